"""world_size-2 test of the row-sharded driver (arrowspace_amd.distributed) on CPU with gloo.

The product's compute backend (HipOps) needs a GPU; here the sharding / exchange / merge logic
is exercised with a stand-in backend defined in THIS file on top of the oracle (tests may use
the oracle as the checker).  What is verified: the sharded build and search reproduce the
single-process oracle results exactly, for equal and for ragged shards.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import arrowspace_amd as A
from arrowspace_amd import distributed as D
from oracle import oracle_ctypes as O
from tests import synth


class _Csr:
    def __init__(self, csr):
        self.csr = csr

    def nnz(self):
        return len(self.csr[1])


class _Gl:
    def __init__(self, csr):
        self.matrix = _Csr(csr)


class _CsrBlock:
    """Row block [lo, lo + count) of a CSR (what as_laplacian_from_knn_rows returns), cut from the oracle's matrix."""

    def __init__(self, csr, lo, count):
        ip, ix, dv = csr
        a, b = int(ip[lo]), int(ip[lo + count])
        self.csr = ((ip[lo: lo + count + 1] - ip[lo]).astype(np.uint64), ix[a:b].copy(), dv[a:b].copy())

    def nnz(self):
        return len(self.csr[1])

    def to_host(self):
        return self.csr


JOIN_SLOTS = 64  # list slots per row of the emulated join


class OracleOps:
    """Test-only stand-in for HipOps (CPU tensors in, CPU tensors out).  With `join=True` it also emulates the sharded
    threshold join in numpy -- thresholds, the pairs a rank contracts (those of the rows it owns against the half of
    the matrix "ahead" of them, listed for BOTH rows of a pair), pack / unpack -- so that the driver's list exchange
    runs on gloo exactly as it does on RCCL; join_finish checks that what arrived is the complete list of every row."""

    def __init__(self, join=False):
        if join:
            self.join_applies = lambda n, d, gp: True
            self.join_capacity = lambda gp: 2 * JOIN_SLOTS

    @staticmethod
    def _gp(gp):
        return O.graph_params(gp.eps, gp.k, gp.topk, gp.p, gp.sigma)

    @staticmethod
    def _cos(X):
        U = X / np.linalg.norm(X, axis=1, keepdims=True)
        return U @ U.T

    def join_thresholds(self, X_full, row_begin, row_count, gp):
        S = self._cos(X_full.numpy())[row_begin: row_begin + row_count]
        kth = np.sort(S, axis=1)[:, -min(S.shape[1], gp.topk + 6)]   # a cut that keeps a few more than topk + 1 per row
        return torch.from_numpy(kth.astype(np.float32))

    def join_pairs(self, X_full, row_begin, row_count, gp, theta_all):
        n = X_full.shape[0]
        S, th = self._cos(X_full.numpy()), theta_all.numpy()
        cnt = np.zeros(n, dtype=np.int32)
        cand = np.zeros((n, 2 * JOIN_SLOTS), dtype=np.int32)

        def put(r, other, score):
            if cnt[r] < JOIN_SLOTS:
                cand[r, 2 * cnt[r]] = other
                cand[r, 2 * cnt[r] + 1] = np.float32(score).view(np.int32)
            cnt[r] += 1
        for i in range(row_begin, row_begin + row_count):
            for off in range(1, n // 2 + 1):        # the pairs this row's owner contracts: the half of the ring ahead
                j = (i + off) % n
                if 2 * off == n and j < i:           # the antipodal pair once, from the smaller index
                    continue
                if S[i, j] >= th[i]:
                    put(i, j, S[i, j])
                if S[i, j] >= th[j]:
                    put(j, i, S[i, j])
        return torch.from_numpy(cnt), torch.from_numpy(cand)

    def join_pack(self, cnt, cand):
        c, a = cnt.numpy(), cand.numpy()
        used = np.minimum(c, JOIN_SLOTS)
        offsets = np.concatenate([[0], np.cumsum(used)]).astype(np.int64)
        packed = np.concatenate([a[r, : 2 * used[r]] for r in range(len(c))] + [np.zeros(0, dtype=np.int32)])
        return torch.from_numpy(offsets), torch.from_numpy(packed.astype(np.int32))

    def join_unpack(self, cnt_src, packed, cap):
        n_src, rows = cnt_src.shape
        out = np.zeros((n_src, rows, cap), dtype=np.int32)
        c, p, at = cnt_src.numpy(), packed.numpy(), 0
        for s in range(n_src):
            for r in range(rows):
                u = 2 * min(int(c[s, r]), cap // 2)
                out[s, r, :u] = p[at: at + u]
                at += u
        assert at == len(p)
        return torch.from_numpy(out)

    def join_finish(self, X_full, row_begin, row_count, gp, theta_all, cnt_src, cand_src):
        S, th = self._cos(X_full.numpy()), theta_all.numpy()
        c, a = cnt_src.numpy(), cand_src.numpy()
        for r in range(row_count):                  # the exchanged lists hold exactly the pairs at or above the cut
            i = row_begin + r
            got = sorted(int(a[s, r, 2 * e]) for s in range(c.shape[0]) for e in range(int(c[s, r])))
            want = [j for j in range(S.shape[0]) if j != i and S[i, j] >= th[i]]
            assert got == want, (i, got, want)
            for s in range(c.shape[0]):
                for e in range(int(c[s, r])):
                    j = int(a[s, r, 2 * e])
                    assert a[s, r, 2 * e + 1] == np.float32(S[i, j]).view(np.int32)
        return self.knn_rows(X_full, row_begin, row_count, gp)

    def laplacian_rows(self, idx_all, dst_all, gp, row_begin, row_count):
        full = O.laplacian_from_knn(idx_all.numpy().astype(np.uint64), dst_all.numpy(), self._gp(gp))
        return _CsrBlock(full, row_begin, row_count)

    def knn_rows(self, X_full, row_begin, row_count, gp):
        idx, dst = O.knn_cosine(X_full.numpy(), self._gp(gp))
        sl = slice(row_begin, row_begin + row_count)
        return torch.from_numpy(idx[sl].astype(np.int64)), torch.from_numpy(dst[sl].copy())

    def laplacian_from_knn(self, idx, dst, gp):
        return _Csr(O.laplacian_from_knn(idx.numpy().astype(np.uint64), dst.numpy(), self._gp(gp)))

    def feature_laplacian(self, centroids, gp, n_items):
        return _Gl(O.build_laplacian(np.ascontiguousarray(np.asarray(centroids).T), self._gp(gp)))

    def lambdas(self, items, gl, taumode):
        return torch.from_numpy(O.taumode_lambdas(items.numpy(), gl.matrix.csr, taumode.kind, taumode.arg))

    def prepare_queries(self, items, queries, gl, taumode):
        return torch.tensor([O.prepare_query_item(q, gl.matrix.csr, taumode.kind, taumode.arg)
                             for q in queries.numpy()], dtype=torch.float64)

    def make_index(self, items, lambdas, taumode, id_offset):
        return (items.numpy(), lambdas.numpy(), int(id_offset))

    def search(self, index, queries, q_lambdas, k, alpha):
        X, lam, off = index
        kk = min(k, len(X))
        idx = np.full((len(queries), kk), -1, dtype=np.int64)
        sc = np.full((len(queries), kk), -np.inf)
        for j, q in enumerate(queries.numpy()):
            i, s = O.search_lambda_aware(X, lam, q, float(q_lambdas[j]), kk, alpha)
            idx[j, : len(i)] = i.astype(np.int64) + off
            sc[j, : len(s)] = s
        return torch.from_numpy(idx), torch.from_numpy(sc)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, d, k, out_dir, join=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X, cent = synth.mixture(n, d, seed=13, n_clusters=8, rank=3)
        lo, hi = D.shard_range(n, rank, world)
        gp = A.GraphParams(0.5, 8, 8, 2.0, 0.25)
        sas = D.ShardedArrowSpace(OracleOps(join), torch.from_numpy(X[lo:hi].copy()), n, gp, A.TauMode.Median)
        gl = sas.build_item_graph()
        assert isinstance(gl, D.ShardedCsr) and (gl.lo, gl.hi) == (lo, hi)
        assert (gl.block is None) == (hi == lo)      # every rank assembled its own row block and nothing else
        csr = gl.to_host()
        assert gl.nnz() == len(csr[1])
        sas.build_lambdas(cent)
        Q = torch.from_numpy(X[[1, n // 2, n - 1]] * 1.02)
        lq = sas.prepare_queries(Q)
        idx, sc = sas.search(Q, lq, k, 0.7)
        lam_all = D.all_gather_rows(sas.lambdas.reshape(-1, 1), sas.sizes).reshape(-1)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), indptr=csr[0], indices=csr[1], data=csr[2],
                 lam=lam_all.numpy(), idx=idx.numpy(), sc=sc.numpy(), lq=lq.numpy(),
                 stats=np.array(sas.lambda_stats), lo=lo, hi=hi)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,k,world,join", [(1024, 5, 2, False), (700, 7, 2, False), (200, 150, 2, False), (600, 5, 3, False),
                                             (1100, 5, 2, True), (1300, 4, 3, True), (600, 6, 3, True)])
def test_sharded_build_and_search_match_single_process(tmp_path, n, k, world, join):
    """Two and three ranks, equal and ragged shards, shards that leave the last ranks WITHOUT rows (n = 600 on three
    ranks: 512 + 88 + 0; n = 200: everything on rank 0) -- every rank still takes part in every collective -- with the
    plain row-range search and with the sharded threshold join (thresholds all-gathered, candidate lists exchanged by
    all_to_all_single with per-pair sizes, the same calls the RCCL run makes)."""
    d = 16
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, d, k, str(tmp_path), join), nprocs=world, join=True)
    X, cent = synth.mixture(n, d, seed=13, n_clusters=8, rank=3)
    gp = O.graph_params(0.5, 8, 8, 2.0, 0.25)
    csr = O.build_laplacian(X, gp)
    csrF = O.build_laplacian(np.ascontiguousarray(cent.T), gp)
    lam = O.taumode_lambdas(X, csrF, O.TAU_MEDIAN)
    Q = X[[1, n // 2, n - 1]] * 1.02
    rs = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    assert int(rs[0]["lo"]) == 0 and int(rs[-1]["hi"]) == n
    assert all(int(a["hi"]) == int(b["lo"]) for a, b in zip(rs, rs[1:]))
    assert all(int(r["lo"]) % 512 == 0 or int(r["lo"]) == n for r in rs)  # shard starts sit on super-tile boundaries (or the shard is empty)
    for r in rs:
        assert np.array_equal(r["indptr"], csr[0]) and np.array_equal(r["indices"], csr[1])
        assert np.array_equal(r["data"], csr[2])
        assert np.array_equal(r["lam"], lam)
        np.testing.assert_allclose(r["stats"], [lam.min(), lam.max(), lam.mean()], rtol=1e-12)
        for j in range(3):
            lq = O.prepare_query_item(Q[j], csrF, O.TAU_MEDIAN)
            assert r["lq"][j] == lq
            oi, osc = O.search_lambda_aware(X, lam, Q[j], lq, k, 0.7)
            assert np.array_equal(r["idx"][j][: len(oi)], oi.astype(np.int64))
            assert np.array_equal(r["sc"][j][: len(oi)], osc)


def _pack_reference(cnt, cand, slots):
    """What as_knn_join_pack produces, in numpy: used slots side by side, offsets in slots."""
    used = np.minimum(cnt, slots)
    offsets = np.concatenate([[0], np.cumsum(used)]).astype(np.int64)
    packed = np.concatenate([cand[r, : 2 * used[r]] for r in range(len(cnt))] + [np.zeros(0, dtype=np.int32)])
    return offsets, packed.astype(np.int32)


def _exchange_worker(rank, world, port, n, slots, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sizes = [D.shard_range(n, r, world)[1] - D.shard_range(n, r, world)[0] for r in range(world)]
        rng = np.random.default_rng(100 + rank)                       # every rank has its own lists for ALL rows
        cnt = rng.integers(0, slots + 3, size=n).astype(np.int32)     # some rows over capacity
        cand = rng.integers(0, 1 << 30, size=(n, 2 * slots)).astype(np.int32)
        offsets, packed = _pack_reference(cnt, cand, slots)
        recv = D.exchange_packed(torch.from_numpy(packed), torch.from_numpy(offsets), sizes)
        cnt_src = D.exchange_row_blocks(torch.from_numpy(cnt), sizes)
        np.savez(os.path.join(out_dir, "x%d.npz" % rank), recv=recv.numpy(), cnt_src=cnt_src.numpy(), cnt=cnt, cand=cand)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,world", [(700, 2), (1300, 3)])
def test_compact_list_exchange(tmp_path, n, world):
    """The candidate lists of the sharded join cross the links as (counts, used slots): every rank must receive,
    source by source and row by row, exactly the used slots of its own rows."""
    slots = 6
    port = _free_port()
    mp.spawn(_exchange_worker, args=(world, port, n, slots, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(tmp_path / ("x%d.npz" % r)) for r in range(world)]
    for g in range(world):
        lo, hi = D.shard_range(n, g, world)
        want, want_cnt = [], []
        for s in range(world):
            cnt, cand = outs[s]["cnt"], outs[s]["cand"]
            want_cnt.append(cnt[lo:hi])
            for r in range(lo, hi):
                want.append(cand[r, : 2 * min(int(cnt[r]), slots)])
        assert np.array_equal(outs[g]["cnt_src"], np.stack(want_cnt))
        assert np.array_equal(outs[g]["recv"], np.concatenate(want + [np.zeros(0, dtype=np.int32)]))


def test_merge_topk_prefers_lower_index_on_ties():
    gi = torch.tensor([[[0, 3]], [[130, 131]]])
    gs = torch.tensor([[[0.9, 0.5]], [[0.9, 0.5]]], dtype=torch.float64)
    idx, sc = D.merge_topk(gi, gs, 3)
    assert idx.tolist() == [[0, 130, 3]] and sc.tolist() == [[0.9, 0.9, 0.5]]


def test_shard_range_covers_everything():
    for n in (1, 127, 128, 129, 1000, 1_000_000, 4_000_000):
        for world in (1, 2, 3, 4, 8):
            spans = [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(lo % 512 == 0 or lo == n for lo, _ in spans)


# ---- the item-graph lambda over row blocks (csrc/sharded.hip, as_compute_taumode_signals_sharded), restated with numpy ----
def _signal_partials(X, indptr, indices, data, lo, hi, n_graph, tail_owner):
    """What one rank sums over ITS row block [lo, hi) of the Laplacian (rows as they are stored: global column ids): a stored
    pair {i, j} once, on the rank that owns its smaller row, and doubled; the diagonal once; the denominator over its own rows
    (and the rows past the graph on `tail_owner`).  Returns (num, edge, sum t^2, den), each one value per signal."""
    S = X.shape[1]
    num, edge, s2, den = np.zeros(S), np.zeros(S), np.zeros(S), np.zeros(S)
    for i in range(lo, hi):
        xi = X[i]
        den += xi * xi
        for e in range(int(indptr[i]), int(indptr[i + 1])):
            j, v = int(indices[e]), data[e]
            if j == i:
                num += xi * v * xi
            elif j > i:
                xj = X[j]
                num += 2.0 * (xi * v * xj)
                w = max(-v, 0.0)
                t = w * (xi - xj) ** 2
                edge += 2.0 * t
                s2 += 2.0 * t * t
    if tail_owner:
        den += (X[n_graph:] ** 2).sum(axis=0)
    return np.stack([num, edge, s2, den])


def _signal_worker(rank, world, port, n, d, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X, _ = synth.mixture(n, d, seed=17, n_clusters=6, rank=3)
        csr = O.build_laplacian(X[: n - 40], O.graph_params(0.5, 8, 8, 2.0, 0.25))   # a graph 40 nodes shorter than the signals
        n_graph = n - 40
        lo, hi = D.shard_range(n_graph, rank, world)
        part = torch.from_numpy(_signal_partials(X, csr[0], csr[1], csr[2], lo, hi, n_graph, rank == 0))
        parts = [torch.empty_like(part) for _ in range(world)]
        dist.all_gather(parts, part)                       # 4 sums per signal and rank
        tot = parts[0].clone()
        for p in parts[1:]:                                 # added in rank order: the same totals on every rank
            tot += p
        num, edge, s2, den = [t.numpy() for t in tot]
        tau = np.array([O.select_tau(X[:, c], O.TAU_MEDIAN) for c in range(d)])   # every rank has the whole columns
        e_raw = np.where(den > 1e-12, num / den, 0.0)
        g = np.where(edge > 0, s2 / (edge * edge), 0.0).clip(0.0, 1.0)
        lam = tau * (e_raw / (e_raw + tau)) + (1.0 - tau) * g
        np.save(os.path.join(out_dir, "sig%d.npy" % rank), lam)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,world", [(1500, 2), (1200, 3)])
def test_item_graph_lambda_sums_shard_by_row_blocks(tmp_path, n, world):
    """SURVEY.md section 8(e), "K7 item-graph signals": partial (num, edge, den) per row block, one exchange of the partial sums,
    the same lambdas as TauMode::compute_synthetic_lambda_csr on one process (src/taumode.rs:552-660) -- the decomposition the
    library's sharded call uses, here on CPU over gloo (the library's own run: tests/test_distributed_gpu.py)."""
    d = 12
    mp.spawn(_signal_worker, args=(world, _free_port(), n, d, str(tmp_path)), nprocs=world, join=True)
    X, _ = synth.mixture(n, d, seed=17, n_clusters=6, rank=3)
    csr = O.build_laplacian(X[: n - 40], O.graph_params(0.5, 8, 8, 2.0, 0.25))
    want = O.taumode_lambdas(np.ascontiguousarray(X.T), csr, O.TAU_MEDIAN)
    lams = [np.load(tmp_path / ("sig%d.npy" % r)) for r in range(world)]
    for lam in lams:
        np.testing.assert_allclose(lam, want, rtol=1e-12, atol=1e-300)
        assert np.array_equal(lam, lams[0])   # bit-identical across the ranks


def test_one_rank_builds_the_item_graph_in_one_call_and_keeps_the_images_for_its_index():
    """Host logic of ShardedArrowSpace on ONE rank (no process group): a backend that offers `build_laplacian` gets the whole item graph
    in one call -- no neighbour table crosses the Python layer -- unless the graph parameters ask for normalised rows (the two-step
    form then, as before), and a backend with a context is told to keep the build's images for the index that follows
    (include/arrowspace_hip.h: "keep_join_images").  A backend without either (the oracle backend above) takes the old path."""
    calls = []

    class FakeCtx:
        def set_option(self, name, value):
            calls.append(("option", name, value))

    class FakeOps(OracleOps):
        def __init__(self):
            super().__init__(False)
            self.ctx = FakeCtx()

        def build_laplacian(self, X_full, gp):
            calls.append(("build_laplacian", tuple(X_full.shape)))
            return "graph"

        def knn_rows(self, X_full, row_begin, row_count, gp):
            calls.append(("knn_rows", row_begin, row_count))
            return super().knn_rows(X_full, row_begin, row_count, gp)

    X, _ = synth.mixture(300, 16, seed=5, n_clusters=4, rank=3)
    gp = A.GraphParams(0.5, 8, 8, 2.0, 0.25)
    sas = D.ShardedArrowSpace(FakeOps(), torch.from_numpy(X), 300, gp, A.TauMode.Median)
    assert sas.build_item_graph() == "graph"
    assert ("option", "keep_join_images", 1) in calls  # (on from the build ... to the creation of the index: build_lambdas sets it back)
    assert ("build_laplacian", (300, 16)) in calls and not any(c[0] == "knn_rows" for c in calls)
    # normalised rows: the one-call form is not taken
    calls.clear()
    gpn = A.GraphParams(0.5, 8, 8, 2.0, 0.25, True, False)
    sas_n = D.ShardedArrowSpace(FakeOps(), torch.from_numpy(X), 300, gpn, A.TauMode.Median)
    sas_n.build_item_graph()
    assert any(c[0] == "knn_rows" for c in calls) and not any(c[0] == "build_laplacian" for c in calls)
    # a backend without a context or a one-call build: nothing is set, the two-step form runs
    plain = D.ShardedArrowSpace(OracleOps(False), torch.from_numpy(X), 300, gp, A.TauMode.Median)
    gl = plain.build_item_graph()
    assert gl is not None
