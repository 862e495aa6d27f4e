"""Row-sharded lambda-tau build and search: one process per GPU, torch.distributed collectives
(backend "nccl" = RCCL over xGMI on the GPU box; "gloo" in the CPU tests).

The path shards by rows.  Exchange steps, nothing else crosses ranks:
  item graph     all-gather of the row shards (the off-diagonal column blocks every rank needs), all-gather of the
                 join thresholds, all-to-all of the candidate lists, all-gather of the per-row neighbour tables
                 (n x (topk+1) x 16 B); the CSR assembly is sharded by rows (every rank assembles its row block from the
                 gathered table: as_laplacian_from_knn_rows), one all-reduce for nnz;
  lambda-tau     none for the values (tau is per item); one small all-reduce for min/max/sum;
  search         all-gather of the per-shard top-k (nq x k pairs) + a stable merge.

`ops` is the compute backend.  The product backend is HipOps (C ABI calls on device tensors);
the CPU tests inject a stand-in defined under tests/ so that the sharding / exchange / merge
logic is exercised with world_size 2 on gloo.  Nothing here imports the oracle.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import api as A
from ._lib import ptr

I64_NONE = -1  # UINT64_MAX seen through int64


def shard_range(n, rank, world):
    """Contiguous row range of `rank`; shard starts are multiples of 512 (the row super tile of the
    symmetric join: every rank's A side reads the shared packed image in place, and the tile pairs
    of the join are owned by whole 512-row super tiles)."""
    per = (n + world - 1) // world
    per = (per + 511) // 512 * 512
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi


class HipOps:
    """Compute backend over libarrowspace_hip.so; all tensors live in HBM."""

    def __init__(self, ctx):
        self.ctx = ctx

    def knn_rows(self, X_full, row_begin, row_count, gp):
        n, d = X_full.shape
        K = gp.topk + 1
        idx = torch.empty((row_count, K), dtype=torch.int64, device=X_full.device)
        dst = torch.empty((row_count, K), dtype=torch.float64, device=X_full.device)
        c = gp.to_c()
        import ctypes as C
        A._adopt_torch_stream(self.ctx)
        self.ctx.check(self.ctx.lib.as_knn_cosine(self.ctx.h, ptr(X_full), n, d, row_begin, row_count, C.byref(c),
                                                  ptr(idx), ptr(dst)))
        return idx, dst

    # -- symmetric threshold join, sharded form (include/arrowspace_hip.h: as_knn_join_*) --
    def join_applies(self, n, d, gp):
        """Does the sharded join apply to an n x d problem?  Depends on n, d, the graph parameters and the context's
        options only, so every rank of a job answers the same (a rank without rows included)."""
        import ctypes as C
        c = gp.to_c()
        return bool(self.ctx.lib.as_knn_join_applies(self.ctx.h, int(n), int(d), C.byref(c)))

    def join_capacity(self, gp):
        import ctypes as C
        c = gp.to_c()
        return int(self.ctx.lib.as_knn_join_capacity(self.ctx.h, C.byref(c)))

    def join_thresholds(self, X_full, row_begin, row_count, gp):
        """Thresholds of this rank's rows, or None when the join does not apply to this problem."""
        import ctypes as C
        from ._lib import AS_ERR_UNSUPPORTED
        n, d = X_full.shape
        theta = torch.empty(row_count, dtype=torch.float32, device=X_full.device)
        c = gp.to_c()
        A._adopt_torch_stream(self.ctx)
        rc = self.ctx.lib.as_knn_join_thresholds(self.ctx.h, ptr(X_full), n, d, row_begin, row_count, C.byref(c), ptr(theta))
        if rc == AS_ERR_UNSUPPORTED:
            return None
        self.ctx.check(rc)
        return theta

    def join_pairs(self, X_full, row_begin, row_count, gp, theta_all):
        import ctypes as C
        n, d = X_full.shape
        c = gp.to_c()
        cap = int(self.ctx.lib.as_knn_join_capacity(self.ctx.h, C.byref(c)))
        cnt = torch.empty(n, dtype=torch.int32, device=X_full.device)
        cand = torch.empty((n, cap), dtype=torch.int32, device=X_full.device)
        A._adopt_torch_stream(self.ctx)
        self.ctx.check(self.ctx.lib.as_knn_join_pairs(self.ctx.h, ptr(X_full), n, d, row_begin, row_count, C.byref(c),
                                                      ptr(theta_all), cap, ptr(cnt), ptr(cand)))
        return cnt, cand

    def join_pack(self, cnt, cand):
        """Used slots of (cnt, cand) side by side + row offsets in slots (as_knn_join_pack)."""
        n, cap = int(cand.shape[0]), int(cand.shape[1])
        offsets = torch.empty(n + 1, dtype=torch.int64, device=cand.device)
        packed = torch.empty(n * cap, dtype=torch.int32, device=cand.device)
        A._adopt_torch_stream(self.ctx)
        self.ctx.check(self.ctx.lib.as_knn_join_pack(self.ctx.h, ptr(cnt), ptr(cand), n, cap, ptr(offsets), ptr(packed)))
        return offsets, packed

    def join_unpack(self, cnt_src, packed, cap):
        """cnt_src [n_src, rows]; packed: the used slots of those n_src x rows lists in that order -> [n_src, rows, cap]."""
        n_src, rows = int(cnt_src.shape[0]), int(cnt_src.shape[1])
        cand_src = torch.empty((n_src, rows, cap), dtype=torch.int32, device=cnt_src.device)
        cnt_c, packed_c = cnt_src.contiguous(), packed.contiguous()  # named: they must outlive the call
        A._adopt_torch_stream(self.ctx)
        self.ctx.check(self.ctx.lib.as_knn_join_unpack(self.ctx.h, ptr(cnt_c), ptr(packed_c), n_src * rows, cap,
                                                       ptr(cand_src)))
        return cand_src

    def join_finish(self, X_full, row_begin, row_count, gp, theta_all, cnt_src, cand_src):
        import ctypes as C
        n, d = X_full.shape
        K = gp.topk + 1
        n_src, cap = int(cand_src.shape[0]), int(cand_src.shape[2])
        idx = torch.empty((row_count, K), dtype=torch.int64, device=X_full.device)
        dst = torch.empty((row_count, K), dtype=torch.float64, device=X_full.device)
        c = gp.to_c()
        cnt_c, cand_c = cnt_src.contiguous(), cand_src.contiguous()  # named: they must outlive the call
        A._adopt_torch_stream(self.ctx)
        self.ctx.check(self.ctx.lib.as_knn_join_finish(self.ctx.h, ptr(X_full), n, d, row_begin, row_count, C.byref(c),
                                                       ptr(theta_all), n_src, cap, ptr(cnt_c), ptr(cand_c), ptr(idx),
                                                       ptr(dst)))
        return idx, dst

    def laplacian_from_knn(self, idx, dist_, gp):
        return A.laplacian_from_knn(idx, dist_, gp, ctx=self.ctx)

    def build_laplacian(self, X_full, gp):
        """The whole item graph in one call (as_build_laplacian): the neighbour table stays in the library's 32-bit form between the
        search and the assembly, the matrix is known to be symmetric (the library's own distances) and carries the order of the
        join for the item-graph lambda."""
        A._adopt_torch_stream(self.ctx)
        return A.build_laplacian_matrix(X_full, gp, ctx=self.ctx).matrix

    def laplacian_rows(self, idx_all, dist_all, gp, row_begin, row_count):
        """This rank's row block of the Laplacian from the all-gathered neighbour table (as_laplacian_from_knn_rows)."""
        A._adopt_torch_stream(self.ctx)
        return A.laplacian_from_knn(idx_all, dist_all, gp, ctx=self.ctx, row_begin=row_begin, row_count=row_count)

    def feature_laplacian(self, centroids, gp, n_items):
        return A.GraphFactory.build_laplacian_matrix_from_k_cluster(
            centroids, gp.eps, gp.k, gp.topk, gp.p, gp.sigma, gp.normalise, gp.sparsity_check, n_items, ctx=self.ctx)

    def lambdas(self, items, gl, taumode):
        asp = A.ArrowSpace(items, taumode, self.ctx)
        asp.compute_taumode(gl)
        return asp.lambdas()

    def prepare_queries(self, items, queries, gl, taumode):
        return A.ArrowSpace(items, taumode, self.ctx).prepare_query_items(queries, gl)

    def make_index(self, items, lambdas, taumode, id_offset):
        asp = A.ArrowSpace(items, taumode, self.ctx)
        asp.update_lambdas(lambdas)
        asp.id_offset = int(id_offset)
        asp._ensure_index()  # as_index_create (residency + exact row norms) belongs to the build, not to the first search
        return asp

    def search(self, index, queries, q_lambdas, k, alpha):
        idx, sc = index.search_lambda_aware_batch(queries, q_lambdas, k, alpha)
        return idx, sc


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _collectives_on(world):
    """Exchange steps run when there is more than one rank -- or, with ARROWSPACE_DIST_FORCE=1, also
    on a single rank with an initialised process group: that is how the RCCL calls (dtypes, stream
    ordering against the HIP kernels) are exercised on a one-GPU test box."""
    import os
    return world > 1 or (os.environ.get("ARROWSPACE_DIST_FORCE") == "1" and dist.is_available() and dist.is_initialized())


def _host_staged():
    """gloo cannot move HBM tensors: stage through the host (test configurations only; the bench
    runs on "nccl" = RCCL, device to device over xGMI)."""
    return dist.get_backend() == "gloo"


def _all_gather_into(out, inp):
    if inp.is_cuda and _host_staged():
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu().contiguous())
        out.copy_(o)
    else:
        dist.all_gather_into_tensor(out, inp.contiguous())


def _all_reduce(t, op):
    if t.is_cuda and _host_staged():
        c = t.cpu()
        dist.all_reduce(c, op=op)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op)


def all_gather_rows(t, sizes):
    """all-gather of row blocks with (possibly) unequal row counts; `sizes` = rows per rank."""
    rank, world = _world()
    if not _collectives_on(world):
        return t
    mx = max(sizes)
    pad = t
    if t.shape[0] < mx:
        pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
    out = torch.empty((world * mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    _all_gather_into(out, pad)
    if all(s == mx for s in sizes):
        return out
    return torch.cat([out[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)


def _all_to_all(out, inp, out_splits, in_splits):
    """all_to_all_single with per-pair split sizes.  ONE code path for every backend: RCCL moves HBM tensors as they
    are; gloo (the CPU tests, and the tests in which several ranks share one GPU) cannot, so device tensors are staged
    through the host around the very same call."""
    if inp.is_cuda and _host_staged():
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu().contiguous(), output_split_sizes=out_splits, input_split_sizes=in_splits)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp.contiguous(), output_split_sizes=out_splits, input_split_sizes=in_splits)


def exchange_row_blocks(t, sizes):
    """t: [n, ...] indexed by GLOBAL row on every rank.  Returns [world, sizes[rank], ...]: from every rank
    the slice of this rank's rows (all-to-all; RCCL over xGMI on the GPU box)."""
    rank, world = _world()
    starts = [sum(sizes[:r]) for r in range(world)]
    mine = sizes[rank]
    if not _collectives_on(world):
        return t[starts[rank]: starts[rank] + mine].unsqueeze(0)
    tail = tuple(t.shape[1:])
    out = torch.empty((world * mine,) + tail, dtype=t.dtype, device=t.device)
    _all_to_all(out, t, [mine] * world, list(sizes))
    return out.view((world, mine) + tail)


def exchange_packed(packed, row_offsets, sizes):
    """Compact form of the list exchange: `packed` = the used slots (2 words each) of all n rows in row order,
    row_offsets [n + 1] in slots.  Every rank receives, source by source, the slots of its own rows
    (all-to-all with per-pair sizes; RCCL over xGMI on the GPU box).  Returns one int32 tensor."""
    rank, world = _world()
    starts = [sum(sizes[:r]) for r in range(world)]
    bounds = row_offsets[torch.tensor(starts + [starts[-1] + sizes[-1]], device=row_offsets.device)].cpu().tolist()
    send = [2 * (bounds[g + 1] - bounds[g]) for g in range(world)]           # words per destination
    if not _collectives_on(world):
        return packed[2 * bounds[rank]: 2 * bounds[rank + 1]]
    send_t = torch.tensor(send, dtype=torch.int64, device=packed.device)
    recv_t = torch.empty(world, dtype=torch.int64, device=packed.device)
    _all_to_all(recv_t, send_t, [1] * world, [1] * world)
    recv = [int(v) for v in recv_t.cpu().tolist()]
    out = torch.empty(sum(recv), dtype=packed.dtype, device=packed.device)
    _all_to_all(out, packed[: 2 * bounds[world]], recv, send)
    return out


class ShardedCsr:
    """The item-graph Laplacian of a row-sharded build: this rank's row block (CSR with global column ids, or None on
    a rank without rows) and the global entry count."""

    def __init__(self, block, lo, hi, n, nnz_total):
        self.block, self.lo, self.hi, self.n, self._nnz = block, lo, hi, n, int(nnz_total)

    def nnz(self):
        return self._nnz

    def shape(self):
        return (self.n, self.n)

    def to_host(self):
        """The WHOLE matrix on every rank (tests, export): all-gather of the row blocks."""
        rank, world = _world()
        if self.block is None:
            ip, ix, dv = np.zeros(1, dtype=np.uint64), np.zeros(0, dtype=np.uint64), np.zeros(0)
        else:
            ip, ix, dv = self.block.to_host()
        if not _collectives_on(world):
            return ip, ix, dv
        parts = [None] * world
        dist.all_gather_object(parts, (ip, ix, dv))
        indptr, base = [np.zeros(1, dtype=np.uint64)], 0
        for p_ip, _, _ in parts:
            indptr.append(p_ip[1:] + np.uint64(base))
            base += int(p_ip[-1])
        return (np.concatenate(indptr), np.concatenate([p[1] for p in parts]), np.concatenate([p[2] for p in parts]))


class ShardedArrowSpace:
    """The row-sharded counterpart of (ArrowSpace, GraphLaplacian) for one rank."""

    def __init__(self, ops, X_shard, n_total, gp, taumode=A.TauMode.Median, libcomm=None):
        """libcomm (a LibComm): the exchange steps of the build, of the lambda statistics and of the search run INSIDE the
        library (csrc/sharded.hip: as_build_laplacian_sharded, as_lambda_stats_sharded, as_search_lambda_aware_sharded) over its
        communicator -- the sharded form of the one-GPU algorithm (row order + int8 first pass + masked f16 pass, every rank on
        its own range of image positions).  Without it the same steps run here on torch.distributed with the stage-level entry
        points (the join in the order of the caller's rows)."""
        self.ops, self.gp, self.taumode, self.libcomm = ops, gp, taumode, libcomm
        self.rank, self.world = _world()
        self.n = int(n_total)
        self.lo, self.hi = shard_range(self.n, self.rank, self.world)
        assert X_shard.shape[0] == self.hi - self.lo, (X_shard.shape, self.lo, self.hi)
        self.X = X_shard
        self.sizes = [shard_range(self.n, r, self.world)[1] - shard_range(self.n, r, self.world)[0]
                      for r in range(self.world)]
        self.lambdas = None
        self.index = None
        # one rank: the index built behind the item graph is over the same rows, untouched in between -- the build keeps its row order
        # and images for it (include/arrowspace_hip.h: "keep_join_images"), and the first search runs on them.  The option is on from
        # the build to the creation of the index only (build_item_graph / build_lambdas below).
        self._keep_images = self.world == 1 and libcomm is None and hasattr(ops, "ctx")

    # -- item graph: n x n Laplacian over the items (build_laplacian_matrix on untransposed rows)
    def build_item_graph(self):
        """Every rank takes part in every collective, with zero-length buffers when it owns no rows (shards are whole
        512-row super tiles, so small inputs leave the last ranks empty); which path runs is decided from
        rank-invariant inputs only."""
        ops, gp, n = self.ops, self.gp, self.n
        if self.libcomm is not None:
            return self.libcomm.build_laplacian(self.X, n, gp)
        if self._keep_images:
            ops.ctx.set_option("keep_join_images", 1)
        if not _collectives_on(self.world) and hasattr(ops, "build_laplacian") and not getattr(gp, "normalise", False):
            return ops.build_laplacian(self.X, gp)                        # one rank: one call
        X_full = all_gather_rows(self.X, self.sizes)                      # RCCL all-gather #1
        rows_here = self.hi - self.lo
        K = gp.topk + 1
        dev = self.X.device
        d = int(self.X.shape[1])
        use_join = _collectives_on(self.world) and hasattr(ops, "join_applies") and ops.join_applies(n, d, gp)
        if use_join:
            # symmetric threshold join: every pair of 512-row super tiles is contracted once, on the rank that owns
            # one of the two; thresholds are all-gathered, candidate lists go to the owners of their rows
            theta = ops.join_thresholds(X_full, self.lo, rows_here, gp) if rows_here else \
                torch.empty(0, dtype=torch.float32, device=dev)
            theta_all = all_gather_rows(theta.reshape(-1, 1), self.sizes).reshape(-1).contiguous()
            cap = ops.join_capacity(gp)
            if rows_here:
                cnt, cand = ops.join_pairs(X_full, self.lo, rows_here, gp, theta_all)
                offsets, packed = ops.join_pack(cnt, cand)
            else:  # nothing contracted here: empty lists for every row
                cnt = torch.zeros(n, dtype=torch.int32, device=dev)
                offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
                packed = torch.empty(0, dtype=torch.int32, device=dev)
            cnt_src = exchange_row_blocks(cnt, self.sizes)                # RCCL all-to-all
            recv = exchange_packed(packed, offsets, self.sizes)           # the used slots only cross the links
            if rows_here:
                cand_src = ops.join_unpack(cnt_src, recv, cap)
                idx, dst = ops.join_finish(X_full, self.lo, rows_here, gp, theta_all, cnt_src, cand_src)
        elif rows_here:
            idx, dst = ops.knn_rows(X_full, self.lo, rows_here, gp)
        if not rows_here:
            idx = torch.empty((0, K), dtype=torch.int64, device=dev)
            dst = torch.empty((0, K), dtype=torch.float64, device=dev)
        idx_all = all_gather_rows(idx, self.sizes)                        # RCCL all-gather #2
        dst_all = all_gather_rows(dst, self.sizes)
        if not _collectives_on(self.world) or not hasattr(ops, "laplacian_rows"):
            return ops.laplacian_from_knn(idx_all, dst_all, gp)           # one rank: the whole matrix
        # K3-K5 sharded by rows: the degree / weight passes over the table are replicated (cheap), the reverse-edge
        # scatter, the unions and the CSR rows are done for this rank's rows only; nnz by all-reduce
        block = ops.laplacian_rows(idx_all, dst_all, gp, self.lo, rows_here) if rows_here else None
        nnz = torch.tensor([block.nnz() if block is not None else 0], dtype=torch.int64, device=dev)
        _all_reduce(nnz, dist.ReduceOp.SUM)
        return ShardedCsr(block, self.lo, self.hi, n, int(nnz))

    # -- reference mapping: F x F graph over the supplied centroids, per-item lambda on the shard
    def build_lambdas(self, centroids):
        self.gl = self.ops.feature_laplacian(centroids, self.gp, self.n)
        if self.hi > self.lo:
            self.lambdas = self.ops.lambdas(self.X, self.gl, self.taumode)
        else:  # a rank without rows: nothing to compute, but it still takes part in the collectives below
            self.lambdas = torch.empty(0, dtype=torch.float64, device=self.X.device)
        lam = self.lambdas if torch.is_tensor(self.lambdas) else torch.from_numpy(np.asarray(self.lambdas))
        if self.libcomm is not None:                                      # logged stats, builder.rs:434-445, inside the library
            self.lambda_stats = self.libcomm.lambda_stats(lam)
            self.index = self.ops.make_index(self.X, self.lambdas, self.taumode, self.lo) if self.hi > self.lo else None
            return self.gl
        stats = torch.stack([lam.min(), -lam.max(), lam.sum()]) if lam.numel() else \
            torch.tensor([float("inf"), float("inf"), 0.0], dtype=torch.float64, device=lam.device)
        if _collectives_on(self.world):                                   # logged stats, builder.rs:434-445
            mm = stats[:2].clone()
            _all_reduce(mm, dist.ReduceOp.MIN)
            sm = stats[2:].clone()
            _all_reduce(sm, dist.ReduceOp.SUM)
            stats = torch.cat([mm, sm])
        self.lambda_stats = (float(stats[0]), float(-stats[1]), float(stats[2]) / self.n)
        self.index = self.ops.make_index(self.X, self.lambdas, self.taumode, self.lo) if self.hi > self.lo else None
        if self._keep_images:
            self.ops.ctx.set_option("keep_join_images", 0)               # (adopted by the index, or released)
        return self.gl

    def prepare_queries(self, queries):
        return self.ops.prepare_queries(self.X, queries, self.gl, self.taumode)

    def search(self, queries, q_lambdas, k, alpha):
        """Global top-k of every query: per-shard top-k (global ids), all-gather, stable merge."""
        if self.libcomm is not None:
            return self.libcomm.search(self.index, queries, q_lambdas, min(k, self.n), alpha, self.lo)
        if self.index is not None:
            idx, sc = self.ops.search(self.index, queries, q_lambdas, k, alpha)
        else:  # no items here: an empty table (padded below to the width of the others)
            nq0 = int(queries.shape[0])
            idx = torch.empty((nq0, 0), dtype=torch.int64, device=queries.device)
            sc = torch.empty((nq0, 0), dtype=torch.float64, device=queries.device)
        idx = idx if torch.is_tensor(idx) else torch.from_numpy(np.asarray(idx).astype(np.int64))
        sc = sc if torch.is_tensor(sc) else torch.from_numpy(np.asarray(sc))
        if not _collectives_on(self.world):
            return idx, sc
        nq, kk = idx.shape
        kmax = torch.tensor([kk], dtype=torch.int64, device=idx.device)
        _all_reduce(kmax, dist.ReduceOp.MAX)
        kmax = int(kmax)
        if kk < kmax:  # a shard smaller than k returns fewer columns
            idx = torch.cat([idx, torch.full((nq, kmax - kk), I64_NONE, dtype=idx.dtype, device=idx.device)], 1)
            sc = torch.cat([sc, torch.full((nq, kmax - kk), float("-inf"), dtype=sc.dtype, device=sc.device)], 1)
        gi = torch.empty((self.world * nq, kmax), dtype=idx.dtype, device=idx.device)
        gs = torch.empty((self.world * nq, kmax), dtype=sc.dtype, device=sc.device)
        _all_gather_into(gi, idx)                                         # RCCL all-gather #3
        _all_gather_into(gs, sc)
        return merge_topk(gi.view(self.world, nq, kmax), gs.view(self.world, nq, kmax), min(k, self.n))


def merge_topk(gi, gs, k):
    """gi/gs: world x nq x kk per-shard results, shards in ascending id order, each sorted by
    (score desc, id asc).  A stable descending sort of the rank-major concatenation keeps the
    lower id first among equal scores (src/core.rs:785 stable sort_by)."""
    world, nq, kk = gi.shape
    ci = gi.permute(1, 0, 2).reshape(nq, world * kk)
    cs = gs.permute(1, 0, 2).reshape(nq, world * kk)
    cs = torch.where(ci == I64_NONE, torch.full_like(cs, float("-inf")), cs)
    order = torch.sort(cs, dim=1, descending=True, stable=True).indices[:, :k]
    return torch.gather(ci, 1, order), torch.gather(cs, 1, order)


# ---- the same build through the library's own communicator (include/arrowspace_hip.h: as_comm_*) ----------------------
class LibComm:
    """as_comm handle: the sharded build runs inside the library (csrc/sharded.hip), the host only supplies a transport.
    `rccl`: RCCL over xGMI, bootstrapped with a 128-byte id (rank 0: LibComm.unique_id(), shipped to the others by the
    host -- here through torch.distributed).  `torch_transport`: the three primitives as callbacks on top of
    torch.distributed with host staging: what lets several ranks share one GPU in the tests, where RCCL cannot run."""

    def __init__(self, ctx, handle, keep=None):
        self.ctx, self.h, self._keep = ctx, handle, keep

    @staticmethod
    def unique_id(ctx):
        import ctypes as C
        buf = C.create_string_buffer(128)
        ctx.check(ctx.lib.as_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def rccl(cls, ctx, unique_id, rank, world):
        import ctypes as C
        h = C.c_void_p()
        ctx.check(ctx.lib.as_comm_create_rccl(ctx.h, C.c_char_p(unique_id), rank, world, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def torch_transport(cls, ctx):
        """Callbacks over torch.distributed (any backend), device buffers staged through the host."""
        import ctypes as C
        from . import _lib as L
        rank, world = _world()
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        hip.hipStreamSynchronize.argtypes = [C.c_void_p]

        def d2h(src, nbytes):
            out = np.empty(int(nbytes), dtype=np.uint8)
            if nbytes:
                assert hip.hipMemcpy(out.ctypes.data, src, int(nbytes), 2) == 0
            return out

        def h2d(dst, arr):
            if arr.size:
                arr = np.ascontiguousarray(arr)
                assert hip.hipMemcpy(dst, arr.ctypes.data, arr.size, 1) == 0

        def all_gather_v(user, send, send_bytes, recv, recv_bytes, recv_offsets, stream):
            try:
                hip.hipStreamSynchronize(stream)
                parts = [None] * world
                dist.all_gather_object(parts, d2h(send, send_bytes))
                for r in range(world):
                    assert parts[r].size == recv_bytes[r]
                    h2d((recv or 0) + recv_offsets[r], parts[r])
                return 0
            except Exception:  # noqa: BLE001 -- a Python exception must not cross the C frame
                return 9

        def all_to_all_v(user, send, send_bytes, send_offsets, recv, recv_bytes, recv_offsets, stream):
            try:
                hip.hipStreamSynchronize(stream)
                outgoing = [d2h((send or 0) + send_offsets[p], send_bytes[p]) for p in range(world)]
                table = [None] * world
                dist.all_gather_object(table, outgoing)  # (test transport: everything everywhere, then pick)
                for p in range(world):
                    part = table[p][rank]
                    assert part.size == recv_bytes[p]
                    h2d((recv or 0) + recv_offsets[p], part)
                return 0
            except Exception:  # noqa: BLE001
                return 9

        def all_reduce_sum_u64(user, buf, count, stream):
            try:
                hip.hipStreamSynchronize(stream)
                mine = d2h(buf, 8 * count).view(np.uint64)
                parts = [None] * world
                dist.all_gather_object(parts, mine)
                h2d(buf, np.sum(np.stack(parts), axis=0, dtype=np.uint64).view(np.uint8))
                return 0
            except Exception:  # noqa: BLE001
                return 9

        cbs = (L.AllGatherV(all_gather_v), L.AllToAllV(all_to_all_v), L.AllReduceSumU64(all_reduce_sum_u64))
        ops = L.CommOpsC(None, *cbs)
        h = C.c_void_p()
        ctx.check(ctx.lib.as_comm_create_custom(ctx.h, C.byref(ops), rank, world, C.byref(h)))
        return cls(ctx, h, keep=(cbs, ops, hip))

    def rank(self):
        return int(self.ctx.lib.as_comm_rank(self.h))

    def world(self):
        return int(self.ctx.lib.as_comm_world(self.h))

    def shard_range(self, n, rank=None):
        import ctypes as C
        lo, hi = C.c_uint64(), C.c_uint64()
        self.ctx.lib.as_shard_range(int(n), self.rank() if rank is None else rank, self.world(), C.byref(lo), C.byref(hi))
        return int(lo.value), int(hi.value)

    def build_laplacian(self, X_shard, n_total, gp):
        """as_build_laplacian_sharded: this rank's row block (ShardedCsr; block None when the rank owns no rows)."""
        import ctypes as C
        ctx = self.ctx
        lo, hi = self.shard_range(n_total)
        assert X_shard.is_cuda and X_shard.is_contiguous() and X_shard.shape[0] == hi - lo
        c = gp.to_c()
        h, nnz = C.c_void_p(), C.c_uint64()
        A._adopt_torch_stream(ctx)
        ctx.check(ctx.lib.as_build_laplacian_sharded(ctx.h, self.h, ptr(X_shard) if hi > lo else None, int(n_total),
                                                     int(X_shard.shape[1]), C.byref(c), C.byref(h), C.byref(nnz)))
        block = A.CsrMatrix(ctx, h) if h.value else None
        return ShardedCsr(block, lo, hi, int(n_total), int(nnz.value))

    def lambda_stats(self, lambdas_shard):
        """as_lambda_stats_sharded: (min, max, mean) of all ranks' lambdas (src/builder.rs:434-445), the same on every rank."""
        import ctypes as C
        ctx = self.ctx
        mn, mx, mean = C.c_double(), C.c_double(), C.c_double()
        lam = lambdas_shard.contiguous() if torch.is_tensor(lambdas_shard) else np.ascontiguousarray(lambdas_shard, dtype=np.float64)
        n_here = int(lam.numel() if torch.is_tensor(lam) else lam.size)
        A._adopt_torch_stream(ctx)
        ctx.check(ctx.lib.as_lambda_stats_sharded(ctx.h, self.h, ptr(lam) if n_here else None, n_here, C.byref(mn), C.byref(mx), C.byref(mean)))
        return (mn.value, mx.value, mean.value)

    def search(self, index, queries, q_lambdas, k, alpha, id_offset):
        """as_search_lambda_aware_sharded: index = this rank's ArrowSpace (None on a rank without rows); queries / q_lambdas
        replicated.  Returns the global (idx, scores), nq x k, on every rank (device tensors when the queries are)."""
        ctx = self.ctx
        nq = int(queries.shape[0])
        if torch.is_tensor(queries):
            Q, LQ = queries.contiguous(), q_lambdas.contiguous()
            idx = torch.empty((nq, k), dtype=torch.int64, device=Q.device)
            sc = torch.empty((nq, k), dtype=torch.float64, device=Q.device)
        else:
            Q, LQ = np.ascontiguousarray(queries, dtype=np.float64), np.ascontiguousarray(q_lambdas, dtype=np.float64)
            idx, sc = np.empty((nq, k), dtype=np.int64), np.empty((nq, k), dtype=np.float64)
        h = None
        if index is not None:
            index._ensure_index()
            h = index._index
        A._adopt_torch_stream(ctx)
        ctx.check(ctx.lib.as_search_lambda_aware_sharded(ctx.h, self.h, h, ptr(Q), ptr(LQ), nq, int(k), float(alpha), int(id_offset),
                                                         ptr(idx), ptr(sc)))
        return idx, sc

    def signal_lambdas(self, X_shard, n_total, block, taumode=None):
        """as_compute_taumode_signals_sharded: the lambdas of the feature columns as signals over the row-sharded item graph
        (block = this rank's CsrMatrix row block, None on a rank without rows).  The same F values on every rank."""
        ctx = self.ctx
        taumode = taumode or A.TauMode.Median
        f = int(X_shard.shape[1])
        lo, hi = self.shard_range(n_total)
        Xs = A._f64(X_shard, ctx)   # float64 or TypeError: a float32 shard must not be read as doubles
        if torch.is_tensor(Xs):
            Xs = Xs.contiguous()
            out = torch.empty(f, dtype=torch.float64, device=Xs.device)
        else:
            out = np.empty(f, dtype=np.float64)
        A._adopt_torch_stream(ctx)
        ctx.check(ctx.lib.as_compute_taumode_signals_sharded(ctx.h, self.h, ptr(Xs) if hi > lo else None, int(n_total), f,
                                                             block.h if block is not None else None, taumode.kind, taumode.arg, ptr(out)))
        return out

    def close(self):
        if self.h:
            self.ctx.lib.as_comm_destroy(self.h)
            self.h = None
