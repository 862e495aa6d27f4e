"""Two ranks sharing the ONE GPU of the test box drive the product backend (HipOps, C ABI) through
the row-sharded driver; collectives run on gloo with host staging (RCCL needs one GPU per rank).
What this covers that the CPU gloo test cannot: as_knn_cosine on a row range with row_begin != 0,
as_laplacian_from_knn on gathered device tables, sharded lambda + index with id_offset, the merge."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, d, out_dir):
    import arrowspace_amd as A
    from arrowspace_amd import distributed as D
    from tests import synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        ctx = A.Context(0)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx._stream = torch.cuda.current_stream().cuda_stream
        ctx.set_option("join_min_rows", 512)  # the symmetric threshold join, sharded form, at test sizes
        X, cent = synth.mixture(n, d, seed=13, n_clusters=16, rank=4)
        lo, hi = D.shard_range(n, rank, world)
        gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
        sas = D.ShardedArrowSpace(D.HipOps(ctx), torch.from_numpy(X[lo:hi].copy()).cuda(), n, gp, A.TauMode.Median)
        gl = sas.build_item_graph()
        assert isinstance(gl, D.ShardedCsr) and (gl.block is None) == (hi == lo)
        csr = gl.to_host()
        sas.build_lambdas(cent)
        Q = torch.from_numpy(X[[1, n // 2, n - 1, 77]] * 1.02).cuda()
        lq = sas.prepare_queries(Q)
        idx, sc = sas.search(Q, lq, 10, 0.7)
        lam_all = D.all_gather_rows(sas.lambdas.reshape(-1, 1), sas.sizes).reshape(-1)
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), indptr=csr[0], indices=csr[1], data=csr[2],
                 lam=lam_all.cpu().numpy(), idx=idx.cpu().numpy(), sc=sc.cpu().numpy(), lq=lq.cpu().numpy(),
                 stats=np.array(sas.lambda_stats), join=np.array(ctx.stat("knn_join") if hi > lo else 1))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,world,d", [(3000, 2, 64), (1111, 2, 64), (3000, 3, 64), (1000, 3, 64), (3000, 2, 200)])
def test_ranks_sharing_one_gpu_match_single_process(tmp_path, n, world, d):
    """2 ranks; 2 ranks with a short second shard; 3 ranks with uneven shards (1024 / 1024 / 952 rows); 3 ranks of
    which the last owns NO rows (512 / 488 / 0) and still takes part in every collective; rows wider than 128 (the
    panelled f16 join over a rank's own super tiles)."""
    import arrowspace_amd as A
    from oracle import oracle_ctypes as O
    from tests import synth
    mp.spawn(_worker, args=(world, _free_port(), n, d, str(tmp_path)), nprocs=world, join=True)
    X, cent = synth.mixture(n, d, seed=13, n_clusters=16, rank=4)
    gp = O.graph_params(0.5, 16, 16, 2.0, 0.25)
    csr = O.build_laplacian(X, gp)
    csrF = O.build_laplacian(np.ascontiguousarray(cent.T), gp)
    lam = O.taumode_lambdas(X, csrF, O.TAU_MEDIAN)
    Q = X[[1, n // 2, n - 1, 77]] * 1.02
    for r in [np.load(tmp_path / ("rank%d.npz" % k)) for k in range(world)]:
        assert int(r["join"]) == 1  # thresholds all-gathered, candidate lists exchanged, own rows finished
        assert np.array_equal(r["indptr"], csr[0]) and np.array_equal(r["indices"], csr[1])
        np.testing.assert_allclose(r["data"], csr[2], rtol=1e-13)
        np.testing.assert_allclose(r["lam"], lam, rtol=1e-12, atol=1e-300)
        np.testing.assert_allclose(r["stats"], [lam.min(), lam.max(), lam.mean()], rtol=1e-10)
        for j in range(4):
            # search ran on the GPU lambdas: compare against the oracle on the same lambdas
            oi, osc = O.search_lambda_aware(X, r["lam"], Q[j], float(r["lq"][j]), 10, 0.7)
            assert np.array_equal(r["idx"][j], oi.astype(np.int64))
            assert np.array_equal(r["sc"][j], osc)


def _rccl_worker(rank, world, port, n, d, out_dir):
    import arrowspace_amd as A
    from arrowspace_amd import distributed as D
    from tests import synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ARROWSPACE_DIST_FORCE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        ctx = A.Context(0)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx._stream = torch.cuda.current_stream().cuda_stream
        ctx.set_option("join_min_rows", 512)
        X, cent = synth.mixture(n, d, seed=13, n_clusters=16, rank=4)
        gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
        sas = D.ShardedArrowSpace(D.HipOps(ctx), torch.from_numpy(X).cuda(), n, gp, A.TauMode.Median)
        for _ in range(3):  # repeated: collectives and kernels interleave on the streams
            csr = sas.build_item_graph().to_host()
            sas.build_lambdas(cent)
            Q = torch.from_numpy(X[[1, n // 2, n - 1, 77]] * 1.02).cuda()
            lq = sas.prepare_queries(Q)
            idx, sc = sas.search(Q, lq, 10, 0.7)
        dist.barrier()
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, "rccl.npz"), indptr=csr[0], indices=csr[1], data=csr[2],
                 lam=sas.lambdas.cpu().numpy(), idx=idx.cpu().numpy(), sc=sc.cpu().numpy(), lq=lq.cpu().numpy(),
                 stats=np.array(sas.lambda_stats))
    finally:
        dist.destroy_process_group()


def test_one_rank_rccl_collectives_match_oracle(tmp_path):
    """backend "nccl" (= RCCL) with ONE rank and the exchange steps forced on: every collective of the sharded
    driver (all-gather of f64 row blocks, i64 / f64 neighbour tables and top-k pairs, min/sum all-reduce,
    barrier) runs through RCCL on device tensors, ordered against the HIP kernels on the context's stream."""
    from oracle import oracle_ctypes as O
    from tests import synth
    n, d = 5000, 64
    mp.spawn(_rccl_worker, args=(1, _free_port(), n, d, str(tmp_path)), nprocs=1, join=True)
    X, cent = synth.mixture(n, d, seed=13, n_clusters=16, rank=4)
    gp = O.graph_params(0.5, 16, 16, 2.0, 0.25)
    csr = O.build_laplacian(X, gp)
    lam = O.taumode_lambdas(X, O.build_laplacian(np.ascontiguousarray(cent.T), gp), O.TAU_MEDIAN)
    r = np.load(tmp_path / "rccl.npz")
    assert np.array_equal(r["indptr"], csr[0]) and np.array_equal(r["indices"], csr[1])
    np.testing.assert_allclose(r["data"], csr[2], rtol=1e-13)
    np.testing.assert_allclose(r["lam"], lam, rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(r["stats"], [lam.min(), lam.max(), lam.mean()], rtol=1e-10)
    Q = X[[1, n // 2, n - 1, 77]] * 1.02
    for j in range(4):
        oi, osc = O.search_lambda_aware(X, r["lam"], Q[j], float(r["lq"][j]), 10, 0.7)
        assert np.array_equal(r["idx"][j], oi.astype(np.int64))
        assert np.array_equal(r["sc"][j], osc)


def _rccl_multi_worker(rank, world, port, n, d, out_dir):
    """One GPU per rank, backend "nccl": the path bench.py takes under torch.distributed.run."""
    import arrowspace_amd as A
    from arrowspace_amd import distributed as D
    from tests import synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        ctx = A.Context(rank)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx._stream = torch.cuda.current_stream().cuda_stream
        ctx.set_option("join_min_rows", 512)
        X, cent = synth.mixture(n, d, seed=13, n_clusters=16, rank=4)
        lo, hi = D.shard_range(n, rank, world)
        gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
        sas = D.ShardedArrowSpace(D.HipOps(ctx), torch.from_numpy(X[lo:hi].copy()).cuda(), n, gp, A.TauMode.Median)
        for _ in range(2):
            gl = sas.build_item_graph()
            sas.build_lambdas(cent)
            Q = torch.from_numpy(X[[1, n // 2, n - 1, 77]] * 1.02).cuda()
            lq = sas.prepare_queries(Q)
            idx, sc = sas.search(Q, lq, 10, 0.7)
        csr = gl.to_host()
        lam_all = D.all_gather_rows(sas.lambdas.reshape(-1, 1), sas.sizes).reshape(-1)
        dist.barrier()
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), indptr=csr[0], indices=csr[1], data=csr[2],
                 lam=lam_all.cpu().numpy(), idx=idx.cpu().numpy(), sc=sc.cpu().numpy(), lq=lq.cpu().numpy(),
                 stats=np.array(sas.lambda_stats), join=np.array(ctx.stat("knn_join") if hi > lo else 1), nnz=np.array(gl.nnz()))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs at least two GPUs (RCCL wants one GPU per rank)")
@pytest.mark.parametrize("n", [20_000, 3_000])
def test_multi_rank_rccl_matches_oracle(tmp_path, n):
    """world = min(device_count, 8) ranks, one GPU each, over RCCL: the all_to_all_single calls with per-pair split
    sizes (thresholds, counts, packed candidate lists), the all-gathers and the sharded CSR assembly of the build the
    driver's scaling run times.  n = 3000 leaves the last ranks of an 8-rank job without rows.  Skipped on the one-GPU
    test box, where the same calls run on gloo with several ranks (test_ranks_sharing_one_gpu_*) and on RCCL with one."""
    from oracle import oracle_ctypes as O
    from tests import synth
    world, d = min(torch.cuda.device_count(), 8), 64
    mp.spawn(_rccl_multi_worker, args=(world, _free_port(), n, d, str(tmp_path)), nprocs=world, join=True)
    X, cent = synth.mixture(n, d, seed=13, n_clusters=16, rank=4)
    gp = O.graph_params(0.5, 16, 16, 2.0, 0.25)
    csr = O.build_laplacian(X, gp)
    lam = O.taumode_lambdas(X, O.build_laplacian(np.ascontiguousarray(cent.T), gp), O.TAU_MEDIAN)
    Q = X[[1, n // 2, n - 1, 77]] * 1.02
    for r in [np.load(tmp_path / ("rank%d.npz" % k)) for k in range(world)]:
        assert int(r["join"]) == 1 or n < 512 * world
        assert int(r["nnz"]) == len(csr[1])
        assert np.array_equal(r["indptr"], csr[0]) and np.array_equal(r["indices"], csr[1])
        np.testing.assert_allclose(r["data"], csr[2], rtol=1e-13)
        np.testing.assert_allclose(r["lam"], lam, rtol=1e-12, atol=1e-300)
        for j in range(4):
            oi, osc = O.search_lambda_aware(X, r["lam"], Q[j], float(r["lq"][j]), 10, 0.7)
            assert np.array_equal(r["idx"][j], oi.astype(np.int64))
            assert np.array_equal(r["sc"][j], osc)


# ---- the same build inside the library: as_comm_* + as_build_laplacian_sharded (csrc/sharded.hip) ----------------------
def _lib_worker(rank, world, port, n, d, out_dir):
    import arrowspace_amd as A
    from arrowspace_amd import distributed as D
    from tests import synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        ctx = A.Context(0)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx._stream = torch.cuda.current_stream().cuda_stream
        ctx.set_option("join_min_rows", 512)
        comm = D.LibComm.torch_transport(ctx)  # the library's exchange sequence over callbacks (ranks share the GPU)
        assert (comm.rank(), comm.world()) == (rank, world)
        lo, hi = comm.shard_range(n)
        assert (lo, hi) == D.shard_range(n, rank, world)
        X, cent = synth.mixture(n, d, seed=13, n_clusters=16, rank=4)
        gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
        # the whole sharded path inside the library: build, lambda statistics, search + merge (no torch collective on the data path)
        sas = D.ShardedArrowSpace(D.HipOps(ctx), torch.from_numpy(X[lo:hi].copy()).cuda(), n, gp, A.TauMode.Median, libcomm=comm)
        gl = sas.build_item_graph()
        join = ctx.stat("sharded_join")
        assert (gl.block is None) == (hi == lo)
        if gl.block is not None:
            assert gl.block.shape() == (hi - lo, n) and gl.block.row_begin() == lo
        csr = gl.to_host()
        sas.build_lambdas(cent)
        Q = torch.from_numpy(X[[1, n // 2, n - 1, 77]] * 1.02).cuda()
        lq = sas.prepare_queries(Q)
        idx, sc = sas.search(Q, lq, 10, 0.7)
        idx_big, sc_big = sas.search(Q, lq, 600, 0.7)   # k beyond the smallest shard: short lists are padded before the merge
        lam_all = D.all_gather_rows(sas.lambdas.reshape(-1, 1), sas.sizes).reshape(-1)
        # item-graph lambdas over the row blocks (as_compute_taumode_signals_sharded): the feature columns as signals
        lam_sig = comm.signal_lambdas(torch.from_numpy(X[lo:hi].copy()).cuda(), n, gl.block, A.TauMode.Median)
        lam_sig_p = comm.signal_lambdas(X[lo:hi].copy(), n, gl.block, A.TauMode.Percentile(0.9))   # host shard, host output
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, "lib%d.npz" % rank), indptr=csr[0], indices=csr[1], data=csr[2], nnz=np.array(gl.nnz()),
                 lam_sig=lam_sig.cpu().numpy(), lam_sig_p=lam_sig_p,
                 join=np.array(join), lam=lam_all.cpu().numpy(), idx=idx.cpu().numpy(), sc=sc.cpu().numpy(), lq=lq.cpu().numpy(),
                 idx_big=idx_big.cpu().numpy(), sc_big=sc_big.cpu().numpy(), stats=np.array(sas.lambda_stats))
        comm.close()
    finally:
        dist.destroy_process_group()


def _lib_fail_worker(rank, world, port, n, d, stage, out_dir):
    import arrowspace_amd as A
    from arrowspace_amd import distributed as D
    from tests import synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        ctx = A.Context(0)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx._stream = torch.cuda.current_stream().cuda_stream
        ctx.set_option("join_min_rows", 512)
        comm = D.LibComm.torch_transport(ctx)
        lo, hi = comm.shard_range(n)
        X, _ = synth.mixture(n, d, seed=13, n_clusters=16, rank=4)
        gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
        sas = D.ShardedArrowSpace(D.HipOps(ctx), torch.from_numpy(X[lo:hi].copy()).cuda(), n, gp, A.TauMode.Median, libcomm=comm)
        Xs = torch.from_numpy(X[lo:hi].copy()).cuda()
        if stage >= 4:   # the calls behind the build: statistics (4), search (5), item-graph lambdas (6; 7: after the sums
            # exchange, where the peers have no second round to meet the failing rank in; 8: a rank-local argument check)
            gl0 = sas.build_item_graph()
            sas.build_lambdas(synth.mixture(n, d, seed=13, n_clusters=16, rank=4)[1])
            Q = torch.from_numpy(X[[1, n // 2]] * 1.02).cuda()
            lq = sas.prepare_queries(Q)

            def call():
                if stage == 4:
                    comm.lambda_stats(sas.lambdas)
                elif stage == 5:
                    sas.search(Q, lq, 10, 0.7)
                else:
                    comm.signal_lambdas(Xs, n, gl0.block, A.TauMode.Median)
        else:
            def call():
                sas.build_item_graph()
        if rank == world - 1:
            ctx.set_option("sharded_inject_failure", stage)  # this rank alone fails, at checkpoint `stage`
        status, text = 0, ""
        try:
            call()
        except A.ArrowSpaceError as e:
            status, text = e.status, str(e)
        ctx.set_option("sharded_inject_failure", 0)
        call()  # the communicator is in step again: the same call succeeds on every rank
        gl = sas.build_item_graph()
        np.savez(os.path.join(out_dir, "fail%d.npz" % rank), status=np.array(status), nnz=np.array(gl.nnz()), text=np.array(text))
        comm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,stage", [(2, 1), (2, 2), (3, 2), (2, 3), (2, 4), (3, 5), (2, 5), (2, 6), (2, 7), (3, 7), (2, 8)])
def test_library_sharded_build_ranks_leave_together_when_one_fails(tmp_path, world, stage):
    """One rank fails locally (test hook: before the row all-gather, before the list exchange, before the nnz all-reduce; in
    the lambda statistics, in the sharded search, in the item-graph lambdas): it returns its own status, its peers return
    AS_ERR_PEER instead of waiting in the collective it never reaches (or merging a result that silently misses a shard), no
    rank hangs, and the next call on the same communicator succeeds."""
    import arrowspace_amd as A
    from arrowspace_amd import _lib
    from oracle import oracle_ctypes as O
    from tests import synth
    n, d = 3000, 64
    mp.spawn(_lib_fail_worker, args=(world, _free_port(), n, d, stage, str(tmp_path)), nprocs=world, join=True)
    X, _ = synth.mixture(n, d, seed=13, n_clusters=16, rank=4)
    nnz = len(O.build_laplacian(X, O.graph_params(0.5, 16, 16, 2.0, 0.25))[1])
    for k in range(world):
        r = np.load(tmp_path / ("fail%d.npz" % k))
        assert int(r["status"]) == ((_lib.AS_ERR_ARG if stage < 4 else _lib.AS_ERR_HIP) if k == world - 1 else _lib.AS_ERR_PEER), str(r["text"])
        if k == world - 1:
            assert "injected failure" in str(r["text"])
        assert int(r["nnz"]) == nnz


@pytest.mark.parametrize("n,world,d", [(3000, 2, 64), (1111, 2, 64), (3000, 3, 64), (1000, 3, 64), (3000, 2, 200)])
def test_library_sharded_build_matches_single_process(tmp_path, n, world, d):
    """as_build_laplacian_sharded with 2 and 3 ranks on the one GPU (transport callbacks over gloo): the stacked row blocks
    are the oracle's Laplacian; uneven shards, a rank without rows, rows wider than 128."""
    from oracle import oracle_ctypes as O
    from tests import synth
    mp.spawn(_lib_worker, args=(world, _free_port(), n, d, str(tmp_path)), nprocs=world, join=True)
    X, cent = synth.mixture(n, d, seed=13, n_clusters=16, rank=4)
    gp = O.graph_params(0.5, 16, 16, 2.0, 0.25)
    csr = O.build_laplacian(X, gp)
    lam = O.taumode_lambdas(X, O.build_laplacian(np.ascontiguousarray(cent.T), gp), O.TAU_MEDIAN)
    Q = X[[1, n // 2, n - 1, 77]] * 1.02
    for r in [np.load(tmp_path / ("lib%d.npz" % k)) for k in range(world)]:
        assert int(r["join"]) == 1 and int(r["nnz"]) == len(csr[1])
        assert np.array_equal(r["indptr"], csr[0]) and np.array_equal(r["indices"], csr[1])
        np.testing.assert_allclose(r["data"], csr[2], rtol=1e-13)
        # as_lambda_stats_sharded (src/builder.rs:434-445) and as_search_lambda_aware_sharded (src/core.rs:778-798)
        np.testing.assert_allclose(r["lam"], lam, rtol=1e-12, atol=1e-300)
        np.testing.assert_allclose(r["stats"], [r["lam"].min(), r["lam"].max(), r["lam"].mean()], rtol=1e-12)
        # as_compute_taumode_signals_sharded: TauMode::compute_synthetic_lambda_csr (src/taumode.rs:552-660) of every feature
        # column over the N x N graph, the sums taken per row block and added in rank order
        np.testing.assert_allclose(r["lam_sig"], O.taumode_lambdas(np.ascontiguousarray(X.T), csr, O.TAU_MEDIAN), rtol=1e-12, atol=1e-300)
        np.testing.assert_allclose(r["lam_sig_p"], O.taumode_lambdas(np.ascontiguousarray(X.T), csr, O.TAU_PERCENTILE, 0.9), rtol=1e-12, atol=1e-300)
        for j in range(4):
            oi, osc = O.search_lambda_aware(X, r["lam"], Q[j], float(r["lq"][j]), 10, 0.7)
            assert np.array_equal(r["idx"][j], oi.astype(np.int64))
            assert np.array_equal(r["sc"][j], osc)
            oi, osc = O.search_lambda_aware(X, r["lam"], Q[j], float(r["lq"][j]), 600, 0.7)
            assert np.array_equal(r["idx_big"][j], oi.astype(np.int64))
            assert np.array_equal(r["sc_big"][j], osc)


def test_library_sharded_build_one_rank_rccl():
    """The RCCL transport of the library with one rank: librccl is loaded at run time, the communicator comes from a unique
    id, and every exchange step of the build runs through ncclSend / ncclRecv / ncclAllReduce on the context's stream."""
    import arrowspace_amd as A
    from arrowspace_amd import distributed as D
    from oracle import oracle_ctypes as O
    from tests import synth
    ctx = A.default_context()
    n, d = 2000, 48
    X, _ = synth.mixture(n, d, seed=5, n_clusters=8, rank=4)
    gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
    comm = D.LibComm.rccl(ctx, D.LibComm.unique_id(ctx), 0, 1)
    try:
        assert comm.shard_range(n) == (0, n)
        gl = comm.build_laplacian(torch.from_numpy(X).cuda(), n, gp)
        ip, ix, dv = gl.block.to_host()
    finally:
        comm.close()
    csr = O.build_laplacian(X, O.graph_params(0.5, 16, 16, 2.0, 0.25))
    assert gl.nnz() == len(csr[1]) and np.array_equal(ip, csr[0]) and np.array_equal(ix, csr[1])
    np.testing.assert_allclose(dv, csr[2], rtol=1e-13)


def _lib_rccl_multi_worker(rank, world, port, n, d, out_dir):
    import arrowspace_amd as A
    from arrowspace_amd import distributed as D
    from tests import synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)  # only carries the 128-byte id and the result gather
    try:
        torch.cuda.set_device(rank)
        ctx = A.Context(rank)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx._stream = torch.cuda.current_stream().cuda_stream
        ctx.set_option("join_min_rows", 512)
        box = [D.LibComm.unique_id(ctx) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        comm = D.LibComm.rccl(ctx, box[0], rank, world)
        lo, hi = comm.shard_range(n)
        X, _ = synth.mixture(n, d, seed=13, n_clusters=16, rank=4)
        gl = comm.build_laplacian(torch.from_numpy(X[lo:hi].copy()).cuda(), n, A.GraphParams(0.5, 16, 16, 2.0, 0.25))
        csr = gl.to_host()
        np.savez(os.path.join(out_dir, "librccl%d.npz" % rank), indptr=csr[0], indices=csr[1], data=csr[2], nnz=np.array(gl.nnz()))
        comm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs at least two GPUs (RCCL wants one GPU per rank)")
@pytest.mark.parametrize("n", [20_000, 3_000])
def test_library_sharded_build_multi_rank_rccl(tmp_path, n):
    """as_build_laplacian_sharded over the library's RCCL transport with min(device_count, 8) ranks, one GPU each."""
    from oracle import oracle_ctypes as O
    from tests import synth
    world, d = min(torch.cuda.device_count(), 8), 64
    mp.spawn(_lib_rccl_multi_worker, args=(world, _free_port(), n, d, str(tmp_path)), nprocs=world, join=True)
    X, _ = synth.mixture(n, d, seed=13, n_clusters=16, rank=4)
    csr = O.build_laplacian(X, O.graph_params(0.5, 16, 16, 2.0, 0.25))
    for r in [np.load(tmp_path / ("librccl%d.npz" % k)) for k in range(world)]:
        assert int(r["nnz"]) == len(csr[1])
        assert np.array_equal(r["indptr"], csr[0]) and np.array_equal(r["indices"], csr[1])
        np.testing.assert_allclose(r["data"], csr[2], rtol=1e-13)


# ---- the sharded build with the one-GPU algorithm: row order + int8 first pass + masked f16 pass, sharded by image positions ----
_POS_STATS = ("sharded_join", "sharded_by_positions", "knn_join_prefilter", "knn_join_prefilter_rotated", "knn_join_row_order",
              "sharded_listed_slots", "sharded_foreign_slots", "knn_uncertified_rows", "sharded_rows_late", "sharded_prepare_by_rows",
              "knn_join_cone_tiles")


def _lib_positions_worker(rank, world, port, n, d, topk, own_rows, out_dir):
    import arrowspace_amd as A
    from arrowspace_amd import distributed as D
    from tests import synth
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        ctx = A.Context(0)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx._stream = torch.cuda.current_stream().cuda_stream
        comm = D.LibComm.torch_transport(ctx)
        ctx.set_option("sharded_own_rows", own_rows)
        ctx.set_option("join_cone_tiles_max_e3", 1000)   # (the tiled first pass whatever share of the tile pairs the bound keeps on this data)
        lo, hi = comm.shard_range(n)
        X, _ = synth.mixture(n, d, seed=29, n_clusters=96, rank=6)
        X[n // 3] = X[5]            # duplicates across shards: ties by row id must survive the trip through image positions
        X[n - 7] = X[5]
        X[n // 2] = 0.0
        gp = A.GraphParams(0.5, 16, topk, 2.0, 0.25)
        gl = comm.build_laplacian(torch.from_numpy(X[lo:hi].copy()).cuda(), n, gp)
        stats = {k: ctx.stat(k) for k in _POS_STATS}
        csr = gl.to_host()
        if rank == 0:
            np.savez(os.path.join(out_dir, "pos.npz"), indptr=csr[0], indices=csr[1], data=csr[2], nnz=np.array(gl.nnz()))
        np.save(os.path.join(out_dir, "pos_stats%d.npy" % rank), np.array([stats[k] for k in sorted(stats)]))
        comm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,world,d,topk,own_rows", [(70_000, 3, 128, 16, 1), (70_000, 3, 128, 16, 0), (66_000, 2, 96, 16, 1), (70_000, 2, 128, 38, 1),
                                                      (66_048, 4, 128, 16, 1), (66_560, 2, 256, 16, 1)])
def test_library_sharded_build_by_image_positions(tmp_path, n, world, d, topk, own_rows):
    """From 65 536 rows the sharded build runs the ONE-GPU algorithm (src/laplacian.rs:203-292 replaced by: rotated
    coordinates -> row order -> int8 first pass -> masked f16 pass -> exact re-rank), every rank on its own range of IMAGE
    positions, the lists exchanged by position and translated back to rows.  The stacked row blocks must be the Laplacian of
    the one-GPU build (itself checked against the oracle at these sizes in test_gpu_stress / test_gpu_parity), bit for bit in
    structure, with duplicates across shards and a zero row.
    own_rows = 1 (the default since round 4): the join is prepared from the ranks' OWN rows -- norms, rotated coordinates and the f16
    image are made per rank and all-gathered (392 B per row at 128 columns), the row order comes from the gathered coordinates --
    and the all-gather of the f64 rows runs on a second stream under the first pass and the f16 pass; 0: rows first, as in round 3."""
    import arrowspace_amd as A
    from tests import synth
    mp.spawn(_lib_positions_worker, args=(world, _free_port(), n, d, topk, own_rows, str(tmp_path)), nprocs=world, join=True)
    X, _ = synth.mixture(n, d, seed=29, n_clusters=96, rank=6)
    X[n // 3] = X[5]
    X[n - 7] = X[5]
    X[n // 2] = 0.0
    ctx = A.default_context()
    gl = A.build_laplacian_matrix(X, A.GraphParams(0.5, 16, topk, 2.0, 0.25), ctx=ctx)
    assert ctx.stat("knn_join") == 1
    ip, ix, dv = gl.matrix.to_host()
    r = np.load(tmp_path / "pos.npz")
    assert int(r["nnz"]) == len(ix)
    assert np.array_equal(r["indptr"], ip) and np.array_equal(r["indices"], ix)
    assert np.array_equal(r["data"], dv)
    keys = sorted(_POS_STATS)
    for k in range(world):
        st = dict(zip(keys, np.load(tmp_path / ("pos_stats%d.npy" % k)).tolist()))
        assert st["sharded_join"] == 1 and st["sharded_by_positions"] == 1, st
        late = own_rows if d <= 128 else 0   # (rows wider than 128: the f64 rows first; their first pass runs on the rank's own positions)
        assert st["sharded_rows_late"] == late and st["sharded_prepare_by_rows"] == late, st
        if d <= 128:   # the cone bound on tile pairs in front of the first pass, on the rank's own range of row tiles
            assert st["knn_join_cone_tiles"] == 1, st
        assert st["knn_join_prefilter"] == 1 and st["knn_join_row_order"] == 1, st
        assert st["sharded_foreign_slots"] * 2 < st["sharded_listed_slots"], st   # most partners of a position live on its own rank


def test_one_rank_of_a_sharded_build_replayed_alone(tmp_path):
    """scripts/rank_of_8.py, the tool behind DESIGN.md section 6's measurement: three processes share the GPU and run
    as_build_laplacian_sharded over a file transport while rank 1 records what it receives; one process then replays rank 1 with every
    collective answered from HBM on the library's own stream (nothing is sent).  The replayed block has the recorded entry counts and
    is, bit for bit, rows [lo, hi) of the one-GPU build of all rows (--check)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "scripts", "rank_of_8.py")
    xdir = str(tmp_path / "exchange")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    common = ["--rows", "200000", "--world", "3", "--rank", "1", "--dir", xdir]
    rec = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), script, "record"] + common, env=env, capture_output=True, text=True, timeout=900)
    assert rec.returncode == 0, rec.stderr[-2000:]
    assert "recorded" in rec.stdout
    out = str(tmp_path / "replay.json")
    rep = subprocess.run([sys.executable, script, "replay", "--check", "--reps", "2", "--out", out] + common, env=env, capture_output=True, text=True,
                         timeout=900)
    assert rep.returncode == 0, rep.stderr[-2000:]
    with open(out) as f:
        leg = json.load(f)
    assert leg["block_equals_rows_of_the_one_gpu_build"] is True
    assert leg["world"] == 3 and leg["rank"] == 1 and leg["rank_ms"] > 0 and leg["received_bytes_from_peers"] > 0
    assert leg["stats"]["sharded_by_positions"] == 1 and leg["collectives"] >= 10
