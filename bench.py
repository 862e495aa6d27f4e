#!/usr/bin/env python3
"""bench.py -- lambda-tau build items/s + lambda-aware search QPS on synthetic N x D rows.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--n 1000000] [--dim 128] [--nq 10000]
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One STEP = one pass of the hot path over the whole synthetic set (BASELINE.json configs[2]):
  build   item graph: cosine k-NN -> eps/top-k capped union CSR Laplacian over the N items
          (with_lambda_graph(eps .5, k 16, topk 16, p 2, sigma .25); MFMA distance block + assembly),
          then the reference mapping: F x F Laplacian over the supplied centroids and the per-item
          lambda-tau (Median tau) for all N items;
  search  prepare_query_item for nq queries + search_lambda_aware(k 10, alpha .7) over the N items.
Inputs are resident in HBM before the timed region.  `value` = N / build seconds (whole job);
`search_qps` = nq / search seconds.  N > 1 row-shards the same N (strong scaling) with RCCL
all-gathers of the row blocks, the neighbour tables and the per-shard top-k.

Beside the contract's fields the JSON line carries (all measured in this run, HIP events on the library's stream):
  roofline          the kernel with the most time per step: the work it EXECUTES per launch over its measured launch time against
                    the peak of its own pipe (int8 / f16 MFMA, or HBM); roofline_kernels: the same for every long kernel
  reference_equivalent  SURVEY.md 8(d)'s 2 n^2 d of the reference's all-pairs loop over the time of the launches that replace
                    it: how much of that loop is never executed -- a separate figure, never a frac
  roofline_scan     the lambda-tau scan (K6/K7/K8): bytes n (8 D + 8) per launch against HBM
  roofline_spmm     the item-graph lambda (the D feature columns as signals over the N x N Laplacian the build returns): bytes
                    nnz 16 + N D 8 per launch against HBM, the gather rate and the f64 vector bound beside it
  roofline_search   the search scan at the benchmark batch (K9, MFMA-bound at nq = 10 k): 2 nq N D flops
  roofline_search_hbm  the search at nq = 1 and nq = 32 with every item row streamed (bandwidth-bound regime): bytes N (8 D + 8) per
                    pass; equivalent_filtered: the same batches on an index searched before (int8 filter: an equivalent, not a bandwidth)
  value_host_inputs the build with the item matrix starting in host memory (PCIe inclusive; never `value`)
  value_hard_data   the item graph on a mixture the rotated first pass cannot exploit (which path ran; never `value`)
  value_cfg1 / value_cfg3_one_gpu / value_wide   the other BASELINE.json configs on this one GPU (100 k x 128; 4 M x 128, which the
                    baseline shards over 8 GPUs; 1 M x 768 with Percentile(0.95) tau and a 100 k-query batch): build ms, the path taken,
                    the kernel with the most time; geometry_sweep: the 1 M x 128 item graph on mixtures of rank 8 ... 128 with the
                    bounds that stayed on.  Never `value`
  cpu_baseline      the oracle on the host cores, bounded sample
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))

sys.path.insert(0, os.path.join(ROOT, "scripts"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# peaks: /opt/skills/guides/MI355X_MICROARCH.md (chip table); f64 MFMA measured (scripts/micro/mfma_f64_peak.hip)
F64_MFMA_PEAK_TFLOPS = 39.3
MFMA16_PEAK_TFLOPS = 2500.0   # dense bf16 / f16 (v_mfma_f32_32x32x16_{bf16,f16}: 32768 flop / 32 clk / SIMD at 2.4 GHz)
INT8_PEAK_TOPS = 5000.0       # dense int8 (v_mfma_i32_32x32x32_i8: twice the 16-bit rate, MI355X_MICROARCH.md)
F32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32
HBM_PEAK_GBS = 8000.0

from source_sha import source_sha16  # noqa: E402
import bench_line  # noqa: E402

SOURCE_SHA16 = source_sha16(ROOT)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=8)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--n", type=int, default=1_000_000)
    p.add_argument("--dim", type=int, default=128)
    p.add_argument("--nq", type=int, default=10_000)
    p.add_argument("--k", type=int, default=10)
    p.add_argument("--alpha", type=float, default=0.7)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extra-legs", action="store_true", help="skip the host-input build and the small-batch search legs")
    p.add_argument("--no-spmm-leg", action="store_true", help="skip the item-graph lambda leg (roofline_spmm)")
    p.add_argument("--rank-of-8-leg", action="store_true", help="add value_cfg3_rank_of_8 (scripts/rank_of_8.py: 8 recording processes on this GPU, "
                                                              "then the replay of rank 3 of 8 at 4 M x 128).  Off by default: half a minute on a "
                                                              "warm box, but eight fresh processes importing torch took 4.7 minutes on a fresh one "
                                                              "-- not inside the driver's timed command; profiles/r06*_rank3_of_8_4M.json holds the result")
    p.add_argument("--no-config-legs", action="store_true", help="skip the other BASELINE configs (value_cfg1, value_cfg3_one_gpu, value_wide) "
                                                                 "and the geometry sweep")
    p.add_argument("--cand-precision", type=int, default=32, choices=[32, 64],
                   help="first candidate stage: 32 = 16-bit / f32 MFMA then f64 MFMA then exact (default); 64 = f64 MFMA first")
    p.add_argument("--join-pipe", type=int, default=16, choices=[16, 48, 32],
                   help="contraction of the join: 16 = one f16 plane (default), 48 = split bf16 in two phases, 32 = f32 MFMA")
    p.add_argument("--cpu-sample-n", type=int, default=32000)
    p.add_argument("--cpu-sample-nq", type=int, default=16)
    return p.parse_args()


def cpu_baseline(args, X_host_fn, cent, n_total):
    """The oracle (CPU restatement of the reference algorithm, OpenMP over the host cores) timed
    on a bounded sample of the same workload.  Reported baseline only."""
    from oracle import oracle_ctypes as O
    ns = min(args.cpu_sample_n, n_total)
    Xs = X_host_fn(0, ns)
    gp = O.graph_params(0.5, 16, 16, 2.0, 0.25)
    t0 = time.time()
    O.build_laplacian(Xs, gp)
    t_graph = time.time() - t0
    csrF = O.build_laplacian(np.ascontiguousarray(cent.T), gp)
    t0 = time.time()
    lam = O.taumode_lambdas(Xs, csrF, O.TAU_MEDIAN)
    t_lam = time.time() - t0
    nqs = args.cpu_sample_nq
    t0 = time.time()
    for j in range(nqs):
        q = Xs[(j * 7919) % ns] * 1.02
        lq = O.prepare_query_item(q, csrF, O.TAU_MEDIAN)
        O.search_lambda_aware(Xs, lam, q, lq if lq != 0.0 else 1e-3, args.k, args.alpha)
    t_search = time.time() - t0
    # graph cost grows as n^2, lambda as n: time the sample, extrapolate to the full N
    t_full = t_graph * (n_total / ns) ** 2 + t_lam * (n_total / ns)
    qps_full = nqs / (t_search * (n_total / ns))
    # the oracle's side of `published_equivalents` (the only two figures the reference publishes for this path):
    # the same inputs and parameters as the GPU side below, timed here because only this leg may call the oracle
    pub = {}
    gp5 = O.graph_params(0.5, 5, 5, 2.0, None)
    for n_items in (50, 100, 200, 400):
        Xp = published_laplacian_rows(n_items, 24)
        O.build_laplacian(Xp, gp5)
        each = []
        for _ in range(5):
            t0 = time.perf_counter()
            csr_p = O.build_laplacian(Xp, gp5)
            each.append(time.perf_counter() - t0)
        pub["build_laplacian_%dx24" % n_items] = {"oracle_cpu_ms": sorted(each)[2] * 1e3, "oracle_nnz": int(len(csr_p[1]))}
    rows_l, Q_l, lam_l, lq_l = published_lookup_inputs()
    t0 = time.perf_counter()
    for j in range(len(Q_l)):
        O.search_lambda_aware(rows_l, lam_l, Q_l[j], lq_l[j], 3, 0.9)
    pub["lookup_top3_batch256"] = {"oracle_cpu_ms": (time.perf_counter() - t0) * 1e3}
    return {
        "value": n_total / t_full, "unit": "items/s", "cores": O.num_threads(), "kind": "port",
        "search_qps": qps_full, "published_equivalents": pub,
        "measured_sample_items_per_s": ns / (t_graph + t_lam),
        "sample": "oracle (C restatement, OpenMP %d threads) on the first %d rows of the same mixture: item graph "
                  "%.2fs + lambdas %.3fs, %d queries %.2fs; value/search_qps are the n^2 (graph) and n (lambda, scan) "
                  "extrapolations to N=%d" % (O.num_threads(), ns, t_graph, t_lam, nqs, t_search, n_total),
    }


def published_laplacian_rows(n_items, n_dims, seed=42):
    """The shape of the reference's synthetic set (benches/base_laplacian_bench.rs:13-29: item i varies around the base pattern
    (i mod 10) / 10, value = |base + 0.01 j + noise|, noise uniform in [-0.1, 0.1)); numpy's generator, not Rust's StdRng."""
    rng = np.random.default_rng(seed)
    i = np.arange(n_items, dtype=np.float64)[:, None]
    j = np.arange(n_dims, dtype=np.float64)[None, :]
    return np.ascontiguousarray(np.abs((i % 10) * 0.1 + j * 0.01 + rng.uniform(-0.1, 0.1, size=(n_items, n_dims))))


def published_lookup_inputs(batch=256, seed=42):
    """benches/index_compute_bench.rs:108-150: the 64 x 24 set of the bench (tests/golden/proteins_64x24.json holds its numbers),
    `batch` queries = random rows x 1.02.  The bench scores with a query lambda of 0.0 through lambda_similarity; search_lambda_aware
    refuses that (src/core.rs:773-776), so items and queries carry a fixed lambda of 0.25 here (the scan's work is the same)."""
    with open(os.path.join(ROOT, "tests", "golden", "proteins_64x24.json")) as f:
        rows = np.array(json.load(f)["rows"], dtype=np.float64)
    rng = np.random.default_rng(seed)
    Q = np.ascontiguousarray(rows[rng.integers(0, len(rows), size=batch)] * 1.02)
    return rows, Q, np.full(len(rows), 0.25), np.full(batch, 0.25)


def published_equivalents(A, ctx):
    """GPU side of the only figures the reference publishes for this path (unknown CPU, older revision):
    build_laplacian_matrix on n x 24 rows-as-nodes sets (eps 0.5, k 5, topk 5, p 2, sigma None; benches/base_laplacian_bench.rs:166-179)
    and the batch-256 top-3 lookup (benches/baselines/base_index_bench.json:15).  Host inputs, as the reference's benches have them."""
    out = {}
    ref_ms = {50: 1.53, 100: 5.53, 200: 20.2, 400: 82.7}
    gp5 = A.GraphParams(0.5, 5, 5, 2.0, None)
    for n_items in (50, 100, 200, 400):
        Xp = published_laplacian_rows(n_items, 24)
        A.build_laplacian_matrix(Xp, gp5, ctx=ctx)
        ctx.synchronize()
        each = []
        for _ in range(9):
            t0 = time.perf_counter()
            gl = A.build_laplacian_matrix(Xp, gp5, ctx=ctx)
            ctx.synchronize()
            each.append(time.perf_counter() - t0)
        out["build_laplacian_%dx24" % n_items] = {"gpu_ms": sorted(each)[4] * 1e3, "gpu_nnz": int(gl.matrix.nnz()),
                                                  "reference_published_ms": ref_ms[n_items],
                                                  "reference_source": "benches/baselines/base_laplacian_bench.json:17-20 (unknown CPU)"}
    rows_l, Q_l, lam_l, lq_l = published_lookup_inputs()
    asp = A.ArrowSpace(rows_l, A.TauMode.Median, ctx)
    asp.update_lambdas(lam_l)
    asp.search_lambda_aware_batch(Q_l, lq_l, 3, 0.9)
    ctx.synchronize()
    each = []
    for _ in range(9):
        t0 = time.perf_counter()
        asp.search_lambda_aware_batch(Q_l, lq_l, 3, 0.9)
        ctx.synchronize()
        each.append(time.perf_counter() - t0)
    out["lookup_top3_batch256"] = {"gpu_ms": sorted(each)[4] * 1e3, "reference_published_ms": 2020.0,
                                   "reference_source": "benches/baselines/base_index_bench.json:15 (arrow alpha 0.9, batch 256; unknown CPU, "
                                                       "dataset of that older revision not recoverable -- the current bench source uses this 64 x 24 set)"}
    out["note"] = ("host inputs, median of 9 calls; at these sizes a call is launch- and transfer-bound (tens of microseconds of kernels), "
                   "so these are latency figures, not throughput; vs_baseline stays null: the published numbers are from an unknown CPU "
                   "and, for the lookup, an unknown dataset")
    return out


def git_head():
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return None


def recorded_traffic(kernel_substr, n, d):
    """HBM bytes per step from the committed rocprofv3 --pmc passes of this same command (scripts/profile_round.sh; PMC
    counters cannot be read from inside this process): `traffic`, with the file and commit it comes from beside it.
    kernel_substr: one name or a tuple of names (a profile group of several kernels: their sum)."""
    if isinstance(kernel_substr, (tuple, list)):
        parts = [recorded_traffic(k, n, d) for k in kernel_substr]
        parts = [p_ for p_ in parts if p_]
        if not parts:
            return None
        return {"bytes_per_launch": sum(p_["bytes_per_launch"] for p_ in parts), "source": parts[0]["source"], "commit": parts[0]["commit"],
                "stale": parts[0]["stale"], "source_sha16": parts[0]["source_sha16"]}
    best = None
    pdir = os.path.join(ROOT, "profiles")
    # (the item-graph lambda has a pass of its own: scripts/profile_round.sh)
    suffix = "_pmc_traffic_spmm.json" if kernel_substr == "signal_sums_kernel" else "_pmc_traffic.json"
    # Which pass: the one RECORDED last (`recorded_unix`, written by scripts/pmc_traffic.py), never the one whose file name sorts
    # last (round 4 quoted r04zzzz, 25 commits old, over r04y).  Passes older than that field are eligible only when no newer one
    # exists, ordered by the commit time noted in profiles/pmc_passes.json.  `stale` says whether the pass was taken on the very
    # sources this process runs from (sha over csrc/ + include/: works on the GPU box, which has no .git).
    passes = []
    try:
        with open(os.path.join(pdir, "pmc_passes.json")) as f:
            legacy_time = json.load(f)
    except (OSError, ValueError):
        legacy_time = {}
    for name in (os.listdir(pdir) if os.path.isdir(pdir) else []):
        if not (re.match(r"r\d\d", name) and name.endswith(suffix)):
            continue
        try:
            with open(os.path.join(pdir, name)) as f:
                j = json.load(f)
        except (OSError, ValueError):
            continue
        if j.get("n") != n or j.get("dim") != d:
            continue
        passes.append((j.get("recorded_unix") or legacy_time.get(name, 0), name, j))
    passes.sort(key=lambda t: (t[0], t[1]))
    for when, name, j in passes[-1:]:   # the newest pass only: a kernel that no longer exists must not bring an old file's bytes along
        for k in j.get("kernels", []):
            if kernel_substr in k.get("kernel", ""):
                # (the PMC passes run ONE step, no warm-up: a kernel's total over the pass is its traffic per step)
                best = {"bytes_per_launch": k.get("hbm_bytes_total"), "launches_per_step": k.get("launches"),
                        "bytes_per_single_launch": k.get("hbm_bytes_per_launch"),
                        "source": "profiles/" + name, "commit": j.get("commit"),
                        "stale": j.get("source_sha16") != SOURCE_SHA16, "source_sha16": j.get("source_sha16")}
    return best


def settle_after_generator():
    """The torch generator of the synthetic rows (RNG, gathers, several GB of fresh allocations) leaves the GPU queue with one pause of
    40 - 100 ms somewhere in the 50 - 150 ms behind it (round 6, scripts/attic/stall_probe.py: with the matrix generated a second
    earlier the pause never falls into a timed call, 0 of 16 runs against 11 of 24; it is the generator's, not the library's: allocator,
    interpreter GC, interrupts, scratch of the lambda kernel and every bench leg in front were ruled out one by one -- DESIGN.md
    section 5).  Generating data is no part of any timed region: wait it out."""
    import torch
    torch.cuda.synchronize()
    time.sleep(1.0)


def config_leg(A, D, synth, ctx, dev, n, d, nq, taumode, k, alpha, rank=8, clusters=256, reps=2, seed=42, search=True, lambdas=True):
    """One BASELINE config (or one geometry of the sweep) on this GPU through the same calls as the timed region: item graph,
    reference-mapping lambdas, a query batch.  Wall times of `reps` repetitions behind one warm-up; the profile of the last one."""
    ops = D.HipOps(ctx)
    X, cent = synth.mixture_device(n, d, dev, seed=seed, n_clusters=clusters, rank=rank)
    settle_after_generator()
    gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
    sas = D.ShardedArrowSpace(ops, X, n, gp, taumode)
    Q = None
    if search and nq:
        qidx = synth.query_indices(n, nq, seed=seed)
        Q = X[torch.from_numpy(qidx.astype(np.int64)).to(dev)] * 1.02

    path_keys = ("knn_join", "knn_join_prefilter", "knn_join_prefilter_rotated", "knn_join_row_order", "knn_join_theta_grouped",
                 "knn_join_cone_tiles", "knn_join_cone_tiles_kept_e6", "knn_join_prefilter_tail_e6", "knn_uncertified_rows",
                 "knn_uncertified_rows_stage1a", "knn_join_order_pivots", "knn_join_order_cover_rounds", "knn_join_order_cover_low_e3",
                 "knn_join_theta_far_rows", "knn_join_theta_dip_rows", "knn_join_candidates", "knn_rerank_pairs_scored")
    path = {}

    def step():
        t0 = time.perf_counter()
        gl = sas.build_item_graph()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        path.update({kk: ctx.stat(kk) for kk in path_keys})   # (read here: the search behind it runs passes that count in the same names)
        if lambdas:
            sas.build_lambdas(cent)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if Q is not None:
            lq = sas.prepare_queries(Q)
            sas.search(Q, lq, k, alpha)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        return gl, t1 - t0, t2 - t1, t3 - t2

    step()
    each = []
    for r in range(reps):
        if r == reps - 1:
            ctx.profile(True)   # (the stage profile of the last repetition)
        gl, a, b, c = step()
        each.append((a, b, c))
    prof = ctx.profile_all()
    ctx.profile(False)
    # the MEDIAN repetition (by its item-graph time): a repetition that runs into the allocator -- the graph handle of the call
    # before is released inside it, and behind the 4 M-row leg that can mean returning gigabytes to the device -- is not the path
    tg, tl, ts = sorted(each)[(reps - 1) // 2]
    stages = {kk: v[0] for kk, v in prof.items() if v[1]}
    inner = ("knn_rerank_link", "knn_rerank_fill", "knn_rerank_score", "knn_rerank_rank")   # (inside knn_rerank)
    top = sorted(((v, kk) for kk, v in stages.items() if kk not in inner), reverse=True)[:4]
    leg = {"n_items": n, "dim": d, "mixture": {"rank": rank, "clusters": clusters, "seed": seed},
           "value": n / (tg + tl), "unit": "items/s", "build_ms": (tg + tl) * 1e3, "item_graph_ms": tg * 1e3, "lambda_ms": tl * 1e3,
           "item_graph_ms_each": [round(e[0] * 1e3, 3) for e in each],
           "item_graph_nnz": int(gl.matrix.nnz() if hasattr(gl, "matrix") else gl.nnz()),
           "dominant_stage": {"name": top[0][1], "ms": top[0][0]} if top else None,
           "top_stages_ms": {kk: v for v, kk in top},
           "path": {kk: path[kk] for kk in path_keys[:15]},
           "candidates_per_row": path["knn_join_candidates"] / max(1, n),
           "rows_fetched_per_row": path["knn_rerank_pairs_scored"] / max(1, n)}
    if Q is not None:
        leg.update({"n_queries": nq, "search_ms": ts * 1e3, "search_qps": nq / ts if ts > 0 else None})
    del sas, X, Q, gl
    torch.cuda.empty_cache()
    return leg


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ARROWSPACE_DIST_FORCE=1 under a one-rank torch.distributed.run: the RCCL path on a one-GPU box
    use_dist = world > 1 or (os.environ.get("ARROWSPACE_DIST_FORCE") == "1" and "RANK" in os.environ)
    # ARROWSPACE_SHARED_GPU=1: every rank runs on GPU 0 (RCCL refuses two ranks on one device, so the library's exchange goes
    # over its callback transport on gloo): the N > 1 bookkeeping of this script and the sharded path of the library on a one-GPU box
    shared_gpu = os.environ.get("ARROWSPACE_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import arrowspace_amd as A
    from arrowspace_amd import distributed as D
    from arrowspace_amd import synth

    def make_ctx():
        c = A.Context(local_rank)
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        c._stream = torch.cuda.current_stream().cuda_stream
        c.set_option("cand_precision", args.cand_precision)
        c.set_option("join_pipe", args.join_pipe)
        return c

    ctx = make_ctx()
    n, d, nq = args.n, args.dim, args.nq
    lo, hi = D.shard_range(n, rank, world)
    # seeded mixture rows, generated on the device (the numpy form of the same generator takes 35 s per 1 M x 128)
    Xd, cent = synth.mixture_device(hi - lo, d, dev, seed=42, row_start=lo)
    settle_after_generator()
    qidx = synth.query_indices(n, nq, seed=42)
    Q = torch.from_numpy(synth.mixture_at(qidx, d, seed=42) * 1.02).to(dev)
    gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
    # N > 1: the exchange steps of the build, the lambda statistics and the search run INSIDE the library over its own RCCL
    # communicator (csrc/sharded.hip): the one-GPU algorithm sharded by image positions.  ARROWSPACE_TORCH_EXCHANGE=1 keeps
    # the torch.distributed form (the join in the order of the caller's rows, no first pass).
    libcomm = None
    transport = "none (one rank)"
    if use_dist and os.environ.get("ARROWSPACE_TORCH_EXCHANGE") == "1":
        transport = "torch.distributed (%s), stage-level entry points" % dist.get_backend()
    elif use_dist and shared_gpu:
        libcomm = D.LibComm.torch_transport(ctx)
        transport = "library exchange over its callback transport on gloo (ranks share one GPU)"
    elif use_dist:
        # The library's own RCCL communicator.  Its bring-up (ncclCommInitRank over the bootstrap sockets) and its first
        # collective run in a side thread under a deadline: a rank that cannot come up says so in an all-reduce on torch's
        # process group, and then EVERY rank continues on the torch.distributed exchange instead (reported in `transport`);
        # nothing is re-executed, and a hung bring-up cannot hang the job.
        # The bring-up thread owns a context and a communicator of its OWN (a context is one per thread and not re-entrant,
        # INTEGRATION.md): on success the whole run adopts that pair; on a timeout the main thread never touches either again
        # (a late success inside the abandoned thread then meets no peer and no shared state), and the vote below runs on a
        # gloo group, so it cannot queue behind hung device work.
        import threading
        vote_group = dist.new_group(backend="gloo")
        box = [D.LibComm.unique_id(ctx) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        result = {}

        def bring_up():
            try:
                torch.cuda.set_device(dev)
                cb = make_ctx()
                c = D.LibComm.rccl(cb, box[0], rank, world)
                c.lambda_stats(torch.zeros(1, dtype=torch.float64, device=dev))   # first collective: a 32-byte all-gather
                cb.synchronize()
                result["ctx"], result["comm"] = cb, c
            except Exception as e:  # noqa: BLE001
                result["error"] = repr(e)

        th = threading.Thread(target=bring_up, daemon=True)
        th.start()
        th.join(float(os.environ.get("ARROWSPACE_RCCL_TIMEOUT_S", "120")))
        came_up = (not th.is_alive()) and "comm" in result
        ok = torch.tensor([1 if came_up else 0], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=vote_group)
        if int(ok) == 1:
            ctx, libcomm = result["ctx"], result["comm"]
            transport = "library exchange over its own RCCL communicator"
        else:
            why = result.get("error", "timed out" if th.is_alive() else "a peer failed")
            transport = "torch.distributed (nccl), stage-level entry points -- FALLBACK: the library's RCCL communicator did not come up (%s)" % why
            sys.stderr.write("bench.py rank %d: %s\n" % (rank, transport))
    ops = D.HipOps(ctx)
    sas = D.ShardedArrowSpace(ops, Xd, n, gp, A.TauMode.Median, libcomm=libcomm)

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    state = {}

    def build_step():
        gl_items = sas.build_item_graph()
        sas.build_lambdas(cent)
        state["nnz"] = gl_items.matrix.nnz() if hasattr(gl_items, "matrix") else gl_items.nnz()
        return gl_items

    def search_step():
        lq = sas.prepare_queries(Q)
        idx, sc = sas.search(Q, lq, args.k, args.alpha)
        state["top"] = (idx, sc)

    for _ in range(args.warmup):
        build_step()
        search_step()
    ctx.profile(True)
    t_build = t_search = 0.0
    sync()
    t_all0 = time.perf_counter()
    for _ in range(args.steps):
        t0 = time.perf_counter()
        build_step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        search_step()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        t_build += t1 - t0
        t_search += t2 - t1
    sync()
    t_all = time.perf_counter() - t_all0
    times = torch.tensor([t_all, t_build, t_search], dtype=torch.float64, device="cpu" if shared_gpu else dev)
    if use_dist:
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
    t_all, t_build, t_search = [float(x) for x in times]
    # device memory in use on every rank's GPU right after the timed steps (hipMemGetInfo: torch's blocks, the library's handles and
    # its workspace cache, i.e. the high-water mark of the step -- the cache keeps what a step took).  Ranks sharing a GPU see the sum.
    free_b, total_b = torch.cuda.mem_get_info(dev)
    mem = torch.tensor([float(total_b - free_b)], dtype=torch.float64, device="cpu" if shared_gpu else dev)
    mem_all = [mem.clone() for _ in range(world)] if use_dist else [mem]
    if use_dist:
        dist.all_gather(mem_all, mem)
    device_memory_used_gb = [round(float(m) / 1e9, 3) for m in mem_all]
    prof = ctx.profile_all()
    ctx.profile(False)
    stats = {k: ctx.stat(k) for k in ("knn_join", "knn_join_pipe", "knn_join_blocks", "knn_join_candidates", "knn_uncertified_rows",
                                      "knn_uncertified_rows_stage1a", "knn_join_delta_e9", "search_uncertified_rows", "search_scan",
                                      "knn_mfma_join_bf16_blocks_full", "knn_join_prefilter", "knn_join_prefilter_rotated",
                                      "knn_join_prefilter_blocks", "knn_join_prefilter_blocks_flagged", "knn_rerank_scored",
                                      "knn_join_row_order", "knn_join_theta_grouped", "knn_join_cone_tiles", "knn_join_cone_tiles_kept_e6",
                                      "knn_rerank_batches", "knn_rerank_pairs_scored", "knn_join_order_chain")}

    # ---- untimed extra legs (rank 0, one GPU): small-batch search, host-resident inputs --------------------------------
    extra = {}
    if rank == 0 and world == 1 and not args.no_extra_legs:
        legs = []
        for b in (1, 32):
            Qb = Q[:b].contiguous()
            lqb = sas.prepare_queries(Qb)
            for _ in range(2):
                sas.search(Qb, lqb, args.k, args.alpha)
            torch.cuda.synchronize()
            reps = 10
            t0 = time.perf_counter()
            for _ in range(reps):
                sas.search(Qb, lqb, args.k, args.alpha)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            bytes_pass = float(n) * (8.0 * d + 8.0)
            legs.append({"nq": b, "ms_per_batch": dt * 1e3, "qps": b / dt, "algorithmic_bytes_per_pass": bytes_pass,
                         "achieved": bytes_pass / dt / 1e9, "frac": bytes_pass / dt / 1e9 / HBM_PEAK_GBS})
        # the same batches with the int8 filter of an index searched before switched off: every batch STREAMS the N (8 D + 8)
        # bytes of the items (what the first search of an index does): a bandwidth, where the filtered form above is an equivalent
        streamed = []
        ctx.set_option("search_stream_filter", 0)
        try:
            for b in (1, 32):
                Qb = Q[:b].contiguous()
                lqb = sas.prepare_queries(Qb)
                for _ in range(2):
                    sas.search(Qb, lqb, args.k, args.alpha)
                torch.cuda.synchronize()
                reps = 10
                ctx.profile(True)
                t0 = time.perf_counter()
                for _ in range(reps):
                    sas.search(Qb, lqb, args.k, args.alpha)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / reps
                pl = ctx.profile_all()
                ctx.profile(False)
                list_ms, list_n = pl.get("search_stream_scan_list", (0.0, 0))
                seq_ms, seq_n = pl.get("search_stream_scan", (0.0, 0))
                bytes_pass = float(n) * (8.0 * d + 8.0)
                k_ms = list_ms / list_n if list_n else dt * 1e3   # HIP events around the one launch that streams the items
                streamed.append({"nq": b, "kernel_ms": k_ms, "sequence_ms": seq_ms / seq_n if seq_n else None,
                                 "wall_ms_per_batch": dt * 1e3, "qps": b / dt, "bytes_streamed_per_pass": bytes_pass,
                                 "achieved": bytes_pass / (k_ms * 1e-3) / 1e9, "frac": bytes_pass / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "wall_GBs": bytes_pass / dt / 1e9, "stream_scan_filter": ctx.stat("stream_scan_filter")})
        finally:
            ctx.set_option("search_stream_filter", 1)
        best = max(streamed, key=lambda leg: leg["frac"])
        extra["roofline_search_hbm"] = {"bound": "hbm", "kernel": "stream_scan_d128_kernel<list> (search_lambda_aware at small batches: the launch that "
                                                                  "streams every item row once per batch, HIP events around it; sequence_ms = "
                                                                  "all launches of a batch, wall_ms_per_batch = with the host)",
                                        "achieved": best["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": best["frac"],
                                        "traffic": None, "legs": streamed,
                                        "equivalent_filtered": {
                                            "what": "EQUIVALENT, not a bandwidth: from the second search of an index on, a rotated int8 image "
                                                    "(32 B per item) replaces the list pass, so ~100 MB are read instead of N (8 D + 8); the "
                                                    "figures below divide the reference's bytes by the measured time",
                                            "legs": [dict(leg, equivalent_GBs=leg.pop("achieved"), equivalent_frac=leg.pop("frac")) for leg in legs]}}
        # the reference's build() starts from host rows (Vec<Vec<f64>>): the same build with the item matrix in host memory
        X_host = Xd.cpu().numpy()
        A.build_laplacian_matrix(X_host[:40000], gp, ctx=ctx)  # staging buffer and copy threads warm
        t0 = time.perf_counter()
        X_up = ctx.upload(X_host)          # as_device_upload: once; graph, lambdas and index all read this copy
        ctx.synchronize()
        t_up = time.perf_counter() - t0
        gl_h = A.build_laplacian_matrix(X_up, gp, ctx=ctx)
        asp_h, _ = A.ArrowSpaceBuilder.new(ctx).with_lambda_graph(0.5, 16, 16, 2.0, 0.25).build(X_up, cent)
        ctx.synchronize()
        t_host = time.perf_counter() - t0
        extra["value_host_inputs"] = {"value": n / t_host, "unit": "items/s", "build_ms": t_host * 1e3, "upload_ms": t_up * 1e3,
                                      "upload_GBs": n * d * 8 / 1e9 / t_up,
                                      "note": "item matrix (%.2f GB) in pageable host memory at the start: uploaded ONCE (as_device_upload, "
                                              "pipelined through pinned staging by several host threads), then the same build on the device "
                                              "copy; PCIe-inclusive" % (n * d * 8 / 1e9)}
        del gl_h, asp_h, X_host, X_up
        # ---- the clustering pre-stage (the step in front of the hot path inside build()) ----------
        # ArrowSpaceBuilder.build(rows) WITHOUT supplied centroids: Two-NN intrinsic dimension + optimal K + pilot radius
        # (src/clustering.rs:36-164,384-492), then incremental clustering over all N rows (:547-910), then the same graph +
        # lambda stages as the bench line.  Rows resident in HBM.  Never in `value` (the bench line supplies the centroids).
        try:
            import arrowspace_amd.api as API
            t0 = time.perf_counter()
            k_opt, radius, idim = API.compute_optimal_k(Xd, None, ctx=ctx)
            ctx.synchronize()
            t_k = time.perf_counter() - t0
            t0 = time.perf_counter()
            cents_p, assign_p, sizes_p = API.incremental_clustering(Xd, k_opt, radius, 4096, ctx=ctx)
            ctx.synchronize()
            t_inc = time.perf_counter() - t0
            t0 = time.perf_counter()
            asp_p, gl_p = A.ArrowSpaceBuilder.new(ctx).with_lambda_graph(0.5, 16, 16, 2.0, 0.25).build(Xd)
            ctx.synchronize()
            t_full = time.perf_counter() - t0
            extra["prestage"] = {"optimal_k_ms": t_k * 1e3, "incremental_clustering_ms": t_inc * 1e3, "build_from_rows_ms": t_full * 1e3,
                                 "k_optimal": int(k_opt), "intrinsic_dimension": int(idim), "radius": float(radius),
                                 "clusters": int(len(cents_p)), "assigned": int((assign_p >= 0).sum()), "batch": 4096,
                                 "items_per_s_build_from_rows": n / t_full,
                                 "note": "build(rows) without supplied centroids: Two-NN + optimal K + pilot radius, incremental "
                                         "clustering of all N rows (batch 4096), F x F graph, per-item lambdas"}
            del asp_p, gl_p
        except Exception as e:  # noqa: BLE001
            extra["prestage"] = {"error": repr(e)}
        # ---- the hard case: data the rotated first pass exploits less well (never in `value`) ----------------------------
        # `value` is measured on SURVEY's rank-8 / 256-cluster mixture, whose energy sits in 31 directions.  Same size and graph
        # parameters on a rank-64 / 4096-cluster mixture (244 rows per cluster, four clusters to a pivot group; 28 % of the energy
        # outside the 31 directions: a looser first pass, which passes a third of the blocks on): same exact results.
        hard = {}
        try:
            Xh, _ = synth.mixture_device(n, d, dev, seed=43, n_clusters=4096, rank=64)
            settle_after_generator()
            A.build_laplacian_matrix(Xh, gp, ctx=ctx)   # warm
            ctx.synchronize()
            each_h, alloc_h = [], []
            alloc_keys = ("handle_cache_malloc_us", "handle_cache_free_us", "handle_cache_wait_us", "handle_cache_mallocs", "workspace_malloc_us", "workspace_mallocs")
            for _ in range(5):
                a0 = [ctx.stat(k) for k in alloc_keys]
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                t0 = time.perf_counter()
                glh = A.build_laplacian_matrix(Xh, gp, ctx=ctx)
                t_ret = time.perf_counter()
                ev1.record()
                ctx.synchronize()
                each_h.append(time.perf_counter() - t0)
                alloc_h.append({k: ctx.stat(k) - v for k, v in zip(alloc_keys, a0)})   # (what the allocator cost inside the call)
                alloc_h[-1]["gpu_timeline_ms"] = round(ev0.elapsed_time(ev1), 3)
                alloc_h[-1]["call_returned_after_ms"] = round((t_ret - t0) * 1e3, 3)
            t_hard = sorted(each_h)[2]   # (the median call: see config_leg)
            hard = {"value": n / t_hard, "unit": "items/s", "build_ms": t_hard * 1e3, "build_ms_each": [round(e * 1e3, 3) for e in each_h],
                    "allocator_inside_each": alloc_h,
                    "item_graph_nnz": int(glh.matrix.nnz()),
                    "data": "rank-64 / 4096-cluster Gaussian mixture (seed 43), same n, d and graph parameters; item graph only "
                            "(the lambda pass adds 0.4 ms)",
                    "path": {k: ctx.stat(k) for k in ("knn_join", "knn_join_prefilter", "knn_join_prefilter_rotated", "knn_join_row_order",
                                                      "knn_join_theta_grouped", "knn_join_prefilter_tail_e6", "knn_join_cone_tiles", "knn_join_cone_tiles_kept_e6", "knn_uncertified_rows",
                                                      "knn_uncertified_rows_stage1a")}}
            if ctx.stat("knn_join_prefilter") == 1:
                hard["path"]["blocks_passed_on_frac"] = ctx.stat("knn_join_prefilter_blocks_flagged") * 2.0 / max(1, ctx.stat("knn_join_prefilter_blocks"))
            del Xh, glh
        except Exception as e:  # noqa: BLE001 -- an extra leg must not take the bench line down
            hard = {"error": repr(e)}
        extra["value_hard_data"] = hard
    if rank == 0 and world == 1 and not args.no_config_legs and n == 1_000_000 and d == 128:
        # ---- the other BASELINE.json configs and the geometry sweep (never in `value`: that stays configs[2]) --------------
        # configs[1] 100 k x 128 (MFMA distance block + Rayleigh kernel); configs[3] 4 M x 128 as ONE GPU runs it (the baseline shards
        # it over 8: that line is `--gpus 8`); configs[4] 1 M x 768, Percentile(0.95) tau, 100 k batched prepare + search.
        if True:
            for name, kw in (("value_cfg1", dict(n=100_000, d=128, nq=10_000, taumode=A.TauMode.Median, reps=3)),
                             ("value_cfg3_one_gpu", dict(n=4_000_000, d=128, nq=10_000, taumode=A.TauMode.Median, reps=2)),
                             ("value_wide", dict(n=1_000_000, d=768, nq=100_000, taumode=A.TauMode.Percentile(0.95), reps=2))):
                try:
                    extra[name] = config_leg(A, D, synth, ctx, dev, k=args.k, alpha=args.alpha, **kw)
                except Exception as e:  # noqa: BLE001 -- an extra leg must not take the bench line down
                    extra[name] = {"error": repr(e)}
            sweep = []
            for rk, cl in ((8, 256), (16, 1024), (32, 2048), (64, 4096), (128, 4096)):
                try:
                    leg = config_leg(A, D, synth, ctx, dev, n=n, d=d, nq=0, taumode=A.TauMode.Median, k=args.k, alpha=args.alpha, rank=rk, clusters=cl,
                                     reps=3, seed=43 if rk > 8 else 42, search=False, lambdas=False)
                    sweep.append({"rank": rk, "clusters": cl, "item_graph_ms": leg["item_graph_ms"], "items_per_s": n / (leg["item_graph_ms"] * 1e-3),
                                  "item_graph_ms_each": leg["item_graph_ms_each"], "dominant_stage": leg["dominant_stage"], "candidates_per_row": leg["candidates_per_row"],
                                  "rows_fetched_per_row": leg["rows_fetched_per_row"],
                                  "bounds_on": {"cone_bound_on_tile_pairs": leg["path"]["knn_join_cone_tiles"],
                                                "tile_pairs_kept": leg["path"]["knn_join_cone_tiles_kept_e6"] * 1e-6,
                                                "grouped_thresholds": leg["path"]["knn_join_theta_grouped"],
                                                "rotated_first_pass": leg["path"]["knn_join_prefilter_rotated"],
                                                "energy_outside_the_31_directions": leg["path"]["knn_join_prefilter_tail_e6"] * 1e-6,
                                                "pivots_of_the_row_order": leg["path"]["knn_join_order_pivots"],
                                                "rounds_of_further_pivots": leg["path"]["knn_join_order_cover_rounds"],
                                                "rows_far_from_the_first_pivots": leg["path"]["knn_join_order_cover_low_e3"] * 1e-3,
                                                "rows_given_whole_sample_thresholds": leg["path"]["knn_join_theta_far_rows"]
                                                + leg["path"]["knn_join_theta_dip_rows"]}})
                except Exception as e:  # noqa: BLE001
                    sweep.append({"rank": rk, "clusters": cl, "error": repr(e)})
            extra["geometry_sweep"] = {"what": "item graph of 1 M x 128 mixtures (noise 0.3 sqrt(rank) per coordinate) of growing rank and cluster count: "
                                               "which bounds stay on, how many pivots the row order ends up with (it draws more where the first "
                                               "1 024 leave rows far from any), and what the build costs",
                                       "legs": sweep}
    if rank == 0 and world == 1 and not args.no_extra_legs:
        try:
            extra["published_equivalents"] = published_equivalents(A, ctx)
        except Exception as e:  # noqa: BLE001
            extra["published_equivalents"] = {"error": repr(e)}
        # ---- every bound off: the dense f16 join over all pairs of super tiles (the floor the bounds are measured against) -------
        try:
            saved = {}
            for opt, val in (("join_prefilter", 0), ("join_cone_tiles", 0), ("join_row_order", 0), ("join_theta_band", 0)):
                saved[opt] = val
                ctx.set_option(opt, val)
            try:
                A.build_laplacian_matrix(Xd, gp, ctx=ctx)
                ctx.synchronize()
                ctx.profile(True)
                t0 = time.perf_counter()
                gl_off = A.build_laplacian_matrix(Xd, gp, ctx=ctx)
                ctx.synchronize()
                t_off = time.perf_counter() - t0
                pl = ctx.profile_all()
                ctx.profile(False)
                join_ms = pl.get("knn_mfma_join_f16", (0.0, 0))[0]
                extra["value_all_bounds_off"] = {
                    "value": n / t_off, "unit": "items/s", "build_ms": t_off * 1e3, "item_graph_nnz": int(gl_off.matrix.nnz()),
                    "join_f16_ms": join_ms, "join_blocks": ctx.stat("knn_join_blocks"),
                    "join_TFLOPs": 2.0 * 32 * 32 * 128 * ctx.stat("knn_join_blocks") / max(join_ms, 1e-9) / 1e9,
                    "roofline": {"bound": "mfma", "kernel": "pair_join_f16_kernel<8, 0, false> (the dense form: every unordered pair of 512-row super tiles, "
                                                            "2 n^2 d / 2 flop + the diagonal, listing included)",
                                 "achieved": 2.0 * 32 * 32 * 128 * ctx.stat("knn_join_blocks") / max(join_ms, 1e-9) / 1e9, "peak": MFMA16_PEAK_TFLOPS,
                                 "unit": "TFLOP/s", "frac": 2.0 * 32 * 32 * 128 * ctx.stat("knn_join_blocks") / max(join_ms, 1e-9) / 1e9 / MFMA16_PEAK_TFLOPS,
                                 "traffic": None, "avg_launch_ms": join_ms},
                    "candidates_per_row": ctx.stat("knn_join_candidates") / max(1, n),
                    "what": "row order, cone bound, int8 first pass and band thresholds off: the f16 join contracts every unordered pair of "
                            "512-row super tiles (item graph only; the same table)"}
                del gl_off
            finally:
                for opt, dflt in (("join_prefilter", 1), ("join_cone_tiles", 1), ("join_row_order", 1), ("join_theta_band", 2)):
                    ctx.set_option(opt, dflt)
        except Exception as e:  # noqa: BLE001
            extra["value_all_bounds_off"] = {"error": repr(e)}
    if rank == 0 and world == 1 and args.rank_of_8_leg and n == 1_000_000 and d == 128:
        # ---- what ONE rank of configs[3] (4 M x 128 row-sharded over 8 GPUs) executes, measured on this GPU ----------------------
        # scripts/rank_of_8.py: eight processes share this GPU and run as_build_laplacian_sharded over a file transport while rank 3
        # records what it receives; then one process replays rank 3 with every collective answered from HBM on the library's stream.
        # Child processes (never an exec from this one); the line must survive their failure.
        try:
            # (the eight recording processes need ~15 GB each: what this process cached during the legs above goes back to the device first)
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            ctx.set_option("release_workspace", 1)
            ctx.set_option("handle_cache_mb", 0)
            ctx.set_option("handle_cache_mb", 2048)
            free_b, total_b = torch.cuda.mem_get_info(dev)
            xdir = "/dev/shm/as_rank_of_8_%d" % os.getpid()
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
                env.pop(k, None)
            script = os.path.join(ROOT, "scripts", "rank_of_8.py")
            t0 = time.perf_counter()
            rec = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                                  "--master-port", str(29600 + os.getpid() % 300), script, "record", "--rows", "4000000", "--dir", xdir],
                                 env=env, capture_output=True, text=True, timeout=600)
            t_rec = time.perf_counter() - t0
            if rec.returncode != 0:
                tb = [ln for ln in rec.stderr.splitlines() if "Error" in ln or "error" in ln or "File \"" in ln]
                raise RuntimeError("record: " + " | ".join(tb[:12])[:1500])
            outp = os.path.join(xdir, "replay.json")
            rep = subprocess.run([sys.executable, script, "replay", "--rows", "4000000", "--dir", xdir, "--reps", "5", "--out", outp],
                                 env=env, capture_output=True, text=True, timeout=600)
            if rep.returncode != 0:
                raise RuntimeError("replay: " + rep.stderr[-400:])
            with open(outp) as f:
                leg = json.load(f)
            leg["leg_wall_s"] = time.perf_counter() - t0
            leg["record_wall_s"] = t_rec
            leg["record_says"] = [ln for ln in rec.stdout.splitlines() if ln.startswith("recorded")][:1]
            leg["device_memory_free_GB_before_the_leg"] = round(free_b / 1e9, 1)
            extra["value_cfg3_rank_of_8"] = leg
        except Exception as e:  # noqa: BLE001
            extra["value_cfg3_rank_of_8"] = {"error": repr(e)[:2000]}
        finally:
            import shutil
            shutil.rmtree(xdir, ignore_errors=True)
    if rank == 0 and world == 1 and not args.no_spmm_leg:
        # ---- the item-graph lambda: the feature columns as signals over the N x N Laplacian the build returns ------------
        # (SURVEY.md 8(a') item-graph mapping, the "SpMV" of north_star; src/taumode.rs:552-660 per signal.)  Algorithmic bytes
        # (SURVEY.md 8(d)): nnz * 16 for the CSR entries + N * D * 8 for the item rows read once; the gathers of the rows a
        # stored pair names come on top and are served on-die where the row order keeps neighbours together.
        try:
            from arrowspace_amd.api import compute_taumode_signals
            gl_sp = A.build_laplacian_matrix(Xd, gp, ctx=ctx)
            nnz_sp = gl_sp.matrix.nnz()
            legs_sp = {}
            for name, tm in (("fixed_tau", A.TauMode.Fixed(0.5)), ("median_tau", A.TauMode.Median)):
                t_call = 0.0
                for it in range(5):
                    if it == 1:
                        ctx.profile(True)
                    t0c = time.perf_counter()
                    lam_sp = compute_taumode_signals(Xd, gl_sp.matrix, tm, ctx=ctx)
                    ctx.synchronize()
                    t_call += (time.perf_counter() - t0c) if it >= 1 else 0.0
                pl = ctx.profile_all()
                ctx.profile(False)
                legs_sp[name] = {k.replace("signal_", "") + "_ms": v[0] / v[1] for k, v in pl.items() if k.startswith("signal_") and v[1]}
                legs_sp[name]["call_ms"] = t_call / 4 * 1e3   # wall time of the whole call (with Median the order statistics run under the sums pass)
                legs_sp[name]["tau_under_the_sums_pass"] = ctx.stat("signal_tau_overlapped")
            # (the sums kernel by itself: in the Median leg the bracket pass of the order statistics shares the memory system with it)
            sums_ms = legs_sp["fixed_tau"]["sums_ms"]
            alg_bytes = nnz_sp * 16.0 + float(n) * d * 8.0
            records = ctx.stat("signal_stream_records")
            valu_floor_ms = records * 7.0 * d / (16.0 * 4 * 256 * 2.4e9) * 1e3   # 16 f64 lanes per clock and SIMD, 1024 SIMDs, 2.4 GHz
            extra["roofline_spmm"] = {
                "bound": "hbm", "kernel": "signal_sums_kernel<false, 2> (item-graph lambda: the upper triangle of the N x N Laplacian "
                                          "streamed once for all D signals, each stored pair gathers the two 1 KB item rows; HIP events "
                                          "around the launch)",
                "achieved": alg_bytes / (sums_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": alg_bytes / (sums_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_bytes": alg_bytes, "avg_launch_ms": sums_ms, "item_graph_nnz": int(nnz_sp), "stream_records": int(records),
                "gathered_bytes": records * 8.0 * d, "gather_rate_GBs": records * 8.0 * d / (sums_ms * 1e-3) / 1e9,
                "f64_valu_floor_ms": valu_floor_ms, "frac_of_f64_valu_bound": valu_floor_ms / sums_ms,
                "row_order": "the join's own (attached to the graph by as_build_laplacian)" if ctx.stat("signal_stream_ordered") else "as stored",
                "legs_ms": legs_sp, "lambda_min_max": [float(lam_sp.min()), float(lam_sp.max())],
                "note": "frac divides the algorithmic bytes (each row once) by the launch time; the pass requests gathered_bytes from "
                        "the caches (a pair's second row), which at this graph (neighbours random inside a 4 MB cluster, as large as "
                        "an XCD's L2) come mostly from the Infinity Cache; median_tau adds the order statistic of every column: "
                        "8 192 sampled values bracket it, one pass over the N x D matrix counts below the bracket and lists what is inside, "
                        "on a side stream under the sums pass (legs_ms.median_tau.call_ms against legs_ms.fixed_tau.call_ms is what it costs)"}
            rec = recorded_traffic("signal_sums_kernel", n, d)
            if rec:
                # (this leg launches the kernel several times in the profiled pass: bytes per single launch)
                extra["roofline_spmm"]["traffic"] = rec["bytes_per_single_launch"]
                extra["roofline_spmm"]["traffic_stale"] = rec["stale"]
                extra["roofline_spmm"]["traffic_source"] = {"file": rec["source"], "commit": rec["commit"], "source_sha16": rec["source_sha16"],
                                                            "what": "bytes that left the L2s (TCC fabric reads x the guide's factor 2, + writes): "
                                                                    "the gathers the L2 misses, most of them served by the Infinity Cache, not HBM"}
            del gl_sp
        except Exception as e:  # noqa: BLE001
            extra["roofline_spmm"] = {"error": repr(e)}

    if rank == 0:
        steps = args.steps
        build_s = t_build / steps
        search_s = t_search / steps
        rows_here = hi - lo
        # ---- every long kernel of the build against ITS OWN pipe: executed work / measured time / peak ----------------
        # (HIP events on the library's stream around each launch group; `roofline` is the one with the most time.)
        # The reference's all-pairs loop (2 n^2 d flop) is NOT what these kernels execute -- the first pass
        # rules most of it out -- so that figure lives in `reference_equivalent`, never in a `frac`.
        names = {16: "knn_mfma_join_f16", 48: "knn_mfma_join_bf16", 32: "knn_mfma_join_f32"}
        pipe = stats["knn_join_pipe"]
        d_pad = (d + 15) // 16 * 16
        kernels = []

        def add(key, kernel, bound, work, peak, unit, what, extra_fields=None, pmc=None):
            ms, launches = prof.get(key, (0.0, 0))
            if not launches or ms <= 0.0:
                return
            avg_ms = ms / steps   # per step: a group may hold several launches of a step (prune + score, items + queries)
            ach = work / (avg_ms * 1e-3) / (1e9 if unit == "GB/s" else 1e12)
            e = {"bound": bound, "kernel": kernel, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
                 "traffic": None, "avg_launch_ms": avg_ms, "launches_per_step": launches / steps, "work_per_launch": work,
                 "work_is": what, "profile_group": key}
            if extra_fields:
                e.update(extra_fields)
            if world == 1:
                rec = recorded_traffic(pmc or kernel.split(" ")[0].split("<")[0], n, d)
                if rec:   # PMC counters cannot be read from inside this process: the committed rocprofv3 --pmc passes of the
                    e["traffic"] = rec["bytes_per_launch"]   # same command (scripts/profile_round.sh), per launch
                    if e.get("gather") and rec["bytes_per_launch"]:
                        # a gather kernel whose rows are shared through the L2s REQUESTS more than the memory system can deliver from
                        # HBM (frac of the requested bytes above 1 is no utilisation): its rate is what left the L2s over its time
                        e["requested_GBs"] = ach
                        e["achieved"] = rec["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9
                        e["frac"] = e["achieved"] / peak
                        e["achieved_is"] = "counter bytes (L2 <-> fabric, profiles/) over the launch time measured here"
                    e["traffic_source"] = {"file": rec["source"], "commit": rec["commit"], "source_sha16": rec["source_sha16"]}
                    e["traffic_stale"] = rec["stale"]
            kernels.append(e)

        if stats["knn_join"] == 1:
            blocks = stats["knn_join_blocks"]
            if stats.get("knn_join_prefilter", 0) == 1:
                rotated = stats.get("knn_join_prefilter_rotated", 0) == 1
                k8 = 32 if rotated else (d + 31) // 32 * 32
                cone = stats.get("knn_join_cone_tiles", 0) == 1
                if prof.get("knn_join_cone_tiles", (0.0, 0))[1]:
                    nt = (n + 127) // 128
                    add("knn_join_cone_tiles",
                        "cone_max_kernel (+ cone_tile_kernel, cone_and_kernel: v_mfma_f32_32x32x16_f16, the rows of every 128-position tile "
                        "against the mean direction of every tile; a bit per pair of tiles that can hold a pair of rows at a threshold)",
                        "mfma", 2.0 * n * nt * 32, MFMA16_PEAK_TFLOPS, "TFLOP/s",
                        "f16 MFMA flops executed (n rows x n/128 directions x 32 coordinates)",
                        {"tile_pairs_kept_frac": stats.get("knn_join_cone_tiles_kept_e6", 0) * 1e-6, "first_pass_behind_it": bool(cone)})
                add("knn_join_prefilter",
                    "pair_prefilter_i8%s_kernel (v_mfma_i32_32x32x32_i8 over %s on %s; "
                    "a bit per 64 x 32 block that can hold a pair)" % ("_narrow" if k8 <= 64 else "",
                                                                          "the pairs of 128-position tiles the cone bound keeps" if cone
                                                                          else "every unordered pair of 512-row super tiles",
                                                                          "the rotated 32-wide int8 image (31 directions + tail norm)"
                                                                          if rotated else "the full-width int8 image of the unit rows"),
                    "mfma", 2.0 * 32 * 32 * k8 * stats["knn_join_prefilter_blocks"], INT8_PEAK_TOPS, "TFLOP/s",
                    "int8 multiply-adds x 2 the MFMAs execute (blocks visited x 32 x 32 x k); peak = dense int8 MFMA",
                    {"blocks_passed_on_frac": stats["knn_join_prefilter_blocks_flagged"] * 2.0 / max(1, stats["knn_join_prefilter_blocks"])})
            if pipe == 16:
                add(names[16], "pair_join_f16_kernel (v_mfma_f32_32x32x16_f16 on one f16 plane of the unit rows; symmetric threshold join "
                               "over the blocks the first pass leaves, listing (row, column, score))",
                    "mfma", 2.0 * 32 * 32 * d_pad * blocks, MFMA16_PEAK_TFLOPS, "TFLOP/s", "f16 MFMA flops executed (blocks contracted x 32 x 32 x d)",
                    pmc="pair_join_f16_kernel<%d, 0, true," % (d_pad // 16) if stats.get("knn_join_prefilter", 0) == 1
                    else "pair_join_f16_kernel<%d, 0, false," % (d_pad // 16))
            elif pipe == 48:
                add(names[48], "pair_join_bf16_kernel (split bf16, two-phase blocks)", "mfma",
                    2.0 * 32 * 32 * d_pad * (blocks + 2.0 * stats["knn_mfma_join_bf16_blocks_full"]), MFMA16_PEAK_TFLOPS, "TFLOP/s",
                    "bf16 MFMA flops executed")
            else:
                t2 = (n + 255) // 256
                add(names[32], "pair_join_mfma32_kernel (v_mfma_f32_32x32x2_f32)", "mfma",
                    2.0 * (t2 * ((t2 - 1) // 2) + t2 + (t2 // 2 if t2 % 2 == 0 else 0)) * 256 * 256 * d, F32_MFMA_PEAK_TFLOPS, "TFLOP/s",
                    "f32 MFMA flops executed")
            scored = stats["knn_rerank_scored"]
            batches = stats.get("knn_rerank_batches", 0)
            if scored > 0 and batches > 0:
                # exact re-rank (rerank_split.h, rerank_pairs.h).  Its kernels one by one, then the whole stage as a group:
                #   scoring  every slot of the stream is a 1 KB row fetched (16 per batch, own rows and padding included) + 20 B of
                #            descriptor per slot + one 8 B key written per pair and side
                #   pruning  the lists read (8 B per listed slot), the records written (12 B per survivor + header)
                pairs = stats.get("knn_rerank_pairs_scored", 0)
                score_bytes = batches * 16.0 * (8.0 * d + 20.0) + pairs * 16.0
                add("knn_rerank_score", "rerank_pairs_score_kernel (the slot stream: batches of 16 rows packed across rows, every unordered pair once; "
                                        "counted vector loads only, products in registers, 16 sequential f64 chains per batch)",
                    "hbm", score_bytes, HBM_PEAK_GBS, "GB/s", "bytes REQUESTED: 16 rows of 8 d bytes per batch + descriptors + keys (rows of one neighbourhood are "
                    "visited side by side on an L2, so part is served on-die: `traffic` is what left the L2s)",
                    {"rows_fetched_per_row": batches * 16.0 / max(1, rows_here), "pairs_scored_per_row": pairs / max(1, rows_here),
                     "survivors_per_row": scored / max(1, rows_here), "batches": batches, "gather": True,
                     # per batch 128 sequential v_add_f64 on 16 lanes (the reference's summation order) + 2048 products on 64 lanes;
                     # a wave-wide f64 instruction occupies its SIMD for 4 cycles; 1024 SIMDs at 2.4 GHz
                     "f64_issue_ms": batches * (128.0 + 2048.0 / 64.0) * 4.0 / (1024.0 * 2.4e9) * 1e3,
                     "f64_issue_is": "time the f64 instructions of the chains and products occupy the SIMDs (divide by avg_launch_ms for the "
                                     "issue fraction: the kernel is bound by the gather, not by the chains)",
                     "achieved_is": "REQUESTED bytes over the launch time (no counter pass of this binary at hand): rows shared through the "
                                    "L2s make this exceed what HBM delivers"}, pmc="rerank_pairs_score_kernel")
                add("knn_rerank_prune", "rerank_prune_kernel (lists -> per-pair bounds -> survivors: a record per row)", "hbm",
                    stats["knn_join_candidates"] * 12.0 + scored * 12.0 + rows_here * 16.0, HBM_PEAK_GBS, "GB/s",
                    "lists (8 B per listed slot + a 4 B plane error gathered) read, records (ids + norms) written; instruction-bound (bisection "
                    "for the K-th lower bound, compaction)", pmc="rerank_prune_kernel")
                rr_bytes = score_bytes + stats["knn_join_candidates"] * 12.0 + scored * 12.0 + rows_here * (16.0 + 17 * 12.0) + scored * 40.0
                add("knn_rerank", "exact re-rank, all five kernels (rerank_prune_kernel, rerank_link_kernel, rerank_fill_kernel, rerank_pairs_score_kernel, "
                                  "rerank_rank_kernel)",
                    "hbm", rr_bytes, HBM_PEAK_GBS, "GB/s", "bytes requested by the five kernels together", {"group": True},
                    pmc=("rerank_pairs_score_kernel", "rerank_prune_kernel", "rerank_link_kernel", "rerank_fill_kernel", "rerank_rank_kernel"))
        ms32, l32 = prof.get("knn_mfma_candidates_f32", (0.0, 0))
        ms64, l64 = prof.get("knn_mfma_candidates", (0.0, 0))
        flops_alg = 2.0 * rows_here * n * d  # 2 n^2 d over the rows this rank owns
        if stats["knn_join"] != 1 and l32 and ms32 >= ms64:
            add("knn_mfma_candidates_f32", "pair_topk_mfma32_kernel (v_mfma_f32_32x32x2_f32; exact f64 re-rank + certification follow)", "mfma",
                flops_alg, F32_MFMA_PEAK_TFLOPS, "TFLOP/s", "2 n^2 d (the full square is contracted)")
        elif stats["knn_join"] != 1 and l64:
            add("knn_mfma_candidates", "pair_topk_mfma_kernel (v_mfma_f64_16x16x4_f64)", "mfma", flops_alg, F64_MFMA_PEAK_TFLOPS, "TFLOP/s",
                "2 n^2 d (the full square is contracted)")
        nnz = float(state.get("nnz", 0))
        if nnz > 0 and world == 1:
            add("graph_assembly", "graph assembly (degree_rows64_kernel, adjacency32_kernel, fill_reverse32_kernel, union_count32r_kernel, write_rows_ranked_kernel, write_rows_mid / _hub_kernel)",
                "hbm", rows_here * 17 * 16.0 + nnz * 16.0 + (rows_here + 1) * 8.0, HBM_PEAK_GBS, "GB/s",
                "SURVEY.md 8(d): n (t+1) 16 B in + nnz 16 B + (n+1) 8 B out", {"group": True},
                pmc=("degree_kernel", "degree32_kernel", "degree_rows64_kernel", "adjacency_kernel", "adjacency32_kernel", "fill_reverse_kernel",
                     "fill_reverse32_kernel", "union_count_kernel", "union_count32_kernel", "union_count32r_kernel", "write_rows_kernel",
                     "write_rows_big_kernel", "write_rows_ranked_kernel", "write_rows_mid_kernel", "write_rows_hub_kernel", "write_rows_serial_kernel",
                     "scan_block_sums", "scan_of_sums", "scan_finish"))
        # `roofline`: the single kernel with the most time (groups of kernels -- the re-rank as a whole, the graph assembly -- are listed in
        # roofline_kernels but do not stand for "the dominant kernel")
        singles = [e for e in kernels if not e.get("group")]
        roofline = max(singles, key=lambda e: e["avg_launch_ms"]) if singles else None
        reference_equivalent = None
        if stats["knn_join"] == 1:
            join_ms = sum(e["avg_launch_ms"] for e in kernels if e["profile_group"] in ("knn_join_prefilter", "knn_join_cone_tiles") + tuple(names.values()))
            if join_ms > 0:
                reference_equivalent = {
                    "what": "SURVEY.md 8(d) prices the neighbour search at the reference's all-pairs loop, 2 n^2 d flop; the launches that "
                            "replace it (cone bound on tile pairs + int8 first pass + f16 pass) take join_ms. This is how much of that loop is never executed, NOT a "
                            "utilisation: no frac is derived from it",
                    "algorithmic_flops": flops_alg, "join_ms": join_ms, "equivalent_TFLOPs": flops_alg / (join_ms * 1e-3) / 1e12}
        lam_ms, lam_n = prof.get("taumode_lambda", (0.0, 0))
        if lam_n:
            # two launches per step hit this group: the N items and the nq queries
            bytes_per_step = (rows_here + nq) * (8.0 * d + 8.0)
            gbps = bytes_per_step * steps / (lam_ms * 1e-3) / 1e9
            extra["roofline_scan"] = {"bound": "hbm", "kernel": "lambda_kernel_v3 (tau select + Rayleigh/dispersion, items + queries)",
                                      "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBS,
                                      "traffic": None, "avg_step_ms": lam_ms / steps, "algorithmic_bytes_per_step": bytes_per_step}
            v3_steps = ctx.stat("taumode_v3_steps")
            if v3_steps > 0 and d == 128:
                # The kernel's other two limits (SURVEY.md 7 asks for the fraction against each).  Per item and Median: the four
                # lanes of an item walk v3_steps edge steps of 7 f64 operations each, 3 x 128 for the diagonal term and the
                # denominator, 4 x 638 min/max for the order statistic (32-sorter 191 compare-exchanges, the two merges, the
                # median's half-cleaner), ~30 to finish; f64 vector peak = CUs x 4 SIMDs x 16 lanes x clock (an FMA counts once).
                # LDS: every edge step reads a 16 B entry and two 8 B values per lane, the row is written once by the DMA and read
                # once into registers; peak 128 B per clock and CU.
                clock_hz, cus = 2.4e9, 256
                ops_item = 28.0 * v3_steps + 384.0 + 2552.0 + 30.0
                lds_item = 4.0 * v3_steps * 32.0 + 2.0 * 8.0 * d
                items_s = (rows_here + nq) * steps / (lam_ms * 1e-3)
                valu_peak, lds_peak = cus * 4 * 16 * clock_hz, cus * 128.0 * clock_hz
                extra["roofline_scan"]["edge_steps_per_lane"] = v3_steps
                extra["roofline_scan"]["f64_valu"] = {"ops_per_item": ops_item, "achieved_Tops": items_s * ops_item / 1e12, "peak_Tops": valu_peak / 1e12,
                                                      "frac": items_s * ops_item / valu_peak}
                extra["roofline_scan"]["lds"] = {"bytes_per_item": lds_item, "achieved_TBs": items_s * lds_item / 1e12, "peak_TBs": lds_peak / 1e12,
                                                 "frac": items_s * lds_item / lds_peak}
                # the limit that binds is the one the kernel is nearest to: its floor (time at that limit's peak) over the measured time
                limits = {"hbm": extra["roofline_scan"]["frac"], "lds": extra["roofline_scan"]["lds"]["frac"],
                          "f64_valu": extra["roofline_scan"]["f64_valu"]["frac"]}
                binding = max(limits, key=limits.get)
                extra["roofline_scan"]["binding_limit"] = binding
                extra["roofline_scan"]["frac_of_binding_limit"] = limits[binding]
                extra["roofline_scan"]["hbm_frac_at_the_binding_limits_peak"] = extra["roofline_scan"]["frac"] / limits[binding]
            if world == 1:
                rec = recorded_traffic("lambda_kernel", n, d)
                if rec:
                    extra["roofline_scan"]["traffic"] = rec["bytes_per_launch"]
                    extra["roofline_scan"]["traffic_source"] = {"file": rec["source"], "commit": rec["commit"], "source_sha16": rec["source_sha16"]}
                    extra["roofline_scan"]["traffic_stale"] = rec["stale"]
        ss_kernel = ("pair_join_f16_kernel<KS, 1> (threshold scan of the search: one-f16-plane products, coarse test on the cosine, "
                     "lambda term against per-query score guesses in the listing)")
        ss_ms, ss_n = prof.get("search_scan_f16", (0.0, 0))
        if not ss_n:
            ss_ms, ss_n = prof.get("search_scan_bf16", (0.0, 0))
            ss_kernel = ("pair_join_bf16_kernel<KS, 2> (threshold scan of the search: split-bf16 products + lambda term against "
                         "per-query score guesses)")
        if ss_n:
            # the threshold scan of the search (stage 0 of K9) at the benchmark batch: one launch per step over all nq queries
            # (+ a short second launch for the queries whose guess was too high)
            fl = 2.0 * nq * rows_here * d
            tf = fl * steps / (ss_ms * 1e-3) / 1e12
            behind_first_pass = ctx.stat("search_scan_prefilter") == 1
            if behind_first_pass:
                # the index adopted the images of the item-graph build (keep_join_images): the scan runs behind the int8 first pass and
                # contracts the flagged blocks only -- 2 nq N D over its time is how much of the all-pairs product is never executed,
                # NOT a utilisation (it exceeds the peak); no frac is derived from it
                ss_kernel = ("scan_slices_f16_kernel (threshold scan of the search behind the int8 first pass: a wave per 64 queries x one "
                             "column super tile, flagged blocks only) + the dense re-scan of the queries whose guess was too high")
                extra["roofline_search"] = {"bound": "mfma", "kernel": ss_kernel, "achieved": None, "peak": MFMA16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                            "frac": None, "traffic": None, "avg_step_ms": ss_ms / steps, "algorithmic_flops_per_step": fl,
                                            "equivalent_TFLOPs_not_a_utilisation": tf}
            else:
                extra["roofline_search"] = {"bound": "mfma", "kernel": ss_kernel,
                                            "achieved": tf, "peak": MFMA16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA16_PEAK_TFLOPS,
                                            "traffic": None, "avg_step_ms": ss_ms / steps, "algorithmic_flops_per_step": fl}
        out = {
            "metric": "lambda_tau_build_items_per_s", "value": n / build_s, "unit": "items/s",
            "candidate_precision": ("int8 first pass (blocks that can hold a pair) -> f16-plane MFMA symmetric threshold join on those (panelled f16 form, no "
                                    "first pass, beyond dim 128) -> second pass -> f32 top-k "
                                    "-> f64 MFMA -> exact f64 scan (results bit-identical to the exact path)")
            if args.cand_precision == 32 else "f64 MFMA -> exact f64 scan",
            "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": t_all / steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "configs[2]: %d x %d low-rank Gaussian mixture (seed 42), item-graph "
                                   "with_lambda_graph(eps=0.5,k=16,topk=16,p=2,sigma=0.25) + per-item lambda-tau "
                                   "(Median) + %d-query prepare+search_lambda_aware(k=%d, alpha=%.1f)"
                                   % (n, d, nq, args.k, args.alpha),
                       "n_items": n, "dim": d, "n_queries": nq, "sharding": "rows x %d" % world, "transport": transport},
            "search_qps": nq / search_s, "build_ms": build_s * 1e3, "search_ms": search_s * 1e3,
            "item_graph_nnz": int(state.get("nnz", -1)),
            "device_memory_used_GB_per_rank": device_memory_used_gb,
            "knn_uncertified_rows": stats["knn_uncertified_rows"], "knn_uncertified_rows_after_second_pass": stats["knn_uncertified_rows_stage1a"],
            "knn_join": stats["knn_join"], "knn_join_pipe": pipe, "knn_join_delta": stats["knn_join_delta_e9"] * 1e-9,
            "knn_join_candidates_per_row": stats["knn_join_candidates"] / max(1, rows_here),
            "search_uncertified_rows": stats["search_uncertified_rows"],
            **({"sharded": {"join_sharded_by_positions": ctx.stat("sharded_by_positions"),
                            "prepared_from_own_rows": ctx.stat("sharded_prepare_by_rows"),
                            "f64_rows_gathered_under_the_join": ctx.stat("sharded_rows_late"),
                            "foreign_share_of_listed_slots": ctx.stat("sharded_foreign_slots") / max(1, ctx.stat("sharded_listed_slots"))}}
               if world > 1 else {}),
            "kernels_ms_per_step": {k: v[0] / steps for k, v in sorted(prof.items())},
            "commit": git_head(),
            "roofline": roofline,
            "roofline_kernels": kernels,
            "reference_equivalent": reference_equivalent,
        }
        out.update(extra)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, lambda a, b: Xd[a:b].cpu().numpy(), cent, n)
            pub_cpu = out["cpu_baseline"].pop("published_equivalents", {})
            if isinstance(out.get("published_equivalents"), dict):
                for key, v in pub_cpu.items():
                    out["published_equivalents"].setdefault(key, {}).update(v)
        # The full record goes to bench_detail.json; stdout carries ONE small line (bench_line.py: below 6 KB, strict JSON) -- round 5's
        # 23 KB line could not be parsed by the driver.
        out["detail_file"] = "bench_detail.json"
        bench_line.write_detail(out, ROOT)
        sys.stdout.flush()
        print(bench_line.compact_line(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
