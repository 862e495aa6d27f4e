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
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# v_mfma_f64_16x16x4_f64 issues every 128 cycles per SIMD on MI355X (measured, scripts/micro/mfma_f64_peak.hip,
# profiles/r01_mfma_f64_issue_rate.txt): 2048 flop / 128 cyc x 1024 SIMDs x 2.4 GHz.  MI355X_MICROARCH.md has no
# f64 MFMA row; the datasheet's FP64-matrix figure is 78.6 TF (reported next to it as frac_of_datasheet_78p6).
F64_MFMA_PEAK_TFLOPS = 39.3
F64_MATRIX_DATASHEET_TFLOPS = 78.6
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 (v_mfma_f32_32x32x16_bf16: 32768 flop / 32 clk / SIMD at 2.4 GHz)
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 flop/clk/SIMD (155 TF measured)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md chip table


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--n", type=int, default=1_000_000)
    p.add_argument("--dim", type=int, default=128)
    p.add_argument("--nq", type=int, default=10_000)
    p.add_argument("--k", type=int, default=10)
    p.add_argument("--alpha", type=float, default=0.7)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cand-precision", type=int, default=32, choices=[32, 64],
                   help="first candidate stage: 32 = f32 MFMA then f64 MFMA then exact (default); 64 = f64 MFMA first")
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
    return {
        "value": n_total / t_full, "unit": "items/s", "cores": O.num_threads(), "kind": "port",
        "search_qps": qps_full,
        "measured_sample_items_per_s": ns / (t_graph + t_lam),
        "sample": "oracle (C restatement, OpenMP %d threads) on the first %d rows of the same mixture: item graph "
                  "%.2fs + lambdas %.3fs, %d queries %.2fs; value/search_qps are the n^2 (graph) and n (lambda, scan) "
                  "extrapolations to N=%d" % (O.num_threads(), ns, t_graph, t_lam, nqs, t_search, n_total),
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ARROWSPACE_DIST_FORCE=1 under a one-rank torch.distributed.run: the RCCL path on a one-GPU box
    use_dist = world > 1 or (os.environ.get("ARROWSPACE_DIST_FORCE") == "1" and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import arrowspace_amd as A
    from arrowspace_amd import distributed as D
    from arrowspace_amd import synth

    ctx = A.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx._stream = torch.cuda.current_stream().cuda_stream
    ctx.set_option("cand_precision", args.cand_precision)
    ops = D.HipOps(ctx)

    n, d, nq = args.n, args.dim, args.nq
    lo, hi = D.shard_range(n, rank, world)
    X_host, cent = synth.mixture(hi - lo, d, seed=42, row_start=lo)
    Xd = torch.from_numpy(X_host).to(dev)
    qidx = synth.query_indices(n, nq, seed=42)
    Q = torch.from_numpy(synth.mixture_at(qidx, d, seed=42) * 1.02).to(dev)
    gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
    sas = D.ShardedArrowSpace(ops, Xd, n, gp, A.TauMode.Median)

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
    times = torch.tensor([t_all, t_build, t_search], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
    t_all, t_build, t_search = [float(x) for x in times]
    prof = ctx.profile_all()
    ctx.profile(False)

    if rank == 0:
        steps = args.steps
        build_s = t_build / steps
        search_s = t_search / steps
        rows_here = hi - lo
        flops_per_launch = 2.0 * rows_here * n * d  # algorithmic: 2*n^2*d over the rows this rank owns
        roofline = None
        ms32, l32 = prof.get("knn_mfma_candidates_f32", (0.0, 0))
        ms64, l64 = prof.get("knn_mfma_candidates", (0.0, 0))
        msj, lj = prof.get("knn_mfma_join_f32", (0.0, 0))
        msb, lb = prof.get("knn_mfma_join_bf16", (0.0, 0))
        # the candidate pass of one step may be split into a bulk launch and a remainder launch:
        # "launch" below = all candidate launches of one step (their flops add up to flops_per_launch)
        l32, l64 = min(l32, steps), min(l64, steps)
        # the symmetric threshold join.  cos(i,j) = cos(j,i): every unordered pair of 256-row tiles is contracted once,
        # so the launch EXECUTES half of the algorithmic 2 n^2 d (plus the diagonal blocks); `achieved` / `frac` price
        # the kernel on the MFMA flops it executes, the reference-equivalent rate (what the reference's full square would
        # need at 2 flops per multiply-add) is given next to it
        t2 = (n + 255) // 256
        pairs = t2 * ((t2 - 1) // 2) + t2 + (t2 // 2 if t2 % 2 == 0 else 0)   # 256 x 256 blocks visited
        if lb and msb >= max(ms32, ms64, msj):
            # split-bf16 contraction in two phases per 32x32 block: the hi planes of every block (KS MFMAs of
            # 32x32x16), the two cross terms (2 KS more) only for the blocks the hi-only products cannot rule out.
            # `achieved` / `frac` price the kernel on the MFMA flops it really executes (block counts reported by the
            # kernel itself); what the full three-term contraction of every visited block would take, and what the
            # reference's full square takes, are given next to it.
            d_pad = (d + 15) // 16 * 16
            blocks = ctx.stat("knn_mfma_join_bf16_blocks")
            full = ctx.stat("knn_mfma_join_bf16_blocks_full")
            per_block_phase = 2.0 * 32 * 32 * d_pad
            executed = per_block_phase * (blocks + 2.0 * full) if blocks else 3 * 2.0 * pairs * 256 * 256 * d_pad
            achieved = executed / (msb / lb * 1e-3) / 1e12
            roofline = {"bound": "mfma", "kernel": "pair_join_bf16_kernel (v_mfma_f32_32x32x16_bf16 on split operands: hi x hi for every "
                                                   "32x32 block, the cross terms for the blocks it cannot rule out; f32 acc; symmetric "
                                                   "threshold join; exact f64 re-rank + certification follow)",
                        "achieved": achieved, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / BF16_MFMA_PEAK_TFLOPS, "traffic": None,
                        "avg_launch_ms": msb / lb, "launches": lb, "executed_flops_per_launch": executed,
                        "blocks_32x32": blocks, "blocks_contracted_in_full": full,
                        "note": "two-phase blocks: the hi planes of every 32x32 block, the cross terms only where a product comes "
                                "within eps of a threshold, so most of the three-term MFMA work is never issued and the kernel is "
                                "bound by what surrounds the MFMAs (LDS-DMA issue, tile barriers, the coarse test: DESIGN.md section 4, "
                                "profiles/r01i_pmc_clock.txt); the three-term form ran at 52 % of this peak, within 8 % of the "
                                "power-limited rate, and took 292 ms",
                        "three_term_flops_of_the_visited_blocks": 3 * 2.0 * pairs * 256 * 256 * d_pad,
                        "three_term_equivalent_TFLOPs": 3 * 2.0 * pairs * 256 * 256 * d_pad / (msb / lb * 1e-3) / 1e12,
                        "algorithmic_flops_per_launch": flops_per_launch,
                        "reference_equivalent_TFLOPs": flops_per_launch / (msb / lb * 1e-3) / 1e12}
        elif lj and msj >= max(ms32, ms64):
            executed = 2.0 * pairs * 256 * 256 * d
            achieved = executed / (msj / lj * 1e-3) / 1e12
            roofline = {"bound": "mfma", "kernel": "pair_join_mfma32_kernel (v_mfma_f32_32x32x2_f32, f32 in / f32 acc, symmetric "
                                                   "threshold join; exact f64 re-rank + certification follow)",
                        "achieved": achieved, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / F32_MFMA_PEAK_TFLOPS, "traffic": None,
                        "avg_launch_ms": msj / lj, "launches": lj, "executed_flops_per_launch": executed,
                        "algorithmic_flops_per_launch": flops_per_launch,
                        "reference_equivalent_TFLOPs": flops_per_launch / (msj / lj * 1e-3) / 1e12}
        elif l32 and ms32 >= ms64:   # dominant kernel: the f32-input MFMA candidate pass
            achieved = flops_per_launch / (ms32 / l32 * 1e-3) / 1e12
            roofline = {"bound": "mfma", "kernel": "pair_topk_mfma32_kernel (v_mfma_f32_32x32x2_f32, f32 in / f32 acc; "
                                                   "exact f64 re-rank + certification follow)",
                        "achieved": achieved, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / F32_MFMA_PEAK_TFLOPS, "traffic": None,
                        "avg_launch_ms": ms32 / l32, "launches": l32}
        elif l64:
            achieved = flops_per_launch / (ms64 / l64 * 1e-3) / 1e12
            roofline = {"bound": "mfma", "kernel": "pair_topk_mfma_kernel (f64 v_mfma_f64_16x16x4_f64)",
                        "achieved": achieved, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / F64_MFMA_PEAK_TFLOPS, "traffic": None,
                        "frac_of_datasheet_78p6": achieved / F64_MATRIX_DATASHEET_TFLOPS,
                        "avg_launch_ms": ms64 / l64, "launches": l64}
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the
        # committed rocprofv3 --pmc measurement of this very command is reported when it matches the workload
        pmc = {}
        try:
            with open(os.path.join(ROOT, "profiles", "r01i_pmc_traffic.json")) as f:
                pmc = json.load(f)
            if roofline and world == 1 and (n, d) == (1_000_000, 128) and pmc.get("dominant_kernel", "pair_topk_mfma32") in roofline["kernel"]:
                roofline["traffic"] = pmc["hbm_bytes_per_step"]
                roofline["traffic_source"] = "profiles/r01i_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
        except (OSError, KeyError, ValueError):
            pmc = {}
        lam_ms, lam_n = prof.get("taumode_lambda", (0.0, 0))
        extra = {}
        if lam_n:
            # two launches per step hit this group: the N items and the nq queries
            bytes_per_step = (rows_here + nq) * (8.0 * d + 8.0)
            gbps = bytes_per_step * steps / (lam_ms * 1e-3) / 1e9
            extra["taumode_scan_GBps"] = gbps
            extra["taumode_scan_frac_hbm"] = gbps / HBM_PEAK_GBS
            # second roofline object: the lambda-tau scan (K6/K7/K8), HBM-side view.  The kernel itself is
            # bound by f64 vector issue + LDS reads at this graph density.
            extra["roofline_scan"] = {"bound": "hbm", "kernel": "lambda_kernel_v3 (tau select + Rayleigh/dispersion, items + queries)",
                                      "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBS,
                                      "traffic": None, "avg_step_ms": lam_ms / steps,
                                      "algorithmic_bytes_per_step": bytes_per_step}
            if world == 1 and (n, d, nq) == (1_000_000, 128, 10_000) and "lambda_scan_hbm_bytes_per_step" in pmc:
                extra["roofline_scan"]["traffic"] = pmc["lambda_scan_hbm_bytes_per_step"]
                extra["roofline_scan"]["traffic_source"] = "profiles/r01i_pmc_traffic.json"
        out = {
            "metric": "lambda_tau_build_items_per_s", "value": n / build_s, "unit": "items/s",
            "candidate_precision": ("split-bf16 MFMA symmetric threshold join (f32 MFMA beyond dim 128) -> f32 top-k "
                                    "-> f64 MFMA -> exact f64 scan (results bit-identical to the exact path)")
            if args.cand_precision == 32 else "f64 MFMA -> exact f64 scan",
            "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": t_all / steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "configs[2]: %d x %d low-rank Gaussian mixture (seed 42), item-graph "
                                   "with_lambda_graph(eps=0.5,k=16,topk=16,p=2,sigma=0.25) + per-item lambda-tau "
                                   "(Median) + %d-query prepare+search_lambda_aware(k=%d, alpha=%.1f)"
                                   % (n, d, nq, args.k, args.alpha),
                       "n_items": n, "dim": d, "n_queries": nq, "sharding": "rows x %d" % world},
            "search_qps": nq / search_s, "build_ms": build_s * 1e3, "search_ms": search_s * 1e3,
            "item_graph_nnz": int(state.get("nnz", -1)),
            "knn_uncertified_rows": ctx.stat("knn_uncertified_rows"),
            "knn_join": ctx.stat("knn_join"), "knn_join_candidates_per_row": ctx.stat("knn_join_candidates") / max(1, rows_here),
            "search_uncertified_rows": ctx.stat("search_uncertified_rows"),
            "kernels_ms_per_step": {k: v[0] / steps for k, v in sorted(prof.items())},
            "roofline": roofline,
        }
        out.update(extra)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, lambda a, b: X_host[a:b], cent, n)
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
