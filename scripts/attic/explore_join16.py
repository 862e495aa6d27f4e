"""Dev check of the f16 join (mfma_join_f16.hip) against the split-bf16 form: identical neighbour tables, stage times,
uncertified rows.  usage: explore_join16.py [n] [d] [pipes...]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
pipes = [int(x) for x in sys.argv[3:]] or [48, 16]
Xd, cent = synth.mixture_device(n, d, "cuda", seed=42)
if os.environ.get("SORTED"):  # rows grouped by their mixture component (the generator interleaves them: row i -> component i % 256)
    Xd = Xd[torch.argsort(torch.arange(n, device="cuda") % 256, stable=True)].contiguous()
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
ctx.set_option("join_min_rows", 512)
if os.environ.get("PREFILTER") is not None: ctx.set_option("join_prefilter", int(os.environ["PREFILTER"]))
if os.environ.get("PREFILTER_MIN") is not None: ctx.set_option("join_prefilter_min_rows", int(os.environ["PREFILTER_MIN"]))
for name in ("join_rank", "join_sample", "join_cap"):
    if os.environ.get(name.upper()): ctx.set_option(name, int(os.environ[name.upper()]))
if os.environ.get("JOIN_THETA_GROUPED") is not None: ctx.set_option("join_theta_grouped", int(os.environ["JOIN_THETA_GROUPED"]))
if os.environ.get("JOIN_ORDER_PIVOTS"): ctx.set_option("join_order_pivots", int(os.environ["JOIN_ORDER_PIVOTS"]))
if os.environ.get("JOIN_BLOCK"): ctx.set_option("join_block", int(os.environ["JOIN_BLOCK"]))
res = {}
for pipe in pipes:
    ctx.set_option("join_pipe", pipe)
    for dbg in ([0, 0, 8] if os.environ.get("STAMPS") else [0, 0]):
        ctx.set_option("debug", dbg)
        ctx.profile(True)
        torch.cuda.synchronize(); t0 = time.time()
        idx, dist = A.knn_cosine(Xd, gp, ctx=ctx)
        torch.cuda.synchronize(); t1 = time.time()
        prof = ctx.profile_all()
    ctx.set_option("debug", 0)
    res[pipe] = (idx.cpu().numpy() if torch.is_tensor(idx) else idx, dist.cpu().numpy() if torch.is_tensor(dist) else dist)
    print("pipe=%d  %.1f ms  cand/row %.1f max %d overflow %d  uncertified %d 1a %d 1b %d 2 %d  blocks %d" % (
        pipe, (t1 - t0) * 1e3, ctx.stat("knn_join_candidates") / n, ctx.stat("knn_join_max_per_row"), ctx.stat("knn_join_overflow_rows"),
        ctx.stat("knn_uncertified_rows"), ctx.stat("knn_uncertified_rows_stage1a"), ctx.stat("knn_uncertified_rows_stage1b"),
        ctx.stat("knn_uncertified_rows_stage2"), ctx.stat("knn_join_blocks")),
        "grouped theta %d bad %d; prefilter %d blocks %d flagged %d delta %.5f rotated %d tail %.4f" % (ctx.stat("knn_join_theta_grouped"), ctx.stat("knn_join_theta_grouped_bad"), ctx.stat("knn_join_prefilter"), ctx.stat("knn_join_prefilter_blocks"),
            ctx.stat("knn_join_prefilter_blocks_flagged"), ctx.stat("knn_join_prefilter_delta_e6") * 1e-6, ctx.stat("knn_join_prefilter_rotated"),
            ctx.stat("knn_join_prefilter_tail_e6") * 1e-6), {k: round(v[0] / v[1], 2) for k, v in prof.items()},
        "TOTALS (ms, launches):", {k: (round(v[0], 2), v[1]) for k, v in prof.items()}, "sum %.1f" % sum(v[0] for v in prof.values()), flush=True)
for dbg in [int(x) for x in os.environ.get("ABLATE", "").split(",") if x]:
    ctx.set_option("join_pipe", pipes[-1]); ctx.set_option("debug", dbg)
    for rep in range(2):
        ctx.profile(True)
        A.knn_cosine(Xd, gp, ctx=ctx); torch.cuda.synchronize()
        prof = ctx.profile_all()
    ctx.set_option("debug", 0)
    print("debug=%d" % dbg, {k: round(v[0] / v[1], 2) for k, v in prof.items() if "join" in k}, flush=True)
for dbg in [int(x) for x in os.environ.get("ABLATE_PRE", "").split(",") if x]:  # bits of the int8 first pass (the rest runs on)
    ctx.set_option("join_pipe", pipes[-1]); ctx.set_option("debug", dbg)
    for rep in range(2):
        ctx.profile(True)
        A.knn_cosine(Xd, gp, ctx=ctx); torch.cuda.synchronize()
        prof = ctx.profile_all()
    ctx.set_option("debug", 0)
    print("debug=%d" % dbg, {k: round(v[0] / v[1], 2) for k, v in prof.items() if "prefilter" in k}, flush=True)
ks = list(res)
for p in ks[1:]:
    print("tables identical (pipe %d vs %d):" % (ks[0], p), np.array_equal(res[ks[0]][0], res[p][0]), np.array_equal(res[ks[0]][1], res[p][1]))
