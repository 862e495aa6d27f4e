"""Where the one slow call of bench.py's value_hard_data sits (the call that needs a second CSR block because the graph of the call
before is still alive): stage profile of every call.  usage: python scripts/attic/stall_probe.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch
import arrowspace_amd as A
from arrowspace_amd import synth
n, d = 1_000_000, 128
ctx = A.default_context()
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
what = sys.argv[1] if len(sys.argv) > 1 else ""
Xh_early = None
if len(sys.argv) > 2 and sys.argv[2] == "pregen":   # the second matrix generated BEFORE anything else, and a second of rest behind the generator
    Xh_early, _ = synth.mixture_device(n, d, "cuda", seed=43, n_clusters=4096, rank=64)
    torch.cuda.synchronize()
if what:  # pieces of the lambda leg of a bench step first: lambdas = all of it; feature / taumode / stats / index = one piece
    from arrowspace_amd import distributed as D
    import numpy as np
    Xd, cent = synth.mixture_device(n, d, "cuda", seed=42)
    ops = D.HipOps(ctx)
    sas = D.ShardedArrowSpace(ops, Xd, n, gp, A.TauMode.Median)
    if what not in ("onlybuilds",):
        gl = ops.feature_laplacian(cent, gp, n)
        lam = ops.lambdas(Xd, gl, A.TauMode.Median)
    torch.cuda.synchronize()
    for _ in range(3):
        if what != "onlylambda":
            sas.build_item_graph()
        if what == "lambdas":
            sas.build_lambdas(cent)
        elif what == "feature":
            ops.feature_laplacian(cent, gp, n)
        elif what == "taumode":
            ops.lambdas(Xd, gl, A.TauMode.Median)
        elif what == "stats":
            lt = lam if torch.is_tensor(lam) else torch.from_numpy(np.asarray(lam)).cuda()
            torch.stack([lt.min(), -lt.max(), lt.sum()]).cpu()
        elif what == "index":
            ops.make_index(Xd, lam, A.TauMode.Median, 0)
        elif what == "builds":
            pass
    torch.cuda.synchronize()
    if len(sys.argv) > 2 and sys.argv[2] == "keep":   # (the first matrix, its graph and its index stay alive)
        kept_alive = (sas, Xd)
    else:
        del sas, Xd
if Xh_early is not None:
    time.sleep(1.0)
    Xh = Xh_early
else:
    Xh, _ = synth.mixture_device(n, d, "cuda", seed=43, n_clusters=4096, rank=64)
    if len(sys.argv) > 2 and sys.argv[2] == "rest":   # generated here, as in the bench, but with a second of rest behind the generator
        torch.cuda.synchronize()
        time.sleep(1.0)
A.build_laplacian_matrix(Xh, gp, ctx=ctx); ctx.synchronize()
keys = ("handle_cache_malloc_us", "handle_cache_mallocs", "workspace_malloc_us", "workspace_mallocs")
prev = None
for it in range(5):
    ctx.profile(True)
    a0 = [ctx.stat(k) for k in keys]
    t0 = time.perf_counter()
    gl = A.build_laplacian_matrix(Xh, gp, ctx=ctx)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    pr = {k: v[0] for k, v in ctx.profile_all().items()}
    ctx.profile(False)
    line = "call %d: %.2f ms  allocator %s" % (it, dt, {k: ctx.stat(k) - v for k, v in zip(keys, a0)})
    if prev:
        diff = {k: round(v - prev.get(k, 0.0), 2) for k, v in pr.items() if abs(v - prev.get(k, 0.0)) > 0.3}
        line += "  stages that differ from the call before: %s" % diff
    print(line, flush=True)
    prev = pr
    keep = gl  # (the graph of this call lives until the next one has returned, as in the bench)
