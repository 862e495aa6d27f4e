"""Second and later searches of an index over data of higher rank: with and without the rotated int8 images (search_rot_tail_e3)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
n, d = 1_000_000, 128
rank, clusters = int(sys.argv[1]), int(sys.argv[2])
ctx = A.default_context()
X, _ = synth.mixture_device(n, d, "cuda", seed=43, n_clusters=clusters, rank=rank)
lam = torch.rand(n, dtype=torch.float64, device="cuda")
for cut in (100, 950):
    ctx.set_option("search_rot_tail_e3", cut)
    asp = A.ArrowSpace(X, A.TauMode.Median, ctx)
    asp.update_lambdas(lam)
    out = []
    for nq in (10000, 1, 32):
        qi = torch.randint(0, n, (nq,), device="cuda", generator=torch.Generator(device="cuda").manual_seed(nq))
        Q = (X[qi] * 1.02).contiguous(); lq = (lam[qi] + 0.01).contiguous()
        ts = []
        for rep in range(4):   # the first search of a batch shape builds what later ones reuse
            torch.cuda.synchronize(); t0 = time.perf_counter()
            idx, sc = asp.search_lambda_aware_batch(Q, lq, 10, 0.7)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        out.append((nq, [round(t, 3) for t in ts], ctx.stat("search_scan_prefilter"), ctx.stat("stream_scan_filter")))
        res = (idx.clone() if torch.is_tensor(idx) else np.array(idx), sc.clone() if torch.is_tensor(sc) else np.array(sc))
        if cut == 100: ref = dict(globals().get("ref", {})); ref[nq] = res
        else:
            a, b = ref[nq]
            same = (torch.equal(a, res[0]) and torch.equal(b, res[1])) if torch.is_tensor(a) else (np.array_equal(a, res[0]) and np.array_equal(b, res[1]))
            out[-1] = out[-1] + ("same" if same else "DIFFERENT",)
    print("rank %d clusters %d cut %.2f:" % (rank, clusters, cut * 1e-3), out, flush=True)
