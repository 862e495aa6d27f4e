"""Small-batch search timing: wall time per batch and the library's per-group device times."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = 128
Xd, cent = synth.mixture_device(n, d, "cuda", seed=42)
lam = torch.rand(n, dtype=torch.float64, device="cuda") * 0.8 + 0.1
asp = A.ArrowSpace(Xd, A.TauMode.Median, ctx)
asp.update_lambdas(lam)
for nq in [int(x) for x in (sys.argv[2:] or ["1", "8", "32"])]:
    Q = (Xd[torch.arange(nq, device="cuda") * 997 % n] * 1.02).contiguous()
    lq = torch.rand(nq, dtype=torch.float64, device="cuda") + 0.1
    for _ in range(3):
        asp.search_lambda_aware_batch(Q, lq, 10, 0.7)
    torch.cuda.synchronize()
    ctx.profile(True)
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        asp.search_lambda_aware_batch(Q, lq, 10, 0.7)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    prof = ctx.profile_all(); ctx.profile(False)
    print("nq=%d  %.3f ms/batch  %.2f TB/s-equivalent  stream=%d overflow=%d" % (nq, dt * 1e3, n * (8 * d + 8) / dt / 1e12,
          ctx.stat("search_stream_scan"), ctx.stat("stream_scan_overflow_rows")), {k: round(v[0] / max(1, v[1]), 3) for k, v in prof.items()}, flush=True)
