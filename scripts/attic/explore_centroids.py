"""Timing of the batch nearest_centroid kernel (csrc/centroids.hip)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
nc = int(sys.argv[3]) if len(sys.argv) > 3 else 256
X, cent = synth.mixture(n, d, seed=42)
Xd = torch.from_numpy(X).cuda()
Cd = torch.from_numpy(np.ascontiguousarray(cent[:nc] if len(cent) >= nc else np.random.default_rng(0).normal(size=(nc, d)))).cuda()
for rep in range(3):
    ctx.profile(True)
    torch.cuda.synchronize(); t0 = time.time()
    idx, d2 = A.nearest_centroids(Xd, Cd, ctx=ctx)
    torch.cuda.synchronize(); t1 = time.time()
    prof = ctx.profile_all()
print("n %d d %d centroids %d: %.2f ms" % (n, d, Cd.shape[0], (t1 - t0) * 1e3), {k: round(v[0] / v[1], 2) for k, v in prof.items()},
      "-> %.1f G (row, centroid, feature) terms/s" % (n * Cd.shape[0] * d / (t1 - t0) / 1e9))
