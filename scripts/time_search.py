#!/usr/bin/env python3
"""A query batch on a fresh index and on the same index again: wall time and every profiled stage.  usage: time_search.py n d nq [opt=value ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import arrowspace_amd as A  # noqa: E402
from arrowspace_amd import synth  # noqa: E402

n, d, nq = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = A.default_context()
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
X, cent = synth.mixture_device(n, d, "cuda", seed=42)
qidx = synth.query_indices(n, nq, seed=42)
Q = X[torch.from_numpy(qidx.astype(np.int64)).cuda()] * 1.02
torch.manual_seed(42)  # (the same lambdas in every run: scripts/scan_search_seeds.py varies them)
lam = torch.rand(n, dtype=torch.float64, device="cuda") * 0.4 + 0.01
lq = lam[torch.from_numpy(qidx.astype(np.int64)).cuda()] + 0.01
asp = A.ArrowSpace(X, A.TauMode.Median, ctx)
asp.update_lambdas(lam)
for call in range(3):
    ctx.profile(True)
    t0 = time.perf_counter()
    asp.search_lambda_aware_batch(Q, lq, 10, 0.7)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    prof = ctx.profile_all()
    ctx.profile(False)
    print("search %d of the index: %.2f ms (%.0f QPS); order %d, first pass %d, pairs listed %d, candidates %d" % (
        call + 1, dt * 1e3, nq / dt, ctx.stat("search_scan_order"), ctx.stat("search_scan_prefilter"), ctx.stat("search_scan_f16_pairs_listed"),
        ctx.stat("search_scan_candidates")))
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0]):
        if v[1]:
            print("  %-36s %8.3f ms" % (k, v[0]))
