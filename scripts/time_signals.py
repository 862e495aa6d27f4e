#!/usr/bin/env python3
"""Item-graph lambda (feature columns as signals over the built graph): stage times by tau mode.  usage: time_signals.py [n] [reps] [opt=value ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch  # noqa: E402

import arrowspace_amd as A  # noqa: E402
from arrowspace_amd import synth  # noqa: E402
from arrowspace_amd.api import compute_taumode_signals  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = A.default_context()
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
Xd, cent = synth.mixture_device(n, 128, "cuda", seed=42)
gl = A.build_laplacian_matrix(Xd, A.GraphParams(0.5, 16, 16, 2.0, 0.25), ctx=ctx)
for name, tm in (("Median", A.TauMode.Median), ("Percentile(0.95)", A.TauMode.Percentile(0.95)), ("Mean", A.TauMode.Mean), ("Fixed", A.TauMode.Fixed(0.5))):
    lam = compute_taumode_signals(Xd, gl.matrix, tm, ctx=ctx)
    ctx.synchronize()
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        lam = compute_taumode_signals(Xd, gl.matrix, tm, ctx=ctx)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    prof = ctx.profile_all()
    ctx.profile(False)
    print("%-18s %.3f ms  %s  fallback columns %d  lambda[0..2] %s" % (name, dt * 1e3, {k: round(v[0] / reps, 3) for k, v in prof.items() if v[1]},
                                                                       ctx.stat("signal_tau_bracket_fallback_columns"), lam[:2].cpu().numpy()))
