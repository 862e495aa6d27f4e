"""Builds on a sequence of different data in one process: per call the wall time, the time spent in the device allocator (handle cache,
workspace) and, for a call much slower than its fastest repetition, the stages that grew.  Found with it (round 5): after 4 M-row and
768-column data were released (torch.cuda.empty_cache) the odd later call takes 30 - 80 ms longer; none of that is spent in the
library's allocator, the time lands in whichever stage is running, and a kernel trace (rocprofv3 --kernel-trace) shows it as one gap
of ~77 ms between two back-to-back launches with no host synchronisation between them (scan_of_sums -> scan_finish): the process's
queues stand still while the driver returns the freed memory.  Hence the bench legs report their MEDIAN repetition."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch
import arrowspace_amd as A
from arrowspace_amd import synth
ctx = A.default_context()
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
keys = ("handle_cache_malloc_us", "handle_cache_free_us", "handle_cache_wait_us", "handle_cache_mallocs", "handle_cache_frees", "handle_cache_hits", "workspace_malloc_us", "workspace_mallocs")
def snap():
    return [ctx.stat(k) for k in keys]
for n, d, rank, cl, seed in ((4000000, 128, 8, 256, 42), (1000000, 768, 8, 256, 42), (1000000, 128, 8, 256, 42), (1000000, 128, 16, 1024, 43), (1000000, 128, 32, 2048, 43),
                             (1000000, 128, 64, 4096, 43), (1000000, 128, 128, 4096, 43), (1000000, 128, 32, 2048, 43)):
    X, _ = synth.mixture_device(n, d, "cuda", seed=seed, n_clusters=cl, rank=rank)
    gl = None
    stages = []
    for rep in range(5):
        s0 = snap()
        ctx.profile(True)
        t0 = time.perf_counter()
        gl = A.build_laplacian_matrix(X, gp, ctx=ctx)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        prof = ctx.profile_all()
        ctx.profile(False)
        s1 = snap()
        stages.append((dt, {k: round(v[0], 2) for k, v in prof.items() if v[1]}))
        print(n, d, rank, cl, "rep", rep, "%.2f ms" % dt, {k.replace("handle_cache_", "h_").replace("workspace_", "ws_"): b - a for k, a, b in zip(keys, s0, s1) if b - a})
    fast = min(stages[1:], key=lambda t: t[0])
    for rep, (dt, st) in enumerate(stages):
        if rep and dt > 1.3 * fast[0]:
            print("   slow rep", rep, "stages above the fastest rep's:", {k: (fast[1].get(k), v) for k, v in st.items() if v > 1.3 * fast[1].get(k, 0) + 0.2},
                  "sum of stages %.1f against %.1f" % (sum(st.values()), sum(fast[1].values())))
    del X, gl
    torch.cuda.empty_cache()
