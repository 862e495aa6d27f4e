"""Builds on a sequence of different data in one process: per call the wall time and the time spent in the device allocator (handle cache,
workspace).  Found with it: the odd call that takes 30 - 60 ms longer after 4 M-row and 768-column data were released spends none of that
in the library's allocator -- the device is still returning the memory torch freed -- which is why the bench legs report their median repetition."""
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
    for rep in range(4):
        s0 = snap()
        t0 = time.perf_counter()
        gl = A.build_laplacian_matrix(X, gp, ctx=ctx)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        s1 = snap()
        print(n, d, rank, cl, "rep", rep, "%.2f ms" % dt, {k.replace("handle_cache_", "h_").replace("workspace_", "ws_"): b - a for k, a, b in zip(keys, s0, s1) if b - a})
    del X, gl
    torch.cuda.empty_cache()
