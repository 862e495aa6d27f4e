#!/usr/bin/env python3
"""Where the clustering pre-stage spends its time at the bench size (rows resident in HBM): per-launch-group times from the
context's profiler next to the wall time.  usage: python scripts/time_prestage.py [n] [d] [batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "arrowspace-rs_amd"))
import torch  # noqa: E402
import arrowspace_amd as A  # noqa: E402
import arrowspace_amd.api as API  # noqa: E402
from arrowspace_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
ctx = A.default_context()
rank = int(os.environ.get("RANK_", "8")); clusters = int(os.environ.get("CLUSTERS_", "256"))
Xd, _ = synth.mixture_device(n, d, torch.device("cuda:0"), seed=42, n_clusters=clusters, rank=rank)   # default: the bench's item matrix
k_opt, radius, idim = API.compute_optimal_k(Xd, None, ctx=ctx)
for rep in range(2):
    ctx.profile(True)
    t0 = time.perf_counter()
    k_opt, radius, idim = API.compute_optimal_k(Xd, None, ctx=ctx)
    ctx.synchronize()
    t1 = time.perf_counter()
    cents, assign, sizes = API.incremental_clustering(Xd, k_opt, radius, batch, ctx=ctx)
    ctx.synchronize()
    t2 = time.perf_counter()
    prof = ctx.profile_all()
    ctx.profile(False)
    print("rep %d: optimal_k %.2f ms (k = %d), incremental clustering %.2f ms (%d clusters; batches: sequential %d, parallel %d, host %d)"
          % (rep, (t1 - t0) * 1e3, k_opt, (t2 - t1) * 1e3, len(cents), ctx.stat("clustering_sequential_batches"),
             ctx.stat("clustering_device_batches"), ctx.stat("clustering_host_batches")))
    for name, (ms, cnt) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
        print("   %-28s %9.3f ms  %6d launches" % (name, ms, cnt))
if os.environ.get("STAMPS"):   # cycles per phase of the chain kernel (s_memtime, 100 MHz), per wave
    ctx.set_option("debug", 0x100000)
    API.incremental_clustering(Xd, k_opt, radius, batch, ctx=ctx)
    ctx.set_option("debug", 0)
    for w in range(8):
        print("   wave %d: measure %8d  reduce %8d  apply %8d  close %8d  (x 10 ns, whole batch)"
              % ((w,) + tuple(ctx.stat("clustering_chain_cycles_w%d_p%d" % (w, q)) for q in range(4))))
