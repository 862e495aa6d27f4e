"""The 10 000-query search of the bench on ONE index several times: the first call scans the image in the order of the items,
later ones the ordered image behind the int8 first pass.  usage: python scripts/attic/search_repeat.py [nq]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch
import arrowspace_amd as A
from arrowspace_amd import synth, distributed as D
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
opts = [a for a in sys.argv[2:] if "=" in a]
n, d = 1_000_000, 128
ctx = A.default_context()
for kv in opts:
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
Xd, cent = synth.mixture_device(n, d, "cuda", seed=42)
import numpy as np
qidx = synth.query_indices(n, nq, seed=42)
Q = Xd[torch.from_numpy(qidx.astype(np.int64)).to("cuda")] * 1.02
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
ops = D.HipOps(ctx)
sas = D.ShardedArrowSpace(ops, Xd, n, gp, A.TauMode.Median)
sas.build_item_graph()
sas.build_lambdas(cent)
for rep in range(5):
    ctx.profile(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lq = sas.prepare_queries(Q)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    sas.search(Q, lq, 10, 0.7)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    pr = ctx.profile_all(); ctx.profile(False)
    print("search %d: prepare %.3f ms  search %.3f ms  " % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3) + "  ".join("%s %.3f" % (k, v[0]) for k, v in sorted(pr.items())),
          " prefilter", ctx.stat("search_scan_prefilter"), "order", ctx.stat("search_scan_order"), "adopted", ctx.stat("index_adopted_join_images"), "kept", ctx.stat("knn_join_images_kept"), flush=True)
