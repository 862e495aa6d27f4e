"""Cycle stamps (`debug` bit 8) of the one-sided f16 pass: the search scan of a 10 k-query batch on a FRESH index of 1 M x 128
(the bench's search leg) and the second pass of the item graph.  Printed by the library on stderr."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch
import arrowspace_amd as A
from arrowspace_amd import distributed as D
from arrowspace_amd import synth
n, d, nq = 1_000_000, 128, 10_000
ctx = A.default_context()
for a in sys.argv[1:]:
    name, val = a.split("="); ctx.set_option(name, int(val))
Xd, cent = synth.mixture_device(n, d, "cuda", seed=42)
qidx = synth.query_indices(n, nq, seed=42)
Q = torch.from_numpy(synth.mixture_at(qidx, d, seed=42) * 1.02).cuda()
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
sas = D.ShardedArrowSpace(D.HipOps(ctx), Xd, n, gp, A.TauMode.Median)
sas.build_item_graph(); sas.build_lambdas(cent)
lq = sas.prepare_queries(Q)
sas.search(Q, lq, 10, 0.7)
for rep in range(2):
    sas.build_lambdas(cent)          # a fresh index, as in the bench
    ctx.set_option("debug", 8)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sas.search(Q, lq, 10, 0.7)
    torch.cuda.synchronize(); print("search on a fresh index (with stamps) %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    ctx.set_option("debug", 0)
ctx.set_option("debug", 8)
sas.build_item_graph()
ctx.set_option("debug", 0)
