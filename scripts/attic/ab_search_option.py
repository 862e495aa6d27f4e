"""A / B of context options on the search leg of the bench (10 k queries on 1 M x 128, a fresh index per call, k = 10,
alpha = 0.7): wall time and the profile groups.  usage: python scripts/ab_search_option.py opt=a,b,c [opt2=v ...]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch
import arrowspace_amd as A
from arrowspace_amd import distributed as D
from arrowspace_amd import synth
n, d, nq = 1_000_000, 128, 10_000
ctx = A.default_context()
sweep = None
for a in sys.argv[1:]:
    name, val = a.split("=")
    if "," in val: sweep = (name, [int(v) for v in val.split(",")])
    else: ctx.set_option(name, int(val))
Xd, cent = synth.mixture_device(n, d, "cuda", seed=42)
qidx = synth.query_indices(n, nq, seed=42)
Q = torch.from_numpy(synth.mixture_at(qidx, d, seed=42) * 1.02).cuda()
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
sas = D.ShardedArrowSpace(D.HipOps(ctx), Xd, n, gp, A.TauMode.Median)
sas.build_item_graph(); sas.build_lambdas(cent)
ref = None
for rep in range(3):
    for v in (sweep[1] if sweep else [None]):
        if sweep: ctx.set_option(sweep[0], v)
        ctx.profile(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lq = sas.prepare_queries(Q)
        idx, sc = sas.search(Q, lq, 10, 0.7)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        pr = ctx.profile_all(); ctx.profile(False)
        if ref is None: ref = (idx.clone(), sc.clone())
        same = bool(torch.equal(idx, ref[0]) and torch.equal(sc, ref[1]))
        top = sorted(pr.items(), key=lambda kv: -kv[1][0])[:6]
        print("%s=%s  search %.2f ms  same %s  cand/query %.1f  " % (sweep[0] if sweep else "-", v, dt * 1e3, same, ctx.stat("search_scan_candidates") / nq) +
              "  ".join("%s %.2f" % (k, ms) for k, (ms, _) in top), flush=True)
