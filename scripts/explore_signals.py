"""Dev timing of the item-graph lambda (signal_lambda.hip) at n x 128: the graph pass under different row orders and rows per
wave, the tau passes, parity of sampled signals against the oracle.
usage: explore_signals.py [n] [check]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
from arrowspace_amd.api import compute_taumode_signals
from oracle import oracle_ctypes as O
ctx = A.default_context()
lib = ctx.lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
check = len(sys.argv) > 2
F = 128
X, cent = synth.mixture(n, F, seed=42)
Xd = torch.from_numpy(X).cuda()
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
t0 = time.time()
gl = A.build_laplacian_matrix(Xd, gp, ctx=ctx)
ctx.synchronize()
nnz = gl.matrix.nnz()
print("item graph: n=%d nnz=%d (%.1f s incl. first-call setup)" % (n, nnz, time.time() - t0))
alg_bytes = nnz * 16 + n * F * 8
out = torch.empty(F, dtype=torch.float64, device="cuda")
T = A.TauMode

def run(name, tm=T.Fixed(0.5), iters=4):
    res = None
    for it in range(iters):
        if it == 1:
            ctx.profile(True)
        res = compute_taumode_signals(Xd, gl.matrix, tm, ctx=ctx)
        ctx.synchronize()
    prof = ctx.profile_all()
    ctx.profile(False)
    line = "  %-34s" % name
    for k, (ms, cnt) in sorted(prof.items()):
        line += " %s %.3f ms" % (k.replace("signal_", ""), ms / cnt)
    ms = prof.get("signal_sums", (0, 1))
    t = ms[0] / ms[1]
    if t > 0:
        line += "  | sums: %.2f TB/s algorithmic (%.3f of 8 TB/s)" % (alg_bytes / t / 1e9, alg_bytes / t / 1e9 / 8.0)
    print(line, flush=True)
    return res

run("as built (the join's order)")
orders = {
    "natural": None,
    "by cluster (i mod 256)": np.argsort(np.arange(n) % 256, kind="stable").astype(np.uint32),
    "random": np.random.default_rng(1).permutation(n).astype(np.uint32),
}
for oname, order in orders.items():
    for ch in (8, 16, 32):
        ctx.set_option("signal_chunk_rows", ch)
        gl.matrix.set_order(order)   # drops the stream: rebuilt under the options
        run("%s chunk=%d" % (oname, ch))
ctx.set_option("signal_chunk_rows", 16)
for waves in (2048, 8192):
    ctx.set_option("signal_waves", waves)
    gl.matrix.set_order(orders["by cluster (i mod 256)"])
    run("by cluster waves=%d (static interleave)" % waves)
ctx.set_option("signal_waves", 0)
gl.matrix.set_order(orders["by cluster (i mod 256)"])
ctx.set_option("signal_sym", 0); gl.matrix.set_order(orders["by cluster (i mod 256)"])
run("by cluster, full stream (sym=0)")
ctx.set_option("signal_sym", 1); gl.matrix.set_order(orders["by cluster (i mod 256)"])
for tm, nm in ((T.Median, "median"), (T.Mean, "mean"), (T.Percentile(0.95), "p95")):
    lam = run("by cluster, tau=%s" % nm, tm)
if check:
    csr = gl.matrix.to_host()
    cols = [0, 1, 17, 64, 127]
    lam = compute_taumode_signals(Xd, gl.matrix, T.Median, ctx=ctx).cpu().numpy()
    t0 = time.time()
    want = O.taumode_lambdas(np.ascontiguousarray(X[:, cols].T), csr, O.TAU_MEDIAN)
    print("oracle %d signals: %.1f s" % (len(cols), time.time() - t0))
    rel = np.abs(lam[cols] - want) / np.abs(want)
    print("lambda", lam[cols], "rel err", rel)
    assert np.all(rel <= 1e-12)
