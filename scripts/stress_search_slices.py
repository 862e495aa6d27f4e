"""Random cases for the search scan by slices and for an index that adopts the images of its item-graph build: the first search of a
fresh index (image in the order of the items, dense scan), its second search (ordered image, int8 first pass, a wave per slice of 64
queries), the first search of an index over the rows of a build with keep_join_images (the build's order and images), and the
one-workgroup-per-512-queries form of the scan must give the same tables; a few queries of every case go to the oracle.
usage: python scripts/stress_search_slices.py <cases> <seed>"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import numpy as np, torch
import arrowspace_amd as A
from arrowspace_amd import synth
from oracle import oracle_ctypes as O
cases, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
ctx = A.default_context()
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
bad = 0
for case in range(cases):
    n = int(rng.integers(66_000, 420_000))
    clusters = int(rng.choice([3, 17, 40, 256, 2000]))
    rank = int(rng.choice([2, 6, 16, 48]))
    nq = int(rng.choice([2048, 2049, 2111, 4096, 5000, 9999, 12_345]))
    k = int(rng.integers(1, 24))
    alpha = float(rng.choice([1.0, 0.9, 0.7, 0.31, 0.05]))
    X, _ = synth.mixture(n, 128, seed=int(rng.integers(1 << 30)), n_clusters=clusters, rank=rank)
    kind = int(rng.integers(4))
    lam = {0: rng.random(n) * 0.4, 1: np.full(n, 0.25), 2: rng.random(n) * 30.0, 3: np.abs(rng.normal(size=n)) * 0.1}[kind]
    qi = rng.integers(0, n, size=nq)
    how = int(rng.integers(3))
    Q = {0: X[qi] * 1.03 + rng.normal(size=(nq, 128)) * 0.01, 1: X[qi].copy(), 2: rng.normal(size=(nq, 128))}[how]
    if rng.random() < 0.3:
        Q[7] = 0.0
    lq = lam[qi] + rng.normal(size=nq) * 0.02
    if rng.random() < 0.3:  # a few duplicates and a zero row among the items
        X[1000:1040] = X[77]
        X[5] = 0.0
    Xd, lamd = torch.from_numpy(X).cuda(), torch.from_numpy(lam).cuda()

    def index_of():
        asp = A.ArrowSpace(Xd, A.TauMode.Median, ctx)
        asp.update_lambdas(lamd)
        asp._ensure_index()
        return asp

    def host(r):
        i, s = r
        return (np.asarray(i.cpu() if torch.is_tensor(i) else i).astype(np.int64), np.asarray(s.cpu() if torch.is_tensor(s) else s))

    t0 = time.time()
    own = index_of()
    r1 = host(own.search_lambda_aware_batch(Q, lq, k, alpha))
    r2 = host(own.search_lambda_aware_batch(Q, lq, k, alpha))
    pre2 = ctx.stat("search_scan_prefilter")
    ctx.set_option("search_scan_slices", 0)
    r3 = host(own.search_lambda_aware_batch(Q, lq, k, alpha))
    ctx.set_option("search_scan_slices", 2)
    ctx.set_option("keep_join_images", 1)
    ctx.set_option("keep_join_images_min_rows", 65536)
    A.build_laplacian_matrix(Xd, gp, ctx=ctx)
    kept = ctx.stat("knn_join_images_kept")
    adopted_ix = index_of()
    adopted = ctx.stat("index_adopted_join_images")
    r4 = host(adopted_ix.search_lambda_aware_batch(Q, lq, k, alpha))
    pre4 = ctx.stat("search_scan_prefilter")
    r5 = host(adopted_ix.search_lambda_aware_batch(Q[:33], lq[:33], k, alpha))
    ctx.set_option("keep_join_images", 0)
    same = all(np.array_equal(r1[0], r[0]) and np.array_equal(r1[1], r[1]) for r in (r2, r3, r4)) and np.array_equal(r1[0][:33], r5[0]) and np.array_equal(r1[1][:33], r5[1])
    ok_oracle = True
    for j in [0, 7, nq - 1] + list(rng.integers(0, nq, size=3)):
        oi, osc = O.search_lambda_aware(X, lam, Q[j], float(lq[j]), k, alpha)
        ok_oracle = ok_oracle and np.array_equal(r1[0][j][:len(oi)], oi.astype(np.int64)) and np.array_equal(r1[1][j][:len(osc)], osc)
    print("case %2d: n %6d clusters %4d rank %2d nq %5d k %2d alpha %.2f lambda kind %d queries %d | first pass on the 2nd search %d, kept %d adopted %d first pass %d | %s %s  %.1f s"
          % (case, n, clusters, rank, nq, k, alpha, kind, how, pre2, kept, adopted, pre4, "same" if same else "DIFFERENT", "oracle ok" if ok_oracle else "ORACLE MISMATCH", time.time() - t0), flush=True)
    bad += 0 if (same and ok_oracle) else 1
    del own, adopted_ix, Xd, lamd
print("ok" if bad == 0 else "FAIL: %d cases" % bad)
