"""A / B of one context option on the item graph of the bench (1 M x 128): wall time of build_laplacian_matrix and the
profile groups, the two settings interleaved on the same box.  usage: python scripts/ab_option.py <option> <a> <b> [n]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch
import arrowspace_amd as A
from arrowspace_amd import synth
opt, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n, d = int(sys.argv[4]) if len(sys.argv) > 4 else 1_000_000, 128
ctx = A.default_context()
Xd, _ = synth.mixture_device(n, d, "cuda", seed=42)
gp = A.GraphParams(0.5, 16, 16, 2.0, 0.25)
A.build_laplacian_matrix(Xd, gp, ctx=ctx)
for rep in range(3):
    for v in (va, vb):
        ctx.set_option(opt, v)
        ctx.profile(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        gl = A.build_laplacian_matrix(Xd, gp, ctx=ctx)
        ctx.synchronize(); dt = time.perf_counter() - t0
        pr = ctx.profile_all(); ctx.profile(False)
        top = sorted(pr.items(), key=lambda kv: -kv[1][0])[:7]
        ip, ix, dv = gl.matrix.to_host()
        import hashlib
        sig = hashlib.sha1(ip.tobytes() + ix.tobytes() + dv.tobytes()).hexdigest()[:10]
        print("%s=%d  build %.2f ms  nnz %d csr %s  " % (opt, v, dt * 1e3, gl.matrix.nnz(), sig) + "  ".join("%s %.2f" % (k.replace("knn_", ""), ms) for k, (ms, _) in top) + "  | basis %.2f pack %.2f thr %.2f band %.2f join %.2f pre %.2f cand %d open %d second %.2f listed %d pretest %d of %d" % (pr.get("knn_join_prefilter_basis", (0, 0))[0], pr.get("pack_unit_rows", (0, 0))[0], pr.get("knn_join_thresholds", (0, 0))[0], pr.get("knn_join_thresholds_band", (0, 0))[0], pr.get("knn_mfma_join_f16", (0, 0))[0], pr.get("knn_join_prefilter", (0, 0))[0], ctx.stat("knn_join_candidates"), ctx.stat("knn_uncertified_rows"), pr.get("knn_mfma_join_f16_second", (0, 0))[0], ctx.stat("knn_mfma_join_f16_blocks_listed"), ctx.stat("knn_mfma_join_f16_blocks_pretest"), ctx.stat("knn_mfma_join_f16_blocks")), flush=True)
        print("      rerank: prune %.3f link %.3f fill %.3f score %.3f rank %.3f  assembly %.3f  long lists %d" % tuple([pr.get("knn_rerank_" + k, (0, 0))[0] for k in ("prune", "link", "fill", "score", "rank")] + [pr.get("graph_assembly", (0, 0))[0], ctx.stat("knn_rerank_long_lists")]), flush=True)
        del gl
