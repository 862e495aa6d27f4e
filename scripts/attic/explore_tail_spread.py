"""How the energy outside the 31 dominant directions is spread over the rows of the bench mixtures (torch; the library's directions
come from a subspace iteration on a sample and span nearly the same space as the top eigenvectors taken here).
usage: explore_tail_spread.py [rank] [clusters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arrowspace-rs_amd"))
import torch
from arrowspace_amd import synth
rank = int(sys.argv[1]) if len(sys.argv) > 1 else 128
clusters = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
X, _ = synth.mixture_device(1000000, 128, "cuda", seed=43, n_clusters=clusters, rank=rank)
U = X / X.norm(dim=1, keepdim=True)
S = U[::61]
w, V = torch.linalg.eigh(S.T @ S)
Q = V[:, -31:]
t2 = (1.0 - (U @ Q).pow(2).sum(dim=1)).clamp(min=0)
q = torch.quantile(t2[::7], torch.tensor([0.01, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99], device="cuda", dtype=t2.dtype))
print("rank %d clusters %d: mean tail^2 %.3f; quantiles 1/10/25/50/75/90/99 %%: %s" % (rank, clusters, float(t2.mean()), [round(float(v), 3) for v in q]))
lab = torch.arange(1000000, device="cuda") % clusters
per = torch.zeros(clusters, device="cuda", dtype=t2.dtype).index_add_(0, lab, t2) / (1000000 / clusters)
qc = torch.quantile(per, torch.tensor([0.1, 0.5, 0.9], device="cuda", dtype=t2.dtype))
print("per cluster mean tail^2: 10/50/90 %%: %s; clusters above 0.6: %.1f %%" % ([round(float(v), 3) for v in qc], float((per > 0.6).float().mean()) * 100))
for thr in (0.85, 0.9):
    # pairs of clusters whose rows cannot be told apart by z_i . z_j >= t_i t_j alone: t_i t_j >= thr - 0.1 (a rough margin)
    t = per.sqrt()
    srt = torch.sort(t).values
    frac = float((t[:, None] * t[None, :] >= thr - 0.1).float().mean())
    print("pairs of clusters with t_I t_J >= %.2f: %.1f %%" % (thr - 0.1, frac * 100))
