"""Seeded synthetic inputs for tests and bench.py.

Counter-based generator (SplitMix64 + Box-Muller), so any row range can be produced
independently (row-sharded ranks generate only their shard) and the same rows come out for any
sharding.  Plain i.i.d. Gaussian rows give an EMPTY graph at eps = 0.5 in D = 128 (pairwise
cosine ~ N(0, 1/sqrt(D))), so the benchmark data is a low-rank Gaussian mixture:
    B ~ N(0,1)^{D x r},  z_m ~ N(0, I_r),  centre c_m = B z_m,
    row x_i = c_{i mod C} + noise * sqrt(r) * N(0, I_D)
"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        z = x + _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _uniform(stream, counter):
    """(0,1) doubles from a (stream, counter) pair; counter is a uint64 array."""
    with np.errstate(over="ignore"):
        key = _splitmix64(np.uint64(stream) * np.uint64(0xD1342543DE82EF95) + np.uint64(0x2545F4914F6CDD1D))
        bits = _splitmix64(counter ^ key)
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def gaussian(stream, start, count):
    """`count` N(0,1) doubles, element e of the stream = Box-Muller of counters 2e, 2e+1."""
    e = np.arange(start, start + count, dtype=np.uint64)
    u1 = _uniform(stream, e * np.uint64(2))
    u2 = _uniform(stream, e * np.uint64(2) + np.uint64(1))
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def gaussian_rows(n, d, seed=1, row_start=0):
    """Plain i.i.d. N(0,1) rows (BASELINE config 1)."""
    return gaussian(seed * 16 + 3, row_start * d, n * d).reshape(n, d)


def mixture_centres(d, seed=42, n_clusters=256, rank=8):
    B = gaussian(seed * 16 + 0, 0, d * rank).reshape(d, rank)
    Z = gaussian(seed * 16 + 1, 0, n_clusters * rank).reshape(n_clusters, rank)
    return Z @ B.T  # C x D


def mixture(n, d, seed=42, n_clusters=256, rank=8, noise=0.3, row_start=0, chunk=1 << 16):
    """Rows [row_start, row_start + n) of the mixture and the C x D centres."""
    cent = mixture_centres(d, seed, n_clusters, rank)
    out = np.empty((n, d), dtype=np.float64)
    scale = noise * np.sqrt(float(rank))
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        r0 = row_start + s
        g = gaussian(seed * 16 + 2, r0 * d, m * d).reshape(m, d)
        cid = (np.arange(r0, r0 + m) % n_clusters)
        out[s:s + m] = cent[cid] + scale * g
    return out, cent


def queries_from(rows_fn, n_total, nq, seed=42):
    """queries = x_{pi(j)} * 1.02 for a seeded index draw (benches/index_compute_bench.rs:121-126)."""
    u = _uniform(seed * 16 + 5, np.arange(nq, dtype=np.uint64))
    idx = np.minimum((u * n_total).astype(np.int64), n_total - 1)
    return idx, rows_fn(idx) * 1.02


def mixture_at(indices, d, seed=42, n_clusters=256, rank=8, noise=0.3):
    """The mixture rows with the given (arbitrary) global indices."""
    cent = mixture_centres(d, seed, n_clusters, rank)
    idx = np.asarray(indices, dtype=np.int64)
    e = (idx[:, None].astype(np.uint64) * np.uint64(d) + np.arange(d, dtype=np.uint64)[None, :]).ravel()
    u1 = _uniform(seed * 16 + 2, e * np.uint64(2))
    u2 = _uniform(seed * 16 + 2, e * np.uint64(2) + np.uint64(1))
    g = (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).reshape(len(idx), d)
    return cent[idx % n_clusters] + noise * np.sqrt(float(rank)) * g


def query_indices(n_total, nq, seed=42):
    u = _uniform(seed * 16 + 5, np.arange(nq, dtype=np.uint64))
    return np.minimum((u * n_total).astype(np.int64), n_total - 1)


# ---- the same generator on the device (torch) ------------------------------------------------------
# numpy takes ~35 s for 1 M x 128 rows; the full-size parity tests and bench.py need 1 M x 768 and 4 M x 128.
# Same counter-based scheme (SplitMix64 on int64 with wrap-around arithmetic + Box-Muller), evaluated by torch
# on the GPU.  log / cos / sqrt come from a different math library than numpy's, so the rows agree with
# `mixture` only to rounding; whoever needs the host copy (the oracle) takes it from the tensor.

def _i64(c):
    c = int(c) & 0xFFFFFFFFFFFFFFFF
    return c - (1 << 64) if c >= (1 << 63) else c


def _lsr_t(z, s):
    return (z >> s) & ((1 << (64 - s)) - 1)


def _splitmix64_t(x):
    z = x + _i64(_GOLD)
    z = (z ^ _lsr_t(z, 30)) * _i64(_M1)
    z = (z ^ _lsr_t(z, 27)) * _i64(_M2)
    return z ^ _lsr_t(z, 31)


def _uniform_t(stream, counter):
    import torch
    with np.errstate(over="ignore"):
        key = int(_splitmix64(np.uint64(stream) * np.uint64(0xD1342543DE82EF95) + np.uint64(0x2545F4914F6CDD1D)))
    bits = _splitmix64_t(counter ^ _i64(key))
    return (_lsr_t(bits, 11).to(torch.float64) + 0.5) * (1.0 / 9007199254740992.0)


def gaussian_device(stream, start, count, device):
    import torch
    e = torch.arange(start, start + count, dtype=torch.int64, device=device)
    u1 = _uniform_t(stream, e * 2)
    u2 = _uniform_t(stream, e * 2 + 1)
    return torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * np.pi * u2)


def mixture_device(n, d, device, seed=42, n_clusters=256, rank=8, noise=0.3, row_start=0, chunk=1 << 18):
    """Rows [row_start, row_start + n) of the mixture as an f64 tensor on `device`, and the C x D centres (numpy)."""
    import torch
    cent = mixture_centres(d, seed, n_clusters, rank)
    cent_t = torch.from_numpy(cent).to(device)
    out = torch.empty((n, d), dtype=torch.float64, device=device)
    scale = noise * np.sqrt(float(rank))
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        r0 = row_start + s
        g = gaussian_device(seed * 16 + 2, r0 * d, m * d, device).reshape(m, d)
        cid = torch.arange(r0, r0 + m, device=device) % n_clusters
        out[s:s + m] = cent_t[cid] + scale * g
    return out, cent
