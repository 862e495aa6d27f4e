// taumode_wide.hip -- K6/K7/K8 for items wider than 128 features (configs[4] of the baseline: F = D = 768).
// Replaces TauMode::select_tau (src/taumode.rs:87-127), TauMode::compute_synthetic_lambda_csr (src/taumode.rs:552-660)
// and their driver loop (src/taumode.rs:230-259) / the batched form of ArrowSpace::prepare_query_item
// (src/core.rs:533-549), as taumode_v3.hip does for F = 128.
//
// The generic kernel (taumode.hip, lambda_kernel<S>) keeps a transposed tile of 64 / S items in LDS and finds tau by a
// 64-step bisection in which every step walks the item's values through LDS again: at F = 768 the tile leaves room for
// five waves per CU and every one of ~3 000 dependent LDS round trips per item is exposed -- 86 ms per million items.
// Here ONE WAVE OWNS ONE ITEM at a time:
//   * its F values are read once, coalesced, into registers (lane l holds features l, 64 + l, ...: NV per lane) -- the
//     next item's loads are issued before this one's arithmetic -- and copied to LDS for the indexed reads of the edge pass;
//   * tau: the order statistic is found on order-preserving integer keys by bisection from the top bit, but a step is NV
//     register compares + ballots + scalar popcounts (no memory), and it runs on the high 32 bits first: as soon as a
//     single value carries the prefix found so far, that value is the answer (distinct values differ well inside their
//     first 32 bits; ties fall through to the low word);
//   * edge pass: lane l takes entries l, 64 + l, ... of the packed (i, j, L_ij) stream -- the upper triangle when L is
//     bitwise symmetric, every pair then standing for (i, j) and (j, i) -- with x_i, x_j from LDS; the three sums are
//     reduced across the wave at the end (a fixed order: deterministic; they differ from the oracle's sequential sums by
//     rounding only: 1e-12 bar).
// Per item ~1.2 k wave instructions for tau and ~12 per 64 entries: the kernel is bound by vector issue, not by HBM.
#include "common.h"

namespace asx {

namespace {

struct __align__(16) PackedEntryW {
  uint32_t i, j;
  double v;
};
constexpr double kTauFloorW = kTauFloor;  // TAU_FLOOR, src/taumode.rs:84

__device__ __forceinline__ uint64_t sort_key_w(double x) {
  const uint64_t u = (uint64_t)__double_as_longlong(x);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double unsort_key_w(uint64_t k) {
  const uint64_t u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)u);
}
__device__ __forceinline__ bool finite_w(double x) { return (((uint64_t)__double_as_longlong(x) >> 52) & 0x7FF) != 0x7FF; }
template <typename T>
__device__ __forceinline__ T wave_sum_w(T v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const uint64_t t = __shfl_xor(v, o, 64);
    v = t < v ? t : v;
  }
  return v;
}
__device__ __forceinline__ uint32_t count_lanes(bool p) { return (uint32_t)__builtin_popcountll(__ballot(p)); }

// k-th smallest (0-based) of the wave's keys (key[q] of lane l = element 64 q + l; elements at or beyond F carry the
// largest key and k < F, so they are never selected)
template <int NV>
__device__ __forceinline__ uint64_t wave_select(const uint64_t (&key)[NV], uint32_t k) {
  // high words: bisection from the top bit
  uint32_t hp = 0;
  for (int b = 31; b >= 0; --b) {
    const uint32_t t = hp | (1u << b);
    uint32_t c = 0;
#pragma unroll
    for (int q = 0; q < NV; ++q) c += count_lanes((uint32_t)(key[q] >> 32) < t);
    if (c <= k) hp = t;
  }
  // the keys that carry this high word: one (the usual case: it is the answer) or several (low words decide)
  uint32_t below = 0, same = 0;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    below += count_lanes((uint32_t)(key[q] >> 32) < hp);
    same += count_lanes((uint32_t)(key[q] >> 32) == hp);
  }
  uint32_t lp = 0;
  if (same == 1) {
    uint32_t mine = 0;
#pragma unroll
    for (int q = 0; q < NV; ++q) mine = (uint32_t)(key[q] >> 32) == hp ? (uint32_t)key[q] : mine;
    bool has = false;
#pragma unroll
    for (int q = 0; q < NV; ++q) has = has || (uint32_t)(key[q] >> 32) == hp;
    const unsigned long long m = __ballot(has);
    lp = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)(__ffsll((long long)m) - 1));
  } else {
    const uint32_t kk = k - below;  // rank among the keys with this high word
    for (int b = 31; b >= 0; --b) {
      const uint32_t t = lp | (1u << b);
      uint32_t c = 0;
#pragma unroll
      for (int q = 0; q < NV; ++q) c += count_lanes((uint32_t)(key[q] >> 32) == hp && (uint32_t)key[q] < t);
      if (c <= kk) lp = t;
    }
  }
  return ((uint64_t)hp << 32) | lp;
}

// SYM: the stream holds the diagonal and j > i only, every off-diagonal entry standing for (i, j) and (j, i)
template <int NV, bool SYM>
__global__ __launch_bounds__(256) void lambda_wide_kernel(const double *__restrict__ items, uint64_t n_items, uint32_t F,
                                                          const PackedEntryW *__restrict__ ent, uint64_t nnz, int mode, double arg,
                                                          int check_finite, int want_lambda, double *__restrict__ out,
                                                          int *__restrict__ err_flag) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double *xs = reinterpret_cast<double *>(smem) + (size_t)wv * 64 * NV;  // the wave's item
  const uint64_t n_waves = (uint64_t)gridDim.x * 4, w0 = (uint64_t)blockIdx.x * 4 + wv;

  double xn[NV];  // the next item, in flight
  auto fetch = [&](uint64_t item) {
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const uint32_t f = (uint32_t)q * 64 + lane;
      xn[q] = (item < n_items && f < F) ? items[item * F + f] : 0.0;
    }
  };
  fetch(w0);
  for (uint64_t item = w0; item < n_items; item += n_waves) {
    double x[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) x[q] = xn[q];
    fetch(item + n_waves);
    bool bad = false;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const bool in = (uint32_t)q * 64 + lane < F;
      bad = bad || (in && !finite_w(x[q]));
      xs[q * 64 + lane] = x[q];
    }
    if (check_finite && __any(bad)) {
      if (lane == 0) atomicOr(err_flag, 1);
    }
    // ---- tau from the item's own values (src/taumode.rs:87-127, 233-234) ----
    double tau;
    if (mode == AS_TAU_FIXED) {
      tau = (finite_w(arg) && arg > 0.0) ? arg : kTauFloorW;
    } else {
      uint32_t m = 0, neg_nf = 0;
      double sum = 0.0;
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const bool in = (uint32_t)q * 64 + lane < F;
        if (in && finite_w(x[q])) {
          ++m;
          sum = sum + x[q];
        } else if (in && ((uint64_t)__double_as_longlong(x[q]) >> 63)) {
          ++neg_nf;
        }
      }
      m = wave_sum_w(m);
      neg_nf = wave_sum_w(neg_nf);
      if (mode == AS_TAU_MEAN) {
        sum = wave_sum_w(sum);
        tau = fmax(m > 0 ? sum / (double)m : 0.0, kTauFloorW);
      } else if (m == 0) {
        tau = kTauFloorW;
      } else {
        uint64_t key[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) key[q] = (uint32_t)q * 64 + lane < F ? sort_key_w(x[q]) : ~0ull;
        if (mode == AS_TAU_PERCENTILE) {
          const double pp = arg != arg ? 0.0 : fmin(fmax(arg, 0.0), 1.0);
          const uint32_t idx = (uint32_t)round((double)(m - 1) * pp);  // half away from zero
          tau = fmax(unsort_key_w(wave_select<NV>(key, idx + neg_nf)), kTauFloorW);
        } else if (m & 1u) {
          tau = fmax(unsort_key_w(wave_select<NV>(key, m / 2 + neg_nf)), kTauFloorW);
        } else {
          const uint32_t k1 = m / 2 - 1 + neg_nf;
          const uint64_t key1 = wave_select<NV>(key, k1);
          // element of rank k1 + 1: key1 again if duplicated, else the smallest key above it
          uint32_t c_le = 0;
          uint64_t nxt = ~0ull;
#pragma unroll
          for (int q = 0; q < NV; ++q) {
            const bool in = (uint32_t)q * 64 + lane < F;
            c_le += (in && key[q] <= key1) ? 1u : 0u;
            if (in && key[q] > key1 && key[q] < nxt) nxt = key[q];
          }
          c_le = wave_sum_w(c_le);
          nxt = wave_min_u64(nxt);
          const uint64_t key2 = (c_le >= k1 + 2) ? key1 : nxt;
          tau = fmax(0.5 * (unsort_key_w(key1) + unsort_key_w(key2)), kTauFloorW);
        }
      }
    }
    if (!want_lambda) {
      if (lane == 0) out[item] = tau;
      continue;
    }
    // ---- one pass over the entries of L (src/taumode.rs:565-639) ----
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double num = 0.0, edge = 0.0, s2 = 0.0;
    for (uint64_t e = lane; e < nnz; e += 64) {
      const PackedEntryW p = ent[e];
      const double xi = xs[p.i], xj = xs[p.j];
      const double term = (xi * p.v) * xj;
      const bool off = p.i != p.j;
      const double w = off ? fmax(-p.v, 0.0) : 0.0;
      const double dd = xi - xj;
      const double t = (w * dd) * dd;
      if (SYM && off) {
        num = num + (term + term);
        edge = edge + (t + t);
        s2 = s2 + (t + t) * t;
      } else {
        num = num + term;
        edge = edge + t;
        s2 = s2 + t * t;
      }
    }
    double den = 0.0;
#pragma unroll
    for (int q = 0; q < NV; ++q) den = den + x[q] * x[q];  // (zeros beyond F)
    num = wave_sum_w(num);
    edge = wave_sum_w(edge);
    s2 = wave_sum_w(s2);
    den = wave_sum_w(den);
    const double e_raw = den > 1e-12 ? num / den : 0.0;
    double g = 0.0;
    if (edge > 0.0) {
      const double e2 = edge * edge;
      if (finite_w(s2) && finite_w(e2) && e2 > 0.0 && s2 > 1e-280) {
        g = s2 / e2;  // sum t^2 / edge^2 == sum (t / edge)^2 up to rounding
      } else {        // extreme magnitudes: the literal two-pass form, shares computed one by one
        double gs = 0.0;
        for (uint64_t e = lane; e < nnz; e += 64) {
          const PackedEntryW p = ent[e];
          const double w = fmax(-p.v, 0.0);
          if (p.i != p.j && w > 0.0) {
            const double dd = xs[p.i] - xs[p.j];
            const double share = ((w * dd) * dd) / edge;
            gs = gs + (SYM ? 2.0 * (share * share) : share * share);
          }
        }
        g = wave_sum_w(gs);
      }
    }
    const double g_raw = fmin(fmax(g, 0.0), 1.0);
    const double e_b = e_raw / (e_raw + tau);
    if (lane == 0) out[item] = tau * e_b + (1.0 - tau) * g_raw;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // everyone is done with xs before the next item overwrites it
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

template <int NV>
int launch_wide(as_ctx *ctx, const double *items, uint64_t n_items, uint32_t F, const void *ent, uint64_t nnz, bool sym, int mode,
                double arg, bool check_finite, bool want_lambda, double *out, int *err) {
  const size_t lds = (size_t)4 * 64 * NV * sizeof(double);
  const unsigned blocks = (unsigned)std::min<uint64_t>(cdiv(n_items, 4), 256ull * 8);
  if (sym) {
    AS_HIP(ctx, hipFuncSetAttribute((const void *)lambda_wide_kernel<NV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((lambda_wide_kernel<NV, true>), dim3(blocks), dim3(256), lds, ctx->stream, items, n_items, F,
                       (const PackedEntryW *)ent, nnz, mode, arg, check_finite ? 1 : 0, want_lambda ? 1 : 0, out, err);
  } else {
    AS_HIP(ctx, hipFuncSetAttribute((const void *)lambda_wide_kernel<NV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((lambda_wide_kernel<NV, false>), dim3(blocks), dim3(256), lds, ctx->stream, items, n_items, F,
                       (const PackedEntryW *)ent, nnz, mode, arg, check_finite ? 1 : 0, want_lambda ? 1 : 0, out, err);
  }
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

}  // namespace

// 128 < F <= 2048.  ent: the packed (i, j, value) stream of L -- its upper triangle (diagonal included) when sym.
// *handled = false when the width is outside this kernel's range (the caller then uses the generic kernel).
int lambda_wide(as_ctx *ctx, const double *items, uint64_t n_items, uint64_t F, const void *ent, uint64_t nnz, bool sym, int mode,
                double arg, bool check_finite, bool want_lambda, double *out, int *err, bool *handled) {
  *handled = false;
  if (F <= 128 || F > 2048 || n_items == 0) return AS_OK;
  *handled = true;
  const uint32_t nv = (uint32_t)((F + 63) / 64);
  ProfScope ps(ctx, want_lambda ? "taumode_lambda" : "select_tau");
#define AS_W(N) return launch_wide<N>(ctx, items, n_items, (uint32_t)F, ent, nnz, sym, mode, arg, check_finite, want_lambda, out, err)
  if (nv <= 4) AS_W(4);
  if (nv <= 8) AS_W(8);
  if (nv <= 12) AS_W(12);
  if (nv <= 16) AS_W(16);
  if (nv <= 24) AS_W(24);
  AS_W(32);
#undef AS_W
}

}  // namespace asx
