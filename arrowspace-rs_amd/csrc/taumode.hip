// taumode.hip -- K6/K7/K8: per-item tau selection and synthetic lambda.
// Replaces TauMode::select_tau (src/taumode.rs:87-127),
// TauMode::compute_synthetic_lambda_csr (src/taumode.rs:552-660),
// TauMode::compute_taumode_lambdas_parallel (src/taumode.rs:174-312) and the batched form of
// ArrowSpace::prepare_query_item (src/core.rs:533-549).
//
// Mapping: the reference nests rayon loops (items outside, CSR rows inside) and copies each
// item row.  Here one wave owns a tile of IPW = 64/S items, staged transposed in LDS
// (xT[f][item], conflict-free: at any instant all lanes of an item group read the same f);
// S lanes share one item and split the CSR entry stream; the CSR itself is a flat stream of
// packed (row, col, value) entries read with wave-(near-)uniform loads.  One pass produces
// num = x'Lx, the edge energy and the sum of squared edge shares; tau comes from a 64-step
// bisection on order-preserving integer keys (no sort, the tile stays intact for the graph pass).
#include <utility>

#include "common.h"
#include "sortnet.h"

namespace asx {

struct __align__(16) PackedEntry {
  uint32_t i, j;
  double v;
};

__global__ void pack_csr_kernel(const uint64_t *__restrict__ indptr, const uint64_t *__restrict__ indices,
                                const double *__restrict__ data, uint64_t n, PackedEntry *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (uint64_t e = indptr[i]; e < indptr[i + 1]; ++e) {
    PackedEntry p;
    p.i = (uint32_t)i;
    p.j = (uint32_t)indices[e];
    p.v = data[e];
    out[e] = p;
  }
}

// Is L bitwise symmetric (L_ij == L_ji for every stored entry)?  Per row: count of entries with j >= i.
__global__ void csr_symmetry_kernel(const uint64_t *__restrict__ indptr, const uint64_t *__restrict__ indices,
                                    const double *__restrict__ data, uint64_t n, uint32_t *__restrict__ upper_cnt,
                                    int *__restrict__ not_symmetric) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t up = 0;
  for (uint64_t e = indptr[i]; e < indptr[i + 1]; ++e) {
    const uint64_t j = indices[e];
    if (j >= i) ++up;
    if (j == i) continue;
    if (j >= n) {
      atomicOr(not_symmetric, 1);
      continue;
    }
    uint64_t lo = indptr[j], hi = indptr[j + 1];  // binary search column i in row j
    bool found = false;
    while (lo < hi) {
      const uint64_t mid = (lo + hi) >> 1;
      const uint64_t c = indices[mid];
      if (c == i) {
        found = __double_as_longlong(data[mid]) == __double_as_longlong(data[e]);
        break;
      }
      if (c < i) lo = mid + 1;
      else hi = mid;
    }
    if (!found) atomicOr(not_symmetric, 1);
  }
  upper_cnt[i] = up;
}

__global__ void pack_upper_kernel(const uint64_t *__restrict__ indptr, const uint64_t *__restrict__ indices,
                                  const double *__restrict__ data, uint64_t n, const uint64_t *__restrict__ uptr,
                                  PackedEntry *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t o = uptr[i];
  for (uint64_t e = indptr[i]; e < indptr[i + 1]; ++e) {
    if (indices[e] < i) continue;
    PackedEntry p;
    p.i = (uint32_t)i;
    p.j = (uint32_t)indices[e];
    p.v = data[e];
    out[o++] = p;
  }
}

static int ensure_packed(as_ctx *ctx, as_csr *L) {
  if (L->packed || L->nnz == 0) return AS_OK;
  AS_HIP(ctx, handle_malloc(&L->packed, sizeof(PackedEntry) * L->nnz));
  hipLaunchKernelGGL(pack_csr_kernel, dim3(cdiv(L->n, 256)), dim3(256), 0, ctx->stream, L->indptr, L->indices,
                     L->data, L->n, (PackedEntry *)L->packed);
  AS_HIP(ctx, hipGetLastError());
  // symmetric matrices (every Laplacian this library builds) also get the upper-triangle stream
  DevBuf<uint32_t> up;
  DevBuf<uint64_t> uptr;
  DevBuf<int> flag;
  AS_TRY(up.alloc(ctx, L->n));
  AS_TRY(uptr.alloc(ctx, L->n + 1));
  AS_TRY(flag.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(flag.p, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL(csr_symmetry_kernel, dim3(cdiv(L->n, 256)), dim3(256), 0, ctx->stream, L->indptr, L->indices,
                     L->data, L->n, up.p, flag.p);
  AS_TRY(exclusive_scan_u32_u64(ctx, up.p, L->n, uptr.p));
  int h = 1;
  uint64_t n_upper = 0;
  AS_HIP(ctx, copy_d2h(ctx, &h, flag.p, sizeof(int)));
  AS_HIP(ctx, copy_d2h(ctx, &n_upper, uptr.p + L->n, sizeof(uint64_t)));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (h == 0 && n_upper > 0) {
    AS_HIP(ctx, handle_malloc(&L->packed_upper, sizeof(PackedEntry) * n_upper));
    hipLaunchKernelGGL(pack_upper_kernel, dim3(cdiv(L->n, 256)), dim3(256), 0, ctx->stream, L->indptr, L->indices,
                       L->data, L->n, uptr.p, (PackedEntry *)L->packed_upper);
    AS_HIP(ctx, hipGetLastError());
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    L->nnz_upper = n_upper;
  }
  return AS_OK;
}

__device__ __forceinline__ uint64_t sort_key(double x) {
  const uint64_t u = (uint64_t)__double_as_longlong(x);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double unsort_key(uint64_t k) {
  const uint64_t u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)u);
}
__device__ __forceinline__ bool finite_bits(double x) {
  return (((uint64_t)__double_as_longlong(x) >> 52) & 0x7FF) != 0x7FF;
}

// reduce over the S lanes that share an item (lanes item + s*IPW)
template <int S, typename T>
__device__ __forceinline__ T group_sum(T v) {
#pragma unroll
  for (int o = 64 / 2; o >= 64 / S && S > 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int S>
__device__ __forceinline__ uint64_t group_min_u64(uint64_t v) {
#pragma unroll
  for (int o = 64 / 2; o >= 64 / S && S > 1; o >>= 1) {
    const uint64_t t = __shfl_xor(v, o, 64);
    v = t < v ? t : v;
  }
  return v;
}

// k-th smallest (0-based) sortable key among the tile column; neg_nf = number of non-finite
// values with the sign bit set (they sort below every finite value and are skipped).
template <int S>
__device__ __forceinline__ uint64_t select_key(const double *xcol, uint32_t F, uint32_t stride, int s, uint32_t k) {
  uint64_t res = 0;
  for (int b = 63; b >= 0; --b) {
    const uint64_t t = res | (1ull << b);
    uint32_t c = 0;
    for (uint32_t f = s; f < F; f += S) c += sort_key(xcol[(size_t)f * stride]) < t ? 1u : 0u;
    c = group_sum<S>(c);
    if (c <= k) res = t;
  }
  return res;
}

template <int S>
__device__ double tile_select_tau(const double *xcol, uint32_t F, uint32_t stride, int s, int mode, double arg) {
  if (mode == AS_TAU_FIXED) return (finite_bits(arg) && arg > 0.0) ? arg : kTauFloor;
  // finite count, sum (index order within a lane), negative non-finite count
  uint32_t m = 0, neg_nf = 0;
  double sum = 0.0;
  for (uint32_t f = s; f < F; f += S) {
    const double x = xcol[(size_t)f * stride];
    if (finite_bits(x)) {
      ++m;
      sum = __dadd_rn(sum, x);
    } else if ((uint64_t)__double_as_longlong(x) >> 63) {
      ++neg_nf;
    }
  }
  m = group_sum<S>(m);
  neg_nf = group_sum<S>(neg_nf);
  if (mode == AS_TAU_MEAN) {
    sum = group_sum<S>(sum);
    const double mean = m > 0 ? __ddiv_rn(sum, (double)m) : 0.0;
    return fmax(mean, kTauFloor);
  }
  if (m == 0) return kTauFloor;
  if (mode == AS_TAU_PERCENTILE) {
    double pp = arg != arg ? 0.0 : fmin(fmax(arg, 0.0), 1.0);
    const uint32_t idx = (uint32_t)round((double)(m - 1) * pp);  // half away from zero
    const double v = unsort_key(select_key<S>(xcol, F, stride, s, idx + neg_nf));
    return fmax(v, kTauFloor);
  }
  // median
  if (m & 1u) {
    const double v = unsort_key(select_key<S>(xcol, F, stride, s, m / 2 + neg_nf));
    return fmax(v, kTauFloor);
  }
  const uint32_t k1 = m / 2 - 1 + neg_nf;
  const uint64_t key1 = select_key<S>(xcol, F, stride, s, k1);
  // element of rank k1+1: key1 again if duplicated, else the smallest key above it
  uint32_t c_le = 0;
  uint64_t nxt = ~0ull;
  for (uint32_t f = s; f < F; f += S) {
    const uint64_t kk = sort_key(xcol[(size_t)f * stride]);
    c_le += kk <= key1 ? 1u : 0u;
    if (kk > key1 && kk < nxt) nxt = kk;
  }
  c_le = group_sum<S>(c_le);
  nxt = group_min_u64<S>(nxt);
  const uint64_t key2 = (c_le >= k1 + 2) ? key1 : nxt;
  const double mid = __dmul_rn(0.5, __dadd_rn(unsort_key(key1), unsort_key(key2)));
  return fmax(mid, kTauFloor);
}

// LDS: xT[F][IPW+1] doubles per wave
template <int S>
__global__ __launch_bounds__(64) void lambda_kernel(const double *__restrict__ items, uint64_t n_items,
                                                    uint32_t F, const PackedEntry *__restrict__ ent,
                                                    uint64_t nnz, int mode, double arg, int check_finite,
                                                    int want_lambda, double *__restrict__ out,
                                                    int *__restrict__ err_flag) {
  constexpr int IPW = 64 / S;
  constexpr int STRIDE = IPW + 1;
  extern __shared__ double xT[];
  const int lane = threadIdx.x;
  const int item_l = lane % IPW;
  const int s = lane / IPW;
  const uint64_t n_tiles = (n_items + IPW - 1) / IPW;

  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t item0 = tile * IPW;
    const uint32_t rows_here = (uint32_t)min((uint64_t)IPW, n_items - item0);
    // ---- stage the tile transposed: coalesced global reads, padded LDS rows ----
    __syncthreads();
    const uint32_t total = (uint32_t)IPW * F;
    const double *src = items + item0 * F;
    bool bad = false;
    for (uint32_t g = lane; g < total; g += 64) {
      const uint32_t r = g / F, e = g - r * F;
      double v = 0.0;
      if (r < rows_here) {
        v = src[g];
        bad |= !finite_bits(v);
      }
      xT[(size_t)e * STRIDE + r] = v;
    }
    if (check_finite && __any(bad)) {
      if (lane == 0) atomicOr(err_flag, 1);
    }
    __syncthreads();
    const double *xcol = xT + item_l;
    // ---- tau from the item's own values (src/taumode.rs:233-234) ----
    const double tau = tile_select_tau<S>(xcol, F, STRIDE, s, mode, arg);
    if (!want_lambda) {
      if (s == 0 && (uint32_t)item_l < rows_here) out[item0 + item_l] = tau;
      continue;
    }
    // ---- one pass over the CSR entries (src/taumode.rs:565-639) ----
    double num = 0.0, edge = 0.0, s2 = 0.0;
    double lnum = 0.0, ledge = 0.0;
    uint32_t cur_row = 0xFFFFFFFFu;
    for (uint64_t e = s; e < nnz; e += S) {
      const PackedEntry p = ent[e];
      if (p.i != cur_row) {  // row-local partial sums, folded in row order (S == 1: as the oracle)
        num = __dadd_rn(num, lnum);
        edge = __dadd_rn(edge, ledge);
        lnum = 0.0;
        ledge = 0.0;
        cur_row = p.i;
      }
      const double xi = xcol[(size_t)p.i * STRIDE];
      const double xj = xcol[(size_t)p.j * STRIDE];
      lnum = __dadd_rn(lnum, __dmul_rn(__dmul_rn(xi, p.v), xj));  // xi * lij * xj
      if (p.i != p.j) {
        const double w = fmax(-p.v, 0.0);
        if (w > 0.0) {
          const double dd = __dsub_rn(xi, xj);
          const double t = __dmul_rn(__dmul_rn(w, dd), dd);  // w * d * d
          ledge = __dadd_rn(ledge, t);
          s2 = __dadd_rn(s2, __dmul_rn(t, t));
        }
      }
    }
    num = __dadd_rn(num, lnum);
    edge = __dadd_rn(edge, ledge);
    num = group_sum<S>(num);
    edge = group_sum<S>(edge);
    s2 = group_sum<S>(s2);
    // denominator over ALL entries of the item (src/taumode.rs:596)
    double den = 0.0;
    for (uint32_t f = s; f < F; f += S) {
      const double x = xcol[(size_t)f * STRIDE];
      den = __dadd_rn(den, __dmul_rn(x, x));
    }
    den = group_sum<S>(den);
    const double e_raw = den > 1e-12 ? __ddiv_rn(num, den) : 0.0;
    double g = 0.0;
    if (edge > 0.0) {
      const double e2 = __dmul_rn(edge, edge);
      if (finite_bits(s2) && finite_bits(e2) && e2 > 0.0 && s2 > 1e-280) {
        g = __ddiv_rn(s2, e2);  // sum t^2 / edge^2 == sum (t/edge)^2 up to rounding
      } else {
        // extreme magnitudes: literal two-pass form, shares computed one by one
        double gs = 0.0;
        for (uint64_t e = s; e < nnz; e += S) {
          const PackedEntry p = ent[e];
          if (p.i == p.j) continue;
          const double w = fmax(-p.v, 0.0);
          if (w > 0.0) {
            const double dd = __dsub_rn(xcol[(size_t)p.i * STRIDE], xcol[(size_t)p.j * STRIDE]);
            const double share = __ddiv_rn(__dmul_rn(__dmul_rn(w, dd), dd), edge);
            gs = __dadd_rn(gs, __dmul_rn(share, share));
          }
        }
        g = group_sum<S>(gs);
      }
    }
    const double g_raw = fmin(fmax(g, 0.0), 1.0);
    const double e_b = __ddiv_rn(e_raw, __dadd_rn(e_raw, tau));
    const double lam = __dadd_rn(__dmul_rn(tau, e_b), __dmul_rn(__dsub_rn(1.0, tau), g_raw));
    if (s == 0 && (uint32_t)item_l < rows_here) out[item0 + item_l] = lam;
  }
}

// ---- fast path (n_features <= 128): 2-wave workgroup per 64-item tile ---------------------------
// Both waves stage the tile transposed into LDS.  Wave 0 then pulls each lane's item into registers
// and runs a Batcher odd-even merge sort network on them (sortnet.h; no data-dependent control flow,
// all indices static) to get tau; wave 1 meanwhile walks the packed CSR stream with wave-uniform
// loads, one item per lane, accumulating x'Lx, the edge energy and the sum of squared edge terms.
// The two meet in LDS and wave 0 writes lambda.  Sums are sequential in CSR order, so with an
// unsymmetric stream lambda agrees with the oracle to the last bits of the G term.
template <int NP, size_t... Is>
__device__ __forceinline__ void sortnet_apply(double (&v)[NP], std::index_sequence<Is...>) {
  auto ce = [&](int a, int b) {
    const double lo = fmin(v[a], v[b]);
    const double hi = fmax(v[a], v[b]);
    v[a] = lo;
    v[b] = hi;
  };
  (ce(SortNet<NP>::kA[Is], SortNet<NP>::kB[Is]), ...);
}

template <int NP>
__device__ __forceinline__ double sorted_pick(const double (&v)[NP], uint32_t idx) {
  double r = v[0];
#pragma unroll
  for (int f = 1; f < NP; ++f) r = idx == (uint32_t)f ? v[f] : r;
  return r;
}

// tau of one item held in xcol[f * stride], f < F <= NP
template <int NP>
__device__ __forceinline__ double regs_select_tau(const double *xcol, uint32_t F, uint32_t stride, int mode,
                                                  double arg, bool &bad) {
  double v[NP];
  uint32_t m = 0;
  double sum = 0.0;
  const double inf = __builtin_inf();
#pragma unroll
  for (int f = 0; f < NP; ++f) {
    double x = inf;
    if ((uint32_t)f < F) {
      x = xcol[(size_t)f * stride];
      if (finite_bits(x)) {
        ++m;
        sum = __dadd_rn(sum, x);
      } else {
        bad = true;
        x = inf;
      }
    }
    v[f] = x;
  }
  if (mode == AS_TAU_FIXED) return (finite_bits(arg) && arg > 0.0) ? arg : kTauFloor;
  if (mode == AS_TAU_MEAN) return fmax(m > 0 ? __ddiv_rn(sum, (double)m) : 0.0, kTauFloor);
  if (m == 0) return kTauFloor;
  sortnet_apply<NP>(v, std::make_index_sequence<SortNet<NP>::kCount>{});
  if (mode == AS_TAU_PERCENTILE) {
    const double pp = arg != arg ? 0.0 : fmin(fmax(arg, 0.0), 1.0);
    const uint32_t idx = (uint32_t)round((double)(m - 1) * pp);
    return fmax(sorted_pick<NP>(v, idx), kTauFloor);
  }
  if (m & 1u) return fmax(sorted_pick<NP>(v, m / 2), kTauFloor);
  const double mid = __dmul_rn(0.5, __dadd_rn(sorted_pick<NP>(v, m / 2 - 1), sorted_pick<NP>(v, m / 2)));
  return fmax(mid, kTauFloor);
}

template <int NP>
__global__ __launch_bounds__(256, 2) void lambda_kernel_v2(const double *__restrict__ items, uint64_t n_items,
                                                           uint32_t F, const PackedEntry *__restrict__ ent,
                                                           uint64_t nnz, int symmetric, int mode, double arg,
                                                           int check_finite, int want_lambda,
                                                           double *__restrict__ out, int *__restrict__ err_flag) {
  constexpr int IPW = 64, STRIDE = 65;
  extern __shared__ double xT[];            // [F][65] tile + [64] tau + [3][64][4] partials
  double *s_tau = xT + (size_t)F * STRIDE;
  double *s_part = s_tau + 64;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;  // wave 0: tau; waves 1..3: CSR stream thirds
  const uint64_t n_tiles = (n_items + IPW - 1) / IPW;
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t item0 = tile * IPW;
    const uint32_t rows_here = (uint32_t)min((uint64_t)IPW, n_items - item0);
    __syncthreads();
    const double *src = items + item0 * F;
    // thread = feature, loop = row: coalesced row reads, 16 independent loads in flight per thread,
    // conflict-free transposed LDS writes (row stride 65 doubles)
    {  // 256 threads: feature = tid & 127, rows split in two halves
      const uint32_t f = tid & 127, r0 = (tid >> 7) * (IPW / 2);
      if (f < F) {
#pragma unroll 16
        for (uint32_t r = r0; r < r0 + IPW / 2; ++r)
          xT[(size_t)f * STRIDE + r] = r < rows_here ? src[(size_t)r * F + f] : 0.0;
      }
    }
    __syncthreads();
    const double *xcol = xT + lane;
    if (wv == 0) {
      bool bad = false;
      const double tau = regs_select_tau<NP>(xcol, F, STRIDE, mode, arg, bad);
      s_tau[lane] = tau;
      if (check_finite && __any(bad && (uint32_t)lane < rows_here) && lane == 0) atomicOr(err_flag, 1);
    } else if (want_lambda) {
      double num = 0.0, edge = 0.0, s2 = 0.0;
      // Entries are fetched 64 at a time (one per lane, coalesced) and broadcast with readlane; they
      // are consumed four at a time so that the eight LDS reads of a group are in flight together
      // (one wave per SIMD: nothing else hides the LDS latency).  Accumulation is flat in stream
      // order (the reference folds row-local sums; the difference is rounding only).
      constexpr int U = 4;
      // this wave's third of the stream, in 64-entry batches
      const uint64_t n_batches = (nnz + 63) / 64;
      const uint64_t b_lo = n_batches * (wv - 1) / 3, b_hi = n_batches * wv / 3;
      for (uint64_t e0 = b_lo * 64; e0 < b_hi * 64 && e0 < nnz; e0 += 64) {
        const uint32_t nb = (uint32_t)min((uint64_t)64, nnz - e0);
        PackedEntry mine;
        mine.i = mine.j = 0;
        mine.v = 0.0;  // padding entries (i = j = 0, v = 0) contribute exactly 0
        if ((uint32_t)lane < nb) mine = ent[e0 + lane];
        const int v_lo = (int)(__double_as_longlong(mine.v) & 0xFFFFFFFFll);
        const int v_hi = (int)(__double_as_longlong(mine.v) >> 32);
        const uint32_t nbu = (nb + U - 1) / U * U;
        for (uint32_t k = 0; k < nbu; k += U) {
          uint32_t pi[U], pj[U];
          double pv[U], xi[U], xj[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            pi[u] = (uint32_t)__builtin_amdgcn_readlane((int)mine.i, k + u);
            pj[u] = (uint32_t)__builtin_amdgcn_readlane((int)mine.j, k + u);
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane(v_lo, k + u);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane(v_hi, k + u);
            pv[u] = __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
            xi[u] = xcol[(size_t)pi[u] * STRIDE];
            xj[u] = xcol[(size_t)pj[u] * STRIDE];
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const double term = __dmul_rn(__dmul_rn(xi[u], pv[u]), xj[u]);  // xi * lij * xj
            const bool off = pi[u] != pj[u];
            const double w = off ? fmax(-pv[u], 0.0) : 0.0;
            const double dd = __dsub_rn(xi[u], xj[u]);
            const double t = __dmul_rn(__dmul_rn(w, dd), dd);  // w * d * d (0 on the diagonal / w = 0)
            if (symmetric && off) {  // stream holds j > i only: each pair stands for (i,j) and (j,i)
              num = __dadd_rn(num, __dadd_rn(term, term));
              edge = __dadd_rn(edge, __dadd_rn(t, t));
              s2 = __dadd_rn(s2, __dmul_rn(__dadd_rn(t, t), t));
            } else {
              num = __dadd_rn(num, term);
              edge = __dadd_rn(edge, t);
              s2 = __dadd_rn(s2, __dmul_rn(t, t));
            }
          }
        }
      }
      double den = 0.0;
      if (wv == 1) {
        for (uint32_t f = 0; f < F; ++f) {
          const double x = xcol[(size_t)f * STRIDE];
          den = __dadd_rn(den, __dmul_rn(x, x));
        }
      }
      double *sp = s_part + (size_t)(wv - 1) * 256;
      sp[lane * 4 + 0] = num;
      sp[lane * 4 + 1] = edge;
      sp[lane * 4 + 2] = s2;
      sp[lane * 4 + 3] = den;
    }
    __syncthreads();
    if (wv == 0 && (uint32_t)lane < rows_here) {
      const double tau = s_tau[lane];
      if (!want_lambda) {
        out[item0 + lane] = tau;
      } else {
        double num = 0.0, edge = 0.0, s2 = 0.0;
        const double den = s_part[lane * 4 + 3];
        for (int w = 0; w < 3; ++w) {  // fixed combination order: deterministic
          num = __dadd_rn(num, s_part[w * 256 + lane * 4 + 0]);
          edge = __dadd_rn(edge, s_part[w * 256 + lane * 4 + 1]);
          s2 = __dadd_rn(s2, s_part[w * 256 + lane * 4 + 2]);
        }
        const double e_raw = den > 1e-12 ? __ddiv_rn(num, den) : 0.0;
        double g = 0.0;
        if (edge > 0.0) {
          const double e2 = __dmul_rn(edge, edge);
          if (finite_bits(s2) && finite_bits(e2) && e2 > 0.0 && s2 > 1e-280) {
            g = __ddiv_rn(s2, e2);
          } else {  // extreme magnitudes: literal two-pass form
            double gs = 0.0;
            for (uint64_t e = 0; e < nnz; ++e) {
              const PackedEntry p = ent[e];
              if (p.i == p.j) continue;
              const double w = fmax(-p.v, 0.0);
              if (w > 0.0) {
                const double dd = __dsub_rn(xcol[(size_t)p.i * STRIDE], xcol[(size_t)p.j * STRIDE]);
                const double share = __ddiv_rn(__dmul_rn(__dmul_rn(w, dd), dd), edge);
                gs = __dadd_rn(gs, symmetric ? __dmul_rn(2.0, __dmul_rn(share, share)) : __dmul_rn(share, share));
              }
            }
            g = gs;
          }
        }
        const double g_raw = fmin(fmax(g, 0.0), 1.0);
        const double e_b = __ddiv_rn(e_raw, __dadd_rn(e_raw, tau));
        out[item0 + lane] = __dadd_rn(__dmul_rn(tau, e_b), __dmul_rn(__dsub_rn(1.0, tau), g_raw));
      }
    }
  }
}

template <int NP>
static int launch_lambda_v2(as_ctx *ctx, const double *items, uint64_t n_items, uint32_t F, const PackedEntry *ent,
                            uint64_t nnz, int symmetric, int mode, double arg, int check_finite, int want_lambda,
                            double *out, int *err) {
  const size_t lds = ((size_t)F * 65 + 64 + 3 * 256) * sizeof(double);
  const uint64_t n_tiles = (n_items + 63) / 64;
  const unsigned blocks = (unsigned)std::min<uint64_t>(n_tiles, 256ull * 8);
  AS_HIP(ctx, hipFuncSetAttribute((const void *)lambda_kernel_v2<NP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
  hipLaunchKernelGGL(lambda_kernel_v2<NP>, dim3(blocks), dim3(256), lds, ctx->stream, items, n_items, F, ent, nnz,
                     symmetric, mode, arg, check_finite, want_lambda, out, err);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

template <int S>
static int launch_lambda(as_ctx *ctx, const double *items, uint64_t n_items, uint32_t F, const PackedEntry *ent,
                         uint64_t nnz, int mode, double arg, int check_finite, int want_lambda, double *out,
                         int *err) {
  constexpr int IPW = 64 / S;
  const size_t lds = (size_t)F * (IPW + 1) * sizeof(double);
  const uint64_t n_tiles = (n_items + IPW - 1) / IPW;
  const unsigned blocks = (unsigned)std::min<uint64_t>(n_tiles, 256ull * 16);
  AS_HIP(ctx, hipFuncSetAttribute((const void *)lambda_kernel<S>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
  hipLaunchKernelGGL(lambda_kernel<S>, dim3(blocks), dim3(64), lds, ctx->stream, items, n_items, F, ent, nnz, mode,
                     arg, check_finite, want_lambda, out, err);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

int lambda_wide(as_ctx *ctx, const double *items, uint64_t n_items, uint64_t F, const void *ent, uint64_t nnz, bool sym, int mode,
                double arg, bool check_finite, bool want_lambda, double *out, int *err, bool *handled);

static int run_lambda(as_ctx *ctx, const double *items, uint64_t n_items, uint64_t F64, const PackedEntry *ent,
                      uint64_t nnz, const PackedEntry *ent_upper, uint64_t nnz_upper, int mode, double arg,
                      bool check_finite, bool want_lambda, double *out) {
  if (n_items == 0) return AS_OK;
  if (mode < AS_TAU_FIXED || mode > AS_TAU_PERCENTILE) return fail(ctx, AS_ERR_ARG, "bad tau_mode");
  const uint32_t F = (uint32_t)F64;
  DevBuf<int> err;
  AS_TRY(err.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(err.p, 0, sizeof(int), ctx->stream));
  if (F <= 128 && ctx->opt_lambda_v2) {
    ProfScope ps(ctx, want_lambda ? "taumode_lambda" : "select_tau");
    const int sym = ent_upper != nullptr && ctx->opt_lambda_v2 != 2;
    if (sym) {
      ent = ent_upper;
      nnz = nnz_upper;
    }
    if (F <= 16) AS_TRY(launch_lambda_v2<16>(ctx, items, n_items, F, ent, nnz, sym, mode, arg, check_finite, want_lambda, out, err.p));
    else if (F <= 32) AS_TRY(launch_lambda_v2<32>(ctx, items, n_items, F, ent, nnz, sym, mode, arg, check_finite, want_lambda, out, err.p));
    else if (F <= 64) AS_TRY(launch_lambda_v2<64>(ctx, items, n_items, F, ent, nnz, sym, mode, arg, check_finite, want_lambda, out, err.p));
    else AS_TRY(launch_lambda_v2<128>(ctx, items, n_items, F, ent, nnz, sym, mode, arg, check_finite, want_lambda, out, err.p));
    if (check_finite) {
      int h = 0;
      AS_HIP(ctx, copy_d2h(ctx, &h, err.p, sizeof(int)));
      AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (h) return fail(ctx, AS_ERR_NONFINITE, as_status_message(AS_ERR_NONFINITE));
    }
    return AS_OK;
  }
  if (F > 128 && ctx->opt_lambda_wide) {  // one wave per item, values in registers (taumode_wide.hip)
    const bool sym = ent_upper != nullptr;
    bool handled = false;
    AS_TRY(lambda_wide(ctx, items, n_items, F, sym ? (const void *)ent_upper : (const void *)ent, sym ? nnz_upper : nnz, sym, mode, arg,
                       check_finite, want_lambda, out, err.p, &handled));
    if (handled) {
      if (check_finite) {
        int h = 0;
        AS_HIP(ctx, copy_d2h(ctx, &h, err.p, sizeof(int)));
        AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (h) return fail(ctx, AS_ERR_NONFINITE, as_status_message(AS_ERR_NONFINITE));
      }
      return AS_OK;
    }
  }
  // lanes per item: smallest S whose tile fits ~40 KB of LDS (4 waves per CU)
  const size_t budget = 40 * 1024;
  int S = 1;
  while (S < 64 && (size_t)F * (64 / S + 1) * 8 > budget) S *= 2;
  if ((size_t)F * (64 / S + 1) * 8 > 150 * 1024) return fail(ctx, AS_ERR_UNSUPPORTED, "n_features too large");
  {
    ProfScope ps(ctx, want_lambda ? "taumode_lambda" : "select_tau");
#define AS_LAUNCH(SS)                                                                                         \
  case SS:                                                                                                    \
    AS_TRY(launch_lambda<SS>(ctx, items, n_items, F, ent, nnz, mode, arg, check_finite, want_lambda, out, err.p)); \
    break;
    switch (S) {
      AS_LAUNCH(1)
      AS_LAUNCH(2)
      AS_LAUNCH(4)
      AS_LAUNCH(8)
      AS_LAUNCH(16)
      AS_LAUNCH(32)
      AS_LAUNCH(64)
    }
#undef AS_LAUNCH
  }
  if (check_finite) {
    int h = 0;
    AS_HIP(ctx, copy_d2h(ctx, &h, err.p, sizeof(int)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h) return fail(ctx, AS_ERR_NONFINITE, as_status_message(AS_ERR_NONFINITE));
  }
  return AS_OK;
}

int taumode_lambdas(as_ctx *ctx, const double *items, uint64_t n_items, uint64_t F, as_csr *L, int mode,
                    double arg, bool check_finite, double *lambdas_out) {
  if (L->n > F) return fail(ctx, AS_ERR_DIM, "graph has more nodes than the items have features");
  AS_TRY(ensure_packed(ctx, L));
  if (ctx->opt_lambda_v3) {
    bool handled = false;
    AS_TRY(lambda_v3(ctx, items, n_items, F, L, mode, arg, check_finite, lambdas_out, &handled));
    if (handled) return AS_OK;
  }
  return run_lambda(ctx, items, n_items, F, (const PackedEntry *)L->packed, L->nnz,
                    (const PackedEntry *)L->packed_upper, L->nnz_upper, mode, arg, check_finite, true, lambdas_out);
}

int select_tau_rows(as_ctx *ctx, const double *rows, uint64_t n_rows, uint64_t len, int mode, double arg,
                    double *taus) {
  return run_lambda(ctx, rows, n_rows, len, nullptr, 0, nullptr, 0, mode, arg, false, false, taus);
}

}  // namespace asx
