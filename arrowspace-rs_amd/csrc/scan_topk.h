// scan_topk.h -- EXACT "one row against all rows" scan with a wave-resident top-K list.
//
// This is the arithmetic-exact leg of the neighbour search (K2) and of the search scan (K9):
// every pair's dot product is a sequential k = 0..d-1 chain of separate f64 multiply and add
// (no FMA), i.e. bit-identical to the reference's `iter().zip().map(|(a,b)| a*b).sum()`
// (src/core.rs:185-197, src/tests/test_helpers.rs:95-99).  It is used
//   (a) as the whole neighbour search / search scan when the MFMA candidate pass is disabled,
//   (b) to re-do the rows the MFMA candidate pass could not certify.
//
// Geometry: 256 threads = 4 waves; wave w owns one "A row" (a graph node / a query); the block
// streams "B rows" (all nodes / all items) in tiles of 64 (one per lane) staged through LDS in
// k-chunks of KC.  The running top-K of a wave's row lives in registers, one entry per lane
// (K <= 64), ordered by (key asc, index asc); insertion is a ballot + one-lane shift.
#pragma once

#include <hip/hip_runtime.h>

#include "common.h"

namespace asx {

constexpr int kScanKC = 128;   // k-chunk (f64 elements)
constexpr int kScanTJ = 64;    // B rows per tile = wave width
constexpr int kScanWaves = 4;  // A rows per block

__device__ __forceinline__ bool lex_less(double ak, uint32_t ai, double bk, uint32_t bi) {
  return ak < bk || (ak == bk && ai < bi);
}

__device__ __forceinline__ double shfl_f64(double v, int src) {
  return __shfl(v, src, 64);
}

// One wave inserts the candidates flagged in `m` (ascending lane order = ascending index).
__device__ __forceinline__ void wave_insert(uint64_t m, double ckey, uint32_t cidx, int lane, int K,
                                            double &my_key, uint32_t &my_idx, double &thr_key,
                                            uint32_t &thr_idx) {
  while (m) {
    const int src = __ffsll((unsigned long long)m) - 1;
    m &= m - 1;
    const double k = shfl_f64(ckey, src);
    const uint32_t id = (uint32_t)__shfl((int)cidx, src, 64);
    if (!lex_less(k, id, thr_key, thr_idx)) continue;
    const bool before = lex_less(my_key, my_idx, k, id);
    const int pos = __popcll(__ballot(before));
    const double up_k = __shfl_up(my_key, 1, 64);
    const uint32_t up_i = (uint32_t)__shfl_up((int)my_idx, 1, 64);
    if (lane == pos) {
      my_key = k;
      my_idx = id;
    } else if (lane > pos) {
      my_key = up_k;
      my_idx = up_i;
    }
    thr_key = shfl_f64(my_key, K - 1);
    thr_idx = (uint32_t)__shfl((int)my_idx, K - 1, 64);
  }
}

// Policy concept:
//   __device__ bool key(double dot, uint32_t a /*global A row*/, uint32_t j /*B row*/, double &key) const
//   returns false when the pair is excluded; sets *err_flag through nan_seen() when needed.
template <typename Policy>
__global__ __launch_bounds__(256) void scan_topk_kernel(
    const double *__restrict__ A, uint64_t lda,       // A rows, leading dimension
    const double *__restrict__ B, uint64_t ldb,       // B rows
    uint32_t nB, uint32_t d,
    const uint32_t *__restrict__ a_ids,               // optional list of A rows (nullptr: a_begin + r)
    uint32_t a_begin, uint32_t a_count, uint32_t K,
    const double *__restrict__ lo_key,                // optional per-A-row exclusive lower bound
    const uint32_t *__restrict__ lo_idx,
    Policy pol, double *__restrict__ out_key, uint32_t *__restrict__ out_idx, uint32_t out_ld,
    uint32_t out_off, int *__restrict__ err_flag) {
  __shared__ double sB[kScanTJ][kScanKC + 1];
  __shared__ double sA[kScanWaves][kScanKC];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;

  for (uint32_t blk = blockIdx.x; blk * kScanWaves < a_count; blk += gridDim.x) {
    const uint32_t r = blk * kScanWaves + wv;  // position in the A list
    const bool have = r < a_count;
    const uint32_t a = have ? (a_ids ? a_ids[r] : a_begin + r) : 0u;

    double my_key = __builtin_inf();
    uint32_t my_idx = kNoIdx;
    double thr_key = __builtin_inf();
    uint32_t thr_idx = kNoIdx;
    double lok = -__builtin_inf();
    uint32_t loi = 0;
    bool use_lo = false;
    if (lo_key && have) {
      lok = lo_key[r];
      loi = lo_idx[r];
      use_lo = true;
    }

    for (uint32_t j0 = 0; j0 < nB; j0 += kScanTJ) {
      double acc = 0.0;
      for (uint32_t k0 = 0; k0 < d; k0 += kScanKC) {
        const uint32_t kc = min((uint32_t)kScanKC, d - k0);
        __syncthreads();  // previous chunk fully consumed
        // stage B tile chunk: 64 rows x kc
        for (uint32_t e = tid; e < kScanTJ * kScanKC; e += 256) {
          const uint32_t rr = e / kScanKC, kk = e % kScanKC;
          const uint32_t j = j0 + rr;
          sB[rr][kk] = (j < nB && kk < kc) ? B[(uint64_t)j * ldb + k0 + kk] : 0.0;
        }
        for (uint32_t e = tid; e < kScanWaves * kScanKC; e += 256) {
          const uint32_t ww = e / kScanKC, kk = e % kScanKC;
          const uint32_t rw = blk * kScanWaves + ww;
          double v = 0.0;
          if (rw < a_count && kk < kc) {
            const uint32_t aw = a_ids ? a_ids[rw] : a_begin + rw;
            v = A[(uint64_t)aw * lda + k0 + kk];
          }
          sA[ww][kk] = v;
        }
        __syncthreads();
        // sequential exact chain: separate multiply and add roundings
        for (uint32_t kk = 0; kk < kc; ++kk) acc = __dadd_rn(acc, __dmul_rn(sA[wv][kk], sB[lane][kk]));
      }
      const uint32_t j = j0 + lane;
      double ckey = __builtin_inf();
      bool valid = false;
      if (have && j < nB) {
        valid = pol.key(acc, a, j, ckey, err_flag);
        if (valid && use_lo) valid = lex_less(lok, loi, ckey, j);
        if (valid) valid = lex_less(ckey, j, thr_key, thr_idx);
      }
      const uint64_t m = __ballot(valid);
      if (m) wave_insert(m, ckey, j, lane, (int)K, my_key, my_idx, thr_key, thr_idx);
    }
    if (have && lane < (int)K) {
      out_key[(uint64_t)r * out_ld + out_off + lane] = my_key;
      out_idx[(uint64_t)r * out_ld + out_off + lane] = my_idx;
    }
  }
}

// ---- policies ---------------------------------------------------------------------------------

// Graph stage: cosine distance, self excluded.  cos := 0 when |a||b| <= 1e-12
// (src/tests/test_helpers.rs:93-105); plain 1 - cos unless rectify.
struct GraphDistPolicy {
  const double *norms;
  int rectify;
  __device__ __forceinline__ bool key(double dot, uint32_t a, uint32_t j, double &out, int *) const {
    if (a == j) return false;
    const double denom = __dmul_rn(norms[a], norms[j]);
    double c = denom > 1e-12 ? __ddiv_rn(dot, denom) : 0.0;
    if (rectify) {
      c = fmin(fmax(c, -1.0), 1.0);
      c = fmax(c, 0.0);
    }
    out = __dsub_rn(1.0, c);
    return !(out != out);  // NaN pairs never enter the list
  }
};

// Search stage: key = -score, score = alpha*cos + (1-alpha)*(1 - min(|lq - li|, 1))
// (src/core.rs:135-138,156-175,229-239).  cos := 0 when the norm product is 0.
struct SearchScorePolicy {
  const double *q_norms;   // per A row (query)
  const double *q_lambdas;
  const double *x_norms;   // per B row (item)
  const double *x_lambdas;
  double alpha;
  __device__ __forceinline__ bool key(double dot, uint32_t a, uint32_t j, double &out, int *err) const {
    const double denom = __dmul_rn(q_norms[a], x_norms[j]);
    const double c = denom > 0.0 ? __ddiv_rn(dot, denom) : 0.0;
    const double diff = fabs(__dsub_rn(q_lambdas[a], x_lambdas[j]));
    const double lam = __dsub_rn(1.0, fmin(diff, 1.0));
    const double s = __dadd_rn(__dmul_rn(alpha, c), __dmul_rn(__dsub_rn(1.0, alpha), lam));
    if (s != s) {
      atomicOr(err, 1);
      return false;
    }
    out = -s;
    return true;
  }
};

}  // namespace asx
