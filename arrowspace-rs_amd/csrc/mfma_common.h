// mfma_common.h -- shared pieces of the MFMA candidate passes (f64 and f32) of K2 / K9.
//
// Scheme (exactness first), see DESIGN.md section 4:
//   1. CANDIDATES  per row, a superset of the wanted top-K from an MFMA contraction over
//      fragment-packed unit rows (never materialising the n x n product);
//   2. EXACT RE-RANK of those candidates with the reference's f64 arithmetic;
//   3. CERTIFICATION against a rigorous rounding bound; rows that fail go to the next, more
//      precise stage (f32 MFMA -> f64 MFMA -> exact scan), so the output is bit-identical to the
//      exact path.
#pragma once

#include <cmath>

#include "scan_topk.h"

namespace asx {

constexpr int kTileRows = 128;       // rows per packed tile
constexpr int kMaxNC = 4;            // A fragments stay in registers up to d_pad = 128 (NC = d_pad / 32)
constexpr uint32_t kQueueCap = 512;  // per-wave queue of elements that passed the threshold filter

// LDS traffic between lanes of ONE wave: the hardware keeps a wave's LDS operations in order, so
// only the compiler has to be told not to move memory operations across this point.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Which global row an A-side row r is (self exclusion, per-row epilogue data).
struct ARows {
  const uint32_t *ids;  // gathered list, or nullptr
  uint32_t row0;        // contiguous range start when ids == nullptr
  __device__ __forceinline__ uint32_t gid(uint32_t r) const { return ids ? ids[r] : row0 + r; }
};

// ---- stage launchers (implemented in mfma_topk_f64.hip / mfma_topk_f32.hip) -------------------
// mode 0: graph (key = -cos, self excluded); mode 1: search (key = -(alpha cos + (1-alpha) lam)).
struct CandProblem {
  int mode;
  // A side
  const double *A;        // source rows (unpacked), leading dimension d
  const double *a_norms;  // indexed by global A id
  ARows arows;
  uint32_t nA;            // rows in this stage (range length or list length)
  // B side
  const double *B;
  const double *b_norms;
  uint32_t nB;
  uint32_t d;
  // search-only epilogue data
  const double *q_lambdas;  // by global A id
  const double *x_lambdas;  // by B row
  double alpha;
  int lam_f32;              // f32 candidate pass: evaluate the lambda term in f32 (small lambdas), else f64
  // output: nA x splits x KC candidates, key = approximate key (ascending = better), idx = B row
  uint32_t KC, splits;
  double *cand_key;
  uint32_t *cand_idx;
};

// Packed operand images.  The B image of the full matrix is built once per call (or cached in an
// index) and shared by all stages of the same precision.
struct PackedF64 {
  DevBuf<double> buf;
  const double *p = nullptr;
  uint32_t tiles = 0;
};
struct PackedF32 {
  DevBuf<float> buf;
  const float *p = nullptr;
  uint32_t tiles = 0;
};

int pack_f64(as_ctx *ctx, const double *R, const double *norms, const uint32_t *row_ids, uint64_t row0, uint64_t n,
             uint32_t d, double *P, uint32_t tiles);
int pack_f32(as_ctx *ctx, const double *R, const double *norms, const uint32_t *row_ids, uint64_t row0, uint64_t n,
             uint32_t d, float *P, uint32_t tiles);
uint32_t packed_units(uint32_t d);                  // 32-column units per packed row (whole 128-wide panels beyond 4)
uint64_t packed_elems(uint32_t tiles, uint32_t d);  // elements of a packed image with `tiles` 128-row tiles

int candidates_f64(as_ctx *ctx, const CandProblem &pb, const double *PA, uint32_t a_tiles, const double *PB,
                   uint32_t b_tiles, const char *prof_name);
int candidates_f32(as_ctx *ctx, const CandProblem &pb, const float *PA, uint32_t a_tiles, const float *PB,
                   uint32_t b_tiles, const char *prof_name);

bool fits_f32(uint32_t KC);  // does a candidate capacity fit the kernel's LDS budget?
bool fits_f64(uint32_t KC);

// rounding bound between an approximate cosine of a stage and the exact (oracle) cosine
inline double cos_delta_f64(uint32_t d) { return (double)(d + 16) * 2.3e-16; }
inline double cos_delta_f32(uint32_t d) {
  const uint32_t dp = packed_units(d) * 32;
  return (double)(dp + 4) * 5.9604644775390625e-08 * 1.01 + 1e-13;  // (d+4) 2^-24 + f64 slack
}

// split-bf16 products (mfma_join_bf16.hip, mfma_theta_bf16.hip): x = hi + lo + e, |e| <= 2^-16 |x| (two roundings to
// 8 significant bits), the pass sums hi*hi + hi*lo + lo*hi exactly representable products in f32:
//   dropped terms  |lo lo'| + |e x'| + |x e'| <= 3 * 2^-16 |x||x'|, summed over k: <= 3 * 2^-16 (Cauchy-Schwarz, unit rows)
//   accumulation   N = 3 * d_pad products plus the carried accumulator of every MFMA; 2^-23 per addition (covers
//                  a truncating adder) on a running sum of magnitude <= 1 + 2^-7
//   operands       two roundings of the unit rows to f32
inline double cos_delta_bf16x3(uint32_t d) {
  const uint32_t ks = (d + 15) / 16, dp = ks * 16;
  const double split = 3.0 * 1.52587890625e-05 * 1.02;
  const double accum = (double)(3 * dp + 3 * ks + 2) * 1.1920928955078125e-07 * 1.02;
  // + the thresholds of the two-phase join blocks: held lowered by eps, eps added back for the exact test (two f32 roundings)
  return split + accum + 2.0 * 5.9604644775390625e-08 + 3e-7 + 1e-13;
}


// one f16 plane of the unit rows scaled by 2^7 (mfma_join_f16.hip): element error <= 2^-11 |x| + 2^-14 (the second term
// covers a flushed subnormal), so  sum |e||x'| + |x||e'| + |e||e'|  <=  2 (2^-11 + 2^-14 sqrt(d_pad) / 128) (1 + 2^-10) in cosine
// units (Cauchy-Schwarz over unit rows; the double rounding f64 -> f32 -> f16 of the packer is inside the 1 + 2^-10);
// the f16 x f16 products are exact in f32; accumulation: d_pad products + the carried accumulator of every MFMA + the
// threshold C = -theta 2^14 the sum starts from, |running sum| < 2^15 so 2^-8 per addition (a truncating adder allowed)
// = 2^-22 in cosine units; one more rounding where the listed score is recovered as acc + theta, and up to three ulps (of a
// value below 2: 3.6e-7) where the masked pass keeps two flag bits in the mantissa of a queued score.
inline double cos_delta_f16_accum(uint32_t d) {
  uint32_t ks = (d + 15) / 16;
  if (ks > 8) ks = (ks + 3) / 4 * 4;  // wide rows are contracted in whole 64-wide panels (zero padded)
  const uint32_t dp = ks * 16;
  return (double)(dp + ks + 4) * 0x1p-22 * 1.02 + 5e-7;
}
inline double cos_delta_f16(uint32_t d) {
  const uint32_t ks = (d + 15) / 16, dp = ks * 16;
  const double plane = 2.0 * (0x1p-11 + 0x1p-14 * std::sqrt((double)dp) / 128.0) * (1.0 + 0x1p-10);
  return plane + cos_delta_f16_accum(d);
}

}  // namespace asx
