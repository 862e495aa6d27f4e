// mfma_stages.h -- what the staged drivers of the MFMA neighbour search (mfma_drivers.hip: K2) and of the search scan
// (search_drivers.hip: K9) share: the bounds of the certification, the generic exact re-rank kernels of joined lists, the pruning
// by stored scores, small helper kernels, and the entry points of the stage kernels in the other translation units.
// (kernels without template parameters are `static`: the header is included by two translation units)
#pragma once
#include "mfma_common.h"

namespace asx {

// ---- exact re-rank + certification ------------------------------------------------------------
// One wave per A row.  C = splits*KC candidates.  Policy = the exact scan's policy (same formulas).
struct GraphBound {
  double delta;
  int rectify;
  // lower bound on the exact key (distance) of any column that is NOT a candidate, given the
  // largest approximate key (= -smallest kept cosine) among full candidate lists
  __device__ __forceinline__ double lower_bound(double worst_approx_key) const {
    const double cmax = -worst_approx_key + delta;  // its exact cosine is at most this
    return rectify ? 1.0 - fmax(cmax, 0.0) : 1.0 - cmax;
  }
};
struct SearchBound {
  double delta;
  __device__ __forceinline__ double lower_bound(double worst_approx_key) const { return worst_approx_key - delta; }
};

// Exact dot product in the reference's arithmetic (sequential multiply-add chain over k, two roundings per
// step, src/core.rs:204-210), with the operands fetched 32 B at a time: each lane walks its own candidate row,
// so an 8-byte load per step would pull every cache line of that row eight times through L1.
__device__ __forceinline__ double exact_dot_seq(const double *__restrict__ x_lds, const double *__restrict__ b,
                                                uint32_t d) {
  double acc = 0.0;
  uint32_t k = 0;
  if (((uintptr_t)b & 31) == 0) {
    typedef double d4 __attribute__((ext_vector_type(4)));
    const d4 *b4 = reinterpret_cast<const d4 *>(b);
    // (Tried in round 5: eight loads in flight per lane, then the 32 steps of their part of the chain.  No gain for the second
    // re-rank of 1 M x 768 (8.9 ms for 4 840 open rows) and the search's re-rank of wide items went 7.9 -> 9.7 ms: the registers
    // of the waiting loads cost more than the round trips they overlap.  Not kept.)
    for (; k + 4 <= d; k += 4) {
      const d4 v = __builtin_nontemporal_load(b4 + (k >> 2));
      acc = __dadd_rn(acc, __dmul_rn(x_lds[k], v.x));
      acc = __dadd_rn(acc, __dmul_rn(x_lds[k + 1], v.y));
      acc = __dadd_rn(acc, __dmul_rn(x_lds[k + 2], v.z));
      acc = __dadd_rn(acc, __dmul_rn(x_lds[k + 3], v.w));
    }
  }
  for (; k < d; ++k) acc = __dadd_rn(acc, __dmul_rn(x_lds[k], b[k]));
  return acc;
}

template <class Policy, class Bound>
__global__ __launch_bounds__(256) void rerank_kernel(const double *__restrict__ A, uint64_t lda,
                                                     const double *__restrict__ B, uint64_t ldb, uint32_t d,
                                                     const uint32_t *__restrict__ a_ids, uint32_t a_begin,
                                                     uint32_t a_count, uint32_t out_row0, uint32_t K, uint32_t KC,
                                                     uint32_t splits, const double *__restrict__ cand_key,
                                                     const uint32_t *__restrict__ cand_idx,
                                                     const uint8_t *__restrict__ force_exact, Policy pol, Bound bnd,
                                                     double *__restrict__ out_key, uint32_t *__restrict__ out_idx,
                                                     uint32_t *__restrict__ fail_rows,
                                                     uint32_t *__restrict__ fail_count, int *__restrict__ err_flag) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t C = splits * KC;
  // per wave: row copy (d) + exact keys (C) + idx (C)
  double *xrow = reinterpret_cast<double *>(smem) + (size_t)wv * (d + C);
  double *ek = xrow + d;
  uint32_t *ei = reinterpret_cast<uint32_t *>(reinterpret_cast<double *>(smem) + (size_t)4 * (d + C)) + (size_t)wv * C;

  for (uint32_t r = blockIdx.x * 4 + wv; r < a_count; r += gridDim.x * 4) {
    const uint32_t a = a_ids ? a_ids[r] : a_begin + r;
    for (uint32_t k = lane; k < d; k += 64) xrow[k] = A[(uint64_t)a * lda + k];
    wave_sync();
    // worst approximate key over FULL lists (a list that is not full saw every column of its split)
    double worst = __builtin_inf();
    bool bounded = false;
    for (uint32_t s = 0; s < splits; ++s) {
      const size_t o = ((size_t)r * splits + s) * KC + KC - 1;
      if (cand_idx[o] != kNoIdx) {  // full list: columns outside it have approx key >= its last key
        bounded = true;
        worst = fmin(worst, cand_key[o]);
      }
    }
    // exact keys of the candidates
    for (uint32_t c = lane; c < C; c += 64) {
      const uint32_t j = cand_idx[(size_t)r * C + c];
      double key = __builtin_inf();
      uint32_t id = kNoIdx;
      if (j != kNoIdx) {
        const double acc = exact_dot_seq(xrow, B + (uint64_t)j * ldb, d);
        double kx;
        if (pol.key(acc, a, j, kx, err_flag)) {
          key = kx;
          id = j;
        }
      }
      ek[c] = key;
      ei[c] = id;
    }
    wave_sync();
    // rank by (key, idx); ranks < K go to the output
    double kth = -__builtin_inf();
    uint32_t n_valid = 0;
    for (uint32_t c = lane; c < C; c += 64) {
      const double kc = ek[c];
      const uint32_t ic = ei[c];
      if (ic == kNoIdx) continue;
      ++n_valid;
      uint32_t rank = 0;
      for (uint32_t c2 = 0; c2 < C; ++c2) rank += lex_less(ek[c2], ei[c2], kc, ic) ? 1u : 0u;
      if (rank < K) {
        out_key[(uint64_t)(r + out_row0) * K + rank] = kc;
        out_idx[(uint64_t)(r + out_row0) * K + rank] = ic;
        kth = fmax(kth, kc);
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      kth = fmax(kth, __shfl_xor(kth, o, 64));
      n_valid += __shfl_xor(n_valid, o, 64);
    }
    for (uint32_t s = n_valid + lane; s < K; s += 64) {  // fewer than K candidates exist at all
      out_key[(uint64_t)(r + out_row0) * K + s] = __builtin_inf();
      out_idx[(uint64_t)(r + out_row0) * K + s] = kNoIdx;
    }
    bool ok = true;
    if (bounded) ok = n_valid >= K && kth < bnd.lower_bound(worst);
    if (force_exact && force_exact[a]) ok = false;
    if (!ok && lane == 0) fail_rows[atomicAdd(fail_count, 1u)] = r + out_row0;
  }
}

// Pairs with 0 < |a||b| <= 1e-12 get cos := 0 in the exact rule but a genuine cosine from the unit
// rows; a row that can form such a pair is always re-done exactly.  (Exactly-zero rows are
// consistent: their unit row is 0, so the product is 0 as well.)
// (min_bits: the bits of the smallest positive norm, as min_norm_kernel leaves them ON THE DEVICE -- read here instead of on the host:
// the read-back in between was a stream synchronisation of every build for one kernel argument)
static __global__ void flag_tiny_norm_rows_kernel(const double *__restrict__ norms, uint32_t n, const unsigned long long *__restrict__ min_bits,
                                           uint8_t *__restrict__ flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const double min_pos_norm = __longlong_as_double((long long)*min_bits);
  if (i < n) {
    const double x = norms[i];
    // non-finite norms as well: an infinite norm meets a NaN norm as cos := 0 under the exact rule (the product of
    // the norms is NaN, which is not > 1e-12), which the unit-row contraction cannot reproduce
    flag[i] = (x != x) || !(x < __builtin_inf()) || (x > 0.0 && !(x * min_pos_norm > 1e-12)) ? 1 : 0;
  }
}

static __global__ void min_norm_kernel(const double *__restrict__ norms, uint32_t n, unsigned long long *out) {
  // norms are >= 0 (or NaN): the bit pattern of a non-negative double orders like the value
  double v = __builtin_inf();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double x = norms[i];
    if (x > 0.0 && x < v) v = x;  // smallest POSITIVE norm
  }
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0) lower_min_u64(out, (unsigned long long)__double_as_longlong(v));
}

static __global__ void scatter_rows_kernel(const uint32_t *__restrict__ rows, uint32_t count, uint32_t K,
                                    const double *__restrict__ src_key, const uint32_t *__restrict__ src_idx,
                                    double *__restrict__ dst_key, uint32_t *__restrict__ dst_idx) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count * K) return;
  const uint32_t r = t / K, s = t - r * K;
  dst_key[(uint64_t)rows[r] * K + s] = src_key[t];
  dst_idx[(uint64_t)rows[r] * K + s] = src_idx[t];
}


static __global__ void set_u64_kernel(unsigned long long *p, unsigned long long v) { *p = v; }

static __global__ void gather_u32_kernel(const uint32_t *__restrict__ table, const uint32_t *__restrict__ pos,
                                  uint32_t count, uint32_t add, uint32_t *__restrict__ out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < count) out[t] = (table ? table[pos[t]] : pos[t]) + add;
}

// key of a listed row for the order of the image: its position there
// t[p] = max(t[p], band[p] - by): the thresholds of the own neighbourhood where they are the higher ones (+inf stays +inf)
// ... and where the sampled threshold is the higher one by no more than `slack`, the certified cut is taken all the same (round 5): a
// row cut at band - by is certified by construction, a row cut at a sampled value that sits within a few deltas of its K-th best
// cosine is a coin toss -- rows of 768 columns have the cosines of a whole cluster within a few deltas of each other, and 2 % of them
// (19.6 k of 1 M) went on to a 36 ms second pass for lists a handful of entries shorter
static __global__ void theta_band_merge_kernel(float *__restrict__ t, const float *__restrict__ band, uint32_t n, float by, float slack) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n && t[p] < __builtin_inff()) {
    const float b = band[p] - by;
    if (b > t[p] || (b == b && t[p] - b <= slack)) t[p] = b;
  }
}
static __global__ void position_keys_kernel(const uint32_t *__restrict__ ids, const uint32_t *__restrict__ inv, uint32_t count, double *__restrict__ keys) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < count) keys[t] = (double)inv[ids[t]];
}

static uint32_t pick_splits(uint32_t a_items, uint32_t b_tiles) {
  // keep >= ~4 work items per CU when the A side is short (query batches, small graphs, fall-back lists)
  uint32_t s = 1;
  while (a_items * s < 1024 && s * 2 <= b_tiles && s < 16) s *= 2;
  return s;
}

// Search scans: the A side (query tiles) is short, so the column range is cut into `s` pieces to fill the
// 256 persistent workgroups.  Rounds of work items are quantised (items / 256 rounded up) and every extra
// piece adds KC candidates per query to the re-rank plus one more list warm-up, so s is chosen to minimise
//   time / ideal sweep time = (rounds * 256) / items  +  s * (re-rank + warm-up cost of one piece / ideal sweep).
// The relative cost was measured at n = 1 M columns (5.7 ms re-rank for 16 pieces beside a 27 ms ideal sweep)
// and scales as 1 / n.
static uint32_t pick_splits_scan(uint32_t a_items, uint32_t b_tiles, uint64_t n_cols) {
  const uint32_t G = 256;
  const uint32_t smax = std::max(1u, std::min(b_tiles / 2, std::max(16u, G / std::max(1u, a_items))));
  const double per_piece = 0.013 * (1.0e6 / (double)std::max<uint64_t>(n_cols, 1));
  uint32_t best = 1;
  double best_cost = 1e300;
  for (uint32_t s = 1; s <= smax; ++s) {
    const uint64_t items = (uint64_t)a_items * s;
    const uint64_t rounds = (items + G - 1) / G;
    const double cost = (double)(rounds * G) / (double)items + per_piece * s;
    if (cost < best_cost - 1e-12) best_cost = cost, best = s;
  }
  return best;
}

static uint32_t cand_capacity(as_ctx *ctx, uint32_t K) {
  uint32_t kc = K + (uint32_t)ctx->opt_cand_slack;
  kc = (kc + 7) / 8 * 8;
  return std::min<uint32_t>(kc, 64);
}

// The exact re-rank kernels stage the A row and up to 1024 candidate keys per wave in LDS: 4 x ((d + C) x 8 + C x 4)
// bytes with C <= 1024 must fit the 160 KB of a CU.  Wider rows (a feature graph over thousands of centroids, build(rows)
// with every item its own centroid) take the exact scan, which tiles k.
static bool rerank_fits(uint64_t d) { return 32 * d + 49152 <= 160 * 1024; }

// column pieces of a top-k pass: the re-rank stages the row and the splits x KC candidates of a wave in LDS
// (launch_rerank: 4 x ((d + C) x 8 + C x 4) bytes), which bounds how finely the columns may be cut
static uint32_t clamp_splits(uint32_t splits, uint32_t d, uint32_t KC) {
  const uint64_t budget = 160 * 1024 / 4;
  const uint64_t room = budget > 8ull * d ? (budget - 8ull * d) / 12 : 0;  // candidates per wave
  return std::max<uint32_t>(1, std::min<uint64_t>(splits, room / std::max<uint32_t>(KC, 1)));
}

constexpr uint32_t kExactDirect = 128;  // fewer failed rows than this go straight to the exact scan ...
// ... when the matrix is small: the exact scan is one wave per row over ALL columns (300 ms at 1 M x 128 however
// few the rows), an MFMA stage over the failed rows a few tens of ms
static bool exact_direct(uint32_t n_rows, uint32_t n_cols) { return n_rows <= kExactDirect && n_cols < 16384; }

template <class Policy, class Bound>
static int launch_rerank(as_ctx *ctx, const char *name, const double *A, const double *B, uint32_t d,
                         const uint32_t *a_ids, uint32_t a_begin, uint32_t a_count, uint32_t out_row0, uint32_t K,
                         uint32_t KC, uint32_t splits, const double *ckey, const uint32_t *cidx, const uint8_t *force, Policy pol,
                         Bound bnd, double *out_key, uint32_t *out_idx, uint32_t *fail_rows, uint32_t *fail_count,
                         int *err) {
  const uint32_t C = splits * KC;
  const size_t lds = (size_t)4 * (d + C) * sizeof(double) + (size_t)4 * C * sizeof(uint32_t);
  ProfScope ps(ctx, name);
  AS_HIP(ctx, hipFuncSetAttribute((const void *)rerank_kernel<Policy, Bound>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((rerank_kernel<Policy, Bound>), dim3(std::min<uint64_t>(cdiv(a_count, 4), 4096)), dim3(256), lds,
                     ctx->stream, A, (uint64_t)d, B, (uint64_t)d, d, a_ids, a_begin, a_count, out_row0, K, KC, splits, ckey, cidx,
                     force, pol, bnd, out_key, out_idx, fail_rows, fail_count, err);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

static int read_u32(as_ctx *ctx, const uint32_t *dev, uint32_t *host) {
  AS_TRY(deferred_stats_issue(ctx));  // (parked counters ride along with this wait: common.h)
  AS_HIP(ctx, copy_d2h(ctx, host, dev, sizeof(uint32_t)));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  deferred_stats_apply(ctx);
  return AS_OK;
}

// ---- threshold join (mfma_join_f32.hip): thresholds, exact re-rank, stage driver ------------------
// hashed 1-in-`rate` sample of the rows, in pseudo-random order: position p holds the sample of block
// t = perm(p), id = rate * t + hash(t) % rate.  A plain stride would alias with periodic row orders, and the ORDER
// matters too: the split-bf16 threshold pass bins the sample by position (mod 32, tile parity) and wants a row's
// best partners spread over the bins; a multiplicative map p * g mod ns keeps residues modulo gcd(ns, 32) and
// leaves them in a few bins.  perm = cycle-walked 4-round Feistel network over the next even power of two.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
static __global__ void sample_ids_kernel(uint32_t ns, uint32_t rate, uint32_t half_bits, uint32_t *__restrict__ ids) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= ns) return;
  const uint32_t mask = (1u << half_bits) - 1u;
  uint32_t x = p;
  do {  // bijection on [0, 2^(2 half_bits)); walking the cycle until it lands below ns makes it one on [0, ns)
    uint32_t l = x >> half_bits, r = x & mask;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
      const uint32_t nl = r;
      r = l ^ (mix32(r * 0x9E3779B9u + k) & mask);
      l = nl;
    }
    x = (l << half_bits) | r;
  } while (x >= ns);
  const uint32_t t = x;
  ids[p] = rate * t + mix32(t ^ 0x5bd1e995u) % rate;
}
static uint32_t sample_half_bits(uint32_t ns) {
  uint32_t b = 1;
  while (b < 16 && (1ull << (2 * b)) < ns) ++b;
  return b;
}

// theta[r] = the m-th best approximate cosine of row r against the sample (the value whose rank among ALL
// rows is about m * rate); padding rows get +inf so that nothing is ever appended for them
static __global__ void join_theta_kernel(const double *__restrict__ ckey, const uint32_t *__restrict__ cidx, uint32_t n,
                                  uint32_t KC, uint32_t m, uint32_t n_pad, float *__restrict__ theta) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_pad) return;
  float t = __builtin_inff();
  if (r < n) {
    const size_t o = (size_t)r * KC + (m - 1);
    t = cidx[o] != kNoIdx ? (float)(-ckey[o]) : -__builtin_inff();
  }
  theta[r] = t;
}

// thresholds of the second pass: a row whose list overflowed (in any of its sources) keeps its threshold, the others
// get the looser one.  theta by global row; theta2 by (row - t2_base); the lists of row g are those of index
// (inv ? inv[g] : g) - list_base: kept by row of the range (sharded by rows), by image position (own join on a permuted image,
// row_order.hip) or by image position of a range of positions (sharded by positions)
static __global__ void second_theta_kernel(const float *__restrict__ theta, const float *__restrict__ theta2,
                                    const uint32_t *__restrict__ cnt, uint32_t n_src, uint32_t t2_base, uint32_t list_base, uint32_t row_count,
                                    uint32_t cap, const uint32_t *__restrict__ ids, uint32_t count, uint32_t padded,
                                    float *__restrict__ out, const uint32_t *__restrict__ inv, const float *__restrict__ band, float band_by) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= padded) return;
  float t = __builtin_inff();
  if (i < count) {
    const uint32_t g = ids[i], r = g - t2_base;
    const uint32_t lr = (inv ? inv[g] : g) - list_base;
    bool over = false;
    for (uint32_t sidx = 0; sidx < n_src; ++sidx) over = over || cnt[(size_t)sidx * row_count + lr] > cap;
    t = over ? theta[g] : theta2[r];
    // never below what the row's own neighbourhood certifies (K rows reach it: mfma_theta_f16.hip, join_thresholds_f16_band)
    if (band && band[g] - band_by > t) t = band[g] - band_by;
  }
  out[i] = t;
}

// ... with thresholds computed afresh for the gathered rows (t1 / t2 by position in ids)
static __global__ void second_theta_fresh_kernel(const float *__restrict__ t1, const float *__restrict__ t2, const float *__restrict__ theta_first,
                                          const uint32_t *__restrict__ cnt,
                                          uint32_t n_src, uint32_t list_base, uint32_t row_count, uint32_t cap,
                                          const uint32_t *__restrict__ ids, uint32_t count, uint32_t padded, float *__restrict__ out,
                                          const uint32_t *__restrict__ inv, const float *__restrict__ band, float band_by) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= padded) return;
  float t = __builtin_inff();
  if (i < count) {
    const uint32_t g = ids[i];
    const uint32_t lr = (inv ? inv[g] : g) - list_base;
    bool over = false;
    for (uint32_t sidx = 0; sidx < n_src; ++sidx) over = over || cnt[(size_t)sidx * row_count + lr] > cap;
    // a list that ran over although the first threshold was the right one is a dense neighbourhood: the tight threshold and
    // twice the room; one that ran over because the first threshold was far too low is an ordinary row: the looser threshold
    t = (over && !(t1[i] - theta_first[g] > 1e-3f)) ? t1[i] : t2[i];
    if (band && band[g] - band_by > t) t = band[g] - band_by;  // (as in second_theta_kernel)
  }
  out[i] = t;
}

// the check of the grouped thresholds: rows perm[i * stride] (count of them), and how many have a grouped threshold more than
// `slack` below the one against the whole sample
static __global__ void theta_checksum_kernel(const float *__restrict__ t, uint32_t n, unsigned long long *__restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(out, (unsigned long long)__float_as_uint(t[i]) * (unsigned long long)(i % 1021 + 1));
}
static __global__ void theta_check_ids_kernel(const uint32_t *__restrict__ perm, uint32_t stride, uint32_t count, uint32_t *__restrict__ ids) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) ids[i] = perm[(size_t)i * stride];
}
// the sample the grouped thresholds draw from, whole: image position 32 j + hash(j) of every 32 (mfma_theta_f16.hip)
static __global__ void theta_check_sample_kernel(const uint32_t *__restrict__ perm, uint32_t n, uint32_t ns, uint32_t *__restrict__ ids) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ns) return;
  uint32_t x = j * 0x9E3779B9u;
  x ^= x >> 15;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  const uint32_t p = j * 32 + (x & 31u);
  ids[j] = perm[p < n ? p : n - 1];
}
static __global__ void theta_check_count_kernel(const float *__restrict__ theta_pos, uint32_t stride, const float *__restrict__ t_full, uint32_t count,
                                         float slack, uint32_t *__restrict__ bad) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count && theta_pos[(size_t)i * stride] < t_full[i] - slack) atomicAdd(bad, 1u);
}

// ids[i] = perm[pos[i]] (~0 stays ~0); then theta_pos[pos[i]] = max(theta_pos[pos[i]], t[i])
static __global__ void rows_of_positions_kernel(const uint32_t *__restrict__ pos, const uint32_t *__restrict__ perm, uint32_t count, uint32_t *__restrict__ ids) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) ids[i] = pos[i] == 0xFFFFFFFFu ? 0xFFFFFFFFu : perm[pos[i]];
}
static __global__ void theta_raise_kernel(float *__restrict__ theta_pos, const uint32_t *__restrict__ pos, const float *__restrict__ t, uint32_t count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count && pos[i] != 0xFFFFFFFFu && t[i] > theta_pos[pos[i]] && t[i] < __builtin_inff()) theta_pos[pos[i]] = t[i];
}
// flag[p] = 1 where theta[p] lies more than `dip` below the median of the 64 positions of its block (one wave a block)
static __global__ __launch_bounds__(256) void theta_dip_flag_kernel(const float *__restrict__ theta, uint32_t n, float dip, uint32_t *__restrict__ flag) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  const float t = p < n ? theta[p] : __builtin_inff();
  uint32_t below = 0;  // values of the block below mine (ties by lane): my rank
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t l = 0; l < 64; ++l) {
    const float o = __shfl(t, (int)l, 64);
    below += (o < t || (o == t && l < lane)) ? 1u : 0u;
  }
  // the value of rank 31 among the finite ones is close enough to a median; infinite thresholds (rows without a direction) rank last
  const unsigned long long at = __ballot(below == 31u);
  const float med = __shfl(t, at ? (int)__builtin_ctzll(at) : 0, 64);
  if (p < n) flag[p] = (t == t && med < __builtin_inff() && t < med - dip) ? 1u : 0u;
}
static __global__ void list_flagged_kernel(const uint32_t *__restrict__ flag, const uint64_t *__restrict__ rank, uint32_t n, uint32_t room,
                                           uint32_t *__restrict__ out) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n && flag[p] && rank[p] < room) out[rank[p]] = p;
}
// flag[inv[ids[i]]] = 1: the image positions of a list of rows (distinct rows: no two writes meet)
static __global__ void flag_positions_kernel(const uint32_t *__restrict__ ids, const uint32_t *__restrict__ inv, uint32_t count, uint32_t *__restrict__ flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) flag[inv[ids[i]]] = 1u;
}
static __global__ void invert_perm_kernel(const uint32_t *__restrict__ perm, uint32_t n, uint32_t *__restrict__ inv) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) inv[perm[i]] = i;
}

// the words the host reads after the re-rank, next to each other: the survivors of the pruning (spread over `slots` words by
// rerank_prune_kernel; parts == nullptr: no pruning ran) and the number of rows left open
static __global__ void join_tail_words_kernel(const unsigned long long *__restrict__ parts, uint32_t slots, const uint32_t *__restrict__ fail_count,
                                              unsigned long long *__restrict__ out) {
  unsigned long long s = 0;
  if (parts)
    for (uint32_t i = threadIdx.x; i < slots; i += 64) s += parts[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) {
    out[0] = s;
    out[1] = *fail_count;
  }
}

static __global__ void pad_theta_kernel(const float *__restrict__ theta, uint32_t n, uint32_t n_pad, float *__restrict__ out) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n_pad) out[r] = r < n ? theta[r] : __builtin_inff();
}

// candidates per row: total, maximum, rows over capacity (diagnostics for the threshold choice)
static __global__ void join_count_stats_kernel(const uint32_t *__restrict__ cnt, uint32_t n, uint32_t cap,
                                        unsigned long long *__restrict__ out) {
  unsigned long long sum = 0, mx = 0, over = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t c = cnt[i];
    sum += c;
    mx = c > mx ? c : mx;
    over += c > cap ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) {
    sum += __shfl_xor(sum, o, 64);
    over += __shfl_xor(over, o, 64);
    const unsigned long long t = __shfl_xor(mx, o, 64);
    mx = t > mx ? t : mx;
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(out, sum);
    atomicMax(out + 1, mx);
    atomicAdd(out + 2, over);
  }
}

// Pruning of a joined candidate list before any candidate row is fetched (one wave; ei / es = ids and approximate
// cosines of the C gathered candidates in LDS).  tau = the K-th largest approximate cosine: K candidates have an
// exact cosine >= tau - delta, so a candidate below tau - 2 delta is strictly worse than K others and cannot be in
// the top K (nor tie with it).  The survivors are compacted in place (order kept); returns their count.
// rectify: distances of negative cosines all collapse to 1 and tie by index, so nothing is dropped at or below 0.
// tau for a list of at most 64 NS scores (NS = its number of 64-score slots: the loops below have exactly that length; with one
// body for 16 slots and a test per slot, the tests were most of the pruning kernel's instructions for the usual 2 - 3 slots)
template <int NS>
__device__ __forceinline__ bool prune_tau(const float *es, uint32_t C, uint32_t K, double delta, int lane, float *tau_out) {
  uint32_t key[NS];
  bool has_nan = false;
  uint32_t kmax = 0u, kmin = 0xFFFFFFFFu;
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    const uint32_t c = (uint32_t)q * 64u + lane;
    key[q] = 0u;
    if (c < C) {
      const float sc = es[c];
      has_nan = has_nan || sc != sc;
      const uint32_t b = __float_as_uint(sc);
      key[q] = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
      kmax = max(kmax, key[q]);
      kmin = min(kmin, key[q]);
    }
  }
  if (__ballot(has_nan)) return false;  // NaN scores: leave the list alone
  // the scores of a list share their leading bits (cosines of one neighbourhood): the search starts below the common prefix
  for (int o = 32; o > 0; o >>= 1) {
    kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, o, 64));
    kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, o, 64));
  }
  const uint32_t diff = kmax ^ kmin;
  const int top = diff ? 31 - __builtin_clz(diff) : -1;   // highest bit in which two scores differ
  uint32_t prefix = top >= 31 ? 0u : (kmax & ~((2u << (top < 0 ? 0 : top)) - 1u));
  if (top < 0) prefix = kmax;
  // ... and stops where a step is worth less than delta / 8: tau only has to be a LOWER bound of the K-th largest score (a
  // smaller tau keeps more candidates, never fewer), and clearing the low bits of a key lowers its score by less than 2^stop
  // units in the last place of the largest score
  int stop = 0;
  if (delta > 0.0) {
    const float top_score = fabsf(__uint_as_float((kmax & 0x80000000u) ? (kmax & 0x7FFFFFFFu) : ~kmax));
    // top_score < 2^e: one unit in the last place <= 2^(e - 24).  Exponents read off the bits (frexpf and floor(log2()) in f64
    // were a tenth of the pruning kernel's instructions): for a normal x, floor(log2 x) is its unbiased exponent and frexp's e is
    // that + 1; a zero score takes e = 0 as with frexpf (a subnormal one too: any stop is valid, it only moves how tight tau is)
    const uint32_t tb = __float_as_uint(top_score) >> 23;
    const int e = tb ? (int)tb - 126 : 0;
    const unsigned long long db = (unsigned long long)__double_as_longlong(delta / 8.0) >> 52;
    const int lg = db ? (int)(db & 0x7FFu) - 1023 : -1023;
    stop = lg - (e - 24);
    stop = stop < 0 ? 0 : stop;
  }
  for (int bit = top; bit >= stop; --bit) {
    const uint32_t trial = prefix | (1u << bit);
    uint32_t cnt = 0;
#pragma unroll
    for (int q = 0; q < NS; ++q) cnt += (uint32_t)__builtin_popcountll(__ballot(key[q] >= trial));  // (slots past C hold key 0 < trial)
    if (cnt >= K) prefix = trial;
  }
  *tau_out = __uint_as_float((prefix & 0x80000000u) ? (prefix & 0x7FFFFFFFu) : ~prefix);
  return true;
}

__device__ __forceinline__ uint32_t prune_candidates(uint32_t *ei, float *es, uint32_t C, uint32_t K, double delta, int rectify,
                                                     int lane) {
  if (C <= K) return C;
  // tau by bisection over the order-preserving integer image of the scores: the largest v with |{s >= v}| >= K.
  // Counting is a compare + ballot + popcount per 64 scores and round (the scores stay in registers); ranking every
  // score against every other one took 3 x C LDS round trips per row and made the pruning pass compute-bound.
  if (C > 64u * 16u) return C;
  float tau = 0.f;
  bool ok;
  const uint32_t ns = (C + 63u) / 64u;  // wave-uniform
  if (ns <= 2) ok = prune_tau<2>(es, C, K, delta, lane, &tau);
  else if (ns <= 3) ok = prune_tau<3>(es, C, K, delta, lane, &tau);
  else if (ns <= 4) ok = prune_tau<4>(es, C, K, delta, lane, &tau);
  else if (ns <= 8) ok = prune_tau<8>(es, C, K, delta, lane, &tau);
  else ok = prune_tau<16>(es, C, K, delta, lane, &tau);
  if (!ok) return C;
  const double thr = (double)tau - 2.0 * delta;
  if (rectify && thr <= 0.0) return C;
  uint32_t out = 0;
  for (uint32_t base = 0; base < C; base += 64) {
    const uint32_t c = base + lane;
    const bool keep = c < C && (double)es[c] >= thr;
    const uint32_t id = c < C ? ei[c] : 0u;
    const unsigned long long m = __ballot(keep);
    wave_sync();
    if (keep) ei[out + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = id;
    out += (uint32_t)__builtin_popcountll(m);
  }
  wave_sync();
  return out;
}

// The same pruning with a bound PER PAIR: the f16 product of rows i and j differs from their exact cosine by at most
// E_i + E_j + E_i E_j (their measured plane errors, Cauchy-Schwarz per pair) + the accumulation term, where the one bound for all
// pairs has twice the LARGEST plane error of the matrix (a third more at 1 M x 128: 28.8 -> 25 rows fetched per row).  err: the
// plane errors by the ids the list holds (image positions), e_self: the row's own; ed: the bounds d_c of the pairs.  lo_c = s_c - d_c
// replaces the score (tau = a lower bound of the K-th largest lo), a candidate stays when s_c + d_c >= tau: K candidates have an
// exact cosine >= tau, which a dropped one cannot reach.
// pair_bound: (rounded up: the three roundings of the sum and of the subtraction from the score are covered by the 4e-7)
__device__ __forceinline__ float pair_bound(float e_self, float ej, float accum, float floor_delta) {
  return __builtin_fmaxf((e_self + ej + e_self * ej) * (1.f + 1e-6f) + accum + 4e-7f, floor_delta);
}
// es[c] = s_c - d_c and ed[c] = d_c on entry (the kernel computes them as it copies the lists)
__device__ __forceinline__ uint32_t prune_candidates_pairwise(uint32_t *ei, float *es, const float *ed, uint32_t C, uint32_t K, float step_f, int rectify,
                                                              int lane) {
  if (C <= K) return C;
  if (C > 64u * 16u) return C;
  float tau = 0.f;
  bool ok;
  const uint32_t ns = (C + 63u) / 64u;  // wave-uniform
  const double step = (double)step_f;  // (the bisection stops an eighth of this below the K-th largest lo: a lower bound)
  if (ns <= 2) ok = prune_tau<2>(es, C, K, step, lane, &tau);
  else if (ns <= 3) ok = prune_tau<3>(es, C, K, step, lane, &tau);
  else if (ns <= 4) ok = prune_tau<4>(es, C, K, step, lane, &tau);
  else if (ns <= 8) ok = prune_tau<8>(es, C, K, step, lane, &tau);
  else ok = prune_tau<16>(es, C, K, step, lane, &tau);
  if (!ok) return C;
  if (rectify && !(tau > 0.f)) return C;  // (cosines rectified to 0 tie: a dropped candidate could still rank by its id)
  uint32_t out = 0;
  for (uint32_t base = 0; base < C; base += 64) {
    const uint32_t c = base + lane;
    const bool keep = c < C && !((double)es[c] + 2.0 * (double)ed[c] < (double)tau);
    const uint32_t id = c < C ? ei[c] : 0u;
    const unsigned long long m = __ballot(keep);
    wave_sync();
    if (keep) ei[out + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = id;
    out += (uint32_t)__builtin_popcountll(m);
  }
  wave_sync();
  return out;
}


template <class Policy, class Bound>
__global__ __launch_bounds__(256) void rerank_join_kernel(const double *__restrict__ X, const double *__restrict__ XA,
                                                          const uint32_t *__restrict__ a_ids, uint32_t d, uint32_t row_begin,
                                                          uint32_t row_count, const float *__restrict__ theta,
                                                          uint32_t n_src, const uint32_t *__restrict__ cnt_src,
                                                          const uint32_t *__restrict__ cand_src, uint32_t cap,
                                                          uint32_t cmax, uint32_t K,
                                                          const uint8_t *__restrict__ force_exact, Policy pol, Bound bnd,
                                                          double prune_delta, int prune_rectify,
                                                          double *__restrict__ out_key, uint32_t *__restrict__ out_idx,
                                                          uint32_t *__restrict__ fail_rows,
                                                          uint32_t *__restrict__ fail_count, int *__restrict__ err_flag) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double *xrow = reinterpret_cast<double *>(smem) + (size_t)wv * (d + cmax);
  double *ek = xrow + d;
  uint32_t *ei = reinterpret_cast<uint32_t *>(reinterpret_cast<double *>(smem) + (size_t)4 * (d + cmax)) + (size_t)wv * cmax;
  for (uint32_t r = blockIdx.x * 4 + wv; r < row_count; r += gridDim.x * 4) {
    const uint32_t a = a_ids ? a_ids[r] : row_begin + r;  // A row (XA = the A matrix: X itself, or the queries)
    const uint32_t ti = a_ids ? r : a;                     // thresholds of a gathered list are indexed by position
    for (uint32_t k = lane; k < d; k += 64) xrow[k] = XA[(uint64_t)a * d + k];
    // gather the candidate ids of all sources into ei[0..C)
    uint32_t C = 0;
    bool overflow = false;
    for (uint32_t sidx = 0; sidx < n_src; ++sidx) {
      const uint32_t total = cnt_src[(size_t)sidx * row_count + r];
      uint32_t take = min(total, cap);
      overflow = overflow || total > cap;
      if (C + take > cmax) {
        overflow = true;
        take = cmax - C;
      }
      const uint2 *src = reinterpret_cast<const uint2 *>(cand_src) + ((size_t)sidx * row_count + r) * cap;
      float *es = reinterpret_cast<float *>(ek);  // approximate cosines, until the exact keys overwrite them
      for (uint32_t c = lane; c < take; c += 64) {
        const uint2 e = src[c];
        ei[C + c] = e.x;
        es[C + c] = __uint_as_float(e.y);
      }
      C += take;
    }
    wave_sync();
    if (prune_delta >= 0.0) C = prune_candidates(ei, reinterpret_cast<float *>(ek), C, K, prune_delta, prune_rectify, lane);
    for (uint32_t c = lane; c < C; c += 64) {
      const uint32_t j = ei[c];
      const double acc = exact_dot_seq(xrow, X + (uint64_t)j * d, d);
      double key = __builtin_inf(), kx;
      uint32_t id = kNoIdx;
      if (pol.key(acc, a, j, kx, err_flag)) {
        key = kx;
        id = j;
      }
      ek[c] = key;
      ei[c] = id;
    }
    wave_sync();
    double kth = -__builtin_inf();
    uint32_t n_valid = 0;
    for (uint32_t c = lane; c < C; c += 64) {
      const double kc = ek[c];
      const uint32_t ic = ei[c];
      if (ic == kNoIdx) continue;
      ++n_valid;
      uint32_t rank = 0;
      for (uint32_t c2 = 0; c2 < C; ++c2) rank += lex_less(ek[c2], ei[c2], kc, ic) ? 1u : 0u;
      if (rank < K) {
        out_key[(uint64_t)r * K + rank] = kc;
        out_idx[(uint64_t)r * K + rank] = ic;
        kth = fmax(kth, kc);
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      kth = fmax(kth, __shfl_xor(kth, o, 64));
      n_valid += __shfl_xor(n_valid, o, 64);
    }
    for (uint32_t s = n_valid + lane; s < K; s += 64) {
      out_key[(uint64_t)r * K + s] = __builtin_inf();
      out_idx[(uint64_t)r * K + s] = kNoIdx;
    }
    // every pair that is not in the lists has an approximate cosine below theta[ti]
    bool ok = !overflow && n_valid >= K && kth < bnd.lower_bound(-(double)theta[ti]);
    if (force_exact && force_exact[a]) ok = false;
    if (!ok && lane == 0) fail_rows[atomicAdd(fail_count, 1u)] = r;
    wave_sync();
  }
}

// The same re-rank for 128-wide rows (1 KB): candidate rows are staged through LDS, one LDS-DMA instruction per row
// (the whole wave fetches 1 KB contiguously), a batch of rows at a time, then one lane per row runs the sequential
// dot product from LDS.  The per-lane walk of the generic kernel fetches every candidate row
// in 32-B pieces: 2-4x the bytes across the fabric (profiles/r01e_pmc_traffic.json) and 64 ms at 1 M rows.
constexpr int kRrPitch = 1040;              // LDS pitch of a staged row (bytes): 16 lanes x 4 banks apart
__host__ __device__ inline size_t rr_key_bytes(uint32_t cmax, uint32_t kmax) {
  const size_t a = (size_t)cmax * 4, b = (size_t)kmax * 8;
  return ((a > b ? a : b) + 15) & ~(size_t)15;
}
typedef __attribute__((address_space(3))) void rr_lds_void;
typedef const __attribute__((address_space(1))) void rr_gbl_void;

template <class Policy, class Bound, int kRrRows, int NBUF>
__global__ __launch_bounds__(256) void rerank_join_d128_kernel(
    const double *__restrict__ X, const double *__restrict__ XA, const uint32_t *__restrict__ a_ids, uint32_t row_begin,
    uint32_t row_count, const float *__restrict__ theta, uint32_t n_src,
    const uint32_t *__restrict__ cnt_src, const uint32_t *__restrict__ cand_src, uint32_t cap, uint32_t cmax, uint32_t K,
    const uint8_t *__restrict__ force_exact, Policy pol, Bound bnd, double prune_delta, int prune_rectify, uint32_t kmax,
    double *__restrict__ out_key,
    uint32_t *__restrict__ out_idx, uint32_t *__restrict__ fail_rows, uint32_t *__restrict__ fail_count,
    int *__restrict__ err_flag) {
  constexpr uint32_t d = 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // per wave: the row (1 KB), the staged candidate rows, the keys of up to kmax candidates that are scored exactly
  // (the same bytes hold the approximate cosines of all cmax gathered candidates until the pruning is done), the ids
  const size_t key_bytes = rr_key_bytes(cmax, kmax);
  const size_t per_wave = 1024 + (size_t)NBUF * kRrRows * kRrPitch + key_bytes + (size_t)cmax * 4;
  unsigned char *base = smem + (size_t)wv * per_wave;
  double *xrow = reinterpret_cast<double *>(base);
  unsigned char *tiles = base + 1024;
  double *ek = reinterpret_cast<double *>(tiles + NBUF * kRrRows * kRrPitch);
  uint32_t *ei = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(ek) + key_bytes);

  for (uint32_t r = blockIdx.x * 4 + wv; r < row_count; r += gridDim.x * 4) {
    const uint32_t a = a_ids ? a_ids[r] : row_begin + r;
    const uint32_t ti = a_ids ? r : a;
    for (uint32_t k = lane; k < d; k += 64) xrow[k] = XA[(uint64_t)a * d + k];
    uint32_t C = 0;
    bool overflow = false;
    for (uint32_t sidx = 0; sidx < n_src; ++sidx) {
      const uint32_t total = cnt_src[(size_t)sidx * row_count + r];
      uint32_t take = min(total, cap);
      overflow = overflow || total > cap;
      if (C + take > cmax) {
        overflow = true;
        take = cmax - C;
      }
      const uint2 *src = reinterpret_cast<const uint2 *>(cand_src) + ((size_t)sidx * row_count + r) * cap;
      float *es = reinterpret_cast<float *>(ek);  // approximate cosines, until the exact keys overwrite them
      for (uint32_t c = lane; c < take; c += 64) {
        const uint2 e = src[c];
        ei[C + c] = e.x;
        es[C + c] = __uint_as_float(e.y);
      }
      C += take;
    }
    wave_sync();
    if (prune_delta >= 0.0) C = prune_candidates(ei, reinterpret_cast<float *>(ek), C, K, prune_delta, prune_rectify, lane);
    if (C > kmax) {  // more survivors than key slots: the row goes on to the next stage
      overflow = true;
      C = kmax;
    }
    const uint32_t n_batches = (C + kRrRows - 1) / kRrRows;
    auto issue = [&](uint32_t b) {
      unsigned char *dst = tiles + (size_t)(b % NBUF) * kRrRows * kRrPitch;
#pragma unroll
      for (int t = 0; t < kRrRows; ++t) {
        const uint32_t c = b * kRrRows + t;
        if (c < C) {  // wave-uniform
          const uint32_t j = ei[c];
          __builtin_amdgcn_global_load_lds((rr_gbl_void *)((const char *)(X + (uint64_t)j * d) + lane * 16),
                                           (rr_lds_void *)(dst + t * kRrPitch), 16, 0, 0);
        }
      }
    };
    if (n_batches && NBUF == 2) issue(0);
    for (uint32_t b = 0; b < n_batches; ++b) {
      if (NBUF == 2) {
        const bool more = b + 1 < n_batches;
        if (more) issue(b + 1);
        // batch b has landed once at most the rows of batch b + 1 are still in flight
        if (more && (b + 2) * kRrRows <= C) {
          if (kRrRows == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      } else {
        issue(b);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      const uint32_t c = b * kRrRows + lane;
      if (lane < kRrRows && c < C) {
        const double *bj = reinterpret_cast<const double *>(tiles + (size_t)(b % NBUF) * kRrRows * kRrPitch + (size_t)lane * kRrPitch);
        double acc = 0.0;
#pragma unroll 8
        for (uint32_t k = 0; k < d; ++k) acc = __dadd_rn(acc, __dmul_rn(xrow[k], bj[k]));
        const uint32_t j = ei[c];
        double key = __builtin_inf(), kx;
        uint32_t id = kNoIdx;
        if (pol.key(acc, a, j, kx, err_flag)) {
          key = kx;
          id = j;
        }
        ek[c] = key;
        ei[c] = id;
      }
      wave_sync();  // the tile of batch b is free for batch b + 2
    }
    double kth = -__builtin_inf();
    uint32_t n_valid = 0;
    for (uint32_t c = lane; c < C; c += 64) {
      const double kc = ek[c];
      const uint32_t ic = ei[c];
      if (ic == kNoIdx) continue;
      ++n_valid;
      uint32_t rank = 0;
      for (uint32_t c2 = 0; c2 < C; ++c2) rank += lex_less(ek[c2], ei[c2], kc, ic) ? 1u : 0u;
      if (rank < K) {
        out_key[(uint64_t)r * K + rank] = kc;
        out_idx[(uint64_t)r * K + rank] = ic;
        kth = fmax(kth, kc);
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      kth = fmax(kth, __shfl_xor(kth, o, 64));
      n_valid += __shfl_xor(n_valid, o, 64);
    }
    for (uint32_t s = n_valid + lane; s < K; s += 64) {
      out_key[(uint64_t)r * K + s] = __builtin_inf();
      out_idx[(uint64_t)r * K + s] = kNoIdx;
    }
    bool ok = !overflow && n_valid >= K && kth < bnd.lower_bound(-(double)theta[ti]);
    if (force_exact && force_exact[a]) ok = false;
    if (!ok && lane == 0) fail_rows[atomicAdd(fail_count, 1u)] = r;
    wave_sync();
  }
}

int join_f32(as_ctx *ctx, const float *P, uint32_t n, uint32_t d, uint32_t I_begin, uint32_t I_count,
             const float *theta, uint32_t cap, uint32_t *cnt, uint32_t *cand, uint32_t *unit_counter,
             const char *prof_name);
static __global__ void count_weird_items_kernel(const double *__restrict__ norms, const double *__restrict__ lams, uint32_t n, int *__restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double a = norms[i], b = lams[i];
  if (!(a == a) || !(b == b) || a >= __builtin_inf() || b >= __builtin_inf() || b <= -__builtin_inf()) atomicAdd(out, 1);
}
int stream_scan_search(as_ctx *ctx, const double *queries, const uint32_t *q_ids, uint32_t nq, const double *items, uint32_t n, uint32_t K,
                       const SearchScorePolicy &pol, double *out_key, uint32_t *out_idx, int *err_flag, std::vector<uint32_t> *redo,
                       const StreamFilter *filter = nullptr);
int stream_scan_graph(as_ctx *ctx, const double *rows, const uint32_t *row_ids, uint32_t n_rows, uint32_t n, uint32_t K,
                      const GraphDistPolicy &pol, double *out_key, uint32_t *out_idx, int *err_flag, std::vector<uint32_t> *redo,
                      const StreamFilter *filter = nullptr);
bool stream_quick_wide_applies(uint64_t nA, uint64_t nB, uint64_t d, uint32_t K);
int stream_quick_wide_search(as_ctx *ctx, const double *queries, uint32_t nq, const double *items, uint32_t n, uint32_t d, uint32_t K,
                             const SearchScorePolicy &pol, double *out_key, uint32_t *out_idx, int *err_flag, std::vector<uint32_t> *redo, const uint32_t *q_ids = nullptr);
int stream_quick_wide_graph(as_ctx *ctx, const double *rows, const uint32_t *row_ids, uint32_t n_rows, uint32_t n, uint32_t d, uint32_t K,
                            const GraphDistPolicy &pol, double *out_key, uint32_t *out_idx, int *err_flag, std::vector<uint32_t> *redo);
constexpr uint32_t kStreamRowsMax = 128;  // leftover rows up to this many take the streaming exact scan, 32 per pass

int join_bf16(as_ctx *ctx, const void *P, uint32_t n, uint32_t d, uint32_t I_begin, uint32_t I_count, const float *theta,
              uint32_t cap, uint32_t *cnt, uint32_t *cand, uint32_t *unit_counter, const char *prof_name);
int join_thresholds_bf16(as_ctx *ctx, const double *rows, const double *norms, uint32_t d, uint32_t row_begin,
                         uint32_t row_count, const uint32_t *sample_ids, uint32_t ns, uint32_t m, float *theta_out,
                         uint32_t m2, float *theta2_out);
int join_bf16_rect(as_ctx *ctx, const void *P, uint32_t n, uint32_t d, const void *PA, uint32_t nA, const uint32_t *a_ids,
                   const float *theta_a, uint32_t cap, uint32_t *cnt, uint32_t *cand, uint32_t *unit_counter,
                   const char *prof_name);
int join_bf16_search(as_ctx *ctx, const void *P, uint32_t n, uint32_t d, const void *PQ, uint32_t nq, const float *guess,
                     const float *q_lam, const float *x_lam, float alpha, float beta, uint32_t cap, uint32_t *cnt,
                     uint32_t *cand, uint32_t *unit_counter, const char *prof_name);
int search_thresholds_bf16(as_ctx *ctx, const void *PQ, uint32_t nq, uint32_t d, const double *items, const double *norms,
                           const uint32_t *sample_ids, uint32_t ns, uint32_t m, const float *q_lam, const float *s_lam,
                           float alpha, float beta, float *score_out, uint32_t m2, float *score2_out);
uint64_t packed_vecs_bf16(uint32_t tiles, uint32_t d);
int pack_bf16(as_ctx *ctx, const double *R, const double *norms, const uint32_t *row_ids, uint64_t row0, uint64_t n,
              uint32_t d, void *P, uint32_t tiles);
int join_f16(as_ctx *ctx, const void *P, uint32_t n, uint32_t d, uint32_t I_begin, uint32_t I_count, const float *theta,
             uint32_t cap, uint32_t *cnt, uint32_t *cand, uint32_t *unit_counter, const char *prof_name, const uint32_t *perm = nullptr,
             const unsigned char *mask = nullptr);
uint64_t prefilter_mask_bytes(uint32_t n, uint32_t I_count);
int prefilter_i8(as_ctx *ctx, const void *Ph, const double *rows, const double *norms, uint32_t n, uint32_t d, uint32_t I_begin,
                 uint32_t I_count, const float *theta, unsigned char *mask, const uint32_t *perm);
int cluster_order(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, uint32_t *perm, float *piv_out = nullptr,
                  uint32_t n_pivots = 0, uint32_t *group_start = nullptr, uint32_t *np_out = nullptr, uint32_t max_pivots = 0,
                  DevBuf<uint32_t> *far_pos = nullptr, uint32_t *far_count = nullptr, DevBuf<uint32_t> *near_out = nullptr,
                  const ShardedRows *own = nullptr);
int order_by_value(as_ctx *ctx, const double *values, uint32_t n, uint32_t *perm);  // row_order.hip
int order_by_pivots(as_ctx *ctx, const double *rows, uint32_t n, uint32_t d, const float *piv, uint32_t *perm, uint32_t np = 256,
                    uint32_t *group_start = nullptr, const float *piv_sub = nullptr, int *keys_out = nullptr, const uint32_t *assign_pre = nullptr,
                    const ShardedRows *own = nullptr);
int scatter_by_perm(as_ctx *ctx, const float *in, const uint32_t *perm, uint32_t n, float *out);
int gather_f32_by_ids(as_ctx *ctx, const float *in, const uint32_t *ids, uint32_t n, float *out);
uint64_t packed_vecs_f16(uint32_t tiles, uint32_t d);
uint32_t packed_tiles_f16(uint64_t n);
int pack_f16(as_ctx *ctx, const double *R, const double *norms, const uint32_t *row_ids, uint64_t row0, uint64_t n, uint32_t d,
             void *P, uint32_t tiles);
int pack_f16_all(as_ctx *ctx, const double *R, const double *norms, uint64_t n, uint32_t d, void *P, uint32_t tiles,
                 const uint32_t *perm = nullptr, float *err_rows = nullptr);
int join_f16_wide(as_ctx *ctx, const void *P, uint32_t n, uint32_t d, uint32_t I_begin, uint32_t I_count, const float *theta,
                  uint32_t cap, uint32_t *cnt, uint32_t *cand, const char *prof_name, const uint32_t *perm = nullptr,
                  const unsigned char *mask = nullptr);
int join_f16_wide_rect(as_ctx *ctx, const void *P, uint32_t n, uint32_t d, const void *PA, uint32_t nA, const uint32_t *a_ids,
                       const float *theta_a, uint32_t cap, uint32_t *cnt, uint32_t *cand, const char *prof_name,
                       const uint32_t *perm = nullptr, const unsigned char *mask = nullptr);
bool rotated_wide_supported(uint32_t d);
int rotated_rows_wide(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, double *Z, double *znorm,
                      double *tail_out, const ShardedRows *sr = nullptr, float *Q_out = nullptr, RotatedPacked *packed = nullptr);
uint32_t rotated_wide_pitch(uint32_t d);
int rotated_project_wide(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, const float *Q, double *Z, double *znorm);
int pack_rotated_from_z(as_ctx *ctx, const double *Z, const uint32_t *perm, uint32_t n, uint32_t tiles, void *P8, double *err_out);
int prefilter_i8_rotated(as_ctx *ctx, const void *P8, double err, uint32_t n, uint32_t I_begin, uint32_t I_count, const float *theta,
                         unsigned char *mask, const uint32_t *tile_mask = nullptr, uint32_t tm_words = 0);

int join_thresholds_f16_wide(as_ctx *ctx, const double *rows, const double *norms, uint32_t d, const void *P_all, uint32_t row_begin,
                             uint32_t row_count, const uint32_t *sample_ids, uint32_t ns, uint32_t m, float *theta_out, uint32_t m2,
                             float *theta2_out);
int join_f16_search(as_ctx *ctx, const void *P, uint32_t n, uint32_t d, const void *PQ, uint32_t nq, const float *cos_floor,
                    const float *g_minus_beta, const float *q_lam, const float *x_lam, float alpha, float beta, uint32_t cap,
                    uint32_t *cnt, uint32_t *cand, uint32_t *unit_counter, const char *prof_name, const uint32_t *perm = nullptr,
                    const uint32_t *perm_a = nullptr, const unsigned char *mask = nullptr, const uint32_t *self_ids = nullptr);
int rotated_image_pack(as_ctx *ctx, const double *rows, const double *norms, const uint32_t *perm, uint32_t n, uint32_t d,
                       const float *Qd, void *P8, uint32_t tiles, double *err_out);
int rotated_basis(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, float *Qd, double *tail_out);
uint64_t rotated_image_bytes(uint32_t tiles);
int prefilter_i8_rect(as_ctx *ctx, const void *P8, uint32_t n, const void *PA, uint32_t a_supers, const float *cos_floor, double delta8,
                      unsigned char *mask, const char *prof_name);
int search_thresholds_f16_wide(as_ctx *ctx, const void *PQ, uint32_t nq, uint32_t d, const double *items, const double *norms,
                               const uint32_t *sample_ids, uint32_t ns, uint32_t m, const float *q_lam, const float *s_lam, float alpha,
                               float beta, float *score_out, uint32_t m2, float *score2_out);
int join_f16_wide_search(as_ctx *ctx, const void *P, uint32_t n, uint32_t d, const void *PQ, uint32_t nq, const float *cos_floor,
                         const float *g_minus_beta, const float *q_lam, const float *x_lam, float alpha, float beta, uint32_t cap,
                         uint32_t *cnt, uint32_t *cand, const char *prof_name, const uint32_t *perm = nullptr, const uint32_t *q_perm = nullptr,
                         const unsigned char *mask = nullptr);
int join_thresholds_f16_wide_grouped(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, const void *P_all,
                                     const uint32_t *perm, const uint32_t *group_start, uint32_t np, const float *piv, uint32_t piv_pitch,
                                     uint32_t piv_dim, uint32_t m, float *theta_out, uint32_t m2, float *theta2_out, const uint32_t *near_pre = nullptr);
int join_thresholds_f16_grouped(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, const void *P_all,
                                const uint32_t *perm, const uint32_t *group_start, uint32_t np, const float *piv, uint32_t piv_pitch,
                                uint32_t piv_dim, uint32_t m, float *theta_out, uint32_t m2, float *theta2_out, uint32_t pos_begin = 0,
                                uint32_t pos_count = 0, const void *row_image = nullptr, const uint32_t *near_pre = nullptr);
int join_thresholds_f16(as_ctx *ctx, const double *rows, const double *norms, uint32_t d, const void *P_all, uint32_t row_begin,
                        uint32_t row_count, const uint32_t *sample_ids, uint32_t ns, uint32_t m, float *theta_out, uint32_t m2,
                        float *theta2_out, const void *row_image = nullptr, uint64_t n_image = 0);
int join_thresholds_f16_wide_band(as_ctx *ctx, uint32_t n, uint32_t d, const void *P_all, uint32_t band, uint32_t m, float *theta_out);
int join_thresholds_f16_band(as_ctx *ctx, uint32_t n, uint32_t d, const void *P_all, uint32_t band, uint32_t m, float *theta_out,
                             uint32_t pos_begin = 0, uint32_t pos_count = 0);
int pack_f16_from_image(as_ctx *ctx, const void *row_image, uint64_t n_src, const uint32_t *row_ids, uint64_t n, uint32_t d, void *P, uint32_t tiles);
int pack_f16_from_rows(as_ctx *ctx, const void *Hrow, uint64_t n_src, const uint32_t *row_ids, uint64_t n, void *P, uint32_t tiles);
int sharded_max_plane_error(as_ctx *ctx, const ShardedRows &sr, double own_err);
int pack_f16_rows_sharded(as_ctx *ctx, const ShardedRows &sr, const double *norms_all, uint64_t n, uint32_t d, void *Prow, uint32_t tiles);

// which pipe contracts the join, and the rounding bound its candidate lists are certified against (the wider of
// the two where the bf16 pipe may have been used: every rank of a sharded build derives the same value)
// (rows wider than 128 take the panelled form of the f16 kernel: mfma_join_f16_wide.hip)
static bool join_on_f16(const as_ctx *ctx, uint32_t d) { return ctx->opt_join_pipe == 16 && d <= 2048; }
static bool join_on_bf16(const as_ctx *ctx, uint32_t d) { return ctx->opt_join_pipe == 48 && d <= 128; }
// the one-sided second pass over what the join leaves open runs on the split-bf16 planes after either 16-bit join
static bool second_pass_bf16(const as_ctx *ctx, uint32_t d) { return (ctx->opt_join_pipe == 16 || ctx->opt_join_pipe == 48) && d <= 128; }
static double rect_delta(uint32_t d) { return std::max(cos_delta_bf16x3(d), cos_delta_f32(d)); }
// f16 plane: with the measured plane error E of the rows (pack_f16_all) 2 E + E^2 replaces the worst-case plane term
static double join_delta(const as_ctx *ctx, uint32_t d) {
  if (join_on_f16(ctx, d)) {
    const double worst = cos_delta_f16(d);
    const double e = ctx->f16_err;
    const double measured = e >= 0.0 ? 2.0 * e + e * e + cos_delta_f16_accum(d) : worst;
    return std::max(std::min(worst, measured), rect_delta(d));
  }
  return join_on_bf16(ctx, d) ? rect_delta(d) : cos_delta_f32(d);
}

// shard starts are multiples of 512: the row super tile of the f16 join (256 for the other pipes)
static bool join_supported(const as_ctx *ctx, uint32_t n, uint32_t d, uint64_t row_begin) {
  return ctx->opt_knn_join && ctx->opt_cand_precision != 64 && d >= 1 && rerank_fits(d) &&
         n >= (uint32_t)std::max<int64_t>(4 * kTileRows, ctx->opt_join_min_rows) &&
         row_begin % (join_on_f16(ctx, d) ? 512 : 256) == 0;
}


}  // namespace asx
