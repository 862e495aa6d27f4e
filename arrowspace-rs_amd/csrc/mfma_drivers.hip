// mfma_drivers.hip -- staged drivers of the MFMA neighbour search (K2) and search scan (K9):
// candidate pass (f32 MFMA, then f64 MFMA for what f32 cannot certify) -> exact re-rank ->
// certification -> exact scan for the remainder.  See mfma_common.h / DESIGN.md section 4.
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
__global__ void flag_tiny_norm_rows_kernel(const double *__restrict__ norms, uint32_t n, double min_pos_norm,
                                           uint8_t *__restrict__ flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const double x = norms[i];
    // non-finite norms as well: an infinite norm meets a NaN norm as cos := 0 under the exact rule (the product of
    // the norms is NaN, which is not > 1e-12), which the unit-row contraction cannot reproduce
    flag[i] = (x != x) || !(x < __builtin_inf()) || (x > 0.0 && !(x * min_pos_norm > 1e-12)) ? 1 : 0;
  }
}

__global__ void min_norm_kernel(const double *__restrict__ norms, uint32_t n, unsigned long long *out) {
  // norms are >= 0 (or NaN): the bit pattern of a non-negative double orders like the value
  double v = __builtin_inf();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double x = norms[i];
    if (x > 0.0 && x < v) v = x;  // smallest POSITIVE norm
  }
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMin(out, (unsigned long long)__double_as_longlong(v));
}

__global__ void scatter_rows_kernel(const uint32_t *__restrict__ rows, uint32_t count, uint32_t K,
                                    const double *__restrict__ src_key, const uint32_t *__restrict__ src_idx,
                                    double *__restrict__ dst_key, uint32_t *__restrict__ dst_idx) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count * K) return;
  const uint32_t r = t / K, s = t - r * K;
  dst_key[(uint64_t)rows[r] * K + s] = src_key[t];
  dst_idx[(uint64_t)rows[r] * K + s] = src_idx[t];
}


__global__ void set_u64_kernel(unsigned long long *p, unsigned long long v) { *p = v; }

__global__ void gather_u32_kernel(const uint32_t *__restrict__ table, const uint32_t *__restrict__ pos,
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
__global__ void theta_band_merge_kernel(float *__restrict__ t, const float *__restrict__ band, uint32_t n, float by, float slack) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n && t[p] < __builtin_inff()) {
    const float b = band[p] - by;
    if (b > t[p] || (b == b && t[p] - b <= slack)) t[p] = b;
  }
}
__global__ void position_keys_kernel(const uint32_t *__restrict__ ids, const uint32_t *__restrict__ inv, uint32_t count, double *__restrict__ keys) {
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
  AS_HIP(ctx, copy_d2h(ctx, host, dev, sizeof(uint32_t)));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
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
__global__ void sample_ids_kernel(uint32_t ns, uint32_t rate, uint32_t half_bits, uint32_t *__restrict__ ids) {
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
__global__ void join_theta_kernel(const double *__restrict__ ckey, const uint32_t *__restrict__ cidx, uint32_t n,
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
__global__ void second_theta_kernel(const float *__restrict__ theta, const float *__restrict__ theta2,
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
__global__ void second_theta_fresh_kernel(const float *__restrict__ t1, const float *__restrict__ t2, const float *__restrict__ theta_first,
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
__global__ void theta_checksum_kernel(const float *__restrict__ t, uint32_t n, unsigned long long *__restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(out, (unsigned long long)__float_as_uint(t[i]) * (unsigned long long)(i % 1021 + 1));
}
__global__ void theta_check_ids_kernel(const uint32_t *__restrict__ perm, uint32_t stride, uint32_t count, uint32_t *__restrict__ ids) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) ids[i] = perm[(size_t)i * stride];
}
// the sample the grouped thresholds draw from, whole: image position 32 j + hash(j) of every 32 (mfma_theta_f16.hip)
__global__ void theta_check_sample_kernel(const uint32_t *__restrict__ perm, uint32_t n, uint32_t ns, uint32_t *__restrict__ ids) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ns) return;
  uint32_t x = j * 0x9E3779B9u;
  x ^= x >> 15;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  const uint32_t p = j * 32 + (x & 31u);
  ids[j] = perm[p < n ? p : n - 1];
}
__global__ void theta_check_count_kernel(const float *__restrict__ theta_pos, uint32_t stride, const float *__restrict__ t_full, uint32_t count,
                                         float slack, uint32_t *__restrict__ bad) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count && theta_pos[(size_t)i * stride] < t_full[i] - slack) atomicAdd(bad, 1u);
}

__global__ void invert_perm_kernel(const uint32_t *__restrict__ perm, uint32_t n, uint32_t *__restrict__ inv) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) inv[perm[i]] = i;
}

__global__ void pad_theta_kernel(const float *__restrict__ theta, uint32_t n, uint32_t n_pad, float *__restrict__ out) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n_pad) out[r] = r < n ? theta[r] : __builtin_inff();
}

// candidates per row: total, maximum, rows over capacity (diagnostics for the threshold choice)
__global__ void join_count_stats_kernel(const uint32_t *__restrict__ cnt, uint32_t n, uint32_t cap,
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

}  // namespace asx
#include "rerank_split.h"
#include "rerank_pairs.h"
namespace asx {

// records (rerank_prune_kernel) -> every pair scored once -> the K best of every row + the rows that could not be certified
// (rerank_pairs.h).  records are by place p: row = order ? order[p] : p, its id a_ids ? a_ids[row] : row_begin + row;
// row2rec: row id -> record of the whole-matrix image order (the inverse of `order`), or nullptr (built here).
__global__ void gather_f64_kernel(const double *__restrict__ src, const uint32_t *__restrict__ idx, uint32_t n, double *__restrict__ dst) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}

struct RerankPairsState {  // what outlives rerank_pairs_launch until the caller's next read-back
  DevBuf<uint32_t> map, s_ids, meta, rowpos, seg_batches;
  DevBuf<uint64_t> seg_off;
  DevBuf<rp_u4> s_desc;
  DevBuf<unsigned long long> counters;
  uint32_t cap_batches = 0;
};
template <int SLOTS>
static int rerank_pairs_launch(as_ctx *ctx, RerankPairsState &st, uint64_t cap_batches, const double *rows, uint32_t d, uint64_t n_total,
                               uint32_t row_begin, uint32_t row_count, uint32_t *records, uint32_t K, int rectify, GraphBound bnd, double *out_dist,
                               uint32_t *out_idx, uint32_t *fail_rows, uint32_t *fail_count, const uint32_t *a_ids, const uint32_t *order,
                               const uint32_t *row2rec) {
  constexpr uint32_t S = 64u * SLOTS;
  const uint32_t n_seg = (row_count + kRpSeg - 1) / kRpSeg;
  const int share = ctx->opt_rerank_share && row_count < kRpShareMax;
  if (share && !row2rec) {
    if (!st.map.p) {
      AS_TRY(st.map.alloc(ctx, n_total));
      AS_HIP(ctx, hipMemsetAsync(st.map.p, 0xFF, sizeof(uint32_t) * n_total, ctx->stream));
      hipLaunchKernelGGL(rerank_row2rec_kernel, dim3(cdiv(row_count, 256)), dim3(256), 0, ctx->stream, row_begin, row_count, a_ids, order, st.map.p);
    }
    row2rec = st.map.p;
  }
  if (cap_batches >= 0xFFFFFFF0ull / 16) return fail(ctx, AS_ERR_UNSUPPORTED, "re-rank: too many pairs for one call");
  st.cap_batches = (uint32_t)cap_batches;
  AS_TRY(st.s_ids.alloc(ctx, (size_t)cap_batches * 16));
  AS_TRY(st.s_desc.alloc(ctx, (size_t)cap_batches * 16));
  if (!st.counters.p) AS_TRY(st.counters.alloc(ctx, kRpCounters));
  AS_HIP(ctx, hipMemsetAsync(st.counters.p, 0, sizeof(unsigned long long) * kRpCounters, ctx->stream));
  AS_TRY(st.meta.alloc(ctx, (size_t)row_count * S));
  AS_TRY(st.rowpos.alloc(ctx, row_count));
  AS_TRY(st.seg_batches.alloc(ctx, n_seg));
  AS_TRY(st.seg_off.alloc(ctx, (size_t)n_seg + 1));
  {
    ProfScope ps(ctx, "knn_rerank_link");
    hipLaunchKernelGGL((rerank_link_kernel<SLOTS>), dim3(std::min<uint32_t>(n_seg, 256u * 32u)), dim3(256), 0, ctx->stream, row_begin, row_count,
                       records, a_ids, order, row2rec, share, st.meta.p, st.rowpos.p, st.seg_batches.p, st.counters.p);
    AS_HIP(ctx, hipGetLastError());
    AS_TRY(exclusive_scan_u32_u64(ctx, st.seg_batches.p, n_seg, st.seg_off.p));
  }
  {
    ProfScope ps(ctx, "knn_rerank_fill");
    hipLaunchKernelGGL((rerank_fill_kernel<SLOTS>), dim3(std::min<uint32_t>(n_seg, 256u * 32u)), dim3(256), 0, ctx->stream, row_begin, row_count,
                       (const uint32_t *)records, a_ids, order, (const uint32_t *)st.meta.p, (const uint32_t *)st.rowpos.p,
                       (const uint32_t *)st.seg_batches.p, (const uint64_t *)st.seg_off.p, st.s_ids.p, reinterpret_cast<uint4 *>(st.s_desc.p),
                       st.cap_batches, st.counters.p);
    AS_HIP(ctx, hipGetLastError());
  }
  {
    ProfScope ps(ctx, "knn_rerank_score");
    const uint32_t rs_grid = 256u * 7u * 8u;  // workgroups (one wave each)
    // (the stream's length is known on the device only: a grid sized for the chip; waves past the stream's end leave at once)
    const uint32_t grid = std::min<uint64_t>(rs_grid, std::max<uint64_t>(8, cap_batches / 8 * 8));
    const uint32_t dbg = (ctx->opt_debug & 64) ? (uint32_t)(ctx->opt_debug >> 8) : 0u;
    const size_t lds = (size_t)kRsRows * kRsPitch;
#define AS_SCORE(PC)                                                                                                                              \
  case PC:                                                                                                                                        \
    hipLaunchKernelGGL((rerank_pairs_score_kernel<PC, SLOTS>), dim3(grid), dim3(64), lds, ctx->stream, rows, (const uint32_t *)st.s_ids.p,         \
                       (const rp_u4 *)st.s_desc.p, (const unsigned long long *)st.counters.p, rectify, records, dbg);                             \
    break;
    switch (d / 128) {
      AS_SCORE(1) AS_SCORE(2) AS_SCORE(3) AS_SCORE(4) AS_SCORE(5) AS_SCORE(6) AS_SCORE(7) AS_SCORE(8)
      default: return fail(ctx, AS_ERR_UNSUPPORTED, "re-rank: row width not covered");
    }
#undef AS_SCORE
    AS_HIP(ctx, hipGetLastError());
  }
  ProfScope ps(ctx, "knn_rerank_rank");
  hipLaunchKernelGGL((rerank_rank_kernel<SLOTS>), dim3(std::min<uint64_t>(cdiv(row_count, 8), 256 * 32)), dim3(256), 0, ctx->stream, row_begin, row_count,
                     (const uint32_t *)records, K, bnd, out_dist, out_idx, fail_rows, fail_count, a_ids, order, st.counters.p);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

// records (rerank_prune_kernel) -> every pair scored once -> the K best of every row + the rows that could not be certified
// (rerank_pairs.h).  records are by place p: row = order ? order[p] : p, its id a_ids ? a_ids[row] : row_begin + row;
// row2rec: row id -> record of the whole-matrix image order (the inverse of `order`), or nullptr (built here).
// No read-back in between: the slot stream gets room for 3 batches per row (half of that is used on the bench data); a call that
// needs more finds out at the caller's read-back of fail_count (*n_fail) and is repeated here with the size the device counted.
template <int SLOTS>
static int rerank_pairs_run(as_ctx *ctx, const double *rows, uint32_t d, uint64_t n_total, uint32_t row_begin, uint32_t row_count, uint32_t *records,
                            uint32_t K, int rectify, GraphBound bnd, double *out_dist, uint32_t *out_idx, uint32_t *fail_rows, uint32_t *fail_count,
                            const uint32_t *a_ids, const uint32_t *order, const uint32_t *row2rec) {
  RerankPairsState st;
  uint64_t cap = (uint64_t)row_count * std::max<uint64_t>(3, (3 * (uint64_t)K + 15) / 16) + 64;
  if (ctx->opt_rerank_stream_cap > 0) cap = (uint64_t)ctx->opt_rerank_stream_cap;  // (test hook: a first attempt that runs out of room)
  for (int attempt = 0; attempt < 2; ++attempt) {
    AS_TRY(rerank_pairs_launch<SLOTS>(ctx, st, cap, rows, d, n_total, row_begin, row_count, records, K, rectify, bnd, out_dist, out_idx, fail_rows,
                                      fail_count, a_ids, order, row2rec));
    unsigned long long h[kRpCounters] = {};
    AS_HIP(ctx, copy_d2h(ctx, h, st.counters.p, sizeof(h)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stats["knn_rerank_batches"] = (int64_t)h[0];
    ctx->stats["knn_rerank_pairs_scored"] = (int64_t)h[1];
    ctx->stats["knn_rerank_stream_retries"] = attempt;
    ctx->stats["knn_rerank_open_list_ran_over"] = (int64_t)h[4];
    ctx->stats["knn_rerank_open_forced"] = (int64_t)h[5];
    ctx->stats["knn_rerank_open_undelivered"] = (int64_t)h[6];
    ctx->stats["knn_rerank_open_too_few"] = (int64_t)h[7];
    ctx->stats["knn_rerank_open_kth_not_clear"] = (int64_t)h[8];
    if (h[2] == 0) return AS_OK;
    cap = h[0] + 64;
    AS_HIP(ctx, hipMemsetAsync(fail_count, 0, sizeof(uint32_t), ctx->stream));  // (the ranking ran on keys nobody wrote)
  }
  return fail(ctx, AS_ERR_HIP, "re-rank: the slot stream overflowed twice");
}

// Exact re-rank of the joined candidates of rows [row_begin, row_begin + row_count) (one wave per row),
// certification against theta.  The candidates of a row come from n_src lists (one per rank that contracted
// tile pairs touching it): cnt_src[s * row_count + r], cand_src[(s * row_count + r) * cap ..].
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
__global__ void count_weird_items_kernel(const double *__restrict__ norms, const double *__restrict__ lams, uint32_t n, int *__restrict__ out) {
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
                             const SearchScorePolicy &pol, double *out_key, uint32_t *out_idx, int *err_flag, std::vector<uint32_t> *redo);
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
                  uint32_t n_pivots = 0, uint32_t *group_start = nullptr, uint32_t *np_out = nullptr);
int order_by_value(as_ctx *ctx, const double *values, uint32_t n, uint32_t *perm);  // row_order.hip
int order_by_pivots(as_ctx *ctx, const double *rows, uint32_t n, uint32_t d, const float *piv, uint32_t *perm, uint32_t np = 256,
                    uint32_t *group_start = nullptr, const float *piv_sub = nullptr);
int scatter_by_perm(as_ctx *ctx, const float *in, const uint32_t *perm, uint32_t n, float *out);
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
                       const uint32_t *perm = nullptr);
bool rotated_wide_supported(uint32_t d);
int rotated_rows_wide(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, double *Z, double *znorm,
                      double *tail_out, const ShardedRows *sr = nullptr, float *Q_out = nullptr);
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
                                     uint32_t piv_dim, uint32_t m, float *theta_out, uint32_t m2, float *theta2_out);
int join_thresholds_f16_grouped(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, const void *P_all,
                                const uint32_t *perm, const uint32_t *group_start, uint32_t np, const float *piv, uint32_t piv_pitch,
                                uint32_t piv_dim, uint32_t m, float *theta_out, uint32_t m2, float *theta2_out, uint32_t pos_begin = 0,
                                uint32_t pos_count = 0, const void *row_image = nullptr);
int join_thresholds_f16(as_ctx *ctx, const double *rows, const double *norms, uint32_t d, const void *P_all, uint32_t row_begin,
                        uint32_t row_count, const uint32_t *sample_ids, uint32_t ns, uint32_t m, float *theta_out, uint32_t m2,
                        float *theta2_out, const void *row_image = nullptr, uint64_t n_image = 0);
int join_thresholds_f16_wide_band(as_ctx *ctx, uint32_t n, uint32_t d, const void *P_all, uint32_t band, uint32_t m, float *theta_out);
int join_thresholds_f16_band(as_ctx *ctx, uint32_t n, uint32_t d, const void *P_all, uint32_t band, uint32_t m, float *theta_out,
                             uint32_t pos_begin = 0, uint32_t pos_count = 0);
int pack_f16_from_image(as_ctx *ctx, const void *row_image, uint64_t n_src, const uint32_t *row_ids, uint64_t n, uint32_t d, void *P, uint32_t tiles);
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

bool knn_join_applies(const as_ctx *ctx, uint32_t n, uint32_t d) { return join_supported(ctx, n, d, 0); }

// theta_out[r] for r in [row_begin, row_begin + row_count): the m-th best approximate cosine of the row against a
// hashed 1-in-`rate` sample of ALL rows (the existing top-k pass with a short list).
int knn_join_thresholds(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, uint32_t row_begin,
                        uint32_t row_count, float *theta_out, const float *packed, float *theta2_out, bool *has_theta2,
                        const void *packedh) {
  if (has_theta2) *has_theta2 = false;
  if (!join_supported(ctx, n, d, row_begin)) return fail(ctx, AS_ERR_UNSUPPORTED, "threshold join not applicable");
  const int64_t rank_opt = ctx->opt_join_rank > 0 ? ctx->opt_join_rank : 4;
  const int64_t sample_opt = ctx->opt_join_sample > 0 ? ctx->opt_join_sample : 32;
  const uint32_t m = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(rank_opt, 64));
  const uint32_t a_tiles = (row_count + kTileRows - 1) / kTileRows, a_alloc = (a_tiles + 1) & ~1u;
  // sample size: whole pairs of 128-row tiles (the bf16 pass bins the sample by tile parity and has no tail
  // handling), about n / join_sample of them; n >= 512 here
  const uint32_t ns = std::max(256u, (n / (uint32_t)std::max<int64_t>(2, sample_opt)) & ~255u);
  const uint32_t rate = n / ns;
  const uint32_t s_tiles = (ns + kTileRows - 1) / kTileRows, s_alloc = (s_tiles + 1) & ~1u;
  const uint32_t KC0 = std::max(8u, (m + 7) / 8 * 8);
  DevBuf<uint32_t> sid, cidx;
  DevBuf<float> S, PA;
  DevBuf<double> ckey;
  AS_TRY(sid.alloc(ctx, ns));
  hipLaunchKernelGGL(sample_ids_kernel, dim3(cdiv(ns, 256)), dim3(256), 0, ctx->stream, ns, rate, sample_half_bits(ns), sid.p);
  if (ctx->opt_join_bf16) {  // thresholds on a 16-bit pipe (mfma_theta_{f16,bf16}.hip); dims they do not cover take the f32 pass
    // second threshold: rank 2 m + 6 (about three times the candidates), for the second pass over what fails
    const uint32_t m2 = std::min<uint32_t>(48, 2 * m + 6);
    // the f16 image of all rows, when the join runs on it, serves as the row operand: one MFMA per k step
    const bool on_f16 = packedh && join_on_f16(ctx, d) && !(ctx->opt_debug & 32);
    const int rc = on_f16 && d > 128
                       ? join_thresholds_f16_wide(ctx, rows, norms, d, packedh, row_begin, row_count, sid.p, ns, m, theta_out, m2,
                                                  theta2_out)
                   : on_f16
                       ? join_thresholds_f16(ctx, rows, norms, d, packedh, row_begin, row_count, sid.p, ns, m, theta_out, m2,
                                             theta2_out)
                       : join_thresholds_bf16(ctx, rows, norms, d, row_begin, row_count, sid.p, ns, m, theta_out, m2, theta2_out);
    if (rc == AS_OK && theta2_out && has_theta2) *has_theta2 = true;
    if (rc != AS_ERR_UNSUPPORTED) return rc;
  }
  const float *pa = packed ? packed + (size_t)(row_begin / kTileRows) * packed_elems(1, d) : nullptr;
  if (!pa) {
    AS_TRY(PA.alloc(ctx, packed_elems(a_alloc, d)));
    AS_TRY(pack_f32(ctx, rows, norms, nullptr, row_begin, row_count, d, PA.p, a_alloc));
    pa = PA.p;
  }

  AS_TRY(S.alloc(ctx, packed_elems(s_alloc, d)));
  AS_TRY(pack_f32(ctx, rows, norms, sid.p, 0, ns, d, S.p, s_alloc));
  AS_TRY(ckey.alloc(ctx, (size_t)row_count * KC0));
  AS_TRY(cidx.alloc(ctx, (size_t)row_count * KC0));
  CandProblem pb0{};
  pb0.mode = 0;
  pb0.A = rows;
  pb0.a_norms = norms;
  pb0.arows = ARows{nullptr, row_begin};
  pb0.nA = row_count;
  pb0.B = rows;
  pb0.b_norms = norms;
  pb0.nB = ns;
  pb0.d = d;
  pb0.KC = KC0;
  pb0.splits = 1;
  pb0.cand_key = ckey.p;
  pb0.cand_idx = cidx.p;
  AS_TRY(candidates_f32(ctx, pb0, pa, a_tiles, S.p, s_tiles, "knn_join_thresholds"));
  hipLaunchKernelGGL(join_theta_kernel, dim3(cdiv(row_count, 256)), dim3(256), 0, ctx->stream, ckey.p, cidx.p, row_count,
                     KC0, m, row_count, theta_out);
  AS_HIP(ctx, hipGetLastError());
  // (no synchronisation to hand the workspace back: a block returned to the cache is reused by later work on this same stream only)
  return AS_OK;
}

// Contracts the tile pairs owned by the super tiles of [row_begin, row_begin + row_count) and appends to
// cnt / cand (indexed by GLOBAL row, n x cap; zeroed here) every pair that reaches a threshold.
int knn_join_pairs(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, uint32_t row_begin,
                   uint32_t row_count, const float *theta_all, uint32_t cap, uint32_t *cnt, uint32_t *cand,
                   const float *packed, const void *packed16, const void *packedh, const uint32_t *perm, const uint32_t *image_perm,
                   const double *rot_z, ConeDirs *cone_keep) {
  if (!join_supported(ctx, n, d, row_begin)) return fail(ctx, AS_ERR_UNSUPPORTED, "threshold join not applicable");
  const uint32_t b_tiles = (n + kTileRows - 1) / kTileRows, b_alloc = (b_tiles + 1) & ~1u;
  const bool f16 = join_on_f16(ctx, d), bf16 = join_on_bf16(ctx, d);
  const uint32_t super = f16 ? 512 : 256;
  const uint32_t T2 = (n + super - 1) / super;
  const uint32_t I_begin = row_begin / super;
  const uint32_t I_end = (uint32_t)(((uint64_t)row_begin + row_count + super - 1) / super);
  DevBuf<float> P, theta;
  DevBuf<unsigned char> P16;
  DevBuf<uint32_t> unitc;
  if (f16) {
    if (!packedh) {
      const uint32_t tiles = packed_tiles_f16(n);
      AS_TRY(P16.alloc(ctx, packed_vecs_f16(tiles, d) * 16));
      AS_TRY(pack_f16_all(ctx, rows, norms, n, d, P16.p, tiles));
      packedh = P16.p;
    }
  } else if (bf16) {
    if (!packed16) {
      AS_TRY(P16.alloc(ctx, packed_vecs_bf16(b_alloc, d) * 16));
      AS_TRY(pack_bf16(ctx, rows, norms, nullptr, 0, n, d, P16.p, b_alloc));
      packed16 = P16.p;
    }
  } else if (!packed) {
    AS_TRY(P.alloc(ctx, packed_elems(b_alloc, d)));
    AS_TRY(pack_f32(ctx, rows, norms, nullptr, 0, n, d, P.p, b_alloc));
    packed = P.p;
  }
  AS_TRY(theta.alloc(ctx, (size_t)T2 * super));
  hipLaunchKernelGGL(pad_theta_kernel, dim3(cdiv((uint64_t)T2 * super, 256)), dim3(256), 0, ctx->stream, theta_all, n, T2 * super,
                     theta.p);
  AS_TRY(unitc.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(cnt, 0, sizeof(uint32_t) * n, ctx->stream));
  AS_HIP(ctx, hipMemsetAsync(unitc.p, 0, sizeof(uint32_t), ctx->stream));
  if (f16) {
    if (d <= 128) {
      // int8 first pass (mfma_prefilter_i8.hip): the f16 pass then visits only the ~2 % of the blocks that can list something
      DevBuf<unsigned char> mask;
      // (only on an image whose rows are grouped by direction, row_order.hip: in the order of a caller's rows -- the bench
      // generator deals its components round-robin -- every block holds some pair of neighbours and the pass rules out nothing;
      // join_prefilter_min_rows = 0 forces it, for the tests)
      const bool pre = ctx->opt_join_prefilter && n >= (uint64_t)std::max<int64_t>(0, ctx->opt_join_prefilter_min_rows) &&
                       (perm || image_perm || ctx->opt_join_prefilter_min_rows == 0);
      if (pre) {
        AS_TRY(mask.alloc(ctx, prefilter_mask_bytes(n, I_end - I_begin)));
        if (rot_z) {  // the caller has the rotated coordinates of the rows (own join: they also gave the row order)
          DevBuf<unsigned char> P8;
          const uint32_t tiles = packed_tiles_f16(n);
          double err = 0.0;
          AS_TRY(P8.alloc(ctx, (size_t)tiles * 4 * 64 * 16));
          {
            ProfScope ps(ctx, "knn_join_prefilter_pack");
            AS_TRY(pack_rotated_from_z(ctx, rot_z, image_perm ? image_perm : perm, n, tiles, P8.p, &err));
          }
          // in front of it, on an image ordered by direction: which pairs of 128-position tiles can hold a pair at all (a cone
          // around each tile's direction against the exact products of the other tile's rows with it: cone_tiles.hip).  Where the
          // bound keeps most pairs (data without directions to tell apart) the pass sweeps everything as before.
          DevBuf<uint32_t> tile_mask;
          const uint32_t tm_words = cone_tile_words(n);
          bool tiled = ctx->opt_join_cone_tiles != 0 && (perm || image_perm) && n >= 65536 && !(ctx->opt_debug & 2048);
          if (tiled) {
            double kept = 1.0;
            AS_TRY(tile_mask.alloc(ctx, (size_t)(I_end - I_begin) * 4 * tm_words));
            AS_TRY(cone_tile_mask(ctx, rot_z, image_perm ? image_perm : perm, n, theta.p, I_begin * 512u, (I_end - I_begin) * 512u, tile_mask.p, &kept,
                                  cone_keep));
            tiled = kept <= (double)ctx->opt_join_cone_tiles_max_e3 * 1e-3;
          }
          ctx->stats["knn_join_cone_tiles"] = tiled ? 1 : 0;
          AS_TRY(prefilter_i8_rotated(ctx, P8.p, err, n, I_begin, I_end - I_begin, theta.p, mask.p, tiled ? tile_mask.p : nullptr, tm_words));
        } else {
          AS_TRY(prefilter_i8(ctx, packedh, rows, norms, n, d, I_begin, I_end - I_begin, theta.p, mask.p, image_perm ? image_perm : perm));
        }
      }
      ctx->stats["knn_join_prefilter"] = pre ? 1 : 0;
      AS_TRY(join_f16(ctx, packedh, n, d, I_begin, I_end - I_begin, theta.p, cap, cnt, cand, unitc.p, "knn_mfma_join_f16", perm,
                      pre ? mask.p : nullptr));
    } else {
      // rows wider than 128: the same first pass on the rotated coordinates the caller has (rotate_wide.hip), reduced to "which
      // pairs of 256-row super tiles hold anything" for the panelled kernel
      DevBuf<unsigned char> mask, P8;
      // (on any range of row super tiles: a rank of a sharded build sweeps its own positions against every column)
      const bool pre = rot_z != nullptr && I_end > I_begin;
      if (pre) {
        const uint32_t tiles = packed_tiles_f16(n);
        double err = 0.0;
        AS_TRY(P8.alloc(ctx, (size_t)tiles * 4 * 64 * 16));
        AS_TRY(mask.alloc(ctx, prefilter_mask_bytes(n, I_end - I_begin)));
        AS_HIP(ctx, hipMemsetAsync(mask.p, 0, prefilter_mask_bytes(n, I_end - I_begin), ctx->stream));  // (both orientations of a pair are looked up)
        {
          ProfScope ps(ctx, "knn_join_prefilter_pack");
          AS_TRY(pack_rotated_from_z(ctx, rot_z, image_perm ? image_perm : perm, n, tiles, P8.p, &err));
        }
        // (behind the same cone bound on tile pairs as the narrow rows' pass)
        DevBuf<uint32_t> tile_mask;
        const uint32_t tm_words = cone_tile_words(n);
        bool tiled = ctx->opt_join_cone_tiles != 0 && (perm || image_perm) && n >= 65536 && !(ctx->opt_debug & 2048);
        if (tiled) {
          double kept = 1.0;
          AS_TRY(tile_mask.alloc(ctx, (size_t)(I_end - I_begin) * 4 * tm_words));
          AS_TRY(cone_tile_mask(ctx, rot_z, image_perm ? image_perm : perm, n, theta.p, I_begin * 512u, (I_end - I_begin) * 512u, tile_mask.p, &kept));
          tiled = kept <= (double)ctx->opt_join_cone_tiles_max_e3 * 1e-3;
        }
        ctx->stats["knn_join_cone_tiles"] = tiled ? 1 : 0;
        AS_TRY(prefilter_i8_rotated(ctx, P8.p, err, n, I_begin, I_end - I_begin, theta.p, mask.p, tiled ? tile_mask.p : nullptr, tm_words));
      }
      ctx->stats["knn_join_prefilter"] = pre ? 1 : 0;
      AS_TRY(join_f16_wide(ctx, packedh, n, d, I_begin, I_end - I_begin, theta.p, cap, cnt, cand, "knn_mfma_join_f16", perm,
                           pre ? mask.p : nullptr));
    }
    ctx->stats["knn_join_blocks"] = ctx->stats["knn_mfma_join_f16_blocks"];
    ctx->stats["knn_join_pipe"] = 16;
  } else if (bf16) {
    AS_TRY(join_bf16(ctx, packed16, n, d, I_begin, I_end - I_begin, theta.p, cap, cnt, cand, unitc.p, "knn_mfma_join_bf16"));
    ctx->stats["knn_join_blocks"] = ctx->stats["knn_mfma_join_bf16_blocks"];
    ctx->stats["knn_join_pipe"] = 48;
  } else {
    AS_TRY(join_f32(ctx, packed, n, d, I_begin, I_end - I_begin, theta.p, cap, cnt, cand, unitc.p, "knn_mfma_join_f32"));
    ctx->stats["knn_join_blocks"] = (int64_t)(I_end - I_begin) * 64;  // at least the diagonal super tiles
    ctx->stats["knn_join_pipe"] = 32;
  }
  return AS_OK;
}

int join_session_open(as_ctx *ctx, const double *rows, uint64_t n, uint64_t d) {
  join_session_close(ctx);
  ctx->f16_err = -1.0;
  auto &js = ctx->join;
  const uint32_t b_tiles = (uint32_t)((n + kTileRows - 1) / kTileRows), b_alloc = (b_tiles + 1) & ~1u;
  void *p = nullptr;
  AS_TRY(ws_alloc(ctx, sizeof(double) * n, &p, &js.norms_cap));
  js.norms = (double *)p;
  AS_TRY(row_norms(ctx, rows, n, d, js.norms));
  if (join_on_f16(ctx, (uint32_t)d)) {
    const uint32_t tiles = packed_tiles_f16(n);
    AS_TRY(ws_alloc(ctx, packed_vecs_f16(tiles, (uint32_t)d) * 16, &p, &js.packedh_cap));
    js.packedh = p;
    AS_TRY(pack_f16_all(ctx, rows, js.norms, n, (uint32_t)d, js.packedh, tiles));
  } else if (!join_on_bf16(ctx, (uint32_t)d)) {  // the f32 image: only the f32 pipe needs it up front (a later f32 stage packs its own)
    AS_TRY(ws_alloc(ctx, sizeof(float) * packed_elems(b_alloc, (uint32_t)d), &p, &js.packed_cap));
    js.packed = (float *)p;
    AS_TRY(pack_f32(ctx, rows, js.norms, nullptr, 0, n, (uint32_t)d, js.packed, b_alloc));
  } else {
    AS_TRY(ws_alloc(ctx, packed_vecs_bf16(b_alloc, (uint32_t)d) * 16, &p, &js.packed16_cap));
    js.packed16 = p;
    AS_TRY(pack_bf16(ctx, rows, js.norms, nullptr, 0, n, (uint32_t)d, js.packed16, b_alloc));
  }
  js.rows = rows;
  js.n = n;
  js.d = d;
  return AS_OK;
}
bool join_session_matches(const as_ctx *ctx, const double *rows, uint64_t n, uint64_t d) {
  return ctx->join.rows == rows && ctx->join.n == n && ctx->join.d == d && ctx->join.norms &&
         (ctx->join.packed || ctx->join.packed16 || ctx->join.packedh);
}
void join_session_close(as_ctx *ctx) {
  auto &js = ctx->join;
  if (js.norms) ws_free(ctx, js.norms, js.norms_cap);
  if (js.packed) ws_free(ctx, js.packed, js.packed_cap);
  if (js.packed16) ws_free(ctx, js.packed16, js.packed16_cap);
  if (js.packedh) ws_free(ctx, js.packedh, js.packedh_cap);
  if (js.theta2) ws_free(ctx, js.theta2, js.theta2_cap);
  js = as_ctx::JoinSession{};
}

uint32_t knn_join_cap(const as_ctx *ctx, uint32_t K, bool sharded) {
  (void)sharded;
  const int64_t cap_opt = ctx->opt_join_cap > 0 ? ctx->opt_join_cap : 512;
  return (uint32_t)std::max<int64_t>(K + 1, std::min<int64_t>(cap_opt, 1024));
}

// ---- the join on an image in the library's own row order: preparation ------------------------------------------------------
// Everything in front of the contraction of the tile pairs: the rotated coordinates of the rows (rotate_wide.hip), the row
// order found on them (row_order.hip), the f16 image packed in that order, the thresholds (grouped sample + check, or the
// plain sample).  All of it is a function of the rows and the options alone and every sum on the way has a fixed order, so
// every rank of a sharded build that holds all rows computes the SAME order, image and thresholds (csrc/sharded.hip calls
// this on every rank and then contracts only the tile pairs of its own range of image positions).
// pos_begin / pos_count / gather (a sharded build): the thresholds -- the one step of the preparation that is a sweep of MFMAs
// per row rather than a pass over the row -- are computed for that range of image positions (rows, when there is no order)
// only, and gather(buffer of n elements, element bytes) completes the buffer on every rank (in place: the caller's own part
// is where it belongs).  Everything else stays replicated.
constexpr double kRotatedTailMax = 0.95;  // share of the rows' energy outside the 31 directions up to which the rotated first pass is taken
int own_join_prepare(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, uint32_t K, OwnJoin &oj,
                     uint32_t pos_begin, uint32_t pos_count, const ShardedRows *sr) {
  const std::function<int(void *, size_t)> *gather = sr ? &sr->gather : nullptr;
  if (!gather) {
    pos_begin = 0;
    pos_count = n;
  }
  // sr (ShardedRows, common.h): `rows` is null until somebody needs the f64 rows of OTHER ranks (need_rows: the steps that have no
  // form on the rank's own rows -- the plain row order, the plain thresholds, widths the sharded packing does not cover); on the
  // path the benchmark shapes take nothing in here does, and the caller starts their all-gather behind this function
  auto need_rows = [&]() -> int {
    if (!rows) AS_TRY(sr->rows_all(&rows));
    return AS_OK;
  };
  if (!sr) oj.sharded_rows = false;
  const uint32_t b_tiles = (n + kTileRows - 1) / kTileRows;
  const uint32_t b_alloc = (b_tiles + 1) & ~1u;
  ctx->f16_err = -1.0;
  oj.cap = knn_join_cap(ctx, K, false);
  AS_TRY(oj.theta_own.alloc(ctx, n));
  AS_TRY(oj.theta2_own.alloc(ctx, n));
  // rows grouped by direction in the operand image (row_order.hip): the thresholds come out by image position
  const bool reorder_narrow = join_on_f16(ctx, d) && d <= 128 && ctx->opt_join_row_order && n >= 65536 && ctx->opt_join_bf16 &&
                              !(ctx->opt_debug & 32);  // (the f16 threshold pass is what reads the image)
  bool &reorder = oj.reorder;
  reorder = reorder_narrow;
  // rows wider than 128 (up to 1024): their rotated coordinates (31 dominant directions + tail norm, rotate_wide.hip) serve
  // the int8 first pass and, being 32 wide, the row order as well
  oj.wide_pre = false;
  if (join_on_f16(ctx, d) && rotated_wide_supported(d) && ctx->opt_join_prefilter && ctx->opt_join_prefilter != 2 &&
      n >= (uint64_t)std::max<int64_t>(0, ctx->opt_join_prefilter_min_rows) && ctx->opt_join_bf16 && !(ctx->opt_debug & 32)) {
    ProfScope ps(ctx, "knn_join_prefilter_basis");
    double tail = 1.0;
    AS_TRY(oj.rot_z.alloc(ctx, (size_t)n * 32));
    AS_TRY(oj.rot_norm.alloc(ctx, n));
    if (sr && d > 128) AS_TRY(need_rows());
    AS_TRY(rotated_rows_wide(ctx, rows, norms, n, d, oj.rot_z.p, oj.rot_norm.p, &tail, rows ? nullptr : sr));
    ctx->stats["knn_join_prefilter_tail_e6"] = (int64_t)(tail * 1e6);
    // Measured at 1 M x 128 (scripts/time_hard.py: mixtures of rank 16 ... 128 with 4 ... 50 000 clusters, duplicated rows, plain
    // Gaussian rows): the rotated 32-wide pass was never slower than the full-width one (within 2 %) and usually twice as fast
    // (rank 64 / 2 048 clusters, tail 0.28: 56 against 122 ms; rank 96 / 1 024 clusters, tail 0.37: 50 against 114) -- a loose
    // tail term still separates clusters -- and where neither pass can rule anything out (Gaussian rows, tail 0.76: 204 against
    // 248 ms) it simply costs a quarter of the other.  Rows wider than 128 have no full-width pass at all, only the dense
    // panelled join, d / 32 times the cost of the pass (400 k x 768, rank 128, tail 0.69: 87 against 173 ms).  So the cut only
    // keeps degenerate directions out.  (Until round 3 it was 0.10; join_prefilter = 2 still forces the full-width image.)
    oj.wide_pre = ctx->opt_join_prefilter == 3 || tail <= kRotatedTailMax;
  }
  // (with the coordinates at hand the order is found on them: 32 columns instead of dim, a quarter of the pivot pass)
  const bool reorder_wide = oj.wide_pre && ctx->opt_join_row_order && n >= 65536;
  if (reorder || reorder_wide) {
    AS_TRY(oj.perm_own.alloc(ctx, n));
    if (!reorder_wide) {
      AS_TRY(need_rows());
      AS_TRY(cluster_order(ctx, rows, norms, n, d, oj.perm_own.p));
    } else {
      AS_TRY(oj.order_piv.alloc(ctx, (size_t)1024 * 128));
      AS_TRY(oj.order_starts.alloc(ctx, 1025));
      AS_TRY(cluster_order(ctx, oj.rot_z.p, oj.rot_norm.p, n, 32, oj.perm_own.p, oj.order_piv.p,
                           (uint32_t)std::max<int64_t>(32, ctx->opt_join_order_pivots), oj.order_starts.p, &oj.order_np));
    }
    AS_TRY(oj.theta_pos.alloc(ctx, n));
    AS_TRY(oj.theta2_pos.alloc(ctx, n));
    reorder = true;
  }
  ctx->stats["knn_join_row_order"] = reorder ? 1 : 0;
  if (!join_on_f16(ctx, d)) oj.f16_err_pos.release();
  if (join_on_f16(ctx, d)) {  // the f16 image serves the thresholds and the join
    const uint32_t tiles = packed_tiles_f16(n);
    AS_TRY(oj.Phown.alloc(ctx, packed_vecs_f16(tiles, d) * 16));
    if (sr && !rows && d <= 128) {
      // sharded by rows: every rank packs its own rows, the image of all rows in ROW order is all-gathered (256 B per row at 128
      // columns), and the image in the order of the join is that one re-arranged -- the same bytes pack_f16_all stores
      if (reorder) {
        AS_TRY(oj.Prow.alloc(ctx, packed_vecs_f16(tiles, d) * 16));
        AS_TRY(pack_f16_rows_sharded(ctx, *sr, norms, n, d, oj.Prow.p, tiles));
        AS_TRY(pack_f16_from_image(ctx, oj.Prow.p, n, oj.perm_own.p, n, d, oj.Phown.p, tiles));
      } else {
        AS_TRY(pack_f16_rows_sharded(ctx, *sr, norms, n, d, oj.Phown.p, tiles));
      }
      oj.sharded_rows = true;
      oj.f16_err_pos.release();  // (the ranks exchange the largest plane error only: one bound for all pairs)
    } else {
      if (sr) AS_TRY(need_rows());
      AS_TRY(oj.f16_err_pos.alloc(ctx, n));  // the rows' plane errors by image position: the pruning's bound per pair
      AS_TRY(pack_f16_all(ctx, rows, norms, n, d, oj.Phown.p, tiles, reorder ? oj.perm_own.p : nullptr, oj.f16_err_pos.p));
    }
  }
  ctx->stats["sharded_prepare_by_rows"] = oj.sharded_rows ? 1 : 0;
  const void *row_image = oj.sharded_rows ? (reorder ? oj.Prow.p : oj.Phown.p) : nullptr;  // the f16 form of any row by id (sharded by rows)
  // thresholds: with the order by 1024 pivots of the coordinates, against the sampled rows of the 16 groups nearest to a row's
  // own (1.7 % of the products of the plain sample, the same thresholds for 99.5 % of the rows; mfma_theta_f16.hip)
  const bool wide_grouped = d > 128 && join_on_f16(ctx, d) && pos_begin == 0 && (pos_count == 0 || pos_count == n) && !gather;  // (rows wider than 128: one GPU)
  // Whatever pass the sampled thresholds come from (grouped sample, plain sample), in the order of the image they are raised to what
  // the row's own neighbourhood certifies: K other rows of its super tile and the ones beside it reach it, so the row is certified
  // by construction and its list is cut near its K-th best cosine instead of its 128th (mfma_theta_f16.hip,
  // join_thresholds_f16_band); the sampled threshold stays for rows whose neighbours live elsewhere.  Computed first: the check of
  // the grouped thresholds below looks at the thresholds the join will use.
  const bool have_band = reorder && oj.Phown.p && ctx->opt_join_theta_band > 0 && (d <= 128 || wide_grouped) && K + 1 <= 32;
  DevBuf<float> tband;
  if (have_band) {
    AS_TRY(tband.alloc(ctx, n));
    AS_HIP(ctx, hipMemsetAsync(tband.p, 0xFF, sizeof(float) * n, ctx->stream));  // (NaN where not computed: never the larger one)
    if (d > 128) AS_TRY(join_thresholds_f16_wide_band(ctx, n, d, oj.Phown.p, (uint32_t)ctx->opt_join_theta_band, K + 1, tband.p));
    else AS_TRY(join_thresholds_f16_band(ctx, n, d, oj.Phown.p, (uint32_t)ctx->opt_join_theta_band, K + 1, tband.p, pos_begin, pos_count));
  }
  auto merge_band = [&]() -> int {  // (own positions; the callers gather afterwards)
    if (have_band && pos_count)
      hipLaunchKernelGGL(theta_band_merge_kernel, dim3(cdiv(pos_count, 256)), dim3(256), 0, ctx->stream, oj.theta_pos.p + pos_begin,
                         (const float *)tband.p + pos_begin, pos_count, (float)(2.0 * join_delta(ctx, d) + 2e-6),
                         (float)((double)std::max<int64_t>(0, ctx->opt_join_theta_band_slack) * join_delta(ctx, d)));
    AS_HIP(ctx, hipGetLastError());
    return AS_OK;
  };
  if (reorder_wide && (d <= 128 || wide_grouped) && oj.order_np >= 256 && ctx->opt_join_theta_grouped && ctx->opt_join_bf16 &&
      !(ctx->opt_debug & 32) && ctx->opt_join_sample <= 0) {
    const int64_t rank_opt = ctx->opt_join_rank > 0 ? ctx->opt_join_rank : 4;
    const uint32_t m = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(rank_opt, 32)), m2 = std::min<uint32_t>(48, 2 * m + 6);
    if (d > 128)
      AS_TRY(join_thresholds_f16_wide_grouped(ctx, rows, norms, n, d, oj.Phown.p, oj.perm_own.p, oj.order_starts.p, oj.order_np, oj.order_piv.p, 32,
                                              32, m, oj.theta_pos.p, m2, oj.theta2_pos.p));
    else
    AS_TRY(join_thresholds_f16_grouped(ctx, rows, norms, n, d, oj.Phown.p, oj.perm_own.p, oj.order_starts.p, oj.order_np, oj.order_piv.p, 32, 32, m,
                                       oj.theta_pos.p, m2, oj.theta2_pos.p, pos_begin, pos_count, row_image));
    AS_TRY(merge_band());
    if (gather) {
      AS_TRY((*gather)(oj.theta_pos.p, sizeof(float)));
      AS_TRY((*gather)(oj.theta2_pos.p, sizeof(float)));
    }
    if (ctx->opt_debug & 65536) {  // diagnostic: a checksum of the thresholds (are they the same from run to run?)
      DevBuf<unsigned long long> cs;
      AS_TRY(cs.alloc(ctx, 1));
      AS_HIP(ctx, hipMemsetAsync(cs.p, 0, sizeof(unsigned long long), ctx->stream));
      hipLaunchKernelGGL(theta_checksum_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, (const float *)oj.theta_pos.p, n, cs.p);
      unsigned long long h = 0;
      AS_HIP(ctx, copy_d2h(ctx, &h, cs.p, sizeof(h)));
      ctx->stats["knn_join_theta_checksum"] = (int64_t)(h & 0x7FFFFFFFFFFFFFFFull);
    }
    // How good they are depends on the data (do the pivots cover its clusters?), so they are CHECKED: 4096 rows spread over
    // the image get their threshold against the whole sample as well (0.1 ms), and if more than 1.0 % of them (0.6 % until round 4) sit more than
    // 5e-3 below it (rows whose neighbours live in groups the sample of their own group does not reach: every one of them
    // lists thousands of pairs) the grouped thresholds are dropped for the plain pass.  (1 M x 128 bench mixture, 1024
    // pivots: 0.24 %; 700 k rows: 0.27 %; 400 k rows, 768 pivots: 2.9 %; 200 k rows: 4.1 %.)
    const uint32_t n_chk = 4096, stride = n / n_chk;
    DevBuf<uint32_t> chk_ids, chk_bad, chk_sid;
    DevBuf<float> chk_t1, chk_t2;
    DevBuf<unsigned char> chk_img;
    const uint32_t ns_all = std::max(256u, (n / 32u) & ~255u);
    AS_TRY(chk_ids.alloc(ctx, n_chk));
    AS_TRY(chk_bad.alloc(ctx, 1));
    AS_TRY(chk_sid.alloc(ctx, ns_all));
    AS_TRY(chk_t1.alloc(ctx, n_chk));
    AS_TRY(chk_t2.alloc(ctx, n_chk));
    AS_TRY(chk_img.alloc(ctx, packed_vecs_f16(n_chk / kTileRows, d) * 16));
    AS_HIP(ctx, hipMemsetAsync(chk_bad.p, 0, sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(theta_check_ids_kernel, dim3(cdiv(n_chk, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)oj.perm_own.p, stride,
                       n_chk, chk_ids.p);
    hipLaunchKernelGGL(theta_check_sample_kernel, dim3(cdiv(ns_all, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)oj.perm_own.p, n,
                       ns_all, chk_sid.p);  // (the same sampled positions, all of them: what differs is the restriction alone)
    if (row_image) AS_TRY(pack_f16_from_image(ctx, row_image, n, chk_ids.p, n_chk, d, chk_img.p, n_chk / kTileRows));
    else AS_TRY(pack_f16(ctx, rows, norms, chk_ids.p, 0, n_chk, d, chk_img.p, n_chk / kTileRows));
    if (d > 128) AS_TRY(join_thresholds_f16_wide(ctx, rows, norms, d, chk_img.p, 0, n_chk, chk_sid.p, ns_all, m, chk_t1.p, m2, chk_t2.p));
    else AS_TRY(join_thresholds_f16(ctx, rows, norms, d, chk_img.p, 0, n_chk, chk_sid.p, ns_all, m, chk_t1.p, m2, chk_t2.p, row_image, n));
    hipLaunchKernelGGL(theta_check_count_kernel, dim3(cdiv(n_chk, 256)), dim3(256), 0, ctx->stream, (const float *)oj.theta_pos.p, stride,
                       (const float *)chk_t1.p, n_chk, 5e-3f, chk_bad.p);
    uint32_t h_bad = 0;
    AS_TRY(read_u32(ctx, chk_bad.p, &h_bad));
    ctx->stats["knn_join_theta_grouped_bad"] = h_bad;
    // (the plain pass grows with the square of the table and what the rows below it add with the table: the share allowed grows
    // with it -- 4 M rows: 0.66 % of the checked rows below, the plain pass 116 ms of 530)
    const uint32_t bad_per_mille = (uint32_t)std::max<int64_t>(0, ctx->opt_join_theta_grouped_bad_e3) * std::min(4u, std::max(1u, n / 1000000u));
    if (h_bad * 1000u <= bad_per_mille * n_chk) {
      oj.have_theta2 = true;
      oj.grouped_theta = true;
    } else {
      if (sr) AS_TRY(need_rows());  // (every rank takes this branch together: the check reads the same thresholds everywhere)
      if (pos_count)
        AS_TRY(knn_join_thresholds(ctx, rows, norms, n, d, pos_begin, pos_count, oj.theta_pos.p + pos_begin, nullptr, oj.theta2_pos.p + pos_begin,
                                   &oj.have_theta2, oj.Phown.p));
      else oj.have_theta2 = true;  // (a rank without positions: what the ranks with positions find, the 16-bit pass always has them)
      AS_TRY(merge_band());
      if (gather) {
        AS_TRY((*gather)(oj.theta_pos.p, sizeof(float)));
        AS_TRY((*gather)(oj.theta2_pos.p, sizeof(float)));
      }
    }
    ctx->stats["knn_join_theta_grouped"] = oj.grouped_theta ? 1 : 0;
  } else {
    ctx->stats["knn_join_theta_grouped"] = 0;
    float *t1 = reorder ? oj.theta_pos.p : oj.theta_own.p, *t2 = reorder ? oj.theta2_pos.p : oj.theta2_own.p;
    if (sr) AS_TRY(need_rows());
    if (pos_count)
      AS_TRY(knn_join_thresholds(ctx, rows, norms, n, d, pos_begin, pos_count, t1 + pos_begin, nullptr, t2 + pos_begin, &oj.have_theta2, oj.Phown.p));
    AS_TRY(merge_band());  // (have_band implies reorder: t1 is theta_pos)
    if (gather) {
      // (whether the pass produced second thresholds depends on the options and the width alone: every rank agrees; a rank
      // without positions takes the answer of the 16-bit passes, which always have them)
      if (!pos_count) oj.have_theta2 = ctx->opt_join_bf16 != 0;
      AS_TRY((*gather)(t1, sizeof(float)));
      if (oj.have_theta2) AS_TRY((*gather)(t2, sizeof(float)));
    }
  }
  if (have_band) {
    // kept by row for the second pass: a row the sampled threshold left open has K rows above this one by construction
    AS_TRY(oj.theta_band_own.alloc(ctx, n));
    AS_TRY(scatter_by_perm(ctx, tband.p, oj.perm_own.p, n, oj.theta_band_own.p));
  } else {
    oj.theta_band_own.release();
  }
  if (reorder) {  // ... and go back to row order for everything behind the join
    AS_TRY(scatter_by_perm(ctx, oj.theta_pos.p, oj.perm_own.p, n, oj.theta_own.p));
    if (oj.have_theta2) AS_TRY(scatter_by_perm(ctx, oj.theta2_pos.p, oj.perm_own.p, n, oj.theta2_own.p));
  }
  if (join_on_bf16(ctx, d)) {
    if (sr) AS_TRY(need_rows());
    AS_TRY(oj.P16own.alloc(ctx, packed_vecs_bf16(b_alloc, d) * 16));
    AS_TRY(pack_bf16(ctx, rows, norms, nullptr, 0, n, d, oj.P16own.p, b_alloc));
  }
  oj.perm = oj.reorder ? oj.perm_own.p : nullptr;
  oj.theta_image = oj.reorder ? oj.theta_pos.p : oj.theta_own.p;
  return AS_OK;
}

// Contracts the tile pairs owned by the 512-position super tiles of [pos_begin, pos_begin + pos_count) of the prepared image.
// cnt (n) / cand (n x cap slots of two words): lists_by_pos = indexed by image position and holding positions, else indexed
// by row and holding rows (the join kernel translates where it appends).
int own_join_pairs(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, uint32_t pos_begin, uint32_t pos_count,
                   OwnJoin &oj, uint32_t *cnt, uint32_t *cand, bool lists_by_pos) {
  // (the directions were tried above and do not carry the rows: the first pass, if any, takes the full-width image without
  // looking for them again)
  const int64_t saved_mode = ctx->opt_join_prefilter;
  if (oj.rot_z.p && !oj.wide_pre && saved_mode == 1) ctx->opt_join_prefilter = 2;
  const int rc_pairs = knn_join_pairs(ctx, rows, norms, n, d, pos_begin, pos_count, oj.theta_image, oj.cap, cnt, cand, nullptr, oj.P16own.p,
                                      oj.Phown.p, (oj.reorder && !lists_by_pos) ? oj.perm_own.p : nullptr,
                                      oj.reorder ? oj.perm_own.p : nullptr, oj.wide_pre ? oj.rot_z.p : nullptr, &oj.cone);
  ctx->opt_join_prefilter = saved_mode;
  return rc_pairs;
}

// ---- K2 driver ----------------------------------------------------------------------------------
int knn_mfma(as_ctx *ctx, const double *rows, const double *norms, uint64_t n64, uint64_t d64, uint64_t row_begin,
             uint64_t row_count64, uint32_t K, int rectify, uint32_t *out_idx, double *out_dist, const JoinSources *js) {
  const uint32_t n = (uint32_t)n64, d = (uint32_t)d64, row_count = (uint32_t)row_count64;
  const int NC = (int)((d + 31) / 32);
  // one extra slot: a row's own index is the best candidate of its list and is dropped by the re-rank
  const uint32_t KC = cand_capacity(ctx, K + 1);
  const bool wide = NC > kMaxNC;  // dim > 128: only the f32 pass has the k-panel loop
  const bool use_f32 = (ctx->opt_cand_precision != 64 || wide) && fits_f32(KC);
  // deep tables (K beyond what the top-k kernels keep in LDS: KC > 40) still take the 16-bit join -- its lists hold 512
  // candidates whatever K is -- and hand what it leaves open straight to the exact scans
  const bool stages_fit = !(wide && !use_f32) && fits_f64(KC);
  const bool deep_join = !stages_fit && !wide && K <= 64 && ctx->opt_cand_precision != 64 &&
                         (join_on_f16(ctx, d) || join_on_bf16(ctx, d)) &&
                         (js != nullptr || (row_begin == 0 && row_count == n && join_supported(ctx, n, d, 0)));
  if ((!stages_fit && !deep_join) || n < 4 * kTileRows || row_count == 0 || !rerank_fits(d))  // small, deep or very wide problems: exact scan
    return knn_exact(ctx, rows, norms, n, d, (js && js->a_ids) ? js->a_ids : nullptr, (js && js->a_ids) ? 0 : row_begin, row_count, K, rectify,
                     out_idx, out_dist, 0);

  const uint32_t b_tiles = (n + kTileRows - 1) / kTileRows;
  const uint32_t b_alloc = (b_tiles + 1) & ~1u;  // even: the f32 pass reads A tiles in pairs

  // rows whose pairs may hit the 0 < |a||b| <= 1e-12 rule are always re-done exactly
  DevBuf<unsigned long long> minb;
  DevBuf<uint8_t> force;
  AS_TRY(minb.alloc(ctx, 1));
  AS_TRY(force.alloc(ctx, n));
  hipLaunchKernelGGL(set_u64_kernel, dim3(1), dim3(1), 0, ctx->stream, minb.p, 0x7FF0000000000000ull);  // +inf
  hipLaunchKernelGGL(min_norm_kernel, dim3(256), dim3(256), 0, ctx->stream, norms, n, minb.p);
  unsigned long long mb = 0;
  AS_HIP(ctx, copy_d2h(ctx, &mb, minb.p, sizeof(mb)));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double min_norm;
  std::memcpy(&min_norm, &mb, sizeof(double));
  hipLaunchKernelGGL(flag_tiny_norm_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, norms, n, min_norm,
                     force.p);

  DevBuf<uint32_t> fail1, fail_count;
  DevBuf<int> err;
  AS_TRY(fail1.alloc(ctx, row_count));
  AS_TRY(fail_count.alloc(ctx, 1));
  AS_TRY(err.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(fail_count.p, 0, sizeof(uint32_t), ctx->stream));
  AS_HIP(ctx, hipMemsetAsync(err.p, 0, sizeof(int), ctx->stream));
  GraphDistPolicy pol{norms, rectify};

  CandProblem pb{};
  pb.mode = 0;
  pb.A = rows;
  pb.a_norms = norms;
  pb.arows = ARows{nullptr, (uint32_t)row_begin};
  pb.nA = row_count;
  pb.B = rows;
  pb.b_norms = norms;
  pb.nB = n;
  pb.d = d;
  pb.KC = KC;

  // ---- stage 1 over the contiguous row range ----
  // The bulk of the rows runs as whole rounds of 256 workgroups; a remainder that would leave most of
  // the chip idle for a full sweep is launched separately with its column range split (tail quantisation).
  PackedF32 B32, A32;
  PackedF64 B64, A64;
  uint32_t n_fail = 0;
  // dim <= 128: symmetric threshold join (half the contractions).  Either the caller has run the join and
  // exchanged the candidate lists (row-sharded build: js), or the range is the whole matrix and it runs here.
  const bool own_join = !js && (use_f32 || deep_join) && row_begin == 0 && row_count == n &&
                        join_supported(ctx, n, d, 0);
  const bool use_join = js != nullptr || own_join;
  ctx->stats["knn_join"] = use_join ? 1 : 0;
  if (!js) ctx->stats["knn_join_prefilter"] = 0;  // (not a leftover of the call before: only the 16-bit joins set it; with js the caller has just run the join)
  JoinSources local{};  // outlives the block below: the later stages still read js->packed
  // own join: the images, the order, the rotated coordinates and the thresholds (own_join_prepare) stay for the later stages;
  // a build sharded by image positions (csrc/sharded.hip) has prepared the same thing and hands it in with the lists
  OwnJoin oj_local;
  const OwnJoin *oj = js ? js->oj : nullptr;
  bool lists_by_pos = false;            // the lists of the own join are indexed by (and hold) image positions
  DevBuf<uint32_t> cnt_own, cand_own;
  if (use_join) {
    if (own_join) {
      AS_TRY(own_join_prepare(ctx, rows, norms, n, d, K, oj_local));
      if (oj_local.reorder) {  // kept for the caller (as_build_laplacian hands it to the graph: common.h, join_order)
        if (ctx->join_order_cap < n) {
          if (ctx->join_order) (void)hipFree(ctx->join_order);
          ctx->join_order = nullptr;
          ctx->join_order_cap = 0;
          if (hipMalloc((void **)&ctx->join_order, sizeof(uint32_t) * n) == hipSuccess) ctx->join_order_cap = n;
          else (void)hipGetLastError();
        }
        if (ctx->join_order_cap >= n) {
          AS_HIP(ctx, hipMemcpyAsync(ctx->join_order, oj_local.perm_own.p, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, ctx->stream));
          ctx->join_order_n = n;
          ++ctx->join_order_seq;
        }
      }
      const uint32_t cap = oj_local.cap;
      AS_TRY(cnt_own.alloc(ctx, n));
      AS_TRY(cand_own.alloc(ctx, (size_t)n * cap * 2));  // two words per slot: row id, approximate cosine
      // 128 columns: the lists stay in image positions and the pruning kernel of the re-rank translates the ~30 survivors of
      // a row (rerank_split.h); other widths: the join kernel translates every entry where it appends it
      lists_by_pos = oj_local.reorder && d == 128 && !(ctx->opt_debug & 16) && ((uintptr_t)rows & 15) == 0 &&
                     ctx->opt_debug != 5;
      if (lists_by_pos) {
        AS_TRY(oj_local.inv_own.alloc(ctx, n));
        hipLaunchKernelGGL(invert_perm_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, oj_local.perm_own.p, n, oj_local.inv_own.p);
      }
      AS_TRY(own_join_pairs(ctx, rows, norms, n, d, 0, n, oj_local, cnt_own.p, cand_own.p, lists_by_pos));
      if (ctx->opt_debug & 0xC0FF & ~(int64_t)(8 | 16 | 32 | 64)) return AS_OK;  // timing experiment (ablation bits): the join only, outputs undefined
      local = JoinSources{nullptr, oj_local.theta_own.p, 1, cap, cnt_own.p, cand_own.p, oj_local.have_theta2 ? oj_local.theta2_own.p : nullptr,
                          oj_local.P16own.p, oj_local.Phown.p};
      js = &local;
      oj = &oj_local;
    }
    DevBuf<unsigned long long> scored;
    // the rows of the range: [row_begin, row_begin + row_count), or the list js->a_ids (a range of image positions)
    const uint32_t *a_ids = js->a_ids;
    const uint32_t *ent_perm = lists_by_pos ? oj->perm_own.p : nullptr;   // list entries are image positions: position -> row
    const uint32_t *list_inv = lists_by_pos ? oj->inv_own.p : nullptr;    // ... and the list of row a sits at inv[a]
    {
      const uint32_t cmax = std::min<uint32_t>(js->n_src * js->cap, std::max<uint32_t>(512, js->cap));
      GraphBound bnd{join_delta(ctx, d), rectify};
      // candidates whose approximate cosine (stored with them) cannot reach the top K are dropped before any row is
      // fetched: 74 -> ~20 rows of 1 KB per row of the table (debug 16: keep them all)
      const double prune_delta = (ctx->opt_debug & 16) ? -1.0 : join_delta(ctx, d);
      ProfScope ps(ctx, "knn_rerank");
      // rows per batch x buffers.  Measured at 1 M rows (re-rank + fall-back re-rank): per-lane walk 66 ms, 8x1 39, 8x2 46,
      // 16x1 37, 16x2 58: occupancy (LDS per wave) beats double buffering.  debug = 10 + {0: 8x1, 1: 8x2, 2: 16x1, 3: 16x2}.
      // with the pruning ~20 candidates of a row are scored exactly: 8 staged rows per batch and key slots for 128
      // candidates keep three workgroups per CU whatever the list capacity (without it: 16 rows, cmax slots)
      const bool pruning = prune_delta >= 0.0;
      const int rr_rows = pruning ? 8 : 16, rr_nbuf = 1;
      const uint32_t kmax = pruning ? std::min<uint32_t>(cmax, std::max<uint32_t>(128, 2 * K)) : cmax;
      const size_t lds128 = (size_t)4 * (1024 + (size_t)rr_nbuf * rr_rows * kRrPitch + rr_key_bytes(cmax, kmax) + (size_t)cmax * 4);
      DevBuf<uint32_t> records;
      DevBuf<double> norms_pos;
      if (d % 128 == 0 && d <= 1024 && pruning && ((uintptr_t)rows & 15) == 0 && (size_t)4 * cmax * 12 <= 64 * 1024 &&
          ctx->opt_debug != 5) {
        // two kernels (rerank_split.h): lists -> pruned records, then one stream of row batches per wave
        const int slots = d == 128 ? 2 : 4;  // 124 / 252 survivors per record
        AS_TRY(records.alloc(ctx, (size_t)row_count * rs_words(slots) + kRsRows));
        AS_TRY(scored.alloc(ctx, 1));
        AS_HIP(ctx, hipMemsetAsync(scored.p, 0, sizeof(unsigned long long), ctx->stream));
        // the rows in the order of the image (a list of image positions is in that order already): neighbours share candidates
        const uint32_t *score_order = (lists_by_pos && !a_ids && row_begin == 0 && row_count == n) ? oj->perm_own.p : nullptr;
        {
          // the bound per pair from the rows' own plane errors, where the lists hold image positions and the image was packed here
          const float *err_pos = (lists_by_pos && ent_perm && join_on_f16(ctx, d) && oj->f16_err_pos.p && ctx->opt_rerank_pair_bound) ? oj->f16_err_pos.p : nullptr;
          // lists that hold ROW ids (rows wider than 128 columns, images in row order): the same bound from the errors by row
          DevBuf<float> err_row;
          if (!err_pos && !lists_by_pos && oj && own_join && oj->f16_err_pos.p && join_on_f16(ctx, d) && ctx->opt_rerank_pair_bound && !a_ids &&
              js->n_src == 1) {
            if (oj->reorder) {
              AS_TRY(err_row.alloc(ctx, n));
              AS_TRY(scatter_by_perm(ctx, oj->f16_err_pos.p, oj->perm_own.p, n, err_row.p));
              err_pos = err_row.p;
            } else {
              err_pos = oj->f16_err_pos.p;
            }
          }
          ctx->stats["knn_rerank_pair_bound"] = err_pos ? 1 : 0;
          if (ent_perm) {  // the norms by image position: a survivor's norm is asked for together with its row id
            AS_TRY(norms_pos.alloc(ctx, n));
            hipLaunchKernelGGL(gather_f64_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, norms, ent_perm, n, norms_pos.p);
          }
          ProfScope ps_prune(ctx, "knn_rerank_prune");
          hipLaunchKernelGGL(rerank_prune_kernel, dim3(std::min<uint64_t>(cdiv(row_count, 4), 256 * 32)), dim3(256), (size_t)4 * cmax * 12,
                             ctx->stream, (uint32_t)row_begin, row_count, js->theta, norms, js->n_src, js->cnt, js->cand, js->cap, cmax, K,
                             force.p, prune_delta, rectify, records.p, slots, ent_perm, list_inv, scored.p, a_ids, score_order, err_pos,
                             (float)(cos_delta_f16_accum(d) * (1.0 + 1e-6)), (float)(rect_delta(d) * (1.0 + 1e-6)), (const double *)norms_pos.p);
        }
        AS_HIP(ctx, hipGetLastError());
        {
          // every unordered pair scored once, the ranking in a kernel of its own (rerank_pairs.h)
          const uint32_t *row2rec = score_order ? list_inv : nullptr;
          if (d == 128) AS_TRY(rerank_pairs_run<2>(ctx, rows, d, n, (uint32_t)row_begin, row_count, records.p, K, rectify, bnd, out_dist, out_idx, fail1.p, fail_count.p, a_ids, score_order, row2rec));
          else AS_TRY(rerank_pairs_run<4>(ctx, rows, d, n, (uint32_t)row_begin, row_count, records.p, K, rectify, bnd, out_dist, out_idx, fail1.p, fail_count.p, a_ids, score_order, row2rec));
        }
      } else if (lists_by_pos) {
        return fail(ctx, AS_ERR_HIP, "internal: lists by image position need the two-kernel re-rank");
      } else if (d == 128 && lds128 <= 160 * 1024 && ((uintptr_t)rows & 15) == 0 && ctx->opt_debug != 5) {
#define AS_RR(R, NB)                                                                                                       \
  AS_HIP(ctx, hipFuncSetAttribute((const void *)rerank_join_d128_kernel<GraphDistPolicy, GraphBound, R, NB>,               \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds128));                               \
  hipLaunchKernelGGL((rerank_join_d128_kernel<GraphDistPolicy, GraphBound, R, NB>),                                        \
                     dim3(std::min<uint64_t>(cdiv(row_count, 4), 256 * 16)), dim3(256), lds128, ctx->stream, rows, rows,   \
                     a_ids, (uint32_t)row_begin, row_count, a_ids ? js->theta_local : js->theta, js->n_src, js->cnt, js->cand, js->cap, cmax, K, force.p, pol, \
                     bnd, prune_delta, rectify, kmax, out_dist, out_idx, fail1.p, fail_count.p, err.p)
        if (rr_rows == 8) { AS_RR(8, 1); }
        else { AS_RR(16, 1); }
#undef AS_RR
      } else {
        const size_t lds = (size_t)4 * (d + cmax) * sizeof(double) + (size_t)4 * cmax * sizeof(uint32_t);
        AS_HIP(ctx, hipFuncSetAttribute((const void *)rerank_join_kernel<GraphDistPolicy, GraphBound>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((rerank_join_kernel<GraphDistPolicy, GraphBound>), dim3(std::min<uint64_t>(cdiv(row_count, 4), 8192)),
                           dim3(256), lds, ctx->stream, rows, rows, a_ids, d, (uint32_t)row_begin, row_count,
                           a_ids ? js->theta_local : js->theta, js->n_src, js->cnt,
                           js->cand, js->cap, cmax, K, force.p, pol, bnd, prune_delta, rectify, out_dist, out_idx, fail1.p,
                           fail_count.p, err.p);
      }
      AS_HIP(ctx, hipGetLastError());
    }
    {  // statistics: candidates per row over all sources
      DevBuf<unsigned long long> st;
      AS_TRY(st.alloc(ctx, 3));
      AS_HIP(ctx, hipMemsetAsync(st.p, 0, 3 * sizeof(unsigned long long), ctx->stream));
      hipLaunchKernelGGL(join_count_stats_kernel, dim3(256), dim3(256), 0, ctx->stream, js->cnt, js->n_src * row_count,
                           js->cap, st.p);
      unsigned long long h[3] = {0, 0, 0};
      AS_HIP(ctx, copy_d2h(ctx, h, st.p, sizeof(h)));
      ctx->stats["knn_join_cap"] = js->cap;
      ctx->stats["knn_join_candidates"] = (int64_t)h[0];
      ctx->stats["knn_join_max_per_row"] = (int64_t)h[1];
      ctx->stats["knn_join_overflow_rows"] = (int64_t)h[2];
      if (scored.p) {  // candidates that survived the pruning: the rows the scoring kernel fetched
        unsigned long long hs = 0;
        AS_HIP(ctx, copy_d2h(ctx, &hs, scored.p, sizeof(hs)));
        AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->stats["knn_rerank_scored"] = (int64_t)hs;
      }
    }
    AS_TRY(read_u32(ctx, fail_count.p, &n_fail));
  } else {
    const uint32_t item_rows = use_f32 ? 2 * kTileRows : kTileRows;  // A rows per work item
    const uint32_t a_tiles_all = (row_count + kTileRows - 1) / kTileRows;
    const float *PA32 = nullptr;
    const double *PA64 = nullptr;
    if (use_f32) {
      AS_TRY(B32.buf.alloc(ctx, packed_elems(b_alloc, d)));
      AS_TRY(pack_f32(ctx, rows, norms, nullptr, 0, n, d, B32.buf.p, b_alloc));
      B32.p = B32.buf.p;
      PA32 = B32.p + (row_begin / kTileRows) * packed_elems(1, d);
      if (row_begin % item_rows != 0) {
        const uint32_t at = (a_tiles_all + 1) & ~1u;
        AS_TRY(A32.buf.alloc(ctx, packed_elems(at, d)));
        AS_TRY(pack_f32(ctx, rows, norms, nullptr, row_begin, row_count, d, A32.buf.p, at));
        PA32 = A32.buf.p;
      }
    } else {
      AS_TRY(B64.buf.alloc(ctx, packed_elems(b_alloc, d)));
      AS_TRY(pack_f64(ctx, rows, norms, nullptr, 0, n, d, B64.buf.p, b_alloc));
      B64.p = B64.buf.p;
      PA64 = B64.p + (row_begin / kTileRows) * packed_elems(1, d);
      if (row_begin % item_rows != 0) {
        AS_TRY(A64.buf.alloc(ctx, packed_elems(a_tiles_all, d)));
        AS_TRY(pack_f64(ctx, rows, norms, nullptr, row_begin, row_count, d, A64.buf.p, a_tiles_all));
        PA64 = A64.buf.p;
      }
    }
    const uint32_t items_all = (row_count + item_rows - 1) / item_rows;
    uint32_t bulk_items = items_all;
    if (items_all >= 512 && items_all % 256 != 0) bulk_items = items_all / 256 * 256;
    uint32_t splits_used = 1;
    for (int part = 0; part < 2; ++part) {
      const uint32_t it0 = part == 0 ? 0 : bulk_items;
      const uint32_t it1 = part == 0 ? bulk_items : items_all;
      if (it0 >= it1) continue;
      const uint32_t r0 = it0 * item_rows;                         // first row of this part (relative)
      const uint32_t rc = std::min(row_count, it1 * item_rows) - r0;  // its row count
      const uint32_t a_tiles = (rc + kTileRows - 1) / kTileRows;
      pb.arows = ARows{nullptr, (uint32_t)row_begin + r0};
      pb.nA = rc;
      pb.splits = clamp_splits(pick_splits(it1 - it0, b_tiles), d, KC);
      if (part == 1 && it1 - it0 < 256)  // remainder round: one work unit per workgroup, as short as possible
        pb.splits = clamp_splits(std::max(1u, std::min({16u, 256u / (it1 - it0), b_tiles})), d, KC);
      splits_used = std::max(splits_used, pb.splits);
      DevBuf<double> ckey;
      DevBuf<uint32_t> cidx;
      AS_TRY(ckey.alloc(ctx, (size_t)rc * pb.splits * KC));
      AS_TRY(cidx.alloc(ctx, (size_t)rc * pb.splits * KC));
      pb.cand_key = ckey.p;
      pb.cand_idx = cidx.p;
      if (use_f32) {
        AS_TRY(candidates_f32(ctx, pb, PA32 + (size_t)(r0 / kTileRows) * packed_elems(1, d), a_tiles, B32.p, b_tiles,
                              "knn_mfma_candidates_f32"));
        if (ctx->opt_debug) continue;  // timing experiment: candidate pass only, outputs undefined
        GraphBound bnd{cos_delta_f32(d), rectify};
        AS_TRY(launch_rerank(ctx, "knn_rerank", rows, rows, d, nullptr, (uint32_t)row_begin + r0, rc, r0, K, KC,
                             pb.splits, ckey.p, cidx.p, force.p, pol, bnd, out_dist, out_idx, fail1.p, fail_count.p,
                             err.p));
      } else {
        AS_TRY(candidates_f64(ctx, pb, PA64 + (size_t)(r0 / kTileRows) * packed_elems(1, d), a_tiles, B64.p, b_tiles,
                              "knn_mfma_candidates"));
        GraphBound bnd{cos_delta_f64(d), rectify};
        AS_TRY(launch_rerank(ctx, "knn_rerank", rows, rows, d, nullptr, (uint32_t)row_begin + r0, rc, r0, K, KC,
                             pb.splits, ckey.p, cidx.p, force.p, pol, bnd, out_dist, out_idx, fail1.p, fail_count.p,
                             err.p));
      }
      // (no synchronisation to hand the workspace back: a block returned to the cache is reused by later work on this same stream only)
    }
    if (ctx->opt_debug) return AS_OK;
    pb.splits = splits_used;
    pb.arows = ARows{nullptr, (uint32_t)row_begin};
    pb.nA = row_count;
    AS_TRY(read_u32(ctx, fail_count.p, &n_fail));
  }
  ctx->stats["knn_uncertified_rows"] = n_fail;
  ctx->stats["knn_join_delta_e9"] = use_join ? (int64_t)(join_delta(ctx, d) * 1e9) : 0;
  ctx->stats["knn_uncertified_rows_stage1a"] = 0;
  ctx->stats["knn_stream_scan_rows"] = 0;
  ctx->stats["knn_uncertified_rows_stage1b"] = 0;
  ctx->stats["knn_uncertified_rows_stage2"] = 0;
  ctx->stats["knn_cand_capacity"] = KC;
  ctx->stats["knn_col_splits"] = pb.splits;
  if (n_fail == 0) return AS_OK;

  // what the later stages need to know about the join behind the lists
  const bool grouped_theta = oj && oj->grouped_theta;                                // thresholds from the sample of the nearest pivot groups
  const bool wide_pre = oj && oj->wide_pre;                                          // the rotated coordinates of the rows are in use
  const double *rot_z = oj ? oj->rot_z.p : nullptr;                                  // ... n x 32
  const uint32_t *perm_used = (oj && oj->reorder) ? oj->perm_own.p : nullptr;        // image position -> row of the f16 image js->packedh
  // the lists of row g: index (lists_inv ? lists_inv[g] : g) - list_base (second_theta_kernel); its looser threshold: theta2[g - t2_base]
  const uint32_t *lists_inv = (lists_by_pos || (js && js->a_ids)) ? oj->inv_own.p : nullptr;
  const uint32_t list_base = (js && js->a_ids) ? js->list_base : (lists_by_pos ? 0u : (uint32_t)row_begin);
  const uint32_t t2_base = (js && js->a_ids) ? 0u : (uint32_t)row_begin;
  // Failed rows go through the stages that follow, each on the gathered list of what the one before
  // could not certify: (after the join only) the f32 top-k pass, the f64 MFMA pass, the exact scan.
  // cur_ids = global ids of the rows still open, cur_pos = their positions in the output range.
  DevBuf<uint32_t> gids;
  AS_TRY(gids.alloc(ctx, n_fail));
  {
    const uint32_t *a_ids_f = js ? js->a_ids : nullptr;  // (a range of image positions: the rows are a list)
    hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(n_fail, 256)), dim3(256), 0, ctx->stream, a_ids_f, fail1.p, n_fail,
                       a_ids_f ? 0u : (uint32_t)row_begin, gids.p);
  }
  const uint32_t *cur_ids = gids.p;
  const uint32_t *cur_pos = fail1.p;
  uint32_t cur_n = n_fail;
  DevBuf<uint32_t> ids_a, pos_a, ids_b, pos_b;

  auto mfma_stage = [&](bool f32, DevBuf<uint32_t> &ids_out, DevBuf<uint32_t> &pos_out, const char *stat) -> int {
    const uint32_t a_tiles = (cur_n + kTileRows - 1) / kTileRows;
    pb.arows = ARows{cur_ids, 0};
    pb.nA = cur_n;
    pb.splits = clamp_splits(pick_splits(f32 ? (a_tiles + 1) / 2 : a_tiles, b_tiles), d, KC);
    DevBuf<double> ckey, ok, A64g;
    DevBuf<uint32_t> cidx, oi, failx;
    DevBuf<float> A32g;
    AS_TRY(ckey.alloc(ctx, (size_t)cur_n * pb.splits * KC));
    AS_TRY(cidx.alloc(ctx, (size_t)cur_n * pb.splits * KC));
    pb.cand_key = ckey.p;
    pb.cand_idx = cidx.p;
    if (f32) {
      if (!B32.p && js && js->packed) B32.p = js->packed;
      if (!B32.p) {
        AS_TRY(B32.buf.alloc(ctx, packed_elems(b_alloc, d)));
        AS_TRY(pack_f32(ctx, rows, norms, nullptr, 0, n, d, B32.buf.p, b_alloc));
        B32.p = B32.buf.p;
      }
      const uint32_t at = (a_tiles + 1) & ~1u;
      AS_TRY(A32g.alloc(ctx, packed_elems(at, d)));
      AS_TRY(pack_f32(ctx, rows, norms, cur_ids, 0, cur_n, d, A32g.p, at));
      AS_TRY(candidates_f32(ctx, pb, A32g.p, a_tiles, B32.p, b_tiles, "knn_mfma_candidates_f32"));
    } else {
      if (!B64.p) {
        AS_TRY(B64.buf.alloc(ctx, packed_elems(b_alloc, d)));
        AS_TRY(pack_f64(ctx, rows, norms, nullptr, 0, n, d, B64.buf.p, b_alloc));
        B64.p = B64.buf.p;
      }
      AS_TRY(A64g.alloc(ctx, packed_elems(a_tiles, d)));
      AS_TRY(pack_f64(ctx, rows, norms, cur_ids, 0, cur_n, d, A64g.p, a_tiles));
      AS_TRY(candidates_f64(ctx, pb, A64g.p, a_tiles, B64.p, b_tiles, "knn_mfma_candidates"));
    }
    AS_TRY(ok.alloc(ctx, (size_t)cur_n * K));
    AS_TRY(oi.alloc(ctx, (size_t)cur_n * K));
    AS_TRY(failx.alloc(ctx, cur_n));
    AS_HIP(ctx, hipMemsetAsync(fail_count.p, 0, sizeof(uint32_t), ctx->stream));
    GraphBound bnd{f32 ? cos_delta_f32(d) : cos_delta_f64(d), rectify};
    AS_TRY(launch_rerank(ctx, "knn_rerank", rows, rows, d, cur_ids, 0u, cur_n, 0u, K, KC, pb.splits, ckey.p, cidx.p,
                         force.p, pol, bnd, ok.p, oi.p, failx.p, fail_count.p, err.p));
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv((uint64_t)cur_n * K, 256)), dim3(256), 0, ctx->stream, cur_pos,
                       cur_n, K, ok.p, oi.p, out_dist, out_idx);
    uint32_t n_next = 0;
    AS_TRY(read_u32(ctx, fail_count.p, &n_next));
    ctx->stats[stat] = n_next;
    if (n_next) {
      AS_TRY(ids_out.alloc(ctx, n_next));
      AS_TRY(pos_out.alloc(ctx, n_next));
      hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(n_next, 256)), dim3(256), 0, ctx->stream, cur_ids, failx.p, n_next,
                         0u, ids_out.p);
      hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(n_next, 256)), dim3(256), 0, ctx->stream, cur_pos, failx.p, n_next,
                         0u, pos_out.p);
      // (no synchronisation to hand the workspace back: a block returned to the cache is reused by later work on this same stream only)
      cur_ids = ids_out.p;
      cur_pos = pos_out.p;
    }
    cur_n = n_next;
    return AS_OK;
  };

  // ---- stage 1a (after a join on the bf16 pipe): the one-sided join over the gathered rows ----
  // A row fails the first pass because its list overflowed (threshold too low for its neighbourhood: same threshold,
  // longer list) or because its K-th neighbour is not clear of the threshold (too high: the looser second threshold).
  // ~1 % of the rows; as a one-sided sweep of the split image this is a few ms, the f32 top-k pass over them 11-22.
  DevBuf<uint32_t> ids_r, pos_r, ids_w, pos_w;
  DevBuf<unsigned char> P16late;
  // rows wider than 128 after the f16 join: the same pass on the f16 plane (one-sided form of mfma_join_f16_wide.hip);
  // the looser thresholds carry it, the bound stays the f16 one
  const bool second_wide = join_on_f16(ctx, d) && d > 128 && js && js->packedh;
  // up to 128 wide after the f16 join: the pass on the f16 plane as well -- the search mode of pair_join_f16_kernel with
  // alpha = 1, beta = 0 and zero lambdas is exactly a one-sided threshold pass (a row lists itself; the re-rank drops it) --
  // which saves building the split-bf16 image of all rows for half a percent of them (opt second_pass_f16 = 0: as before)
  // (where the streaming exact scan takes what is left: the f16 bound leaves a few more rows open than the split form)
  const bool second_h = join_on_f16(ctx, d) && d == 128 && js && js->packedh && ctx->opt_second_pass_f16 && ctx->opt_stream_scan &&
                        stream_scan_applies(1, n, d, K);
  if (use_join && js->theta2 && (second_pass_bf16(ctx, d) || second_wide) && cur_n > 0 && cur_n <= n / 4) {
    const uint32_t nf = cur_n, cap2 = 1024;
    const void *split_image = (second_wide || second_h) ? js->packedh : js->packed16;
    if (!split_image) {  // the join ran on the f16 plane: the split-bf16 image of all rows is only built when rows are left
      AS_TRY(P16late.alloc(ctx, packed_vecs_bf16(b_alloc, d) * 16));
      AS_TRY(pack_bf16(ctx, rows, norms, nullptr, 0, n, d, P16late.p, b_alloc));
      split_image = P16late.p;
    }
    const uint32_t bt = (nf + kTileRows - 1) / kTileRows, b_al = second_h ? (bt + 3) & ~3u : (bt + 1) & ~1u, pad = b_al * kTileRows;
    // The f16 pass behind a ONE-SIDED cone bound (cone_tiles.hip: the cones of the image's tiles are kept from the first pass): the
    // open rows are put in the order of the image first -- a wave's 64 rows are then neighbours, and most column tiles lie outside
    // what any of them can reach.
    DevBuf<uint32_t> ids_s, pos_s;
    const bool cone_second = second_h && oj && oj->cone.valid && oj->rot_z.p && oj->inv_own.p && ctx->opt_join_cone_second && nf >= 512 &&
                             !(ctx->opt_debug & 2048);
    if (cone_second) {
      DevBuf<double> keys;
      DevBuf<uint32_t> ord;
      AS_TRY(keys.alloc(ctx, nf));
      AS_TRY(ord.alloc(ctx, nf));
      AS_TRY(ids_s.alloc(ctx, nf));
      AS_TRY(pos_s.alloc(ctx, nf));
      hipLaunchKernelGGL(position_keys_kernel, dim3(cdiv(nf, 256)), dim3(256), 0, ctx->stream, cur_ids, (const uint32_t *)oj->inv_own.p, nf, keys.p);
      AS_TRY(order_by_value(ctx, keys.p, nf, ord.p));
      hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(nf, 256)), dim3(256), 0, ctx->stream, cur_ids, (const uint32_t *)ord.p, nf, 0u, ids_s.p);
      hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(nf, 256)), dim3(256), 0, ctx->stream, cur_pos, (const uint32_t *)ord.p, nf, 0u, pos_s.p);
      AS_HIP(ctx, hipGetLastError());
      cur_ids = ids_s.p;
      cur_pos = pos_s.p;
    }
    DevBuf<unsigned char> PA;
    DevBuf<float> theta_a, zeros;
    DevBuf<uint32_t> cnt_b, cand_b, unitc, oi, failx;
    DevBuf<double> ok;
    AS_TRY(PA.alloc(ctx, ((second_wide || second_h) ? packed_vecs_f16(b_al, d) : packed_vecs_bf16(b_al, d)) * 16));
    AS_TRY(theta_a.alloc(ctx, pad));
    AS_TRY(cnt_b.alloc(ctx, pad));
    AS_TRY(cand_b.alloc(ctx, (size_t)pad * cap2 * 2));
    AS_TRY(unitc.alloc(ctx, 1));
    AS_TRY(ok.alloc(ctx, (size_t)nf * K));
    AS_TRY(oi.alloc(ctx, (size_t)nf * K));
    AS_TRY(failx.alloc(ctx, nf));
    if (second_wide || second_h) AS_TRY(pack_f16(ctx, rows, norms, cur_ids, 0, nf, d, PA.p, b_al));
    else AS_TRY(pack_bf16(ctx, rows, norms, cur_ids, 0, nf, d, PA.p, b_al));
    const float *band_own = ((second_h || second_wide) && oj && oj->theta_band_own.p && ctx->opt_join_theta_band_second) ? oj->theta_band_own.p : nullptr;
    const float band_by = (float)(2.0 * join_delta(ctx, d) + 2e-6);
    if (grouped_theta && second_h && !(ctx->opt_debug & 131072)) {  // (debug bit 131072: the first pass's own second thresholds)
      // the thresholds of the first pass came from the sampled rows of the nearest pivot groups only; a row that is open after it
      // may be one whose neighbours live elsewhere (a cluster without a pivot of its own), so the second pass takes thresholds
      // against the whole sample for the few thousand rows it sweeps (PA is their f16 image)
      const int64_t rank_opt = ctx->opt_join_rank > 0 ? ctx->opt_join_rank : 4;
      const uint32_t m = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(rank_opt, 32)), m2 = std::min<uint32_t>(48, 2 * m + 6);
      const uint32_t ns = std::max(256u, (n / 32u) & ~255u);
      DevBuf<uint32_t> sid;
      DevBuf<float> t1, t2;
      AS_TRY(sid.alloc(ctx, ns));
      AS_TRY(t1.alloc(ctx, pad));
      AS_TRY(t2.alloc(ctx, pad));
      hipLaunchKernelGGL(sample_ids_kernel, dim3(cdiv(ns, 256)), dim3(256), 0, ctx->stream, ns, n / ns, sample_half_bits(ns), sid.p);
      AS_TRY(join_thresholds_f16(ctx, rows, norms, d, PA.p, 0, nf, sid.p, ns, m, t1.p, m2, t2.p));
      hipLaunchKernelGGL(second_theta_fresh_kernel, dim3(cdiv(pad, 256)), dim3(256), 0, ctx->stream, (const float *)t1.p, (const float *)t2.p,
                         js->theta, js->cnt, js->n_src, list_base, row_count, js->cap, cur_ids, nf, pad, theta_a.p, lists_inv, band_own, band_by);
    } else {
      hipLaunchKernelGGL(second_theta_kernel, dim3(cdiv(pad, 256)), dim3(256), 0, ctx->stream, js->theta, js->theta2, js->cnt,
                         js->n_src, t2_base, list_base, row_count, js->cap, cur_ids, nf, pad, theta_a.p, lists_inv, band_own, band_by);
    }
    AS_HIP(ctx, hipMemsetAsync(cnt_b.p, 0, sizeof(uint32_t) * pad, ctx->stream));
    AS_HIP(ctx, hipMemsetAsync(unitc.p, 0, sizeof(uint32_t), ctx->stream));
    AS_HIP(ctx, hipMemsetAsync(fail_count.p, 0, sizeof(uint32_t), ctx->stream));
    if (second_h) {
      const size_t zn = std::max<size_t>(((size_t)n + 511) / 512 * 512, pad);
      AS_TRY(zeros.alloc(ctx, zn));
      AS_HIP(ctx, hipMemsetAsync(zeros.p, 0, sizeof(float) * zn, ctx->stream));
      DevBuf<unsigned char> mask2;
      bool masked = false;
      if (cone_second) {
        double kept = 1.0;
        AS_TRY(mask2.alloc(ctx, (size_t)(pad / 512u) * 2 * ((n + 511) / 512) * 8));
        AS_TRY(cone_rect_mask(ctx, oj->rot_z.p, cur_ids, nf, theta_a.p, pad, oj->cone, n, mask2.p, &kept));
        masked = kept <= (double)ctx->opt_join_cone_tiles_max_e3 * 1e-3;
      }
      ctx->stats["knn_join_cone_second"] = masked ? 1 : 0;
      AS_TRY(join_f16_search(ctx, split_image, n, d, PA.p, nf, theta_a.p, theta_a.p, zeros.p, zeros.p, 1.f, 0.f, cap2, cnt_b.p, cand_b.p,
                             unitc.p, "knn_mfma_join_f16_second", perm_used, nullptr, masked ? mask2.p : nullptr, cur_ids));
    } else if (second_wide)
      AS_TRY(join_f16_wide_rect(ctx, split_image, n, d, PA.p, nf, cur_ids, theta_a.p, cap2, cnt_b.p, cand_b.p, "knn_mfma_join_f16_second",
                                perm_used));
    else
      AS_TRY(join_bf16_rect(ctx, split_image, n, d, PA.p, nf, cur_ids, theta_a.p, cap2, cnt_b.p, cand_b.p, unitc.p,
                            "knn_mfma_join_bf16_second"));
    const double delta2 = (second_wide || second_h) ? join_delta(ctx, d) : rect_delta(d);
    {
      GraphBound bnd{delta2, rectify};
      ProfScope ps(ctx, "knn_rerank");
      const size_t lds = (size_t)4 * (d + cap2) * sizeof(double) + (size_t)4 * cap2 * sizeof(uint32_t);
      AS_HIP(ctx, hipFuncSetAttribute((const void *)rerank_join_kernel<GraphDistPolicy, GraphBound>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL((rerank_join_kernel<GraphDistPolicy, GraphBound>), dim3(std::min<uint64_t>(cdiv(nf, 4), 8192)),
                         dim3(256), lds, ctx->stream, rows, rows, cur_ids, d, 0u, nf, theta_a.p, 1u, cnt_b.p, cand_b.p, cap2, cap2,
                         K, force.p, pol, bnd, (ctx->opt_debug & 16) ? -1.0 : delta2, rectify, ok.p, oi.p, failx.p,
                         fail_count.p, err.p);
      AS_HIP(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv((uint64_t)nf * K, 256)), dim3(256), 0, ctx->stream, cur_pos, nf, K, ok.p,
                       oi.p, out_dist, out_idx);
    uint32_t n_next = 0;
    AS_TRY(read_u32(ctx, fail_count.p, &n_next));
    ctx->stats["knn_uncertified_rows_stage1a"] = n_next;
    if (n_next) {
      AS_TRY(ids_r.alloc(ctx, n_next));
      AS_TRY(pos_r.alloc(ctx, n_next));
      hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(n_next, 256)), dim3(256), 0, ctx->stream, cur_ids, failx.p, n_next, 0u,
                         ids_r.p);
      hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(n_next, 256)), dim3(256), 0, ctx->stream, cur_pos, failx.p, n_next, 0u,
                         pos_r.p);
      // (no synchronisation to hand the workspace back: a block returned to the cache is reused by later work on this same stream only)
      cur_ids = ids_r.p;
      cur_pos = pos_r.p;
    }
    cur_n = n_next;
    if (cur_n == 0) return AS_OK;
  }

  // ---- a handful of rows left: one exact pass over the matrix per 32 of them (stream_scan.hip) instead of MFMA sweeps ----
  if (ctx->opt_stream_scan && cur_n <= kStreamRowsMax && stream_scan_applies(std::min<uint32_t>(cur_n, 32), n, d, K)) {
    DevBuf<double> fk;
    DevBuf<uint32_t> fi, redo_ids, redo_pos_d;
    AS_TRY(fk.alloc(ctx, (size_t)cur_n * K));
    AS_TRY(fi.alloc(ctx, (size_t)cur_n * K));
    std::vector<uint32_t> redo_all;
    // with the rotated coordinates of the own join at hand the list pass of the scan is the int8 filter + the pairs it leaves
    // (stream_scan.hip): the image of all rows and of each batch are packed from the coordinates
    DevBuf<unsigned char> f_p8, f_q8;
    StreamFilter flt{};
    bool use_flt = false;
    if (wide_pre && rot_z && d == 128 && ctx->opt_search_stream_filter && n < (1u << 27)) {
      const uint32_t tiles = packed_tiles_f16(n);
      double e_items = 0.0;
      AS_TRY(f_p8.alloc(ctx, (size_t)tiles * 4 * 64 * 16));
      AS_TRY(f_q8.alloc(ctx, (size_t)4 * 64 * 16));
      AS_TRY(pack_rotated_from_z(ctx, rot_z, nullptr, n, tiles, f_p8.p, &e_items));
      flt.P8 = f_p8.p;
      flt.err_items = (float)(e_items * (1.0 + 1e-6));
      use_flt = true;
    }
    for (uint32_t c0 = 0; c0 < cur_n; c0 += 32) {
      const uint32_t cn = std::min<uint32_t>(32, cur_n - c0);
      std::vector<uint32_t> redo;
      if (use_flt) {
        double e_a = 0.0;
        AS_TRY(pack_rotated_from_z(ctx, rot_z, cur_ids + c0, cn, 1, f_q8.p, &e_a));
        flt.PQ8 = f_q8.p;
        flt.err_a = (float)(e_a * (1.0 + 1e-6));
      }
      AS_TRY(stream_scan_graph(ctx, rows, cur_ids + c0, cn, n, K, pol, fk.p + (size_t)c0 * K, fi.p + (size_t)c0 * K, err.p, &redo,
                               use_flt ? &flt : nullptr));
      for (uint32_t r : redo) redo_all.push_back(c0 + r);
    }
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv((uint64_t)cur_n * K, 256)), dim3(256), 0, ctx->stream, cur_pos, cur_n, K, fk.p,
                       fi.p, out_dist, out_idx);
    AS_HIP(ctx, hipGetLastError());
    ctx->stats["knn_stream_scan_rows"] = cur_n;
    if (redo_all.empty()) return AS_OK;
    // rows whose candidate list overflowed (a sample that misses their neighbourhood): the exact scan, below
    AS_TRY(redo_pos_d.alloc(ctx, redo_all.size()));
    AS_HIP(ctx, copy_h2d(ctx, redo_pos_d.p, redo_all.data(), sizeof(uint32_t) * redo_all.size()));
    const uint32_t nr = (uint32_t)redo_all.size();
    AS_TRY(ids_r.alloc(ctx, nr));
    AS_TRY(pos_r.alloc(ctx, nr));
    hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(nr, 256)), dim3(256), 0, ctx->stream, cur_ids, redo_pos_d.p, nr, 0u, ids_r.p);
    hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(nr, 256)), dim3(256), 0, ctx->stream, cur_pos, redo_pos_d.p, nr, 0u, pos_r.p);
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    cur_ids = ids_r.p;
    cur_pos = pos_r.p;
    cur_n = nr;
    DevBuf<double> fk2;
    DevBuf<uint32_t> fi2;
    AS_TRY(fk2.alloc(ctx, (size_t)cur_n * K));
    AS_TRY(fi2.alloc(ctx, (size_t)cur_n * K));
    AS_TRY(knn_exact(ctx, rows, norms, n, d, cur_ids, 0, cur_n, K, rectify, fi2.p, fk2.p, 0));
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv((uint64_t)cur_n * K, 256)), dim3(256), 0, ctx->stream, cur_pos, cur_n, K, fk2.p,
                       fi2.p, out_dist, out_idx);
    AS_HIP(ctx, hipGetLastError());
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return AS_OK;
  }

  // ---- rows of 256 .. 1024 values, a handful left: the quick pass over all rows + exact chains (stream_scan.hip) instead of an
  // f32 top-k sweep for them (47 ms for ten rows at 1 M x 768) ----
  if (ctx->opt_stream_scan && ctx->opt_search_stream_quick && cur_n <= kStreamRowsMax &&
      stream_quick_wide_applies(std::min<uint32_t>(cur_n, 32), n, d, K)) {
    DevBuf<double> fk;
    DevBuf<uint32_t> fi, redo_pos_d;
    AS_TRY(fk.alloc(ctx, (size_t)cur_n * K));
    AS_TRY(fi.alloc(ctx, (size_t)cur_n * K));
    std::vector<uint32_t> redo_all;
    for (uint32_t c0 = 0; c0 < cur_n; c0 += 32) {
      const uint32_t cn = std::min<uint32_t>(32, cur_n - c0);
      std::vector<uint32_t> redo;
      AS_TRY(stream_quick_wide_graph(ctx, rows, cur_ids + c0, cn, n, d, K, pol, fk.p + (size_t)c0 * K, fi.p + (size_t)c0 * K, err.p, &redo));
      for (uint32_t r : redo) redo_all.push_back(c0 + r);
    }
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv((uint64_t)cur_n * K, 256)), dim3(256), 0, ctx->stream, cur_pos, cur_n, K, fk.p, fi.p,
                       out_dist, out_idx);
    AS_HIP(ctx, hipGetLastError());
    ctx->stats["knn_stream_scan_rows"] = cur_n;
    if (redo_all.empty()) {
      AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
      return AS_OK;
    }
    const uint32_t nr = (uint32_t)redo_all.size();
    AS_TRY(redo_pos_d.alloc(ctx, nr));
    AS_HIP(ctx, copy_h2d(ctx, redo_pos_d.p, redo_all.data(), sizeof(uint32_t) * nr));
    AS_TRY(ids_w.alloc(ctx, nr));  // (own buffers: cur_ids / cur_pos may live in ids_r / pos_r)
    AS_TRY(pos_w.alloc(ctx, nr));
    hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(nr, 256)), dim3(256), 0, ctx->stream, cur_ids, redo_pos_d.p, nr, 0u, ids_w.p);
    hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(nr, 256)), dim3(256), 0, ctx->stream, cur_pos, redo_pos_d.p, nr, 0u, pos_w.p);
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    cur_ids = ids_w.p;
    cur_pos = pos_w.p;
    cur_n = nr;
  }

  // ---- stage 1b (after the join): f32 top-k pass over the gathered rows ----
  if (use_join && stages_fit && !exact_direct(cur_n, n)) {
    AS_TRY(mfma_stage(true, ids_a, pos_a, "knn_uncertified_rows_stage1b"));
    if (cur_n == 0) return AS_OK;
  }
  // ---- stage 2: f64 MFMA over the gathered rows ----
  if (use_f32 && stages_fit && !wide && !exact_direct(cur_n, n)) {
    AS_TRY(mfma_stage(false, ids_b, pos_b, "knn_uncertified_rows_stage2"));
    if (cur_n == 0) return AS_OK;
  }

  // ---- last stage: exact scan ----
  DevBuf<double> fk;
  DevBuf<uint32_t> fi;
  AS_TRY(fk.alloc(ctx, (size_t)cur_n * K));
  AS_TRY(fi.alloc(ctx, (size_t)cur_n * K));
  AS_TRY(knn_exact(ctx, rows, norms, n, d, cur_ids, 0, cur_n, K, rectify, fi.p, fk.p, 0));
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv((uint64_t)cur_n * K, 256)), dim3(256), 0, ctx->stream, cur_pos,
                     cur_n, K, fk.p, fi.p, out_dist, out_idx);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

// ---- K9 driver ----------------------------------------------------------------------------------
// max |v| over a vector as the bits of a non-negative double (atomicMax on the integer image is the
// max on the value); a NaN counts as +inf so that the caller takes the f64 epilogue.
__global__ void absmax_kernel(const double *__restrict__ v, uint64_t n, unsigned long long *__restrict__ out) {
  double m = 0.0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const double a = fabs(v[i]);
    m = a != a ? __builtin_inf() : fmax(m, a);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}

// ---- threshold scan of the search (stage 0 of K9) --------------------------------------------------
// score(q, i) = alpha cos(q, i) + beta lam(q, i), lam = 1 - min(|lambda_q - lambda_i|, 1) (src/core.rs:229-239).
// Every query gets a guess g_q for its K-th best score (the m-th best approximate score over a sample of the items,
// mfma_theta_bf16.hip); the search form of the one-sided bf16 join lists every item whose approximate score
// reaches g_q; the exact re-rank scores the listed items in f64, and a query is certified when its K-th best exact
// score is clear of what an unlisted item can reach, g_q + delta (delta: the cosine bound of the split-bf16
// products scaled by alpha, the f32 evaluation of the lambda term, the f32 roundings of the sum).  Any guess is
// correct; a guess above the true K-th score fails the certification and the query goes on to the MFMA top-k /
// exact stages.
__global__ void gather_lambda_f32_kernel(const double *__restrict__ lam, const uint32_t *__restrict__ ids, uint32_t count,
                                         uint32_t padded, float *__restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < padded) out[i] = i < count ? (float)lam[ids ? ids[i] : i] : 0.f;
}
// guesses padded to whole super tiles: +inf on padding rows (never a candidate), -inf for a NaN guess (everything
// is listed, the list overflows, the query falls through)
__global__ void scan_guess_kernel(const float *__restrict__ guess, const uint32_t *__restrict__ ids, uint32_t nq,
                                  uint32_t padded, float *__restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= padded) return;
  float g = __builtin_inff();
  if (i < nq) {
    g = guess[ids ? ids[i] : i];
    if (!(g == g)) g = -__builtin_inff();
  }
  out[i] = g;
}

// the same for the wide form of the scan (mfma_join_f16_wide.hip): what the cosine alone must reach, (g - beta) / alpha,
// and g - beta
__global__ void scan_guess_wide_kernel(const float *__restrict__ guess, const uint32_t *__restrict__ ids, uint32_t nq, uint32_t padded,
                                       float alpha, float beta, float *__restrict__ cos_floor, float *__restrict__ g_minus_beta) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= padded) return;
  float g = __builtin_inff();
  if (i < nq) {
    g = guess[ids ? ids[i] : i];
    if (!(g == g)) g = -__builtin_inff();
  }
  g_minus_beta[i] = g - beta;
  cos_floor[i] = (g - beta) / alpha;
}

int search_mfma(as_ctx *ctx, as_index *ix, const double *queries, const double *q_norms, const double *q_lambdas,
                uint64_t nq64, uint32_t k, double alpha, uint32_t *out_idx, double *out_key) {
  const uint32_t n = (uint32_t)ix->n, d = (uint32_t)ix->d, nq = (uint32_t)nq64;
  const int NC = (int)((d + 31) / 32);
  const uint32_t K = std::min<uint32_t>(k, n);
  const uint32_t KC = cand_capacity(ctx, K);
  const bool wide = NC > kMaxNC;
  const bool use_f32 = (ctx->opt_cand_precision != 64 || wide) && fits_f32(KC);
  // deep lists (large k) and small matrices do not fit the MFMA top-k passes; the threshold scan (stage 0) has no
  // such limit, so what it leaves open then goes straight to the exact scan
  const bool mfma_ok = !((wide && !use_f32) || !fits_f64(KC) || n < 4 * kTileRows);
  if (nq == 0 || n < 4 * kTileRows || !rerank_fits(d))
    return search_exact(ctx, ix, queries, q_norms, q_lambdas, nullptr, nq, nq, k, alpha, out_idx, out_key);
  // an index that is searched again gets its items ordered by direction and a rotated int8 image (below); with those a small
  // batch reads 32 MB + the few blocks its queries can reach instead of streaming the 1 GB of rows
  const uint32_t prior_calls = ix->scan_calls.fetch_add(1);
  // shared for the whole search; a lazy build drops it, takes the index exclusively, builds (re-checking), synchronises the stream
  // and comes back (as_index, common.h).  Pointers of the index are read where they are used, after the builds.
  std::shared_lock<std::shared_mutex> reading(ix->rw);
#define AS_IX_BUILD(cond, ...)                                      \
  if (cond) {                                                       \
    reading.unlock();                                               \
    {                                                               \
      std::unique_lock<std::shared_mutex> building(ix->rw);         \
      if (cond) {                                                   \
        __VA_ARGS__                                                 \
        AS_HIP(ctx, hipStreamSynchronize(ctx->stream));             \
      }                                                             \
    }                                                               \
    reading.lock();                                                 \
  }
  const bool ordered_ok = !wide && n >= 65536 && ctx->opt_join_pipe == 16 && ctx->opt_search_scan_f16 && ctx->opt_search_scan &&
                          (ctx->opt_search_scan_order == 2 || (ctx->opt_search_scan_order == 1 && prior_calls >= 1));
  // items wider than 128 (round 5): the same two things -- an image in an order that groups the items by direction, found on their
  // rotated coordinates (rotate_wide.hip), and the rotated int8 image for a rectangular first pass in front of the panelled scan
  const bool ordered_ok_w = wide && n >= 65536 && rotated_wide_supported((uint32_t)d) && ctx->opt_join_pipe == 16 && ctx->opt_search_scan &&
                            ctx->opt_search_prefilter &&
                            // (a batch of 8 192 queries or more pays for the images -- ~30 ms per million items of 768 columns -- within
                            // its own scan, 1.5 ms per thousand queries there: such a batch gets them at the first search already)
                            (ctx->opt_search_scan_order == 2 || (ctx->opt_search_scan_order == 1 && (prior_calls >= 1 || nq >= 8192)));
  const bool prefer_staged = ordered_ok && ctx->opt_search_prefilter && d > 32 && ix->rot_state >= 0 && alpha > 0.0 && alpha <= 1.0 &&
                             nq >= (uint32_t)std::max<int64_t>(1, ctx->opt_search_staged_min_batch);
  // one query or a small batch: the bandwidth-bound regime.  One exact pass over the items (stream_scan.hip), no
  // candidate stages, nothing to certify.
  if (ctx->opt_stream_scan && stream_scan_applies(nq, n, d, K) && K == k && !prefer_staged) {
    DevBuf<int> err0;
    AS_TRY(err0.alloc(ctx, 1));
    AS_HIP(ctx, hipMemsetAsync(err0.p, 0, sizeof(int), ctx->stream));
    SearchScorePolicy pol0{q_norms, q_lambdas, ix->norms, ix->lambdas, alpha};
    std::vector<uint32_t> redo;
    // an index that is searched again keeps a rotated int8 image of its items (32 bytes each, in their own order): the list
    // pass then streams the ~1-3 % of the items that can reach some query's cut instead of all of them
    StreamFilter flt{};
    const StreamFilter *use_filter = nullptr;
    if (ctx->opt_search_stream_filter && prior_calls >= 1 && alpha > 0.0 && alpha <= 1.0 && ix->rot_state >= 0 && ix->weird_items != 1) {
      AS_IX_BUILD(ix->weird_items < 0 || (ix->weird_items == 0 && ix->rot_state == 0) || (ix->weird_items == 0 && ix->rot_state == 1 && !ix->packed8_nat),
      if (ix->weird_items < 0) {  // (a non-finite norm or lambda must reach the exact arithmetic: no filter for such an index)
        DevBuf<int> wc;
        AS_TRY(wc.alloc(ctx, 1));
        AS_HIP(ctx, hipMemsetAsync(wc.p, 0, sizeof(int), ctx->stream));
        hipLaunchKernelGGL(count_weird_items_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, ix->norms, ix->lambdas, n, wc.p);
        int h = 0;
        AS_HIP(ctx, copy_d2h(ctx, &h, wc.p, sizeof(h)));
        AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ix->weird_items = h ? 1 : 0;
      }
      if (ix->weird_items == 0 && ix->rot_state == 0) {
        double tail = 1.0;
        if (!ix->rot_q) AS_HIP(ctx, handle_malloc((void **)&ix->rot_q, sizeof(float) * 32 * 128));  // (kept on failure: the next call reuses it)
        AS_TRY(rotated_basis(ctx, ix->items, ix->norms, n, d, ix->rot_q, &tail));
        ix->rot_state = tail <= (double)ctx->opt_search_rot_tail_e3 * 1e-3 ? 1 : -1;
      }
      if (ix->weird_items == 0 && ix->rot_state == 1 && !ix->packed8_nat) {
        const uint32_t tiles = packed_tiles_f16(n);
        AS_HIP(ctx, handle_malloc(&ix->packed8_nat, rotated_image_bytes(tiles)));
        AS_TRY(rotated_image_pack(ctx, ix->items, ix->norms, nullptr, n, d, ix->rot_q, ix->packed8_nat, tiles, &ix->rot_err_nat));
      }
      )
      if (ix->weird_items == 0 && ix->rot_state == 1 && ix->packed8_nat) {
        flt = StreamFilter{ix->packed8_nat, (float)(ix->rot_err_nat * (1.0 + 1e-6)), ix->rot_q, q_norms, q_lambdas, ix->norms, ix->lambdas, alpha};
        use_filter = &flt;
      }
    }
    AS_TRY(stream_scan_search(ctx, queries, nullptr, nq, ix->items, n, K, pol0, out_key, out_idx, err0.p, &redo, use_filter));
    int h_err0 = 0;
    AS_HIP(ctx, copy_d2h(ctx, &h_err0, err0.p, sizeof(int)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_err0) return fail(ctx, AS_ERR_NAN_SCORE, as_status_message(AS_ERR_NAN_SCORE));
    ctx->stats["search_scan"] = 0;
    ctx->stats["search_stream_scan"] = 1;
    if (redo.empty()) return AS_OK;
    // a candidate list overflowed (a sample that misses the query's neighbourhood): the whole batch takes the staged path
  } else if (ctx->opt_stream_scan && ctx->opt_search_stream_quick && stream_quick_wide_applies(nq, n, d, K) && K == k) {
    // items of 256 .. 1024 values, a small batch: the quick pass over the items (sample, then all of them) and the reference's
    // chain for the pairs it leaves (stream_scan.hip): ~1 ms per 6 GB instead of the staged path's 22 ms at one query
    DevBuf<int> err0;
    AS_TRY(err0.alloc(ctx, 1));
    AS_HIP(ctx, hipMemsetAsync(err0.p, 0, sizeof(int), ctx->stream));
    SearchScorePolicy pol0{q_norms, q_lambdas, ix->norms, ix->lambdas, alpha};
    std::vector<uint32_t> redo;
    AS_TRY(stream_quick_wide_search(ctx, queries, nq, ix->items, n, d, K, pol0, out_key, out_idx, err0.p, &redo));
    int h_err0 = 0;
    AS_HIP(ctx, copy_d2h(ctx, &h_err0, err0.p, sizeof(int)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_err0) return fail(ctx, AS_ERR_NAN_SCORE, as_status_message(AS_ERR_NAN_SCORE));
    ctx->stats["search_scan"] = 0;
    ctx->stats["search_stream_scan"] = 1;
    if (redo.empty()) return AS_OK;
  }
  else ctx->stats["search_stream_scan"] = 0;
  const uint32_t b_tiles = (n + kTileRows - 1) / kTileRows;
  const uint32_t b_alloc = (b_tiles + 1) & ~1u;
  auto ensure64 = [&]() -> int {
    AS_IX_BUILD(!ix->packed,
      AS_HIP(ctx, handle_malloc((void **)&ix->packed, sizeof(double) * packed_elems(b_alloc, d)));
      AS_TRY(pack_f64(ctx, ix->items, ix->norms, nullptr, 0, n, d, ix->packed, b_alloc));
    )
    return AS_OK;
  };
  auto ensure32 = [&]() -> int {
    AS_IX_BUILD(!ix->packed32,
      AS_HIP(ctx, handle_malloc((void **)&ix->packed32, sizeof(float) * packed_elems(b_alloc, d)));
      AS_TRY(pack_f32(ctx, ix->items, ix->norms, nullptr, 0, n, d, ix->packed32, b_alloc));
    )
    return AS_OK;
  };

  DevBuf<uint32_t> fail1, fail_count;
  DevBuf<int> err;
  AS_TRY(fail1.alloc(ctx, nq));
  AS_TRY(fail_count.alloc(ctx, 1));
  AS_TRY(err.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(fail_count.p, 0, sizeof(uint32_t), ctx->stream));
  AS_HIP(ctx, hipMemsetAsync(err.p, 0, sizeof(int), ctx->stream));
  SearchScorePolicy pol{q_norms, q_lambdas, ix->norms, ix->lambdas, alpha};

  CandProblem pb{};
  pb.mode = 1;
  pb.A = queries;
  pb.a_norms = q_norms;
  pb.arows = ARows{nullptr, 0};
  pb.nA = nq;
  pb.B = ix->items;
  pb.b_norms = ix->norms;
  pb.nB = n;
  pb.d = d;
  pb.q_lambdas = q_lambdas;
  pb.x_lambdas = ix->lambdas;
  pb.alpha = alpha;
  pb.KC = KC;

  uint32_t n_fail = 0;
  // ---- stage 0: threshold scan on the bf16 pipe (see the note above gather_lambda_f32_kernel) ----
  // (items wider than 128: the same scan on the panelled f16 kernel, mfma_join_f16_wide.hip)
  const bool scan_wide = wide && d <= 2048 && d % 2 == 0 && ctx->opt_join_pipe == 16;
  // up to 128 wide: the scan itself on one f16 plane (MODE 1 of pair_join_f16_kernel), guesses still from split-bf16 products
  const bool scan_h = !wide && ctx->opt_join_pipe == 16 && ctx->opt_search_scan_f16;
  bool scan = ctx->opt_search_scan && ctx->opt_cand_precision != 64 && (!wide || scan_wide) && alpha > 0.0 && alpha <= 1.0 && K < 1024 &&
              n >= (uint32_t)std::max<int64_t>(4 * kTileRows, ctx->opt_join_min_rows);
  double lsum = 0.0;
  if (scan) {  // the f32 lambda term needs lambdas of moderate size (as in the f32 top-k pass below)
    DevBuf<unsigned long long> amax;
    AS_TRY(amax.alloc(ctx, 2));
    AS_HIP(ctx, hipMemsetAsync(amax.p, 0, 2 * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(absmax_kernel, dim3(std::min<uint64_t>(1024, cdiv(nq, 256))), dim3(256), 0, ctx->stream, q_lambdas,
                       (uint64_t)nq, amax.p);
    hipLaunchKernelGGL(absmax_kernel, dim3(std::min<uint64_t>(1024, cdiv(n, 256))), dim3(256), 0, ctx->stream, ix->lambdas,
                       (uint64_t)n, amax.p + 1);
    AS_HIP(ctx, hipGetLastError());
    double hmax[2] = {0.0, 0.0};
    AS_HIP(ctx, copy_d2h(ctx, hmax, amax.p, sizeof(hmax)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    lsum = hmax[0] + hmax[1];
    scan = lsum <= 64.0;  // NaN lambdas fail this too
  }
  ctx->stats["search_scan"] = scan ? 1 : 0;
  if (!scan && !mfma_ok)
    return search_exact(ctx, ix, queries, q_norms, q_lambdas, nullptr, nq, nq, k, alpha, out_idx, out_key);
  if (scan) {
    const double beta = 1.0 - alpha;
    // up to 128 wide: the image in an order that groups the items by direction (the listing of the scan then meets its pairs in
    // few blocks, and the first pass below can rule the others out); the scan translates positions back.  Ordering the items,
    // finding their directions and the second image cost ~5 ms at 1 M items, so an index gets them at its SECOND scan
    // (search_scan_order = 2: at the first): one-shot use pays nothing, a serving index pays once.
    const bool want_order = scan_h && ordered_ok;
    const uint32_t ns_want = std::max(256u, (n / (uint32_t)std::max<int64_t>(2, ctx->opt_search_scan_sample)) & ~255u);
    const bool want_order_w = scan_wide && ordered_ok_w && ix->rot_state >= 0;
    auto scan_state_missing = [&]() -> bool {
      return (want_order && !ix->scan_perm && ix->packedh) || ((scan_wide || scan_h) && !ix->packedh) ||
             (want_order_w && ix->rot_state == 0) || (want_order_w && ix->rot_state == 1 && (!ix->scan_perm || !ix->packed8 || !ix->lam32_pos)) ||
             (scan_h && ctx->opt_search_prefilter && d > 32 && (want_order || ix->scan_perm) && (ix->rot_state == 0 || (ix->rot_state == 1 && !ix->packed8))) ||
             (!scan_wide && !scan_h && !ix->packed16) ||
             !ix->lam32 || (scan_h && ix->scan_perm && !ix->lam32_pos) || !ix->scan_sample || ix->scan_ns != ns_want;
    };
    AS_IX_BUILD(scan_state_missing(),
    if (want_order && !ix->scan_perm && ix->packedh) {  // (the image of the first scan was in the order of the items)
      (void)handle_free(ix->packedh);
      ix->packedh = nullptr;
    }
    if (want_order_w && (ix->rot_state == 0 || (ix->rot_state == 1 && (!ix->scan_perm || !ix->packed8)))) {
      // wide items: rotated coordinates -> order, ordered f16 image, rotated int8 image by position; the directions stay with the index
      const uint32_t tiles = packed_tiles_f16(n);
      DevBuf<double> Zi, zn;
      double tail = 1.0;
      AS_TRY(Zi.alloc(ctx, (size_t)n * 32));
      AS_TRY(zn.alloc(ctx, n));
      if (!ix->rot_qw) AS_HIP(ctx, handle_malloc((void **)&ix->rot_qw, sizeof(float) * 31 * (size_t)rotated_wide_pitch(d)));
      AS_TRY(rotated_rows_wide(ctx, ix->items, ix->norms, n, d, Zi.p, zn.p, &tail, nullptr, ix->rot_qw));
      // (the cut of the narrow items -- a tenth of the energy outside the directions -- weighs a 32-wide pass against a 128-wide scan;
      // here the scan is d / 32 times the pass, as for the join of wide rows: the images are kept unless the directions carry nothing)
      ix->rot_state = tail <= std::max(0.5, (double)ctx->opt_search_rot_tail_e3 * 1e-3) ? 1 : -1;
      ctx->stats["search_rot_tail_e6"] = (int64_t)(tail * 1e6);
      if (ix->rot_state == 1) {
        if (ix->packedh) {  // (the image of the first scan was in the order of the items)
          (void)handle_free(ix->packedh);
          ix->packedh = nullptr;
        }
        if (!ix->scan_perm) AS_HIP(ctx, handle_malloc((void **)&ix->scan_perm, sizeof(uint32_t) * (size_t)n));
        if (!ix->scan_piv) AS_HIP(ctx, handle_malloc((void **)&ix->scan_piv, sizeof(float) * 256 * 128));
        AS_TRY(cluster_order(ctx, Zi.p, zn.p, n, 32, ix->scan_perm, ix->scan_piv));
        AS_HIP(ctx, handle_malloc(&ix->packedh, packed_vecs_f16(tiles, d) * 16));
        AS_TRY(pack_f16_all(ctx, ix->items, ix->norms, n, d, ix->packedh, tiles, ix->scan_perm));
        ix->f16_err = ctx->f16_err;
        if (!ix->packed8) AS_HIP(ctx, handle_malloc(&ix->packed8, rotated_image_bytes(tiles)));
        AS_TRY(pack_rotated_from_z(ctx, Zi.p, ix->scan_perm, n, tiles, ix->packed8, &ix->rot_err));
      }
    }
    if ((scan_wide || scan_h) && !ix->packedh) {
      const uint32_t tiles = packed_tiles_f16(n);
      if (want_order) {
        AS_HIP(ctx, handle_malloc((void **)&ix->scan_perm, sizeof(uint32_t) * (size_t)n));
        AS_HIP(ctx, handle_malloc((void **)&ix->scan_piv, sizeof(float) * 256 * 128));
        AS_TRY(cluster_order(ctx, ix->items, ix->norms, n, d, ix->scan_perm, ix->scan_piv));
      }
      AS_HIP(ctx, handle_malloc(&ix->packedh, packed_vecs_f16(tiles, d) * 16));
      AS_TRY(pack_f16_all(ctx, ix->items, ix->norms, n, d, ix->packedh, tiles, ix->scan_perm));
      ix->f16_err = ctx->f16_err;
    }
    if (scan_h && ix->scan_perm && ctx->opt_search_prefilter && d > 32) {
      const uint32_t tiles = packed_tiles_f16(n);
      if (ix->rot_state == 0) {
        double tail = 1.0;
        if (!ix->rot_q) AS_HIP(ctx, handle_malloc((void **)&ix->rot_q, sizeof(float) * 32 * 128));
        AS_TRY(rotated_basis(ctx, ix->items, ix->norms, n, d, ix->rot_q, &tail));
        ix->rot_state = tail <= (double)ctx->opt_search_rot_tail_e3 * 1e-3 ? 1 : -1;
      }
      // (the directions may have been found by a small batch before, for its own image in the order of the items: the image by
      // position is this path's to build -- it used to be built only together with the directions, and a large batch that came
      // after a small one then handed the first pass a null image)
      if (ix->rot_state == 1 && !ix->packed8) {
        AS_HIP(ctx, handle_malloc(&ix->packed8, rotated_image_bytes(tiles)));
        AS_TRY(rotated_image_pack(ctx, ix->items, ix->norms, ix->scan_perm, n, d, ix->rot_q, ix->packed8, tiles, &ix->rot_err));
      }
    }
    if (!scan_wide && !scan_h && !ix->packed16) {
      AS_HIP(ctx, handle_malloc(&ix->packed16, packed_vecs_bf16(b_alloc, d) * 16));
      AS_TRY(pack_bf16(ctx, ix->items, ix->norms, nullptr, 0, n, d, ix->packed16, b_alloc));
    }
    if (!ix->lam32) {  // padded to whole 512-row super tiles (the images of both forms of the scan end there)
      const uint32_t lpad = (n + 511) / 512 * 512;
      AS_HIP(ctx, handle_malloc((void **)&ix->lam32, sizeof(float) * (size_t)lpad));
      hipLaunchKernelGGL(gather_lambda_f32_kernel, dim3(cdiv((uint64_t)lpad, 256)), dim3(256), 0, ctx->stream,
                         ix->lambdas, (const uint32_t *)nullptr, n, lpad, ix->lam32);
    }
    if ((scan_h || scan_wide) && ix->scan_perm && !ix->lam32_pos) {  // ... and by position in the ordered image
      const uint32_t lpad = (n + 511) / 512 * 512;
      AS_HIP(ctx, handle_malloc((void **)&ix->lam32_pos, sizeof(float) * (size_t)lpad));
      hipLaunchKernelGGL(gather_lambda_f32_kernel, dim3(cdiv((uint64_t)lpad, 256)), dim3(256), 0, ctx->stream, ix->lambdas,
                         (const uint32_t *)ix->scan_perm, n, lpad, ix->lam32_pos);
    }
    if (ix->scan_sample && ix->scan_ns != ns_want) {  // the option changed since the sample was drawn
      (void)handle_free(ix->scan_sample);
      (void)handle_free(ix->scan_sample_lam);
      ix->scan_sample = nullptr;
      ix->scan_sample_lam = nullptr;
    }
    if (!ix->scan_sample) {
      const uint32_t ns = ns_want;
      AS_HIP(ctx, handle_malloc((void **)&ix->scan_sample, sizeof(uint32_t) * ns));
      AS_HIP(ctx, handle_malloc((void **)&ix->scan_sample_lam, sizeof(float) * ns));
      hipLaunchKernelGGL(sample_ids_kernel, dim3(cdiv(ns, 256)), dim3(256), 0, ctx->stream, ns, n / ns, sample_half_bits(ns),
                         ix->scan_sample);
      hipLaunchKernelGGL(gather_lambda_f32_kernel, dim3(cdiv(ns, 256)), dim3(256), 0, ctx->stream, ix->lambdas,
                         ix->scan_sample, ns, ns, ix->scan_sample_lam);
      AS_HIP(ctx, hipGetLastError());
      ix->scan_ns = ns;
    }
    )
    const float *x_lam_h = ((scan_h || scan_wide) && ix->scan_perm) ? ix->lam32_pos : ix->lam32;  // what the f16 scan reads: by image position
    const uint32_t ns = ix->scan_ns;
    const uint32_t m = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(ctx->opt_search_scan_rank, 32));
    const uint32_t a_tiles = (nq + kTileRows - 1) / kTileRows, a_alloc = scan_h ? (a_tiles + 3) & ~3u : (a_tiles + 1) & ~1u;
    const uint32_t padded = a_alloc * kTileRows;  // whole super tiles of the kernel that scans (512 rows on the f16 plane, else 256)
    const uint32_t cap = (uint32_t)std::max<int64_t>(K + 1, std::min<int64_t>(ctx->opt_search_scan_cap, 1024));
    DevBuf<unsigned char> PQ, PQh;
    DevBuf<float> qlam, guess, guess2, guess_pad;
    // the looser guess of the second pass over what fails: rank 2 m + 6, as far as its ~rank x rate candidates fit the lists
    const uint32_t m2 = std::max(m, std::min<uint32_t>(std::min<uint32_t>(48, 2 * m + 6), 512u / std::max(1u, n / ns)));
    DevBuf<uint32_t> cnt, cand, unitc;
    AS_TRY(PQ.alloc(ctx, (scan_wide ? packed_vecs_f16(a_alloc, d) : packed_vecs_bf16(a_alloc, d)) * 16));
    AS_TRY(qlam.alloc(ctx, padded));
    AS_TRY(guess.alloc(ctx, nq));
    AS_TRY(guess2.alloc(ctx, nq));
    AS_TRY(guess_pad.alloc(ctx, padded));
    AS_TRY(cnt.alloc(ctx, padded));
    AS_TRY(cand.alloc(ctx, (size_t)padded * cap * 2));
    AS_TRY(unitc.alloc(ctx, 1));
    double cos_delta = cos_delta_bf16x3(d);  // bound between a listed product and the exact cosine
    DevBuf<float> floor_pad, gmb_pad;
    // a large batch is put in the order of the items' image too (the same pivots): positions in the query image -> queries
    DevBuf<uint32_t> qperm;
    // ... and on an index whose items are in their own order a batch goes in the order of its LAMBDAS: a row group of the scan
    // then spans a narrow lambda range, and a lane can rule its column out by the lambda term from two scalars (the cosine alone
    // reaching its floor sent half of all blocks of a 10 k-query batch through the walk; mfma_join_f16.hip)
    const bool wide_pre = scan_wide && ix->scan_perm && ix->rot_state == 1 && ix->packed8 && ix->rot_qw && ctx->opt_search_prefilter;
    DevBuf<double> Zq, zqn;  // wide items behind the first pass: the batch's rotated coordinates
    if (wide_pre) {
      AS_TRY(Zq.alloc(ctx, (size_t)nq * 32));
      AS_TRY(zqn.alloc(ctx, nq));
      AS_TRY(rotated_project_wide(ctx, queries, q_norms, nq, d, ix->rot_qw, Zq.p, zqn.p));
    }
    const bool q_by_pivots = (scan_h && ix->scan_perm && nq >= 2048) || (wide_pre && nq >= 2048);
    const bool q_by_lambda = scan_h && !q_by_pivots && nq >= 1024 && ctx->opt_search_lambda_order && is_device_ptr(q_lambdas);
    const bool q_order = q_by_pivots || q_by_lambda;
    if (q_order) {
      AS_TRY(qperm.alloc(ctx, nq));
      if (q_by_pivots && wide_pre) AS_TRY(order_by_pivots(ctx, Zq.p, nq, 32, ix->scan_piv, qperm.p));
      else if (q_by_pivots) AS_TRY(order_by_pivots(ctx, queries, nq, d, ix->scan_piv, qperm.p));
      else AS_TRY(order_by_value(ctx, q_lambdas, nq, qperm.p));
    }
    ctx->stats["search_scan_lambda_order"] = q_by_lambda ? 1 : 0;
    if (scan_h) {
      AS_TRY(PQh.alloc(ctx, packed_vecs_f16(a_alloc, d) * 16));
      AS_TRY(pack_f16_all(ctx, queries, q_norms, nq, d, PQh.p, a_alloc, q_order ? qperm.p : nullptr));
      const double e = std::max(ctx->f16_err, ix->f16_err);
      cos_delta = (ctx->f16_err >= 0.0 && ix->f16_err >= 0.0) ? std::min(cos_delta_f16(d), 2.0 * e + e * e + cos_delta_f16_accum(d))
                                                                : cos_delta_f16(d);
      AS_TRY(floor_pad.alloc(ctx, padded));
      AS_TRY(gmb_pad.alloc(ctx, padded));
    }
    if (scan_wide) {  // one f16 plane: the measured plane errors of the two images (mfma_join_f16.hip)
      AS_TRY(pack_f16_all(ctx, queries, q_norms, nq, d, PQ.p, a_alloc, q_order ? qperm.p : nullptr));
      const double e = std::max(ctx->f16_err, ix->f16_err);
      cos_delta = (ctx->f16_err >= 0.0 && ix->f16_err >= 0.0) ? std::min(cos_delta_f16(d), 2.0 * e + e * e + cos_delta_f16_accum(d))
                                                                : cos_delta_f16(d);
      AS_TRY(floor_pad.alloc(ctx, padded));
      AS_TRY(gmb_pad.alloc(ctx, padded));
    } else {
      AS_TRY(pack_bf16(ctx, queries, q_norms, nullptr, 0, nq, d, PQ.p, a_alloc));
    }
    hipLaunchKernelGGL(gather_lambda_f32_kernel, dim3(cdiv(padded, 256)), dim3(256), 0, ctx->stream, q_lambdas,
                       (const uint32_t *)nullptr, nq, padded, qlam.p);
    DevBuf<float> qlam_pos;  // ... by position in the query image, for the f16 scan
    if (q_order) {
      AS_TRY(qlam_pos.alloc(ctx, padded));
      hipLaunchKernelGGL(gather_lambda_f32_kernel, dim3(cdiv(padded, 256)), dim3(256), 0, ctx->stream, q_lambdas,
                         (const uint32_t *)qperm.p, nq, padded, qlam_pos.p);
    }
    if (scan_wide && q_order) {  // (the image is in the batch's own order: the guesses come out by position and go back to the queries)
      DevBuf<float> g1, g2;
      AS_TRY(g1.alloc(ctx, padded));
      AS_TRY(g2.alloc(ctx, padded));
      AS_TRY(search_thresholds_f16_wide(ctx, PQ.p, nq, d, ix->items, ix->norms, ix->scan_sample, ns, m, qlam_pos.p, ix->scan_sample_lam,
                                        (float)alpha, (float)beta, g1.p, m2, g2.p));
      AS_TRY(scatter_by_perm(ctx, g1.p, qperm.p, nq, guess.p));
      AS_TRY(scatter_by_perm(ctx, g2.p, qperm.p, nq, guess2.p));
    } else if (scan_wide)
      AS_TRY(search_thresholds_f16_wide(ctx, PQ.p, nq, d, ix->items, ix->norms, ix->scan_sample, ns, m, qlam.p, ix->scan_sample_lam,
                                        (float)alpha, (float)beta, guess.p, m2, guess2.p));
    else
      AS_TRY(search_thresholds_bf16(ctx, PQ.p, nq, d, ix->items, ix->norms, ix->scan_sample, ns, m, qlam.p, ix->scan_sample_lam,
                                    (float)alpha, (float)beta, guess.p, m2, guess2.p));
    hipLaunchKernelGGL(scan_guess_kernel, dim3(cdiv(padded, 256)), dim3(256), 0, ctx->stream, guess.p, (const uint32_t *)nullptr,
                       nq, padded, guess_pad.p);
    AS_HIP(ctx, hipMemsetAsync(cnt.p, 0, sizeof(uint32_t) * padded, ctx->stream));
    AS_HIP(ctx, hipMemsetAsync(unitc.p, 0, sizeof(uint32_t), ctx->stream));
    if (scan_h) {
      hipLaunchKernelGGL(scan_guess_wide_kernel, dim3(cdiv(padded, 256)), dim3(256), 0, ctx->stream, guess.p,
                         (const uint32_t *)(q_order ? qperm.p : nullptr), nq, padded, (float)alpha, (float)beta, floor_pad.p, gmb_pad.p);
      // int8 first pass on the rotated images of the items (kept with the index) and of this batch: which blocks can hold a
      // pair whose cosine alone reaches what the query's guess asks for
      DevBuf<unsigned char> PQ8, mask;
      const bool pre = ix->scan_perm && ix->rot_state == 1 && ctx->opt_search_prefilter;
      if (pre) {
        double eq = 0.0;
        AS_TRY(PQ8.alloc(ctx, rotated_image_bytes(a_alloc)));
        AS_TRY(rotated_image_pack(ctx, queries, q_norms, q_order ? qperm.p : nullptr, nq, d, ix->rot_q, PQ8.p, a_alloc, &eq));
        const double E = std::max(eq, ix->rot_err), delta8 = (2.0 * E + E * E) * (1.0 + 1e-6) + 1e-4;
        const uint32_t T2i = (n + 511) / 512, a_supers = a_alloc / 4;
        AS_TRY(mask.alloc(ctx, (size_t)a_supers * 2 * T2i * 8));
        AS_TRY(prefilter_i8_rect(ctx, ix->packed8, n, PQ8.p, a_supers, floor_pad.p, delta8, mask.p, "search_scan_prefilter"));
      }
      ctx->stats["search_scan_prefilter"] = pre ? 1 : 0;
      ctx->stats["search_scan_order"] = q_by_pivots ? 1 : 0;
      AS_TRY(join_f16_search(ctx, ix->packedh, n, d, PQh.p, nq, floor_pad.p, gmb_pad.p, q_order ? qlam_pos.p : qlam.p, x_lam_h,
                             (float)alpha, (float)beta, cap, cnt.p, cand.p, unitc.p, "search_scan_f16", ix->scan_perm,
                             q_order ? qperm.p : nullptr, pre ? mask.p : nullptr));
    } else if (scan_wide) {
      // (floor / g - beta by position of the query image; padded to whole 512-position super tiles of the first pass: +inf beyond nq)
      const uint32_t a_alloc4 = (a_alloc + 3) & ~3u, padded4 = a_alloc4 * kTileRows;
      DevBuf<float> floor4, gmb4;
      AS_TRY(floor4.alloc(ctx, padded4));
      AS_TRY(gmb4.alloc(ctx, padded4));
      hipLaunchKernelGGL(scan_guess_wide_kernel, dim3(cdiv(padded4, 256)), dim3(256), 0, ctx->stream, guess.p,
                         (const uint32_t *)(q_order ? qperm.p : nullptr), nq, padded4, (float)alpha, (float)beta, floor4.p, gmb4.p);
      DevBuf<unsigned char> PQ8, mask;
      if (wide_pre) {
        double eq = 0.0;
        AS_TRY(PQ8.alloc(ctx, rotated_image_bytes(a_alloc4)));
        AS_TRY(pack_rotated_from_z(ctx, Zq.p, q_order ? qperm.p : nullptr, nq, a_alloc4, PQ8.p, &eq));
        const double E = std::max(eq, ix->rot_err), delta8 = (2.0 * E + E * E) * (1.0 + 1e-6) + 1e-4;
        const uint32_t T2i = (n + 511) / 512, a_supers = a_alloc4 / 4;
        AS_TRY(mask.alloc(ctx, (size_t)a_supers * 2 * T2i * 8));
        AS_TRY(prefilter_i8_rect(ctx, ix->packed8, n, PQ8.p, a_supers, floor4.p, delta8, mask.p, "search_scan_prefilter"));
      }
      ctx->stats["search_scan_prefilter"] = wide_pre ? 1 : 0;
      ctx->stats["search_scan_order"] = (q_by_pivots && wide_pre) ? 1 : 0;
      AS_TRY(join_f16_wide_search(ctx, ix->packedh, n, d, PQ.p, nq, floor4.p, gmb4.p, q_order ? qlam_pos.p : qlam.p, x_lam_h, (float)alpha,
                                  (float)beta, cap, cnt.p, cand.p, "search_scan_f16", ix->scan_perm, q_order ? qperm.p : nullptr,
                                  wide_pre ? mask.p : nullptr));
    } else {
      AS_TRY(join_bf16_search(ctx, ix->packed16, n, d, PQ.p, nq, guess_pad.p, qlam.p, ix->lam32, (float)alpha, (float)beta, cap,
                              cnt.p, cand.p, unitc.p, "search_scan_bf16"));
    }
    {
      // unlisted items: approximate score < g_q.  Approximate vs exact score: alpha x the cosine bound of the pass,
      // the lambda term in f32 with beta folded into its operands (four roundings on numbers of size <= lsum),
      // the f32 roundings of alpha, beta and of the two fused multiply-adds.
      const double lam_err = std::fabs(beta) * (lsum * 0x1p-22 + 0x1p-21);
      const SearchBound bnd{std::fabs(alpha) * cos_delta + lam_err + (std::fabs(alpha) + std::fabs(beta)) * 5.2e-7 + 1e-12};
      ProfScope ps(ctx, "search_rerank");
      // the lists carry the approximate score minus beta: candidates that cannot reach the top K are dropped before
      // their rows are fetched, as in the item graph (prune_candidates)
      const uint32_t kmax = std::min<uint32_t>(cap, std::max<uint32_t>(128, 2 * K));
      const size_t lds128 = (size_t)4 * (1024 + (size_t)8 * kRrPitch + rr_key_bytes(cap, kmax) + (size_t)cap * 4);
      if (d == 128 && lds128 <= 160 * 1024 && ((uintptr_t)ix->items & 15) == 0) {
        AS_HIP(ctx, hipFuncSetAttribute((const void *)rerank_join_d128_kernel<SearchScorePolicy, SearchBound, 8, 1>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds128));
        hipLaunchKernelGGL((rerank_join_d128_kernel<SearchScorePolicy, SearchBound, 8, 1>),
                           dim3(std::min<uint64_t>(cdiv(nq, 4), 256 * 16)), dim3(256), lds128, ctx->stream, ix->items, queries,
                           (const uint32_t *)nullptr, 0u, nq, guess_pad.p, 1u, cnt.p, cand.p, cap, cap, K,
                           (const uint8_t *)nullptr, pol, bnd, bnd.delta, 0, kmax, out_key, out_idx, fail1.p, fail_count.p,
                           err.p);
      } else {
        const size_t lds = (size_t)4 * (d + cap) * sizeof(double) + (size_t)4 * cap * sizeof(uint32_t);
        AS_HIP(ctx, hipFuncSetAttribute((const void *)rerank_join_kernel<SearchScorePolicy, SearchBound>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((rerank_join_kernel<SearchScorePolicy, SearchBound>), dim3(std::min<uint64_t>(cdiv(nq, 4), 8192)),
                           dim3(256), lds, ctx->stream, ix->items, queries, (const uint32_t *)nullptr, d, 0u, nq, guess_pad.p, 1u,
                           cnt.p, cand.p, cap, cap, K, (const uint8_t *)nullptr, pol, bnd, bnd.delta, 0, out_key, out_idx, fail1.p,
                           fail_count.p, err.p);
      }
      AS_HIP(ctx, hipGetLastError());
    }
    {  // statistics: candidates per query
      DevBuf<unsigned long long> st;
      AS_TRY(st.alloc(ctx, 3));
      AS_HIP(ctx, hipMemsetAsync(st.p, 0, 3 * sizeof(unsigned long long), ctx->stream));
      hipLaunchKernelGGL(join_count_stats_kernel, dim3(64), dim3(256), 0, ctx->stream, cnt.p, nq, cap, st.p);
      unsigned long long h[3] = {0, 0, 0};
      AS_HIP(ctx, copy_d2h(ctx, h, st.p, sizeof(h)));
      AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
      ctx->stats["search_scan_candidates"] = (int64_t)h[0];
      ctx->stats["search_scan_max_per_query"] = (int64_t)h[1];
      ctx->stats["search_scan_overflow_queries"] = (int64_t)h[2];
    }
    AS_TRY(read_u32(ctx, fail_count.p, &n_fail));
    ctx->stats["search_scan_uncertified_first"] = n_fail;
    // ---- stage 0b: the queries whose guess was too high, again with the looser one ----
    if (n_fail > 0 && n_fail <= nq / 4 + 16) {
      const uint32_t nf = n_fail;
      const uint32_t bt = (nf + kTileRows - 1) / kTileRows, b_al = scan_h ? (bt + 3) & ~3u : (bt + 1) & ~1u, pad_b = b_al * kTileRows;
      const uint32_t cap2 = 1024;
      DevBuf<unsigned char> PQb;
      DevBuf<float> qlam_b, guess_b;
      DevBuf<uint32_t> cnt_b, cand_b, oi, fail_b, ids_b;
      DevBuf<double> ok;
      AS_TRY(PQb.alloc(ctx, ((scan_wide || scan_h) ? packed_vecs_f16(b_al, d) : packed_vecs_bf16(b_al, d)) * 16));
      AS_TRY(qlam_b.alloc(ctx, pad_b));
      AS_TRY(guess_b.alloc(ctx, pad_b));
      AS_TRY(cnt_b.alloc(ctx, pad_b));
      AS_TRY(cand_b.alloc(ctx, (size_t)pad_b * cap2 * 2));
      AS_TRY(ok.alloc(ctx, (size_t)nf * K));
      AS_TRY(oi.alloc(ctx, (size_t)nf * K));
      AS_TRY(fail_b.alloc(ctx, nf));
      AS_TRY(ids_b.alloc(ctx, nf));
      if (scan_wide || scan_h) AS_TRY(pack_f16(ctx, queries, q_norms, fail1.p, 0, nf, d, PQb.p, b_al));
      else AS_TRY(pack_bf16(ctx, queries, q_norms, fail1.p, 0, nf, d, PQb.p, b_al));
      hipLaunchKernelGGL(gather_lambda_f32_kernel, dim3(cdiv(pad_b, 256)), dim3(256), 0, ctx->stream, q_lambdas, fail1.p, nf, pad_b,
                         qlam_b.p);
      hipLaunchKernelGGL(scan_guess_kernel, dim3(cdiv(pad_b, 256)), dim3(256), 0, ctx->stream, guess2.p, fail1.p, nf, pad_b,
                         guess_b.p);
      AS_HIP(ctx, hipMemsetAsync(cnt_b.p, 0, sizeof(uint32_t) * pad_b, ctx->stream));
      AS_HIP(ctx, hipMemsetAsync(unitc.p, 0, sizeof(uint32_t), ctx->stream));
      AS_HIP(ctx, hipMemsetAsync(fail_count.p, 0, sizeof(uint32_t), ctx->stream));
      if (scan_wide || scan_h) {
        DevBuf<float> floor_b, gmb_b;
        AS_TRY(floor_b.alloc(ctx, pad_b));
        AS_TRY(gmb_b.alloc(ctx, pad_b));
        hipLaunchKernelGGL(scan_guess_wide_kernel, dim3(cdiv(pad_b, 256)), dim3(256), 0, ctx->stream, guess2.p, fail1.p, nf, pad_b,
                           (float)alpha, (float)beta, floor_b.p, gmb_b.p);
        if (scan_h)
          AS_TRY(join_f16_search(ctx, ix->packedh, n, d, PQb.p, nf, floor_b.p, gmb_b.p, qlam_b.p, x_lam_h, (float)alpha, (float)beta, cap2,
                                 cnt_b.p, cand_b.p, unitc.p, "search_scan_f16", ix->scan_perm));
        else
          AS_TRY(join_f16_wide_search(ctx, ix->packedh, n, d, PQb.p, nf, floor_b.p, gmb_b.p, qlam_b.p, x_lam_h, (float)alpha, (float)beta,
                                    cap2, cnt_b.p, cand_b.p, "search_scan_f16", ix->scan_perm));
      } else {
        AS_TRY(join_bf16_search(ctx, ix->packed16, n, d, PQb.p, nf, guess_b.p, qlam_b.p, ix->lam32, (float)alpha, (float)beta, cap2,
                                cnt_b.p, cand_b.p, unitc.p, "search_scan_bf16"));
      }
      {
        const double lam_err = std::fabs(beta) * (lsum * 0x1p-22 + 0x1p-21);
        const SearchBound bnd{std::fabs(alpha) * cos_delta + lam_err + (std::fabs(alpha) + std::fabs(beta)) * 5.2e-7 + 1e-12};
        ProfScope ps(ctx, "search_rerank");
        const size_t lds = (size_t)4 * (d + cap2) * sizeof(double) + (size_t)4 * cap2 * sizeof(uint32_t);
        AS_HIP(ctx, hipFuncSetAttribute((const void *)rerank_join_kernel<SearchScorePolicy, SearchBound>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((rerank_join_kernel<SearchScorePolicy, SearchBound>), dim3(std::min<uint64_t>(cdiv(nf, 4), 8192)),
                           dim3(256), lds, ctx->stream, ix->items, queries, (const uint32_t *)fail1.p, d, 0u, nf, guess_b.p, 1u,
                           cnt_b.p, cand_b.p, cap2, cap2, K, (const uint8_t *)nullptr, pol, bnd, bnd.delta, 0, ok.p, oi.p, fail_b.p,
                           fail_count.p, err.p);
        AS_HIP(ctx, hipGetLastError());
      }
      hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv((uint64_t)nf * K, 256)), dim3(256), 0, ctx->stream, fail1.p, nf, K, ok.p,
                         oi.p, out_key, out_idx);
      uint32_t nf2 = 0;
      AS_TRY(read_u32(ctx, fail_count.p, &nf2));
      if (nf2) {  // what is still open: positions in the gathered list -> query ids, back into fail1
        hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(nf2, 256)), dim3(256), 0, ctx->stream, fail1.p, fail_b.p, nf2, 0u, ids_b.p);
        AS_HIP(ctx, hipMemcpyAsync(fail1.p, ids_b.p, sizeof(uint32_t) * nf2, hipMemcpyDeviceToDevice, ctx->stream));
        AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
      }
      n_fail = nf2;
    }
  } else {
    const uint32_t a_tiles = (nq + kTileRows - 1) / kTileRows;
    const uint32_t a_items = use_f32 ? (a_tiles + 1) / 2 : a_tiles;
    pb.splits = clamp_splits(pick_splits_scan(a_items, b_tiles, n), d, KC);
    DevBuf<double> ckey;
    DevBuf<uint32_t> cidx;
    AS_TRY(ckey.alloc(ctx, (size_t)nq * pb.splits * KC));
    AS_TRY(cidx.alloc(ctx, (size_t)nq * pb.splits * KC));
    pb.cand_key = ckey.p;
    pb.cand_idx = cidx.p;
    if (use_f32) {
      AS_TRY(ensure32());
      const uint32_t at = (a_tiles + 1) & ~1u;
      DevBuf<float> PA;
      AS_TRY(PA.alloc(ctx, packed_elems(at, d)));
      AS_TRY(pack_f32(ctx, queries, q_norms, nullptr, 0, nq, d, PA.p, at));
      // lambda term of the epilogue: f32 when the lambdas are small enough for its error to stay far below
      // the cosine error (the usual case: lambda in [0, 1]), f64 otherwise (mfma_topk_f32.hip)
      DevBuf<unsigned long long> amax;
      AS_TRY(amax.alloc(ctx, 2));
      AS_HIP(ctx, hipMemsetAsync(amax.p, 0, 2 * sizeof(unsigned long long), ctx->stream));
      hipLaunchKernelGGL(absmax_kernel, dim3(std::min<uint64_t>(1024, cdiv(nq, 256))), dim3(256), 0, ctx->stream,
                         q_lambdas, (uint64_t)nq, amax.p);
      hipLaunchKernelGGL(absmax_kernel, dim3(std::min<uint64_t>(1024, cdiv(n, 256))), dim3(256), 0, ctx->stream,
                         ix->lambdas, (uint64_t)n, amax.p + 1);
      AS_HIP(ctx, hipGetLastError());
      double hmax[2] = {0.0, 0.0};
      AS_HIP(ctx, copy_d2h(ctx, hmax, amax.p, sizeof(hmax)));
      const double lsum = hmax[0] + hmax[1];
      pb.lam_f32 = lsum <= 64.0 ? 1 : 0;
      ctx->stats["search_lambda_term_f32"] = pb.lam_f32;
      AS_TRY(candidates_f32(ctx, pb, PA.p, a_tiles, ix->packed32, b_tiles, "search_mfma_candidates_f32"));
      // f32 pass: cosine error scaled by |alpha|, the lambda term's error scaled by |beta| (f32 form only),
      // plus the f32 roundings of a sum bounded by |alpha| + |beta| (product, fma)
      const double beta_abs = std::fabs(1.0 - alpha);
      const double lam_err = pb.lam_f32 ? beta_abs * (lsum * 0x1p-23 + 0x1p-22) : 0.0;
      SearchBound bnd{std::fabs(alpha) * cos_delta_f32(d) + lam_err + (std::fabs(alpha) + beta_abs) * 2.6e-7 + 1e-12};
      AS_TRY(launch_rerank(ctx, "search_rerank", queries, ix->items, d, nullptr, 0u, nq, 0u, K, KC, pb.splits, ckey.p,
                           cidx.p, (const uint8_t *)nullptr, pol, bnd, out_key, out_idx, fail1.p, fail_count.p, err.p));
    } else {
      AS_TRY(ensure64());
      DevBuf<double> PA;
      AS_TRY(PA.alloc(ctx, packed_elems(a_tiles, d)));
      AS_TRY(pack_f64(ctx, queries, q_norms, nullptr, 0, nq, d, PA.p, a_tiles));
      AS_TRY(candidates_f64(ctx, pb, PA.p, a_tiles, ix->packed, b_tiles, "search_mfma_candidates"));
      SearchBound bnd{std::fabs(alpha) * cos_delta_f64(d) + 1e-15};
      AS_TRY(launch_rerank(ctx, "search_rerank", queries, ix->items, d, nullptr, 0u, nq, 0u, K, KC, pb.splits, ckey.p,
                           cidx.p, (const uint8_t *)nullptr, pol, bnd, out_key, out_idx, fail1.p, fail_count.p, err.p));
    }
    AS_TRY(read_u32(ctx, fail_count.p, &n_fail));
  }
  ctx->stats["search_uncertified_rows"] = n_fail;
  ctx->stats["search_uncertified_rows_stage2"] = 0;
  ctx->stats["search_cand_capacity"] = KC;
  ctx->stats["search_col_splits"] = pb.splits;

  const uint32_t *exact_ids = fail1.p;
  uint32_t n_exact = n_fail;
  DevBuf<uint32_t> fail2, ids2;
  if (mfma_ok && (use_f32 || scan) && !wide && n_fail > 0 && !exact_direct(n_fail, n)) {
    const uint32_t a_tiles = (n_fail + kTileRows - 1) / kTileRows;
    pb.arows = ARows{fail1.p, 0};
    pb.nA = n_fail;
    pb.splits = clamp_splits(pick_splits(a_tiles, b_tiles), d, KC);
    DevBuf<double> ckey, PA, ok;
    DevBuf<uint32_t> cidx, oi;
    AS_TRY(ckey.alloc(ctx, (size_t)n_fail * pb.splits * KC));
    AS_TRY(cidx.alloc(ctx, (size_t)n_fail * pb.splits * KC));
    pb.cand_key = ckey.p;
    pb.cand_idx = cidx.p;
    AS_TRY(ensure64());
    AS_TRY(PA.alloc(ctx, packed_elems(a_tiles, d)));
    AS_TRY(pack_f64(ctx, queries, q_norms, fail1.p, 0, n_fail, d, PA.p, a_tiles));
    AS_TRY(candidates_f64(ctx, pb, PA.p, a_tiles, ix->packed, b_tiles, "search_mfma_candidates"));
    AS_TRY(ok.alloc(ctx, (size_t)n_fail * K));
    AS_TRY(oi.alloc(ctx, (size_t)n_fail * K));
    AS_TRY(fail2.alloc(ctx, n_fail));
    AS_HIP(ctx, hipMemsetAsync(fail_count.p, 0, sizeof(uint32_t), ctx->stream));
    SearchBound bnd{std::fabs(alpha) * cos_delta_f64(d) + 1e-15};
    AS_TRY(launch_rerank(ctx, "search_rerank", queries, ix->items, d, fail1.p, 0u, n_fail, 0u, K, KC, pb.splits, ckey.p,
                         cidx.p, (const uint8_t *)nullptr, pol, bnd, ok.p, oi.p, fail2.p, fail_count.p, err.p));
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv((uint64_t)n_fail * K, 256)), dim3(256), 0, ctx->stream, fail1.p,
                       n_fail, K, ok.p, oi.p, out_key, out_idx);
    uint32_t n_fail2 = 0;
    AS_TRY(read_u32(ctx, fail_count.p, &n_fail2));
    ctx->stats["search_uncertified_rows_stage2"] = n_fail2;
    n_exact = n_fail2;
    if (n_fail2) {
      AS_TRY(ids2.alloc(ctx, n_fail2));
      hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(n_fail2, 256)), dim3(256), 0, ctx->stream, fail1.p, fail2.p,
                         n_fail2, 0u, ids2.p);
      exact_ids = ids2.p;
    }
  }
  int h_err = 0;
  AS_HIP(ctx, copy_d2h(ctx, &h_err, err.p, sizeof(int)));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (h_err) return fail(ctx, AS_ERR_NAN_SCORE, as_status_message(AS_ERR_NAN_SCORE));
  if (n_exact) {
    DevBuf<double> fk;
    DevBuf<uint32_t> fi;
    AS_TRY(fk.alloc(ctx, (size_t)n_exact * K));
    AS_TRY(fi.alloc(ctx, (size_t)n_exact * K));
    AS_TRY(search_exact(ctx, ix, queries, q_norms, q_lambdas, exact_ids, n_exact, nq, K, alpha, fi.p, fk.p));
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv((uint64_t)n_exact * K, 256)), dim3(256), 0, ctx->stream,
                       exact_ids, n_exact, K, fk.p, fi.p, out_key, out_idx);
    AS_HIP(ctx, hipGetLastError());
  }
  return AS_OK;
}
#undef AS_IX_BUILD

}  // namespace asx
