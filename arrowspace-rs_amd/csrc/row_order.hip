// row_order.hip -- an order of the rows in which neighbours sit together, for the OPERAND IMAGE of the symmetric join only.
//
// The join (mfma_join_f16.hip) contracts every pair of row tiles whatever the order of the rows, but what happens AFTER the
// MFMAs of a 64 x 32 block depends on it: a block that holds a pair above a threshold goes through the element-wise test
// and the listing, and with neighbours scattered over the whole image (the bench's generator deals the mixture components
// round-robin: row i belongs to component i mod 256) a quarter of all blocks hold one.  With the rows grouped by direction
// the same pairs fall into few, dense blocks: measured at 1 M x 128 with the rows sorted by their true component the join
// drops from 110 to 97 ms.  So the image is packed in a permuted order -- rows assigned to the nearest (by cosine) of 256
// rows sampled as pivots, stable counting sort by pivot -- and the kernel translates positions back to row ids where it
// appends to the lists.  Nothing else sees the permutation: thresholds are scattered back to row order, the re-rank, its
// tie-breaks by row id and every later stage work on the rows as given.  Any permutation is correct; this one is cheap
// (about 2 ms at 1 M x 128) and needs no clustering from the caller.
#include <vector>
#include <algorithm>

#include "common.h"

namespace asx {

namespace {

constexpr int kPivots = 256;      // pivots of an order found on the rows themselves
constexpr int kPivotsMax = 4096;  // ... on 32 rotated coordinates (a quarter of the arithmetic per pivot: csrc/rotate_wide.hip), where
                                  // 1024 instead of 256 pivots leave the f16 pass of the join half the blocks; beyond 1024 (the
                                  // assignment then runs in passes of 1024 pivots) where the data has more clusters than that
constexpr int kPivotsPass = 1024;  // pivots of one pass of the assignment (64 KB of fragments in LDS) = the most of a second set
constexpr int kPivotChunk = 32;  // pivots per LDS pass

// unit pivots, f32, P x dp (dp = d rounded up to 4, zero padded)
__global__ void gather_pivots_kernel(const double *__restrict__ rows, const double *__restrict__ norms, const uint32_t *__restrict__ ids,
                                     uint32_t d, uint32_t dp, float *__restrict__ piv) {
  const uint32_t p = blockIdx.x;
  const uint32_t r = ids[p];
  const double nr = norms[r];
  for (uint32_t k = threadIdx.x; k < dp; k += blockDim.x)
    piv[(size_t)p * dp + k] = (k < d && nr > 0.0 && nr < __builtin_inf()) ? (float)(rows[(size_t)r * d + k] / nr) : 0.f;
}

// pivot of every row: the one with the largest dot product (the row's own norm does not matter for the argmax; rows with a
// zero or non-finite norm go to pivot 0).  One thread per row, the row in registers, pivots through LDS (broadcast reads).
template <int DP>  // padded width, a multiple of 4, <= 128
__global__ __launch_bounds__(256) void assign_pivots_kernel(const double *__restrict__ rows, uint32_t n, uint32_t d,
                                                            const float *__restrict__ piv, uint32_t *__restrict__ assign, int np) {
  __shared__ float s_piv[kPivotChunk * DP];
  const uint32_t row = blockIdx.x * 256 + threadIdx.x;
  float x[DP];
#pragma unroll
  for (int k = 0; k < DP; ++k) x[k] = (row < n && (uint32_t)k < d) ? (float)rows[(size_t)row * d + k] : 0.f;
  float best = -__builtin_inff();
  uint32_t arg = 0;
  for (int p0 = 0; p0 < np; p0 += kPivotChunk) {
    __syncthreads();
    for (int e = threadIdx.x; e < kPivotChunk * DP; e += 256) s_piv[e] = piv[(size_t)p0 * DP + e];
    __syncthreads();
    for (int p = 0; p < kPivotChunk; ++p) {
      float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
#pragma unroll
      for (int k = 0; k < DP; k += 4) {
        const float4 q = *reinterpret_cast<const float4 *>(&s_piv[p * DP + k]);
        acc0 = __builtin_fmaf(x[k], q.x, acc0);
        acc1 = __builtin_fmaf(x[k + 1], q.y, acc1);
        acc2 = __builtin_fmaf(x[k + 2], q.z, acc2);
        acc3 = __builtin_fmaf(x[k + 3], q.w, acc3);
      }
      const float s = (acc0 + acc1) + (acc2 + acc3);
      if (s > best) {
        best = s;
        arg = (uint32_t)(p0 + p);
      }
    }
  }
  if (row < n) assign[row] = arg;
}

// The same argmax for the 32 rotated coordinates on the matrix pipe (round 3): a 1 M x 32 . 32 x 1024 product is 6.6e10 flop, 1.5 ms
// of f32 vector work above, and a rounding of the operands to f16 moves an assignment only between pivots whose products lie
// within 1e-3 of each other -- any assignment gives a correct image order, and this one is as deterministic as the other.
// A wave owns 32 rows (A fragments: 2 x 8 halves per lane, straight from the f64 coordinates), the np pivots sit in LDS as f16
// B fragments, two v_mfma_f32_32x32x16_f16 per 32 pivots; every lane keeps the best (value, pivot) of its 16 rows over the
// pivots of its column, in pivot order with a strict '>', and the 32 columns meet at the end (lower pivot on equal values).
typedef float ro_v16f32 __attribute__((ext_vector_type(16)));
typedef _Float16 ro_f16x8 __attribute__((ext_vector_type(8)));
// More than 1024 pivots: passes of up to 1024 (piv = the pass's first pivot, p_base its index), the best key of the passes so far
// carried in key_io (flags: 1 = read it, 2 = write it, 4 = this pass is the last and writes the assignment); `bits` low bits of the
// key hold the pivot (10 up to 1024 pivots -- the assignment round 4 measured with -- 12 beyond).
__global__ __launch_bounds__(256) void assign_pivots_mfma_kernel(const double *__restrict__ Z, uint32_t n, const float *__restrict__ piv,
                                                                 uint32_t np, uint32_t *__restrict__ assign, uint32_t p_base = 0,
                                                                 uint32_t bits = 10, int *__restrict__ key_io = nullptr, uint32_t flags = 4) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ro_smem[];
  ro_f16x8 *s_b = reinterpret_cast<ro_f16x8 *>(ro_smem);  // [np / 32 tiles][2 k steps][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, lc = lane & 31, lh = lane >> 5;
  for (uint32_t e = tid; e < (np / 32) * 2 * 64; e += 256) {
    const uint32_t l = e & 63, ks = (e >> 6) & 1, t = e >> 7;
    const float *src = piv + (size_t)(t * 32 + (l & 31)) * 32 + 16 * ks + 8 * (l >> 5);
    ro_f16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (_Float16)src[j];
    s_b[e] = v;
  }
  __syncthreads();
  // (a workgroup walks chunks of 128 rows: the 64 KB of pivot fragments above are built once per workgroup, not once per 128 rows
  // -- with a workgroup per chunk the setup was most of the 0.32 ms this took at 1 M rows)
  for (uint32_t chunk = blockIdx.x; chunk * 128u < n; chunk += gridDim.x) {
  const uint32_t row0 = (chunk * 4 + (tid >> 6)) * 32;
  if (row0 >= n) continue;
  ro_f16x8 a[2];
  {
    const uint32_t row = min(row0 + (uint32_t)lc, n - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const double *src = Z + (size_t)row * 32 + 16 * ks + 8 * lh;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = (float)src[j];
        a[ks][j] = (_Float16)(x == x ? x : 0.f);
      }
    }
  }
  // best (value, pivot) as ONE integer: the product's bits with the low ten replaced by 1023 - pivot (np <= 1024).  Integer order is
  // float order for positive values, a larger key wins, an equal value goes to the lower pivot; the ten bits given up (1.2e-4
  // relative) are below what the f16 operands resolve, and a row whose products are all negative gets SOME pivot -- any assignment
  // gives a correct order.  Two instructions per product instead of a compare and two selects (the loop was bound by them).
  int best[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) best[r] = (int)0x80000000u;
  const uint32_t inv_max = (1u << bits) - 1u;
  ro_v16f32 zero;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;
  for (uint32_t t = 0; t < np / 32; ++t) {
    const ro_f16x8 b0 = s_b[(t * 2 + 0) * 64 + lane], b1 = s_b[(t * 2 + 1) * 64 + lane];
    ro_v16f32 x = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b0, zero, 0, 0, 0);
    x = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b1, x, 0, 0, 0);
    const uint32_t inv = inv_max - (p_base + t * 32 + (uint32_t)lc);
#pragma unroll
    for (int r = 0; r < 16; ++r) best[r] = max(best[r], (int)((__float_as_uint(x[r]) & ~inv_max) | inv));
  }
  // accumulator register r of lane (lh, lc) = row 4 lh + (r & 3) + 8 (r >> 2) of the tile, column lc: the columns meet here
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int bv = best[r];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) bv = max(bv, __shfl_xor(bv, o, 64));
    const uint32_t row = row0 + 4 * lh + (r & 3) + 8 * (r >> 2);
    if (lc == 0 && row < n) {
      if (flags & 1u) bv = max(bv, key_io[row]);
      if (flags & 2u) key_io[row] = bv;
      if (flags & 4u) assign[row] = inv_max - ((uint32_t)bv & inv_max);
    }
  }
  }
}

// stable counting sort by pivot, blocks of 256 rows: counts[c * n_blocks + b] -> (exclusive scan) -> positions
__global__ __launch_bounds__(256) void order_count_kernel(const uint32_t *__restrict__ assign, uint32_t n, uint32_t n_blocks,
                                                          uint32_t *__restrict__ counts, int np) {
  __shared__ uint32_t s_cnt[kPivotsMax];
  for (int c = threadIdx.x; c < np; c += 256) s_cnt[c] = 0;
  __syncthreads();
  const uint32_t row = blockIdx.x * 256 + threadIdx.x;
  if (row < n) atomicAdd(&s_cnt[assign[row]], 1u);
  __syncthreads();
  for (int c = threadIdx.x; c < np; c += 256) counts[(size_t)c * n_blocks + blockIdx.x] = s_cnt[c];
}
__global__ __launch_bounds__(256) void order_place_kernel(const uint32_t *__restrict__ assign, uint32_t n, uint32_t n_blocks,
                                                          const uint64_t *__restrict__ offsets, uint32_t *__restrict__ perm) {
  __shared__ uint32_t s_c[256];
  const uint32_t row = blockIdx.x * 256 + threadIdx.x;
  const uint32_t c = row < n ? assign[row] : 0xFFFFFFFFu;
  s_c[threadIdx.x] = c;
  __syncthreads();
  if (row >= n) return;
  uint32_t rank = 0;  // earlier rows of this block with the same pivot: keeps the sort stable (and deterministic)
  for (uint32_t t = 0; t < threadIdx.x; ++t) rank += s_c[t] == c ? 1u : 0u;
  perm[offsets[(size_t)c * n_blocks + blockIdx.x] + rank] = row;
}

__global__ void scatter_f32_by_perm_kernel(const float *__restrict__ in, const uint32_t *__restrict__ perm, uint32_t n,
                                           float *__restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[perm[i]] = in[i];
}

// Inside one pivot's group (workgroup g: places [start[g], start[g + 1]) of perm) the rows are put in the order of their key of
// the second pivot set: a counting sort in LDS, stable and free of atomics whose order could vary (chunks of 256 places in
// turn: a row's rank among the earlier rows of its chunk with the same key + the key's running cursor).  Groups beyond kSubMax
// rows stay as they are.
constexpr uint32_t kSubMax = 12288;
__global__ __launch_bounds__(256) void order_sub_kernel(const uint32_t *__restrict__ group_start, const uint32_t *__restrict__ key_of_row, int np,
                                                        uint32_t *__restrict__ perm) {
  __shared__ uint32_t s_cur[kPivotsPass];
  __shared__ uint32_t s_rows[kSubMax];
  __shared__ uint32_t s_key[256];
  __shared__ uint32_t s_scan[256];
  const uint32_t lo = group_start[blockIdx.x], hi = group_start[blockIdx.x + 1], len = hi - lo;
  if (len < 2 || len > kSubMax) return;
  for (int c = threadIdx.x; c < np; c += 256) s_cur[c] = 0;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < len; i += 256) atomicAdd(&s_cur[key_of_row[perm[lo + i]]], 1u);  // (a sum: any order)
  __syncthreads();
  // exclusive scan of the np counts (np <= kPivotsPass: four per thread)
  uint32_t mine[4], tot = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = threadIdx.x * 4 + q;
    mine[q] = c < np ? s_cur[c] : 0u;
    tot += mine[q];
  }
  s_scan[threadIdx.x] = tot;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const uint32_t v = threadIdx.x >= (uint32_t)o ? s_scan[threadIdx.x - o] : 0u;
    __syncthreads();
    s_scan[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t base = s_scan[threadIdx.x] - tot;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = threadIdx.x * 4 + q;
    if (c < np) s_cur[c] = base;
    base += mine[q];
  }
  __syncthreads();
  for (uint32_t i0 = 0; i0 < len; i0 += 256) {
    const uint32_t i = i0 + threadIdx.x;
    const uint32_t row = i < len ? perm[lo + i] : 0u;
    const uint32_t k = i < len ? key_of_row[row] : 0xFFFFFFFFu;
    s_key[threadIdx.x] = k;
    __syncthreads();
    uint32_t rank = 0, same = 0;
    for (uint32_t t = 0; t < 256; ++t) {
      const bool eq = s_key[t] == k;
      rank += (eq && t < threadIdx.x) ? 1u : 0u;
      same += eq ? 1u : 0u;
    }
    if (i < len) s_rows[s_cur[k] + rank] = row;
    __syncthreads();
    if (i < len && rank == 0) s_cur[k] += same;  // (the first row of the chunk with this key moves its cursor on)
    __syncthreads();
  }
  for (uint32_t i = threadIdx.x; i < len; i += 256) perm[lo + i] = s_rows[i];
}

__global__ void shift_ids_kernel(uint32_t *__restrict__ ids, uint32_t np, uint32_t by) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < np) ids[p] += by;
}
// The pivots in the order of a greedy chain (round 5): from pivot 0, always on to the most similar pivot not yet taken (ties: the lower
// index).  The groups of the image are laid out in the order of their pivots, and the sampled pivots come in the order of the rows
// they were drawn from -- so the four groups a cluster of the bench mixture falls into sat anywhere in the image, and with them the
// rows' neighbours: tile pairs that hold a pair lay all over the square.  Along the chain the groups of one cluster are adjacent and
// similar clusters follow each other: the band thresholds see more of a row's true neighbourhood, the tile pairs the cone bound keeps
// sit near the diagonal, and what the exact re-rank and the item-graph lambda gather is closer in the caches.  Any order is correct;
// this one is a function of the pivots alone (the same on every rank).
// Two kernels: the 32 most similar pivots of every pivot (a workgroup each), then ONE wave walks the chain through those lists held in
// LDS -- a step is a list read, a look at the taken flags and a ballot (~0.1 us; the whole-set argmax per step it replaces took 2 us
// of one workgroup's vector issue, 2.3 ms for 1 024 pivots).  Where all 32 of a pivot's list are taken the chain goes on at the
// lowest pivot not yet taken: late in the walk, and any order is correct.
constexpr uint32_t kChainList = 32;  // (16 beyond 2 048 pivots: the lists of the walk stay in 128 KB of LDS)
__global__ __launch_bounds__(256) void pivot_near_lists_kernel(const float *__restrict__ piv, uint32_t dpv, uint32_t np, uint16_t *__restrict__ lists,
                                                               uint32_t list_len) {
  __shared__ float s_sim[kPivotsMax];
  __shared__ float s_best[256];
  __shared__ uint32_t s_at[256];
  const uint32_t g = blockIdx.x;
  for (uint32_t h = threadIdx.x; h < np; h += 256) {
    float a = 0.f;
#pragma unroll
    for (uint32_t k = 0; k < 32; ++k) a = __builtin_fmaf(piv[(size_t)g * dpv + k], piv[(size_t)h * dpv + k], a);
    s_sim[h] = (h == g || !(a == a)) ? -__builtin_inff() : a;
  }
  __syncthreads();
  for (uint32_t k = 0; k < list_len; ++k) {
    float best = -__builtin_inff();
    uint32_t at = 0xFFFFFFFFu;
    for (uint32_t h = threadIdx.x; h < np; h += 256)
      if (s_sim[h] > best) {
        best = s_sim[h];
        at = h;
      }
    s_best[threadIdx.x] = best;
    s_at[threadIdx.x] = at;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o && (s_best[threadIdx.x + o] > s_best[threadIdx.x] ||
                                   (s_best[threadIdx.x + o] == s_best[threadIdx.x] && s_at[threadIdx.x + o] < s_at[threadIdx.x]))) {
        s_best[threadIdx.x] = s_best[threadIdx.x + o];
        s_at[threadIdx.x] = s_at[threadIdx.x + o];
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      lists[(size_t)g * list_len + k] = s_at[0] == 0xFFFFFFFFu ? (uint16_t)0xFFFFu : (uint16_t)s_at[0];
      if (s_at[0] != 0xFFFFFFFFu) s_sim[s_at[0]] = -__builtin_inff();
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(64) void pivot_chain_kernel(const uint16_t *__restrict__ lists, uint32_t np, uint32_t *__restrict__ order,
                                                         uint32_t list_len) {
  extern __shared__ unsigned char smem[];
  uint16_t *s_list = reinterpret_cast<uint16_t *>(smem);                     // np x list_len
  unsigned char *s_taken = smem + (size_t)np * list_len * sizeof(uint16_t);  // np
  const uint32_t lane = threadIdx.x;
  for (uint32_t i = lane; i < np * list_len; i += 64) s_list[i] = lists[i];
  for (uint32_t i = lane; i < np; i += 64) s_taken[i] = i == 0 ? 1 : 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  uint32_t cur = 0, low = 1;  // low: every pivot below it is taken
  if (lane == 0) order[0] = 0;
  for (uint32_t step = 1; step < np; ++step) {
    const uint32_t c = lane < list_len ? s_list[cur * list_len + lane] : 0xFFFFu;
    const bool ok = c < np && s_taken[c] == 0;
    const unsigned long long m = __ballot(ok);
    uint32_t nxt;
    if (m) {
      nxt = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)__builtin_ctzll(m));
    } else {
      while (low < np && s_taken[low]) ++low;  // (wave-uniform: every lane reads the same flag)
      nxt = low < np ? low : 0u;
    }
    if (lane == 0) {
      s_taken[nxt] = 1;
      order[step] = nxt;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    cur = nxt;
  }
}
// inv[order[p]] = p; near16[p][0] = p, near16[p][k] = inv[lists[order[p]][k - 1]]: the 16 nearest groups of every group in the
// order of the chain (what the grouped thresholds ask for: mfma_theta_f16.hip, near_groups_kernel), from the chain's own lists
__global__ void chain_inverse_kernel(const uint32_t *__restrict__ order, uint32_t np, uint32_t *__restrict__ inv) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < np) inv[order[p]] = p;
}
__global__ void chain_near16_kernel(const uint32_t *__restrict__ order, const uint32_t *__restrict__ inv, const uint16_t *__restrict__ lists,
                                    uint32_t list_len, uint32_t np, uint32_t *__restrict__ near16) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= np * 16u) return;
  const uint32_t p = t >> 4, k = t & 15u;
  uint32_t v = p;
  if (k) {
    const uint32_t o = lists[(size_t)order[p] * list_len + (k - 1u)];
    v = o < np ? inv[o] : 0xFFFFFFFFu;
  }
  near16[t] = v;
}
// The assignment behind rounds of further pivots without another pass over the pivots: a row's pivot is the better of its best
// of the first set (key_a, `bits_a` pivot bits) and its best of the drawn ones (key_b, 12 pivot bits), both numbered as the pivots
// lay BEFORE the last chain; inv: where each went.  key_out: the same as ONE key with `bits_out` pivot bits (the product's low bits
// are the pivot's: a tie of the two within them goes to the first set -- any assignment gives a correct order).
__global__ void assign_remap_kernel(const int *__restrict__ key_a, uint32_t bits_a, const int *__restrict__ key_b, const uint32_t *__restrict__ inv,
                                    uint32_t n, uint32_t bits_out, uint32_t *__restrict__ assign, int *__restrict__ key_out) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const uint32_t ma = (1u << bits_a) - 1u, mb = 4095u, mo = (1u << bits_out) - 1u;
  const int ka = key_a[r], kb = key_b[r];
  const int va = (int)((uint32_t)ka & ~ma), vb = (int)((uint32_t)kb & ~mb);  // (integer order is float order for the positive products)
  const bool second = vb > va;
  const uint32_t old = second ? mb - ((uint32_t)kb & mb) : ma - ((uint32_t)ka & ma);
  const uint32_t now = inv[old];
  assign[r] = now;
  key_out[r] = (int)(((uint32_t)(second ? vb : va) & ~mo) | (mo - now));
}
// how many times the chain of the pivots steps from one cluster into another: pivots g, g + 1 whose cosine lies below `cut`
__global__ void chain_breaks_kernel(const float *__restrict__ piv, uint32_t dpv, uint32_t np, float cut, uint32_t *__restrict__ breaks) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  bool brk = false;
  if (g + 1 < np) {
    float a = 0.f;
    for (int k = 0; k < 32; ++k) a = __builtin_fmaf(piv[(size_t)g * dpv + k], piv[(size_t)(g + 1) * dpv + k], a);
    brk = !(a >= cut);
  }
  const unsigned long long m = __ballot(brk);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(breaks, (uint32_t)__builtin_popcountll(m));
}
__global__ void pivot_permute_kernel(const float *__restrict__ piv, const uint32_t *__restrict__ order, uint32_t dpv, float *__restrict__ out) {
  const uint32_t p = blockIdx.x;
  for (uint32_t k = threadIdx.x; k < dpv; k += blockDim.x) out[(size_t)p * dpv + k] = piv[(size_t)order[p] * dpv + k];
}

__global__ void order_pivot_ids_kernel(uint32_t n, uint32_t *__restrict__ ids, uint32_t np) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= np) return;
  uint64_t off = (uint64_t)p * 2654435761ull;
  if (np > (uint32_t)kPivotsPass) {  // (the multiplier alone walks n / np in a short cycle: with rows dealt round-robin over C clusters,
    uint32_t x = p * 0x9E3779B9u;    // 4 096 pivots fell into a third of 2 048 clusters.  Up to 1 024 pivots it stays as measured.)
    x ^= x >> 15;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    off = x;
  }
  ids[p] = (uint32_t)(((uint64_t)p * n) / np + off % (n / np));
}

}  // namespace

// Of the rows drawn in a round (unit vectors, c x pitch), those that lie within `cut` of an EARLIER one are dropped -- two rows of one
// cluster would spend two pivots on it -- and the first `want` of the others (their count rounded down to whole 32s) are kept, in
// the order they were drawn.  Twice as many rows are drawn as pivots are wanted: 1 024 draws into 3 000 clusters hit 870 different
// ones, 2 048 draws give the 1 024.  keep[j] then out[..] / n_out.
__global__ __launch_bounds__(256) void cover_distinct_kernel(const float *__restrict__ cand, uint32_t pitch, uint32_t c, float cut,
                                                             uint32_t *__restrict__ keep) {
  __shared__ int s_near;
  const uint32_t j = blockIdx.x;
  if (threadIdx.x == 0) s_near = 0;
  __syncthreads();
  float x[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) x[k] = cand[(size_t)j * pitch + k];
  bool near = false;
  for (uint32_t i = threadIdx.x; i < j && i < c; i += 256) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) a = __builtin_fmaf(x[k], cand[(size_t)i * pitch + k], a);
    near |= a > cut;
  }
  if (near) s_near = 1;
  __syncthreads();
  if (threadIdx.x == 0) keep[j] = s_near ? 0u : 1u;
}
__global__ __launch_bounds__(1024) void cover_keep_kernel(const float *__restrict__ cand, uint32_t pitch, uint32_t c, const uint32_t *__restrict__ keep,
                                                          uint32_t want, float *__restrict__ out, uint32_t *__restrict__ n_out) {
  __shared__ uint32_t s_scan[1024];
  __shared__ uint32_t s_base;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  uint32_t total = 0;
  for (uint32_t j0 = 0; j0 < c; j0 += 1024) {  // first the count, so that the number kept is whole 32s
    const uint32_t j = j0 + threadIdx.x;
    const uint32_t k = j < c ? keep[j] : 0u;
    total += __syncthreads_count(k != 0);
  }
  const uint32_t take = min(want, total) / 32u * 32u;
  for (uint32_t j0 = 0; j0 < c; j0 += 1024) {
    const uint32_t j = j0 + threadIdx.x;
    const uint32_t k = j < c ? keep[j] : 0u;
    s_scan[threadIdx.x] = k;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const uint32_t v = threadIdx.x >= (uint32_t)o ? s_scan[threadIdx.x - o] : 0u;
      __syncthreads();
      s_scan[threadIdx.x] += v;
      __syncthreads();
    }
    const uint32_t at = s_base + s_scan[threadIdx.x] - k;
    if (k && at < take)
      for (uint32_t q = 0; q < pitch; ++q) out[(size_t)at * pitch + q] = cand[(size_t)j * pitch + q];
    __syncthreads();
    if (threadIdx.x == 1023) s_base += s_scan[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_out = take;
}

// diagnostic (debug 524288): rows whose pivot on the matrix pipe lies more than 4e-3 (of the row's norm) below the f32 argmax
__global__ void assign_check_kernel(const double *__restrict__ Z, uint32_t n, const float *__restrict__ piv, const uint32_t *__restrict__ a,
                                    const uint32_t *__restrict__ b, uint32_t *__restrict__ bad) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  float sa = 0.f, sb = 0.f, nn = 0.f;
  for (int k = 0; k < 32; ++k) {
    const float x = (float)Z[(size_t)r * 32 + k];
    sa = __builtin_fmaf(x, piv[(size_t)a[r] * 32 + k], sa);
    sb = __builtin_fmaf(x, piv[(size_t)b[r] * 32 + k], sb);
    nn = __builtin_fmaf(x, x, nn);
  }
  if (fabsf(sa - sb) > 4e-3f * sqrtf(nn)) atomicAdd(bad, 1u);
}
// ---- do the pivots cover the data? (round 5) ---------------------------------------------------------------------------------
// A pivot is a sampled row; where the data has more clusters than pivots, the rows of a cluster without one go to whichever pivots of
// OTHER clusters lie nearest -- a few of them, far from each other -- and the image no longer holds a row's neighbours together: the
// grouped thresholds fail their check, the cone bound keeps every tile pair, and a 1 M x 128 table takes 40 ms instead of 17 (rank-32
// mixture of 2 048 clusters under 1 024 pivots).  The assignment already knows: such a row's cosine with its pivot is far below that
// of a row whose cluster has one.  So the cosines are binned, rows below 0.8 of the 90 % point count as not covered, and where they
// are many the next 1 024 pivots are drawn FROM them (every round covers 1 - exp(-pivots / clusters left) of what is left, where
// pivots drawn from all rows waste most of theirs on clusters that have one), up to 4 096 pivots or 240 rows per pivot.  Any set of
// pivots gives a correct order; this one is a function of the rows alone.
__device__ __forceinline__ float cover_cosine(const int *__restrict__ ka, uint32_t mask_a, const int *__restrict__ kb, uint32_t mask_b,
                                              const double *__restrict__ norms, uint32_t r, bool *usable) {
  const double nr = norms[r];
  *usable = nr > 0.0 && nr < __builtin_inf();
  float v = __uint_as_float((uint32_t)ka[r] & mask_a);
  if (kb) v = fmaxf(v, __uint_as_float((uint32_t)kb[r] & mask_b));
  return *usable ? v / (float)nr : 0.f;
}
// hist[256]: the cosine of every row with its pivot in steps of 1 / 256 (negative ones in bin 0); hist[256]: rows without a direction
__global__ __launch_bounds__(256) void cover_hist_kernel(const int *__restrict__ ka, uint32_t mask_a, const int *__restrict__ kb, uint32_t mask_b,
                                                         const double *__restrict__ norms, uint32_t n, uint32_t *__restrict__ hist) {
  __shared__ uint32_t s_h[257];
  s_h[threadIdx.x] = 0;
  if (threadIdx.x == 0) s_h[256] = 0;
  __syncthreads();
  for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < n; r += gridDim.x * 256) {
    bool usable;
    const float c = cover_cosine(ka, mask_a, kb, mask_b, norms, r, &usable);
    atomicAdd(&s_h[usable ? (uint32_t)fminf(255.f, fmaxf(0.f, c * 256.f)) : 256u], 1u);
  }
  __syncthreads();
  if (s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_h[threadIdx.x]);
  if (threadIdx.x == 0 && s_h[256]) atomicAdd(&hist[256], s_h[256]);
}
__global__ void cover_flag_kernel(const int *__restrict__ ka, uint32_t mask_a, const int *__restrict__ kb, uint32_t mask_b,
                                  const double *__restrict__ norms, uint32_t n, uint32_t below_bin, uint32_t *__restrict__ flag) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  bool usable;
  const float c = cover_cosine(ka, mask_a, kb, mask_b, norms, r, &usable);
  flag[r] = (usable && (uint32_t)fminf(255.f, fmaxf(0.f, c * 256.f)) < below_bin) ? 1u : 0u;
}
// the same by image position (perm[p] = row): flag[p]; then the flagged positions in ascending order
__global__ void cover_flag_pos_kernel(const int *__restrict__ ka, uint32_t mask_a, const double *__restrict__ norms, const uint32_t *__restrict__ perm,
                                      uint32_t n, uint32_t below_bin, uint32_t *__restrict__ flag) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  bool usable;
  const float c = cover_cosine(ka, mask_a, nullptr, 0u, norms, perm[p], &usable);
  flag[p] = (usable && (uint32_t)fminf(255.f, fmaxf(0.f, c * 256.f)) < below_bin) ? 1u : 0u;
}
__global__ void cover_list_pos_kernel(const uint32_t *__restrict__ flag, const uint64_t *__restrict__ rank, uint32_t n, uint32_t room,
                                      uint32_t *__restrict__ out) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n && flag[p] && rank[p] < room) out[rank[p]] = p;
}
// ids[j], j < m: one of the rows not covered out of each of m even strata of their ranks (rank[n] = how many there are, >= m)
__global__ void cover_draw_kernel(const uint32_t *__restrict__ flag, const uint64_t *__restrict__ rank, uint32_t n, uint32_t m,
                                  uint32_t *__restrict__ ids) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n || !flag[r]) return;
  const uint64_t U = rank[n], k = rank[r], per = U / m;
  const uint64_t j = min((uint64_t)m - 1, k / per);
  uint32_t x = (uint32_t)j * 0x9E3779B9u;
  x ^= x >> 15;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  if (k == j * per + x % per) ids[j] = r;
}

// perm[pos] = row: the rows grouped by the nearest (by cosine) of 256 GIVEN unit pivots (piv: 256 rows of pitch = dim rounded up
// to 32, f32, device), stable within a pivot.  Any n >= 1, dim <= 128.
__global__ void order_group_starts_kernel(const uint64_t *__restrict__ offsets, uint32_t n_blocks, uint32_t np, uint32_t n,
                                          uint32_t *__restrict__ out) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < np) out[g] = (uint32_t)offsets[(size_t)g * n_blocks];
  if (g == np) out[g] = n;
}

// group_start (optional, np + 1 values, device): where the rows of every pivot begin in the order
// keys_out (optional, n values; rotated coordinates only): every row's best (product, pivot) key of the assignment, key_value_bits()
// of it being the product
// own (a build sharded by rows that holds every row's coordinates): the pivots of a rank's OWN rows are found here and the
// assignments (and keys) all-gathered -- 4 B a row instead of every rank sweeping all rows against all pivots; a row's pivot is a
// function of the row and the pivots alone, so every rank ends with the same order
int order_by_pivots(as_ctx *ctx, const double *rows, uint32_t n, uint32_t d, const float *piv, uint32_t *perm, uint32_t np,
                    uint32_t *group_start, const float *piv_sub, int *keys_out, const uint32_t *assign_pre, const ShardedRows *own) {
  if (d > 128 || n == 0 || np == 0 || np > (uint32_t)kPivotsMax || np % kPivotChunk != 0)
    return fail(ctx, AS_ERR_UNSUPPORTED, "row order: shape not covered");
  const uint32_t dp = (d + 3) / 4 * 4, n_blocks = (n + 255) / 256;
  const uint32_t dpv = (dp + 31) / 32 * 32;  // pitch of the pivots = the kernel's register count
  // (assign_pre: the pivot of every row is known already -- cluster_order behind its rounds of further pivots)
  struct {
    DevBuf<uint32_t> own;
    uint32_t *p = nullptr;
  } assign;
  DevBuf<uint32_t> counts;
  DevBuf<uint64_t> offsets;
  if (assign_pre) {
    assign.p = const_cast<uint32_t *>(assign_pre);
  } else {
    AS_TRY(assign.own.alloc(ctx, n));
    assign.p = assign.own.p;
  }
  AS_TRY(counts.alloc(ctx, (size_t)np * n_blocks));
  AS_TRY(offsets.alloc(ctx, (size_t)np * n_blocks + 1));
#define AS_ASSIGN(DPV)                                                                                                         \
  hipLaunchKernelGGL(assign_pivots_kernel<DPV>, dim3(n_blocks), dim3(256), 0, ctx->stream, rows, n, d, piv, assign.p, (int)np)
  const bool on_mfma = d == 32 && np % 32 == 0;
  // the rows this call sweeps: all of them, or (own) the rank's own range, in place
  const bool sharded = own && on_mfma && !assign_pre && (uint64_t)own->lo + own->count <= n;
  const uint32_t r_lo = sharded ? (uint32_t)own->lo : 0u, r_n = sharded ? (uint32_t)own->count : n;
  const double *r_rows = rows + (size_t)r_lo * d;
  ctx->stats["knn_join_order_assign_sharded"] = sharded ? 1 : 0;
  if (assign_pre) {
  } else if (on_mfma && np <= (uint32_t)kPivotsPass) {  // the rotated coordinates: on the matrix pipe
    if (r_n)
      hipLaunchKernelGGL(assign_pivots_mfma_kernel, dim3(std::min(cdiv(r_n, 128), 512u)), dim3(256), (size_t)(np / 32) * 2 * 64 * 16, ctx->stream, r_rows, r_n,
                         piv, np, assign.p + r_lo, 0u, 10u, keys_out ? keys_out + r_lo : nullptr, keys_out ? 6u : 4u);
  } else if (on_mfma) {  // ... in passes of 1024 pivots, the best key so far carried from pass to pass
    DevBuf<int> keys;
    if (!keys_out) AS_TRY(keys.alloc(ctx, n));
    for (uint32_t p0 = 0; p0 < np && r_n; p0 += kPivotsPass) {
      const uint32_t pc = std::min<uint32_t>(kPivotsPass, np - p0);
      const uint32_t flags = (p0 ? 1u : 0u) | (p0 + pc < np ? 2u : (keys_out ? 6u : 4u));
      hipLaunchKernelGGL(assign_pivots_mfma_kernel, dim3(std::min(cdiv(r_n, 128), 512u)), dim3(256), (size_t)(pc / 32) * 2 * 64 * 16, ctx->stream, r_rows, r_n,
                         piv + (size_t)p0 * 32, pc, assign.p + r_lo, p0, 12u, (keys_out ? keys_out : keys.p) + r_lo, flags);
    }
  } else
  switch (dpv / 32) {
    case 1: AS_ASSIGN(32); break;
    case 2: AS_ASSIGN(64); break;
    case 3: AS_ASSIGN(96); break;
    default: AS_ASSIGN(128); break;
  }
  if (on_mfma && !assign_pre && (ctx->opt_debug & 524288)) {
    DevBuf<uint32_t> ref, bad;
    AS_TRY(ref.alloc(ctx, n));
    AS_TRY(bad.alloc(ctx, 1));
    AS_HIP(ctx, hipMemsetAsync(bad.p, 0, sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(assign_pivots_kernel<32>, dim3(n_blocks), dim3(256), 0, ctx->stream, rows, n, d, piv, ref.p, (int)np);
    hipLaunchKernelGGL(assign_check_kernel, dim3(n_blocks), dim3(256), 0, ctx->stream, rows, n, piv, (const uint32_t *)assign.p, (const uint32_t *)ref.p, bad.p);
    uint32_t h = 0;
    AS_HIP(ctx, copy_d2h(ctx, &h, bad.p, sizeof(h)));
    ctx->stats["knn_join_order_assign_off"] = h;
  }
#undef AS_ASSIGN
  if (sharded) {
    AS_HIP(ctx, hipGetLastError());
    AS_TRY(own->gather(assign.p, sizeof(uint32_t)));
    if (keys_out) AS_TRY(own->gather(keys_out, sizeof(int)));
  }
  hipLaunchKernelGGL(order_count_kernel, dim3(n_blocks), dim3(256), 0, ctx->stream, assign.p, n, n_blocks, counts.p, (int)np);
  AS_HIP(ctx, hipGetLastError());
  AS_TRY(exclusive_scan_u32_u64(ctx, counts.p, (uint64_t)np * n_blocks, offsets.p));
  hipLaunchKernelGGL(order_place_kernel, dim3(n_blocks), dim3(256), 0, ctx->stream, assign.p, n, n_blocks, offsets.p, perm);
  // piv_sub: inside a pivot's group the rows are then put in the order of their nearest pivot of a SECOND set: rows of one group
  // that are near each other stay together, so where a group holds several small clusters a 64-row block of the image holds one
  // of them, not a mix (1 M x 128, 4 096 clusters of 244 rows: the masked f16 pass 84 -> 50 ms).  (The second nearest pivot of
  // the FIRST set, kept by the same pass, was tried as the key: the top-2 bookkeeping made the assignment 2 ms slower.)
  DevBuf<uint32_t> assign_sub, starts_own;
  if (piv_sub && on_mfma) {
    AS_TRY(assign_sub.alloc(ctx, n));
    uint32_t *gs = group_start;
    if (!gs) {
      AS_TRY(starts_own.alloc(ctx, np + 1));
      gs = starts_own.p;
    }
    // (opt_join_order_sub = the divisor of the second set's size: 1 = as many pivots as the first, 2 = half, ...)
    const uint32_t np_sub = std::min<uint32_t>(kPivotsPass, std::max<uint32_t>(32u, np / (uint32_t)std::max<int64_t>(1, ctx->opt_join_order_sub) / 32u * 32u));
    if (r_n)
      hipLaunchKernelGGL(assign_pivots_mfma_kernel, dim3(std::min(cdiv(r_n, 128), 1024u)), dim3(256), (size_t)(np_sub / 32) * 2 * 64 * 16, ctx->stream, r_rows, r_n,
                         piv_sub, np_sub, assign_sub.p + r_lo);
    if (sharded) {
      AS_HIP(ctx, hipGetLastError());
      AS_TRY(own->gather(assign_sub.p, sizeof(uint32_t)));
    }
    hipLaunchKernelGGL(order_group_starts_kernel, dim3(cdiv(np + 1, 256)), dim3(256), 0, ctx->stream, offsets.p, n_blocks, np, n, gs);
    hipLaunchKernelGGL(order_sub_kernel, dim3(np), dim3(256), 0, ctx->stream, (const uint32_t *)gs, (const uint32_t *)assign_sub.p, (int)np, perm);
  }
  if (group_start)
    hipLaunchKernelGGL(order_group_starts_kernel, dim3(cdiv(np + 1, 256)), dim3(256), 0, ctx->stream, offsets.p, n_blocks, np, n, group_start);
  AS_HIP(ctx, hipGetLastError());
  // (no synchronisation to hand the workspace back: a block returned to the cache is reused by later work on this same stream only)
  return AS_OK;
}

// piv (np x dpv, 32 coordinates used) -> the same pivots in the order of the greedy chain
// inv_out (optional, np): where each pivot went; near_out (optional): np x 16, the nearest groups of every group in the new order
static int chain_pivots(as_ctx *ctx, float *piv, uint32_t np, uint32_t dpv, uint32_t *inv_out = nullptr, DevBuf<uint32_t> *near_out = nullptr) {
  if (near_out) near_out->release();
  if (np > (uint32_t)kPivotsMax || np < 2) return inv_out ? fail(ctx, AS_ERR_UNSUPPORTED, "row order: chain of this many pivots") : AS_OK;
  const uint32_t list_len = np > 2048 ? kChainList / 2 : kChainList;
  DevBuf<uint32_t> order, inv_own;
  DevBuf<uint16_t> lists;
  DevBuf<float> tmp;
  AS_TRY(order.alloc(ctx, np));
  AS_TRY(lists.alloc(ctx, (size_t)np * list_len));
  AS_TRY(tmp.alloc(ctx, (size_t)np * dpv));
  const size_t lds = (size_t)np * list_len * sizeof(uint16_t) + np;
  AS_HIP(ctx, hipFuncSetAttribute((const void *)pivot_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(pivot_near_lists_kernel, dim3(np), dim3(256), 0, ctx->stream, (const float *)piv, dpv, np, lists.p, list_len);
  hipLaunchKernelGGL(pivot_chain_kernel, dim3(1), dim3(64), lds, ctx->stream, (const uint16_t *)lists.p, np, order.p, list_len);
  hipLaunchKernelGGL(pivot_permute_kernel, dim3(np), dim3(64), 0, ctx->stream, (const float *)piv, (const uint32_t *)order.p, dpv, tmp.p);
  AS_HIP(ctx, hipGetLastError());
  AS_HIP(ctx, hipMemcpyAsync(piv, tmp.p, sizeof(float) * (size_t)np * dpv, hipMemcpyDeviceToDevice, ctx->stream));
  if (inv_out || near_out) {
    uint32_t *inv = inv_out;
    if (!inv) {
      AS_TRY(inv_own.alloc(ctx, np));
      inv = inv_own.p;
    }
    hipLaunchKernelGGL(chain_inverse_kernel, dim3(cdiv(np, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)order.p, np, inv);
    if (near_out) {
      AS_TRY(near_out->alloc(ctx, (size_t)np * 16));
      hipLaunchKernelGGL(chain_near16_kernel, dim3(cdiv(np * 16, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)order.p, (const uint32_t *)inv,
                         (const uint16_t *)lists.p, list_len, np, near_out->p);
    }
    AS_HIP(ctx, hipGetLastError());
  }
  return AS_OK;
}

// perm[pos] = row: the rows grouped by their nearest pivot (device array of n), the pivots being 256 of the rows themselves.
// dim <= 128, n >= 256 * 256.  piv_out (optional, 256 x 128 floats, device): the pivots, for order_by_pivots on other rows.
// n_pivots: 0 = 256; up to 1024 (a multiple of 32; clamped so that a pivot stands for 512 rows or more, 256 at least).  piv_out: n_pivots x 128.
// group_start / np_out (optional): order_by_pivots' group starts (room for 1025 values) and the number of pivots used.
int order_label_stats(as_ctx *ctx, const uint32_t *perm, uint32_t n, uint32_t C);
int cluster_order(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, uint32_t *perm, float *piv_out,
                  uint32_t n_pivots, uint32_t *group_start, uint32_t *np_out, uint32_t max_pivots, DevBuf<uint32_t> *far_pos,
                  uint32_t *far_count, DevBuf<uint32_t> *near_out, const ShardedRows *own) {
  if (far_count) *far_count = 0;
  if (d > 128 || n < (uint32_t)kPivots * 256u) return fail(ctx, AS_ERR_UNSUPPORTED, "row order: shape not covered");
  // (more than 256 only where a pivot still stands for 512 rows or more: the thresholds of the own join are taken against the
  // sampled rows -- one in 32 -- of a few groups)
  // (beyond 1 024 -- asked for, or grown below where the data has more clusters than that -- a pivot may stand for 240 rows)
  const uint32_t per_pivot = n_pivots > (uint32_t)kPivotsPass ? 240u : 512u;
  uint32_t np = std::min<uint32_t>(std::max<uint32_t>(n_pivots ? n_pivots : (uint32_t)kPivots, 32u) / 32u * 32u,
                                   std::min<uint32_t>((uint32_t)kPivotsMax, std::max<uint32_t>((uint32_t)kPivots, n / per_pivot / 32u * 32u)));
  ProfScope ps(ctx, "knn_join_row_order");
  const uint32_t dp = (d + 3) / 4 * 4;
  // max_pivots: room in piv_out / group_start for a set grown where the first one does not cover the data (rotated coordinates only)
  const uint32_t np_room = std::min<uint32_t>(std::min<uint32_t>(max_pivots, (uint32_t)kPivotsMax), n / 240u / 32u * 32u);
  const bool may_grow = d == 32 && piv_out && np_room > np && np % 32 == 0 && ctx->opt_join_order_cover > 0;
  DevBuf<uint32_t> ids;
  DevBuf<float> piv_own;
  AS_TRY(ids.alloc(ctx, np));
  float *piv = piv_out;
  if (!piv) {
    AS_TRY(piv_own.alloc(ctx, (size_t)np * 128));
    piv = piv_own.p;
  }
  hipLaunchKernelGGL(order_pivot_ids_kernel, dim3(cdiv(np, 256)), dim3(256), 0, ctx->stream, n, ids.p, np);
  const uint32_t dpv = (dp + 31) / 32 * 32;
  hipLaunchKernelGGL(gather_pivots_kernel, dim3(np), dim3(128), 0, ctx->stream, rows, norms, ids.p, d, dpv, piv);
  AS_HIP(ctx, hipGetLastError());
  const bool chain = d == 32 && ctx->opt_join_order_chain;
  if (near_out) near_out->release();
  if (chain) AS_TRY(chain_pivots(ctx, piv, np, dpv, nullptr, near_out));
  ctx->stats["knn_join_order_chain"] = chain ? 1 : 0;
  DevBuf<float> piv_sub;
  const bool with_sub = d == 32 && ctx->opt_join_order_sub;
  if (with_sub) {  // a second set of pivots (other rows), for the order inside the groups
    DevBuf<uint32_t> ids2;
    AS_TRY(ids2.alloc(ctx, np));
    AS_TRY(piv_sub.alloc(ctx, (size_t)np * 128));
    hipLaunchKernelGGL(order_pivot_ids_kernel, dim3(cdiv(np, 256)), dim3(256), 0, ctx->stream, n - n / (2 * np), ids2.p, np);
    hipLaunchKernelGGL(shift_ids_kernel, dim3(cdiv(np, 256)), dim3(256), 0, ctx->stream, ids2.p, np, n / (2 * np));
    hipLaunchKernelGGL(gather_pivots_kernel, dim3(np), dim3(128), 0, ctx->stream, rows, norms, ids2.p, d, dpv, piv_sub.p);
    AS_HIP(ctx, hipGetLastError());
    if (chain && ctx->opt_join_order_chain > 1) AS_TRY(chain_pivots(ctx, piv_sub.p, np, dpv));  // (the order inside the groups too)
  }
  DevBuf<int> keys;
  float cluster_cut = 0.75f;  // (where the coverage was not measured: what 0.8 of a typical 90 % point is)
  if (may_grow) AS_TRY(keys.alloc(ctx, n));
  AS_TRY(order_by_pivots(ctx, rows, n, d, piv, perm, np, group_start, with_sub ? piv_sub.p : nullptr, may_grow ? keys.p : nullptr, nullptr, own));
  ctx->stats["knn_join_order_pivots"] = np;
  if (may_grow) {
    // the first set's assignment is the one round 4 measured; what follows runs only where it leaves many rows far from their pivot
    const uint32_t np_first = np, mask_a = np_first > (uint32_t)kPivotsPass ? ~4095u : ~1023u;
    DevBuf<int> keys_new;
    DevBuf<uint32_t> hist, flag, cand_ids, keep;
    DevBuf<uint64_t> rank;
    DevBuf<float> cand;
    AS_TRY(hist.alloc(ctx, 257));
    uint32_t rounds = 0, last_below = 0;
    for (;;) {
      AS_HIP(ctx, hipMemsetAsync(hist.p, 0, 257 * sizeof(uint32_t), ctx->stream));
      hipLaunchKernelGGL(cover_hist_kernel, dim3(std::min(cdiv(n, 256), 1024u)), dim3(256), 0, ctx->stream, (const int *)keys.p, mask_a,
                         (const int *)(rounds ? keys_new.p : nullptr), ~4095u, norms, n, hist.p);
      uint32_t hh[257];
      AS_HIP(ctx, copy_d2h(ctx, hh, hist.p, sizeof(hh)));
      const uint64_t usable = (uint64_t)n - hh[256];
      uint64_t run = 0;
      int q90 = 255;
      for (int i = 0; i < 256; ++i) {
        run += hh[i];
        if (run * 10 >= usable * 9) {
          q90 = i;
          break;
        }
      }
      const uint32_t below = (uint32_t)(q90 * 8 / 10);
      uint64_t low = 0;
      for (uint32_t i = 0; i < below; ++i) low += hh[i];
      last_below = low ? below : 0u;
      cluster_cut = (float)below * (1.f / 256.f);
      if (rounds == 0) ctx->stats["knn_join_order_cover_low_e3"] = (int64_t)(low * 1000 / std::max<uint64_t>(1, usable));
      ctx->stats["knn_join_order_cover_left_e3"] = (int64_t)(low * 1000 / std::max<uint64_t>(1, usable));
      // (the first round where a seventh of the rows lie far from their pivot, later ones down to 3 %)
      const uint64_t trigger_e3 = rounds == 0 ? (uint64_t)ctx->opt_join_order_cover : 30u;
      const uint32_t want = std::min<uint32_t>(kPivotsPass, np_room - np);
      // (twice as many rows drawn as pivots wanted, fewer where few rows are left: 8 rows or more per draw)
      const uint32_t draws = (uint32_t)std::min<uint64_t>(2 * (uint64_t)want, low / 8);
      if (want == 0 || low * 1000 < trigger_e3 * usable || draws < 32) break;
      if (rounds == 0) {
        AS_TRY(keys_new.alloc(ctx, n));
        AS_TRY(flag.alloc(ctx, n));
        AS_TRY(rank.alloc(ctx, (size_t)n + 1));
        AS_TRY(cand.alloc(ctx, (size_t)2 * kPivotsPass * dpv));
        AS_TRY(cand_ids.alloc(ctx, 2 * kPivotsPass));
        AS_TRY(keep.alloc(ctx, 2 * kPivotsPass + 1));
      }
      hipLaunchKernelGGL(cover_flag_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, (const int *)keys.p, mask_a,
                         (const int *)(rounds ? keys_new.p : nullptr), ~4095u, norms, n, below, flag.p);
      AS_TRY(exclusive_scan_u32_u64(ctx, flag.p, n, rank.p));
      hipLaunchKernelGGL(cover_draw_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)flag.p, (const uint64_t *)rank.p, n, draws,
                         cand_ids.p);
      hipLaunchKernelGGL(gather_pivots_kernel, dim3(draws), dim3(128), 0, ctx->stream, rows, norms, (const uint32_t *)cand_ids.p, d, dpv, cand.p);
      hipLaunchKernelGGL(cover_distinct_kernel, dim3(draws), dim3(256), 0, ctx->stream, (const float *)cand.p, dpv, draws, (float)below * (1.f / 256.f), keep.p);
      hipLaunchKernelGGL(cover_keep_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const float *)cand.p, dpv, draws, (const uint32_t *)keep.p, want,
                         piv + (size_t)np * dpv, keep.p + 2 * kPivotsPass);
      uint32_t m = 0;
      AS_HIP(ctx, copy_d2h(ctx, &m, keep.p + 2 * kPivotsPass, sizeof(m)));
      if (m == 0) break;
      // the new pivots against every row: the best of the drawn sets so far in keys_new (12 pivot bits; only its product is read)
      hipLaunchKernelGGL(assign_pivots_mfma_kernel, dim3(std::min(cdiv(n, 128), 512u)), dim3(256), (size_t)(m / 32) * 2 * 64 * 16, ctx->stream, rows, n,
                         (const float *)(piv + (size_t)np * dpv), m, (uint32_t *)nullptr, np, 12u, keys_new.p, (rounds ? 1u : 0u) | 2u);
      AS_HIP(ctx, hipGetLastError());
      np += m;
      ++rounds;
    }
    ctx->stats["knn_join_order_cover_rounds"] = rounds;
    if (rounds && chain) {
      // the whole set along one chain; every row's pivot is the better of its two bests so far, renumbered (no pass over 4 096
      // pivots again: 1.2 ms)
      DevBuf<uint32_t> inv, assign;
      AS_TRY(inv.alloc(ctx, np));
      AS_TRY(assign.alloc(ctx, n));
      AS_TRY(chain_pivots(ctx, piv, np, dpv, inv.p, near_out));
      hipLaunchKernelGGL(assign_remap_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, (const int *)keys.p, np_first > (uint32_t)kPivotsPass ? 12u : 10u,
                         (const int *)keys_new.p, (const uint32_t *)inv.p, n, np > (uint32_t)kPivotsPass ? 12u : 10u, assign.p, keys.p);
      AS_HIP(ctx, hipGetLastError());
      AS_TRY(order_by_pivots(ctx, rows, n, d, piv, perm, np, group_start, with_sub ? piv_sub.p : nullptr, nullptr, assign.p, nullptr));
      ctx->stats["knn_join_order_pivots"] = np;
    } else if (rounds) {  // (no chain: the pivots stay where they are, the rows are assigned again)
      if (near_out) near_out->release();
      AS_TRY(order_by_pivots(ctx, rows, n, d, piv, perm, np, group_start, with_sub ? piv_sub.p : nullptr, keys.p, nullptr, nullptr));
      ctx->stats["knn_join_order_pivots"] = np;
    }
    // The rows still far from their pivot, by image position: their neighbours are not where the image order puts a row's
    // neighbours, so the caller takes their thresholds from the whole sample (own_join_prepare).  (The cut of the last count.)
    if (far_pos && far_count && last_below > 0) {
      const uint32_t mask_f = np > (uint32_t)kPivotsPass ? ~4095u : ~1023u;
      if (!flag.p) AS_TRY(flag.alloc(ctx, n));
      if (!rank.p) AS_TRY(rank.alloc(ctx, (size_t)n + 1));
      hipLaunchKernelGGL(cover_flag_pos_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, (const int *)keys.p, mask_f, norms, (const uint32_t *)perm, n,
                         last_below, flag.p);
      AS_TRY(exclusive_scan_u32_u64(ctx, flag.p, n, rank.p));
      uint64_t h_far = 0;
      AS_HIP(ctx, copy_d2h(ctx, &h_far, rank.p + n, sizeof(h_far)));
      ctx->stats["knn_join_order_far_rows"] = (int64_t)h_far;
      if (h_far > 0 && h_far * 100 <= (uint64_t)n * 8) {  // (more than 8 %: the check of the thresholds decides for all rows)
        const uint32_t room = (uint32_t)((h_far + 511) / 512 * 512);
        AS_TRY(far_pos->alloc(ctx, room));
        AS_HIP(ctx, hipMemsetAsync(far_pos->p, 0xFF, sizeof(uint32_t) * room, ctx->stream));
        hipLaunchKernelGGL(cover_list_pos_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)flag.p, (const uint64_t *)rank.p, n, room,
                           far_pos->p);
        AS_HIP(ctx, hipGetLastError());
        *far_count = (uint32_t)h_far;
      }
    }
  }
  if (np_out) *np_out = np;
  // How many rows a cluster holds, roughly: along the chain the pivots of one cluster follow each other, so n / (1 + the steps of the
  // chain between pivots less similar than the cut of the coverage) -- the caller sizes the band of the own-neighbourhood thresholds
  // by it (own_join_prepare: a band that covers a cluster certifies cuts near a row's K-th best; 4 M x 128 and 1 M x 768 with clusters of
  // 4 - 16 k rows build 3.5 % faster with 3 - 4 super tiles on either side than with 2, clusters of 1 000 rows 2 % slower).
  ctx->stats["knn_join_order_cluster_rows"] = 0;
  if (chain && np >= 2) {
    DevBuf<uint32_t> brk;
    AS_TRY(brk.alloc(ctx, 1));
    AS_HIP(ctx, hipMemsetAsync(brk.p, 0, sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(chain_breaks_kernel, dim3(cdiv(np, 256)), dim3(256), 0, ctx->stream, (const float *)piv, dpv, np, cluster_cut, brk.p);
    uint32_t hb = 0;
    AS_HIP(ctx, copy_d2h(ctx, &hb, brk.p, sizeof(hb)));
    ctx->stats["knn_join_order_cluster_rows"] = (int64_t)(n / (1u + hb));
  }
  if (ctx->opt_debug_label_mod > 0) AS_TRY(order_label_stats(ctx, perm, n, (uint32_t)ctx->opt_debug_label_mod));
  return AS_OK;
}

// diagnostic ("debug_label_mod" = C: row r carries label r mod C, as the bench mixtures deal their clusters): how many different
// labels the 128 positions of a tile hold, and how many of its rows carry the tile's most frequent label
namespace {
__global__ __launch_bounds__(128) void tile_labels_kernel(const uint32_t *__restrict__ perm, uint32_t n, uint32_t C, uint32_t *__restrict__ distinct,
                                                          uint32_t *__restrict__ major) {
  __shared__ uint32_t s_l[128];
  const uint32_t T = blockIdx.x, t = threadIdx.x, pos = T * 128u + t;
  s_l[t] = pos < n ? perm[pos] % C : 0xFFFFFFFFu;
  __syncthreads();
  const uint32_t mine = s_l[t];
  uint32_t same = 0, first = 1;
  for (uint32_t k = 0; k < 128; ++k) {
    same += s_l[k] == mine ? 1u : 0u;
    if (k < t && s_l[k] == mine) first = 0;
  }
  __shared__ uint32_t s_d, s_m;
  if (t == 0) s_d = 0, s_m = 0;
  __syncthreads();
  if (mine != 0xFFFFFFFFu) {
    if (first) atomicAdd(&s_d, 1u);
    atomicMax(&s_m, same);
  }
  __syncthreads();
  if (t == 0) distinct[T] = s_d, major[T] = s_m;
}
}  // namespace
int order_label_stats(as_ctx *ctx, const uint32_t *perm, uint32_t n, uint32_t C) {
  const uint32_t ntiles = (n + 127u) / 128u;
  DevBuf<uint32_t> dd, mm;
  AS_TRY(dd.alloc(ctx, ntiles));
  AS_TRY(mm.alloc(ctx, ntiles));
  hipLaunchKernelGGL(tile_labels_kernel, dim3(ntiles), dim3(128), 0, ctx->stream, perm, n, C, dd.p, mm.p);
  std::vector<uint32_t> hd(ntiles), hm(ntiles);
  AS_HIP(ctx, copy_d2h(ctx, hd.data(), dd.p, sizeof(uint32_t) * ntiles));
  AS_HIP(ctx, copy_d2h(ctx, hm.data(), mm.p, sizeof(uint32_t) * ntiles));
  std::sort(hd.begin(), hd.end());
  std::sort(hm.begin(), hm.end());
  uint64_t sum_m = 0;
  for (uint32_t v : hm) sum_m += v;
  ctx->stats["order_tile_labels_q10"] = hd[ntiles / 10];
  ctx->stats["order_tile_labels_q50"] = hd[ntiles / 2];
  ctx->stats["order_tile_labels_q90"] = hd[(size_t)ntiles * 9 / 10];
  ctx->stats["order_tile_major_share_e3"] = (int64_t)(sum_m * 1000 / std::max<uint64_t>(1, n));
  return AS_OK;
}

// out[i] = in[ids[i]], i < n (an id of ~0: 0)
__global__ void gather_f32_by_ids_kernel(const float *__restrict__ in, const uint32_t *__restrict__ ids, uint32_t n, float *__restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = ids[i] == 0xFFFFFFFFu ? 0.f : in[ids[i]];
}
int gather_f32_by_ids(as_ctx *ctx, const float *in, const uint32_t *ids, uint32_t n, float *out) {
  hipLaunchKernelGGL(gather_f32_by_ids_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, in, ids, n, out);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

// out[perm[i]] = in[i], i < n
int scatter_by_perm(as_ctx *ctx, const float *in, const uint32_t *perm, uint32_t n, float *out) {
  hipLaunchKernelGGL(scatter_f32_by_perm_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, in, perm, n, out);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

// perm[p] = the index of the p-th smallest value (stable: equal values keep their order; NaNs last): the queries of a batch in
// the order of their lambdas, so that the 32 rows of a row group of the search scan share a narrow lambda range (mfma_join_f16.hip)
//
// A sort of (order-preserving 64-bit key, index) pairs, hand-written: chunks of 4096 pairs are sorted in LDS by one workgroup
// each (bitonic network), then runs are merged pairwise, every element finding its place by a binary search in the partner run.
// The pair (key, index) is a total order (indices are distinct), so the result is THE stable order whatever network sorts it.
// 10 k queries: one chunk pass + two merge passes, three launches on the context's stream and no host round trip (a library
// radix sort stood here through round 3: four launches of its own and a synchronisation to hand its buffers back).
namespace {
constexpr uint32_t kSortChunk = 4096;
struct KeyId {
  unsigned long long k;
  uint32_t i;
};
__device__ __forceinline__ bool keyid_less(unsigned long long ka, uint32_t ia, unsigned long long kb, uint32_t ib) { return ka < kb || (ka == kb && ia < ib); }
__device__ __forceinline__ unsigned long long value_key(double x) {
  unsigned long long b = (unsigned long long)__double_as_longlong(x);
  b = (b >> 63) ? ~b : (b | 0x8000000000000000ull);  // order-preserving
  return x != x ? ~0ull : b;
}
// one workgroup per chunk: keys from the values, bitonic sort of (key, index) in LDS, sorted chunk out
__global__ __launch_bounds__(1024) void sort_chunks_kernel(const double *__restrict__ v, uint32_t n, unsigned long long *__restrict__ keys,
                                                           uint32_t *__restrict__ ids) {
  __shared__ unsigned long long s_k[kSortChunk];
  __shared__ uint32_t s_i[kSortChunk];
  const uint32_t base = blockIdx.x * kSortChunk;
  // (the network spans the next power of two of what the chunk holds, not the whole chunk: the ~1 600 open rows of a second pass are
  // sorted in 66 rounds over 2 048 slots instead of 91 over 8 192 -- 75 -> 15 us)
  uint32_t lim = 2;
  while (lim < kSortChunk && base + lim < n) lim <<= 1;
  for (uint32_t t = threadIdx.x; t < lim; t += 1024) {
    const uint32_t g = base + t;
    s_k[t] = g < n ? value_key(v[g]) : ~0ull;
    s_i[t] = g < n ? g : 0xFFFFFFFFu;  // padding sorts behind everything (a NaN has key ~0 too, and a smaller index)
  }
  __syncthreads();
  for (uint32_t k = 2; k <= lim; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t t = threadIdx.x; t < lim; t += 1024) {
        const uint32_t p = t ^ j;
        if (p > t) {
          const unsigned long long ka = s_k[t], kb = s_k[p];
          const uint32_t ia = s_i[t], ib = s_i[p];
          const bool up = (t & k) == 0;
          if (keyid_less(kb, ib, ka, ia) == up) {
            s_k[t] = kb, s_i[t] = ib;
            s_k[p] = ka, s_i[p] = ia;
          }
        }
      }
      __syncthreads();
    }
  for (uint32_t t = threadIdx.x; t < lim; t += 1024) {
    const uint32_t g = base + t;
    if (g < n) {
      keys[g] = s_k[t];
      ids[g] = s_i[t];
    }
  }
}
// runs of `run` sorted pairs -> runs of 2 `run`: element e of a run goes to (its index in the run) + (elements of the partner run
// that precede it)
__global__ void merge_runs_kernel(const unsigned long long *__restrict__ k_in, const uint32_t *__restrict__ i_in, uint32_t n, uint32_t run,
                                  unsigned long long *__restrict__ k_out, uint32_t *__restrict__ i_out) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const uint32_t r = e / run, pair0 = (r & ~1u) * run;
  const uint32_t other0 = (r & 1u) ? pair0 : pair0 + run, other1 = min(other0 + run, n);
  const unsigned long long k = k_in[e];
  const uint32_t i = i_in[e];
  uint32_t lo = min(other0, n), hi = other1;  // (a last run without a partner: nothing precedes)
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (keyid_less(k_in[mid], i_in[mid], k, i)) lo = mid + 1;
    else hi = mid;
  }
  const uint32_t dst = pair0 + (e - r * run) + (lo - min(other0, n));
  k_out[dst] = k;
  i_out[dst] = i;
}
}  // namespace
int order_by_value(as_ctx *ctx, const double *values, uint32_t n, uint32_t *perm) {
  if (n == 0) return AS_OK;
  DevBuf<unsigned long long> ka, kb;
  DevBuf<uint32_t> ib;
  AS_TRY(ka.alloc(ctx, n));
  AS_TRY(kb.alloc(ctx, n));
  AS_TRY(ib.alloc(ctx, n));
  // the number of merge passes decides where the chunk pass writes, so that the last pass lands in `perm`
  uint32_t passes = 0;
  for (uint64_t run = kSortChunk; run < n; run <<= 1) ++passes;
  unsigned long long *k_cur = ka.p, *k_nxt = kb.p;
  uint32_t *i_cur = (passes & 1u) ? ib.p : perm, *i_nxt = (passes & 1u) ? perm : ib.p;
  hipLaunchKernelGGL(sort_chunks_kernel, dim3(cdiv(n, kSortChunk)), dim3(1024), 0, ctx->stream, values, n, k_cur, i_cur);
  for (uint64_t run = kSortChunk; run < n; run <<= 1) {
    hipLaunchKernelGGL(merge_runs_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, (const unsigned long long *)k_cur, (const uint32_t *)i_cur, n,
                       (uint32_t)run, k_nxt, i_nxt);
    std::swap(k_cur, k_nxt);
    std::swap(i_cur, i_nxt);
  }
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;  // (the workspace blocks go back to the cache of the ONE stream every later user runs on: common.h, ws_alloc)
}

}  // namespace asx
