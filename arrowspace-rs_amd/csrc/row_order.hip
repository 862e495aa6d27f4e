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
#include "common.h"

namespace asx {

namespace {

constexpr int kPivots = 256;      // pivots of an order found on the rows themselves
constexpr int kPivotsMax = 1024;  // ... on 32 rotated coordinates (a quarter of the arithmetic per pivot: csrc/rotate_wide.hip), where
                                  // 1024 instead of 256 pivots leave the f16 pass of the join half the blocks
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
__global__ __launch_bounds__(256) void assign_pivots_mfma_kernel(const double *__restrict__ Z, uint32_t n, const float *__restrict__ piv,
                                                                 uint32_t np, uint32_t *__restrict__ assign) {
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
  ro_v16f32 zero;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;
  for (uint32_t t = 0; t < np / 32; ++t) {
    const ro_f16x8 b0 = s_b[(t * 2 + 0) * 64 + lane], b1 = s_b[(t * 2 + 1) * 64 + lane];
    ro_v16f32 x = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b0, zero, 0, 0, 0);
    x = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b1, x, 0, 0, 0);
    const uint32_t inv = 1023u - (t * 32 + (uint32_t)lc);
#pragma unroll
    for (int r = 0; r < 16; ++r) best[r] = max(best[r], (int)((__float_as_uint(x[r]) & 0xFFFFFC00u) | inv));
  }
  // accumulator register r of lane (lh, lc) = row 4 lh + (r & 3) + 8 (r >> 2) of the tile, column lc: the columns meet here
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int bv = best[r];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) bv = max(bv, __shfl_xor(bv, o, 64));
    const uint32_t ba = 1023u - ((uint32_t)bv & 1023u);
    const uint32_t row = row0 + 4 * lh + (r & 3) + 8 * (r >> 2);
    if (lc == 0 && row < n) assign[row] = ba;
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
  __shared__ uint32_t s_cur[kPivotsMax];
  __shared__ uint32_t s_rows[kSubMax];
  __shared__ uint32_t s_key[256];
  __shared__ uint32_t s_scan[256];
  const uint32_t lo = group_start[blockIdx.x], hi = group_start[blockIdx.x + 1], len = hi - lo;
  if (len < 2 || len > kSubMax) return;
  for (int c = threadIdx.x; c < np; c += 256) s_cur[c] = 0;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < len; i += 256) atomicAdd(&s_cur[key_of_row[perm[lo + i]]], 1u);  // (a sum: any order)
  __syncthreads();
  // exclusive scan of the np counts (np <= 1024: four per thread)
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
constexpr uint32_t kChainList = 32;
__global__ __launch_bounds__(256) void pivot_near_lists_kernel(const float *__restrict__ piv, uint32_t dpv, uint32_t np, uint16_t *__restrict__ lists) {
  __shared__ float s_sim[1024];
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
  for (uint32_t k = 0; k < kChainList; ++k) {
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
      lists[(size_t)g * kChainList + k] = s_at[0] == 0xFFFFFFFFu ? (uint16_t)0xFFFFu : (uint16_t)s_at[0];
      if (s_at[0] != 0xFFFFFFFFu) s_sim[s_at[0]] = -__builtin_inff();
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(64) void pivot_chain_kernel(const uint16_t *__restrict__ lists, uint32_t np, uint32_t *__restrict__ order) {
  extern __shared__ unsigned char smem[];
  uint16_t *s_list = reinterpret_cast<uint16_t *>(smem);                       // np x 32
  unsigned char *s_taken = smem + (size_t)np * kChainList * sizeof(uint16_t);  // np
  const uint32_t lane = threadIdx.x;
  for (uint32_t i = lane; i < np * kChainList; i += 64) s_list[i] = lists[i];
  for (uint32_t i = lane; i < np; i += 64) s_taken[i] = i == 0 ? 1 : 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  uint32_t cur = 0, low = 1;  // low: every pivot below it is taken
  if (lane == 0) order[0] = 0;
  for (uint32_t step = 1; step < np; ++step) {
    const uint32_t c = lane < kChainList ? s_list[cur * kChainList + lane] : 0xFFFFu;
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
__global__ void pivot_permute_kernel(const float *__restrict__ piv, const uint32_t *__restrict__ order, uint32_t dpv, float *__restrict__ out) {
  const uint32_t p = blockIdx.x;
  for (uint32_t k = threadIdx.x; k < dpv; k += blockDim.x) out[(size_t)p * dpv + k] = piv[(size_t)order[p] * dpv + k];
}

__global__ void order_pivot_ids_kernel(uint32_t n, uint32_t *__restrict__ ids, uint32_t np) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < np) ids[p] = (uint32_t)(((uint64_t)p * n) / np + (((uint64_t)p * 2654435761ull) % (n / np)));
}

}  // namespace

// perm[pos] = row: the rows grouped by the nearest (by cosine) of 256 GIVEN unit pivots (piv: 256 rows of pitch = dim rounded up
// to 32, f32, device), stable within a pivot.  Any n >= 1, dim <= 128.
__global__ void order_group_starts_kernel(const uint64_t *__restrict__ offsets, uint32_t n_blocks, uint32_t np, uint32_t n,
                                          uint32_t *__restrict__ out) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < np) out[g] = (uint32_t)offsets[(size_t)g * n_blocks];
  if (g == np) out[g] = n;
}

// group_start (optional, np + 1 values, device): where the rows of every pivot begin in the order
int order_by_pivots(as_ctx *ctx, const double *rows, uint32_t n, uint32_t d, const float *piv, uint32_t *perm, uint32_t np,
                    uint32_t *group_start, const float *piv_sub) {
  if (d > 128 || n == 0 || np == 0 || np > (uint32_t)kPivotsMax || np % kPivotChunk != 0)
    return fail(ctx, AS_ERR_UNSUPPORTED, "row order: shape not covered");
  const uint32_t dp = (d + 3) / 4 * 4, n_blocks = (n + 255) / 256;
  const uint32_t dpv = (dp + 31) / 32 * 32;  // pitch of the pivots = the kernel's register count
  DevBuf<uint32_t> assign, counts;
  DevBuf<uint64_t> offsets;
  AS_TRY(assign.alloc(ctx, n));
  AS_TRY(counts.alloc(ctx, (size_t)np * n_blocks));
  AS_TRY(offsets.alloc(ctx, (size_t)np * n_blocks + 1));
#define AS_ASSIGN(DPV)                                                                                                         \
  hipLaunchKernelGGL(assign_pivots_kernel<DPV>, dim3(n_blocks), dim3(256), 0, ctx->stream, rows, n, d, piv, assign.p, (int)np)
  const size_t lds_mfma = (size_t)(np / 32) * 2 * 64 * 16;
  if (d == 32 && np % 32 == 0 && lds_mfma <= 64 * 1024) {  // the rotated coordinates: on the matrix pipe
    hipLaunchKernelGGL(assign_pivots_mfma_kernel, dim3(std::min(cdiv(n, 128), 512u)), dim3(256), lds_mfma, ctx->stream, rows, n, piv, np, assign.p);
  } else
  switch (dpv / 32) {
    case 1: AS_ASSIGN(32); break;
    case 2: AS_ASSIGN(64); break;
    case 3: AS_ASSIGN(96); break;
    default: AS_ASSIGN(128); break;
  }
#undef AS_ASSIGN
  hipLaunchKernelGGL(order_count_kernel, dim3(n_blocks), dim3(256), 0, ctx->stream, assign.p, n, n_blocks, counts.p, (int)np);
  AS_HIP(ctx, hipGetLastError());
  AS_TRY(exclusive_scan_u32_u64(ctx, counts.p, (uint64_t)np * n_blocks, offsets.p));
  hipLaunchKernelGGL(order_place_kernel, dim3(n_blocks), dim3(256), 0, ctx->stream, assign.p, n, n_blocks, offsets.p, perm);
  // piv_sub: inside a pivot's group the rows are then put in the order of their nearest pivot of a SECOND set: rows of one group
  // that are near each other stay together, so where a group holds several small clusters a 64-row block of the image holds one
  // of them, not a mix (1 M x 128, 4 096 clusters of 244 rows: the masked f16 pass 84 -> 50 ms).  (The second nearest pivot of
  // the FIRST set, kept by the same pass, was tried as the key: the top-2 bookkeeping made the assignment 2 ms slower.)
  DevBuf<uint32_t> assign_sub, starts_own;
  if (piv_sub && d == 32 && np % 32 == 0 && lds_mfma <= 64 * 1024) {
    AS_TRY(assign_sub.alloc(ctx, n));
    uint32_t *gs = group_start;
    if (!gs) {
      AS_TRY(starts_own.alloc(ctx, np + 1));
      gs = starts_own.p;
    }
    // (opt_join_order_sub = the divisor of the second set's size: 1 = as many pivots as the first, 2 = half, ...)
    const uint32_t np_sub = std::max<uint32_t>(32u, np / (uint32_t)std::max<int64_t>(1, ctx->opt_join_order_sub) / 32u * 32u);
    hipLaunchKernelGGL(assign_pivots_mfma_kernel, dim3(std::min(cdiv(n, 128), 1024u)), dim3(256), (size_t)(np_sub / 32) * 2 * 64 * 16, ctx->stream, rows, n, piv_sub,
                       np_sub, assign_sub.p);
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
static int chain_pivots(as_ctx *ctx, float *piv, uint32_t np, uint32_t dpv) {
  if (np > 1024 || np < 2) return AS_OK;
  DevBuf<uint32_t> order;
  DevBuf<uint16_t> lists;
  DevBuf<float> tmp;
  AS_TRY(order.alloc(ctx, np));
  AS_TRY(lists.alloc(ctx, (size_t)np * kChainList));
  AS_TRY(tmp.alloc(ctx, (size_t)np * dpv));
  const size_t lds = (size_t)np * kChainList * sizeof(uint16_t) + np;
  AS_HIP(ctx, hipFuncSetAttribute((const void *)pivot_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(pivot_near_lists_kernel, dim3(np), dim3(256), 0, ctx->stream, (const float *)piv, dpv, np, lists.p);
  hipLaunchKernelGGL(pivot_chain_kernel, dim3(1), dim3(64), lds, ctx->stream, (const uint16_t *)lists.p, np, order.p);
  hipLaunchKernelGGL(pivot_permute_kernel, dim3(np), dim3(64), 0, ctx->stream, (const float *)piv, (const uint32_t *)order.p, dpv, tmp.p);
  AS_HIP(ctx, hipGetLastError());
  AS_HIP(ctx, hipMemcpyAsync(piv, tmp.p, sizeof(float) * (size_t)np * dpv, hipMemcpyDeviceToDevice, ctx->stream));
  return AS_OK;
}

// perm[pos] = row: the rows grouped by their nearest pivot (device array of n), the pivots being 256 of the rows themselves.
// dim <= 128, n >= 256 * 256.  piv_out (optional, 256 x 128 floats, device): the pivots, for order_by_pivots on other rows.
// n_pivots: 0 = 256; up to 1024 (a multiple of 32; clamped so that a pivot stands for 512 rows or more, 256 at least).  piv_out: n_pivots x 128.
// group_start / np_out (optional): order_by_pivots' group starts (room for 1025 values) and the number of pivots used.
int cluster_order(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, uint32_t *perm, float *piv_out,
                  uint32_t n_pivots, uint32_t *group_start, uint32_t *np_out) {
  if (d > 128 || n < (uint32_t)kPivots * 256u) return fail(ctx, AS_ERR_UNSUPPORTED, "row order: shape not covered");
  // (more than 256 only where a pivot still stands for 512 rows or more: the thresholds of the own join are taken against the
  // sampled rows -- one in 32 -- of a few groups)
  const uint32_t np = std::min<uint32_t>(std::max<uint32_t>(n_pivots ? n_pivots : (uint32_t)kPivots, 32u) / 32u * 32u,
                                         std::min<uint32_t>((uint32_t)kPivotsMax, std::max<uint32_t>((uint32_t)kPivots, n / 512u / 32u * 32u)));
  ProfScope ps(ctx, "knn_join_row_order");
  const uint32_t dp = (d + 3) / 4 * 4;
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
  if (np_out) *np_out = np;
  const bool chain = d == 32 && ctx->opt_join_order_chain;
  if (chain) AS_TRY(chain_pivots(ctx, piv, np, dpv));
  ctx->stats["knn_join_order_chain"] = chain ? 1 : 0;
  DevBuf<float> piv_sub;
  if (d == 32 && ctx->opt_join_order_sub) {  // a second set of pivots (other rows), for the order inside the groups
    DevBuf<uint32_t> ids2;
    AS_TRY(ids2.alloc(ctx, np));
    AS_TRY(piv_sub.alloc(ctx, (size_t)np * 128));
    hipLaunchKernelGGL(order_pivot_ids_kernel, dim3(cdiv(np, 256)), dim3(256), 0, ctx->stream, n - n / (2 * np), ids2.p, np);
    hipLaunchKernelGGL(shift_ids_kernel, dim3(cdiv(np, 256)), dim3(256), 0, ctx->stream, ids2.p, np, n / (2 * np));
    hipLaunchKernelGGL(gather_pivots_kernel, dim3(np), dim3(128), 0, ctx->stream, rows, norms, ids2.p, d, dpv, piv_sub.p);
    AS_HIP(ctx, hipGetLastError());
    if (chain && ctx->opt_join_order_chain > 1) AS_TRY(chain_pivots(ctx, piv_sub.p, np, dpv));  // (the order inside the groups too)
    return order_by_pivots(ctx, rows, n, d, piv, perm, np, group_start, piv_sub.p);
  }
  return order_by_pivots(ctx, rows, n, d, piv, perm, np, group_start, nullptr);
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
  for (uint32_t t = threadIdx.x; t < kSortChunk; t += 1024) {
    const uint32_t g = base + t;
    s_k[t] = g < n ? value_key(v[g]) : ~0ull;
    s_i[t] = g < n ? g : 0xFFFFFFFFu;  // padding sorts behind everything (a NaN has key ~0 too, and a smaller index)
  }
  __syncthreads();
  for (uint32_t k = 2; k <= kSortChunk; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t t = threadIdx.x; t < kSortChunk; t += 1024) {
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
  for (uint32_t t = threadIdx.x; t < kSortChunk; t += 1024) {
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
