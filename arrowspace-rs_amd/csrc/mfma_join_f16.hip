// mfma_join_f16.hip -- the symmetric threshold join (scheme and meaning of the lists: mfma_join_f32.hip) contracted
// on the 16-bit matrix pipe from ONE f16 plane of the unit rows (v_mfma_f32_32x32x16_f16: the bf16 rate).
//
// Why f16 and why one plane.  A candidate pass only needs a rigorous bound between its product s and the exact
// cosine: what is not in row r's list has s < theta[r], hence an exact cosine below theta[r] + delta, and the exact
// f64 re-rank certifies a row only if its K-th neighbour is clear of that (mfma_drivers.hip).  bf16 keeps 8
// significant bits: a single plane is good to 3.9e-3, too coarse to list by, so the bf16 form of this kernel carried
// a second (lo) plane through LDS and a second MFMA phase for the blocks near a threshold.  f16 keeps 11 bits: unit
// rows scaled by 2^7 (no element is subnormal unless it is below 5e-7 of the row's norm) give
//     |s - cos| <= 2 * (2^-11 + 2^-14 sqrt(d_pad) / 128) + f32 accumulation  =  cos_delta_f16(d) ~ 1.02e-3 at d = 128,
// which costs ~0.1 % more uncertified rows and a handful more row fetches in the re-rank than 1e-4 does -- and the
// whole lo-plane machinery goes: half the bytes staged per tile, a third of the MFMAs on the blocks that listed
// something, no second phase, 64 instead of 128 registers of A fragments per 64 rows.
//
// Geometry (chosen for bytes per flop: the bf16 form moved 32 B/clk/CU from L2 to LDS, half of what the path gives):
// a workgroup = 8 waves x 64 rows = one 512-row super tile; B is staged 256 columns (64 KB at dim 128) at a time
// through a 2-deep LDS-DMA ring, one barrier per stage; every LDS fragment read feeds two MFMAs (the wave's two 32-row
// groups).  Per stage a wave issues 128 MFMAs, 64 ds_read_b128, 8 DMA pieces.
//
// Threshold test without per-element compares: the first MFMA of a block takes C = -theta_row (16 registers per row
// group that stay put for the whole unit), so the block ends with acc = s - theta_row; "some product reaches its row's
// threshold" is max(acc) >= 0 (8 v_max3), and the column side, s >= theta_col, is tested conservatively on the same
// maximum as max(acc) >= min(0, theta_col - max_rows(theta_row)).  Blocks that pass (~13 %) run the exact element-wise
// test and queue their pairs; listed scores are acc + theta_row.
#include <hip/hip_fp16.h>

#include <type_traits>

#include "mfma_common.h"

namespace asx {

uint32_t f16_ksteps(uint32_t d);  // mfma_join_f16_wide.hip: ceil(d / 16), whole 64-wide panels beyond 128

namespace {

typedef float v16f32 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

constexpr int kSuperH = 512;        // rows per super tile (four packed 128-row tiles)
constexpr uint32_t kQueueJoin = 256;    // queued appends per wave (one accumulator register adds at most 128)
constexpr uint32_t kQueueSearch = 160;  // ... in the search mode, whose per-unit arrays take the room (one side is listed: <= 64 per pass)
constexpr float kOperandScale = 128.f;     // unit rows are stored times 2^7 ...
constexpr float kProductScale = 16384.f;   // ... so products (and thresholds inside the kernel) carry 2^14

// Packed f16 image: tiles of 128 rows; fragment (T, ks, s) = 64 lanes x 8 halves (16 B per lane):
//   P[((T * KS + ks) * 4 + s) * 64 + l][j] = f16( 128 * unit(R)[128 T + 32 s + (l & 31)][16 ks + 8 (l >> 5) + j] )
// = the A / B operand of v_mfma_f32_32x32x16_f16 as it stands.  Rows beyond n, columns beyond d and non-finite
// quotients are 0 (a row with a non-finite norm is re-done exactly whatever its products say: flag_tiny_norm_rows_kernel).
// err_part (optional, rows x 2 KS doubles): the squared error of the thread's 8 stored values, || stored / 128 - unit ||^2 over its
// piece of the row, for f16_error_reduce_kernel: the plane error comes out of the packing pass instead of a second read of the rows.
__global__ void pack_f16_kernel(const double *__restrict__ R, const double *__restrict__ norms,
                                const uint32_t *__restrict__ row_ids, uint64_t row0, uint64_t n, uint32_t d, uint32_t KS,
                                uint64_t total_frag_lanes, f16x8 *__restrict__ P, double *__restrict__ err_part = nullptr) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total_frag_lanes) return;
  const uint32_t l = g & 63;
  const uint32_t s = (g >> 6) & 3;
  const uint64_t tk = g >> 8;
  const uint32_t ks = (uint32_t)(tk % KS);
  const uint64_t T = tk / KS;
  const uint64_t row = T * kTileRows + s * 32 + (l & 31);
  const uint32_t k0 = ks * 16 + 8 * (l >> 5);
  f16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (_Float16)0.0f;
  double e2 = 0.0;
  if (row < n && !(row_ids && row_ids[row] == 0xFFFFFFFFu)) {  // (an id of ~0: an empty slot, a zero row)
    const uint64_t src = row_ids ? row_ids[row] : row0 + row;
    const double nr = norms[src];
    if (nr > 0.0) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k0 + j < d) {
          const double u = R[src * d + k0 + j] / nr;
          const float x = (float)u * kOperandScale;
          v[j] = (x == x && __builtin_fabsf(x) <= 2.f * kOperandScale) ? (_Float16)x : (_Float16)0.0f;
          if (err_part && nr < __builtin_inf()) {  // (the measure f16_row_error_kernel takes: rows with a finite positive norm)
            const double e = (double)(float)v[j] * (1.0 / kOperandScale) - u;
            e2 += e * e;
          }
        }
    }
  }
  P[g] = v;
  if (err_part && row < n) err_part[row * (2ull * KS) + 2ull * ks + (l >> 5)] = e2;
}
// The same image from an f16 image of ALL rows in ROW order (`src`: the layout above with row = row id, n_src rows) instead of the
// f64 rows: position p of P takes the 2 KS vectors of row ids[p] (ids null: row p; an id of ~0 or a position beyond `count`: zeros).
// Pure data movement, so P holds exactly what pack_f16_kernel stores for the same ids.  A build sharded by rows (csrc/sharded.hip)
// packs its own rows once, all-gathers that image (256 B per row at 128 columns instead of the 1 KB of the row) and takes every
// image of the preparation -- the one in the order of the join, the samples of the thresholds -- from it.
__global__ void f16_image_gather_kernel(const f16x8 *__restrict__ src, uint64_t n_src, const uint32_t *__restrict__ ids, uint64_t count, uint32_t KS,
                                        uint64_t total_frag_lanes, f16x8 *__restrict__ P) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total_frag_lanes) return;
  const uint32_t l = g & 63;
  const uint32_t s = (g >> 6) & 3;
  const uint64_t tk = g >> 8;
  const uint32_t ks = (uint32_t)(tk % KS);
  const uint64_t T = tk / KS;
  const uint64_t pos = T * kTileRows + s * 32 + (l & 31);
  f16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (_Float16)0.0f;
  if (pos < count) {
    const uint64_t id = ids ? (uint64_t)ids[pos] : pos;
    if (!(ids && ids[pos] == 0xFFFFFFFFu) && id < n_src)
      v = src[(((id / kTileRows) * KS + ks) * 4 + ((id % kTileRows) >> 5)) * 64 + (id & 31) + (l & 32)];
  }
  P[g] = v;
}
// The same image (rows of exactly 128 values) from the rows' f16 form ROW-MAJOR in row order (H: n_src x 128 halves, what
// rotate_wide.hip's fused pass leaves): the 128 positions of tile T take the 256 bytes of rows ids[128 T ..] (ids null: row p; an id
// of ~0 or beyond n_src, or a position beyond n: zeros) -- sixteen lanes a row, each one 16-byte fragment vector -- through an LDS copy of the tile.
__global__ __launch_bounds__(256) void f16_image_from_rows_kernel(const f16x8 *__restrict__ H, uint64_t n_src, const uint32_t *__restrict__ ids,
                                                                   uint64_t n, f16x8 *__restrict__ P) {
  constexpr int KS = 8;
  __shared__ f16x8 s_tile[KS * 4 * 64];  // 32 KB: [ks][s][lane]
  const int tid = threadIdx.x, sub = tid & 15;  // sub: values 8 sub .. 8 sub + 7 of a row = k step sub / 2, half sub % 2
  const uint64_t T = blockIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int q = 16 * i + (tid >> 4);  // position in the tile
    const uint64_t pos = T * kTileRows + (uint64_t)q;
    f16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (_Float16)0.0f;
    if (pos < n) {
      const uint32_t id = ids ? ids[pos] : (uint32_t)pos;
      if (id != 0xFFFFFFFFu && (uint64_t)id < n_src) v = H[(uint64_t)id * 16 + sub];
    }
    s_tile[((sub >> 1) * 4 + (q >> 5)) * 64 + (q & 31) + 32 * (sub & 1)] = v;
  }
  __syncthreads();
  f16x8 *out = P + T * (uint64_t)(KS * 4 * 64);
#pragma unroll
  for (int q = 0; q < KS * 4 * 64 / 256; ++q) out[q * 256 + tid] = s_tile[q * 256 + tid];
}
// ---- rows of exactly 128 values: image and plane error in ONE coalesced pass over the rows (round 3) ---------------------------
// The per-fragment-lane packer above reads a row in sixteen 64-byte pieces from sixteen different waves and ran at 0.6 TB/s
// (1.66 ms per GB), the error kernel read the rows again (1.09 ms).  Here a workgroup owns a 128-row tile of the image: each of
// its four waves reads 32 rows, a row per instruction (lane l takes values 2 l, 2 l + 1: 1 KB contiguous), divides by the norm,
// rounds to f16, writes the two halves to their place in an LDS copy of the tile's fragment layout and keeps the squared error
// of its two values; a halving butterfly adds those over the 64 lanes per row (a fixed order), and the 32 KB tile goes out
// with 16-byte stores.  Same values as pack_f16_kernel stores, same error measure as f16_row_error_kernel.
// WIDE (rows of a multiple of 128 values, round 5): block (T, c) does the same for columns [128 c, 128 c + 128) of the tile's rows --
// k steps [8 c, 8 c + 8) of the image, 32 KB that lie together -- and leaves the squared error of each row's piece in
// part[position * chunks + c]; f16_error_reduce_kernel adds a row's pieces in index order.  (Before: the per-fragment-lane packer
// + a second read of the rows for the error, 5.6 ms for 1 M x 768 = 2.2 TB/s over two reads.)
template <bool WIDE>
__global__ __launch_bounds__(256) void pack_f16_tile128_kernel(const double *__restrict__ R, const double *__restrict__ norms,
                                                               const uint32_t *__restrict__ row_ids, uint64_t n, f16x8 *__restrict__ P,
                                                               unsigned int *__restrict__ err_bits, float *__restrict__ err_rows,
                                                               uint32_t d, double *__restrict__ part) {
  constexpr int KS = 8;
  __shared__ f16x8 s_tile[KS * 4 * 64];  // 32 KB: [ks][s][lane]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // (WIDE: the blocks of a tile follow each other, so the pieces of a row are asked for close in time)
  const uint32_t chunks = WIDE ? d / 128u : 1u, chunk = WIDE ? blockIdx.x % chunks : 0u;
  const uint64_t T = WIDE ? blockIdx.x / chunks : blockIdx.x;
  const uint64_t pitch = WIDE ? (uint64_t)d : 128u;
  typedef double pk_d2 __attribute__((ext_vector_type(2)));
  // lane l holds values k = 2 l, 2 l + 1 of a row: k step l / 8, half (l % 8) / 4, elements 2 (l % 4) and + 1 of that vector
  const int ks = lane >> 3, half = (lane >> 2) & 1, j2 = (lane & 3);
  float worst = 0.f;
  constexpr int kBatch = 8;
  for (int r0 = 0; r0 < 32; r0 += kBatch) {
    pk_d2 x[kBatch];
    double nr[kBatch];
    bool live[kBatch];
#pragma unroll
    for (int b = 0; b < kBatch; ++b) {
      const uint64_t pos = T * kTileRows + (uint64_t)wv * 32 + (uint64_t)(r0 + b);
      live[b] = pos < n && !(row_ids && row_ids[pos] == 0xFFFFFFFFu);
      const uint64_t src = live[b] ? (row_ids ? (uint64_t)row_ids[pos] : pos) : 0;
      nr[b] = live[b] ? norms[src] : 0.0;
      x[b] = reinterpret_cast<const pk_d2 *>(R + src * pitch + (uint64_t)chunk * 128u)[lane];
    }
    double e2[kBatch];
#pragma unroll
    for (int b = 0; b < kBatch; ++b) {
      _Float16 h0 = (_Float16)0.0f, h1 = (_Float16)0.0f;
      e2[b] = 0.0;
      if (live[b] && nr[b] > 0.0) {
        const double u0 = x[b].x / nr[b], u1 = x[b].y / nr[b];
        const float f0 = (float)u0 * kOperandScale, f1 = (float)u1 * kOperandScale;
        h0 = (f0 == f0 && __builtin_fabsf(f0) <= 2.f * kOperandScale) ? (_Float16)f0 : (_Float16)0.0f;
        h1 = (f1 == f1 && __builtin_fabsf(f1) <= 2.f * kOperandScale) ? (_Float16)f1 : (_Float16)0.0f;
        if (nr[b] < __builtin_inf()) {
          const double d0 = (double)(float)h0 * (1.0 / kOperandScale) - u0, d1 = (double)(float)h1 * (1.0 / kOperandScale) - u1;
          e2[b] = d0 * d0 + d1 * d1;
        }
      }
      const int rr = r0 + b;  // row of the wave's 32: sub-tile s = wv, lane' = rr + 32 half
      _Float16 *dst = reinterpret_cast<_Float16 *>(&s_tile[(ks * 4 + wv) * 64 + rr + 32 * half]) + 2 * j2;
      dst[0] = h0;
      dst[1] = h1;
    }
    // per-row sums over the 64 lanes: 8 values per lane -> halving butterfly (lane bits 5, 4, 3), then full adds (bits 2, 1, 0)
    {
      const bool up5 = (lane >> 5) & 1, up4 = (lane >> 4) & 1, up3 = (lane >> 3) & 1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double keep = up5 ? e2[i + 4] : e2[i], send = up5 ? e2[i] : e2[i + 4];
        e2[i] = keep + __shfl_xor(send, 32, 64);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const double keep = up4 ? e2[i + 2] : e2[i], send = up4 ? e2[i] : e2[i + 2];
        e2[i] = keep + __shfl_xor(send, 16, 64);
      }
      {
        const double keep = up3 ? e2[1] : e2[0], send = up3 ? e2[0] : e2[1];
        e2[0] = keep + __shfl_xor(send, 8, 64);
      }
      double t = e2[0];
      t += __shfl_xor(t, 4, 64);
      t += __shfl_xor(t, 2, 64);
      t += __shfl_xor(t, 1, 64);
      // (the eight lanes of group (bits 5, 4, 3) hold the sum of row 4 b5 + 2 b4 + b3 of the batch)
      const uint64_t pos = T * kTileRows + (uint64_t)wv * 32 + (uint64_t)(r0 + 4 * (int)up5 + 2 * (int)up4 + (int)up3);
      if (WIDE) {
        if ((lane & 7) == 0 && pos < n) part[pos * chunks + chunk] = t;
      } else {
        const float e = (float)(__builtin_sqrt(t) * (1.0 + 1e-6));  // rounded up
        if (e == e) worst = __builtin_fmaxf(worst, e);
        // the plane error of each row by image position, for the per-pair bound of the pruning (rerank_split.h); NaN -> +inf: such
        // a row prunes nothing
        if (err_rows && (lane & 7) == 0 && pos < n) err_rows[pos] = e == e ? e : __builtin_inff();
      }
    }
  }
  if (!WIDE) {
    for (int o = 32; o > 0; o >>= 1) worst = __builtin_fmaxf(worst, __shfl_xor(worst, o, 64));
    if (lane == 0) raise_max_u32(err_bits, __float_as_uint(worst));  // non-negative floats order like their bits
  }
  __syncthreads();
  f16x8 *out = P + (T * (uint64_t)chunks + chunk) * (uint64_t)(KS * 4 * 64);
#pragma unroll
  for (int q = 0; q < KS * 4 * 64 / 256; ++q) out[q * 256 + tid] = s_tile[q * 256 + tid];
}

// E = max over the rows of sqrt(sum of the 2 KS partial sums, in index order) (1 + 1e-6): see f16_row_error_kernel
__global__ void f16_error_reduce_kernel(const double *__restrict__ err_part, uint64_t n, uint32_t slots, unsigned int *__restrict__ out_bits,
                                        float *__restrict__ err_rows) {
  const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float e = 0.f;
  if (row < n) {
    double e2 = 0.0;
    for (uint32_t s_ = 0; s_ < slots; ++s_) e2 += err_part[row * slots + s_];
    const float ev = (float)(__builtin_sqrt(e2) * (1.0 + 1e-6));  // rounded up
    if (ev == ev) e = ev;
    if (err_rows) err_rows[row] = ev == ev ? ev : __builtin_inff();  // (by image position: the pruning's bound per pair)
  }
  for (int o = 32; o > 0; o >>= 1) e = __builtin_fmaxf(e, __shfl_xor(e, o, 64));
  if ((threadIdx.x & 63) == 0) raise_max_u32(out_bits, __float_as_uint(e));  // non-negative floats order like their bits
}

__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(size_t)(lds_void *)p; }

// LDS traffic between the lanes of ONE wave (queues, listing slots).  The hardware keeps a wave's LDS operations in order, so all
// that is needed is that they have been issued and completed (lgkmcnt) and that the compiler moves nothing across.  wave_sync()
// (mfma_common.h) says the same with two wavefront-scope fences -- which hipcc lowers to s_waitcnt vmcnt(0) as well: every listed
// block then waited for the stores and returning atomics of the flush before it and for the LDS-DMA of the next stage.
__device__ __forceinline__ void lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

// LDS fragment reads issued and waited for by hand (the compiler would wait each one out on the spot)
template <int OFF>
__device__ __forceinline__ void lds_read16(f16x8 &dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int PENDING>
__device__ __forceinline__ void lds_wait1(f16x8 &x) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(PENDING));
}
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// thresholds inside the kernel: clamped to [-1.5, 1.5] (no product of unit rows is outside [-1.01, 1.01], so the
// lists are the same and |C| stays below 2^15 for the accumulation bound), +inf beyond and for NaN; times 2^14
__device__ __forceinline__ float thr_scaled(float t) {
  if (!(t <= 1.5f)) return __builtin_inff();
  return __builtin_fmaxf(t, -1.5f) * kProductScale;
}

__device__ __forceinline__ float max16(const v16f32 &a) {
  float m = __builtin_fmaxf(__builtin_fmaxf(a[0], a[1]), a[2]);
  m = __builtin_fmaxf(__builtin_fmaxf(m, a[3]), a[4]);
  m = __builtin_fmaxf(__builtin_fmaxf(m, a[5]), a[6]);
  m = __builtin_fmaxf(__builtin_fmaxf(m, a[7]), a[8]);
  m = __builtin_fmaxf(__builtin_fmaxf(m, a[9]), a[10]);
  m = __builtin_fmaxf(__builtin_fmaxf(m, a[11]), a[12]);
  m = __builtin_fmaxf(__builtin_fmaxf(m, a[13]), a[14]);
  return __builtin_fmaxf(m, a[15]);
}

__device__ __forceinline__ void append16(uint32_t *__restrict__ cnt, uint32_t *__restrict__ cand, uint32_t cap, uint32_t target,
                                         uint32_t other, float score) {
  const uint32_t pos = atomicAdd(cnt + target, 1u);
  if (pos < cap) reinterpret_cast<uint2 *>(cand)[(size_t)target * cap + pos] = make_uint2(other, __float_as_uint(score));
}

// P = f16 image of ALL rows, T2 super tiles of 512 rows (zero padded).  theta: T2 * 512 thresholds, +inf on padding rows.
// the units of the symmetric masked pass (MODE 0) that hold at least one flagged stage, in (nearly) their order: list[0] = count
__global__ __launch_bounds__(256) void join_unit_list_kernel(const unsigned long long *__restrict__ mask, uint32_t T2, uint32_t I_begin, uint32_t I_count,
                                                             uint32_t W, uint32_t *__restrict__ list) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n_cblocks = (T2 + W - 1) / W, total = n_cblocks * I_count, n_norm = (T2 - 1) / 2;
  bool keep = false;
  if (u < total) {  // (the unit geometry of pair_join_f16_kernel, MODE 0)
    const uint32_t c = u / I_count, I = I_begin + (u - c * I_count);
    const uint32_t B0 = c * W, Wc = min(W, T2 - B0);
    const bool antipodal = (T2 % 2 == 0) && (I < T2 / 2) && T2 > 1;
    const uint32_t L = 1 + n_norm + (antipodal ? 1u : 0u);
    const uint32_t delta = B0 >= I ? B0 - I : B0 + T2 - I;
    uint32_t p_lo, p_hi;
    if (delta + Wc <= T2) {
      p_lo = delta;
      p_hi = min(delta + Wc, L);
    } else {
      p_lo = 0;
      p_hi = min(delta + Wc - T2, L);
    }
    if (p_hi > p_lo) {
      const uint32_t steps = (p_hi - p_lo) * 2;
      for (uint32_t t = 0; t < steps && !keep; ++t) {
        uint32_t S = I + p_lo + (t >> 1);
        S = S >= T2 ? S - T2 : S;
        keep = mask[(size_t)(I - I_begin) * 2 * T2 + 2 * S + (t & 1u)] != 0;
      }
    }
  }
  const unsigned long long m = __ballot(keep);
  uint32_t base = 0;
  const int lane = threadIdx.x & 63;
  if (lane == 0 && m) base = atomicAdd(list, (uint32_t)__builtin_popcountll(m));
  base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
  if (keep) list[1 + base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = u;
}

// Work units (column block of W super tiles, row super tile I) are dealt from an atomic counter, column block by
// column block, so that the workgroups running together stream the same columns.  Super tile I meets the column
// super tiles I .. I + (T2 - 1) / 2 (mod T2), the antipodal one only from the smaller index: every unordered pair of
// super tiles once, the same work for every I.
// MODE 1: the threshold scan of the search on the same machinery (one-sided).  PA = f16 image of the queries (whole 512-row
// super tiles, [I_begin, I_begin + I_count) of them), every one of which meets every column super tile of P (the items);
// theta = what the cosine alone must reach, (g_q - beta) / alpha, by query position (+inf on padding): the coarse test;
// s.q_g = g_q - beta, s.q_lam / s.x_lam = f32 lambdas of the queries / of all items (padded to whole super tiles).  A pair
// is listed for its query when alpha s - min(|beta lambda_q - beta lambda_x|, beta) >= g_q - beta, with that value as score.
struct F16SearchTerm {
  const float *q_lam, *q_g, *x_lam;
  float alpha, beta;
  uint32_t nq;
  // the one-sided second pass of the item graph on this mode: the row ids of the "queries" (by position in their image), so
  // that a row does not list ITSELF.  (It used to, "the re-rank drops it" -- but the pruning of the re-rank takes its cut from
  // the K-th best listed score, and with the row's own 1.0 in the list that is the (K - 1)-th best of the others: one rank too
  // high, enough to prune the true K-th neighbour when two candidates are 4e-6 apart and the f16 errors point the wrong way.
  // Found by scripts/stress_first_pass.py, n = 211 570, K = 39.)
  const uint32_t *self_ids;
};
// MASKED (MODE 0): mask = what the int8 first pass found (mfma_prefilter_i8.hip): one byte per (row super tile, stage of 256
// columns, wave), a bit per 32-column block that may hold a pair.  The unit visits only the stages with a bit set and a wave
// contracts only its own blocks (the fragment ring then restarts at every block).
constexpr int kFlushRounds = 3;  // rounds of 64 queue entries whose returning atomics go out together (each costs five registers)
// DIAG = false: the diagnostics (debug bits, cycle stamps) compiled out -- the form the build launches when no debug bit is set
// (128 columns behind the first pass: the scalar registers and branches they cost sit in the listing, which has none to spare).
template <int KS, int MODE, bool MASKED, bool DIAG = true>
__global__ __launch_bounds__(512, 2) void pair_join_f16_kernel(const f16x8 *__restrict__ P, const f16x8 *__restrict__ PA, uint32_t T2,
                                                               uint32_t I_begin, uint32_t I_count, uint32_t W, uint32_t n,
                                                               const float *__restrict__ theta, uint32_t cap,
                                                               uint32_t *__restrict__ cnt, uint32_t *__restrict__ cand,
                                                               uint32_t *__restrict__ unit_counter, int dbg_in,
                                                               unsigned long long *__restrict__ stamps, F16SearchTerm st,
                                                               const uint32_t *__restrict__ perm,
                                                               const unsigned long long *__restrict__ mask,
                                                               const uint32_t *__restrict__ perm_a,
                                                               const uint32_t *__restrict__ unit_list = nullptr) {
  // unit_list (or null): [0] = how many, [1 ..] = the units that hold a flagged stage (join_unit_list_kernel) -- behind the cone
  // bound 19 of 20 units are empty, and a workgroup that draws one pays a returning atomic, the read of its mask and three
  // barriers to find that out (0.6 ms of the 4.9 of the pass at 1 M rows)
  constexpr int kTileVecs = KS * 4 * 64;     // 16-B vectors of a 128-row tile
  constexpr int kStageVecs = 2 * kTileVecs;  // a stage = 256 columns
  constexpr int kBlocks = 8;                 // 32-column blocks per stage
  const int dbg = DIAG ? dbg_in : 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS: [2 stages of B][2 x 256 column thresholds][8 queues of (owner, other)][8 queues of scores][8 x 4 listing slots]
  f16x8 *bufB = reinterpret_cast<f16x8 *>(smem);
  float *colthr = reinterpret_cast<float *>(smem + (size_t)2 * kStageVecs * 16);
  __shared__ uint32_t s_unit;
  __shared__ uint32_t s_nlist[2];                  // MASKED: the unit's stages that hold something, in order
  __shared__ unsigned char s_step[128];            // (bytes: at 128 columns the kernel's LDS is within 2 KB of the 160 KB)
  __shared__ unsigned long long s_bits[128];

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lh = lane >> 5, lc = lane & 31;
  // passing pairs go to a per-wave LDS queue as (list owner, other row, score) and are appended 64 at a time: an append is
  // a returning atomic (~2 us before its store can go out) and, taken one by one, makes its wave late at the barrier
  // (a variable of its own, not a piece of the dynamic array the LDS-DMA writes: in front of every LDS access that may alias a
  // pending DMA the compiler waits for vmcnt(0) -- the pieces of the next stage in flight, the stores of the flush before)
  constexpr uint32_t kQueueH = MODE == 1 ? kQueueSearch : kQueueJoin;
  constexpr int kSlotFloats = 36;
  constexpr size_t kSideBytes = (size_t)8 * kQueueH * (sizeof(uint2) + sizeof(float)) + (size_t)8 * 4 * kSlotFloats * sizeof(float) +
                                (MODE == 1 ? (size_t)2 * kSuperH * sizeof(float) : 0);
  __shared__ __attribute__((aligned(16))) unsigned char s_side[kSideBytes];
  uint2 *queue = reinterpret_cast<uint2 *>(s_side) + (size_t)wv * kQueueH;
  float *queue_s = reinterpret_cast<float *>(s_side + (size_t)8 * kQueueH * sizeof(uint2)) + (size_t)wv * kQueueH;
  // listing scratch of the wave: 4 slots of (16 acc, 16 acc + theta_row, lane, theta_col, 2 unused) floats
  float *ex = reinterpret_cast<float *>(s_side + (size_t)8 * kQueueH * (sizeof(uint2) + sizeof(float))) + (size_t)wv * 4 * kSlotFloats;
  // MODE 1: lambda and g - beta of the unit's 512 queries
  float *s_qlam = reinterpret_cast<float *>(s_side + (size_t)8 * kQueueH * (sizeof(uint2) + sizeof(float))) + 8 * 4 * kSlotFloats;
  float *s_qg = s_qlam + kSuperH;
  uint32_t qn = 0;  // wave-uniform
  auto lanes_below = [&](unsigned long long m) -> uint32_t {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  };
  unsigned long long t_flush = 0, t_emit = 0;
  auto flush_queue = [&]() {
    unsigned long long tf0 = 0;
    if (dbg & 8) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tf0)::"memory");
    lds_sync();
    if constexpr (MODE == 0 && MASKED) {
      // one entry per pair (row, column): bit 0 / 1 of the score = listed for the row / the column.  All the returning atomics of
      // the queue's (up to four) rounds go out back to back and the stores follow: taken round by round (atomic, wait, store) a
      // flush cost a wave four to eight exposed round trips (55 k cycles, stamps of round 3).
      // (kFlushRounds at a time: five registers per round; four rounds spill)
      constexpr int R = kFlushRounds;
      const uint32_t fq_addr = lds_addr(queue) + 8u * (uint32_t)lane, fs_addr = lds_addr(queue_s) + 4u * (uint32_t)lane;
      for (uint32_t k0 = 0; k0 * 64u < qn; k0 += R) {
        uint32_t ta[R], tb[R], ts[R], pa[R], pb[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
          const uint32_t e = (k0 + (uint32_t)k) * 64u + lane;
          ts[k] = 0;
          ta[k] = tb[k] = 0;
          if (e < qn && !(dbg & 16384)) {
            // (read by hand: in front of an LDS read of its own the compiler waits for the stores of the rounds before -- vmcnt(0))
            unsigned long long vxy;
            uint32_t vs;
            asm volatile(
                "ds_read_b64 %0, %2\n\t"
                "ds_read_b32 %1, %3\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(vxy), "=&v"(vs)
                : "v"(fq_addr + (k0 + (uint32_t)k) * 512u), "v"(fs_addr + (k0 + (uint32_t)k) * 256u)
                : "memory");
            const uint32_t vx = (uint32_t)vxy, vy = (uint32_t)(vxy >> 32);
            if (vx != vy) {
              ta[k] = perm ? perm[vx] : vx;
              tb[k] = perm ? perm[vy] : vy;
              ts[k] = vs;
            }
          }
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
          pa[k] = pb[k] = 0;
          if (ts[k] & 1u) pa[k] = atomicAdd(cnt + ta[k], 1u);
          if (ts[k] & 2u) pb[k] = atomicAdd(cnt + tb[k], 1u);
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
          const uint32_t sc = ts[k] & ~3u;
          if ((ts[k] & 1u) && pa[k] < cap) reinterpret_cast<uint2 *>(cand)[(size_t)ta[k] * cap + pa[k]] = make_uint2(tb[k], sc);
          if ((ts[k] & 2u) && pb[k] < cap) reinterpret_cast<uint2 *>(cand)[(size_t)tb[k] * cap + pb[k]] = make_uint2(ta[k], sc);
        }
      }
    } else {
    for (uint32_t base = 0; base < qn; base += 64) {
      const uint32_t e = base + lane;
      if (e < qn && !(dbg & 16384)) {
        // positions in the image -> row ids where the image is packed in a permuted order (row_order.hip); a list of the
        // search mode is owned by a query position, which stays
        const uint2 v = queue[e];
        {
          // (MODE 1: a list is owned by a query, perm_a: position in the query image -> query)
          const uint32_t owner = MODE == 0 ? (perm ? perm[v.x] : v.x) : (perm_a ? perm_a[v.x] : v.x), other = perm ? perm[v.y] : v.y;
          append16(cnt, cand, cap, owner, other, queue_s[e]);
        }
      }
    }
    }
    lds_sync();
    qn = 0;
    if (dbg & 8) {
      unsigned long long tf1;
      asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tf1)::"memory");
      t_flush += tf1 - tf0;
    }
  };
  const uint32_t n_norm = (T2 - 1) / 2;
  const uint32_t n_cblocks = (T2 + W - 1) / W;
  const uint32_t total_units = n_cblocks * I_count;

  // DMA pieces (1 KB each) of the stage that starts at packed tile `tile0`: KS per wave (KS <= 8), plus the thresholds of
  // its 256 columns (one piece, wave 7).  part < 0: all of them, else the pieces that go with block `part` of the stage before.
  auto stage = [&](uint32_t tile0, int buf, int part) {
    const f16x8 *src = P + (size_t)tile0 * kTileVecs + lane;
    f16x8 *dst = bufB + (size_t)buf * kStageVecs;
    if (part < 0) {
#pragma unroll
      for (int p = 0; p < KS; ++p)
        __builtin_amdgcn_global_load_lds((gbl_void *)(src + (wv * KS + p) * 64), (lds_void *)(dst + (wv * KS + p) * 64), 16, 0, 0);
    } else if constexpr (KS % kBlocks == 0) {  // KS / 8 pieces with every block
#pragma unroll
      for (int j = 0; j < KS / kBlocks; ++j) {
        const int p = part * (KS / kBlocks) + j;
        __builtin_amdgcn_global_load_lds((gbl_void *)(src + (wv * KS + p) * 64), (lds_void *)(dst + (wv * KS + p) * 64), 16, 0, 0);
      }
    } else if (part < KS) {  // one piece with each of the first KS blocks
      __builtin_amdgcn_global_load_lds((gbl_void *)(src + (wv * KS + part) * 64), (lds_void *)(dst + (wv * KS + part) * 64), 16, 0, 0);
    }
    if (wv == 7 && part <= 0)  // thresholds of the stage's columns (MODE 1: their lambdas)
      __builtin_amdgcn_global_load_lds((gbl_void *)((MODE == 1 ? st.x_lam : theta) + (size_t)tile0 * kTileRows + lane * 4),
                                       (lds_void *)(colthr + buf * 256), 16, 0, 0);
  };

  // diagnostics (as_ctx_set_option("debug", bits)): 2 no listing, 4 no DMA of the next stage, 8 cycle stamps, 128 no test
  auto now = [&]() -> unsigned long long {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
  };
  unsigned long long t_unit = 0, t_blocks = 0, t_dma = 0, t_bar = 0, t_mark = 0;
  unsigned long long n_blk = 0, n_pass = 0;  // 32x32 blocks contracted / blocks that went through the element-wise test
  unsigned long long n_pre = 0;              // blocks whose one-maximum pretest passed (the sign-bit word is formed for these)
  const bool stamp = (dbg & 8) != 0;
  if (stamp) t_mark = now();

  for (;;) {
    __syncthreads();
    if (tid == 0) s_unit = atomicAdd(unit_counter, 1u);
    __syncthreads();
    uint32_t u = s_unit;
    if (unit_list && unit_list[0] * 2u <= total_units) {  // (a list of most units buys nothing: they are drawn in their own order)
      if (u >= unit_list[0]) break;
      u = unit_list[1 + u];
    } else if (u >= total_units) break;
    const uint32_t c = u / I_count, I = I_begin + (u - c * I_count);
    const uint32_t B0 = c * W, Wc = min(W, T2 - B0);
    uint32_t p_lo, p_hi;
    if (MODE == 1) {  // every column super tile of the block, by absolute index
      p_lo = B0;
      p_hi = B0 + Wc;
    } else {
      const bool antipodal = (T2 % 2 == 0) && (I < T2 / 2) && T2 > 1;
      const uint32_t L = 1 + n_norm + (antipodal ? 1u : 0u);  // offsets of this super tile: 0 = diagonal
      const uint32_t delta = B0 >= I ? B0 - I : B0 + T2 - I;  // offset of the block's first super tile
      if (delta + Wc <= T2) {
        p_lo = delta;
        p_hi = min(delta + Wc, L);
      } else {  // the block straddles I: its tail is offsets 0 .. (W <= T2 / 4 keeps the head out of the pattern)
        p_lo = 0;
        p_hi = min(delta + Wc - T2, L);
      }
      if (p_hi <= p_lo) continue;
    }
    uint32_t steps = (p_hi - p_lo) * 2;  // 256-column stages to visit
    if constexpr (MASKED) {  // ... of which only those the first pass flagged (steps <= 2 W <= 128)
      if (tid < 128) {
        unsigned long long b = 0;
        if ((uint32_t)tid < steps) {
          uint32_t S = (MODE == 1 ? 0u : I) + p_lo + ((uint32_t)tid >> 1);
          S = S >= T2 ? S - T2 : S;
          b = mask[(size_t)(I - I_begin) * 2 * T2 + 2 * S + ((uint32_t)tid & 1u)];
        }
        const unsigned long long have = __ballot(b != 0);
        if (lane == 0) s_nlist[wv] = (uint32_t)__builtin_popcountll(have);
        __syncthreads();
        if (b != 0) {
          const uint32_t at = (wv == 1 ? s_nlist[0] : 0u) + lanes_below(have);
          s_step[at] = (unsigned char)tid;
          s_bits[at] = b;
        }
      } else {
        __syncthreads();
      }
      __syncthreads();
      steps = s_nlist[0] + s_nlist[1];
      if (steps == 0) continue;
    }

    // the wave's 64 rows: two groups of 32, A fragments in registers for the whole unit
    f16x8 a0[KS], a1[KS];
    {
      const size_t T = (size_t)4 * I + (wv >> 1);
      const int s0 = 2 * (wv & 1);
      const f16x8 *a_src = (MODE == 1 ? PA : P) + T * kTileVecs + (size_t)s0 * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        a0[ks] = a_src[(size_t)ks * 256];
        a1[ks] = a_src[(size_t)ks * 256 + 64];
      }
    }
    const uint32_t row_base0 = I * kSuperH + wv * 64 + 4 * lh, row_base1 = row_base0 + 32;  // + (r & 3) + 8 * (r >> 2)
    v16f32 nth0, nth1;              // minus the scaled row thresholds: the C operand of a block's first MFMA
    float tmax0 = -__builtin_inff(), tmax1 = -__builtin_inff();  // largest finite scaled threshold of each row group
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float t0 = thr_scaled(theta[row_base0 + (r & 3) + 8 * (r >> 2)]);
      const float t1 = thr_scaled(theta[row_base1 + (r & 3) + 8 * (r >> 2)]);
      nth0[r] = -t0;
      nth1[r] = -t1;
      if (t0 < __builtin_inff()) tmax0 = __builtin_fmaxf(tmax0, t0);
      if (t1 < __builtin_inff()) tmax1 = __builtin_fmaxf(tmax1, t1);
    }
    // (the compiler must not see that nth is minus something it has: it then keeps the thresholds themselves and rebuilds nth with
    // 16 v_xor in front of every block's first MFMA -- as the C operand nth can be read where it sits)
    if constexpr (MASKED && MODE == 0) asm volatile("" : "+v"(nth0), "+v"(nth1));
    tmax0 = __builtin_fmaxf(tmax0, __shfl_xor(tmax0, 32, 64));
    tmax1 = __builtin_fmaxf(tmax1, __shfl_xor(tmax1, 32, 64));
    // (the same in every lane of a group of 32 ... and, as the larger of the two, in the wave: one scalar register)
    tmax0 = tmax1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, __builtin_fmaxf(tmax0, tmax1))));
    // does the wave hold rows that never list (theta = +inf)?  wave-uniform: the listing masks them by hand (emit)
    bool dead0 = false;
    {
      bool d0 = false;
#pragma unroll
      for (int r = 0; r < 16; ++r) d0 = d0 || nth0[r] == -__builtin_inff() || nth1[r] == -__builtin_inff();
      dead0 = __any(d0);
    }
    const bool dead1 = dead0;

    auto tile_of = [&](uint32_t step) -> uint32_t {  // first packed tile of a stage
      uint32_t S = (MODE == 1 ? 0u : I) + p_lo + (step >> 1);
      S = S >= T2 ? S - T2 : S;
      return 4 * S + 2 * (step & 1u);
    };

    // Listing of a block that passed the coarse test.  Exact per-lane tests first (row side: max(acc) >= 0; column side:
    // max(acc + theta_row) >= theta_col); the few lanes that hold a pair (a block of clustered data typically has one)
    // put their 16 values into the wave's LDS slots, four lanes at a time, and the 64 lanes take one (slot, register)
    // entry each: one pass of ballots and queue writes instead of a 16-step walk over the accumulator registers.
    float qlo0 = 0.f, qhi0 = 0.f, qlo1 = 0.f, qhi1 = 0.f;  // MODE 1: lambda range of the rows of the wave's two groups (set below, per unit)
    auto emit = [&](const v16f32 &acc, const v16f32 &nth, uint32_t row_group0, uint32_t colbase, float tc, float tmax, bool dead) -> bool {
      if constexpr (MODE == 0 && MASKED) {
        // A block that gets here (the int8 first pass has removed the rest) holds ~9 pairs on average, dozens inside a cluster,
        // and the listing is written for instruction count.  Per accumulator register: two compares into scalar masks, nothing
        // more if both are empty; else ONE queue entry per pair -- (row, column, score) with the two low mantissa bits of the
        // score saying whose list it goes to (the flush expands them; cos_delta_f16_accum allows for the three ulps) -- written
        // under the mask as exec.  Compares, selects and the predicated LDS writes are hand-issued: from bool expressions the
        // compiler builds every mask twice and spills them.  Rows beyond n never list (theta = +inf makes acc = -inf and
        // acc - nth a NaN), columns beyond n have tc = +inf and z = +inf, the pair (i, i) is dropped by the flush.  When the
        // queue cannot take a register's pairs the walk stops, the queue is flushed (one call site) and the walk resumes there.
        const uint32_t col = colbase + lc, row0 = row_group0 + 4 * lh;
        const float z = col < n ? 0.f : __builtin_inff();
        const uint32_t q_addr = lds_addr(queue), s_addr = lds_addr(queue_s);
        auto cmp_ge = [](float a, float b) -> unsigned long long {
          unsigned long long m;
          asm("v_cmp_ge_f32 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
          return m;
        };
        auto pick = [](unsigned long long m, uint32_t v) -> uint32_t {  // v where the lane's bit of m is set, else 0
          uint32_t r;
          asm("v_cndmask_b32 %0, 0, %2, %1" : "=v"(r) : "s"(m), "v"(v));
          return r;
        };
        // Most blocks hold no pair at all, and ONE maximum per lane settles most of those: a product that reaches its row's
        // threshold has acc >= 0, one that reaches its column's has acc = vr - t_row >= theta_col - t_row >= theta_col - (the
        // largest finite threshold of the group's rows).
        if (!(dbg & 0x800000)) {
          if (!__any(max16(acc) >= __builtin_fminf(0.f, tc - tmax))) return false;
        }
        if (!(dbg & 0x800000)) {  // (debug bit 23: the register-by-register walk below, kept as the form for blocks too dense for the queue)
          // Every lane looks at its own 16 products (one column, 16 rows).  Which of them pass is read off SIGN bits: acc[r] < 0 fails
          // the row's threshold, (acc[r] - nth[r]) - theta_col < 0 the column's, and v_alignbit shifts a sign into a word in one
          // instruction -- two bits per product, 64 instructions for the block where compares, selects and ors took 112.  A wave-wide
          // prefix sum (six DPP adds, no LDS round trip) gives each lane its place in the queue, and it writes its pairs there one
          // register at a time under its own predicate.  (No scalar mask is formed per register: the walk that did -- two compares
          // into SGPR pairs, population counts, a branch and an exec-masked write per register -- spent 9 k cycles per listed
          // block on scalar round trips.)
          n_pre += 1;
          if (dbg & 0x2000000) return false;  // (ablation: the pretest alone)
          uint32_t w = 0;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float d2 = (acc[r] - nth[r]) - tc;
            w = __builtin_amdgcn_alignbit(w, __float_as_uint(d2), 31);      // ... so register r ends at bits 2 (15 - r) + 1: the column's test,
            w = __builtin_amdgcn_alignbit(w, __float_as_uint(acc[r]), 31);  //     2 (15 - r): the row's
          }
          uint32_t pass = col < n ? ~w : 0u;  // (a column beyond n: theta_col = +inf has failed its side already, the row side must not list it)
          if (dead) {  // rows that never list (theta = +inf: padding) hold acc = -inf and acc - nth = NaN, whose sign says nothing
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (nth[r] == -__builtin_inff()) pass &= ~(3u << (2 * (15 - r)));
          }
          if (!__any(pass != 0u)) return false;
          if (dbg & 0x4000000) return true;  // (ablation: pretest + the sign-bit word, nothing queued)
          const uint32_t c = (uint32_t)__builtin_popcount((pass | (pass >> 1)) & 0x55555555u);
          uint32_t inc = c;  // inclusive prefix sum over the wave: inside rows of 16 lanes, then across them
          inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xF, 0xF, true);  // row_shr:1
          inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xF, 0xF, true);  // row_shr:2
          inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xF, 0xF, true);  // row_shr:4
          inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xF, 0xF, true);  // row_shr:8
          inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x142, 0xA, 0xF, true);  // row_bcast:15 into rows 1 and 3
          inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x143, 0xC, 0xF, true);  // row_bcast:31 into rows 2 and 3
          const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
          if (tot <= kQueueH) {
            // (flushed while the queue still fits ONE batch of the flush's rounds -- one exposed round trip of returning atomics per
            // flush instead of two; only a block with more pairs than that fills the queue beyond)
            if (dbg & 0x8000000) return true;  // (ablation: ... + the prefix sums, nothing queued)
            if (qn && qn + tot > 64u * (uint32_t)kFlushRounds) flush_queue();
            // (the two LDS writes of an entry are issued by hand: in front of a write of its own the compiler waits for every
            // pending vector memory operation -- vmcnt(0): the scattered stores of the flush before, the LDS-DMA of the next stage)
            const uint32_t o = qn + inc - c;
            uint32_t qa = q_addr + 8u * o, sa = s_addr + 4u * o;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const uint32_t f = (pass >> (2 * (15 - r))) & 3u;  // bit 0: listed for the row, bit 1: for the column (what the flush reads)
              if (f) {
                const float vr = acc[r] - nth[r];
                const uint32_t sc = (__float_as_uint(vr * (1.f / kProductScale)) & ~3u) | f;
                const uint32_t row = row0 + (r & 3) + 8 * (r >> 2);
                asm volatile(
                    "ds_write2_b32 %0, %2, %3 offset1:1\n\t"
                    "ds_write_b32 %1, %4\n\t"
                    "v_add_u32 %0, 8, %0\n\t"
                    "v_add_u32 %1, 4, %1"
                    : "+v"(qa), "+v"(sa)
                    : "v"(row), "v"(col), "v"(sc)
                    : "memory");
              }
            }
            qn += tot;
            lds_sync();
            return true;
          }
        }
        uint32_t resume = 0;
        bool listed = false;
        for (;;) {
          bool full = false;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (full || (uint32_t)r < resume) continue;
            const float vr = acc[r] - nth[r];
            const unsigned long long m1 = cmp_ge(acc[r], z), m2 = cmp_ge(vr, tc), mm = m1 | m2;
            if (mm == 0) continue;
            const uint32_t cnt_r = (uint32_t)__builtin_popcountll(mm);
            if (qn + cnt_r > kQueueH) {
              full = true;
              resume = (uint32_t)r;
              continue;
            }
            const uint32_t o = qn + lanes_below(mm);
            const uint32_t sb = (__float_as_uint(vr * (1.f / kProductScale)) & ~3u) | pick(m1, 1u) | pick(m2, 2u);
            const unsigned long long rc = ((unsigned long long)col << 32) | (row0 + (r & 3) + 8 * (r >> 2));
            unsigned long long saved;
            asm volatile(
                "s_and_saveexec_b64 %0, %1\n\t"
                "ds_write_b64 %2, %3\n\t"
                "ds_write_b32 %4, %5\n\t"
                "s_mov_b64 exec, %0"
                : "=&s"(saved)
                : "s"(mm), "v"(q_addr + 8u * o), "v"(rc), "v"(s_addr + 4u * o), "v"(sb)
                : "memory");
            qn += cnt_r;
            listed = true;
          }
          if (!full) break;
          flush_queue();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        lds_sync();
        return listed;
      } else {
      v16f32 v;
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[r] - nth[r];
      bool hit = max16(acc) >= 0.f || (MODE == 0 && max16(v) >= tc);
      if constexpr (MODE == 1) {
        // The cosine alone reaching its floor says little where the lambda term matters: half of all blocks of a 10 k-query batch
        // got here, 9 % hold a pair.  A pair passes the walk below when alpha * cos - pen >= g - beta with pen = min(|beta lq -
        // beta lx|, beta), i.e. alpha * (cos - floor) >= pen, and pen >= min(beta * dist(lx, [lo, hi]), beta) for every row of the
        // group: one more compare per lane, on the maximum it has already taken.  (4e-6 x magnitudes covers the roundings between
        // this form and the walk's; the test only decides who is walked.)  It bites when the batch is in the order of its lambdas.
        const bool g1 = ((row_group0 >> 5) & 1u) != 0;
        const float lo = g1 ? qlo1 : qlo0, hi = g1 ? qhi1 : qhi0;
        const float dist = __builtin_fmaxf(0.f, __builtin_fmaxf(lo - tc, tc - hi));
        const float pen = __builtin_fminf(st.beta * dist, st.beta);
        const float slack = 4e-6f * (1.f + st.beta * (__builtin_fabsf(tc) + __builtin_fmaxf(__builtin_fabsf(lo), __builtin_fabsf(hi))));
        if (!(dbg & 0x400000)) hit = hit && (max16(acc) * (st.alpha * (1.f / kProductScale)) + slack >= pen);  // (debug bit 22: without the bound)
      }
      unsigned long long m = __ballot(hit);
      if (m == 0) return false;
      do {
        const uint32_t rank = lanes_below(m);
        if (hit && ((m >> lane) & 1ull) && rank < 4) {
          float *sl = ex + rank * kSlotFloats;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            reinterpret_cast<float4 *>(sl)[q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            reinterpret_cast<float4 *>(sl)[4 + q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
          }
          sl[32] = __uint_as_float((uint32_t)lane);
          sl[33] = tc;
        }
        lds_sync();
        const uint32_t filled = min(4u, (uint32_t)__builtin_popcountll(m));
        const uint32_t slot = (uint32_t)lane >> 4, r = (uint32_t)lane & 15u;
        const float *sl = ex + slot * kSlotFloats;
        const bool valid = slot < filled;
        const float ap = sl[r], vv = sl[16 + r], tcs = sl[33];
        const uint32_t src = __float_as_uint(sl[32]);
        const uint32_t row = row_group0 + 4 * (src >> 5) + (r & 3) + 8 * (r >> 2), col = colbase + (src & 31);
        bool inside = valid && col < n;
        float sc = vv * (1.f / kProductScale);
        bool w1, w2;
        if (MODE == 1) {  // tcs = lambda of the column; the query's lambda and g - beta from the unit's arrays
          const uint32_t rl = row - I * kSuperH;
          inside = inside && row < st.nq;
          if (st.self_ids && inside) inside = st.self_ids[row] != (perm ? perm[col] : col);
          sc = __builtin_fmaf(st.alpha, sc, -__builtin_fminf(__builtin_fabsf(__builtin_fmaf(st.beta, s_qlam[rl & (kSuperH - 1)], -st.beta * tcs)), st.beta));
          w1 = inside && ap >= 0.f && sc >= s_qg[rl & (kSuperH - 1)];
          w2 = false;
        } else {
          inside = inside && row < n && row != col;
          w1 = inside && ap >= 0.f;
          w2 = inside && vv >= tcs;
        }
        const unsigned long long m1 = __ballot(w1), m2 = __ballot(w2);
        const uint32_t n1 = (uint32_t)__builtin_popcountll(m1), n2 = (uint32_t)__builtin_popcountll(m2);
        if (qn + n1 + n2 > kQueueH) flush_queue();
        if (w1) {
          queue[qn + lanes_below(m1)] = make_uint2(row, col);
          queue_s[qn + lanes_below(m1)] = sc;
        }
        if (w2) {
          queue[qn + n1 + lanes_below(m2)] = make_uint2(col, row);
          queue_s[qn + n1 + lanes_below(m2)] = sc;
        }
        qn += n1 + n2;
        lds_sync();
        m &= m - 1;  // the four lowest lanes are done
        m &= m - 1;
        m &= m - 1;
        m &= m - 1;
      } while (m);
      return true;
      }
    };

    f16x8 fa[4];
    __syncthreads();
    if (MODE == 1) {  // (no DMA is pending here)
      s_qlam[tid] = st.q_lam[(size_t)I * kSuperH + tid];
      s_qg[tid] = st.q_g[(size_t)I * kSuperH + tid];
    }
    auto step_at = [&](uint32_t li) -> uint32_t { return MASKED ? __builtin_amdgcn_readfirstlane((uint32_t)s_step[li]) : li; };
    stage(tile_of(step_at(0)), 0, -1);
    __syncthreads();
    if (MODE == 1) {  // the lambda range of the wave's two row groups (rows beyond the batch left out): what emit() bounds the lambda term by
      const uint32_t my_row = I * kSuperH + (uint32_t)wv * 64u + (uint32_t)lane;
      const bool live = my_row < st.nq;
      const float ql = s_qlam[wv * 64 + lane];
      float lo = live ? ql : __builtin_inff(), hi = live ? ql : -__builtin_inff();
      // (a floor below -1.5 enters the accumulator clamped, thr_scaled: cos - floor is then not what the accumulator holds, and
      // the bound does not apply to a group with such a row -- queries whose lambda term is at its floor for every item)
      const unsigned long long clamped = __ballot(live && !(theta[my_row] >= -1.5f));
      if (clamped & (lane < 32 ? 0xFFFFFFFFull : 0xFFFFFFFF00000000ull)) {
        lo = -__builtin_inff();
        hi = __builtin_inff();
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        lo = __builtin_fminf(lo, __shfl_xor(lo, o, 64));
        hi = __builtin_fmaxf(hi, __shfl_xor(hi, o, 64));
      }
      qlo0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lo), 0));
      qhi0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hi), 0));
      qlo1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lo), 32));
      qhi1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hi), 32));
    }
    if (stamp) {
      const unsigned long long t = now();
      t_unit += t - t_mark;
      t_mark = t;
    }
    for (uint32_t li = 0; li < steps; ++li) {
      const uint32_t step = step_at(li);
      const uint32_t my_blocks = MASKED ? __builtin_amdgcn_readfirstlane((uint32_t)(s_bits[li] >> (8 * wv)) & 0xFFu) : 0xFFu;
      const uint32_t tile0 = tile_of(step);
      const int pbuf = (int)(li & 1u);
      const bool diag = MODE == 0 && (p_lo + (step >> 1)) == 0;
      const bool more = li + 1 < steps && !(dbg & 4);
      const uint32_t tile_next = more ? tile_of(step_at(li + 1)) : 0;
      const uint32_t b_addr = lds_addr(bufB + (size_t)pbuf * kStageVecs) + (uint32_t)lane * 16u;
      const float *tcol = colthr + pbuf * 256 + lc;  // thresholds of this stage's columns (LDS)

      // The 8 blocks of the stage, as a real loop (one copy of the listing code: unrolled, its rare path costs
      // hundreds of registers in live ranges).  Fragments are requested three k steps ahead, across block boundaries;
      // the last block requests into the bytes behind the stage, which nobody reads.
      // kRing (KS a multiple of 4): the ring of four fragment registers runs across the block boundaries.  Other widths
      // restart it at every block (one exposed LDS latency per block; not the benchmark shapes).
      constexpr bool kRing = KS % 4 == 0 && !MASKED;
      uint32_t blk = b_addr;  // fragment base of the block
      if constexpr (kRing) {
        lds_read16<0 * 4 * 64 * 16>(fa[0], blk);
        lds_read16<1 * 4 * 64 * 16>(fa[1], blk);
        lds_read16<2 * 4 * 64 * 16>(fa[2], blk);
      }
      // MASKED: a wave that lists something issues its share of the next stage behind its blocks -- the compiler waits for
      // every pending LDS-DMA (vmcnt(0)) in front of the LDS accesses of the listing, which with the pieces in flight cost
      // 4 000 cycles per listed block; a wave without blocks in this stage issues at once
      if (MASKED && more && my_blocks == 0) stage(tile_next, pbuf ^ 1, -1);
#pragma nounroll
      for (int nn = 0; nn < kBlocks; ++nn) {
        // fragment base of the next block: sub-tile nn + 1 of the stage (4 sub-tiles of 32 columns per packed tile)
        const uint32_t nxt = b_addr + (uint32_t)((((nn + 1) >> 2) * kTileVecs + ((nn + 1) & 3) * 64) * 16);
        if (!MASKED && more && wv < 4) stage(tile_next, pbuf ^ 1, nn);
        if constexpr (MASKED) {
          // only the blocks the first pass flagged for this wave (its share of the DMA goes out either way); the two row groups
          // one after the other through one set of accumulators (the listing below wants the registers)
          if ((my_blocks >> nn) & 1u) {
            const uint32_t colbase = tile0 * kTileRows + nn * 32;
            const float tc = MODE == 1 ? tcol[nn * 32] /* lambda of the lane's column */
                                       : ((diag || colbase + lc >= n) ? __builtin_inff() : thr_scaled(tcol[nn * 32]));
            auto half = [&](const f16x8(&a)[KS], const v16f32 &nth, float tmax, uint32_t row_base, bool dead) {
              lds_read16<0>(fa[0], blk);
              if constexpr (KS > 1) lds_read16<1 * 4 * 64 * 16>(fa[1], blk);
              if constexpr (KS > 2) lds_read16<2 * 4 * 64 * 16>(fa[2], blk);
              v16f32 x;
              static_for<0, KS>([&](auto ks_c) {
                constexpr int ks = decltype(ks_c)::value;
                constexpr int ahead = ks + 3;
                if constexpr (ahead < KS) lds_read16<ahead * 4 * 64 * 16>(fa[ahead & 3], blk);
                constexpr int pending = (KS - 1 - ks) < 3 ? (KS - 1 - ks) : 3;
                lds_wait1<pending>(fa[ks & 3]);
                x = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], fa[ks & 3], ks == 0 ? nth : x, 0, 0, 0);
              });
              n_blk += 1;
              if (!(dbg & 128) && !(dbg & 2)) {  // (no coarse test: the first pass was one)
                const unsigned long long te0 = stamp ? now() : 0;
                n_pass += emit(x, nth, row_base - 4 * lh, colbase, tc, tmax, dead) ? 1 : 0;
                if (stamp) t_emit += now() - te0;
              }
            };
            half(a0, nth0, tmax0, row_base0, dead0);
            half(a1, nth1, tmax1, row_base1, dead1);
          }
          blk = nxt;
          continue;
        }
        if constexpr (!kRing) {
          lds_read16<0>(fa[0], blk);
          if constexpr (KS > 1) lds_read16<1 * 4 * 64 * 16>(fa[1], blk);
          if constexpr (KS > 2) lds_read16<2 * 4 * 64 * 16>(fa[2], blk);
        }
        v16f32 x0, x1;
        static_for<0, KS>([&](auto ks_c) {
          constexpr int ks = decltype(ks_c)::value;
          constexpr int ahead = ks + 3;  // k step requested now: of this block, or (ring) of the next one
          if constexpr (ahead < KS) lds_read16<ahead * 4 * 64 * 16>(fa[ahead & 3], blk);
          else if constexpr (kRing) lds_read16<(ahead - KS) * 4 * 64 * 16>(fa[ahead & 3], nxt);
          constexpr int pending = kRing ? 3 : ((KS - 1 - ks) < 3 ? (KS - 1 - ks) : 3);
          lds_wait1<pending>(fa[ks & 3]);
          x0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[ks], fa[ks & 3], ks == 0 ? nth0 : x0, 0, 0, 0);
          x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[ks], fa[ks & 3], ks == 0 ? nth1 : x1, 0, 0, 0);
        });
        if (more && wv >= 4) stage(tile_next, pbuf ^ 1, nn);
        // coarse test: some product reaches its row's threshold (max(acc) >= 0) or may reach its column's (conservative:
        // theta_col minus the largest row threshold of the group)
        n_blk += 2;
        if (!(dbg & 128)) {
          const uint32_t colbase = tile0 * kTileRows + nn * 32;
          const float tc = MODE == 1 ? tcol[nn * 32] /* lambda of the lane's column */
                                     : ((diag || colbase + lc >= n) ? __builtin_inff() : thr_scaled(tcol[nn * 32]));
          const bool h0 = MODE == 1 ? max16(x0) >= 0.f : max16(x0) >= __builtin_fminf(0.f, tc - tmax0);
          const bool h1 = MODE == 1 ? max16(x1) >= 0.f : max16(x1) >= __builtin_fminf(0.f, tc - tmax1);
          if (__any(h0 || h1) && !(dbg & 2)) {
            if (__any(h0)) n_pass += emit(x0, nth0, row_base0 - 4 * lh, colbase, tc, tmax0, dead0) ? 1 : 0;
            if (__any(h1)) n_pass += emit(x1, nth1, row_base1 - 4 * lh, colbase, tc, tmax1, dead1) ? 1 : 0;
          }
        }
        blk = nxt;
      }
      if (MASKED && more && my_blocks != 0) stage(tile_next, pbuf ^ 1, -1);
      if (stamp) {
        unsigned long long t = now();
        t_blocks += t - t_mark;
        t_mark = t;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t = now();
        t_dma += t - t_mark;
        t_mark = t;
      }
      __syncthreads();  // the DMA of the next stage has landed; everyone is done reading bufB[pbuf]
      if (stamp) {
        const unsigned long long t = now();
        t_bar += t - t_mark;
        t_mark = t;
      }
    }
  }
  flush_queue();
  if (lane == 0) {
    atomicAdd(stamps + 5, n_blk);
    atomicAdd(stamps + 7, n_pass);
    if (!stamp && n_pre) atomicAdd(stamps + 1, n_pre);
  }
  if (stamp && lane == 0) {
    atomicAdd(stamps + 0, t_unit);
    atomicAdd(stamps + 1, t_blocks);
    atomicAdd(stamps + 2, t_dma);
    atomicAdd(stamps + 3, t_bar);
    atomicAdd(stamps + 4, 1ull);
    atomicAdd(stamps + 6, t_emit);
    atomicAdd(stamps + 8, t_flush);
  }
}


// ---- the one-sided masked scan, a wave on its own (round 6) -------------------------------------------------------------------------
// MODE 1 of pair_join_f16_kernel behind the rectangular first pass shares a stage of 256 columns among the eight waves of a workgroup
// (64 queries each) through LDS, a barrier a stage.  With the batch in the order of the items' image the eight slices of a super tile
// look at DIFFERENT columns: in a stage one or two waves hold flagged blocks and the others wait -- cycle stamps of a 10 000-query
// batch at 1 M items: 1.1 M cycles in the blocks, 3.8 M at the barriers.  Here nothing is shared: a work unit is (a slice of 64
// queries, G stages of 256 columns), drawn by ONE wave from the list of the units that hold a flagged block (scan_slice_units_kernel);
// the wave keeps its queries' fragments in registers, asks for the eight k-step fragments of a flagged block straight from the image
// (1 KB a fragment, contiguous; the next block's are on their way while this one is contracted) and lists through its own queue.
// No barrier, no LDS staging; the image is read through the L2 once per slice that needs a block instead of once per super tile.
__global__ __launch_bounds__(256) void scan_slice_flags_kernel(const unsigned long long *__restrict__ mask, uint32_t T2, uint32_t a_supers, uint32_t G,
                                                               uint32_t *__restrict__ flags) {
  // G: stages of 256 columns per unit.  flags[u] = 1 where unit u = slice * n_cblocks + c holds a flagged block
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n_cblocks = (2 * T2 + G - 1) / G, total = n_cblocks * a_supers * 8u;
  if (u >= total) return;
  const uint32_t slice = u / n_cblocks, c = u - slice * n_cblocks, I = slice >> 3, w = slice & 7u;
  const uint32_t s0 = c * G, s1 = min(2 * T2, s0 + G);
  bool keep = false;
  for (uint32_t t = s0; t < s1 && !keep; ++t) keep = ((mask[(size_t)I * 2 * T2 + t] >> (8 * w)) & 0xFFull) != 0;
  flags[u] = keep ? 1u : 0u;
}
// ... and the list of them IN ORDER (list[0] = how many): consecutive entries are neighbours of one slice, which is what lets a wave
// draw a run of units and keep its queries' fragments (off: the exclusive scan of flags, off[total] = the count)
__global__ __launch_bounds__(256) void scan_slice_list_kernel(const uint32_t *__restrict__ flags, const uint64_t *__restrict__ off, uint32_t total,
                                                              uint32_t *__restrict__ list) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u == 0) list[0] = (uint32_t)off[total];
  if (u < total && flags[u]) list[1 + off[u]] = u;
}

constexpr int kSliceWaves = 4;  // waves of a workgroup of the scan below (they share nothing but the workgroup)
template <int KS>
__global__ __launch_bounds__(64 * kSliceWaves, 2) void scan_slices_f16_kernel(const f16x8 *__restrict__ P, const f16x8 *__restrict__ PA, uint32_t T2,
                                                                              uint32_t a_supers, uint32_t G, uint32_t n, const float *__restrict__ theta,
                                                                              uint32_t cap, uint32_t *__restrict__ cnt, uint32_t *__restrict__ cand,
                                                                              uint32_t *__restrict__ unit_counter, F16SearchTerm st,
                                                                              const uint32_t *__restrict__ perm, const unsigned long long *__restrict__ mask,
                                                                              const uint32_t *__restrict__ perm_a, const uint32_t *__restrict__ unit_list,
                                                                              unsigned long long *__restrict__ stamps, uint32_t run_max) {
  constexpr int kTileVecs = KS * 4 * 64;
  constexpr int kSlotFloats = 36;
  __shared__ uint2 s_queue[kSliceWaves][kQueueSearch];
  __shared__ float s_score[kSliceWaves][kQueueSearch];
  __shared__ __attribute__((aligned(16))) float s_ex[kSliceWaves][4 * kSlotFloats];
  __shared__ float s_qlam[kSliceWaves][64], s_qg[kSliceWaves][64];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lh = lane >> 5, lc = lane & 31;
  uint2 *queue = s_queue[wv];
  float *queue_s = s_score[wv], *ex = s_ex[wv], *qlam = s_qlam[wv], *qg = s_qg[wv];
  uint32_t qn = 0;  // wave-uniform
  auto lanes_below = [&](unsigned long long m) -> uint32_t {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  };
  auto flush_queue = [&]() {
    lds_sync();
    for (uint32_t base = 0; base < qn; base += 64) {
      const uint32_t e = base + lane;
      if (e < qn) {
        const uint2 v = queue[e];
        append16(cnt, cand, cap, perm_a ? perm_a[v.x] : v.x, perm ? perm[v.y] : v.y, queue_s[e]);
      }
    }
    lds_sync();
    qn = 0;
  };
  const uint32_t n_cblocks = (2 * T2 + G - 1) / G;
  const uint32_t n_units = unit_list[0];
  unsigned long long n_blk = 0, n_pass = 0;
  uint32_t cur_slice = 0xFFFFFFFFu;
  f16x8 a0[KS], a1[KS];
  v16f32 nth0, nth1;
  float qlo0 = 0.f, qhi0 = 0.f, qlo1 = 0.f, qhi1 = 0.f;
  uint32_t row0 = 0;  // first query position of the slice

  // the listing of pair_join_f16_kernel's search mode (the same tests, the same scores), the slice's lambdas from the wave's own arrays
  auto emit = [&](const v16f32 &acc, const v16f32 &nth, uint32_t row_group0, bool g1, uint32_t colbase, float tc) -> bool {
    v16f32 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[r] - nth[r];
    const float top = max16(acc);
    bool hit = top >= 0.f;
    {
      const float lo = g1 ? qlo1 : qlo0, hi = g1 ? qhi1 : qhi0;
      const float dist = __builtin_fmaxf(0.f, __builtin_fmaxf(lo - tc, tc - hi));
      const float pen = __builtin_fminf(st.beta * dist, st.beta);
      const float slack = 4e-6f * (1.f + st.beta * (__builtin_fabsf(tc) + __builtin_fmaxf(__builtin_fabsf(lo), __builtin_fabsf(hi))));
      hit = hit && (top * (st.alpha * (1.f / kProductScale)) + slack >= pen);
    }
    unsigned long long m = __ballot(hit);
    if (m == 0) return false;
    do {
      const uint32_t rank = lanes_below(m);
      if (hit && ((m >> lane) & 1ull) && rank < 4) {
        float *sl = ex + rank * kSlotFloats;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          reinterpret_cast<float4 *>(sl)[q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
          reinterpret_cast<float4 *>(sl)[4 + q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        }
        sl[32] = __uint_as_float((uint32_t)lane);
        sl[33] = tc;
      }
      lds_sync();
      const uint32_t filled = min(4u, (uint32_t)__builtin_popcountll(m));
      const uint32_t slot = (uint32_t)lane >> 4, r = (uint32_t)lane & 15u;
      const float *sl = ex + slot * kSlotFloats;
      const bool valid = slot < filled;
      const float ap = sl[r], vv = sl[16 + r], tcs = sl[33];
      const uint32_t src = __float_as_uint(sl[32]);
      const uint32_t row = row_group0 + 4 * (src >> 5) + (r & 3) + 8 * (r >> 2), col = colbase + (src & 31);
      bool inside = valid && col < n && row < st.nq;
      if (st.self_ids && inside) inside = st.self_ids[row] != (perm ? perm[col] : col);
      const uint32_t rl = (row - row0) & 63u;
      float sc = vv * (1.f / kProductScale);
      sc = __builtin_fmaf(st.alpha, sc, -__builtin_fminf(__builtin_fabsf(__builtin_fmaf(st.beta, qlam[rl], -st.beta * tcs)), st.beta));
      const bool w1 = inside && ap >= 0.f && sc >= qg[rl];
      const unsigned long long m1 = __ballot(w1);
      const uint32_t n1 = (uint32_t)__builtin_popcountll(m1);
      if (qn + n1 > kQueueSearch) flush_queue();
      if (w1) {
        queue[qn + lanes_below(m1)] = make_uint2(row, col);
        queue_s[qn + lanes_below(m1)] = sc;
      }
      qn += n1;
      lds_sync();
      m &= m - 1;  // the four lowest lanes are done
      m &= m - 1;
      m &= m - 1;
      m &= m - 1;
    } while (m);
    return true;
  };

  // a run of units per draw: one returning atomic, one read of the list and one of the mask for all of them (three dependent round
  // trips a run instead of a unit -- most units hold one or two blocks), and a run is nearly always of one slice
  const uint32_t run = min(run_max, 64u / G);
  // (drawn in the order of the list from ONE counter.  Measured at 10 000 queries x 1 M items, 41 600 units: 1.04 ms; dealt round
  // robin without a counter 1.27 -- units differ by a factor of thirty in what they hold --, drawn from 16 counters over interleaved
  // parts of the list 1.27 as well; runs of 2 / 4 / 8 units a draw 1.16 / 1.46 / 2.13: the heavy units of a slice sit side by side)
  for (;;) {
    uint32_t u0 = 0;
    if (lane == 0) u0 = atomicAdd(unit_counter, run);
    u0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)u0);
    if (u0 >= n_units) break;
    const uint32_t n_here = min(run, n_units - u0);
    const uint32_t k_l = (uint32_t)lane / G, t_l = (uint32_t)lane - k_l * G;  // lane -> (unit of the run, stage of the unit)
    uint32_t unit_l = 0, bits_l = 0;
    if (k_l < n_here) {
      unit_l = unit_list[1 + u0 + k_l];
      const uint32_t slice_l = unit_l / n_cblocks, c_l = unit_l - slice_l * n_cblocks, sg = c_l * G + t_l;
      if (sg < 2 * T2) bits_l = (uint32_t)(mask[(size_t)(slice_l >> 3) * 2 * T2 + sg] >> (8 * (slice_l & 7u))) & 0xFFu;
    }
    const unsigned long long have_all = __ballot(bits_l != 0);
   for (uint32_t k = 0; k < n_here; ++k) {
    const uint32_t unit = (uint32_t)__builtin_amdgcn_readlane((int)unit_l, (int)(k * G));
    const uint32_t slice = unit / n_cblocks, c = unit - slice * n_cblocks, I = slice >> 3, w = slice & 7u;
    const uint32_t s0 = c * G;
    unsigned long long have = (have_all >> (k * G)) & ((G >= 64u) ? ~0ull : ((1ull << G) - 1ull));
    const uint32_t my_bits = bits_l;  // (stage t of the unit sits in lane k G + t)
    const uint32_t lane0 = k * G;
    if (have == 0) continue;
    if (slice != cur_slice) {  // the slice's 64 queries: fragments, thresholds, lambdas
      cur_slice = slice;
      row0 = I * kSuperH + w * 64u;
      const size_t T = (size_t)4 * I + (w >> 1);
      const f16x8 *a_src = PA + T * kTileVecs + (size_t)(2 * (w & 1u)) * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        a0[ks] = a_src[(size_t)ks * 256];
        a1[ks] = a_src[(size_t)ks * 256 + 64];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        nth0[r] = -thr_scaled(theta[row0 + 4 * lh + (r & 3) + 8 * (r >> 2)]);
        nth1[r] = -thr_scaled(theta[row0 + 32 + 4 * lh + (r & 3) + 8 * (r >> 2)]);
      }
      lds_sync();  // (the listing of the unit before is done with the arrays)
      const float ql = st.q_lam[row0 + lane];
      qlam[lane] = ql;
      qg[lane] = st.q_g[row0 + lane];
      const bool live = row0 + (uint32_t)lane < st.nq;
      float lo = live ? ql : __builtin_inff(), hi = live ? ql : -__builtin_inff();
      // (a floor below -1.5 enters the accumulator clamped: the lambda bound does not apply to a group with such a row)
      const unsigned long long clamped = __ballot(live && !(theta[row0 + lane] >= -1.5f));
      if (clamped & (lane < 32 ? 0xFFFFFFFFull : 0xFFFFFFFF00000000ull)) {
        lo = -__builtin_inff();
        hi = __builtin_inff();
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        lo = __builtin_fminf(lo, __shfl_xor(lo, o, 64));
        hi = __builtin_fmaxf(hi, __shfl_xor(hi, o, 64));
      }
      qlo0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lo), 0));
      qhi0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hi), 0));
      qlo1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lo), 32));
      qhi1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hi), 32));
      lds_sync();
    }
    // the flagged blocks of the unit in order; the fragments of the next one are asked for before this one is contracted
    uint32_t st_i = (uint32_t)__builtin_ctzll(have), byte = (uint32_t)__builtin_amdgcn_readlane((int)my_bits, (int)(lane0 + st_i));
    have &= have - 1;
    auto block_at = [&](uint32_t stage_i, uint32_t nn, uint32_t *colbase) -> const f16x8 * {
      const uint32_t step = s0 + stage_i, tile0 = 4 * (step >> 1) + 2 * (step & 1u);
      *colbase = tile0 * kTileRows + nn * 32;
      return P + ((size_t)(tile0 + (nn >> 2)) * KS * 4 + (nn & 3u)) * 64 + lane;
    };
    f16x8 fb[KS];
    float tc;
    uint32_t colbase;
    {
      const uint32_t nn = (uint32_t)__builtin_ctz(byte);
      byte &= byte - 1;
      const f16x8 *src = block_at(st_i, nn, &colbase);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) fb[ks] = src[(size_t)ks * 256];
      tc = st.x_lam[colbase + lc];
    }
    for (;;) {
      bool more = true;
      if (byte == 0) {
        if (have == 0) more = false;
        else {
          st_i = (uint32_t)__builtin_ctzll(have);
          byte = (uint32_t)__builtin_amdgcn_readlane((int)my_bits, (int)(lane0 + st_i));
          have &= have - 1;
        }
      }
      f16x8 fn[KS];
      float tcn = 0.f;
      uint32_t colbase_n = 0;
      if (more) {
        const uint32_t nn = (uint32_t)__builtin_ctz(byte);
        byte &= byte - 1;
        const f16x8 *src = block_at(st_i, nn, &colbase_n);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) fn[ks] = src[(size_t)ks * 256];
        tcn = st.x_lam[colbase_n + lc];
      }
      {
        v16f32 x;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) x = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[ks], fb[ks], ks == 0 ? nth0 : x, 0, 0, 0);
        n_pass += emit(x, nth0, row0, false, colbase, tc) ? 1 : 0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) x = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[ks], fb[ks], ks == 0 ? nth1 : x, 0, 0, 0);
        n_pass += emit(x, nth1, row0 + 32, true, colbase, tc) ? 1 : 0;
        n_blk += 2;
      }
      if (!more) break;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) fb[ks] = fn[ks];
      tc = tcn;
      colbase = colbase_n;
    }
   }
  }
  flush_queue();
  if (lane == 0 && stamps) {
    atomicAdd(stamps + 5, n_blk);
    atomicAdd(stamps + 7, n_pass);
  }
}

}  // namespace

// 16-B vectors of an f16 image with `tiles` 128-row tiles
uint64_t packed_vecs_f16(uint32_t tiles, uint32_t d) { return (uint64_t)tiles * f16_ksteps(d) * 4 * 64; }
// tiles of the f16 image of n rows: whole 512-row super tiles
uint32_t packed_tiles_f16(uint64_t n) { return (uint32_t)((n + kSuperH - 1) / kSuperH) * 4; }

int pack_f16(as_ctx *ctx, const double *R, const double *norms, const uint32_t *row_ids, uint64_t row0, uint64_t n, uint32_t d,
             void *P, uint32_t tiles) {
  const uint32_t KS = f16_ksteps(d);
  const uint64_t total = packed_vecs_f16(tiles, d);  // one thread per fragment lane
  ProfScope ps(ctx, "pack_unit_rows");
  hipLaunchKernelGGL(pack_f16_kernel, dim3(cdiv(total, 256)), dim3(256), 0, ctx->stream, R, norms, row_ids, row0, n, d, KS, total,
                     (f16x8 *)P);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

// E = max over the rows of || stored row / 128 - unit row ||_2, the size of what the f16 plane lost (one wave per row,
// the same arithmetic as the packer, differences taken in f64).  For unit rows a, b with stored rows a + ea, b + eb:
// |stored product - cos| = |a.eb + ea.b + ea.eb| <= 2 E + E^2 (Cauchy-Schwarz), about half the worst case of 2^-11 per
// element that cos_delta_f16 assumes: rounding errors of 128 elements do not all sit at the edge of their binade.
__global__ __launch_bounds__(256) void f16_row_error_kernel(const double *__restrict__ R, const double *__restrict__ norms, uint64_t n,
                                                            uint32_t d, unsigned int *__restrict__ out_bits, const uint32_t *__restrict__ perm,
                                                            float *__restrict__ err_rows) {
  // (perm / err_rows: the error of the row at image position p = perm[p] goes to err_rows[p])
  const int lane = threadIdx.x & 63;
  float worst = 0.f;
  for (uint64_t pos = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); pos < n; pos += (uint64_t)gridDim.x * 4) {
    const uint64_t row = perm ? (uint64_t)perm[pos] : pos;
    const double nr = norms[row];
    double e2 = 0.0;
    if (nr > 0.0 && nr < __builtin_inf()) {
      for (uint32_t k = lane; k < d; k += 64) {
        const double u = R[row * d + k] / nr;
        const float x = (float)u * kOperandScale;
        const _Float16 h = (x == x && __builtin_fabsf(x) <= 2.f * kOperandScale) ? (_Float16)x : (_Float16)0.0f;
        const double e = (double)(float)h * (1.0 / kOperandScale) - u;
        e2 += e * e;
      }
    }
    for (int o = 32; o > 0; o >>= 1) e2 += __shfl_xor(e2, o, 64);
    const float e = (float)(__builtin_sqrt(e2) * (1.0 + 1e-6)) ;  // rounded up
    if (e == e) worst = __builtin_fmaxf(worst, e);
    if (err_rows && lane == 0) err_rows[pos] = e == e ? e : __builtin_inff();
  }
  if (lane == 0) raise_max_u32(out_bits, __float_as_uint(worst));  // non-negative floats order like their bits
}

// packs ALL n rows and leaves the measured plane error in ctx->f16_err (what join_delta certifies against)
// (perm: image position -> row, or nullptr)
// (err_rows: the plane error of every row by image position, or nullptr; paths that do not measure it per row fill in the largest)
__global__ void fill_f32_kernel(float *__restrict__ p, uint64_t n, float v) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
int pack_f16_all(as_ctx *ctx, const double *R, const double *norms, uint64_t n, uint32_t d, void *P, uint32_t tiles,
                 const uint32_t *perm, float *err_rows) {
  DevBuf<unsigned int> bits;
  AS_TRY(bits.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(bits.p, 0, sizeof(unsigned int), ctx->stream));
  const uint32_t KS = f16_ksteps(d);
  ctx->stats["knn_join_pack_wide_tiles"] = 0;
  if (d == 128 && ((uintptr_t)R & 15) == 0) {
    ProfScope ps(ctx, "pack_unit_rows");
    hipLaunchKernelGGL(pack_f16_tile128_kernel<false>, dim3(tiles), dim3(256), 0, ctx->stream, R, norms, perm, n, (f16x8 *)P, bits.p, err_rows,
                       d, (double *)nullptr);
    err_rows = nullptr;  // (written)
  } else if (d % 128 == 0 && (uint64_t)tiles * (d / 128) < 0x7FFFFFFFull && ((uintptr_t)R & 15) == 0 && ctx->opt_join_pack_wide_tiles) {
    // a multiple of 128 columns: the same pass, a block per (tile, 128 columns)
    DevBuf<double> part;
    const uint32_t chunks = d / 128;
    ctx->stats["knn_join_pack_wide_tiles"] = 1;
    AS_TRY(part.alloc(ctx, (size_t)n * chunks));
    ProfScope ps(ctx, "pack_unit_rows");
    hipLaunchKernelGGL(pack_f16_tile128_kernel<true>, dim3(tiles * chunks), dim3(256), 0, ctx->stream, R, norms, perm, n, (f16x8 *)P, bits.p,
                       (float *)nullptr, d, part.p);
    hipLaunchKernelGGL(f16_error_reduce_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, (const double *)part.p, n, chunks, bits.p, err_rows);
    err_rows = nullptr;  // (written)
  } else if (KS <= 8) {
    // up to 128 columns: the plane error comes out of the packing pass (16 partial sums per row, 128 B against the 1 KB of a
    // second read of the row: pack 1.66 + error 1.09 ms at 1 M x 128 before)
    DevBuf<double> part;
    AS_TRY(part.alloc(ctx, (size_t)n * 2 * KS));
    const uint64_t total = packed_vecs_f16(tiles, d);
    {
      ProfScope ps(ctx, "pack_unit_rows");
      hipLaunchKernelGGL(pack_f16_kernel, dim3(cdiv(total, 256)), dim3(256), 0, ctx->stream, R, norms, perm, (uint64_t)0, n, d, KS, total,
                         (f16x8 *)P, part.p);
      hipLaunchKernelGGL(f16_error_reduce_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, (const double *)part.p, n, 2 * KS, bits.p, err_rows);
      err_rows = nullptr;  // (written)
    }
  } else {
    AS_TRY(pack_f16(ctx, R, norms, perm, 0, n, d, P, tiles));
    hipLaunchKernelGGL(f16_row_error_kernel, dim3((unsigned)std::min<uint64_t>(cdiv(n, 4), 256 * 32)), dim3(256), 0, ctx->stream, R, norms,
                       n, d, bits.p, perm, err_rows);
    err_rows = nullptr;  // (written)
  }
  AS_HIP(ctx, hipGetLastError());
  unsigned int h = 0;
  AS_HIP(ctx, copy_d2h(ctx, &h, bits.p, sizeof(h)));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  float e;
  std::memcpy(&e, &h, sizeof(e));
  ctx->f16_err = (double)e;
  if (err_rows) {
    hipLaunchKernelGGL(fill_f32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, err_rows, n, e);
    AS_HIP(ctx, hipGetLastError());
  }
  return AS_OK;
}

int pack_f16_from_rows(as_ctx *ctx, const void *Hrow, uint64_t n_src, const uint32_t *row_ids, uint64_t n, void *P, uint32_t tiles) {
  ProfScope ps(ctx, "pack_unit_rows");
  hipLaunchKernelGGL(f16_image_from_rows_kernel, dim3(tiles), dim3(256), 0, ctx->stream, (const f16x8 *)Hrow, n_src, row_ids, n, (f16x8 *)P);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

int pack_f16_from_image(as_ctx *ctx, const void *row_image, uint64_t n_src, const uint32_t *row_ids, uint64_t n, uint32_t d, void *P, uint32_t tiles) {
  // (ctx->rowmajor_image: the source is the rows' f16 form ROW-MAJOR, not an image in fragment layout -- own_join_prepare says so)
  if (row_image && row_image == ctx->rowmajor_image && d == 128) return pack_f16_from_rows(ctx, row_image, n_src, row_ids, n, P, tiles);
  const uint32_t KS = f16_ksteps(d);
  const uint64_t total = packed_vecs_f16(tiles, d);
  ProfScope ps(ctx, "pack_unit_rows");
  hipLaunchKernelGGL(f16_image_gather_kernel, dim3(cdiv(total, 256)), dim3(256), 0, ctx->stream, (const f16x8 *)row_image, n_src, row_ids, n, KS, total,
                     (f16x8 *)P);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

// The f16 image of ALL rows in row order on every rank of a build sharded by rows: each rank packs its own rows (pack_f16_all:
// the same stored values and the same plane error as on one GPU), the pieces are all-gathered in place (shard starts are multiples
// of 512 rows, so a rank's rows are whole tiles), and ctx->f16_err becomes the largest plane error over all ranks.
int sharded_max_plane_error(as_ctx *ctx, const ShardedRows &sr, double own_err);
int pack_f16_rows_sharded(as_ctx *ctx, const ShardedRows &sr, const double *norms_all, uint64_t n, uint32_t d, void *Prow, uint32_t tiles) {
  const uint32_t KS = f16_ksteps(d);
  const uint64_t tile_bytes = (uint64_t)KS * 4 * 64 * 16;
  const uint64_t used_tiles = (n + kTileRows - 1) / kTileRows;
  if (tiles > used_tiles)  // (the padding of the last super tile: rows of zeros)
    AS_HIP(ctx, hipMemsetAsync((char *)Prow + used_tiles * tile_bytes, 0, (tiles - used_tiles) * tile_bytes, ctx->stream));
  double own_err = 0.0;
  if (sr.count) {
    const uint32_t own_tiles = (sr.count + kTileRows - 1) / kTileRows;
    AS_TRY(pack_f16_all(ctx, sr.own, norms_all + sr.lo, sr.count, d, (char *)Prow + (uint64_t)(sr.lo / kTileRows) * tile_bytes, own_tiles, nullptr, nullptr));
    own_err = ctx->f16_err;
  }
  std::vector<uint64_t> bytes((size_t)sr.world), offs((size_t)sr.world), eight((size_t)sr.world, 8), at8((size_t)sr.world);
  for (int r = 0; r < sr.world; ++r) {
    const uint64_t t0 = sr.row_lo[(size_t)r] / kTileRows, t1 = (sr.row_hi[(size_t)r] + kTileRows - 1) / kTileRows;
    bytes[(size_t)r] = sr.row_hi[(size_t)r] > sr.row_lo[(size_t)r] ? (t1 - t0) * tile_bytes : 0;
    offs[(size_t)r] = t0 * tile_bytes;
    at8[(size_t)r] = (uint64_t)r * 8;
  }
  AS_TRY(sr.gather_v(Prow, bytes.data(), offs.data()));
  (void)eight, (void)at8;
  return sharded_max_plane_error(ctx, sr, own_err);
}

// ctx->f16_err = the largest plane error over all ranks (every rank contributes the one of its own rows)
int sharded_max_plane_error(as_ctx *ctx, const ShardedRows &sr, double own_err) {
  std::vector<uint64_t> eight((size_t)sr.world, 8), at8((size_t)sr.world);
  for (int r = 0; r < sr.world; ++r) at8[(size_t)r] = (uint64_t)r * 8;
  DevBuf<double> errs;
  AS_TRY(errs.alloc(ctx, (size_t)sr.world));
  AS_HIP(ctx, copy_h2d(ctx, errs.p + sr.rank, &own_err, sizeof(double)));
  AS_TRY(sr.gather_v(errs.p, eight.data(), at8.data()));
  std::vector<double> h((size_t)sr.world);
  AS_HIP(ctx, copy_d2h(ctx, h.data(), errs.p, sizeof(double) * sr.world));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double e = 0.0;
  for (double v : h) e = std::max(e, v);
  ctx->f16_err = e;
  return AS_OK;
}

static size_t f16_join_lds(uint32_t KS, bool search) {  // the dynamic part: two stages of B + the column thresholds (the queues are static)
  (void)search;
  return (size_t)2 * 2 * KS * 4 * 64 * 16 + 2 * 256 * sizeof(float);
}

// Threshold join over super tiles [I_begin, I_begin + I_count) of the T2 = ceil(n / 512) super tiles of the f16 image P
// (all n rows).  theta: T2 * 512 thresholds (+inf on padding).  cnt / cand (n x cap) are indexed by global row; cnt must be
// zeroed by the caller, unit_counter too.  dim <= 128; AS_ERR_UNSUPPORTED beyond.
int join_f16(as_ctx *ctx, const void *P, uint32_t n, uint32_t d, uint32_t I_begin, uint32_t I_count, const float *theta,
             uint32_t cap, uint32_t *cnt, uint32_t *cand, uint32_t *unit_counter, const char *prof_name, const uint32_t *perm,
             const unsigned char *mask) {
  const uint32_t KS = (d + 15) / 16;
  if (!(KS >= 1 && KS <= 8)) return fail(ctx, AS_ERR_UNSUPPORTED, "f16 join: dim not covered");
  const uint32_t T2 = (n + kSuperH - 1) / kSuperH;
  const size_t lds = f16_join_lds(KS, false);
  uint32_t w_opt = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(ctx->opt_join_block, 64));
  // The masked pass over a SHORT range of row super tiles (a rank of a sharded build): nearly all flagged stages of a row super tile
  // sit around the diagonal, in one or two units of 64 column super tiles -- 244 row super tiles (1 M rows, 8 ranks) leave some 400
  // heavy units for 256 workgroups, and the longest of them is the pass: 1.68 ms for a sixth of the one-GPU pass's blocks, 1.04 with
  // units of one column super tile (977 row super tiles -- 4 M rows, 8 ranks --: 64 and 8 within 3 %, 2 a sixth slower).
  if (mask && ctx->opt_join_masked_block > 0) w_opt = (uint32_t)std::min<int64_t>(ctx->opt_join_masked_block, 64);
  else if (mask && I_count < 512) w_opt = std::min(w_opt, std::max(1u, I_count / 128));
  const uint32_t W = std::max(1u, std::min(w_opt, T2 / 4));
  const uint64_t units = (uint64_t)((T2 + W - 1) / W) * I_count;
  const unsigned grid = (unsigned)std::min<uint64_t>(units, 256u);
  auto parked = std::make_unique<DeferredStat>();  // the kernel's counters: statistics, read back with the next wait for the stream
  DevBuf<unsigned long long> &stamps = parked->buf;
  AS_TRY(stamps.alloc(ctx, 10));
  AS_HIP(ctx, hipMemsetAsync(stamps.p, 0, 10 * sizeof(unsigned long long), ctx->stream));
  DevBuf<uint32_t> unit_list;  // behind the first pass: the units that hold anything (the kernel walks these alone)
  if (mask && ctx->opt_join_unit_list && units < (1ull << 31)) {
    AS_TRY(unit_list.alloc(ctx, units + 1));
    AS_HIP(ctx, hipMemsetAsync(unit_list.p, 0, sizeof(uint32_t), ctx->stream));
  }
  {
    ProfScope ps(ctx, prof_name);
    if (unit_list.p)
      hipLaunchKernelGGL(join_unit_list_kernel, dim3((unsigned)cdiv(units, (uint64_t)256)), dim3(256), 0, ctx->stream, (const unsigned long long *)mask, T2,
                         I_begin, I_count, W, unit_list.p);
    // rows of 128 values behind the first pass: the form with lister waves ("join_lister", mfma_join_f16_lister.h)
    {
#define AS_JOIN(K)                                                                                                        \
  case K:                                                                                                                 \
    if (mask && (K != 8 || ctx->opt_debug == 0)) { /* (the diagnostics of this form exist at 8 k steps only) */    \
      AS_HIP(ctx, hipFuncSetAttribute((const void *)pair_join_f16_kernel<K, 0, true, false>,                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                             \
      hipLaunchKernelGGL((pair_join_f16_kernel<K, 0, true, false>), dim3(grid), dim3(512), lds, ctx->stream,       \
                         (const f16x8 *)P, (const f16x8 *)nullptr, T2, I_begin, I_count, W, n, theta, cap, cnt, cand,     \
                         unit_counter, 0, stamps.p, F16SearchTerm{}, perm, (const unsigned long long *)mask,              \
                         (const uint32_t *)nullptr, (const uint32_t *)unit_list.p);                                       \
    } else if (mask) {                                                                                                    \
      AS_HIP(ctx, hipFuncSetAttribute((const void *)pair_join_f16_kernel<8, 0, true>,                                     \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                             \
      hipLaunchKernelGGL((pair_join_f16_kernel<8, 0, true>), dim3(grid), dim3(512), lds, ctx->stream, (const f16x8 *)P,   \
                         (const f16x8 *)nullptr, T2, I_begin, I_count, W, n, theta, cap, cnt, cand, unit_counter,         \
                         (int)ctx->opt_debug, stamps.p, F16SearchTerm{}, perm, (const unsigned long long *)mask,          \
                         (const uint32_t *)nullptr, (const uint32_t *)unit_list.p);                                       \
    } else {                                                                                                              \
      AS_HIP(ctx, hipFuncSetAttribute((const void *)pair_join_f16_kernel<K, 0, false>,                                    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                             \
      hipLaunchKernelGGL((pair_join_f16_kernel<K, 0, false>), dim3(grid), dim3(512), lds, ctx->stream, (const f16x8 *)P,  \
                         (const f16x8 *)nullptr, T2, I_begin, I_count, W, n, theta, cap, cnt, cand, unit_counter,         \
                         (int)ctx->opt_debug, stamps.p, F16SearchTerm{}, perm, (const unsigned long long *)nullptr,       \
                         (const uint32_t *)nullptr);                                                                      \
    }                                                                                                                     \
    break;
    switch (KS) {
      AS_JOIN(1) AS_JOIN(2) AS_JOIN(3) AS_JOIN(4) AS_JOIN(5) AS_JOIN(6) AS_JOIN(7) AS_JOIN(8)
      default: return fail(ctx, AS_ERR_UNSUPPORTED, "f16 join: dim not covered");
    }
#undef AS_JOIN
    }
    AS_HIP(ctx, hipGetLastError());
  }
  if (!(ctx->opt_debug & 8)) {
    parked->host.resize(10);
    parked->keys = {{std::string(prof_name) + "_blocks", 5}, {std::string(prof_name) + "_blocks_listed", 7}, {std::string(prof_name) + "_blocks_pretest", 1}};
    if (std::string(prof_name) == "knn_mfma_join_f16") parked->keys.push_back({"knn_join_blocks", 5});  // (mfma_drivers.hip copies it at once)
    ctx->deferred.push_back(std::move(parked));
  } else {
    unsigned long long h[10];
    AS_HIP(ctx, copy_d2h(ctx, h, stamps.p, sizeof(h)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stats[std::string(prof_name) + "_blocks"] = (int64_t)h[5];       // 32x32 blocks contracted (KS MFMAs each)
    ctx->stats[std::string(prof_name) + "_blocks_listed"] = (int64_t)h[7];  // ... that listed at least one pair
    if (ctx->opt_debug & 8) {
      const double w = (double)std::max<unsigned long long>(1, h[4]);
      std::fprintf(stderr,
                   "[join_f16 stamps] per wave, Mcycles: unit prologue %.2f  blocks %.2f  dma wait %.2f  barrier %.2f; %llu blocks, "
                   "%.3f listed; in the listing %.2f of which flushing %.2f\n",
                   h[0] / w / 1e6, h[1] / w / 1e6, h[2] / w / 1e6, h[3] / w / 1e6, h[5], h[5] ? (double)h[7] / h[5] : 0.0, h[6] / w / 1e6,
                   h[8] / w / 1e6);
    }
  }
  return AS_OK;
}

// The threshold scan of the search on one f16 plane (MODE 1 of the kernel; the contract of join_bf16_search in
// mfma_join_bf16.hip, with the guess handed over as cos_floor = (g - beta) / alpha and g_minus_beta, both by query position and
// padded to whole 512-row super tiles with +inf).  PQ: f16 image of the queries (whole 512-row super tiles).
int join_f16_search(as_ctx *ctx, const void *P, uint32_t n, uint32_t d, const void *PQ, uint32_t nq, const float *cos_floor,
                    const float *g_minus_beta, const float *q_lam, const float *x_lam, float alpha, float beta, uint32_t cap,
                    uint32_t *cnt, uint32_t *cand, uint32_t *unit_counter, const char *prof_name, const uint32_t *perm,
                    const uint32_t *perm_a, const unsigned char *mask, const uint32_t *self_ids) {
  const uint32_t KS = (d + 15) / 16;
  if (!(KS >= 1 && KS <= 8)) return fail(ctx, AS_ERR_UNSUPPORTED, "f16 search scan: dim not covered");
  const uint32_t T2 = (n + kSuperH - 1) / kSuperH, a_supers = (nq + kSuperH - 1) / kSuperH;
  const size_t lds = f16_join_lds(KS, true);
  const uint32_t w_opt = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(ctx->opt_join_block, 64));
  // (on an image ordered by direction, and all the more behind the first pass, a unit's work is what its blocks hold, not how
  // many columns it spans: small units balance it)
  // ... and the one-sided pass has few row super tiles (20 for 10 k queries, a dozen for the rows of a second pass): column blocks of at
  // most 32 super tiles keep a thousand units for 256 CUs (64: search scan 4.3 -> 5.5 ms, second pass of the 4 M table 29 -> 42 ms)
  const uint32_t w_one = std::min(w_opt, 32u);
  // ... and a short batch (the few hundred queries a second guess re-scans: ONE row super tile) takes narrower blocks still, so that
  // there are about a thousand units whatever the batch (62 units of 32 column super tiles left three quarters of the CUs idle:
  // 0.6 ms for 5 % of the scan's work)
  const uint32_t w_fill = (uint32_t)std::max<uint64_t>(1, (uint64_t)T2 * a_supers / 1024);
  const uint32_t w_masked = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(ctx->opt_join_one_sided_block, 32));
  const uint32_t W = std::max(1u, std::min(std::min((mask || (perm && !(ctx->opt_debug & 262144))) ? std::min(w_one, w_masked) : w_one, w_fill), std::max(1u, T2 / 4)));
  const uint64_t units = (uint64_t)((T2 + W - 1) / W) * a_supers;
  const unsigned grid = (unsigned)std::min<uint64_t>(units, 256u);
  DevBuf<unsigned long long> stamps;
  AS_TRY(stamps.alloc(ctx, 10));
  AS_HIP(ctx, hipMemsetAsync(stamps.p, 0, 10 * sizeof(unsigned long long), ctx->stream));
  const F16SearchTerm st{q_lam, g_minus_beta, x_lam, alpha, beta, nq, self_ids};
  ProfScope ps(ctx, prof_name);
  // (the second pass of an item graph -- self_ids -- too, "join_second_slices": its open rows are in the order of the image, a slice of
  // 64 of them spans a few per cent of the columns: 0.70 -> 0.5 ms at 1 M rows.  The first version of this kernel -- runs of units
  // from an unordered list, four column super tiles a unit -- had lost there, 0.83.)
  if (mask && KS == 8 && ctx->opt_search_scan_slices > 0 && (!self_ids || ctx->opt_join_second_slices) && !(ctx->opt_debug & 8)) {
    // behind the first pass: a wave per (slice of 64 queries, a few stages of 256 columns), nothing shared (scan_slices_f16_kernel)
    const uint32_t Ws = (uint32_t)std::min<int64_t>(ctx->opt_search_scan_slices, 64);  // stages of 256 columns per unit
    const uint64_t s_units = (uint64_t)((2 * T2 + Ws - 1) / Ws) * a_supers * 8;
    if (s_units < (1ull << 31)) {
      DevBuf<uint32_t> unit_list, flags;
      DevBuf<uint64_t> off;
      AS_TRY(unit_list.alloc(ctx, s_units + 1));
      AS_TRY(flags.alloc(ctx, s_units + 1));
      AS_TRY(off.alloc(ctx, s_units + 1));
      hipLaunchKernelGGL(scan_slice_flags_kernel, dim3((unsigned)cdiv(s_units, (uint64_t)256)), dim3(256), 0, ctx->stream, (const unsigned long long *)mask, T2,
                         a_supers, Ws, flags.p);
      AS_TRY(exclusive_scan_u32_u64(ctx, flags.p, s_units, off.p));
      hipLaunchKernelGGL(scan_slice_list_kernel, dim3((unsigned)cdiv(s_units, (uint64_t)256)), dim3(256), 0, ctx->stream, (const uint32_t *)flags.p,
                         (const uint64_t *)off.p, (uint32_t)s_units, unit_list.p);
      hipLaunchKernelGGL(scan_slices_f16_kernel<8>, dim3(512), dim3(64 * kSliceWaves), 0, ctx->stream, (const f16x8 *)P, (const f16x8 *)PQ, T2, a_supers, Ws, n,
                         cos_floor, cap, cnt, cand, unit_counter, st, perm, (const unsigned long long *)mask, perm_a, (const uint32_t *)unit_list.p,
                         stamps.p, (uint32_t)std::max<int64_t>(1, std::min<int64_t>(ctx->opt_search_scan_run, 16)));
      AS_HIP(ctx, hipGetLastError());
      return AS_OK;
    }
  }
  // (diagnostics compiled out unless a debug bit asks for them: the product instantiation, as in the build's first pass)
  const bool diag = ctx->opt_debug != 0;
#define AS_LAUNCH(K, M, DG)                                                                                               \
  {                                                                                                                       \
    AS_HIP(ctx, hipFuncSetAttribute((const void *)pair_join_f16_kernel<K, 1, M, DG>,                               \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                               \
    hipLaunchKernelGGL((pair_join_f16_kernel<K, 1, M, DG>), dim3(grid), dim3(512), lds, ctx->stream,               \
                       (const f16x8 *)P, (const f16x8 *)PQ, T2, 0u, a_supers, W, n, cos_floor, cap, cnt, cand,            \
                       unit_counter, (int)ctx->opt_debug, stamps.p, st, perm,                                             \
                       (const unsigned long long *)(M ? mask : nullptr), perm_a);                                         \
  }
#define AS_JOIN(K)                                       \
  case K:                                                \
    if (K == 8 && !diag) {                               \
      if (mask) AS_LAUNCH(K, true, false)                \
      else AS_LAUNCH(K, false, false)                    \
    } else if (mask) AS_LAUNCH(K, true, true)            \
    else AS_LAUNCH(K, false, true)                       \
    break;
  switch (KS) {
    AS_JOIN(1) AS_JOIN(2) AS_JOIN(3) AS_JOIN(4) AS_JOIN(5) AS_JOIN(6) AS_JOIN(7) AS_JOIN(8)
    default: return fail(ctx, AS_ERR_UNSUPPORTED, "f16 search scan: dim not covered");
  }
#undef AS_LAUNCH
#undef AS_JOIN
  AS_HIP(ctx, hipGetLastError());
  if (ctx->opt_debug & 8) {  // diagnostic: where a wave's cycles go (the stamps of the kernel; scripts/stamps_one_sided.py)
    unsigned long long h[10];
    AS_HIP(ctx, copy_d2h(ctx, h, stamps.p, sizeof(h)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const double w = (double)std::max<unsigned long long>(1, h[4]);
    std::fprintf(stderr,
                 "[join_f16 one-sided stamps, %s, W %u, %llu units] per wave, Mcycles: unit prologue %.2f  blocks %.2f  dma wait %.2f  barrier %.2f; "
                 "%llu blocks, %.3f through the walk; in the listing %.2f of which flushing %.2f\n",
                 prof_name, W, (unsigned long long)units, h[0] / w / 1e6, h[1] / w / 1e6, h[2] / w / 1e6, h[3] / w / 1e6, h[5],
                 h[5] ? (double)h[7] / h[5] : 0.0, h[6] / w / 1e6, h[8] / w / 1e6);
  }
  return AS_OK;
}

}  // namespace asx
