// mfma_prefilter_i8.hip -- an 8-bit first pass of the symmetric threshold join (mfma_join_f16.hip): which 64 x 32 blocks of the
// all-pairs product can hold a pair that reaches a threshold at all.
//
// The join contracts every pair of row tiles on the f16 plane although, with the rows grouped by direction in the image
// (row_order.hip), only ~2 % of its 64 x 32 blocks list anything: the thresholds sit at about the 130th best cosine of a row
// and the rest of the matrix is far below them.  Telling those blocks apart needs the products, but not 11 bits of them:
// v_mfma_i32_32x32x32_i8 runs at twice the f16 rate, integer accumulation is exact, and an int8 image of the unit rows
// (one global step = largest |element| / 127) is good to  |s8 - cos| <= 2 E8 + E8^2  with E8 = the MEASURED largest row error
// norm (1.3e-2 on the bench data), i.e. delta8 ~ 2.5e-2.  A block none of whose products reaches  min(theta_row, theta_col)
// - delta8  holds no pair with an exact cosine at or above either threshold, so leaving it out of the f16 pass keeps the
// meaning of the lists (what is not in r's list has a cosine below theta[r] + delta16).  Measured on the bench data the
// wider margin lets 2.2 % of the blocks through instead of 1.7 %.
//
// Same geometry as the f16 kernel (8 waves x 64 rows = a 512-row super tile, stages of 256 columns through a 2-deep LDS-DMA
// ring, A fragments in registers, the threshold as the C operand, the same units and the same cyclic half pattern), so
// that the result is exactly what the f16 pass wants to know: one byte per (row super tile, stage of 256 columns, wave) with
// a bit per 32-column block.  The f16 kernel then visits only the stages with a bit set and each wave only its own blocks.
#include "mfma_common.h"

namespace asx {

uint32_t packed_tiles_f16(uint64_t n);

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int v16i32 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

constexpr int kSuperH = 512;
constexpr int kNever = 1 << 29;          // integer threshold nothing reaches (sums and differences of two stay inside int)
constexpr float kOperandScale = 128.f;   // the f16 image holds 128 x the unit rows

__device__ __forceinline__ int quantise(float h, float c) {
  const float q = __builtin_rintf(h * c);
  return (int)__builtin_fminf(__builtin_fmaxf(q, -127.f), 127.f);
}

// largest |value| of an f16 image (as the bits of a non-negative float)
__global__ __launch_bounds__(256) void image_absmax_kernel(const f16x8 *__restrict__ P, uint64_t vecs, unsigned int *__restrict__ out_bits) {
  float m = 0.f;
  for (uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < vecs; g += (uint64_t)gridDim.x * 256) {
    const f16x8 v = P[g];
#pragma unroll
    for (int j = 0; j < 8; ++j) m = __builtin_fmaxf(m, __builtin_fabsf((float)v[j]));
  }
  for (int o = 32; o > 0; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) raise_max_u32(out_bits, __float_as_uint(m));
}

// int8 image from the f16 image (same tiles of 128 rows, same order of the rows):
//   P8[((T * KS8 + k8) * 4 + s) * 64 + l] byte j = q( Ph value of row 128 T + 32 s + (l & 31), column 32 k8 + 16 (l >> 5) + j ),
// q(h) = clamp(rint(h * c), +-127).  Both operands of an MFMA come from this one image, so which k a byte stands for inside a
// k step does not matter as long as it is the same on both sides.
__global__ __launch_bounds__(256) void pack_i8_kernel(const f16x8 *__restrict__ Ph, uint32_t KS, uint32_t KS8, uint64_t total, float c,
                                                      i32x4 *__restrict__ P8) {
  const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= total) return;
  const uint32_t l = g & 63, s = (g >> 6) & 3;
  const uint64_t tk = g >> 8;
  const uint32_t k8 = (uint32_t)(tk % KS8);
  const uint64_t T = tk / KS8;
  const uint32_t ks = 2 * k8 + (l >> 5);
  i32x4 out = {0, 0, 0, 0};
  if (ks < KS) {
    const f16x8 *src = Ph + ((T * KS + ks) * 4 + s) * 64 + (l & 31);
    const f16x8 v0 = src[0], v1 = src[32];
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lo |= ((uint32_t)quantise((float)v0[4 * w + j], c) & 0xFFu) << (8 * j);
        hi |= ((uint32_t)quantise((float)v1[4 * w + j], c) & 0xFFu) << (8 * j);
      }
      out[w] = (int)lo;
      out[2 + w] = (int)hi;
    }
  }
  P8[g] = out;
}

// E8 = max over the rows of || q / (128 c) - unit row ||_2 (the arithmetic of pack_f16_kernel and pack_i8_kernel repeated on
// the row, differences in f64; one wave per row): |s8 / (128 c)^2 - cos| <= 2 E8 + E8^2 by Cauchy-Schwarz.
__global__ __launch_bounds__(256) void i8_row_error_kernel(const double *__restrict__ R, const double *__restrict__ norms, uint64_t n,
                                                           uint32_t d, float c, unsigned int *__restrict__ out_bits) {
  const int lane = threadIdx.x & 63;
  const double unit_step = 1.0 / ((double)c * (double)kOperandScale);
  float worst = 0.f;
  for (uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n; row += (uint64_t)gridDim.x * 4) {
    const double nr = norms[row];
    double e2 = 0.0;
    if (nr > 0.0 && nr < __builtin_inf()) {
      for (uint32_t k = lane; k < d; k += 64) {
        const double u = R[row * d + k] / nr;
        const float x = (float)u * kOperandScale;
        const _Float16 h = (x == x && __builtin_fabsf(x) <= 2.f * kOperandScale) ? (_Float16)x : (_Float16)0.0f;
        const double e = (double)quantise((float)h, c) * unit_step - u;
        e2 += e * e;
      }
    }
    for (int o = 32; o > 0; o >>= 1) e2 += __shfl_xor(e2, o, 64);
    const float e = (float)(__builtin_sqrt(e2) * (1.0 + 1e-6));
    if (e == e) worst = __builtin_fmaxf(worst, e);
  }
  if (lane == 0) raise_max_u32(out_bits, __float_as_uint(worst));
}

// integer thresholds: s8 >= T  <=>  s8 / (128 c)^2 >= theta - delta8  (kNever where the f16 kernel never lists: thr_scaled)
__global__ void i8_thresholds_kernel(const float *__restrict__ theta, uint64_t count, double delta8, double inv_step2, int *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float t = theta[i];
  int T = kNever;
  if (t <= 1.5f) {
    const double x = __builtin_ceil(((double)__builtin_fmaxf(t, -1.5f) - delta8) * inv_step2);
    T = (int)__builtin_fmax(__builtin_fmin(x, (double)(kNever - 1)), -(double)(kNever - 1));
  }
  out[i] = T;
}

// ---- the rotated, truncated image -------------------------------------------------------------------------------------------
// For ANY matrix Q with orthonormal columns, unit rows u = Q z + r (z = Q^T u, r orthogonal to the columns) have
//     cos(i, j) = z_i . z_j + r_i . r_j  <=  z_i . z_j + t_i t_j,      t = |r| = sqrt(1 - |z|^2).
// With Q = 31 directions that carry most of the energy of the rows (subspace iteration on the covariance of a sample) the right
// side is a 32-wide product -- 31 coordinates and the tail norm as the 32nd -- that bounds the cosine from above: ONE int8 MFMA k
// step per tile instead of four, whatever the width of the rows.  How tight it is depends on the data (t_i t_j is the slack: 0.07
// on the bench mixture, where 92 % of the energy sits in 31 directions; useless on isotropic rows, for which the caller keeps
// the full-width image), never whether it holds.  Quantised with the fixed step 1/127 (|z_k| <= 1, t <= 1): z to nearest, t
// upwards, so  cos <= s8 / 127^2 + 2 Ez + Ez^2  with Ez = the largest quantisation error norm of a row's z (measured here).
constexpr int kRotDims = 31;

// covariance of a sample of the unit rows (every `stride`-th row, f32): 64 rows per trip through LDS, a thread owns 8 x 8 entries
__global__ __launch_bounds__(256) void cov_sample_kernel(const double *__restrict__ R, const double *__restrict__ norms, uint64_t n,
                                                         uint32_t d, uint64_t stride, uint32_t n_sample, float *__restrict__ cov) {
  __shared__ float s_x[64][129];
  const int a = threadIdx.x >> 4, b = threadIdx.x & 15;
  float acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
  for (uint32_t base = blockIdx.x * 64; base < n_sample; base += gridDim.x * 64) {
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 128; e += 256) {
      const uint32_t r = e >> 7, k = e & 127;
      const uint64_t row = (uint64_t)(base + r) * stride;
      float x = 0.f;
      if (base + r < n_sample && row < n && k < d) {
        const double nr = norms[row];
        if (nr > 0.0 && nr < __builtin_inf()) x = (float)(R[row * d + k] / nr);
      }
      s_x[r][k] = x == x ? x : 0.f;
    }
    __syncthreads();
    for (int r = 0; r < 64; ++r) {
      float xa[8], xb[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        xa[i] = s_x[r][8 * a + i];
        xb[i] = s_x[r][8 * b + i];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_fmaf(xa[i], xb[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) cov[(size_t)blockIdx.x * 128 * 128 + (8 * a + i) * 128 + 8 * b + j] = acc[i][j];
}
// ... whose per-block parts (cov above: gridDim.x x 128 x 128) are added in block order: a fixed order, so that the directions are
// the same from run to run (a float atomicAdd is not)
__global__ void cov_reduce_kernel(const float *__restrict__ part, uint32_t blocks, float *__restrict__ cov) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  float s = 0.f;
  for (uint32_t b = 0; b < blocks; ++b) s += part[(size_t)b * 128 * 128 + e];
  cov[e] = s;
}

// image position p (row perm[p], or p) -> 32 bytes: q_k = rint(127 z_k) for the 31 directions, then ceil(127 t); laid out as
// the int8 MFMA operand (KS8 = 1: the two 16-byte halves of a row sit 32 lanes apart).  One thread per position, the row in
// registers, the directions broadcast from LDS.  err_bits: the largest || q / 127 - z || over the rows (float bits).
template <int DP>  // row width padded to 32
__global__ __launch_bounds__(256) void rotate_pack_i8_kernel(const double *__restrict__ R, const double *__restrict__ norms,
                                                             const uint32_t *__restrict__ perm, uint32_t n, uint32_t d,
                                                             uint32_t positions, const float *__restrict__ Q /* [32][128] */,
                                                             i32x4 *__restrict__ P8, unsigned int *__restrict__ err_bits) {
  __shared__ float s_q[kRotDims * DP];
  for (int e = threadIdx.x; e < kRotDims * DP; e += 256) s_q[e] = Q[(e / DP) * 128 + (e % DP)];
  __syncthreads();
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  float x[DP];
  bool live = false;
  if (p < n) {
    const uint64_t row = perm ? perm[p] : p;
    const double nr = norms[row];
    live = nr > 0.0 && nr < __builtin_inf();
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      const float v = (live && (uint32_t)k < d) ? (float)(R[row * d + k] / nr) : 0.f;
      x[k] = v == v ? v : 0.f;
    }
  } else {
#pragma unroll
    for (int k = 0; k < DP; ++k) x[k] = 0.f;
  }
  uint32_t bytes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float zz = 0.f, e2 = 0.f;
  for (int j = 0; j < kRotDims; ++j) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int k = 0; k < DP; k += 4) {
      const float4 q = *reinterpret_cast<const float4 *>(&s_q[j * DP + k]);
      a0 = __builtin_fmaf(x[k], q.x, a0);
      a1 = __builtin_fmaf(x[k + 1], q.y, a1);
      a2 = __builtin_fmaf(x[k + 2], q.z, a2);
      a3 = __builtin_fmaf(x[k + 3], q.w, a3);
    }
    const float z = (a0 + a1) + (a2 + a3);
    zz = __builtin_fmaf(z, z, zz);
    const int q8 = quantise(z, 127.f);
    const float e = (float)q8 * (1.f / 127.f) - z;
    e2 = __builtin_fmaf(e, e, e2);
    bytes[j >> 2] |= ((uint32_t)q8 & 0xFFu) << (8 * (j & 3));
  }
  if (live) {  // the tail norm, rounded up (2e-5: what f32 and a basis orthonormal to 1e-6 can lose)
    const float t = __builtin_sqrtf(__builtin_fmaxf(1.f - zz, 0.f) + 2e-5f);
    const int tq = (int)__builtin_fminf(__builtin_ceilf(t * 127.f), 127.f);
    bytes[7] |= (uint32_t)tq << 24;
  }
  if (p < positions) {
    const size_t v = ((size_t)(p >> 7) * 4 + ((p >> 5) & 3)) * 64 + (p & 31);
    P8[v] = i32x4{(int)bytes[0], (int)bytes[1], (int)bytes[2], (int)bytes[3]};
    P8[v + 32] = i32x4{(int)bytes[4], (int)bytes[5], (int)bytes[6], (int)bytes[7]};
  }
  float e = __builtin_sqrtf(e2) * (1.f + 1e-6f);
  for (int o = 32; o > 0; o >>= 1) e = __builtin_fmaxf(e, __shfl_xor(e, o, 64));
  if ((threadIdx.x & 63) == 0) raise_max_u32(err_bits, __float_as_uint(e));
}

// smallest threshold of every 32 consecutive positions (the scalar side of the narrow kernel's block test)
__global__ void i8_block_min_kernel(const int *__restrict__ thr, uint64_t blocks, int *__restrict__ out) {
  const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= blocks) return;
  int m = kNever;
  for (int j = 0; j < 32; ++j) m = min(m, thr[b * 32 + j]);
  out[b] = m;
}

// The same image for a handful of rows (the queries of a small batch: a thread per row takes 60 us of dependent loads for one
// row): a wave per row, lane l holds columns l and l + 64, the directions straight from L2.  dim <= 128.
__global__ __launch_bounds__(64) void rotate_pack_i8_few_kernel(const double *__restrict__ R, const double *__restrict__ norms,
                                                                const uint32_t *__restrict__ perm, uint32_t n, uint32_t d,
                                                                const float *__restrict__ Q /* [32][128] */, i32x4 *__restrict__ P8,
                                                                unsigned int *__restrict__ err_bits) {
  const uint32_t p = blockIdx.x, lane = threadIdx.x;
  float x0 = 0.f, x1 = 0.f;
  bool live = false;
  if (p < n) {
    const uint64_t row = perm ? perm[p] : p;
    const double nr = norms[row];
    live = nr > 0.0 && nr < __builtin_inf();
    if (live && lane < d) x0 = (float)(R[row * d + lane] / nr);
    if (live && lane + 64 < d) x1 = (float)(R[row * d + lane + 64] / nr);
    x0 = x0 == x0 ? x0 : 0.f;
    x1 = x1 == x1 ? x1 : 0.f;
  }
  float zz = 0.f, e2 = 0.f;
  int mine = 0;  // lane j < 31: q_j, lane 31: the tail
  for (int j = 0; j < kRotDims; ++j) {
    float a = __builtin_fmaf(x0, Q[j * 128 + lane], x1 * Q[j * 128 + 64 + lane]);
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    zz = __builtin_fmaf(a, a, zz);
    const int q8 = quantise(a, 127.f);
    const float e = (float)q8 * (1.f / 127.f) - a;
    e2 = __builtin_fmaf(e, e, e2);
    if ((int)lane == j) mine = q8;
  }
  if (lane == kRotDims && live) mine = (int)__builtin_fminf(__builtin_ceilf(__builtin_sqrtf(__builtin_fmaxf(1.f - zz, 0.f) + 2e-5f) * 127.f), 127.f);
  // bytes 0 .. 31 of the row = lanes 0 .. 31; words of four
  uint32_t w = ((uint32_t)mine & 0xFFu) << (8 * (lane & 3));
  w |= __shfl_xor(w, 1, 64);
  w |= __shfl_xor(w, 2, 64);  // lanes 4 k .. 4 k + 3 hold word k
  const uint32_t w0 = __shfl(w, 0, 64), w1 = __shfl(w, 4, 64), w2 = __shfl(w, 8, 64), w3 = __shfl(w, 12, 64);
  const uint32_t w4 = __shfl(w, 16, 64), w5 = __shfl(w, 20, 64), w6 = __shfl(w, 24, 64), w7 = __shfl(w, 28, 64);
  if (lane == 0) {
    const size_t v = ((size_t)(p >> 7) * 4 + ((p >> 5) & 3)) * 64 + (p & 31);
    P8[v] = i32x4{(int)w0, (int)w1, (int)w2, (int)w3};
    P8[v + 32] = i32x4{(int)w4, (int)w5, (int)w6, (int)w7};
    raise_max_u32(err_bits, __float_as_uint(__builtin_sqrtf(e2) * (1.f + 1e-6f)));
  }
}

__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(size_t)(lds_void *)p; }
template <int OFF>
__device__ __forceinline__ void lds_read16(i32x4 &dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int PENDING>
__device__ __forceinline__ void lds_wait1(i32x4 &x) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(PENDING));
}
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}
__device__ __forceinline__ int max16(const v16i32 &a) {
  int m = max(max(a[0], a[1]), a[2]);
  m = max(max(m, a[3]), a[4]);
  m = max(max(m, a[5]), a[6]);
  m = max(max(m, a[7]), a[8]);
  m = max(max(m, a[9]), a[10]);
  m = max(max(m, a[11]), a[12]);
  m = max(max(m, a[13]), a[14]);
  return max(m, a[15]);
}

// Units, pairing of the super tiles and the stages: exactly those of pair_join_f16_kernel<KS, 0> (mfma_join_f16.hip), which
// reads mask[((I - I_begin) * 2 T2 + stage) * 8 + wave], stage = first packed tile of the 256 columns / 2, bit = 32-column block.
template <int KS8>
__global__ __launch_bounds__(512, 2) void pair_prefilter_i8_kernel(const i32x4 *__restrict__ P8, uint32_t T2, uint32_t I_begin,
                                                                   uint32_t I_count, uint32_t W, uint32_t n,
                                                                   const int *__restrict__ thr, unsigned char *__restrict__ mask,
                                                                   uint32_t *__restrict__ unit_counter, int dbg,
                                                                   unsigned long long *__restrict__ stamps) {
  constexpr int kTileVecs = KS8 * 4 * 64;    // 16-B vectors of a 128-row tile
  constexpr int kStageVecs = 2 * kTileVecs;  // a stage = 256 columns
  constexpr int kBlocks = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  i32x4 *bufB = reinterpret_cast<i32x4 *>(smem);
  int *colthr = reinterpret_cast<int *>(smem + (size_t)2 * kStageVecs * 16);
  __shared__ uint32_t s_unit;

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lh = lane >> 5, lc = lane & 31;
  const uint32_t n_norm = (T2 - 1) / 2;
  const uint32_t n_cblocks = (T2 + W - 1) / W;
  const uint32_t total_units = n_cblocks * I_count;
  unsigned long long n_blk = 0, n_flag = 0;
  // diagnostics (as_ctx_set_option("debug", bits)): 8 cycle stamps, 256 no test, 512 no DMA of the next stage
  auto now = [&]() -> unsigned long long {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
  };
  unsigned long long t_unit = 0, t_blocks = 0, t_dma = 0, t_bar = 0, t_mark = 0;
  const bool stamp = (dbg & 8) != 0;
  if (stamp) t_mark = now();

  auto stage = [&](uint32_t tile0, int buf, int part) {
    const i32x4 *src = P8 + (size_t)tile0 * kTileVecs + lane;
    i32x4 *dst = bufB + (size_t)buf * kStageVecs;
    if (part < 0) {
#pragma unroll
      for (int p = 0; p < KS8; ++p)
        __builtin_amdgcn_global_load_lds((gbl_void *)(src + (wv * KS8 + p) * 64), (lds_void *)(dst + (wv * KS8 + p) * 64), 16, 0, 0);
    } else if (part < KS8) {  // one piece with each of the first KS8 blocks
      __builtin_amdgcn_global_load_lds((gbl_void *)(src + (wv * KS8 + part) * 64), (lds_void *)(dst + (wv * KS8 + part) * 64), 16, 0, 0);
    }
    if (wv == 7 && part <= 0)  // thresholds of the stage's columns
      __builtin_amdgcn_global_load_lds((gbl_void *)(thr + (size_t)tile0 * kTileRows + lane * 4), (lds_void *)(colthr + buf * 256), 16, 0, 0);
  };

  for (;;) {
    __syncthreads();
    if (tid == 0) s_unit = atomicAdd(unit_counter, 1u);
    __syncthreads();
    const uint32_t u = s_unit;
    if (u >= total_units) break;
    const uint32_t c = u / I_count, I = I_begin + (u - c * I_count);
    const uint32_t B0 = c * W, Wc = min(W, T2 - B0);
    uint32_t p_lo, p_hi;
    {
      const bool antipodal = (T2 % 2 == 0) && (I < T2 / 2) && T2 > 1;
      const uint32_t L = 1 + n_norm + (antipodal ? 1u : 0u);
      const uint32_t delta = B0 >= I ? B0 - I : B0 + T2 - I;
      if (delta + Wc <= T2) {
        p_lo = delta;
        p_hi = min(delta + Wc, L);
      } else {
        p_lo = 0;
        p_hi = min(delta + Wc - T2, L);
      }
      if (p_hi <= p_lo) continue;
    }
    const uint32_t steps = (p_hi - p_lo) * 2;

    i32x4 a0[KS8], a1[KS8];
    {
      const size_t T = (size_t)4 * I + (wv >> 1);
      const int s0 = 2 * (wv & 1);
      const i32x4 *a_src = P8 + T * kTileVecs + (size_t)s0 * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < KS8; ++ks) {
        a0[ks] = a_src[(size_t)ks * 256];
        a1[ks] = a_src[(size_t)ks * 256 + 64];
      }
    }
    const uint32_t row_base0 = I * kSuperH + wv * 64 + 4 * lh, row_base1 = row_base0 + 32;  // + (r & 3) + 8 * (r >> 2)
    v16i32 nth0, nth1;  // minus the row thresholds: the C operand of a block's first MFMA
    int tmax0 = -kNever, tmax1 = -kNever;  // largest finite threshold of each row group
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t0 = thr[row_base0 + (r & 3) + 8 * (r >> 2)];
      const int t1 = thr[row_base1 + (r & 3) + 8 * (r >> 2)];
      nth0[r] = -t0;
      nth1[r] = -t1;
      if (t0 < kNever) tmax0 = max(tmax0, t0);
      if (t1 < kNever) tmax1 = max(tmax1, t1);
    }
    tmax0 = max(tmax0, __shfl_xor(tmax0, 32, 64));
    tmax1 = max(tmax1, __shfl_xor(tmax1, 32, 64));

    auto tile_of = [&](uint32_t step) -> uint32_t {
      uint32_t S = I + p_lo + (step >> 1);
      S = S >= T2 ? S - T2 : S;
      return 4 * S + 2 * (step & 1u);
    };

    i32x4 fa[4];
    __syncthreads();
    stage(tile_of(0), 0, -1);
    __syncthreads();
    if (stamp) {
      const unsigned long long t = now();
      t_unit += t - t_mark;
      t_mark = t;
    }
    for (uint32_t step = 0; step < steps; ++step) {
      const uint32_t tile0 = tile_of(step);
      const int pbuf = (int)(step & 1u);
      const bool diag = (p_lo + (step >> 1)) == 0;
      const bool more = step + 1 < steps && !(dbg & 512);
      const uint32_t tile_next = more ? tile_of(step + 1) : 0;
      const uint32_t b_addr = lds_addr(bufB + (size_t)pbuf * kStageVecs) + (uint32_t)lane * 16u;
      const int *tcol = colthr + pbuf * 256 + lc;
      uint32_t bits = 0;  // wave-uniform: blocks of this stage that may hold a pair

      constexpr bool kRing = KS8 % 4 == 0;  // (see the f16 kernel)
      uint32_t blk = b_addr;
      if constexpr (kRing) {
        lds_read16<0 * 4 * 64 * 16>(fa[0], blk);
        lds_read16<1 * 4 * 64 * 16>(fa[1], blk);
        lds_read16<2 * 4 * 64 * 16>(fa[2], blk);
      }
      // The blocks of the stage, two per trip: the test of a block is issued behind the MFMAs of the NEXT one (a second pair of
      // accumulators), so that a wave's vector work runs in the shadow of its own MFMAs.  With one pair the two waves of a
      // SIMD, released together by the stage barrier, stay in step -- both contracting, then both testing with the matrix
      // pipe idle: 870 cycles per block and wave against 512 of MFMAs.
      auto block_addr = [&](int nn) -> uint32_t { return b_addr + (uint32_t)(((nn >> 2) * kTileVecs + (nn & 3) * 64) * 16); };
      auto contract = [&](int nn, uint32_t at, uint32_t nxt, v16i32 &x0, v16i32 &x1) {
        if (more && wv < 4) stage(tile_next, pbuf ^ 1, nn);
        if constexpr (!kRing) {
          lds_read16<0>(fa[0], at);
          if constexpr (KS8 > 1) lds_read16<1 * 4 * 64 * 16>(fa[1], at);
          if constexpr (KS8 > 2) lds_read16<2 * 4 * 64 * 16>(fa[2], at);
        }
        static_for<0, KS8>([&](auto ks_c) {
          constexpr int ks = decltype(ks_c)::value;
          constexpr int ahead = ks + 3;
          if (!(dbg & 1024)) {
          if constexpr (ahead < KS8) lds_read16<ahead * 4 * 64 * 16>(fa[ahead & 3], at);
          else if constexpr (kRing) lds_read16<(ahead - KS8) * 4 * 64 * 16>(fa[ahead & 3], nxt);
          constexpr int pending = kRing ? 3 : ((KS8 - 1 - ks) < 3 ? (KS8 - 1 - ks) : 3);
          lds_wait1<pending>(fa[ks & 3]);
          }
          x0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0[ks], fa[ks & 3], ks == 0 ? nth0 : x0, 0, 0, 0);
          x1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1[ks], fa[ks & 3], ks == 0 ? nth1 : x1, 0, 0, 0);
        });
        if (more && wv >= 4) stage(tile_next, pbuf ^ 1, nn);
        n_blk += 2;
      };
      // coarse: some product reaches its row's threshold, or may reach its column's (theta_col minus the largest row
      // threshold of the group); then, per lane, the same two tests on the values themselves
      auto test = [&](int nn, const v16i32 &x0, const v16i32 &x1) {
        if (dbg & 256) {
          bits |= 1u << nn;
          return;
        }
        const uint32_t colbase = tile0 * kTileRows + nn * 32;
        const int tc = (diag || colbase + lc >= n) ? kNever : tcol[nn * 32];
        const int m0 = max16(x0), m1 = max16(x1);
        const bool h0 = m0 >= min(0, tc - tmax0), h1 = m1 >= min(0, tc - tmax1);
        if (__any(h0 || h1)) {
          bool hit = m0 >= 0 || m1 >= 0;
          if (!hit && tc < kNever) {
            v16i32 v0, v1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              v0[r] = x0[r] - nth0[r];
              v1[r] = x1[r] - nth1[r];
            }
            hit = max16(v0) >= tc || max16(v1) >= tc;
          }
          if (__any(hit)) {
            bits |= 1u << nn;
            n_flag += 1;
          }
        }
      };
      v16i32 xa0, xa1, xb0, xb1;
#pragma nounroll
      for (int np = 0; np < kBlocks; np += 2) {
        const uint32_t at1 = block_addr(np + 1), at2 = block_addr(np + 2);
        contract(np, blk, at1, xa0, xa1);
        if (np > 0) test(np - 1, xb0, xb1);
        contract(np + 1, at1, at2, xb0, xb1);
        test(np, xa0, xa1);
        blk = at2;
      }
      test(kBlocks - 1, xb0, xb1);
      if (lane == 0) mask[((size_t)(I - I_begin) * 2 * T2 + (tile0 >> 1)) * 8 + wv] = (unsigned char)bits;
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
  if (lane == 0) {
    atomicAdd(stamps + 0, n_blk);
    atomicAdd(stamps + 1, n_flag);
  }
  if (stamp && lane == 0) {
    atomicAdd(stamps + 2, t_unit);
    atomicAdd(stamps + 3, t_blocks);
    atomicAdd(stamps + 4, t_dma);
    atomicAdd(stamps + 5, t_bar);
    atomicAdd(stamps + 6, 1ull);
  }
}

// The same pass for images of one or two k steps (rows up to 64 wide, and the rotated image): with so few MFMAs per block the
// kernel is bound by the vector work of the test (16 v_max3 for the 32 values of a lane against 2 to 4 MFMAs), so everything
// else is kept out of the way: stages of 512 columns (16 blocks between barriers), the whole next stage requested right behind
// the barrier, the ring of four fragment registers running across blocks and packed tiles (a trip of the loop = the four
// blocks of one packed tile, fully unrolled), the test of a block issued behind the MFMAs of the next (two sets of
// accumulators), and the column thresholds read by hand -- a compiler-generated LDS read would wait for the pending DMA.
// RECT: the one-sided form of the search scan -- the rows are those of a second image PA (the queries: I_count super tiles of
// it, thresholds thr_a by their position), every one of which meets every column super tile of P8; no column side.
// OCC4: one accumulator set (the test of a block right behind its MFMAs) in 128 registers, so that two workgroups share a CU:
// four waves per SIMD instead of two interleave by themselves and cover each other's barriers.
// TILED (OCC4, not RECT): tile_mask (cone_tiles.hip) says which pairs of 128-position tiles can hold a pair at all -- row tile
// 4 (I - I_begin) + a of the call against column tile 4 S + c, a bit each, tm_words words per row tile.  A unit then visits only
// the column super tiles with a bit set for any of its four row tiles (their list is made in LDS, as the masked f16 pass lists
// its stages), and a wave contracts only the 128-column trips its own row tile needs; the ring of fragment registers restarts
// at every trip.  The output mask of what is not visited must be zero (the launcher clears it).
template <int KS8, bool RECT, bool OCC4 = false, bool TILED = false>
__global__ __launch_bounds__(512, OCC4 ? 4 : 2) void pair_prefilter_i8_narrow_kernel(const i32x4 *__restrict__ P8, const i32x4 *__restrict__ PA,
                                                                          const int *__restrict__ thr_a, uint32_t T2, uint32_t I_begin,
                                                                          uint32_t I_count, uint32_t W, uint32_t n,
                                                                          const int *__restrict__ thr, const int *__restrict__ thr_min,
                                                                          unsigned char *__restrict__ mask,
                                                                          uint32_t *__restrict__ unit_counter, int dbg,
                                                                          unsigned long long *__restrict__ stamps,
                                                                          const uint32_t *__restrict__ tile_mask = nullptr, uint32_t tm_words = 0) {
  static_assert(!TILED || (OCC4 && !RECT && KS8 == 1), "the tiled form exists for the rotated image at four waves per SIMD");
  static_assert(KS8 == 1 || KS8 == 2, "one or two k steps");
  constexpr int kTileVecs = KS8 * 4 * 64;    // 16-B vectors of a 128-row tile
  constexpr int kStageVecs = 4 * kTileVecs;  // a stage = 512 columns = one column super tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  i32x4 *bufB = reinterpret_cast<i32x4 *>(smem);
  int *colthr = reinterpret_cast<int *>(smem + (size_t)2 * kStageVecs * 16);  // 2 x 512, then the block minima (2 x 64 slots),
  int *colmin = colthr + 2 * 512;                                             // then 8 KB nobody owns: the ring reads ahead
  __shared__ uint32_t s_unit;
  __shared__ uint32_t s_st[64];  // TILED: the unit's column super tiles with anything to do: step | 16 tile bits << 8
  __shared__ uint32_t s_nl;

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lh = lane >> 5, lc = lane & 31;
  const uint32_t n_norm = (T2 - 1) / 2;
  const uint32_t n_cblocks = (T2 + W - 1) / W;
  const uint32_t total_units = n_cblocks * I_count;
  unsigned long long n_blk = 0, n_flag = 0;
  auto now = [&]() -> unsigned long long {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
  };
  unsigned long long t_unit = 0, t_blocks = 0, t_dma = 0, t_bar = 0, t_mark = 0;
  const bool stamp = (dbg & 8) != 0;
  if (stamp) t_mark = now();

  // the 16 KS8 + 2 pieces (1 KB each) of column super tile S: 2 KS8 per wave, the thresholds with waves 6 and 7
  auto stage = [&](uint32_t S, int buf) {
    const i32x4 *src = P8 + (size_t)4 * S * kTileVecs + lane;
    i32x4 *dst = bufB + (size_t)buf * kStageVecs;
#pragma unroll
    for (int p = 0; p < 2 * KS8; ++p)
      __builtin_amdgcn_global_load_lds((gbl_void *)(src + (wv * 2 * KS8 + p) * 64), (lds_void *)(dst + (wv * 2 * KS8 + p) * 64), 16, 0, 0);
    if (!RECT && wv >= 6)
      __builtin_amdgcn_global_load_lds((gbl_void *)(thr + (size_t)S * kSuperH + (wv - 6) * 256 + lane * 4),
                                       (lds_void *)(colthr + buf * 512 + (wv - 6) * 256), 16, 0, 0);
    if (!RECT && wv == 5)  // the minima of its 16 blocks (lanes 16 .. repeat them)
      __builtin_amdgcn_global_load_lds((gbl_void *)(thr_min + (size_t)S * 16 + (lane & 15)), (lds_void *)(colmin + buf * 64), 4, 0, 0);
  };

  for (;;) {
    __syncthreads();
    if (tid == 0) s_unit = atomicAdd(unit_counter, 1u);
    __syncthreads();
    const uint32_t u = s_unit;
    if (u >= total_units) break;
    const uint32_t c = u / I_count, I = I_begin + (u - c * I_count);
    const uint32_t B0 = c * W, Wc = min(W, T2 - B0);
    uint32_t p_lo, p_hi;
    if (RECT) {  // every column super tile of the block, by absolute index
      p_lo = B0;
      p_hi = B0 + Wc;
    } else {
      const bool antipodal = (T2 % 2 == 0) && (I < T2 / 2) && T2 > 1;
      const uint32_t L = 1 + n_norm + (antipodal ? 1u : 0u);
      const uint32_t delta = B0 >= I ? B0 - I : B0 + T2 - I;
      if (delta + Wc <= T2) {
        p_lo = delta;
        p_hi = min(delta + Wc, L);
      } else {
        p_lo = 0;
        p_hi = min(delta + Wc - T2, L);
      }
      if (p_hi <= p_lo) continue;
    }
    const uint32_t steps = p_hi - p_lo;  // column super tiles to visit
    uint32_t n_list = steps;             // ... of which (TILED) those with a tile pair to look at
    if constexpr (TILED) {
      if (tid < 64) {  // (steps <= W <= 64: one lane of wave 0 per column super tile)
        uint32_t bits16 = 0;
        if ((uint32_t)tid < steps) {
          uint32_t S = I + p_lo + (uint32_t)tid;
          S = S >= T2 ? S - T2 : S;
          const uint32_t ct = 4u * S;  // its four column tiles: one nibble of a word
#pragma unroll
          for (uint32_t a = 0; a < 4; ++a)
            bits16 |= ((tile_mask[(size_t)(4u * (I - I_begin) + a) * tm_words + (ct >> 5)] >> (ct & 31u)) & 0xFu) << (4u * a);
        }
        const unsigned long long have = __ballot(bits16 != 0);
        if (bits16) s_st[__builtin_amdgcn_mbcnt_hi((uint32_t)(have >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)have, 0u))] = (uint32_t)tid | (bits16 << 8);
        if (tid == 0) s_nl = (uint32_t)__builtin_popcountll(have);
      }
      __syncthreads();
      n_list = s_nl;
      if (n_list == 0) continue;
    }

    i32x4 a0[KS8], a1[KS8];
    {
      const size_t T = (size_t)4 * I + (wv >> 1);
      const int s0 = 2 * (wv & 1);
      const i32x4 *a_src = (RECT ? PA : P8) + T * kTileVecs + (size_t)s0 * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < KS8; ++ks) {
        a0[ks] = a_src[(size_t)ks * 256];
        a1[ks] = a_src[(size_t)ks * 256 + 64];
      }
    }
    const uint32_t row_base0 = I * kSuperH + wv * 64 + 4 * lh, row_base1 = row_base0 + 32;
    v16i32 nth0, nth1;
    int tmax0 = -kNever, tmax1 = -kNever;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t0 = (RECT ? thr_a : thr)[row_base0 + (r & 3) + 8 * (r >> 2)];
      const int t1 = (RECT ? thr_a : thr)[row_base1 + (r & 3) + 8 * (r >> 2)];
      nth0[r] = -t0;
      nth1[r] = -t1;
      if (t0 < kNever) tmax0 = max(tmax0, t0);
      if (t1 < kNever) tmax1 = max(tmax1, t1);
    }
    tmax0 = max(tmax0, __shfl_xor(tmax0, 32, 64));
    tmax1 = max(tmax1, __shfl_xor(tmax1, 32, 64));
    const int tmaxw = __builtin_amdgcn_readfirstlane(max(tmax0, tmax1));  // the largest finite threshold of the wave's 64 rows
    auto super_of = [&](uint32_t step) -> uint32_t {
      const uint32_t S = (RECT ? 0u : I) + p_lo + step;
      return S >= T2 ? S - T2 : S;
    };

    i32x4 fa[4];
    auto step_at = [&](uint32_t li) -> uint32_t { return TILED ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(s_st[li] & 0xFFu)) : li; };
    __syncthreads();
    stage(super_of(step_at(0)), 0);
    __syncthreads();
    if (stamp) {
      const unsigned long long t = now();
      t_unit += t - t_mark;
      t_mark = t;
    }
    for (uint32_t li = 0; li < n_list; ++li) {
      const uint32_t step = step_at(li);
      // TILED: the 128-column trips of this stage that the wave's own row tile (two waves each) needs
      const uint32_t my_trips = TILED ? ((uint32_t)__builtin_amdgcn_readfirstlane((int)(s_st[li] >> 8)) >> (4u * ((uint32_t)wv >> 1))) & 0xFu : 0xFu;
      const uint32_t S = super_of(step);
      const int pbuf = (int)(li & 1u);
      const bool diag = RECT || (p_lo + step) == 0;  // no column side
      if (li + 1 < n_list) stage(super_of(step_at(li + 1)), pbuf ^ 1);  // (buffer pbuf ^ 1 was let go at the barrier)
      const uint32_t b_addr = lds_addr(bufB + (size_t)pbuf * kStageVecs) + (uint32_t)lane * 16u;
      const uint32_t t_addr = lds_addr(colthr + pbuf * 512) + (uint32_t)lc * 4u;
      uint32_t bits = 0;  // wave-uniform: blocks of this stage that may hold a pair

      // byte offset of fragment g of a trip (block g / KS8, k step g % KS8); blocks 4 .. are those of the next packed tile
      auto frag_off = [](int g) constexpr -> int { return ((g / KS8) >> 2) * kTileVecs * 16 + ((g / KS8) & 3) * 64 * 16 + (g % KS8) * 4 * 64 * 16; };
      // The test of a block.  Coarse, per lane one maximum over its 32 values against a SCALAR: some product reaches its row's
      // threshold (>= 0) or may reach a column's (the smallest threshold of the block's 32 columns minus the largest of the
      // wave's rows) -- 16 v_max3 and a compare.  The blocks that pass (~10 %) take the two tests per lane on the values.
      int bm = kNever;  // lane l (mod 16): the smallest column threshold of block l of this stage
      if (!RECT)
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(bm) : "v"(lds_addr(colmin + pbuf * 64) + (uint32_t)(lane & 15) * 4u) : "memory");
      auto test = [&](int nn, const v16i32 &x0, const v16i32 &x1) {
        // (four independent chains: a single one of 16 dependent v_max3 leaves the wave waiting on itself)
        int ma = max(max(x0[0], x0[1]), x0[2]), mb = max(max(x0[8], x0[9]), x0[10]);
        int mc = max(max(x1[0], x1[1]), x1[2]), md = max(max(x1[8], x1[9]), x1[10]);
        ma = max(max(ma, x0[3]), x0[4]);
        mb = max(max(mb, x0[11]), x0[12]);
        mc = max(max(mc, x1[3]), x1[4]);
        md = max(max(md, x1[11]), x1[12]);
        ma = max(max(ma, x0[5]), x0[6]);
        mb = max(max(mb, x0[13]), x0[14]);
        mc = max(max(mc, x1[5]), x1[6]);
        md = max(max(md, x1[13]), x1[14]);
        ma = max(max(ma, x0[7]), x0[15]);
        mc = max(max(mc, x1[7]), x1[15]);
        ma = max(max(ma, mb), mc);
        const int m = max(ma, md);
        const int cmin = diag ? kNever : __builtin_amdgcn_readlane(bm, nn);
        const int rhs = min(0, cmin - tmaxw);
        if (__any(m >= rhs)) {
          bool hit = m >= 0;
          if (!diag && !__any(hit)) {
            int tc;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(tc) : "v"(t_addr + (uint32_t)nn * 128u) : "memory");
            if (tc < kNever) {  // (rare path, written for few live registers: a running maximum, no second set of 32 values)
              int mv = x0[0] - nth0[0];
#pragma unroll
              for (int r = 1; r < 16; ++r) mv = max(mv, x0[r] - nth0[r]);
#pragma unroll
              for (int r = 0; r < 16; ++r) mv = max(mv, x1[r] - nth1[r]);
              hit = hit || mv >= tc;
            }
          }
          if (__any(hit)) {
            bits |= 1u << nn;
            n_flag += 1;
          }
        }
      };
      if constexpr (!TILED) {
        lds_read16<frag_off(0)>(fa[0], b_addr);
        lds_read16<frag_off(1)>(fa[1], b_addr);
        lds_read16<frag_off(2)>(fa[2], b_addr);
      }
      v16i32 xa0, xa1, xb0, xb1;
#pragma nounroll
      for (int trip = 0; trip < 4; ++trip) {
        const uint32_t base = b_addr + (uint32_t)(trip * kTileVecs * 16);
        if constexpr (TILED) {
          if (!((my_trips >> trip) & 1u)) continue;
          lds_read16<frag_off(0)>(fa[0], base);
          lds_read16<frag_off(1)>(fa[1], base);
          lds_read16<frag_off(2)>(fa[2], base);
        }
        static_for<0, 4>([&](auto b_c) {
          constexpr int b = decltype(b_c)::value;
          static_for<0, KS8>([&](auto ks_c) {
            constexpr int ks = decltype(ks_c)::value;
            constexpr int f = b * KS8 + ks;
            if constexpr (TILED) {  // the look-ahead ends with the trip: 3, 2, 1, 0 reads behind the fragment waited for
              if constexpr (f + 3 < 4 * KS8) lds_read16<frag_off(f + 3)>(fa[(f + 3) & 3], base);
              lds_wait1<(4 * KS8 - 1 - f) < 3 ? (4 * KS8 - 1 - f) : 3>(fa[f & 3]);
            } else {
            lds_read16<frag_off(f + 3)>(fa[(f + 3) & 3], base);
            lds_wait1<3>(fa[f & 3]);
            }
            if constexpr (OCC4 || (b & 1) == 0) {
              xa0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0[ks], fa[f & 3], ks == 0 ? nth0 : xa0, 0, 0, 0);
              xa1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1[ks], fa[f & 3], ks == 0 ? nth1 : xa1, 0, 0, 0);
            } else {
              xb0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0[ks], fa[f & 3], ks == 0 ? nth0 : xb0, 0, 0, 0);
              xb1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1[ks], fa[f & 3], ks == 0 ? nth1 : xb1, 0, 0, 0);
            }
          });
          if constexpr (OCC4) {
            test(4 * trip + b, xa0, xa1);
          } else if constexpr (b == 0) {  // the block before this one, in the shadow of the MFMAs just issued
            if (trip > 0) test(4 * trip - 1, xb0, xb1);
          } else if constexpr ((b & 1) == 1) {
            test(4 * trip + b - 1, xa0, xa1);
          } else {
            test(4 * trip + b - 1, xb0, xb1);
          }
        });
        n_blk += 8;
      }
      if constexpr (!OCC4) test(15, xb0, xb1);
      if (lane < 2)
        mask[((size_t)(I - I_begin) * 2 * T2 + 2 * S + (uint32_t)lane) * 8 + wv] = (unsigned char)(bits >> (8 * lane));
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
  if (lane == 0) {
    atomicAdd(stamps + 0, n_blk);
    atomicAdd(stamps + 1, n_flag);
  }
  if (stamp && lane == 0) {
    atomicAdd(stamps + 2, t_unit);
    atomicAdd(stamps + 3, t_blocks);
    atomicAdd(stamps + 4, t_dma);
    atomicAdd(stamps + 5, t_bar);
    atomicAdd(stamps + 6, 1ull);
  }
}

}  // namespace

// 31 orthonormal directions (d x 31, column j at Q[j * 128 ..], f32, row 31 zero) that carry most of trace(cov): four rounds of
// subspace iteration from a fixed start, modified Gram-Schmidt in f64.  Returns the share of the energy outside them.
static double dominant_directions(const float *cov, uint32_t d, float *Q /* 32 x 128 */) {
  const int m = kRotDims;
  std::vector<double> C((size_t)d * d), Y((size_t)m * d), Z((size_t)m * d);
  double trace = 0.0;
  for (uint32_t i = 0; i < d; ++i) {
    for (uint32_t j = 0; j < d; ++j) C[(size_t)i * d + j] = 0.5 * ((double)cov[i * 128 + j] + (double)cov[j * 128 + i]);
    trace += C[(size_t)i * d + i];
  }
  uint64_t seed = 0x9E3779B97F4A7C15ull;
  for (auto &y : Y) {
    seed = seed * 6364136223846793005ull + 1442695040888963407ull;
    y = (double)((seed >> 11) & 0xFFFFF) / 1048576.0 - 0.5;
  }
  auto orthonormalise = [&](std::vector<double> &V) {
    for (int j = 0; j < m; ++j) {
      double *vj = &V[(size_t)j * d];
      for (int pass = 0; pass < 2; ++pass)
        for (int i = 0; i < j; ++i) {
          const double *vi = &V[(size_t)i * d];
          double dot = 0.0;
          for (uint32_t k = 0; k < d; ++k) dot += vi[k] * vj[k];
          for (uint32_t k = 0; k < d; ++k) vj[k] -= dot * vi[k];
        }
      double nn = 0.0;
      for (uint32_t k = 0; k < d; ++k) nn += vj[k] * vj[k];
      if (nn > 1e-300) {
        const double inv = 1.0 / std::sqrt(nn);
        for (uint32_t k = 0; k < d; ++k) vj[k] *= inv;
      } else {  // (a direction the others already span: any unit vector orthogonal to them would do; leave it out)
        for (uint32_t k = 0; k < d; ++k) vj[k] = 0.0;
      }
    }
  };
  orthonormalise(Y);
  for (int it = 0; it < 4; ++it) {
    for (int j = 0; j < m; ++j)
      for (uint32_t i = 0; i < d; ++i) {
        double acc = 0.0;
        const double *ci = &C[(size_t)i * d], *yj = &Y[(size_t)j * d];
        for (uint32_t k = 0; k < d; ++k) acc += ci[k] * yj[k];
        Z[(size_t)j * d + i] = acc;
      }
    Y.swap(Z);
    orthonormalise(Y);
  }
  double captured = 0.0;
  for (int j = 0; j < m; ++j) {
    const double *yj = &Y[(size_t)j * d];
    for (uint32_t i = 0; i < d; ++i) {
      double acc = 0.0;
      const double *ci = &C[(size_t)i * d];
      for (uint32_t k = 0; k < d; ++k) acc += ci[k] * yj[k];
      captured += yj[i] * acc;
    }
  }
  for (int j = 0; j < 32; ++j)
    for (uint32_t k = 0; k < 128; ++k) Q[j * 128 + k] = (j < m && k < d) ? (float)Y[(size_t)j * d + k] : 0.f;
  return trace > 0.0 ? std::max(0.0, 1.0 - captured / trace) : 1.0;
}

// The rotated image of `positions` = tiles * 128 image positions (position p = row perm[p], or p; zero beyond n) for a given
// basis Qd (device, 32 x 128).  err_out: the largest quantisation error norm of a row's coordinates.
int rotated_image_pack_async(as_ctx *ctx, const double *rows, const double *norms, const uint32_t *perm, uint32_t n, uint32_t d,
                             const float *Qd, void *P8, uint32_t tiles, unsigned int *err_bits_dev) {
  struct { unsigned int *p; } bits{err_bits_dev};  // (atomicMax into it: zeroed by the caller)
  const uint32_t positions = tiles * kTileRows;
  if (positions <= 512) {  // a small batch of queries: a wave per position (the thread-per-row form is latency-bound there)
    hipLaunchKernelGGL(rotate_pack_i8_few_kernel, dim3(positions), dim3(64), 0, ctx->stream, rows, norms, perm, n, d, Qd, (i32x4 *)P8, bits.p);
    AS_HIP(ctx, hipGetLastError());
    return AS_OK;
  }
#define AS_ROT(DPV)                                                                                                          \
  hipLaunchKernelGGL(rotate_pack_i8_kernel<DPV>, dim3(cdiv(positions, 256)), dim3(256), 0, ctx->stream, rows, norms, perm, n, d, \
                     positions, Qd, (i32x4 *)P8, bits.p)
  switch ((d + 31) / 32) {
    case 1: AS_ROT(32); break;
    case 2: AS_ROT(64); break;
    case 3: AS_ROT(96); break;
    default: AS_ROT(128); break;
  }
#undef AS_ROT
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}
int rotated_image_pack(as_ctx *ctx, const double *rows, const double *norms, const uint32_t *perm, uint32_t n, uint32_t d,
                       const float *Qd, void *P8, uint32_t tiles, double *err_out) {
  DevBuf<unsigned int> bits;
  AS_TRY(bits.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(bits.p, 0, sizeof(unsigned int), ctx->stream));
  const uint32_t positions = tiles * kTileRows;
#define AS_ROT(DPV)                                                                                                          \
  hipLaunchKernelGGL(rotate_pack_i8_kernel<DPV>, dim3(cdiv(positions, 256)), dim3(256), 0, ctx->stream, rows, norms, perm, n, d, \
                     positions, Qd, (i32x4 *)P8, bits.p)
  switch ((d + 31) / 32) {
    case 1: AS_ROT(32); break;
    case 2: AS_ROT(64); break;
    case 3: AS_ROT(96); break;
    default: AS_ROT(128); break;
  }
#undef AS_ROT
  AS_HIP(ctx, hipGetLastError());
  unsigned int h = 0;
  AS_HIP(ctx, copy_d2h(ctx, &h, bits.p, sizeof(h)));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  float e;
  std::memcpy(&e, &h, sizeof(e));
  *err_out = (double)e;
  return AS_OK;
}

// 31 dominant directions of the rows (covariance of a 32 k sample, subspace iteration on the host) -> Qd (device, 32 x 128
// floats); tail_out: the share of the sample's energy outside them.
int rotated_basis(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, float *Qd, double *tail_out) {
  if (d > 128) return fail(ctx, AS_ERR_UNSUPPORTED, "rotated image: dim not covered");
  DevBuf<float> covd, covp;
  AS_TRY(covd.alloc(ctx, 128 * 128));
  const uint64_t stride = std::max<uint64_t>(1, n / 32768);
  const uint32_t n_sample = (uint32_t)((n + stride - 1) / stride);
  const uint32_t cov_blocks = (uint32_t)std::min<uint64_t>(cdiv(n_sample, 64), 256);
  AS_TRY(covp.alloc(ctx, (size_t)cov_blocks * 128 * 128));
  hipLaunchKernelGGL(cov_sample_kernel, dim3(cov_blocks), dim3(256), 0, ctx->stream, rows, norms, (uint64_t)n, d, stride, n_sample, covp.p);
  hipLaunchKernelGGL(cov_reduce_kernel, dim3(128 * 128 / 256), dim3(256), 0, ctx->stream, (const float *)covp.p, cov_blocks, covd.p);
  AS_HIP(ctx, hipGetLastError());
  std::vector<float> cov(128 * 128), Q(32 * 128);
  AS_HIP(ctx, copy_d2h(ctx, cov.data(), covd.p, cov.size() * sizeof(float)));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *tail_out = dominant_directions(cov.data(), d, Q.data());
  AS_HIP(ctx, hipMemcpyAsync(Qd, Q.data(), Q.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (Q is a local)
  return AS_OK;
}

uint64_t rotated_image_bytes(uint32_t tiles) { return (uint64_t)tiles * 4 * 64 * 16; }

// One-sided first pass of the search scan on rotated images: PA = image of the queries (a_supers super tiles of 512 positions),
// cos_floor = what the cosine of a pair must reach, by query position (+inf on padding), delta8 = the bound of the two images.
// mask: a_supers x 2 T2 x 8 bytes, as pair_join_f16_kernel<KS, 1, true> reads it.
int prefilter_i8_rect(as_ctx *ctx, const void *P8, uint32_t n, const void *PA, uint32_t a_supers, const float *cos_floor, double delta8,
                      unsigned char *mask, const char *prof_name) {
  const uint32_t T2 = (n + kSuperH - 1) / kSuperH;
  DevBuf<int> thr_a;
  DevBuf<uint32_t> unitc;
  DevBuf<unsigned long long> stamps;
  AS_TRY(thr_a.alloc(ctx, (size_t)a_supers * kSuperH));
  AS_TRY(unitc.alloc(ctx, 1));
  AS_TRY(stamps.alloc(ctx, 8));
  AS_HIP(ctx, hipMemsetAsync(unitc.p, 0, sizeof(uint32_t), ctx->stream));
  AS_HIP(ctx, hipMemsetAsync(stamps.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(i8_thresholds_kernel, dim3(cdiv((uint64_t)a_supers * kSuperH, 256)), dim3(256), 0, ctx->stream, cos_floor,
                     (uint64_t)a_supers * kSuperH, delta8, 127.0 * 127.0, thr_a.p);
  const uint32_t w_opt = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(ctx->opt_join_block, 64));
  const uint32_t W = std::max(1u, std::min(std::min(w_opt, 8u), std::max(1u, T2 / 4)));
  const uint64_t units = (uint64_t)((T2 + W - 1) / W) * a_supers;
  const unsigned grid = (unsigned)std::min<uint64_t>(units, 256u);
  const size_t lds_n = (size_t)2 * 4 * 4 * 64 * 16 + 2 * 512 * sizeof(int) + 2 * 64 * sizeof(int) + 8192;
  {
    ProfScope ps(ctx, prof_name);
    AS_HIP(ctx, hipFuncSetAttribute((const void *)pair_prefilter_i8_narrow_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds_n));
    hipLaunchKernelGGL((pair_prefilter_i8_narrow_kernel<1, true>), dim3(grid), dim3(512), lds_n, ctx->stream, (const i32x4 *)P8,
                       (const i32x4 *)PA, (const int *)thr_a.p, T2, 0u, a_supers, W, n, (const int *)nullptr, (const int *)nullptr, mask,
                       unitc.p, (int)(ctx->opt_debug & ~8), stamps.p);
    AS_HIP(ctx, hipGetLastError());
  }
  unsigned long long h[8];
  AS_HIP(ctx, copy_d2h(ctx, h, stamps.p, sizeof(h)));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stats[std::string(prof_name) + "_blocks"] = (int64_t)h[0];
  ctx->stats[std::string(prof_name) + "_blocks_flagged"] = (int64_t)h[1];
  return AS_OK;
}

uint64_t prefilter_mask_bytes(uint32_t n, uint32_t I_count) { return (uint64_t)I_count * 2 * ((n + kSuperH - 1) / kSuperH) * 8; }

// Ph: f16 image of all n rows (packed_tiles_f16(n) tiles, any row order); rows / norms: the rows themselves (any order: only the
// largest quantisation error over them is taken); theta: T2 * 512 thresholds by image position (+inf on padding), as join_f16
// takes them.  mask: prefilter_mask_bytes(n, I_count) bytes, every byte the f16 pass reads is written.  dim <= 128.
int prefilter_launch(as_ctx *ctx, const void *P8v, uint32_t KS8, double delta8, double inv_step2, uint32_t n, uint32_t I_begin,
                     uint32_t I_count, const float *theta, unsigned char *mask, const uint32_t *tile_mask = nullptr, uint32_t tm_words = 0);

// The pass on a rotated image the caller has (rows wider than 128: rotate_wide.hip): err = its largest quantisation error norm.
int prefilter_i8_rotated(as_ctx *ctx, const void *P8, double err, uint32_t n, uint32_t I_begin, uint32_t I_count, const float *theta,
                         unsigned char *mask, const uint32_t *tile_mask, uint32_t tm_words) {
  const double delta8 = (2.0 * err + err * err) * (1.0 + 1e-6) + 1e-4;
  ctx->stats["knn_join_prefilter_rotated"] = 1;
  return prefilter_launch(ctx, P8, 1, delta8, 127.0 * 127.0, n, I_begin, I_count, theta, mask, tile_mask, tm_words);
}

// perm: image position -> row of Ph (nullptr: the rows as given).
int prefilter_i8(as_ctx *ctx, const void *Ph, const double *rows, const double *norms, uint32_t n, uint32_t d, uint32_t I_begin,
                 uint32_t I_count, const float *theta, unsigned char *mask, const uint32_t *perm) {
  const uint32_t KS = (d + 15) / 16;
  uint32_t KS8 = (KS + 1) / 2;
  if (!(KS >= 1 && KS <= 8)) return fail(ctx, AS_ERR_UNSUPPORTED, "int8 prefilter: dim not covered");
  const uint32_t tiles = packed_tiles_f16(n);
  // rows wider than one k step: 31 dominant directions + the tail norm, if they carry the rows (opt: 1 decide, 2 never, 3 always)
  bool rotated = false;
  DevBuf<float> Qd;
  if (KS8 > 1 && ctx->opt_join_prefilter != 2) {
    ProfScope ps(ctx, "knn_join_prefilter_basis");
    AS_TRY(Qd.alloc(ctx, 32 * 128));
    double tail = 1.0;
    AS_TRY(rotated_basis(ctx, rows, norms, n, d, Qd.p, &tail));
    ctx->stats["knn_join_prefilter_tail_e6"] = (int64_t)(tail * 1e6);
    rotated = ctx->opt_join_prefilter == 3 || tail <= 0.95;  // (as own_join_prepare: mfma_drivers.hip, kRotatedTailMax)
  }
  ctx->stats["knn_join_prefilter_rotated"] = rotated ? 1 : 0;
  if (rotated) KS8 = 1;
  const uint64_t vecs_h = (uint64_t)tiles * KS * 4 * 64, vecs_8 = (uint64_t)tiles * KS8 * 4 * 64;
  DevBuf<unsigned char> P8;
  DevBuf<unsigned int> bits;
  AS_TRY(P8.alloc(ctx, vecs_8 * 16));
  AS_TRY(bits.alloc(ctx, 2));
  AS_HIP(ctx, hipMemsetAsync(bits.p, 0, 2 * sizeof(unsigned int), ctx->stream));
  float c = 1.f, e8 = 0.f;
  if (rotated) {
    ProfScope ps(ctx, "knn_join_prefilter_pack");
    double e = 0.0;
    AS_TRY(rotated_image_pack(ctx, rows, norms, perm, n, d, Qd.p, P8.p, tiles, &e));
    e8 = (float)e;
    c = 127.f / kOperandScale;  // (128 c)^2 = 127^2 below
  } else {
    ProfScope ps(ctx, "knn_join_prefilter_pack");
    hipLaunchKernelGGL(image_absmax_kernel, dim3((unsigned)std::min<uint64_t>(cdiv(vecs_h, 256), 256 * 16)), dim3(256), 0, ctx->stream,
                       (const f16x8 *)Ph, vecs_h, bits.p);
    AS_HIP(ctx, hipGetLastError());
    unsigned int h = 0;
    AS_HIP(ctx, copy_d2h(ctx, &h, bits.p, sizeof(h)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float amax;
    std::memcpy(&amax, &h, sizeof(amax));
    c = (amax > 0.f && amax < __builtin_inff()) ? 127.f / amax : 1.f;
    hipLaunchKernelGGL(pack_i8_kernel, dim3(cdiv(vecs_8, 256)), dim3(256), 0, ctx->stream, (const f16x8 *)Ph, KS, KS8, vecs_8, c,
                       (i32x4 *)P8.p);
    hipLaunchKernelGGL(i8_row_error_kernel, dim3((unsigned)std::min<uint64_t>(cdiv(n, 4), 256 * 32)), dim3(256), 0, ctx->stream, rows,
                       norms, (uint64_t)n, d, c, bits.p + 1);
    AS_HIP(ctx, hipGetLastError());
    AS_HIP(ctx, copy_d2h(ctx, &h, bits.p + 1, sizeof(h)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(&e8, &h, sizeof(e8));
  }
  const double E = (double)e8, delta8 = (2.0 * E + E * E) * (1.0 + 1e-6) + (rotated ? 1e-4 : 1e-9);
  const double inv_step2 = ((double)c * (double)kOperandScale) * ((double)c * (double)kOperandScale);
  return prefilter_launch(ctx, P8.p, KS8, delta8, inv_step2, n, I_begin, I_count, theta, mask);
}

// the pass itself on a finished int8 image of KS8 k steps: s8 >= T  <=>  s8 / inv_step2 >= theta - delta8
int prefilter_launch(as_ctx *ctx, const void *P8v, uint32_t KS8, double delta8, double inv_step2, uint32_t n, uint32_t I_begin,
                     uint32_t I_count, const float *theta, unsigned char *mask, const uint32_t *tile_mask, uint32_t tm_words) {
  const uint32_t T2 = (n + kSuperH - 1) / kSuperH;
  struct { const void *p; } P8{P8v};
  DevBuf<int> thr;
  DevBuf<uint32_t> unitc;
  auto parked = std::make_unique<DeferredStat>();
  DevBuf<unsigned long long> &stamps = parked->buf;
  AS_TRY(thr.alloc(ctx, (size_t)T2 * kSuperH));
  AS_TRY(unitc.alloc(ctx, 1));
  AS_TRY(stamps.alloc(ctx, 8));
  AS_HIP(ctx, hipMemsetAsync(unitc.p, 0, sizeof(uint32_t), ctx->stream));
  AS_HIP(ctx, hipMemsetAsync(stamps.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
  ctx->stats["knn_join_prefilter_delta_e6"] = (int64_t)(delta8 * 1e6);
  hipLaunchKernelGGL(i8_thresholds_kernel, dim3(cdiv((uint64_t)T2 * kSuperH, 256)), dim3(256), 0, ctx->stream, theta,
                     (uint64_t)T2 * kSuperH, delta8, inv_step2, thr.p);
  DevBuf<int> thr_min;
  AS_TRY(thr_min.alloc(ctx, (size_t)T2 * 16));
  hipLaunchKernelGGL(i8_block_min_kernel, dim3(cdiv((uint64_t)T2 * 16, 256)), dim3(256), 0, ctx->stream, thr.p, (uint64_t)T2 * 16,
                     thr_min.p);
  AS_HIP(ctx, hipGetLastError());
  const size_t lds = (size_t)2 * 2 * KS8 * 4 * 64 * 16 + 2 * 256 * sizeof(int);
  const uint32_t w_opt = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(ctx->opt_join_block, 64));
  const uint32_t W = std::max(1u, std::min(w_opt, T2 / 4));  // (the f16 pass uses the same W: the masks are indexed by stage, not by unit)
  const uint64_t units = (uint64_t)((T2 + W - 1) / W) * I_count;
  const unsigned grid = (unsigned)std::min<uint64_t>(units, 256u);
  {
    ProfScope ps(ctx, "knn_join_prefilter");
#define AS_PRE(K)                                                                                                            \
  case K:                                                                                                                    \
    AS_HIP(ctx, hipFuncSetAttribute((const void *)pair_prefilter_i8_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                    (int)lds));                                                                              \
    hipLaunchKernelGGL((pair_prefilter_i8_kernel<K>), dim3(grid), dim3(512), lds, ctx->stream, (const i32x4 *)P8.p, T2,      \
                       I_begin, I_count, W, n, thr.p, mask, unitc.p, (int)ctx->opt_debug, stamps.p);                         \
    break;
#define AS_PRE_NARROW(K)                                                                                                     \
  case K: {                                                                                                                  \
    const size_t lds_n = (size_t)2 * 4 * K * 4 * 64 * 16 + 2 * 512 * sizeof(int) + 2 * 64 * sizeof(int) + 8192;              \
    if (tile_mask && K == 1 && !(ctx->opt_debug & 2048)) { /* behind the cone bound on tile pairs (cone_tiles.hip) */      \
      AS_HIP(ctx, hipMemsetAsync(mask, 0, prefilter_mask_bytes(n, I_count), ctx->stream)); /* (what is not visited) */    \
      AS_HIP(ctx, hipFuncSetAttribute((const void *)pair_prefilter_i8_narrow_kernel<1, false, true, true>,                   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_n));                              \
      hipLaunchKernelGGL((pair_prefilter_i8_narrow_kernel<1, false, true, true>), dim3((unsigned)std::min<uint64_t>(units, 512u)), \
                         dim3(512), lds_n, ctx->stream, (const i32x4 *)P8.p, (const i32x4 *)nullptr, (const int *)nullptr,   \
                         T2, I_begin, I_count, W, n, thr.p, thr_min.p, mask, unitc.p, (int)ctx->opt_debug, stamps.p,         \
                         tile_mask, tm_words);                                                                               \
    } else if (!(ctx->opt_debug & 2048)) {  /* (debug bit 2048: the two-set form at one workgroup per CU) */                \
      AS_HIP(ctx, hipFuncSetAttribute((const void *)pair_prefilter_i8_narrow_kernel<K, false, true>,                         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_n));                              \
      hipLaunchKernelGGL((pair_prefilter_i8_narrow_kernel<K, false, true>), dim3((unsigned)std::min<uint64_t>(units, 512u)), \
                         dim3(512), lds_n, ctx->stream, (const i32x4 *)P8.p, (const i32x4 *)nullptr, (const int *)nullptr,   \
                         T2, I_begin, I_count, W, n, thr.p, thr_min.p, mask, unitc.p, (int)ctx->opt_debug, stamps.p);        \
    } else {                                                                                                                 \
      AS_HIP(ctx, hipFuncSetAttribute((const void *)pair_prefilter_i8_narrow_kernel<K, false>,                               \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_n));                              \
      hipLaunchKernelGGL((pair_prefilter_i8_narrow_kernel<K, false>), dim3(grid), dim3(512), lds_n, ctx->stream,             \
                         (const i32x4 *)P8.p, (const i32x4 *)nullptr, (const int *)nullptr, T2, I_begin, I_count, W, n,      \
                         thr.p, thr_min.p, mask, unitc.p, (int)ctx->opt_debug, stamps.p);                                    \
    }                                                                                                                        \
  } break;
    switch (KS8) {
      AS_PRE_NARROW(1) AS_PRE_NARROW(2) AS_PRE(3) AS_PRE(4)
      default: return fail(ctx, AS_ERR_UNSUPPORTED, "int8 prefilter: dim not covered");
    }
#undef AS_PRE_NARROW
#undef AS_PRE
    AS_HIP(ctx, hipGetLastError());
  }
  if (!(ctx->opt_debug & 8)) {  // statistics only: read back with the next wait for the stream (common.h: DeferredStat)
    parked->host.resize(8);
    parked->keys = {{"knn_join_prefilter_blocks", 0}, {"knn_join_prefilter_blocks_flagged", 1}};
    ctx->deferred.push_back(std::move(parked));
    return AS_OK;
  }
  unsigned long long h[8];
  AS_HIP(ctx, copy_d2h(ctx, h, stamps.p, sizeof(h)));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->opt_debug & 8) {
    const double w = (double)std::max<unsigned long long>(1, h[6]);
    std::fprintf(stderr, "[prefilter_i8 stamps] per wave, Mcycles: unit prologue %.2f  blocks %.2f  dma wait %.2f  barrier %.2f\n",
                 h[2] / w / 1e6, h[3] / w / 1e6, h[4] / w / 1e6, h[5] / w / 1e6);
  }
  ctx->stats["knn_join_prefilter_blocks"] = (int64_t)h[0];          // 32 x 32 blocks contracted
  ctx->stats["knn_join_prefilter_blocks_flagged"] = (int64_t)h[1];  // 64 x 32 blocks left to the f16 pass
  return AS_OK;
}

}  // namespace asx
