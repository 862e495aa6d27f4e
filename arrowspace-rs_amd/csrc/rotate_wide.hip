// rotate_wide.hip -- the rotated coordinates of the rows of the own join (mfma_prefilter_i8.hip: 31 dominant directions + the norm
// of what is left bound the cosine of two unit rows from above by a 32-wide product).  Written for rows wider than 128, where a
// covariance matrix is no longer a small thing to iterate on the host, and used from 33 columns on: the directions come from subspace iteration on a SAMPLE of the rows with both products on
// the device -- Z = S Q (sample x 31), Y = S^T Z (dim x 31) -- and only the orthonormalisation of Y (dim x 31, f64) on the host.
// The coordinates of all rows are kept (n x 32 doubles: z_0 .. z_30, tail norm t): the row order of the join's image is found on
// them (cluster_order on 32 columns instead of dim), and the int8 image is packed from them once the order is known.
#include <map>
#include <mutex>
#include <vector>

#include "common.h"

namespace asx {

namespace {

constexpr int kDirs = 31;
constexpr uint32_t kMaxWide = 1024;  // 31 directions of that width fit LDS as f32 (127 KB)

// z = Q^T u for `count` rows (row index = first + i * stride), one wave per row and trip: the row in registers (lane l holds
// columns l, l + 64, ...), the directions in LDS.  out: count x 32 doubles (z_0 .. z_30 as computed in f32, then t = sqrt(max(0,
// 1 - |z|^2) + 2e-5): the tail norm rounded up by what f32 and a basis orthonormal to 1e-6 can lose), rows with a zero or
// non-finite norm all zero.  tail_part (optional): per WAVE of the grid (4 * gridDim.x doubles) its sum of 1 - |z|^2 over its rows, added
// up by the host in index order: every sum on this path has a fixed order, so that the directions, the row order found on them and
// with it which rows reach which stage are the same from run to run (a float atomicAdd here made a failure reproduce 1 time in 5).
template <int NV>  // columns per lane = ceil(dim / 64) <= 16
__global__ __launch_bounds__(256) void zproj_kernel(const double *__restrict__ R, const double *__restrict__ norms, uint64_t first,
                                                    uint64_t stride, uint32_t count, uint32_t d, const float *__restrict__ Q,
                                                    uint32_t pitch, double *__restrict__ out, double *__restrict__ tail_part) {
  extern __shared__ float s_q[];  // kDirs x pitch
  for (uint32_t e = threadIdx.x; e < (uint32_t)kDirs * pitch; e += 256) s_q[e] = Q[e];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  double tails = 0.0;
  for (uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6); i < count; i += gridDim.x * 4) {
    const uint64_t row = first + (uint64_t)i * stride;
    const double nr = norms[row];
    const bool live = nr > 0.0 && nr < __builtin_inf();
    float x[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const uint32_t k = (uint32_t)lane + 64u * j;
      const float v = (live && k < d) ? (float)(R[row * d + k] / nr) : 0.f;
      x[j] = v == v ? v : 0.f;
    }
    float zz = 0.f, mine = 0.f;  // lane j < 31 keeps z_j
    for (int dir = 0; dir < kDirs; ++dir) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) a = __builtin_fmaf(x[j], s_q[dir * pitch + lane + 64 * j], a);
      for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
      zz = __builtin_fmaf(a, a, zz);
      if (lane == dir) mine = a;
    }
    const float t = live ? __builtin_sqrtf(__builtin_fmaxf(1.f - zz, 0.f) + 2e-5f) : 0.f;
    if (lane == kDirs) mine = t;
    if (lane < 32) out[(size_t)i * 32 + lane] = (double)mine;
    if (live) tails += (double)__builtin_fmaxf(1.f - zz, 0.f);
  }
  if (tail_part && lane == 0) tail_part[blockIdx.x * 4 + (threadIdx.x >> 6)] = tails;
}

// the same for ALL rows up to 128 wide: a thread per row with the row in registers and the directions broadcast from LDS (the
// 186 cross-lane shuffles per row of the wave-per-row form cost more than its coalesced loads save: 1.9 against 0.9 ms at 1 M x 128)
template <int DP>  // row width padded to 32
__global__ __launch_bounds__(256) void zproj_thread_kernel(const double *__restrict__ R, const double *__restrict__ norms, uint32_t n,
                                                           uint32_t d, const float *__restrict__ Q, uint32_t pitch,
                                                           double *__restrict__ out, double *__restrict__ tail_part) {
  __shared__ float s_q[kDirs * DP];
  for (int e = threadIdx.x; e < kDirs * DP; e += 256) s_q[e] = Q[(size_t)(e / DP) * pitch + (e % DP)];
  __syncthreads();
  const uint32_t row = blockIdx.x * 256 + threadIdx.x;
  double tails = 0.0;
  if (row < n) {
    const double nr = norms[row];
    const bool live = nr > 0.0 && nr < __builtin_inf();
    float x[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      const float v = (live && (uint32_t)k < d) ? (float)(R[(uint64_t)row * d + k] / nr) : 0.f;
      x[k] = v == v ? v : 0.f;
    }
    float zz = 0.f;
    double *o = out + (size_t)row * 32;
    for (int j = 0; j < kDirs; ++j) {
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
      o[j] = (double)z;
    }
    o[kDirs] = live ? (double)__builtin_sqrtf(__builtin_fmaxf(1.f - zz, 0.f) + 2e-5f) : 0.0;
    if (live) tails = (double)__builtin_fmaxf(1.f - zz, 0.f);
  }
  for (int o2 = 32; o2 > 0; o2 >>= 1) tails += __shfl_xor(tails, o2, 64);  // (butterfly: the same tree in every lane and run)
  if (tail_part && (threadIdx.x & 63) == 0) tail_part[blockIdx.x * 4 + (threadIdx.x >> 6)] = tails;
}

// The same values (the four chains of a direction run over k = c, c + 4, ... in the same order) with the rows staged through LDS in
// chunks of 16 columns, a row segment of 128 B per 16 lanes: each thread walking its own 1 KB row in global memory kept the pass at
// 1.2 TB/s (0.85 ms at 1 M x 128).
template <int DP>
__global__ __launch_bounds__(256) void zproj_tile_kernel(const double *__restrict__ R, const double *__restrict__ norms, uint32_t n,
                                                         uint32_t d, const float *__restrict__ Q, uint32_t pitch,
                                                         double *__restrict__ out, double *__restrict__ tail_part) {
  __shared__ __attribute__((aligned(16))) float s_q[kDirs * DP];
  __shared__ double tile[256 * 17];
  const int tid = threadIdx.x;
  for (int e = tid; e < kDirs * DP; e += 256) s_q[e] = Q[(size_t)(e / DP) * pitch + (e % DP)];
  const uint32_t row0 = blockIdx.x * 256u, row = row0 + (uint32_t)tid;
  const double nr = row < n ? norms[row] : 0.0;
  const bool live = row < n && nr > 0.0 && nr < __builtin_inf();
  float acc[kDirs][4];
#pragma unroll
  for (int j = 0; j < kDirs; ++j)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[j][c] = 0.f;
#pragma unroll 1
  for (int c0 = 0; c0 < DP; c0 += 16) {
    __syncthreads();  // (the chunk before has been consumed; the first time: the directions are in place)
#pragma unroll
    for (int q = 0; q < 16; ++q) {  // element e = q * 256 + tid of the 256 x 16 chunk: row e / 16, column e % 16
      const int e = q * 256 + tid, r = e >> 4, c = e & 15;
      tile[r * 17 + c] = (row0 + (uint32_t)r < n && (uint32_t)(c0 + c) < d) ? R[(uint64_t)(row0 + r) * d + c0 + c] : 0.0;
    }
    __syncthreads();
    float x[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float v = (live && (uint32_t)(c0 + c) < d) ? (float)(tile[tid * 17 + c] / nr) : 0.f;
      x[c] = v == v ? v : 0.f;
    }
#pragma unroll
    for (int j = 0; j < kDirs; ++j)
#pragma unroll
      for (int k = 0; k < 16; k += 4) {
        const float4 q = *reinterpret_cast<const float4 *>(&s_q[j * DP + c0 + k]);
        acc[j][0] = __builtin_fmaf(x[k], q.x, acc[j][0]);
        acc[j][1] = __builtin_fmaf(x[k + 1], q.y, acc[j][1]);
        acc[j][2] = __builtin_fmaf(x[k + 2], q.z, acc[j][2]);
        acc[j][3] = __builtin_fmaf(x[k + 3], q.w, acc[j][3]);
      }
  }
  double tails = 0.0;
  if (row < n) {
    float zz = 0.f;
    double *o = out + (size_t)row * 32;
#pragma unroll
    for (int j = 0; j < kDirs; ++j) {
      const float z = (acc[j][0] + acc[j][1]) + (acc[j][2] + acc[j][3]);
      zz = __builtin_fmaf(z, z, zz);
      o[j] = (double)z;
    }
    o[kDirs] = live ? (double)__builtin_sqrtf(__builtin_fmaxf(1.f - zz, 0.f) + 2e-5f) : 0.0;
    if (live) tails = (double)__builtin_fmaxf(1.f - zz, 0.f);
  }
  for (int o2 = 32; o2 > 0; o2 >>= 1) tails += __shfl_xor(tails, o2, 64);  // (butterfly: the same tree in every lane and run)
  if (tail_part && (tid & 63) == 0) tail_part[blockIdx.x * 4 + (tid >> 6)] = tails;
}

// Rows of exactly 128 values, 16-byte aligned: the same coordinates again (the same four chains per direction, the same divisions),
// with a workgroup staging 64 WHOLE rows -- every wave instruction a full 1-KB row, sixteen of them in flight per wave, divided
// by the norm and kept in LDS as f32 -- and four threads per row taking eight directions each.  The chunked form above moves the
// GB of a million rows in 0.8 ms; this one in 0.5 ms (its LDS reads: 144 bytes per 32 multiply-adds).  The matrix-pipe form of
// the wider rows (zproj_mfma_kernel) was measured here in round 5: 0.74 ms.  tail_part: [block][4 waves], blocks of 64 rows.
__global__ __launch_bounds__(256) void zproj_rows128_kernel(const double *__restrict__ R, const double *__restrict__ norms, uint32_t n,
                                                            const float *__restrict__ Q, uint32_t pitch, double *__restrict__ out,
                                                            double *__restrict__ tail_part) {
  // (a direction per 132 floats: the four threads of a row read four neighbouring directions at once -- 528 bytes apart, four
  // different groups of banks; 8 directions apart they sat 4 KB apart, on the same banks, and the kernel was no faster than the old)
  constexpr int kQS = 132;
  __shared__ __attribute__((aligned(16))) float s_q[kDirs * kQS];
  __shared__ float xf[64 * 129];
  __shared__ float zs[64 * 33];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int e = tid; e < kDirs * 128; e += 256) s_q[(e >> 7) * kQS + (e & 127)] = Q[(size_t)(e >> 7) * pitch + (e & 127)];
  const uint32_t row0 = blockIdx.x * 64u;
  typedef double d2 __attribute__((ext_vector_type(2)));
  d2 v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {  // wave wv takes rows wv, wv + 4, ...: lane l the values 2 l, 2 l + 1
    const uint32_t r = row0 + (uint32_t)(4 * q + wv);
    v[q] = r < n ? reinterpret_cast<const d2 *>(R + (uint64_t)r * 128)[lane] : d2{0.0, 0.0};
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const uint32_t r = row0 + (uint32_t)(4 * q + wv);
    const double nr = r < n ? norms[r] : 0.0;
    const bool live = r < n && nr > 0.0 && nr < __builtin_inf();
    const float f0 = live ? (float)(v[q].x / nr) : 0.f, f1 = live ? (float)(v[q].y / nr) : 0.f;
    float *dst = xf + (size_t)(4 * q + wv) * 129 + 2 * lane;
    dst[0] = f0 == f0 ? f0 : 0.f;
    dst[1] = f1 == f1 ? f1 : 0.f;
  }
  __syncthreads();
  const int rr = tid >> 2, part = tid & 3;
  {
    float acc[8][4];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[jj][c] = 0.f;
    const float *xr = xf + (size_t)rr * 129;
#pragma unroll 4
    for (int k0 = 0; k0 < 128; k0 += 4) {
      const float x0 = xr[k0], x1 = xr[k0 + 1], x2 = xr[k0 + 2], x3 = xr[k0 + 3];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int j = min(4 * jj + part, kDirs - 1);  // (direction 31 does not exist: the last thread of a row repeats 30 and drops it)
        const float4 q = *reinterpret_cast<const float4 *>(&s_q[j * kQS + k0]);
        acc[jj][0] = __builtin_fmaf(x0, q.x, acc[jj][0]);
        acc[jj][1] = __builtin_fmaf(x1, q.y, acc[jj][1]);
        acc[jj][2] = __builtin_fmaf(x2, q.z, acc[jj][2]);
        acc[jj][3] = __builtin_fmaf(x3, q.w, acc[jj][3]);
      }
    }
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
      if (4 * jj + part < kDirs) zs[rr * 33 + 4 * jj + part] = (acc[jj][0] + acc[jj][1]) + (acc[jj][2] + acc[jj][3]);
  }
  __syncthreads();
  const uint32_t row = row0 + (uint32_t)rr;
  double tails = 0.0;
  if (row < n) {
    double *o = out + (size_t)row * 32;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
      if (8 * part + jj < kDirs) o[8 * part + jj] = (double)zs[rr * 33 + 8 * part + jj];  // (each thread a contiguous 64 bytes of the row)
    if (part == 0) {
      const double nr = norms[row];
      const bool live = nr > 0.0 && nr < __builtin_inf();
      float zz = 0.f;
      for (int j = 0; j < kDirs; ++j) {
        const float z = zs[rr * 33 + j];
        zz = __builtin_fmaf(z, z, zz);
      }
      o[kDirs] = live ? (double)__builtin_sqrtf(__builtin_fmaxf(1.f - zz, 0.f) + 2e-5f) : 0.0;
      if (live) tails = (double)__builtin_fmaxf(1.f - zz, 0.f);
    }
  }
  for (int o2 = 32; o2 > 0; o2 >>= 1) tails += __shfl_xor(tails, o2, 64);
  if (tail_part && lane == 0) tail_part[blockIdx.x * 4 + wv] = tails;
}

// ---- one pass over rows of exactly 128 values (round 6): the rotated coordinates AND the f16 form of the rows ---------------------
// zproj_rows128_kernel above and pack_f16_tile128_kernel (mfma_join_f16.hip) each read the GB of a million rows, divide every value by
// its row's norm and round it -- 0.49 + 0.37 ms, the second one in the order of the join, which is only known once the coordinates
// are.  Here a workgroup stages 64 whole rows once and leaves both:
//  * the rows' f16 form exactly as the packer stores it (the same quotient, the same rounding, the same squared-error butterfly per
//    row), ROW-MAJOR and in row order (Hrow: n x 128 halves, 256 B a row -- lane l writes its two values, a row per wave
//    instruction); the image in the order of the join is that one re-arranged once the order is known (f16_image_from_rows_kernel:
//    256 B per row instead of 1 KB);
//  * the coordinates on the matrix pipe: Z = U Q is 2 n 128 32 flop behind n KB, but four threads per row each walking eight
//    directions through LDS made zproj_rows128_kernel LDS-bound at 2 TB/s.  v_mfma_f32_32x32x2_f32 keeps f32 products and f32
//    accumulation (the tail term's 2e-5 allows for f32 and a basis orthonormal to 1e-6: unchanged); wave (g, h) takes rows
//    32 g .. 32 g + 31 and the k range [64 h, 64 h + 64) -- lane (lh, lc) supplies u[row lc][64 h + 32 lh + t] and
//    q[dir lc][the same k] at step t (the order in which k is summed is free; this one reads LDS without bank conflicts and keeps
//    the wave's 32 values of Q in registers) -- and the two halves meet in LDS in a fixed order.
// err_rows: the plane error of each row BY ROW; err_bits: the largest.  Grid: ceil(n / 64) blocks.
typedef float zp_v16f32 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 4) void zproj_pack_rows128_kernel(const double *__restrict__ R, const double *__restrict__ norms, uint32_t n,
                                                                 const float *__restrict__ Q, uint32_t pitch, double *__restrict__ out,
                                                                 double *__restrict__ tail_part, _Float16 *__restrict__ Hrow,
                                                                 float *__restrict__ err_rows, unsigned int *__restrict__ err_bits) {
  constexpr float kScale = 128.f;  // (kOperandScale of mfma_join_f16.hip: unit rows are stored times 2^7)
  // (the products take the place of the operand once every wave has read it: 33 KB a workgroup instead of 50, four workgroups a CU)
  __shared__ float xf[64 * 129];
  float *zs = xf;             // 64 x 33
  float *zp = xf + 64 * 33;   // the partial products of the upper k half
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lh = lane >> 5, lc = lane & 31;
  const uint32_t row0 = blockIdx.x * 64u;
  typedef double d2 __attribute__((ext_vector_type(2)));
  d2 v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {  // wave wv takes rows wv, wv + 4, ...: lane l the values 2 l, 2 l + 1
    const uint32_t r = row0 + (uint32_t)(4 * q + wv);
    v[q] = r < n ? reinterpret_cast<const d2 *>(R + (uint64_t)r * 128)[lane] : d2{0.0, 0.0};
  }
  // the wave's 32 values of Q for the matrix pipe (direction lc -- 31 does not exist: zeros --, k = 64 h + 32 lh + t)
  const int g = wv & 1, h = wv >> 1;
  float qreg[32];
#pragma unroll
  for (int t = 0; t < 32; ++t) qreg[t] = lc < kDirs ? Q[(size_t)lc * pitch + 64 * h + 32 * lh + t] : 0.f;
  float worst = 0.f;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {  // two batches of eight rows: the packer's butterfly adds a row's 64 partial errors in ITS order
    double e2[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int q = 8 * qb + b, lr = 4 * q + wv;
      const uint32_t r = row0 + (uint32_t)lr;
      const double nr = r < n ? norms[r] : 0.0;
      // the coordinates' operand (zproj_rows128_kernel: rows with a zero or non-finite norm are all zero)
      const bool live_z = r < n && nr > 0.0 && nr < __builtin_inf();
      const double u0 = v[q].x / nr, u1 = v[q].y / nr;
      const float f0 = live_z ? (float)u0 : 0.f, f1 = live_z ? (float)u1 : 0.f;
      float *dst = xf + (size_t)lr * 129 + 2 * lane;
      dst[0] = f0 == f0 ? f0 : 0.f;
      dst[1] = f1 == f1 ? f1 : 0.f;
      // the image's operand (pack_f16_tile128_kernel: any positive norm, the quotient times 2^7, out of range or NaN -> 0)
      _Float16 h0 = (_Float16)0.0f, h1 = (_Float16)0.0f;
      e2[b] = 0.0;
      if (r < n && nr > 0.0) {
        const float g0 = (float)u0 * kScale, g1 = (float)u1 * kScale;
        h0 = (g0 == g0 && __builtin_fabsf(g0) <= 2.f * kScale) ? (_Float16)g0 : (_Float16)0.0f;
        h1 = (g1 == g1 && __builtin_fabsf(g1) <= 2.f * kScale) ? (_Float16)g1 : (_Float16)0.0f;
        if (nr < __builtin_inf()) {
          const double d0 = (double)(float)h0 * (1.0 / kScale) - u0, d1 = (double)(float)h1 * (1.0 / kScale) - u1;
          e2[b] = d0 * d0 + d1 * d1;
        }
      }
      if (r < n) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        reinterpret_cast<h2 *>(Hrow + (uint64_t)r * 128)[lane] = h2{h0, h1};
      }
    }
    // per-row sums over the 64 lanes: 8 values per lane -> halving butterfly (lane bits 5, 4, 3), then full adds (bits 2, 1, 0)
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
    const uint32_t r = row0 + (uint32_t)(4 * (8 * qb + 4 * (int)up5 + 2 * (int)up4 + (int)up3) + wv);
    const float e = (float)(__builtin_sqrt(t) * (1.0 + 1e-6));  // rounded up
    if (e == e) worst = __builtin_fmaxf(worst, e);
    if ((lane & 7) == 0 && r < n) err_rows[r] = e == e ? e : __builtin_inff();  // NaN -> +inf: such a row prunes nothing
  }
  for (int o = 32; o > 0; o >>= 1) worst = __builtin_fmaxf(worst, __shfl_xor(worst, o, 64));
  if (lane == 0) raise_max_u32(err_bits, __float_as_uint(worst));  // non-negative floats order like their bits
  __syncthreads();
  {
    zp_v16f32 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float *xa = xf + (size_t)(32 * g + lc) * 129 + 64 * h + 32 * lh;
#pragma unroll
    for (int t = 0; t < 32; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[t], qreg[t], acc, 0, 0, 0);
    // acc[r] of lane (lh, lc) = the product of row 32 g + 4 lh + (r & 3) + 8 (r >> 2) with direction lc over the wave's k half
    __syncthreads();  // (every wave is done with the operand: its place is the products' now)
    float *dstz = h ? zp : zs;
#pragma unroll
    for (int r = 0; r < 16; ++r) dstz[(32 * g + 4 * lh + (r & 3) + 8 * (r >> 2)) * 33 + lc] = acc[r];
  }
  __syncthreads();
  const int rr = tid >> 2, part = tid & 3;
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {  // the two k halves in a fixed order
    const int j = 8 * part + jj;
    zs[rr * 33 + j] = zs[rr * 33 + j] + zp[rr * 33 + j];
  }
  __syncthreads();
  const uint32_t row = row0 + (uint32_t)rr;
  double tails = 0.0;
  if (row < n) {
    double *o = out + (size_t)row * 32;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
      if (8 * part + jj < kDirs) o[8 * part + jj] = (double)zs[rr * 33 + 8 * part + jj];  // (each thread a contiguous 64 bytes of the row)
    if (part == 0) {
      const double nr = norms[row];
      const bool live = nr > 0.0 && nr < __builtin_inf();
      float zz = 0.f;
      for (int j = 0; j < kDirs; ++j) {
        const float z = zs[rr * 33 + j];
        zz = __builtin_fmaf(z, z, zz);
      }
      o[kDirs] = live ? (double)__builtin_sqrtf(__builtin_fmaxf(1.f - zz, 0.f) + 2e-5f) : 0.0;
      if (live) tails = (double)__builtin_fmaxf(1.f - zz, 0.f);
    }
  }
  for (int o2 = 32; o2 > 0; o2 >>= 1) tails += __shfl_xor(tails, o2, 64);
  if (tail_part && lane == 0) tail_part[blockIdx.x * 4 + wv] = tails;
}

// One round of the subspace iteration on a sample of rows of exactly 128 values in ONE kernel (round 6): zproj_kernel (a wave a row,
// 186 shuffles a row) + yacc_kernel (32 of its 256 threads at work at 128 columns) took 54 + 37 us for 16 384 sample rows, four
// times a build.  Here a workgroup stages 64 sample rows as f32 unit rows in LDS, a thread takes eight directions of one row for the
// coordinates (the row's values come conflict-free out of LDS, the directions are a broadcast), and then 15.5 of the 31 x 128
// elements of the block's part of Y = S^T Z, each a sum over its 64 rows in row order.  yreduce_kernel adds the blocks' parts in its
// fixed tree.  (f32 sums in another order than the two kernels': the directions differ in their last bits -- any basis orthonormal
// to 1e-6 serves, and this one is the same in every run and on every rank.)
__global__ __launch_bounds__(256) void zy_sample128_kernel(const double *__restrict__ R, const double *__restrict__ norms, uint64_t stride, uint32_t count,
                                                           const float *__restrict__ Q, uint32_t pitch, float *__restrict__ Ypart) {
  __shared__ float s_u[64 * 129];
  __shared__ float s_q[kDirs * 129];
  __shared__ float s_z[64 * 33];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint32_t i0 = blockIdx.x * 64u;
  for (int e = tid; e < kDirs * 128; e += 256) s_q[(e >> 7) * 129 + (e & 127)] = Q[(size_t)(e >> 7) * pitch + (e & 127)];
  typedef double d2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int q = 0; q < 16; ++q) {  // wave wv takes rows wv, wv + 4, ...: lane l the values 2 l, 2 l + 1
    const uint32_t lr = (uint32_t)(4 * q + wv), i = i0 + lr;
    float f0 = 0.f, f1 = 0.f;
    if (i < count) {
      const uint64_t row = (uint64_t)i * stride;
      const double nr = norms[row];
      if (nr > 0.0 && nr < __builtin_inf()) {
        const d2 v = reinterpret_cast<const d2 *>(R + row * 128)[lane];
        f0 = (float)(v.x / nr);
        f1 = (float)(v.y / nr);
        f0 = f0 == f0 ? f0 : 0.f;
        f1 = f1 == f1 ? f1 : 0.f;
      }
    }
    s_u[lr * 129 + 2 * lane] = f0;
    s_u[lr * 129 + 2 * lane + 1] = f1;
  }
  __syncthreads();
  {  // coordinates: thread (row rr, part) -> directions 8 part .. 8 part + 7
    const int rr = tid >> 2, part = tid & 3;
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.f;
    for (int k = 0; k < 128; ++k) {
      const float x = s_u[rr * 129 + k];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (8 * part + j < kDirs) a[j] = __builtin_fmaf(x, s_q[(8 * part + j) * 129 + k], a[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (8 * part + j < kDirs) s_z[rr * 33 + 8 * part + j] = a[j];
  }
  __syncthreads();
  float *yp = Ypart + (size_t)blockIdx.x * kDirs * pitch;
  for (int e = tid; e < kDirs * 128; e += 256) {  // (dir, k): consecutive threads consecutive k
    const int dir = e >> 7, k = e & 127;
    float y = 0.f;
#pragma unroll 8
    for (int r = 0; r < 64; ++r) y = __builtin_fmaf(s_z[r * 33 + dir], s_u[r * 129 + k], y);
    yp[(size_t)dir * pitch + k] = y;
  }
}

// Y[block][dir][k] = sum over the sample rows of the block (i = block, block + grid, ...: a fixed order) of z[row][dir] * u[row][k];
// yreduce_kernel adds the blocks' parts in block order.  (Was a float atomicAdd into one Y: the directions differed in their last
// bits from run to run, and with them the row order and the routing of rows through the stages.)
__global__ __launch_bounds__(256) void yacc_kernel(const double *__restrict__ R, const double *__restrict__ norms, uint64_t stride,
                                                   uint32_t count, uint32_t d, const double *__restrict__ Z, uint32_t pitch,
                                                   float *__restrict__ Y) {
  __shared__ float s_z[32];
  float acc[kDirs][4];
#pragma unroll
  for (int a = 0; a < kDirs; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = 0.f;
  for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
    const uint64_t row = (uint64_t)i * stride;
    const double nr = norms[row];
    __syncthreads();
    if (threadIdx.x < 32) s_z[threadIdx.x] = (float)Z[(size_t)i * 32 + threadIdx.x];
    __syncthreads();
    if (!(nr > 0.0 && nr < __builtin_inf())) continue;
    float u[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint32_t k = threadIdx.x + 256u * c;
      const float v = k < d ? (float)(R[row * d + k] / nr) : 0.f;
      u[c] = v == v ? v : 0.f;
    }
#pragma unroll
    for (int a = 0; a < kDirs; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[a][c] = __builtin_fmaf(s_z[a], u[c], acc[a][c]);
  }
#pragma unroll
  for (int a = 0; a < kDirs; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const uint32_t k = threadIdx.x + 256u * c;
      if (k < d) Y[((size_t)blockIdx.x * kDirs + a) * pitch + k] = acc[a][c];
    }
}
// (a fixed TREE: 16 threads per element add 1/16 of the blocks each, in block order, and one of them adds the 16 sums in
// order -- the same sum in every run, 48 dependent additions instead of 512)
__global__ __launch_bounds__(256) void yreduce_kernel(const float *__restrict__ part, uint32_t blocks, uint32_t elems, float *__restrict__ Y) {
  __shared__ float s_p[16][17];
  const uint32_t el = threadIdx.x & 15u, j = threadIdx.x >> 4;  // 16 elements x 16 slices per workgroup
  const uint32_t e = blockIdx.x * 16u + el;
  const uint32_t per = (blocks + 15u) / 16u;
  float s = 0.f;
  if (e < elems)
    for (uint32_t b = j * per; b < min(blocks, (j + 1) * per); ++b) s += part[(size_t)b * elems + e];
  s_p[el][j] = s;
  __syncthreads();
  if (j == 0 && e < elems) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += s_p[el][q];
    Y[e] = t;
  }
}

// out[0] = the sum of `count` partial tail sums (thread t adds parts t, t + 1024, ... in that order, then a fixed tree over the
// threads); out[1] = the float whose bits err_bits holds (the largest plane error of a packing pass), or 0
__global__ __launch_bounds__(1024) void tail_reduce_kernel(const double *__restrict__ parts, uint32_t count, const unsigned int *__restrict__ err_bits,
                                                           double *__restrict__ out) {
  __shared__ double s_t[1024];
  double s = 0.0;
  for (uint32_t i = threadIdx.x; i < count; i += 1024) s += parts[i];
  s_t[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s_t[threadIdx.x] += s_t[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = s_t[0];
    out[1] = err_bits ? (double)__uint_as_float(*err_bits) : 0.0;
  }
}
__global__ void znorm_kernel(const double *__restrict__ Z, uint32_t n, double *__restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int j = 0; j < 32; ++j) s += Z[(size_t)i * 32 + j] * Z[(size_t)i * 32 + j];
  out[i] = __builtin_sqrt(s);
}

// ---- a build sharded by rows (ShardedRows, common.h) --------------------------------------------------------------------------
// the rank's share of the strided sample, side by side: S[i] = row i * stride for the sample indices [i0, i1) whose rows it holds
__global__ void copy_sample_rows_kernel(const double *__restrict__ own, uint64_t lo, uint64_t stride, uint64_t i0, uint64_t i1, uint32_t d,
                                        double *__restrict__ S) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (i1 - i0) * d) return;
  const uint64_t i = i0 + e / d, k = e % d;
  S[i * d + k] = own[(i * stride - lo) * d + k];
}
__global__ void sample_norms_kernel(const double *__restrict__ norms, uint64_t stride, uint32_t ns, double *__restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ns) out[i] = norms[(uint64_t)i * stride];
}
// the coordinates are f32 values kept in doubles (zproj_*_kernel): they cross the links as f32, exactly
__global__ void z_to_f32_kernel(const double *__restrict__ Z, uint64_t count, float *__restrict__ out) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < count) out[e] = (float)Z[e];
}
__global__ void z_from_f32_kernel(const float *__restrict__ in, uint64_t count, double *__restrict__ Z) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < count) Z[e] = (double)in[e];
}

typedef int i32x4 __attribute__((ext_vector_type(4)));

// the int8 image (one k step: mfma_prefilter_i8.hip) of image position p = row perm[p] (or p) from the kept coordinates:
// q_k = rint(127 z_k), the tail upwards; err_bits: the largest || q / 127 - z || of a row
__global__ __launch_bounds__(256) void pack_from_z_kernel(const double *__restrict__ Z, const uint32_t *__restrict__ perm, uint32_t n,
                                                          uint32_t positions, i32x4 *__restrict__ P8, unsigned int *__restrict__ err_bits) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  uint32_t bytes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float e2 = 0.f;
  if (p < n) {
    const double *z = Z + (size_t)(perm ? perm[p] : p) * 32;
    for (int j = 0; j < kDirs; ++j) {
      const float zj = (float)z[j];
      const float q = __builtin_fminf(__builtin_fmaxf(__builtin_rintf(zj * 127.f), -127.f), 127.f);
      const float e = q * (1.f / 127.f) - zj;
      e2 = __builtin_fmaf(e, e, e2);
      bytes[j >> 2] |= ((uint32_t)(int)q & 0xFFu) << (8 * (j & 3));
    }
    const float t = (float)z[kDirs];
    const int tq = t > 0.f ? (int)__builtin_fminf(__builtin_ceilf(t * 127.f), 127.f) : 0;
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

// The same coordinates on the matrix pipe, for rows of 256 .. 1024 columns (a multiple of 32): Z = X Q is a GEMM of 2 n d 32 flop
// behind n d 8 bytes, but zproj_kernel above gives a row to a wave and ends each of its 31 products in a wave-wide reduction: 14 ms
// for 1 M x 768, ten times the time of its bytes (round 5: the largest stage of the wide build's preparation).  Here a wave owns 32
// rows; a lane (row r = lane & 31, half h = lane >> 5) loads the 16 consecutive columns 32 c + 16 h .. of its row -- whole 64-byte
// sectors -- and step s of v_mfma_f32_32x32x2_f32 takes each lane's s-th value: the k index of the instruction's two slices is then
// 32 c + 16 h + s on both operands (the order in which k is summed is free), the directions coming from LDS in [k][32] form.  f32
// products and sums as before (the unit row is formed in f64 first); the epilogue is the other kernel's: t = sqrt(max(0, 1 - |z|^2)
// + 2e-5), rows without a direction all zero, the wave's sum of 1 - |z|^2 in a fixed order.
typedef float zp_v16f32 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void zproj_mfma_kernel(const double *__restrict__ R, const double *__restrict__ norms, uint32_t n, uint32_t d,
                                                         const float *__restrict__ Q, uint32_t pitch, double *__restrict__ out,
                                                         double *__restrict__ tail_part) {
  extern __shared__ float zp_smem[];
  float *s_q = zp_smem;                       // [d][33]: s_q[k * 33 + dir], dir 31 = 0
  float *s_z = zp_smem + (size_t)d * 33;      // [8 waves][32 rows][33]
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6, r = lane & 31u, h = lane >> 5;
  for (uint32_t e = tid; e < d * 32u; e += 512) {
    const uint32_t k = e >> 5, dir = e & 31u;
    s_q[k * 33 + dir] = dir < (uint32_t)kDirs ? Q[(size_t)dir * pitch + k] : 0.f;
  }
  __syncthreads();
  const uint32_t row0 = (blockIdx.x * 8u + wv) * 32u;
  double tails = 0.0;
  if (row0 < n) {
    const uint32_t row = min(row0 + r, n - 1u);
    const double nr = norms[row];
    const bool live = nr > 0.0 && nr < __builtin_inf() && row0 + r < n;
    const double inv = live ? 1.0 / nr : 0.0;
    const double *src = R + (size_t)row * d + 16u * h;
    zp_v16f32 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (uint32_t c = 0; c < d / 32u; ++c) {
      double x[16];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const double2 v = *reinterpret_cast<const double2 *>(src + 32u * c + 2 * q);
        x[2 * q] = v.x, x[2 * q + 1] = v.y;
      }
      const float *bq = s_q + (size_t)(32u * c + 16u * h) * 33 + r;
#pragma unroll
      for (int sidx = 0; sidx < 16; ++sidx) {
        const float a = live ? (float)(x[sidx] * inv) : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a == a ? a : 0.f, bq[sidx * 33], acc, 0, 0, 0);
      }
    }
    // accumulator register i of lane (h, r): row 8 (i >> 2) + (i & 3) + 4 h of the wave's 32, direction r
    float *zt = s_z + (size_t)wv * 32 * 33;
#pragma unroll
    for (int i = 0; i < 16; ++i) zt[(8 * (i >> 2) + (i & 3) + 4 * (int)h) * 33 + r] = acc[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (h == 0) {
      const uint32_t mine = row0 + r;
      const double nm = mine < n ? norms[mine] : 0.0;
      const bool lv = mine < n && nm > 0.0 && nm < __builtin_inf();
      float zz = 0.f;
      float z[32];
#pragma unroll
      for (int k = 0; k < kDirs; ++k) {
        z[k] = lv ? zt[r * 33 + k] : 0.f;
        zz = __builtin_fmaf(z[k], z[k], zz);
      }
      z[kDirs] = lv ? __builtin_sqrtf(__builtin_fmaxf(1.f - zz, 0.f) + 2e-5f) : 0.f;
      if (mine < n) {
        double *o = out + (size_t)mine * 32;
#pragma unroll
        for (int k = 0; k < 32; ++k) o[k] = (double)z[k];
      }
      if (lv) tails = (double)__builtin_fmaxf(1.f - zz, 0.f);
    }
  }
  if (tail_part) {  // lanes 0 .. 31 of the wave, added in lane order (a fixed order: the same tail in every run)
    double tsum = 0.0;
    for (int l = 0; l < 32; ++l) tsum += __shfl(tails, l, 64);
    if (lane == 0) tail_part[blockIdx.x * 8u + wv] = tsum;
  }
}
static bool zproj_mfma_applies(uint32_t d) { return d >= 256 && d <= 960 && d % 32 == 0; }  // (the directions of 960 columns and the eight waves' tiles fill the 160 KB of LDS)
static uint32_t zproj_mfma_parts(uint32_t n) { return cdiv(n, 256) * 8u; }  // entries of tail_part
static int launch_zproj_mfma(as_ctx *ctx, const double *R, const double *norms, uint32_t n, uint32_t d, const float *Q, uint32_t pitch, double *out,
                             double *tail_part) {
  const size_t lds = ((size_t)d * 33 + (size_t)8 * 32 * 33) * sizeof(float);
  AS_HIP(ctx, hipFuncSetAttribute((const void *)zproj_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(zproj_mfma_kernel, dim3(cdiv(n, 256)), dim3(512), lds, ctx->stream, R, norms, n, d, Q, pitch, out, tail_part);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

template <class... A>
int launch_zproj(as_ctx *ctx, uint32_t nv, unsigned grid, size_t lds, A... a) {
#define AS_Z(NV)                                                                                                              \
  case NV:                                                                                                                    \
    AS_HIP(ctx, hipFuncSetAttribute((const void *)zproj_kernel<NV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));  \
    hipLaunchKernelGGL(zproj_kernel<NV>, dim3(grid), dim3(256), lds, ctx->stream, a...);                                      \
    break;
  switch (nv) {
    AS_Z(1) AS_Z(2) AS_Z(3) AS_Z(4) AS_Z(6) AS_Z(8) AS_Z(12) AS_Z(16)
    default: return fail(ctx, AS_ERR_UNSUPPORTED, "rotated coordinates: width not covered");
  }
#undef AS_Z
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

}  // namespace

bool rotated_wide_supported(uint32_t d) { return d > 32 && d <= kMaxWide; }  // (up to 32 wide the rows themselves are one k step)

// Z: n x 32 doubles (device), znorm: n doubles (their norms, ~1): the rotated coordinates of all rows.  tail_out: the share of
// the sample's energy outside the 31 directions (the caller decides whether the bound is worth using).
// sr (a build sharded by rows; `rows` is not read, `norms` are those of ALL rows; rows up to 128 wide): the directions come from
// the same strided sample -- every rank contributes the sample rows it holds, 16 MB in all -- each rank projects its OWN rows, and
// the coordinates (128 B per row) and the per-block tail sums are all-gathered: the same Z, norms and tail as on one GPU.
// Q_out (optional, device, 31 x rotated_wide_pitch(d) floats): the directions themselves, for rotated_project_wide on other rows.
uint32_t rotated_wide_pitch(uint32_t d) { return (d + 63) / 64 * 64; }
static uint32_t zproj_width(uint32_t pitch) {
  uint32_t nv = pitch / 64;  // round up to an instantiated width
  for (uint32_t c : {1u, 2u, 3u, 4u, 6u, 8u, 12u, 16u})
    if (nv <= c) return c;
  return nv;
}
// packed (or null): where the rows are exactly 128 values wide and the build is not sharded, the pass over all rows also leaves
// their f16 form row-major in ROW order (packed->Hrow, n x 128 halves), every row's plane error (packed->err_rows, by row) and the
// largest of them (packed->f16_err), and sets packed->done -- zproj_pack_rows128_kernel: one read of the rows instead of two.
int rotated_rows_wide(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, double *Z, double *znorm,
                      double *tail_out, const ShardedRows *sr, float *Q_out, RotatedPacked *packed) {
  if (packed) packed->done = false;
  if (!rotated_wide_supported(d)) return fail(ctx, AS_ERR_UNSUPPORTED, "rotated coordinates: width not covered");
  if (sr && d > 128) return fail(ctx, AS_ERR_UNSUPPORTED, "rotated coordinates of a sharded build: rows up to 128 wide");
  const uint32_t pitch = (d + 63) / 64 * 64;
  uint32_t nv = pitch / 64;  // round up to an instantiated width
  for (uint32_t c : {1u, 2u, 3u, 4u, 6u, 8u, 12u, 16u})
    if (nv <= c) {
      nv = c;
      break;
    }
  const size_t lds = (size_t)kDirs * pitch * sizeof(float);
  const uint64_t stride = std::max<uint64_t>(1, n / (uint64_t)std::max<int64_t>(1024, std::min<int64_t>(ctx->opt_join_basis_sample, 65536)));
  const uint32_t ns = (uint32_t)((n + stride - 1) / stride);
  DevBuf<float> Qd, Yd, Ypart;
  DevBuf<double> Zs, tpart;
  DevBuf<unsigned int> err_bits;
  // (rows of exactly 128 values: a round of the iteration is one kernel over blocks of 64 sample rows, zy_sample128_kernel)
  const bool fused_round = d == 128 && ctx->opt_join_basis_fused && (((uintptr_t)(sr ? sr->own : rows)) & 15) == 0;
  const uint32_t y_blocks = fused_round ? (uint32_t)cdiv(ns, 64) : std::min<uint32_t>(ns, 512);
  // (128 aligned columns: zproj_rows128_kernel, blocks of 64 rows; other widths up to 128: blocks of 256)
  const bool rows128 = d == 128 && (((uintptr_t)(sr ? sr->own : rows)) & 15) == 0 && !(ctx->opt_debug & 2048);
  const uint32_t z_rows = rows128 ? 64u : 256u;
  const uint32_t z_grid = d <= 128 ? (uint32_t)cdiv(n, z_rows) : std::min<uint32_t>((uint32_t)cdiv(n, 4), 256 * 8);
  AS_TRY(Qd.alloc(ctx, (size_t)kDirs * pitch));
  AS_TRY(Yd.alloc(ctx, (size_t)kDirs * pitch));
  AS_TRY(Ypart.alloc(ctx, (size_t)y_blocks * kDirs * pitch));
  AS_TRY(Zs.alloc(ctx, (size_t)ns * 32));
  AS_TRY(tpart.alloc(ctx, (size_t)z_grid * 4));
  uint32_t tail_parts = 0;  // (entries of tpart where the projection of all rows has a grid of its own)
  // the sample the directions are found on: rows 0, stride, 2 stride, ... of the matrix, or (sharded) the same rows side by side
  const double *s_rows = rows, *s_norms = norms;
  uint64_t s_stride = stride;
  DevBuf<double> S, Sn;
  if (sr) {
    AS_TRY(S.alloc(ctx, (size_t)ns * d));
    AS_TRY(Sn.alloc(ctx, ns));
    std::vector<uint64_t> bytes((size_t)sr->world), offs((size_t)sr->world);
    for (int r = 0; r < sr->world; ++r) {
      const uint64_t i0 = (sr->row_lo[(size_t)r] + stride - 1) / stride, i1 = (sr->row_hi[(size_t)r] + stride - 1) / stride;
      bytes[(size_t)r] = (i1 - i0) * d * sizeof(double);
      offs[(size_t)r] = i0 * d * sizeof(double);
    }
    const uint64_t i0 = ((uint64_t)sr->lo + stride - 1) / stride, i1 = ((uint64_t)sr->lo + sr->count + stride - 1) / stride;
    if (i1 > i0)
      hipLaunchKernelGGL(copy_sample_rows_kernel, dim3(cdiv((i1 - i0) * d, 256)), dim3(256), 0, ctx->stream, sr->own, (uint64_t)sr->lo, stride, i0, i1, d, S.p);
    hipLaunchKernelGGL(sample_norms_kernel, dim3(cdiv(ns, 256)), dim3(256), 0, ctx->stream, norms, stride, ns, Sn.p);
    AS_HIP(ctx, hipGetLastError());
    AS_TRY(sr->gather_v(S.p, bytes.data(), offs.data()));
    s_rows = S.p;
    s_norms = Sn.p;
    s_stride = 1;
  }
  std::vector<float> Q((size_t)kDirs * pitch, 0.f), Yh((size_t)kDirs * pitch);
  std::vector<double> V((size_t)kDirs * d);
  uint64_t seed = 0x9E3779B97F4A7C15ull;
  for (auto &v : V) {
    seed = seed * 6364136223846793005ull + 1442695040888963407ull;
    v = (double)((seed >> 11) & 0xFFFFF) / 1048576.0 - 0.5;
  }
  // (eight partial sums in a fixed order: a single chain of d dependent additions per product made the four calls 0.8 ms of
  // host time with the GPU idle; any basis orthonormal to 1e-6 serves, and this one is the same in every run and on every rank)
  auto dot8 = [&](const double *a, const double *b) -> double {
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t k = 0;
    for (; k + 8 <= d; k += 8)
      for (int q = 0; q < 8; ++q) s[q] += a[k + q] * b[k + q];
    for (; k < d; ++k) s[k & 7] += a[k] * b[k];
    return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  };
  auto orthonormalise = [&]() {  // modified Gram-Schmidt, twice; a direction the others span is left out (zero)
    for (int j = 0; j < kDirs; ++j) {
      double *vj = &V[(size_t)j * d];
      for (int pass = 0; pass < 2; ++pass)
        for (int i = 0; i < j; ++i) {
          const double *vi = &V[(size_t)i * d];
          const double dot = dot8(vi, vj);
          for (uint32_t k = 0; k < d; ++k) vj[k] -= dot * vi[k];
        }
      const double nn = dot8(vj, vj);
      const double inv = nn > 1e-300 ? 1.0 / std::sqrt(nn) : 0.0;
      for (uint32_t k = 0; k < d; ++k) vj[k] *= inv;
    }
    for (int j = 0; j < kDirs; ++j)
      for (uint32_t k = 0; k < pitch; ++k) Q[(size_t)j * pitch + k] = k < d ? (float)V[(size_t)j * d + k] : 0.f;
  };
  {
    // the starting basis is a function of d alone (a fixed generator, a fixed Gram-Schmidt): computed once per width and process --
    // its 31 x 30 sequential projections took 48 us of every build with the GPU idle
    static std::mutex start_mu;
    static std::map<uint32_t, std::pair<std::vector<double>, std::vector<float>>> start_basis;
    std::lock_guard<std::mutex> lock(start_mu);
    auto it = start_basis.find(d);
    if (it == start_basis.end()) {
      orthonormalise();
      it = start_basis.emplace(d, std::make_pair(V, Q)).first;
    }
    V = it->second.first;
    Q = it->second.second;
  }
  const int iters = (int)std::max<int64_t>(1, std::min<int64_t>(ctx->opt_join_basis_iters, 8));
  for (int it = 0; it < iters; ++it) {  // Q <- orth(S^T S Q)
    AS_HIP(ctx, hipMemcpyAsync(Qd.p, Q.data(), Q.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (fused_round) {
      hipLaunchKernelGGL(zy_sample128_kernel, dim3(y_blocks), dim3(256), 0, ctx->stream, s_rows, s_norms, s_stride, ns, (const float *)Qd.p, pitch, Ypart.p);
    } else {
      AS_HIP(ctx, hipMemsetAsync(Ypart.p, 0, (size_t)y_blocks * kDirs * pitch * sizeof(float), ctx->stream));  // (columns d .. pitch stay 0)
      AS_TRY(launch_zproj(ctx, nv, std::min<uint32_t>(cdiv(ns, 4), 1024), lds, s_rows, s_norms, (uint64_t)0, s_stride, ns, d, (const float *)Qd.p,
                          pitch, Zs.p, (double *)nullptr));
      hipLaunchKernelGGL(yacc_kernel, dim3(y_blocks), dim3(256), 0, ctx->stream, s_rows, s_norms, s_stride, ns, d,
                         (const double *)Zs.p, pitch, Ypart.p);
    }
    hipLaunchKernelGGL(yreduce_kernel, dim3(cdiv((uint32_t)kDirs * pitch, 16)), dim3(256), 0, ctx->stream, (const float *)Ypart.p, y_blocks,
                       (uint32_t)kDirs * pitch, Yd.p);
    AS_HIP(ctx, hipGetLastError());
    AS_HIP(ctx, copy_d2h(ctx, Yh.data(), Yd.p, Yh.size() * sizeof(float)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int j = 0; j < kDirs; ++j)
      for (uint32_t k = 0; k < d; ++k) V[(size_t)j * d + k] = (double)Yh[(size_t)j * pitch + k];
    orthonormalise();
  }
  AS_HIP(ctx, hipMemcpyAsync(Qd.p, Q.data(), Q.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  if (d <= 128) {
    // (sharded: the rank's own rows into their place in Z and in the per-block tail sums -- shard starts are multiples of 512, so
    // its blocks of 256 rows are blocks of the whole matrix)
    const double *p_rows = sr ? sr->own : rows, *p_norms = sr ? norms + sr->lo : norms;
    const uint32_t p_n = sr ? sr->count : n;
    double *p_z = sr ? Z + (size_t)sr->lo * 32 : Z, *p_t = sr ? tpart.p + (size_t)(sr->lo / z_rows) * 4 : tpart.p;
#define AS_ZT(DPV) hipLaunchKernelGGL(zproj_tile_kernel<DPV>, dim3(cdiv(p_n, 256)), dim3(256), 0, ctx->stream, p_rows, p_norms, p_n, d, (const float *)Qd.p, pitch, p_z, p_t)
    if (p_n && rows128 && packed && packed->Hrow && packed->err_rows) {  // (sharded: the rank's own rows into their place, as Z below)
      AS_TRY(err_bits.alloc(ctx, 1));
      AS_HIP(ctx, hipMemsetAsync(err_bits.p, 0, sizeof(unsigned int), ctx->stream));
      ProfScope ps(ctx, "pack_unit_rows");  // (the packing's share cannot be told apart: the whole pass is booked here, under the basis stage)
      hipLaunchKernelGGL(zproj_pack_rows128_kernel, dim3(cdiv(p_n, 64)), dim3(256), 0, ctx->stream, p_rows, p_norms, p_n, (const float *)Qd.p, pitch, p_z, p_t,
                         (_Float16 *)packed->Hrow + (sr ? (size_t)sr->lo * 128 : 0), packed->err_rows + (sr ? sr->lo : 0), err_bits.p);
      packed->done = true;
    } else if (!p_n && rows128 && packed && packed->Hrow && packed->err_rows && sr) {
      packed->done = true;  // (a rank without rows: nothing of its own to pack, the others' rows arrive with the gather)
    } else if (p_n && rows128)
      hipLaunchKernelGGL(zproj_rows128_kernel, dim3(cdiv(p_n, 64)), dim3(256), 0, ctx->stream, p_rows, p_norms, p_n, (const float *)Qd.p, pitch, p_z, p_t);
    else if (p_n) switch ((d + 31) / 32) {
      case 2: AS_ZT(64); break;
      case 3: AS_ZT(96); break;
      default: AS_ZT(128); break;
    }
#undef AS_ZT
    AS_HIP(ctx, hipGetLastError());
    if (sr) {
      DevBuf<float> zf;
      AS_TRY(zf.alloc(ctx, (size_t)n * 32));
      if (p_n)
        hipLaunchKernelGGL(z_to_f32_kernel, dim3(cdiv((uint64_t)p_n * 32, 256)), dim3(256), 0, ctx->stream, (const double *)p_z, (uint64_t)p_n * 32,
                           zf.p + (size_t)sr->lo * 32);
      AS_TRY(sr->gather(zf.p, 32 * sizeof(float)));
      hipLaunchKernelGGL(z_from_f32_kernel, dim3(cdiv((uint64_t)n * 32, 256)), dim3(256), 0, ctx->stream, (const float *)zf.p, (uint64_t)n * 32, Z);
      AS_HIP(ctx, hipGetLastError());
      std::vector<uint64_t> bytes((size_t)sr->world), offs((size_t)sr->world);
      for (int r = 0; r < sr->world; ++r) {
        const uint64_t b0 = sr->row_lo[(size_t)r] / z_rows, b1 = (sr->row_hi[(size_t)r] + z_rows - 1) / z_rows;
        bytes[(size_t)r] = sr->row_hi[(size_t)r] > sr->row_lo[(size_t)r] ? (b1 - b0) * 4 * sizeof(double) : 0;
        offs[(size_t)r] = b0 * 4 * sizeof(double);
      }
      AS_TRY(sr->gather_v(tpart.p, bytes.data(), offs.data()));
    }
  } else if (zproj_mfma_applies(d) && ((uintptr_t)rows & 15) == 0 && ctx->opt_join_zproj_mfma) {
    AS_TRY(tpart.alloc(ctx, zproj_mfma_parts(n)));  // (a tail sum per wave of THIS kernel's grid)
    tail_parts = zproj_mfma_parts(n);
    AS_TRY(launch_zproj_mfma(ctx, rows, norms, n, d, (const float *)Qd.p, pitch, Z, tpart.p));
  } else {
    AS_TRY(launch_zproj(ctx, nv, z_grid, lds, rows, norms, (uint64_t)0, (uint64_t)1, n, d, (const float *)Qd.p, pitch, Z, tpart.p));
  }
  hipLaunchKernelGGL(znorm_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, (const double *)Z, n, znorm);
  AS_HIP(ctx, hipGetLastError());
  if (Q_out) AS_HIP(ctx, hipMemcpyAsync(Q_out, Qd.p, sizeof(float) * (size_t)kDirs * pitch, hipMemcpyDeviceToDevice, ctx->stream));
  // the tail sums added up on the device (a fixed tree: the same value in every run and on every rank) and read back as ONE word --
  // the half megabyte of per-wave parts used to cross to the host and be added there, 0.14 ms with the GPU idle -- together with
  // the largest plane error when the pass packed the rows
  const uint32_t n_parts = tail_parts ? tail_parts : z_grid * 4;
  DevBuf<double> tsum;
  AS_TRY(tsum.alloc(ctx, 2));
  hipLaunchKernelGGL(tail_reduce_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const double *)tpart.p, n_parts, (const unsigned int *)err_bits.p, tsum.p);
  AS_HIP(ctx, hipGetLastError());
  double h[2] = {0.0, 0.0};
  AS_HIP(ctx, copy_d2h(ctx, h, tsum.p, sizeof(h)));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (Q and the locals are done with)
  *tail_out = h[0] / std::max<double>(1.0, (double)n);
  if (packed && packed->done) packed->f16_err = err_bits.p ? h[1] : 0.0;
  return AS_OK;
}

// The rotated coordinates of OTHER rows (the queries of a search batch) in directions found before (Q: 31 x rotated_wide_pitch(d)
// floats, rotated_rows_wide's Q_out): Z (n x 32: 31 coordinates and the norm of what they leave out of the unit row), znorm.
int rotated_project_wide(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, const float *Q, double *Z, double *znorm) {
  if (!rotated_wide_supported(d)) return fail(ctx, AS_ERR_UNSUPPORTED, "rotated coordinates: width not covered");
  if (n == 0) return AS_OK;
  const uint32_t pitch = rotated_wide_pitch(d), nv = zproj_width(pitch);
  const size_t lds = (size_t)kDirs * pitch * sizeof(float);
  const uint32_t z_grid = std::min<uint32_t>((uint32_t)cdiv(n, 4), 256 * 8);
  DevBuf<double> tpart;
  if (zproj_mfma_applies(d) && ((uintptr_t)rows & 15) == 0 && ctx->opt_join_zproj_mfma && n >= 4096) {
    AS_TRY(launch_zproj_mfma(ctx, rows, norms, n, d, Q, pitch, Z, nullptr));
  } else {
    AS_TRY(tpart.alloc(ctx, (size_t)z_grid * 4));
    AS_TRY(launch_zproj(ctx, nv, z_grid, lds, rows, norms, (uint64_t)0, (uint64_t)1, n, d, Q, pitch, Z, tpart.p));
  }
  hipLaunchKernelGGL(znorm_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, (const double *)Z, n, znorm);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

// P8: tiles * 128 positions (zero beyond n); err_out: the largest quantisation error norm of a row
int pack_rotated_from_z(as_ctx *ctx, const double *Z, const uint32_t *perm, uint32_t n, uint32_t tiles, void *P8, double *err_out) {
  DevBuf<unsigned int> bits;
  AS_TRY(bits.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(bits.p, 0, sizeof(unsigned int), ctx->stream));
  const uint32_t positions = tiles * 128;
  hipLaunchKernelGGL(pack_from_z_kernel, dim3(cdiv(positions, 256)), dim3(256), 0, ctx->stream, Z, perm, n, positions, (i32x4 *)P8, bits.p);
  AS_HIP(ctx, hipGetLastError());
  unsigned int h = 0;
  AS_HIP(ctx, copy_d2h(ctx, &h, bits.p, sizeof(h)));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  float e;
  std::memcpy(&e, &h, sizeof(e));
  *err_out = (double)e;
  return AS_OK;
}

}  // namespace asx
