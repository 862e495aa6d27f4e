// knn_exact.hip -- exact row norms and the exact neighbour scan (K1 + exact leg of K2).
// Replaces the CosinePair queries of src/laplacian.rs:211,221,246 with the rule set of
// SURVEY.md Appendix A/B (self excluded, (distance asc, index asc), cos := 0 on tiny norms).
#include "scan_topk.h"

namespace asx {

// One thread per row, sequential sum of squares (src/core.rs:212-216).  The rows of a block are staged through LDS
// sixteen columns at a time: lanes read 128-B segments of the rows (whole cache lines, used at once), then every
// thread adds the sixteen squares of its row in order.  (Each thread walking its own row in global memory made the
// pass 0.8 ms per million 128-wide rows; it runs four times per build + search step.)
constexpr int kNormRows = 256, kNormCols = 16, kNormPitch = kNormCols + 1;
__global__ __launch_bounds__(kNormRows) void row_norms_kernel(const double *__restrict__ rows, uint64_t n, uint64_t d,
                                                              double *__restrict__ norms) {
  __shared__ double tile[kNormRows * kNormPitch];
  const uint64_t row0 = (uint64_t)blockIdx.x * kNormRows;
  const int tid = threadIdx.x;
  double s = 0.0;
  for (uint64_t c0 = 0; c0 < d; c0 += kNormCols) {
    const int cols = (int)((d - c0) < (uint64_t)kNormCols ? (d - c0) : (uint64_t)kNormCols);
    __syncthreads();  // the chunk before has been consumed
#pragma unroll
    for (int q = 0; q < kNormCols; ++q) {  // element e = q * 256 + tid of the 256 x 16 chunk: row e / 16, column e % 16
      const int e = q * kNormRows + tid, r = e / kNormCols, c = e % kNormCols;
      if (row0 + r < n && c < cols) tile[r * kNormPitch + c] = rows[(row0 + r) * d + c0 + c];
    }
    __syncthreads();
    if (row0 + tid < n)
      for (int c = 0; c < cols; ++c) {
        const double v = tile[tid * kNormPitch + c];
        s = __dadd_rn(s, __dmul_rn(v, v));
      }
  }
  if (row0 + tid < n) norms[row0 + tid] = __dsqrt_rn(s);
}

// Rows of exactly 128 values, 16-byte aligned: a workgroup stages 64 WHOLE rows (64 KB: sixteen 16-byte loads per thread, every
// one of them a full 1-KB row per wave instruction, all in flight together), then one thread per row walks its row in LDS in
// index order -- the same sum.  The chunked form above moves a GB in 0.74 ms (eight rounds of loads and two barriers each per
// block); this one runs at the speed of the loads.
constexpr int kNorm128Rows = 64, kNorm128Pitch = 129;
__global__ __launch_bounds__(256) void row_norms_d128_kernel(const double *__restrict__ rows, uint64_t n, double *__restrict__ norms) {
  extern __shared__ __attribute__((aligned(16))) double tile128[];  // 64 x 129
  const uint64_t row0 = (uint64_t)blockIdx.x * kNorm128Rows;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  typedef double d2 __attribute__((ext_vector_type(2)));
  d2 v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {  // wave wv takes rows wv, wv + 4, ...: lane l the values 2 l, 2 l + 1
    const uint64_t r = row0 + (uint64_t)(4 * q + wv);
    v[q] = r < n ? reinterpret_cast<const d2 *>(rows + r * 128)[lane] : d2{0.0, 0.0};
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    double *dst = tile128 + (size_t)(4 * q + wv) * kNorm128Pitch + 2 * lane;
    dst[0] = v[q].x;
    dst[1] = v[q].y;
  }
  __syncthreads();
  if (tid < kNorm128Rows && row0 + tid < n) {
    const double *src = tile128 + (size_t)tid * kNorm128Pitch;
    double s = 0.0;
#pragma unroll 16
    for (int c = 0; c < 128; ++c) {
      const double x = src[c];
      s = __dadd_rn(s, __dmul_rn(x, x));
    }
    norms[row0 + tid] = __dsqrt_rn(s);
  }
}

// StandardScaler (src/laplacian.rs:146-151): per-column z-score.  smartcore's source is absent;
// the oracle's convention is used (population variance, zero std -> 1; PARITY UNPINNED).  One thread
// per column walks the rows in order (the sums are sequential, as the oracle's), lanes read
// consecutive columns of a row: coalesced.
__global__ void column_stats_kernel(const double *__restrict__ rows, uint64_t n, uint64_t d,
                                    double *__restrict__ mean, double *__restrict__ sd) {
  const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d) return;
  double m = 0.0;
  for (uint64_t r = 0; r < n; ++r) m = __dadd_rn(m, rows[r * d + c]);
  m = __ddiv_rn(m, (double)n);
  double v = 0.0;
  for (uint64_t r = 0; r < n; ++r) {
    const double t = __dsub_rn(rows[r * d + c], m);
    v = __dadd_rn(v, __dmul_rn(t, t));
  }
  double s = __dsqrt_rn(__ddiv_rn(v, (double)n));
  if (s == 0.0) s = 1.0;
  mean[c] = m;
  sd[c] = s;
}
__global__ void standardise_kernel(const double *__restrict__ rows, uint64_t total, uint64_t d,
                                   const double *__restrict__ mean, const double *__restrict__ sd,
                                   double *__restrict__ out) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total) return;
  const uint64_t c = g % d;
  out[g] = __ddiv_rn(__dsub_rn(rows[g], mean[c]), sd[c]);
}

int standardise_rows(as_ctx *ctx, const double *rows, uint64_t n, uint64_t d, double *out) {
  DevBuf<double> mean, sd;
  AS_TRY(mean.alloc(ctx, d));
  AS_TRY(sd.alloc(ctx, d));
  ProfScope ps(ctx, "standardise");
  hipLaunchKernelGGL(column_stats_kernel, dim3(cdiv(d, 64)), dim3(64), 0, ctx->stream, rows, n, d, mean.p, sd.p);
  hipLaunchKernelGGL(standardise_kernel, dim3(cdiv(n * d, 256)), dim3(256), 0, ctx->stream, rows, n * d, d, mean.p, sd.p,
                     out);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

// A few rows (the F feature rows of a centroid matrix, a handful of queries): a wave per row -- its lanes bring 512 values at a
// time into LDS side by side, lane 0 adds their squares in index order (the same sum).  The block form above walks ONE workgroup
// through d / 16 rounds of loads and barriers: 68 us for 128 rows of 256 values.
__global__ __launch_bounds__(256) void row_norms_few_kernel(const double *__restrict__ rows, uint64_t n, uint64_t d, double *__restrict__ norms) {
  __shared__ double chunk[4][512];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t r = (uint64_t)blockIdx.x * 4 + wv;
  if (r >= n) return;
  double s = 0.0;
  for (uint64_t c0 = 0; c0 < d; c0 += 512) {
    const uint32_t cols = (uint32_t)((d - c0) < 512 ? (d - c0) : 512);
    for (uint32_t c = lane; c < cols; c += 64) chunk[wv][c] = rows[r * d + c0 + c];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane == 0)
      for (uint32_t c = 0; c < cols; ++c) {
        const double v = chunk[wv][c];
        s = __dadd_rn(s, __dmul_rn(v, v));
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (lane == 0) norms[r] = __dsqrt_rn(s);
}

int row_norms(as_ctx *ctx, const double *rows, uint64_t n, uint64_t d, double *norms) {
  if (n == 0) return AS_OK;
  ProfScope ps(ctx, "row_norms");
  if (n <= 2048) {
    hipLaunchKernelGGL(row_norms_few_kernel, dim3(cdiv(n, 4)), dim3(256), 0, ctx->stream, rows, n, d, norms);
    AS_HIP(ctx, hipGetLastError());
    return AS_OK;
  }
  if (d == 128 && ((uintptr_t)rows & 15) == 0 && n >= 4096) {
    const size_t lds = (size_t)kNorm128Rows * kNorm128Pitch * sizeof(double);
    AS_HIP(ctx, hipFuncSetAttribute((const void *)row_norms_d128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(row_norms_d128_kernel, dim3(cdiv(n, kNorm128Rows)), dim3(256), lds, ctx->stream, rows, n, norms);
  } else
  hipLaunchKernelGGL(row_norms_kernel, dim3(cdiv(n, kNormRows)), dim3(kNormRows), 0, ctx->stream, rows, n, d, norms);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

// k > 64 is served by repeated passes, each continuing strictly after the last (distance, index) of the one before
// (the wave-resident list holds 64 entries), as the search scan does.
int knn_exact(as_ctx *ctx, const double *rows, const double *norms, uint64_t n, uint64_t d,
              const uint32_t *row_ids, uint64_t row_begin, uint64_t row_count, uint32_t K,
              int rectify, uint32_t *out_idx, double *out_dist, uint64_t /*unused*/) {
  if (row_count == 0) return AS_OK;
  if (K == 0 || K > 256) return fail(ctx, AS_ERR_UNSUPPORTED, "topk + 1 must be in 1..256");
  DevBuf<int> err;
  AS_TRY(err.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(err.p, 0, sizeof(int), ctx->stream));
  GraphDistPolicy pol{norms, rectify};
  const unsigned blocks = (unsigned)std::min<uint64_t>(cdiv(row_count, kScanWaves), 2048);
  DevBuf<double> lo_key;
  DevBuf<uint32_t> lo_idx;
  for (uint32_t off = 0; off < K; off += 64) {
    const uint32_t kk = std::min<uint32_t>(64, K - off);
    const double *lk = nullptr;
    const uint32_t *li = nullptr;
    if (off > 0) {  // continue after entry off - 1 of every row (a row that ran out of neighbours keeps +inf / none)
      if (!lo_key.p) {
        AS_TRY(lo_key.alloc(ctx, row_count));
        AS_TRY(lo_idx.alloc(ctx, row_count));
      }
      AS_HIP(ctx, hipMemcpy2DAsync(lo_key.p, sizeof(double), out_dist + (off - 1), sizeof(double) * K, sizeof(double), row_count,
                                   hipMemcpyDeviceToDevice, ctx->stream));
      AS_HIP(ctx, hipMemcpy2DAsync(lo_idx.p, sizeof(uint32_t), out_idx + (off - 1), sizeof(uint32_t) * K, sizeof(uint32_t),
                                   row_count, hipMemcpyDeviceToDevice, ctx->stream));
      lk = lo_key.p;
      li = lo_idx.p;
    }
    ProfScope ps(ctx, "knn_exact_scan");
    hipLaunchKernelGGL(scan_topk_kernel<GraphDistPolicy>, dim3(blocks), dim3(256), 0, ctx->stream,
                       rows, d, rows, d, (uint32_t)n, (uint32_t)d, row_ids, (uint32_t)row_begin,
                       (uint32_t)row_count, kk, lk, li, pol, out_dist, out_idx, K, off, err.p);
    AS_HIP(ctx, hipGetLastError());
  }
  return AS_OK;
}

// self_in_results (include/arrowspace_hip.h): the row itself competes for one of the K places with its own distance
// 1 - x.x / (|x||x|) in the reference arithmetic; if it lands inside the first K (ties: lower index) the last
// neighbour is pushed out.  One thread per row of the table.
__global__ void self_in_results_kernel(const double *__restrict__ rows, const double *__restrict__ norms, uint64_t d,
                                       uint64_t row_begin, uint64_t row_count, uint32_t K, int rectify,
                                       uint32_t *__restrict__ idx, double *__restrict__ dist) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= row_count) return;
  const uint32_t last = idx[r * K + K - 1];
  if (last == kNoIdx) return;  // fewer than K other rows: nobody is pushed out
  const uint64_t i = row_begin + r;
  double s = 0.0;
  for (uint64_t c = 0; c < d; ++c) {
    const double v = rows[i * d + c];
    s = __dadd_rn(s, __dmul_rn(v, v));
  }
  const double denom = __dmul_rn(norms[i], norms[i]);
  double c = denom > 1e-12 ? __ddiv_rn(s, denom) : 0.0;
  if (rectify) {
    c = fmin(fmax(c, -1.0), 1.0);
    c = fmax(c, 0.0);
  }
  const double ds = __dsub_rn(1.0, c);
  const double dl = dist[r * K + K - 1];
  if (ds < dl || (ds == dl && (uint32_t)i < last)) {
    idx[r * K + K - 1] = kNoIdx;
    dist[r * K + K - 1] = __builtin_inf();
  }
}

int apply_self_in_results(as_ctx *ctx, const double *rows, const double *norms, uint64_t d, uint64_t row_begin,
                          uint64_t row_count, uint32_t K, int rectify, uint32_t *idx, double *dist) {
  if (row_count == 0) return AS_OK;
  hipLaunchKernelGGL(self_in_results_kernel, dim3(cdiv(row_count, 256)), dim3(256), 0, ctx->stream, rows, norms, d, row_begin,
                     row_count, K, rectify, idx, dist);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

}  // namespace asx
