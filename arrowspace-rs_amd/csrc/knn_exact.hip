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

int row_norms(as_ctx *ctx, const double *rows, uint64_t n, uint64_t d, double *norms) {
  if (n == 0) return AS_OK;
  ProfScope ps(ctx, "row_norms");
  hipLaunchKernelGGL(row_norms_kernel, dim3(cdiv(n, kNormRows)), dim3(kNormRows), 0, ctx->stream, rows, n, d, norms);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

int knn_exact(as_ctx *ctx, const double *rows, const double *norms, uint64_t n, uint64_t d,
              const uint32_t *row_ids, uint64_t row_begin, uint64_t row_count, uint32_t K,
              int rectify, uint32_t *out_idx, double *out_dist, uint64_t /*unused*/) {
  if (row_count == 0) return AS_OK;
  if (K == 0 || K > 64) return fail(ctx, AS_ERR_UNSUPPORTED, "topk + 1 must be in 1..64");
  DevBuf<int> err;
  AS_TRY(err.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(err.p, 0, sizeof(int), ctx->stream));
  GraphDistPolicy pol{norms, rectify};
  const unsigned blocks = (unsigned)std::min<uint64_t>(cdiv(row_count, kScanWaves), 2048);
  {
    ProfScope ps(ctx, "knn_exact_scan");
    hipLaunchKernelGGL(scan_topk_kernel<GraphDistPolicy>, dim3(blocks), dim3(256), 0, ctx->stream,
                       rows, d, rows, d, (uint32_t)n, (uint32_t)d, row_ids, (uint32_t)row_begin,
                       (uint32_t)row_count, K, (const double *)nullptr, (const uint32_t *)nullptr, pol,
                       out_dist, out_idx, K, 0u, err.p);
  }
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

}  // namespace asx
