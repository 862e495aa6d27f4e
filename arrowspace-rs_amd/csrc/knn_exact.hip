// knn_exact.hip -- exact row norms and the exact neighbour scan (K1 + exact leg of K2).
// Replaces the CosinePair queries of src/laplacian.rs:211,221,246 with the rule set of
// SURVEY.md Appendix A/B (self excluded, (distance asc, index asc), cos := 0 on tiny norms).
#include "scan_topk.cuh"

namespace asx {

// One thread per row, sequential sum of squares (src/core.rs:212-216).  Rows are short
// (<= a few KB) and this runs once per build, so the strided row walk is served by L2.
__global__ void row_norms_kernel(const double *__restrict__ rows, uint64_t n, uint64_t d,
                                 double *__restrict__ norms) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double *r = rows + i * d;
  double s = 0.0;
  for (uint64_t c = 0; c < d; ++c) s = __dadd_rn(s, __dmul_rn(r[c], r[c]));
  norms[i] = __dsqrt_rn(s);
}

int row_norms(as_ctx *ctx, const double *rows, uint64_t n, uint64_t d, double *norms) {
  if (n == 0) return AS_OK;
  ProfScope ps(ctx, "row_norms");
  hipLaunchKernelGGL(row_norms_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, rows, n, d, norms);
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
