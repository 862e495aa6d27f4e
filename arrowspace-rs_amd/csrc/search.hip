// search.hip -- exact leg of K9: blended cosine + lambda scan with per-query top-k.
// Replaces ArrowSpace::search_lambda_aware (src/core.rs:760-798) and
// ArrowItem::lambda_similarity (src/core.rs:135-239), batched over queries.
#include "scan_topk.h"

namespace asx {

// Exact scan for the listed queries (q_ids == nullptr: all nq).  Results are written in list
// order (row r of out = r-th listed query).  k > 64 is served by repeated passes, each
// continuing strictly after the last (key, index) of the previous one.
int search_exact(as_ctx *ctx, const as_index *ix, const double *queries, const double *q_norms,
                 const double *q_lambdas, const uint32_t *q_ids, uint64_t nq_sel, uint64_t /*nq*/, uint32_t k,
                 double alpha, uint32_t *out_idx, double *out_key) {
  if (nq_sel == 0 || k == 0) return AS_OK;
  DevBuf<int> err;
  AS_TRY(err.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(err.p, 0, sizeof(int), ctx->stream));
  SearchScorePolicy pol{q_norms, q_lambdas, ix->norms, ix->lambdas, alpha};
  const unsigned blocks = (unsigned)std::min<uint64_t>(cdiv(nq_sel, kScanWaves), 2048);
  DevBuf<double> lo_key;
  DevBuf<uint32_t> lo_idx;
  for (uint32_t off = 0; off < k; off += 64) {
    const uint32_t kk = std::min<uint32_t>(64, k - off);
    const double *lk = nullptr;
    const uint32_t *li = nullptr;
    if (off > 0) {  // continue after entry off-1 of every query
      if (!lo_key.p) {
        AS_TRY(lo_key.alloc(ctx, nq_sel));
        AS_TRY(lo_idx.alloc(ctx, nq_sel));
      }
      AS_HIP(ctx, hipMemcpy2DAsync(lo_key.p, sizeof(double), out_key + (off - 1), sizeof(double) * k, sizeof(double),
                                   nq_sel, hipMemcpyDeviceToDevice, ctx->stream));
      AS_HIP(ctx, hipMemcpy2DAsync(lo_idx.p, sizeof(uint32_t), out_idx + (off - 1), sizeof(uint32_t) * k,
                                   sizeof(uint32_t), nq_sel, hipMemcpyDeviceToDevice, ctx->stream));
      lk = lo_key.p;
      li = lo_idx.p;
    }
    ProfScope ps(ctx, "search_exact_scan");
    hipLaunchKernelGGL(scan_topk_kernel<SearchScorePolicy>, dim3(blocks), dim3(256), 0, ctx->stream, queries, ix->d,
                       ix->items, ix->d, (uint32_t)ix->n, (uint32_t)ix->d, q_ids, 0u, (uint32_t)nq_sel, kk, lk, li,
                       pol, out_key, out_idx, k, off, err.p);
    AS_HIP(ctx, hipGetLastError());
  }
  int h = 0;
  AS_HIP(ctx, copy_d2h(ctx, &h, err.p, sizeof(int)));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (h) return fail(ctx, AS_ERR_NAN_SCORE, as_status_message(AS_ERR_NAN_SCORE));
  return AS_OK;
}

}  // namespace asx
