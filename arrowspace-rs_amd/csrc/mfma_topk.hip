// mfma_topk.hip -- MFMA candidate pass (placeholder wiring: forwards to the exact scan until the
// f64 MFMA kernel lands in the next commit).
#include "common.h"

namespace asx {

int knn_mfma(as_ctx *ctx, const double *rows, const double *norms, uint64_t n, uint64_t d, uint64_t row_begin,
             uint64_t row_count, uint32_t K, int rectify, uint32_t *out_idx, double *out_dist) {
  return knn_exact(ctx, rows, norms, n, d, nullptr, row_begin, row_count, K, rectify, out_idx, out_dist, 0);
}

int search_mfma(as_ctx *ctx, as_index *ix, const double *queries, const double *q_norms, const double *q_lambdas,
                uint64_t nq, uint32_t k, double alpha, uint32_t *out_idx, double *out_score) {
  return search_exact(ctx, ix, queries, q_norms, q_lambdas, nullptr, nq, nq, k, alpha, out_idx, out_score);
}

}  // namespace asx
