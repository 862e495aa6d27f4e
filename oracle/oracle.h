/*
 * oracle.h -- CPU restatement of arrowspace's lambda-tau build + lambda-aware search path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
 * only as the checker / the reported CPU baseline -- never as the thing measured or shipped.
 * The product (arrowspace-rs_amd/) has no CPU path and never links or imports this.
 *
 * PARITY STATUS.  The reference (tuned-org-uk/arrowspace-rs v0.18.1) is pure Rust; there is no
 * Rust toolchain in the build image, so no reference binary (oracle/_ref) can exist.  This
 * restatement is pinned against every known-answer test the reference holds for the path
 * (select_tau table, lambda scale invariance, ArrowItem doctests, the 64x24 top-3 anchor, the
 * Laplacian invariants on the literal matrices of src/tests/test_laplacian.rs) -- see
 * tests/test_oracle_golden.py.  The nearest-neighbour arithmetic itself lives in smartcore
 * 0.4.5 `CosinePair` (Cargo.lock:1719-1722), whose source is absent from /root/reference:
 * for exact neighbour sets, tie order, self-exclusion and zero-norm rows the rule set of
 * SURVEY.md Appendix A/B is used and that part is "PARITY UNPINNED".
 *
 * All arithmetic is IEEE f64, sequential-by-index, compiled with -ffp-contract=off so that
 * a*b+c is two roundings exactly as rustc emits it.
 */
#ifndef ARROWSPACE_ORACLE_H
#define ARROWSPACE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mirrors GraphParams, reference src/graph.rs:94-102 */
typedef struct {
  double eps;
  uint64_t k;    /* stored, never used by the graph code (src/laplacian.rs uses topk only) */
  uint64_t topk;
  double p;
  double sigma;  /* used when has_sigma != 0, else 1.0 (src/laplacian.rs:254) */
  int32_t has_sigma;
  int32_t normalise;      /* StandardScaler z-score (src/laplacian.rs:146-151); parity unpinned */
  int32_t sparsity_check; /* src/graph.rs:185-193, enforced by the factory, not here */
  int32_t rectify;        /* 0: dist = 1 - cos (default); 1: dist = 1 - max(0, cos) */
  int32_t self_in_results; /* 0 (default): query_row_top_k does not return the query row, up to
                            * topk + 1 neighbours; 1: it does (the row itself competes for one of the topk + 1 places
                            * with its own distance and is then dropped by the caller's i != j: src/laplacian.rs:224,252) */
  int32_t reserved;
} aso_graph_params;

typedef struct {
  uint64_t n;
  uint64_t nnz;
  uint64_t *indptr;  /* n + 1 */
  uint64_t *indices; /* nnz, ascending inside a row, diagonal in place */
  double *data;      /* nnz */
} aso_csr;

enum { ASO_TAU_FIXED = 0, ASO_TAU_MEDIAN = 1, ASO_TAU_MEAN = 2, ASO_TAU_PERCENTILE = 3 };

enum {
  ASO_OK = 0,
  ASO_ERR_SHAPE = -1,       /* "items should be at least of shape (2,2)" laplacian.rs:129-134 */
  ASO_ERR_NONFINITE = -2,   /* core.rs:534-537 */
  ASO_ERR_ZERO_LAMBDA = -3, /* core.rs:773-776 */
  ASO_ERR_NAN_SCORE = -4,   /* partial_cmp().unwrap() on NaN, core.rs:785 */
  ASO_ERR_DIM = -5,         /* graph larger than the signal, taumode.rs:570-575 would index OOB */
  ASO_ERR_ALLOC = -6
};

/* builder.rs:225-233 */
uint64_t aso_define_result_k(uint64_t k, uint64_t topk);

/* laplacian.rs:203-292 restricted to the neighbour search: per node the first (topk+1)
 * non-self candidates by (distance asc, index asc).  nbr_idx/nbr_dist are n*(topk+1);
 * rows with fewer than topk+1 candidates are padded with idx = UINT64_MAX. */
int aso_knn_cosine(const double *rows, uint64_t n, uint64_t d, const aso_graph_params *p,
                   uint64_t *nbr_idx, double *nbr_dist);

/* the same rule for a list of rows (row_ids[n_rows]); outputs are n_rows x (topk+1) */
int aso_knn_cosine_rows(const double *rows, uint64_t n, uint64_t d, const aso_graph_params *p,
                        const uint64_t *row_ids, uint64_t n_rows, uint64_t *nbr_idx, double *nbr_dist);

/* laplacian.rs:122-417: rows are nodes (row-major n x d) -> n x n CSR Laplacian */
int aso_build_laplacian(const double *rows, uint64_t n, uint64_t d, const aso_graph_params *p,
                        aso_csr *out);
/* laplacian.rs:217-417 from a complete neighbour table (n x (topk+1), as aso_knn_cosine writes) */
int aso_laplacian_from_knn(const uint64_t *nbr_idx, const double *nbr_dist, uint64_t n,
                           const aso_graph_params *p, aso_csr *out);
void aso_csr_free(aso_csr *m);

/* taumode.rs:87-127 */
double aso_select_tau(const double *v, uint64_t len, int mode, double arg);

/* taumode.rs:552-660 (== :381-519) */
int aso_synthetic_lambda(const double *x, uint64_t len, const aso_csr *L, double tau,
                         double *lambda_out);

/* taumode.rs:230-259 */
int aso_taumode_lambdas(const double *items, uint64_t n_items, uint64_t n_features,
                        const aso_csr *L, int mode, double arg, double *lambdas_out);

/* core.rs:533-549 (without projection) */
int aso_prepare_query_item(const double *q, uint64_t n_features, const aso_csr *L, int mode,
                           double arg, double *lambda_out);

/* core.rs:135-239 */
double aso_lambda_similarity(const double *q, double lambda_q, const double *x, double lambda_x,
                             uint64_t d, double alpha);

/* core.rs:760-798; writes min(k, n_items) results, returns that count or a negative error */
int64_t aso_search_lambda_aware(const double *items, const double *lambdas, uint64_t n_items,
                                uint64_t d, const double *q, double lambda_q, uint64_t k,
                                double alpha, uint64_t *idx_out, double *score_out);

/* core.rs:802-928; up to k results (the union can be smaller than k only when n_items < k) */
int64_t aso_search_lambda_aware_hybrid(const double *items, const double *lambdas, uint64_t n_items,
                                       uint64_t d, const double *q, double lambda_q, uint64_t k,
                                       double alpha, uint64_t *idx_out, double *score_out);

/* core.rs:944-976 with a prepared query lambda; outputs must hold n_items entries */
int aso_nearest_centroids(const double *rows, uint64_t n, uint64_t d, const double *centroids, uint64_t n_centroids,
                          uint64_t *idx_out, double *dist2_out);
int64_t aso_range_search(const double *lambdas, uint64_t n_items, double lambda_q, double eps,
                         uint64_t *idx_out, double *dist_out);

/* ---- clustering pre-stage and JL projection (prestage.c; see its header for what is pinned) ------------------------- */
int aso_pre_sample_indices(uint64_t n, uint64_t m, uint64_t seed, uint64_t *out);
/* src/clustering.rs:117-147; NaN where the reference skips the row */
int aso_two_nn_ratios(const double *rows, uint64_t n, uint64_t f, const uint64_t *sample_ids, uint64_t ns, double *ratios);
uint64_t aso_intrinsic_dimension(uint64_t n, uint64_t f, const double *ratios, uint64_t ns);
int aso_kmeans_lloyd(const double *rows, uint64_t n, uint64_t f, uint64_t k, uint64_t max_iter, uint64_t seed,
                     uint64_t *assign, double *cents);
double aso_calinski_harabasz(const double *rows, uint64_t n, uint64_t f, const uint64_t *assign, uint64_t k);
uint64_t aso_step2_calinski_harabasz(const double *rows, uint64_t n, uint64_t f, uint64_t k_min, uint64_t k_max, uint64_t seed);
double aso_threshold_from_pilot(const double *rows, uint64_t n, uint64_t f, uint64_t k, uint64_t seed);
int aso_compute_optimal_k(const double *rows, uint64_t n, uint64_t f, uint64_t seed, uint64_t *k_out, double *radius_out,
                          uint64_t *id_out);
int64_t aso_incremental_clustering(const double *rows, uint64_t n, uint64_t f, uint64_t max_clusters, double radius,
                                   uint64_t batch, double *cents, int64_t *assign, uint64_t *counts);
uint64_t aso_jl_dimension(uint64_t n_points, double epsilon);
int aso_jl_project(const double *rows, uint64_t n, uint64_t d, const double *G, uint64_t r, double *out);

/* threads used by the OpenMP loops (for the cpu_baseline report) */
int aso_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
