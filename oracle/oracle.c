/*
 * oracle.c -- CPU restatement (plain C, f64) of the arrowspace lambda-tau path.
 * TEST INFRASTRUCTURE ONLY; see oracle.h for the rules and the parity status.
 *
 * Every function cites the reference lines it follows (paths relative to /root/reference).
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp)
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TAU_FLOOR 1e-10 /* src/taumode.rs:84 */

int aso_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* src/builder.rs:225-233 -- k itself is never read again by the graph code */
uint64_t aso_define_result_k(uint64_t k, uint64_t topk) {
  if (k <= 5) return 3;
  if (k < 10) return 4;
  return topk;
}

/* sequential sum of squares then sqrt: src/core.rs:212-216, src/tests/test_helpers.rs:76-79 */
static double row_norm(const double *a, uint64_t d) {
  double s = 0.0;
  for (uint64_t c = 0; c < d; ++c) s += a[c] * a[c];
  return sqrt(s);
}

static double row_dot(const double *a, const double *b, uint64_t d) {
  double s = 0.0;
  for (uint64_t c = 0; c < d; ++c) s += a[c] * b[c];
  return s;
}

/* cosine distance of the graph stage.  Plain 1 - cos; cos := 0 when |a||b| <= 1e-12
 * (in-tree model src/tests/test_helpers.rs:93-105; smartcore CosinePair source absent). */
static double graph_distance(double dot, double ni, double nj, int rectify) {
  double denom = ni * nj;
  double c = denom > 1e-12 ? dot / denom : 0.0;
  if (rectify) {
    if (c > 1.0) c = 1.0;
    if (c < -1.0) c = -1.0;
    if (c < 0.0) c = 0.0;
  }
  return 1.0 - c;
}

/* StandardScaler (src/laplacian.rs:146-151): per-column z-score.  smartcore source absent:
 * population variance, zero std -> 1.  PARITY UNPINNED. */
static double *standardise(const double *rows, uint64_t n, uint64_t d) {
  double *out = (double *)malloc(sizeof(double) * n * d);
  if (!out) return NULL;
  for (uint64_t c = 0; c < d; ++c) {
    double m = 0.0;
    for (uint64_t r = 0; r < n; ++r) m += rows[r * d + c];
    m /= (double)n;
    double v = 0.0;
    for (uint64_t r = 0; r < n; ++r) {
      double t = rows[r * d + c] - m;
      v += t * t;
    }
    double sd = sqrt(v / (double)n);
    if (sd == 0.0) sd = 1.0;
    for (uint64_t r = 0; r < n; ++r) out[r * d + c] = (rows[r * d + c] - m) / sd;
  }
  return out;
}


/* self_in_results: the query row takes part in the ranking with its own distance,
 * 1 - x.x / (|x||x|) in the same arithmetic (not exactly 0), ties by index; if it lands inside the first K it is
 * dropped afterwards (src/laplacian.rs:224,252 filter i != j), which leaves K - 1 neighbours. */
static void apply_self_in_results(const double *R, uint64_t d, const double *norms, uint64_t i, int rectify, uint64_t K,
                                  uint64_t cnt, uint64_t *bi, double *bd) {
  if (cnt < K) return; /* fewer than K other rows exist: the row pushes nobody out */
  const double ds = graph_distance(row_dot(R + i * d, R + i * d, d), norms[i], norms[i], rectify);
  if (ds < bd[K - 1] || (ds == bd[K - 1] && i < bi[K - 1])) {
    bi[K - 1] = UINT64_MAX;
    bd[K - 1] = INFINITY;
  }
}

/* K2: per node the first (topk+1) candidates j != i ordered by (distance asc, index asc).
 * src/laplacian.rs:211,221,246 via CosinePair; ordering of src/tests/test_helpers.rs:115-119 */
static int knn_core(const double *R, uint64_t n, uint64_t d, const aso_graph_params *p,
                    uint64_t *nbr_idx, double *nbr_dist) {
  const uint64_t K = p->topk + 1;
  double *norms = (double *)malloc(sizeof(double) * n);
  if (!norms) return ASO_ERR_ALLOC;
  for (uint64_t i = 0; i < n; ++i) norms[i] = row_norm(R + i * d, d);
  const int rectify = p->rectify;
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t ii = 0; ii < (int64_t)n; ++ii) {
    const uint64_t i = (uint64_t)ii;
    uint64_t *bi = nbr_idx + i * K;
    double *bd = nbr_dist + i * K;
    uint64_t cnt = 0;
    for (uint64_t j = 0; j < n; ++j) {
      if (j == i) continue;
      double dist = graph_distance(row_dot(R + i * d, R + j * d, d), norms[i], norms[j], rectify);
      /* j ascends, so an equal distance never displaces an earlier entry */
      if (cnt == K && !(dist < bd[K - 1])) continue;
      uint64_t pos = cnt < K ? cnt : K - 1;
      while (pos > 0 && dist < bd[pos - 1]) {
        bd[pos] = bd[pos - 1];
        bi[pos] = bi[pos - 1];
        --pos;
      }
      bd[pos] = dist;
      bi[pos] = j;
      if (cnt < K) ++cnt;
    }
    for (uint64_t s = cnt; s < K; ++s) {
      bi[s] = UINT64_MAX;
      bd[s] = INFINITY;
    }
    if (p->self_in_results) apply_self_in_results(R, d, norms, i, rectify, K, cnt, bi, bd);
  }
  free(norms);
  return ASO_OK;
}

/* the same neighbour rule for a list of rows only (sampled parity checks at full size) */
int aso_knn_cosine_rows(const double *R, uint64_t n, uint64_t d, const aso_graph_params *p,
                        const uint64_t *row_ids, uint64_t n_rows, uint64_t *nbr_idx, double *nbr_dist) {
  if (n < 2 || d < 2) return ASO_ERR_SHAPE;
  const uint64_t K = p->topk + 1;
  double *norms = (double *)malloc(sizeof(double) * n);
  if (!norms) return ASO_ERR_ALLOC;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)n; ++i) norms[i] = row_norm(R + (uint64_t)i * d, d);
  const int rectify = p->rectify;
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t rr = 0; rr < (int64_t)n_rows; ++rr) {
    const uint64_t i = row_ids[rr];
    uint64_t *bi = nbr_idx + (uint64_t)rr * K;
    double *bd = nbr_dist + (uint64_t)rr * K;
    uint64_t cnt = 0;
    for (uint64_t j = 0; j < n; ++j) {
      if (j == i) continue;
      double dist = graph_distance(row_dot(R + i * d, R + j * d, d), norms[i], norms[j], rectify);
      if (cnt == K && !(dist < bd[K - 1])) continue;
      uint64_t pos = cnt < K ? cnt : K - 1;
      while (pos > 0 && dist < bd[pos - 1]) {
        bd[pos] = bd[pos - 1];
        bi[pos] = bi[pos - 1];
        --pos;
      }
      bd[pos] = dist;
      bi[pos] = j;
      if (cnt < K) ++cnt;
    }
    for (uint64_t s = cnt; s < K; ++s) {
      bi[s] = UINT64_MAX;
      bd[s] = INFINITY;
    }
    if (p->self_in_results) apply_self_in_results(R, d, norms, i, rectify, K, cnt, bi, bd);
  }
  free(norms);
  return ASO_OK;
}

int aso_knn_cosine(const double *rows, uint64_t n, uint64_t d, const aso_graph_params *p,
                   uint64_t *nbr_idx, double *nbr_dist) {
  if (n < 2 || d < 2) return ASO_ERR_SHAPE;
  return knn_core(rows, n, d, p, nbr_idx, nbr_dist);
}

typedef struct {
  uint64_t j;
  double w;
  double score;
  uint64_t pos;
} cand_t;

static int cand_by_score_desc(const void *a, const void *b) {
  const cand_t *x = (const cand_t *)a, *y = (const cand_t *)b;
  if (x->score > y->score) return -1;
  if (x->score < y->score) return 1;
  /* reference: sort_unstable_by (ties unspecified); oracle: keep candidate order */
  return (x->pos > y->pos) - (x->pos < y->pos);
}

typedef struct {
  uint64_t j;
  double w;
} edge_t;

static int edge_by_col(const void *a, const void *b) {
  const edge_t *x = (const edge_t *)a, *y = (const edge_t *)b;
  return (x->j > y->j) - (x->j < y->j);
}

/* K3..K5 from a complete neighbour table: src/laplacian.rs:217-417 */
int aso_laplacian_from_knn(const uint64_t *nbr_idx, const double *nbr_dist, uint64_t n,
                           const aso_graph_params *p, aso_csr *out) {
  memset(out, 0, sizeof(*out));
  const uint64_t K = p->topk + 1;
  const double sigma = p->has_sigma ? p->sigma : 1.0; /* src/laplacian.rs:254 */
  uint64_t *deg = (uint64_t *)calloc(n, sizeof(uint64_t));
  if (!deg) return ASO_ERR_ALLOC;

  /* degrees: src/laplacian.rs:217-227 */
  uint64_t deg_sum = 0;
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t c = 0;
    for (uint64_t s = 0; s < K; ++s)
      if (nbr_idx[i * K + s] != UINT64_MAX && nbr_dist[i * K + s] <= p->eps) ++c;
    deg[i] = c;
    deg_sum += c;
  }
  const double avg_degree = (double)deg_sum / (double)n; /* :229 */
  const int sparsify = avg_degree > 10.0;                /* :230 */

  /* directed adjacency rows: src/laplacian.rs:243-288 */
  cand_t *adj = (cand_t *)malloc(sizeof(cand_t) * n * K);
  uint64_t *adj_len = (uint64_t *)calloc(n, sizeof(uint64_t));
  uint64_t *row_cnt = (uint64_t *)calloc(n + 1, sizeof(uint64_t));
  if (!adj || !adj_len || !row_cnt) return ASO_ERR_ALLOC;
  for (uint64_t i = 0; i < n; ++i) {
    cand_t *v = adj + i * K;
    uint64_t len = 0;
    for (uint64_t s = 0; s < K; ++s) {
      uint64_t j = nbr_idx[i * K + s];
      double dist = nbr_dist[i * K + s];
      if (j == UINT64_MAX || !(dist <= p->eps)) continue;
      double w = 1.0 / (1.0 + pow(dist / sigma, p->p)); /* :253-254 */
      if (w > 1e-12) {
        v[len].j = j;
        v[len].w = w;
        v[len].score = sparsify ? w * sqrt((double)(deg[i] * deg[j])) : w; /* :258-262 */
        v[len].pos = len;
        ++len;
      }
    }
    if (sparsify && len > 2) { /* :274-280 */
      qsort(v, len, sizeof(cand_t), cand_by_score_desc);
      uint64_t keep = len / 2;
      if (keep < 1) keep = 1;
      len = keep;
    }
    adj_len[i] = len;
  }

  /* union symmetrisation: src/laplacian.rs:295-346.  (i,j) and (j,i) both get w; when both
   * directions exist they carry the same w (w depends only on the pair distance). */
  for (uint64_t i = 0; i < n; ++i)
    for (uint64_t s = 0; s < adj_len[i]; ++s) {
      row_cnt[i + 1]++;
      row_cnt[adj[i * K + s].j + 1]++;
    }
  for (uint64_t i = 0; i < n; ++i) row_cnt[i + 1] += row_cnt[i];
  const uint64_t tot = row_cnt[n];
  edge_t *ed = (edge_t *)malloc(sizeof(edge_t) * (tot ? tot : 1));
  uint64_t *cur = (uint64_t *)malloc(sizeof(uint64_t) * (n + 1));
  if (!ed || !cur) return ASO_ERR_ALLOC;
  memcpy(cur, row_cnt, sizeof(uint64_t) * (n + 1));
  for (uint64_t i = 0; i < n; ++i)
    for (uint64_t s = 0; s < adj_len[i]; ++s) {
      uint64_t j = adj[i * K + s].j;
      double w = adj[i * K + s].w;
      ed[cur[i]].j = j;
      ed[cur[i]++].w = w;
      ed[cur[j]].j = i;
      ed[cur[j]++].w = w;
    }
  /* per-row sort by column + dedupe: :339-341 */
  uint64_t *sym_len = (uint64_t *)calloc(n, sizeof(uint64_t));
  if (!sym_len) return ASO_ERR_ALLOC;
  uint64_t nnz = 0;
  for (uint64_t i = 0; i < n; ++i) {
    edge_t *r = ed + row_cnt[i];
    uint64_t len = row_cnt[i + 1] - row_cnt[i];
    qsort(r, len, sizeof(edge_t), edge_by_col);
    uint64_t o = 0;
    for (uint64_t s = 0; s < len; ++s) {
      if (r[s].j == i) continue; /* src != dst, :331 */
      if (o > 0 && r[o - 1].j == r[s].j) continue;
      r[o++] = r[s];
    }
    sym_len[i] = o;
    nnz += o + 1; /* explicit diagonal, :369-370 */
  }

  /* Laplacian triplets -> CSR: src/laplacian.rs:349-417, :160 */
  out->n = n;
  out->nnz = nnz;
  out->indptr = (uint64_t *)malloc(sizeof(uint64_t) * (n + 1));
  out->indices = (uint64_t *)malloc(sizeof(uint64_t) * nnz);
  out->data = (double *)malloc(sizeof(double) * nnz);
  if (!out->indptr || !out->indices || !out->data) return ASO_ERR_ALLOC;
  uint64_t o = 0;
  for (uint64_t i = 0; i < n; ++i) {
    out->indptr[i] = o;
    const edge_t *r = ed + row_cnt[i];
    double degree = 0.0; /* :369 sequential sum in column order */
    for (uint64_t s = 0; s < sym_len[i]; ++s) degree += r[s].w;
    int diag_done = 0;
    for (uint64_t s = 0; s < sym_len[i]; ++s) {
      if (!diag_done && r[s].j > i) {
        out->indices[o] = i;
        out->data[o++] = degree;
        diag_done = 1;
      }
      out->indices[o] = r[s].j;
      out->data[o++] = -r[s].w;
    }
    if (!diag_done) {
      out->indices[o] = i;
      out->data[o++] = degree;
    }
  }
  out->indptr[n] = o;

  free(deg);
  free(adj);
  free(adj_len);
  free(row_cnt);
  free(ed);
  free(cur);
  free(sym_len);
  return ASO_OK;
}


int aso_build_laplacian(const double *rows, uint64_t n, uint64_t d, const aso_graph_params *p,
                        aso_csr *out) {
  memset(out, 0, sizeof(*out));
  /* src/laplacian.rs:128-134 */
  if (n < 2 || d < 2) return ASO_ERR_SHAPE;
  const uint64_t K = p->topk + 1;
  double *scaled = NULL;
  const double *R = rows;
  if (p->normalise) {
    scaled = standardise(rows, n, d);
    if (!scaled) return ASO_ERR_ALLOC;
    R = scaled;
  }
  uint64_t *nbr_idx = (uint64_t *)malloc(sizeof(uint64_t) * n * K);
  double *nbr_dist = (double *)malloc(sizeof(double) * n * K);
  if (!nbr_idx || !nbr_dist) return ASO_ERR_ALLOC;
  int rc = knn_core(R, n, d, p, nbr_idx, nbr_dist);
  free(scaled);
  if (!rc) rc = aso_laplacian_from_knn(nbr_idx, nbr_dist, n, p, out);
  free(nbr_idx);
  free(nbr_dist);
  return rc;
}

void aso_csr_free(aso_csr *m) {
  if (!m) return;
  free(m->indptr);
  free(m->indices);
  free(m->data);
  memset(m, 0, sizeof(*m));
}

static int dbl_cmp(const void *a, const void *b) {
  double x = *(const double *)a, y = *(const double *)b;
  return (x > y) - (x < y);
}

/* src/taumode.rs:87-127 */
double aso_select_tau(const double *v, uint64_t len, int mode, double arg) {
  if (mode == ASO_TAU_FIXED) return (isfinite(arg) && arg > 0.0) ? arg : TAU_FLOOR;
  if (mode == ASO_TAU_MEAN) {
    double sum = 0.0;
    uint64_t cnt = 0;
    for (uint64_t i = 0; i < len; ++i)
      if (isfinite(v[i])) {
        sum += v[i];
        ++cnt;
      }
    double m = cnt > 0 ? sum / (double)cnt : 0.0;
    return m > TAU_FLOOR ? m : TAU_FLOOR;
  }
  double *s = (double *)malloc(sizeof(double) * (len ? len : 1));
  uint64_t m = 0;
  for (uint64_t i = 0; i < len; ++i)
    if (isfinite(v[i])) s[m++] = v[i];
  if (m == 0) {
    free(s);
    return TAU_FLOOR;
  }
  qsort(s, m, sizeof(double), dbl_cmp);
  double r;
  if (mode == ASO_TAU_PERCENTILE) {
    double pp = arg < 0.0 ? 0.0 : (arg > 1.0 ? 1.0 : arg); /* NaN: Rust clamp keeps NaN -> idx 0 */
    if (isnan(arg)) pp = 0.0;
    uint64_t idx = (uint64_t)round((double)(m - 1) * pp);
    r = s[idx];
  } else if (m % 2 == 1) {
    r = s[m / 2];
  } else {
    r = 0.5 * (s[m / 2 - 1] + s[m / 2]);
  }
  free(s);
  return r > TAU_FLOOR ? r : TAU_FLOOR;
}

/* src/taumode.rs:552-660.  Row-local partial sums, rows combined in row order (the
 * reference's rayon reduce order is unspecified). */
int aso_synthetic_lambda(const double *x, uint64_t len, const aso_csr *L, double tau,
                         double *lambda_out) {
  if (L->n > len) return ASO_ERR_DIM;
  double num = 0.0, edge = 0.0;
  for (uint64_t i = 0; i < L->n; ++i) {
    double xi = x[i], ln = 0.0, le = 0.0;
    for (uint64_t e = L->indptr[i]; e < L->indptr[i + 1]; ++e) {
      uint64_t j = L->indices[e];
      double lij = L->data[e];
      ln += xi * lij * x[j]; /* :575 */
      if (i != j) {
        double w = -lij > 0.0 ? -lij : 0.0; /* :578 */
        if (w > 0.0) {
          double dd = xi - x[j];
          le += w * dd * dd; /* :581 */
        }
      }
    }
    num += ln;
    edge += le;
  }
  double den = 0.0;
  for (uint64_t i = 0; i < len; ++i) den += x[i] * x[i]; /* :596 over ALL entries */
  double e_raw = den > 1e-12 ? num / den : 0.0;          /* :597-602 */
  double g = 0.0;
  if (edge > 0.0) { /* :611-639 */
    for (uint64_t i = 0; i < L->n; ++i) {
      double xi = x[i], lg = 0.0;
      for (uint64_t e = L->indptr[i]; e < L->indptr[i + 1]; ++e) {
        uint64_t j = L->indices[e];
        if (i == j) continue;
        double lij = L->data[e];
        double w = -lij > 0.0 ? -lij : 0.0;
        if (w > 0.0) {
          double dd = xi - x[j];
          double share = (w * dd * dd) / edge;
          lg += share * share;
        }
      }
      g += lg;
    }
  }
  double g_raw = g < 0.0 ? 0.0 : (g > 1.0 ? 1.0 : g); /* :641 */
  double e_bounded = e_raw / (e_raw + tau);           /* :642 */
  *lambda_out = tau * e_bounded + (1.0 - tau) * g_raw; /* :647 */
  return ASO_OK;
}

/* src/taumode.rs:230-259: tau from the item's own values, then the CSR form */
int aso_taumode_lambdas(const double *items, uint64_t n_items, uint64_t n_features,
                        const aso_csr *L, int mode, double arg, double *lambdas_out) {
  if (L->n > n_features) return ASO_ERR_DIM;
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < (int64_t)n_items; ++r) {
    const double *x = items + (uint64_t)r * n_features;
    double tau = aso_select_tau(x, n_features, mode, arg);
    aso_synthetic_lambda(x, n_features, L, tau, &lambdas_out[r]);
  }
  return ASO_OK;
}

/* src/core.rs:533-549 (projection off) */
int aso_prepare_query_item(const double *q, uint64_t n_features, const aso_csr *L, int mode,
                           double arg, double *lambda_out) {
  for (uint64_t i = 0; i < n_features; ++i)
    if (!isfinite(q[i])) return ASO_ERR_NONFINITE;
  double tau = aso_select_tau(q, n_features, mode, arg);
  return aso_synthetic_lambda(q, n_features, L, tau, lambda_out);
}

/* src/core.rs:156-175 with :229-239 and :135-138 */
double aso_lambda_similarity(const double *q, double lambda_q, const double *x, double lambda_x,
                             uint64_t d, double alpha) {
  double denom = row_norm(q, d) * row_norm(x, d);
  double cosv = denom > 0.0 ? row_dot(q, x, d) / denom : 0.0;
  double diff = fabs(lambda_q - lambda_x);
  double lam = 1.0 - (diff < 1.0 ? diff : 1.0);
  return alpha * cosv + (1.0 - alpha) * lam;
}

/* src/core.rs:760-798: full scan, stable descending sort, truncate(k) */
int64_t aso_search_lambda_aware(const double *items, const double *lambdas, uint64_t n_items,
                                uint64_t d, const double *q, double lambda_q, uint64_t k,
                                double alpha, uint64_t *idx_out, double *score_out) {
  if (lambda_q == 0.0) return ASO_ERR_ZERO_LAMBDA;
  uint64_t kk = k < n_items ? k : n_items;
  uint64_t cnt = 0;
  for (uint64_t i = 0; i < n_items; ++i) {
    double s = aso_lambda_similarity(q, lambda_q, items + i * d, lambdas[i], d, alpha);
    if (isnan(s)) return ASO_ERR_NAN_SCORE;
    if (kk == 0) continue;
    /* (score desc, index asc): i ascends, so an equal score never displaces */
    if (cnt == kk && !(s > score_out[kk - 1])) continue;
    uint64_t pos = cnt < kk ? cnt : kk - 1;
    while (pos > 0 && s > score_out[pos - 1]) {
      score_out[pos] = score_out[pos - 1];
      idx_out[pos] = idx_out[pos - 1];
      --pos;
    }
    score_out[pos] = s;
    idx_out[pos] = i;
    if (cnt < kk) ++cnt;
  }
  return (int64_t)kk;
}

/* src/core.rs:802-928 search_lambda_aware_hybrid.  Union of (a) every item with cosine > 0.9999
 * (score = cosine), (b) the lambda-aware top-k (score = blended, unless already in (a)), (c) the
 * best-cosine item (score = cosine unless already present); sorted by score descending
 * (reference: sort_unstable_by, ties unspecified; oracle: index ascending), truncated to k.
 * Returns the number of results written. */
typedef struct {
  uint64_t idx;
  double score;
} hyb_t;
static int hyb_cmp(const void *a, const void *b) {
  const hyb_t *x = (const hyb_t *)a, *y = (const hyb_t *)b;
  if (x->score > y->score) return -1;
  if (x->score < y->score) return 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}
int64_t aso_search_lambda_aware_hybrid(const double *items, const double *lambdas, uint64_t n_items,
                                       uint64_t d, const double *q, double lambda_q, uint64_t k,
                                       double alpha, uint64_t *idx_out, double *score_out) {
  if (k == 0 || n_items == 0) return 0;
  const double beta = 1.0 - alpha;
  const double qn = row_norm(q, d);
  hyb_t *all = (hyb_t *)malloc(sizeof(hyb_t) * n_items);   /* blended score of everything */
  double *cosv = (double *)malloc(sizeof(double) * n_items);
  if (!all || !cosv) return ASO_ERR_ALLOC;
  uint64_t sem_top = 0;
  double sem_best = -INFINITY;
  for (uint64_t i = 0; i < n_items; ++i) {
    const double denom = qn * row_norm(items + i * d, d);
    const double c = denom > 0.0 ? row_dot(q, items + i * d, d) / denom : 0.0; /* :229-239 */
    const double diff = fabs(lambda_q - lambdas[i]);
    const double lam = 1.0 - (diff < 1.0 ? diff : 1.0);
    cosv[i] = c;
    all[i].idx = i;
    all[i].score = alpha * c + beta * lam; /* :830-832 */
    if (c > sem_best) {                     /* :835-837 strict: first index of the maximum */
      sem_best = c;
      sem_top = i;
    }
  }
  /* lambda top-k by (score desc, idx asc) */
  qsort(all, n_items, sizeof(hyb_t), hyb_cmp);
  const uint64_t kk = k < n_items ? k : n_items;
  /* union */
  hyb_t *u = (hyb_t *)malloc(sizeof(hyb_t) * (n_items + kk + 1));
  char *in = (char *)calloc(n_items, 1);
  if (!u || !in) return ASO_ERR_ALLOC;
  uint64_t m = 0;
  for (uint64_t i = 0; i < n_items; ++i)
    if (cosv[i] > 0.9999) { /* :840-842 */
      u[m].idx = i;
      u[m++].score = cosv[i];
      in[i] = 1;
    }
  for (uint64_t t = 0; t < kk; ++t)
    if (!in[all[t].idx]) {
      u[m++] = all[t];
      in[all[t].idx] = 1;
    }
  if (!in[sem_top]) {
    u[m].idx = sem_top;
    u[m++].score = sem_best;
  }
  qsort(u, m, sizeof(hyb_t), hyb_cmp);
  const uint64_t r = m < k ? m : k;
  for (uint64_t t = 0; t < r; ++t) {
    idx_out[t] = u[t].idx;
    score_out[t] = u[t].score;
  }
  free(all);
  free(cosv);
  free(u);
  free(in);
  return (int64_t)r;
}

/* src/core.rs:944-976 range_search (query lambda already prepared): every item with
 * lambda_q - lambda_i <= eps, in index order, with that signed difference. */
int64_t aso_range_search(const double *lambdas, uint64_t n_items, double lambda_q, double eps,
                         uint64_t *idx_out, double *dist_out) {
  uint64_t m = 0;
  for (uint64_t i = 0; i < n_items; ++i) {
    const double dist = lambda_q - lambdas[i];
    if (dist <= eps) {
      idx_out[m] = i;
      dist_out[m++] = dist;
    }
  }
  return (int64_t)m;
}

/* src/clustering.rs:913-928 nearest_centroid, applied to a batch of rows (its callers, :590, :721, :764, walk the rows):
 * squared Euclidean distance as a sequential sum of (a - b)^2, strict '<' (the first of equal minima wins),
 * (0, +inf) when no distance compares below +inf (NaN rows). */
int aso_nearest_centroids(const double *rows, uint64_t n, uint64_t d, const double *centroids, uint64_t n_centroids,
                          uint64_t *idx_out, double *dist2_out) {
  if (!rows || !centroids || !idx_out || !dist2_out) return -1;
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < (int64_t)n; ++r) {
    const double *x = rows + (uint64_t)r * d;
    uint64_t best = 0;
    double best_d2 = INFINITY;
    for (uint64_t c = 0; c < n_centroids; ++c) {
      const double *y = centroids + c * d;
      double d2 = 0.0;
      for (uint64_t k = 0; k < d; ++k) {
        const double diff = x[k] - y[k];
        d2 += diff * diff;
      }
      if (d2 < best_d2) {
        best_d2 = d2;
        best = c;
      }
    }
    idx_out[r] = best;
    dist2_out[r] = best_d2;
  }
  return 0;
}
