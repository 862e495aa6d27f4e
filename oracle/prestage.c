/*
 * prestage.c -- CPU restatement of the clustering pre-stage (src/clustering.rs) and of the JL projection
 * (src/reduction.rs) of the reference.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * What is pinned and what is not.  The arithmetic of every step follows the reference line by line (cited at each
 * function).  Three things in the reference cannot be reproduced without its Rust dependencies and are replaced by
 * the build's own seeded definitions, the SAME here and in the library (so the two agree bit for bit, while
 * agreement with a reference run is statistical only: SURVEY.md section 8f rank 1, src/tests/test_taumode.rs:328-428):
 *   - every `StdRng::seed_from_u64(..)` + `shuffle` draw (sample of 500 / 1000 rows)      -> pre_sample_indices()
 *   - smartcore `KMeans::fit` (initialisation and stopping rule unknown: source absent)   -> pre_kmeans_lloyd():
 *     seeded choice of k distinct rows, Lloyd steps, nearest_centroid tie rule of src/clustering.rs:913-928
 *   - the ChaCha8 Gaussian stream of ImplicitProjection (src/reduction.rs:186-199)       -> the matrix is an input
 * The incremental clustering is restated for a batch size B: the snapshot of src/clustering.rs:571-594 is taken once
 * per batch of B rows (a legal schedule of the reference's parallel mode) and rows also see the centroids created
 * earlier in their own batch; a batch ENDS EARLY with the row that creates the last centroid the cap allows (from then on
 * no centroid can appear, a row's fate depends on the snapshot alone, and the next batch starts with a snapshot of them
 * all); B = 1 is the reference's sequential mode (`with_seed`, :900-901) exactly.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static uint64_t sm64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

/* m distinct indices of [0, n): the first m steps of a Fisher-Yates shuffle of 0..n-1 driven by SplitMix64 */
int aso_pre_sample_indices(uint64_t n, uint64_t m, uint64_t seed, uint64_t *out) {
  uint64_t *idx = (uint64_t *)malloc(sizeof(uint64_t) * (n ? n : 1));
  if (!idx) return ASO_ERR_ALLOC;
  for (uint64_t i = 0; i < n; ++i) idx[i] = i;
  if (m > n) m = n;
  for (uint64_t t = 0; t < m; ++t) {
    const uint64_t j = t + sm64(sm64(seed) ^ sm64(t + 0x51ED270B27B4F3CFull)) % (n - t);
    const uint64_t tmp = idx[t];
    idx[t] = idx[j];
    idx[j] = tmp;
    out[t] = idx[t];
  }
  free(idx);
  return ASO_OK;
}

/* squared Euclidean distance, sequential over the features: src/clustering.rs:129-134,918-922 */
static double dist2(const double *a, const double *b, uint64_t f) {
  double d2 = 0.0;
  for (uint64_t j = 0; j < f; ++j) {
    const double diff = a[j] - b[j];
    d2 += diff * diff;
  }
  return d2;
}

/* Two-NN ratios d2 / d1 of the sampled rows (src/clustering.rs:117-147); NaN where the reference returns None */
int aso_two_nn_ratios(const double *rows, uint64_t n, uint64_t f, const uint64_t *sample_ids, uint64_t ns, double *ratios) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int64_t ss = 0; ss < (int64_t)ns; ++ss) {
    const uint64_t i = sample_ids[ss];
    double b1 = INFINITY, b2 = INFINITY;
    uint64_t cnt = 0;
    for (uint64_t j = 0; j < n; ++j) {
      if (j == i) continue;
      const double d = sqrt(dist2(rows + i * f, rows + j * f, f));
      ++cnt;
      if (d < b1) {
        b2 = b1;
        b1 = d;
      } else if (d < b2) {
        b2 = d;
      }
    }
    ratios[ss] = (cnt >= 2 && b1 > 1e-12) ? b2 / b1 : NAN;
  }
  return ASO_OK;
}

/* src/clustering.rs:101-164 given the ratios */
uint64_t aso_intrinsic_dimension(uint64_t n, uint64_t f, const double *ratios, uint64_t ns) {
  if (n < 10) return f < 2 ? f : 2;
  double sum = 0.0;
  uint64_t cnt = 0;
  for (uint64_t s = 0; s < ns; ++s)
    if (ratios[s] == ratios[s]) {
      sum += ratios[s];
      ++cnt;
    }
  if (cnt == 0) return f < 3 ? f : 3;
  const double mean = sum / (double)cnt;
  const double id = mean > 1.001 ? 1.0 / log(mean) : (double)f;
  double r = round(id);
  if (r < 1.0) r = 1.0;
  if (r > (double)f) r = (double)f;
  return (uint64_t)r;
}

/* src/clustering.rs:913-928: first of equal minima */
static uint64_t nearest(const double *row, const double *cents, uint64_t k, uint64_t f, double *d2_out) {
  uint64_t best = 0;
  double bd = INFINITY;
  for (uint64_t c = 0; c < k; ++c) {
    const double d = dist2(row, cents + c * f, f);
    if (d < bd) {
      bd = d;
      best = c;
    }
  }
  *d2_out = bd;
  return best;
}

/* the build's own Lloyd k-means in place of smartcore's (see the header): assignments[n], centroids[k x f] */
int aso_kmeans_lloyd(const double *rows, uint64_t n, uint64_t f, uint64_t k, uint64_t max_iter, uint64_t seed,
                     uint64_t *assign, double *cents) {
  if (n == 0) return ASO_OK;
  if (k > n) k = n;
  uint64_t *init = (uint64_t *)malloc(sizeof(uint64_t) * k);
  uint64_t *prev = (uint64_t *)malloc(sizeof(uint64_t) * n);
  uint64_t *count = (uint64_t *)malloc(sizeof(uint64_t) * k);
  double *sum = (double *)malloc(sizeof(double) * k * f);
  if (!init || !prev || !count || !sum) return ASO_ERR_ALLOC;
  aso_pre_sample_indices(n, k, seed, init);
  for (uint64_t c = 0; c < k; ++c) memcpy(cents + c * f, rows + init[c] * f, sizeof(double) * f);
  for (uint64_t it = 0;; ++it) {
    int same = it > 0;
    for (uint64_t i = 0; i < n; ++i) {
      double d;
      const uint64_t a = nearest(rows + i * f, cents, k, f, &d);
      if (it > 0 && prev[i] != a) same = 0;
      assign[i] = a;
    }
    if (same || it == max_iter) break;
    memcpy(prev, assign, sizeof(uint64_t) * n);
    memset(sum, 0, sizeof(double) * k * f);
    memset(count, 0, sizeof(uint64_t) * k);
    for (uint64_t i = 0; i < n; ++i) { /* sums in row order */
      const uint64_t a = assign[i];
      for (uint64_t j = 0; j < f; ++j) sum[a * f + j] += rows[i * f + j];
      ++count[a];
    }
    for (uint64_t c = 0; c < k; ++c)
      if (count[c] > 0)
        for (uint64_t j = 0; j < f; ++j) cents[c * f + j] = sum[c * f + j] / (double)count[c]; /* empty: unchanged */
  }
  free(init);
  free(prev);
  free(count);
  free(sum);
  return ASO_OK;
}

/* centroids and counts of an assignment: sums in row order then divided (src/clustering.rs:330-352,387-408) */
static void centroids_of(const double *rows, uint64_t n, uint64_t f, const uint64_t *assign, uint64_t k, double *cents,
                         uint64_t *count) {
  memset(cents, 0, sizeof(double) * k * f);
  memset(count, 0, sizeof(uint64_t) * k);
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t a = assign[i];
    if (a >= k) continue;
    for (uint64_t j = 0; j < f; ++j) cents[a * f + j] += rows[i * f + j];
    ++count[a];
  }
  for (uint64_t c = 0; c < k; ++c)
    if (count[c] > 0)
      for (uint64_t j = 0; j < f; ++j) cents[c * f + j] /= (double)count[c];
}

/* src/clustering.rs:314-381 (the reference's rayon sums are unordered; sequential here) */
double aso_calinski_harabasz(const double *rows, uint64_t n, uint64_t f, const uint64_t *assign, uint64_t k) {
  if (k <= 1 || k >= n) return 0.0;
  double *g = (double *)calloc(f, sizeof(double));
  double *cents = (double *)malloc(sizeof(double) * k * f);
  uint64_t *count = (uint64_t *)malloc(sizeof(uint64_t) * k);
  for (uint64_t j = 0; j < f; ++j) {
    double s = 0.0;
    for (uint64_t i = 0; i < n; ++i) s += rows[i * f + j];
    g[j] = s / (double)n;
  }
  centroids_of(rows, n, f, assign, k, cents, count);
  double bgss = 0.0;
  for (uint64_t c = 0; c < k; ++c)
    if (count[c] > 0) bgss += (double)count[c] * dist2(cents + c * f, g, f);
  double wgss = 0.0;
  for (uint64_t i = 0; i < n; ++i)
    if (assign[i] < k) wgss += dist2(rows + i * f, cents + assign[i] * f, f);
  free(g);
  free(cents);
  free(count);
  if (wgss < 1e-10) return 0.0;
  return (bgss / (double)(k - 1)) / (wgss / (double)(n - k));
}

static double best_ch(const double *rows, uint64_t n, uint64_t f, uint64_t k, uint64_t seed0, uint64_t *assign, double *cents) {
  double best = 0.0;
  int have = 0;
  for (uint64_t trial = 0; trial < 3; ++trial) {
    aso_kmeans_lloyd(rows, n, f, k, 20, seed0 + trial, assign, cents);
    const double s = aso_calinski_harabasz(rows, n, f, assign, k);
    if (!have || s >= best) { /* max_by keeps the last of equal maxima */
      best = s;
      have = 1;
    }
  }
  return best;
}

/* src/clustering.rs:167-310 on the (already sampled) rows */
uint64_t aso_step2_calinski_harabasz(const double *rows, uint64_t n, uint64_t f, uint64_t k_min, uint64_t k_max, uint64_t seed) {
  if (n < 10) return k_min;
  const uint64_t range = k_max - k_min, step = range <= 5 ? 1 : (range <= 15 ? 2 : 3);
  uint64_t *assign = (uint64_t *)malloc(sizeof(uint64_t) * n);
  double *cents = (double *)malloc(sizeof(double) * (k_max + 1) * f);
  uint64_t best_k = k_min;
  double best_score = -INFINITY;
  int have = 0;
  for (uint64_t k = k_min; k <= k_max; k += step) {
    if (!(k < n && k >= 2)) continue;
    const double sc = best_ch(rows, n, f, k, seed + k * 1000, assign, cents) - 0.8 * (double)k * log((double)n);
    /* max_by (:229-241): the greater score; equal or incomparable scores: the larger k (k ascends here) */
    if (!have || !(sc < best_score)) {
      best_k = k;
      best_score = sc;
      have = 1;
    }
  }
  if (!have) best_score = -INFINITY;
  if (step > 1) {
    uint64_t fine[5];
    fine[0] = best_k >= step - 1 ? best_k - (step - 1) : 0;
    fine[1] = best_k >= 1 ? best_k - 1 : 0;
    fine[2] = best_k;
    fine[3] = best_k + 1 < k_max ? best_k + 1 : k_max;
    fine[4] = best_k + step - 1 < k_max ? best_k + step - 1 : k_max;
    uint64_t fk = 0;
    double fs = 0.0;
    int fhave = 0;
    for (int t = 0; t < 5; ++t) {
      const uint64_t k = fine[t];
      if (!(k >= k_min && k <= k_max && k < n)) continue;
      if (k >= k_min && (k - k_min) % step == 0) continue; /* already a candidate */
      const double sc = best_ch(rows, n, f, k, seed + k * 10000, assign, cents) - 0.8 * (double)k * log((double)n);
      if (!fhave || !(sc < fs)) { /* the list is in ascending k: same rule */
        fk = k;
        fs = sc;
        fhave = 1;
      }
    }
    if (fhave && fs > best_score) {
      best_k = fk;
      best_score = fs;
    }
  }
  free(assign);
  free(cents);
  return best_k < k_max ? best_k : k_max;
}

static int cmp_f64(const void *a, const void *b) {
  const double x = *(const double *)a, y = *(const double *)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

/* src/clustering.rs:384-492 on the (already sampled) rows */
double aso_threshold_from_pilot(const double *rows, uint64_t n, uint64_t f, uint64_t k, uint64_t seed) {
  if (k > n) k = n;
  uint64_t *assign = (uint64_t *)malloc(sizeof(uint64_t) * n);
  double *cents = (double *)malloc(sizeof(double) * k * f);
  uint64_t *count = (uint64_t *)malloc(sizeof(uint64_t) * k);
  double *dists = (double *)malloc(sizeof(double) * n);
  aso_kmeans_lloyd(rows, n, f, k, 20, seed + 100000, assign, cents);
  centroids_of(rows, n, f, assign, k, cents, count);
  uint64_t nd = 0;
  for (uint64_t i = 0; i < n; ++i)
    if (assign[i] < k) dists[nd++] = dist2(rows + i * f, cents + assign[i] * f, f);
  double radius;
  if (nd == 0) {
    radius = 1.0;
  } else {
    qsort(dists, nd, sizeof(double), cmp_f64);
    uint64_t idx = (uint64_t)ceil((double)nd * 0.9);
    if (idx > nd - 1) idx = nd - 1;
    const double p90 = dists[idx];
    double min_inter = INFINITY;
    int any_inter = 0;
    for (uint64_t a = 0; a < k; ++a)
      for (uint64_t b = a + 1; b < k; ++b)
        if (count[a] > 0 && count[b] > 0) {
          const double d = dist2(cents + a * f, cents + b * f, f);
          any_inter = 1;
          if (d < min_inter) min_inter = d;
        }
    const double ratio = (min_inter < INFINITY && min_inter > 0.0) ? p90 / min_inter : 1.0;
    if (p90 < 1e-8 || ratio < 0.01) {
      radius = any_inter ? fmax(min_inter * 0.15, 1e-6) : 1e-6;
    } else {
      radius = fmax(p90 * 1.5, 1e-6);
    }
  }
  free(assign);
  free(cents);
  free(count);
  free(dists);
  return radius;
}

/* src/clustering.rs:36-95: (k_optimal, radius, intrinsic dimension) */
int aso_compute_optimal_k(const double *rows, uint64_t n, uint64_t f, uint64_t seed, uint64_t *k_out, double *radius_out,
                          uint64_t *id_out) {
  /* step 1 */
  const uint64_t ns_id = n < 500 ? n : 500;
  uint64_t *ids = (uint64_t *)malloc(sizeof(uint64_t) * 1000);
  double *ratios = (double *)malloc(sizeof(double) * 500);
  uint64_t id_est;
  if (n < 10) {
    id_est = f < 2 ? f : 2;
  } else {
    aso_pre_sample_indices(n, ns_id, seed + 1, ids);
    aso_two_nn_ratios(rows, n, f, ids, ns_id, ratios);
    id_est = aso_intrinsic_dimension(n, f, ratios, ns_id);
  }
  uint64_t k_min = (uint64_t)ceil(sqrt((double)n / 10.0));
  if (k_min < 2) k_min = 2;
  uint64_t k_max = f;
  if (n / 10 < k_max) k_max = n / 10;
  if (5 * id_est < k_max) k_max = 5 * id_est;
  if ((uint64_t)pow((double)n, 0.5) < k_max) k_max = (uint64_t)pow((double)n, 0.5);
  if (k_max < k_min + 1) k_max = k_min + 1;
  if (k_max > n / 2) k_max = n / 2;
  /* sample of <= 1000 rows */
  const uint64_t ss = n < 1000 ? n : 1000;
  double *sample = (double *)malloc(sizeof(double) * ss * f);
  if (n > ss) {
    aso_pre_sample_indices(n, ss, seed, ids);
    for (uint64_t s = 0; s < ss; ++s) memcpy(sample + s * f, rows + ids[s] * f, sizeof(double) * f);
  } else {
    memcpy(sample, rows, sizeof(double) * n * f);
  }
  const uint64_t k_opt = aso_step2_calinski_harabasz(sample, ss, f, k_min, k_max, seed);
  *radius_out = aso_threshold_from_pilot(sample, ss, f, k_opt, seed);
  *k_out = k_opt;
  *id_out = id_est;
  free(ids);
  free(ratios);
  free(sample);
  return ASO_OK;
}

/* src/clustering.rs:547-910 without sampling, restated for a batch size (see the header).  centroids: max_clusters x f,
 * assignments: -1 = none; returns the number of clusters. */
int64_t aso_incremental_clustering(const double *rows, uint64_t n, uint64_t f, uint64_t max_clusters, double radius,
                                   uint64_t batch, double *cents, int64_t *assign, uint64_t *counts) {
  uint64_t nc = 0;
  if (batch == 0) batch = 1;
  double *snap_d2 = (double *)malloc(sizeof(double) * batch);
  uint64_t *snap_i = (uint64_t *)malloc(sizeof(uint64_t) * batch);
  for (uint64_t i = 0; i < n; ++i) assign[i] = -1;
  uint64_t done = 0;
  for (uint64_t b0 = 0; b0 < n; b0 += done) {
    const uint64_t bn = n - b0 < batch ? n - b0 : batch;
    const uint64_t nc_snap = nc;
    done = bn;
    /* phase 1 for the whole batch: nearest centroid of the snapshot (:571-594) */
    for (uint64_t r = 0; r < bn; ++r) {
      if (nc_snap == 0) {
        snap_i[r] = 0;
        snap_d2[r] = INFINITY;
      } else {
        snap_i[r] = nearest(rows + (b0 + r) * f, cents, nc_snap, f, &snap_d2[r]);
      }
    }
    /* phases 2-3 in row order; the centroids created earlier in this batch are seen with their current values */
    for (uint64_t r = 0; r < bn; ++r) {
      const double *row = rows + (b0 + r) * f;
      uint64_t bi = snap_i[r];
      double bd = snap_d2[r];
      for (uint64_t c = nc_snap; c < nc; ++c) {
        const double d = dist2(row, cents + c * f, f);
        if (d < bd) {
          bd = d;
          bi = c;
        }
      }
      if (nc == 0) { /* first centroid (:637-656) */
        memcpy(cents, row, sizeof(double) * f);
        counts[0] = 1;
        assign[b0 + r] = 0;
        nc = 1;
      } else if (nc < max_clusters && bd > radius * 0.5) { /* new cluster (:670-713) */
        memcpy(cents + nc * f, row, sizeof(double) * f);
        counts[nc] = 1;
        assign[b0 + r] = (int64_t)nc;
        ++nc;
      } else if (bd <= radius) { /* assign + running mean (:714-768) */
        const double k_new = (double)counts[bi] + 1.0;
        for (uint64_t j = 0; j < f; ++j) cents[bi * f + j] += (row[j] - cents[bi * f + j]) / k_new;
        counts[bi] += 1;
        assign[b0 + r] = (int64_t)bi;
      } else if (bd <= radius * 1.5) { /* soft outlier: counted, centroid not moved (:769-800) */
        counts[bi] += 1;
        assign[b0 + r] = (int64_t)bi;
      } /* else dropped (:801-826) */
      if (nc_snap < max_clusters && nc == max_clusters) { /* the cap was reached inside this batch: it ends here */
        done = r + 1;
        break;
      }
    }
  }
  free(snap_d2);
  free(snap_i);
  return (int64_t)nc;
}

/* src/reduction.rs:126-140 */
uint64_t aso_jl_dimension(uint64_t n_points, double epsilon) {
  const double log_n = log((double)n_points);
  const double eps_sq = pow(epsilon, 2.0);
  const uint64_t jl = (uint64_t)ceil(8.0 * log_n / eps_sq);
  return jl < 32 ? 32 : jl;
}

/* src/reduction.rs:180-203 with the Gaussian matrix G (original_dim x reduced_dim, the order the reference draws it in)
 * given: result[r] += original * sample * scale, original-major */
int aso_jl_project(const double *rows, uint64_t n, uint64_t d, const double *G, uint64_t r, double *out) {
  const double scale = 1.0 / sqrt((double)r);
#pragma omp parallel for schedule(static)
  for (int64_t ii = 0; ii < (int64_t)n; ++ii) {
    double *o = out + (uint64_t)ii * r;
    for (uint64_t t = 0; t < r; ++t) o[t] = 0.0;
    for (uint64_t i = 0; i < d; ++i) {
      const double x = rows[(uint64_t)ii * d + i];
      for (uint64_t t = 0; t < r; ++t) o[t] += x * G[i * r + t] * scale;
    }
  }
  return ASO_OK;
}
