/*
 * arrowspace_hip.h -- C ABI of libarrowspace_hip.so: the MI355X (gfx950) implementation of
 * arrowspace's lambda-tau index build and lambda-aware search path.
 *
 * This is the drop-in boundary.  The reference (tuned-org-uk/arrowspace-rs v0.18.1) is a pure
 * Rust rlib with no FFI of its own (Cargo.toml:60-62), so the boundary sits on the stage seams
 * the reference already has (trait EigenMaps, src/eigenmaps.rs:93-172); every entry point below
 * names the reference function whose body it replaces.  A Rust maintainer binds these with a
 * plain `extern "C"` block (INTEGRATION.md shows the shim).
 *
 * Conventions
 *   - All matrices are row-major f64; indices are u64 at the boundary (usize in the reference).
 *   - Every data pointer may be a HOST pointer or a DEVICE (HBM) pointer; the library asks
 *     the HIP runtime which it is.  Host buffers are staged over PCIe on the context's stream;
 *     device buffers are used in place (this is how bench.py and the multi-GPU driver keep the
 *     inputs resident in HBM).
 *   - Inputs are borrowed for the duration of the call.  Outputs are either caller-provided
 *     buffers or library-owned handles released by the matching *_free / *_destroy.
 *   - Every function returns AS_OK (0) or an as_status error; as_last_error() gives the text.
 *     The reference signals the same conditions by panicking; the messages are kept verbatim so
 *     a host shim can re-panic with them.
 *   - One as_ctx per host thread (not re-entrant).  There is NO CPU fallback: without a gfx950
 *     device as_ctx_create fails with AS_ERR_NO_DEVICE.
 */
#ifndef ARROWSPACE_HIP_H
#define ARROWSPACE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AS_ABI_VERSION 5

typedef enum {
  AS_OK = 0,
  AS_ERR_SHAPE = 1,       /* "items should be at least of shape (2,2): (d,n)"  src/laplacian.rs:129-134 */
  AS_ERR_NONFINITE = 2,   /* "Query item contains invalid values (NaN or infinity)..." src/core.rs:534-537 */
  AS_ERR_ZERO_LAMBDA = 3, /* "Lambda of the item is 0.0, prepare the item before searching" src/core.rs:773-776 */
  AS_ERR_NAN_SCORE = 4,   /* partial_cmp().unwrap() on a NaN score, src/core.rs:785 */
  AS_ERR_DIM = 5,         /* graph has more nodes than the signal has entries (src/taumode.rs:570-575 would index out of bounds); query dimension mismatch src/core.rs:510-516 */
  AS_ERR_TOO_SPARSE = 6,  /* "Resulting laplacian matrix is too sparse" src/graph.rs:185-193 */
  AS_ERR_ARG = 7,         /* null pointer / bad enum / unsupported size */
  AS_ERR_UNSUPPORTED = 8, /* outside the accelerated path (e.g. topk + 1 > 256, k > 4096) */
  AS_ERR_HIP = 9,         /* HIP runtime failure; text in as_last_error */
  AS_ERR_NO_DEVICE = 10,  /* no gfx950 device: there is no CPU fallback */
  AS_ERR_PEER = 11        /* as_build_laplacian_sharded: another rank failed locally; this rank stopped with it */
} as_status;

/* TauMode, src/taumode.rs:75-82 */
typedef enum { AS_TAU_FIXED = 0, AS_TAU_MEDIAN = 1, AS_TAU_MEAN = 2, AS_TAU_PERCENTILE = 3 } as_tau_mode;

/* GraphParams, src/graph.rs:94-102 (same field names) */
typedef struct {
  double eps;             /* maximum cosine distance kept */
  uint64_t k;             /* stored only; the graph code reads topk (src/laplacian.rs:211,246) */
  uint64_t topk;          /* neighbours requested per node = topk + 1 */
  double p;               /* kernel exponent */
  double sigma;           /* used when has_sigma != 0, else 1.0 (src/laplacian.rs:254) */
  int32_t has_sigma;      /* Option<f64> discriminant */
  int32_t normalise;      /* 1: per-column z-score first (StandardScaler, src/laplacian.rs:146-151; population variance, zero std -> 1: smartcore's convention is unpinned) */
  int32_t sparsity_check; /* carried for the host-side GraphFactory check (src/graph.rs:185-193) */
  int32_t rectify;        /* 0: distance = 1 - cos (default).  1: 1 - max(0, cos) */
  /* smartcore's CosinePair (source absent) either leaves the query row out of
   * query_row_top_k(i, topk + 1) or returns it as its own nearest neighbour; the reference then filters i != j
   * (src/laplacian.rs:224,252), so a node keeps up to topk + 1 neighbours in the first case (0, default) and up to topk
   * in the second (1: the row competes with its own distance 1 - x.x / (|x||x|) for one of the topk + 1 places). */
  int32_t self_in_results;
  int32_t reserved;       /* 0 */
} as_graph_params;

typedef struct as_ctx as_ctx;     /* device, stream, workspace */
typedef struct as_csr as_csr;     /* CSR Laplacian resident in HBM (sprs::CsMat<f64> counterpart) */
typedef struct as_index as_index; /* items + lambdas resident in HBM (ArrowSpace.data / .lambdas) */
typedef struct as_jl as_jl;       /* a seeded Gaussian projection resident in HBM (ImplicitProjection, src/reduction.rs:166-203) */

/* ---- context -------------------------------------------------------------------------- */
int as_abi_version(void);
int as_ctx_create(int device, as_ctx **out);
void as_ctx_destroy(as_ctx *ctx);
const char *as_last_error(const as_ctx *ctx);
const char *as_status_message(int status); /* the reference's panic text for a status */
/* Run on an existing HIP stream (e.g. PyTorch's current stream; NULL = HIP's default stream).
 * Until this is called the context uses a private stream and every call is synchronous with
 * the host.  Afterwards calls whose outputs are device buffers are stream-ordered (no host
 * sync); host outputs are always complete on return. */
int as_ctx_set_stream(as_ctx *ctx, void *hip_stream);
int as_ctx_synchronize(as_ctx *ctx);
/* Options.  None of them changes a result -- only which stage produces it; defaults in parentheses.  The 35 below are the supported
 * ones; csrc/common.h (struct as_ctx) lists, next to these, the switches that exist for the tests and for measurements only (forms
 * of one stage that must agree bit for bit, failure injection, sample sizes): they are accepted here too but are no interface.
 *  stages
 *   "knn_mfma" (1), "search_mfma" (1)   MFMA candidate passes + exact re-rank + certification; 0 = the exact brute-force scan only
 *   "cand_precision" (32)       first candidate pass of the non-join path on the f32 pipe, or 64: on the f64 pipe
 *   "knn_join" (1)              symmetric threshold join for whole-matrix item graphs of "join_min_rows" (32768) rows or more
 *   "join_pipe" (16)            its contraction: 16 = one f16 plane, 48 = split bf16 in two phases (dim <= 128), 32 = f32 MFMA
 *   "join_sample", "join_rank", "join_cap" (0 = automatic: 32 / 4 / 512)   threshold sample 1-in-N, its order statistic, list slots per row
 *   "stream_scan" (1)           streaming exact scans for the rows / queries the candidate stages leave open
 *  the join's preparation
 *   "join_prefilter" (1)        int8 first pass in front of the f16 join, from "join_prefilter_min_rows" (65536) rows: 1 = on the rotated
 *                               32-wide image (31 dominant directions + the tail norm), 2 = full-width image always, 3 = rotated always, 0 = off
 *   "join_row_order" (1)        rows grouped by direction in the operand image: "join_order_pivots" (1024; 2048 beyond 3 M rows while the
 *                               default stands) pivots on the rotated coordinates,
 *                               "join_order_sub" (4): inside a group by a second set of pivots / 4, "join_order_chain" (1): the groups laid
 *                               out along a greedy chain of similar pivots (neighbours of a row end up near it in the image);
 *                               "join_order_cover" (150): where more than this many per mille of the rows lie far from their pivot (data
 *                               with more clusters than pivots) further pivots are drawn from those rows, up to 4096 (0 = never)
 *   "join_theta_grouped" (1)    thresholds against the sampled rows of the 16 pivot groups nearest to a row's own (checked against the plain sample)
 *   "join_theta_band" (2)       thresholds raised to what a row's own neighbourhood in the ordered image certifies (512-position super tiles
 *                               on either side; 0 = off); "join_theta_band_slack" (4): the certified cut is also taken where the sampled
 *                               threshold is higher by at most this many rounding bounds
 *   "join_theta_far" (1)        rows far from their pivot, and rows whose threshold lies 0.05 below the median of their 64-position block,
 *                               take the larger of it and the threshold against the whole sample
 *   "join_cone_tiles" (1), "join_cone_second" (1)   cone bound on pairs of 128-position tiles in front of the first pass / its one-sided
 *                               form in front of the second pass
 *   "join_cone_groups" (1)      the cone bound in two levels from 2048 tiles on: cones around 32 tiles first (0 = the fine level alone)
 *   "join_order_pivots_auto" (1) the first pivots follow the row count (see "join_order_pivots"; 0 = the number given, always)
 *   "join_theta_rows64" (1)     the threshold sweeps give a wave 64 rows (two MFMAs per fragment read, two workgroups per CU; 0 = 32 rows)
 *   "join_fused_rows_pass" (1)  rows of exactly 128 values: the rotated coordinates and the f16 image from one pass over the rows (0 = two)
 *   "join_pack_wide_tiles" (1)  the f16 image of rows of a multiple of 128 columns in one coalesced pass (0 = the per-fragment packer)
 *   "join_open_order" (1)       the rows the first re-rank leaves open go on in the order of the image (0 = in the order they were flagged)
 *  (the last six exist so that the tests can build the same table both ways; every setting gives the same result)
 *  the exact re-rank
 *   "rerank_pair_bound" (1)     the pruning bounds each pair by its two rows' own f16 plane errors
 *   "rerank_share" (1)          a pair two rows both keep is fetched and scored once, its key handed to both (0: every row scores all its own)
 *   "rerank_image_order" (1)    rows wider than 128: the records of the re-rank lie in the order of the join's image (0 = in row order)
 *  search
 *   "search_scan" (1)           threshold scan of a query batch on the 16-bit pipe (0 < alpha <= 1)
 *   "search_scan_order", "search_prefilter", "search_stream_filter" (1)   an index of 65536 items or more that is searched again keeps
 *                               ordered / rotated images: 1 = from the second search on, 0 = never (2 = at the first, for "search_scan_order")
 *   "search_lambda_order" (1)   a batch of 1024 queries or more is scanned in the order of its lambdas
 *  item-graph lambda
 *   "signal_sym" (1)            a bitwise symmetric graph is streamed as its upper triangle
 *   "signal_order" (1)          rows visited in the order attached to the graph (as_csr_set_order; as_build_laplacian attaches the join's)
 *  resources
 *   "handle_cache_mb" (2048)    memory of destroyed handles kept for the next allocation of the same size (process-wide; environment
 *                               ARROWSPACE_HANDLE_CACHE_MB; 0 = off and released)
 *   "release_workspace"         (an action: any value) the context's cached workspace blocks -- a call keeps what it took for the next
 *                               one -- are returned to the device now
 *   "sharded_agreement" (1)     every collective of a sharded call behind a one-word status agreement (as_*_sharded)
 *  tuning constants and test hooks (every setting gives the same results; kept as options so that the tests and the measurement
 *  scripts can take the other branch; defaults in parentheses)
 *   "cand_slack" (8) candidates kept beyond topk + 1 by the top-k passes; "join_bf16" (1) thresholds of the bf16 / f32 joins from bf16
 *   products; "join_block" (64) column super tiles per unit of the join passes; "join_one_sided_block" (2) the same for the one-sided
 *   f16 pass behind a mask or on an ordered image; "join_unit_list" (1) the masked f16 pass walks a list of the units that hold
 *   anything; "join_basis_iters" (4) rounds of subspace iteration behind the rotated coordinates, "join_basis_sample" (16384) rows of
 *   the strided sample they run on (8192: 0.1 - 0.4 ms less per build up to rank 64, 0.8 ms more at rank 128); "join_zproj_mfma" (1) those
 *   coordinates of 256 .. 1024-wide rows on the matrix pipe; "join_theta_rate" (32) the grouped thresholds' sample, one position
 *   in N; "join_theta_parts" (1) threshold passes with few rows split the sample over several units; "join_theta_grouped_bad_e3" (10)
 *   per mille of checked rows that may sit below the plain sample's thresholds before the grouped ones are dropped;
 *   "join_theta_dip_e3" (50) how far (per mille) below its block's median a threshold sends a row to the whole sample;
 *   "join_theta_band_auto" (1) the band grows with the clusters the row order found; "join_theta_band_second" (1) the second pass
 *   never cuts below what the band certified; "join_cone_tiles_max_e3" (600) / "join_cone_split_above_e3" (80) per mille of tile
 *   pairs kept beyond which the cone bound is dropped / taken again with two cones per tile; "join_second_prefilter" (1) wide rows:
 *   an int8 first pass in front of the second pass; "second_pass_f16" (1) the second pass on the f16 plane (0: split bf16);
 *   "rerank_stream_cap" (0) test hook: batches the re-rank's slot stream gets room for at first; "assembly_lanes" (2) graph assembly
 *   with 32 lanes a row (0: a thread a row); "lambda_v2" (1), "lambda_v3" (1), "lambda_wide" (1) which lambda-tau kernel a width
 *   takes; "nearest_mfma" (1), "cluster_on_device" (1) the pre-stage's nearest-centroid search on the matrix pipe / its batches on
 *   the device; "search_scan_f16" (1), "search_scan_sample" (32), "search_scan_rank" (3), "search_scan_cap" (512),
 *   "search_rot_tail_e3" (100), "search_staged_min_batch" (2^30), "search_stream_quick" (1) the search scan's plane, its guess
 *   (the m-th best of a 1-in-N sample), list slots per query, when a re-searched index keeps rotated images, batch sizes of the
 *   staged and the quick streaming form; "signal_min_len" (9000), "signal_tau_bracket" (1), "signal_tau_overlap" (1) item-graph
 *   lambda: from which signal length as_compute_taumode takes it, the order statistics from a sampled bracket, on a side stream;
 *   "sharded_own_rows" (1) a sharded build prepares the join from the ranks' own rows; "sharded_order_assign" (1) ... and finds
 *   the pivots of own rows only (assignments all-gathered); "join_basis_fused" (1) rows of 128 values: a round of the subspace iteration that finds the 31 directions in
 *   one kernel (0: two); "keep_join_images" (0) an item-graph build (one GPU, rows of 128 values) keeps its row order, the f16
 *   image in it, the pivots, the 31 directions and the rotated int8 image with the context (~300 MB per million rows, until the next
 *   build, as_index_create or 0 here), "keep_join_images_min_rows" (262144) from this many rows; "index_adopt_join_images" (1) as_index_create takes them over when its items are those rows --
 *   the same device pointer, shape and row norms; THE CALLER MUST NOT HAVE WRITTEN TO THE ROWS IN BETWEEN -- and the first search then
 *   runs on the ordered image behind the int8 first pass (stat index_adopted_join_images);
 *   "join_second_slices" (1) the second pass of an item graph on that scan too;
 *   "search_scan_run" (1) units a wave of that scan draws at a time; "search_scan_slices" (2) the one-sided f16 scan behind a first pass: a wave per (64 queries, N stages of 256 columns) on its
 *   own (0: a workgroup per 512 queries, a stage of columns shared through LDS); "join_masked_block" (0) column super tiles per work unit of the masked f16 pass alone (0: "join_block", narrower for a
 *   short range of rows); "rerank_prune_halves" (1) the re-rank's pruning takes a row per half wave, lists of at most 128 entries in registers;
 *   "sharded_split_assembly" (1) degrees, weights and pruning of a row on its
 *   owner, the adjacency exchanged in place of the neighbour tables; "sharded_inject_failure" (0) test hook: this rank fails at checkpoint N;
 *   "debug_label_mod" (0) diagnostic: label statistics per image tile
 *   "keep_open_rows" (0)        diagnostic: 1 = the ids of the rows the first exact re-rank of an item graph leaves open (the rows that
 *                               go on to the second pass, the top-k passes or an exact scan) are kept and read back as the stats
 *                               knn_open_rows_kept (how many) and knn_open_row_<i>
 *   "debug"                     diagnostics only (outputs may be undefined) */
int as_ctx_set_option(as_ctx *ctx, const char *name, int64_t value);
int64_t as_ctx_get_stat(const as_ctx *ctx, const char *name);

/* Per-kernel device timing (HIP events on the context's stream).  Enable, run calls, then read
 * the accumulated milliseconds / launch count of a named kernel group. */
int as_profile_enable(as_ctx *ctx, int on);
int as_profile_reset(as_ctx *ctx);
int as_profile_get(as_ctx *ctx, const char *name, double *ms_total, uint64_t *launches);
/* writes up to cap names separated by '\n' */
int as_profile_names(as_ctx *ctx, char *buf, uint64_t cap);

/* ---- residency for a host caller -----------------------------------------------------------------------------------------
 * The reference's build() walks host rows (Vec<Vec<f64>>, src/builder.rs:249-304 makes three host copies of them).  Here
 * every data pointer may be an HBM pointer, so a host that builds the graph, the lambdas and the index from the same item
 * matrix uploads it ONCE and passes the device pointer to every call.  as_device_upload allocates `bytes` of HBM and fills
 * them from host memory (pipelined through the context's pinned staging buffer: several host threads copy chunk c + 1 while
 * the DMA of chunk c is in flight); as_device_download copies back; as_device_free releases.  A host pointer passed straight
 * to a call goes through the same pipeline once per call. */
int as_device_upload(as_ctx *ctx, const void *host, uint64_t bytes, void **dev_out);
int as_device_download(as_ctx *ctx, const void *dev, uint64_t bytes, void *host_out);
void as_device_free(as_ctx *ctx, void *dev);

/* ---- builder helper --------------------------------------------------------------------- */
/* ArrowSpaceBuilder::define_result_k, src/builder.rs:225-233 */
uint64_t as_define_result_k(uint64_t k, uint64_t topk);

/* ---- graph: replaces build_laplacian_matrix, src/laplacian.rs:122-178 -------------------- */
/* rows: n_nodes x dim, ROWS ARE NODES (GraphFactory passes centroids^T, src/graph.rs:170-172;
 * the item graph passes the items themselves, src/energymaps.rs:270).  Output: n_nodes x
 * n_nodes CSR, columns ascending, explicit diagonal (also when 0, src/laplacian.rs:369-370). */
int as_build_laplacian(as_ctx *ctx, const double *rows, uint64_t n_nodes, uint64_t dim,
                       const as_graph_params *params, as_csr **out);

/* Stage-level entry: the neighbour search alone (CosinePair::query_row_top_k for every row of
 * [row_begin, row_begin + row_count), src/laplacian.rs:211,221,246).  nbr_idx / nbr_dist are
 * row_count x (topk+1), ordered (distance asc, index asc); missing entries idx = UINT64_MAX. */
int as_knn_cosine(as_ctx *ctx, const double *rows, uint64_t n_nodes, uint64_t dim,
                  uint64_t row_begin, uint64_t row_count, const as_graph_params *params,
                  uint64_t *nbr_idx, double *nbr_dist);

/* ---- clustering pre-stage: replaces the row loops around nearest_centroid, src/clustering.rs:913-928 (callers
 * :590, :721, :764) ------------------------------------------------------------------------------------------
 * For every row the index of its nearest centroid and the squared Euclidean distance, in the reference's
 * arithmetic (sequential sum of (a - b)^2 over the features; strict '<' over the centroids in index order, so the
 * first of equal minima wins; (0, +inf) for a row whose distances are all NaN).  rows: n_rows x dim, centroids:
 * n_centroids x dim, both host or device; out_idx[n_rows], out_dist2[n_rows]. */
int as_nearest_centroids(as_ctx *ctx, const double *rows, uint64_t n_rows, uint64_t dim, const double *centroids,
                         uint64_t n_centroids, uint64_t *out_idx, double *out_dist2);

/* ---- clustering pre-stage: ClusteringHeuristic + run_incremental_clustering_with_sampling, src/clustering.rs ---------
 * What runs inside ArrowSpaceBuilder::build immediately before the graph (start_clustering, src/eigenmaps.rs:175-235).
 * Every pass over all N rows runs on the device in the reference's arithmetic (the 500 x N Two-NN distances, the
 * nearest-centroid searches); the bookkeeping on the <= 1000-row sample and the centroid updates run on the host in
 * the reference's order.  The reference's random draws (StdRng shuffles, smartcore KMeans, both absent here) are
 * replaced by the build's own seeded definitions (csrc/prestage.hip), so results agree with the oracle bit for bit
 * and with a reference run statistically (src/tests/test_taumode.rs:328-428 asks no more of its own two runs).
 * Small outputs (k, radius, ratios, centroids, assignments, counts) are HOST pointers; `rows` may be host or device. */
/* Two-NN ratios d2 / d1 of the sampled rows (estimate_intrinsic_dimension, :117-147); NaN where the reference skips a row */
int as_two_nn_ratios(as_ctx *ctx, const double *rows, uint64_t n_rows, uint64_t n_features, const uint64_t *sample_ids,
                     uint64_t n_samples, double *ratios_out);
/* kmeans_lloyd (:494-531) with the build's own seeded Lloyd in place of smartcore's: assignments_out[n_rows],
 * centroids_out[k x n_features] (or NULL) */
int as_kmeans_lloyd(as_ctx *ctx, const double *rows, uint64_t n_rows, uint64_t n_features, uint64_t k, uint64_t max_iter,
                    uint64_t seed, uint64_t *assignments_out, double *centroids_out);
/* calinski_harabasz_score (:314-381) */
int as_calinski_harabasz(as_ctx *ctx, const double *rows, uint64_t n_rows, uint64_t n_features, const uint64_t *assignments,
                         uint64_t k, double *score_out);
/* compute_optimal_k (:36-72): step 1 bounds + Two-NN intrinsic dimension, step 2 Calinski-Harabasz sweep on a 1000-row
 * sample, step 3 radius from a pilot k-means */
int as_compute_optimal_k(as_ctx *ctx, const double *rows, uint64_t n_rows, uint64_t n_features, uint64_t seed, uint64_t *k_out,
                         double *radius_out, uint64_t *id_out);
/* run_incremental_clustering_with_sampling without a sampler (:547-910).  batch = 1 is the reference's sequential mode
 * (with_seed) exactly; a larger batch takes the snapshot of :571-594 once per batch (a legal schedule of its parallel
 * mode) and lets a row also see the centroids created earlier in its batch; a batch ends with the row that creates the
 * last centroid max_clusters allows (the next one starts with a snapshot of all of them).  No row visits the host for
 * n_features <= 1024.  centroids_out: max_clusters x n_features (the first *n_clusters_out rows are written),
 * assignments_out[n_rows] (-1: dropped), counts_out[max_clusters]. */
int as_incremental_clustering(as_ctx *ctx, const double *rows, uint64_t n_rows, uint64_t n_features, uint64_t max_clusters,
                              double radius, uint64_t batch, double *centroids_out, uint64_t *n_clusters_out,
                              int64_t *assignments_out, uint64_t *counts_out);

/* ---- JL projection: src/reduction.rs ------------------------------------------------------------------------------------ */
uint64_t as_jl_dimension(uint64_t n_points, double epsilon); /* compute_jl_dimension, :126-140 */
/* ImplicitProjection (:166-203): an original_dim x reduced_dim N(0,1) matrix drawn from `seed` (the build's own
 * counter-based stream: the reference's ChaCha8 draw is not reproducible without its crate), kept in HBM */
int as_jl_create(as_ctx *ctx, uint64_t original_dim, uint64_t reduced_dim, uint64_t seed, as_jl **out);
void as_jl_destroy(as_jl *jl);
uint64_t as_jl_original_dim(const as_jl *jl);
uint64_t as_jl_reduced_dim(const as_jl *jl);
int as_jl_matrix(as_ctx *ctx, const as_jl *jl, double *g_out); /* the matrix, original_dim x reduced_dim, row-major */
/* project_matrix / ImplicitProjection::project (:142-164,180-203), batched: out[n_rows x reduced_dim];
 * out[t] = sum_i (x[i] * G[i][t]) * (1 / sqrt(reduced_dim)), i ascending, rounded as the reference rounds it.
 * ArrowSpace::project_query (src/core.rs:509-529) is this call on the queries. */
int as_jl_project(as_ctx *ctx, const as_jl *jl, const double *rows, uint64_t n_rows, double *out);

/* Row-sharded form of the same neighbour search (one process per GPU).  cos(i,j) =
 * cos(j,i): every unordered pair of 256-row tiles is contracted once, by the rank that owns one of the two
 * tiles, and serves both rows -- so the candidate lists of a rank's rows are partly produced on other ranks.
 * rows = ALL n_nodes rows in HBM (after the all-gather of the shards); [row_begin, row_begin + row_count) =
 * this rank's rows, row_begin a multiple of 512.  All pointers below are device pointers.
 *   1. as_knn_join_thresholds: theta_out[row_count], the per-row candidate thresholds of this rank's rows;
 *      the host all-gathers them into theta_all[n_nodes].
 *   2. as_knn_join_pairs: contracts this rank's tile pairs; cnt[n_nodes] (zeroed by the call) and
 *      cand[n_nodes x cap] receive, indexed by GLOBAL row, every pair that reaches a threshold.  cap =
 *      as_knn_join_capacity() = 32-bit words per row: a list slot is two words, the other row's index and the
 *      approximate cosine (f32 bits) that put it there (the re-rank uses it to drop, before fetching any row,
 *      the candidates that cannot reach the top K); cnt counts slots.  The host
 *      sends each rank the slices of its rows (all-to-all): cnt_src[n_src x row_count],
 *      cand_src[n_src x row_count x cap].
 *   3. as_knn_join_finish: exact f64 re-rank + certification of this rank's rows from the n_src lists; rows
 *      that cannot be certified go through the same later stages as in as_knn_cosine.  Output as there.
 * The three calls form ONE sequence over the same `rows` on one context: the first builds the row norms and the
 * packed operand image and keeps them in the context, the last releases them (calling 2 or 3 out of sequence
 * only costs rebuilding them).
 * AS_ERR_UNSUPPORTED when the join does not apply (small n, normalise, f64-first mode, host
 * pointers): the caller then uses as_knn_cosine on its row range.
 * as_knn_join_applies: 1 when the sequence applies to an n_nodes x dim problem under the context's options (it depends
 * on nothing else, so every rank of a job gets the same answer, a rank that owns no rows included), else 0. */
int as_knn_join_applies(as_ctx *ctx, uint64_t n_nodes, uint64_t dim, const as_graph_params *params);
int as_knn_join_thresholds(as_ctx *ctx, const double *rows, uint64_t n_nodes, uint64_t dim, uint64_t row_begin,
                           uint64_t row_count, const as_graph_params *params, float *theta_out);
uint32_t as_knn_join_capacity(as_ctx *ctx, const as_graph_params *params);
int as_knn_join_pairs(as_ctx *ctx, const double *rows, uint64_t n_nodes, uint64_t dim, uint64_t row_begin,
                      uint64_t row_count, const as_graph_params *params, const float *theta_all, uint32_t cap,
                      uint32_t *cnt, uint32_t *cand);
/* The exchange between steps 2 and 3 in compact form (optional: the dense arrays work too).  as_knn_join_pack: the
 * used slots of rows [0, n_rows) of (cnt, cand) written side by side into packed (capacity n_rows x cap words),
 * row_offsets[n_rows + 1] = exclusive scan of min(cnt, cap / 2) in SLOTS (2 words each) -- the owner g of a row range
 * gets packed[2 row_offsets[lo_g] .. 2 row_offsets[hi_g]) and cnt[lo_g .. hi_g).  as_knn_join_unpack: the inverse on
 * the receiving side, for n_rows = n_src x row_count rows in source-major order: cand_out as as_knn_join_finish takes
 * it (only the used slots are written).  Device pointers. */
int as_knn_join_pack(as_ctx *ctx, const uint32_t *cnt, const uint32_t *cand, uint64_t n_rows, uint32_t cap,
                     uint64_t *row_offsets, uint32_t *packed);
int as_knn_join_unpack(as_ctx *ctx, const uint32_t *cnt, const uint32_t *packed, uint64_t n_rows, uint32_t cap,
                       uint32_t *cand_out);
int as_knn_join_finish(as_ctx *ctx, const double *rows, uint64_t n_nodes, uint64_t dim, uint64_t row_begin,
                       uint64_t row_count, const as_graph_params *params, const float *theta_all, uint32_t n_src,
                       uint32_t cap, const uint32_t *cnt_src, const uint32_t *cand_src, uint64_t *nbr_idx,
                       double *nbr_dist);

/* Stage-level entry: _build_adjacency's filter/weight/sparsify + _symmetrise_adjancency +
 * _build_sparse_laplacian (src/laplacian.rs:217-417) from a complete neighbour table. */
int as_laplacian_from_knn(as_ctx *ctx, const uint64_t *nbr_idx, const double *nbr_dist,
                          uint64_t n_nodes, const as_graph_params *params, as_csr **out);

/* Row-sharded form: rows [row_begin, row_begin + row_count) of the same Laplacian, from the complete (all-gathered)
 * neighbour table.  Every rank of a sharded build assembles its own row block: the table passes that need all rows
 * (degrees, weights, pruning: src/laplacian.rs:217-288) are replicated, the reverse-edge scatter, the unions and the
 * CSR rows (src/laplacian.rs:295-417) are done for the range only.  The handle that comes out is a ROW BLOCK:
 * as_csr_n = row_count rows, as_csr_cols = n_nodes, GLOBAL column ids, indptr starting at 0, as_csr_row_begin = its
 * first row.  Stacking the blocks of consecutive ranges gives exactly the CSR of as_laplacian_from_knn.  A row block is
 * not accepted as the graph of as_compute_taumode (AS_ERR_UNSUPPORTED); as_compute_taumode_signals_sharded takes one per rank. */
int as_laplacian_from_knn_rows(as_ctx *ctx, const uint64_t *nbr_idx, const double *nbr_dist, uint64_t n_nodes,
                               const as_graph_params *params, uint64_t row_begin, uint64_t row_count, as_csr **out);

uint64_t as_csr_n(const as_csr *m);         /* rows */
uint64_t as_csr_cols(const as_csr *m);      /* columns (= rows unless the handle is a row block) */
uint64_t as_csr_row_begin(const as_csr *m); /* first global row of a row block, else 0 */
uint64_t as_csr_nnz(const as_csr *m);
/* copy out (host or device destinations): indptr[n+1], indices[nnz], data[nnz] */
int as_csr_export(as_ctx *ctx, const as_csr *m, uint64_t *indptr, uint64_t *indices, double *data);
/* wrap an existing CSR (e.g. one loaded by the host) */
int as_csr_import(as_ctx *ctx, uint64_t n, uint64_t nnz, const uint64_t *indptr,
                  const uint64_t *indices, const double *data, as_csr **out);
void as_csr_free(as_csr *m);

/* ---- row-sharded build, one process per GPU: what replaces build_laplacian_matrix
 * (src/laplacian.rs:122-178) when the items are spread over several GPUs.  The library runs the whole exchange
 * sequence through a communicator: all-gather of the row shards; then every rank prepares the SAME join (rotated
 * coordinates, row order, f16 / int8 images, thresholds: functions of the rows alone, every sum in a fixed order), runs the
 * int8 first pass and the masked f16 pass over the tile pairs of ITS range of image positions, and the candidate lists go to
 * the owners of their positions in compact form (all-to-all: counts, sizes, slots); every rank finishes the rows of its
 * positions exactly; all-gather of the neighbour tables (by position, put back in row order); all-reduce of nnz.
 *   - as_comm_create_rccl: RCCL over xGMI.  Rank 0 calls as_comm_unique_id and ships the 128 bytes to the other ranks
 *     by whatever means the host has; every rank then creates its communicator.  librccl is loaded at run time
 *     (AS_ERR_UNSUPPORTED when it is not there).  Collectives are issued on the context's stream.
 *   - as_comm_create_custom: the caller's own transport, three primitives over DEVICE buffers.  Byte counts and offsets
 *     are arrays of `world` entries indexed by peer; every rank calls them in the same order; a primitive must have
 *     completed, or be ordered on `stream`, when it returns.
 * as_shard_range: the rows of a rank, contiguous, shard starts multiples of 512 (the row super tile of the join); ranks
 * past the end own no rows and still take part in every step.
 * as_build_laplacian_sharded: rows_shard = this rank's rows ((hi - lo) x dim; host or device, a host shard is staged once); *out_block = its ROW
 * BLOCK of the Laplacian (see as_laplacian_from_knn_rows; NULL when the rank owns no rows), *nnz_total = nnz of the
 * whole matrix.  Stacking the blocks of ranks 0 .. world - 1 gives exactly the CSR of as_build_laplacian on all rows.
 * Every rank must make the call.  A rank that fails locally (an allocation, a kernel) returns its own status and its peers
 * return AS_ERR_PEER instead of waiting for it: every data collective of the build is preceded by a one-word all-reduce
 * of "has anyone failed" (option "sharded_agreement", default 1, the same value on every rank; 0 = none: a failing rank
 * then leaves its peers waiting).
 * A transport that fails INSIDE a collective is the transport's own error path.  No block is left behind on failure. */
typedef struct as_comm as_comm;
typedef struct as_comm_ops {
  void *user;
  /* every rank contributes send_bytes bytes; rank r's contribution lands at recv + recv_offsets[r] (recv_bytes[r] bytes) */
  int (*all_gather_v)(void *user, const void *send, uint64_t send_bytes, void *recv, const uint64_t *recv_bytes,
                      const uint64_t *recv_offsets, void *stream);
  /* send + send_offsets[p] (send_bytes[p] bytes) goes to peer p; peer p's part lands at recv + recv_offsets[p] */
  int (*all_to_all_v)(void *user, const void *send, const uint64_t *send_bytes, const uint64_t *send_offsets, void *recv,
                      const uint64_t *recv_bytes, const uint64_t *recv_offsets, void *stream);
  /* in place, element-wise sum over the ranks */
  int (*all_reduce_sum_u64)(void *user, uint64_t *buf, uint64_t count, void *stream);
} as_comm_ops;
int as_comm_unique_id(void *id128);
int as_comm_create_rccl(as_ctx *ctx, const void *id128, int rank, int world, as_comm **out);
int as_comm_create_custom(as_ctx *ctx, const as_comm_ops *ops, int rank, int world, as_comm **out);
void as_comm_destroy(as_comm *comm);
int as_comm_rank(const as_comm *comm);
int as_comm_world(const as_comm *comm);
void as_shard_range(uint64_t n, int rank, int world, uint64_t *lo, uint64_t *hi);
int as_build_laplacian_sharded(as_ctx *ctx, as_comm *comm, const double *rows_shard, uint64_t n_total, uint64_t dim,
                               const as_graph_params *params, as_csr **out_block, uint64_t *nnz_total);
/* The logged lambda statistics of a sharded build (src/builder.rs:434-445: min, max and mean over ALL items' lambdas):
 * lambdas_shard = this rank's n_here lambdas (host or device; n_here may be 0).  Every rank gets the same three values: the
 * shards' (min, max, sum, count) are all-gathered and combined in rank order, every partial sum in a fixed order. */
int as_lambda_stats_sharded(as_ctx *ctx, as_comm *comm, const double *lambdas_shard, uint64_t n_here, double *min_out,
                            double *max_out, double *mean_out);
/* ArrowSpace::search_lambda_aware (src/core.rs:760-798) over a row-sharded index: every rank searches its shard
 * (as_search_lambda_aware with id_offset = its first global row; index_shard NULL on a rank without rows), the per-shard
 * top-k lists are all-gathered and merged by (score desc, global index asc) -- what the reference's stable sort of all items
 * gives.  queries / q_lambdas are replicated (the same on every rank); idx_out / score_out (nq x k, host or device) hold the
 * same global result on every rank.  The reference panics for the whole query when one item's score is NaN or the query's
 * lambda is 0; here a rank whose local stage fails (those two, or an allocation) returns its own status and every other rank
 * returns AS_ERR_PEER -- the status agreement of the sharded build in front of the exchange ("sharded_agreement"; with 0 the
 * failing shard's lists are exchanged empty and only that rank reports the error).  as_lambda_stats_sharded above and
 * as_compute_taumode_signals_sharded below leave together in the same way. */
int as_search_lambda_aware_sharded(as_ctx *ctx, as_comm *comm, const as_index *index_shard, const double *queries,
                                   const double *q_lambdas, uint64_t nq, uint64_t k, double alpha, uint64_t id_offset,
                                   uint64_t *idx_out, double *score_out);

/* Item-graph lambdas over a row-sharded graph ("K7 item-graph signals"): items_shard = this rank's
 * item rows ((hi - lo) x n_features, as_shard_range; host or device), L_block = its ROW BLOCK of the Laplacian
 * (as_build_laplacian_sharded; NULL on a rank without rows).  The shards are all-gathered (a stored pair (i, j) reads row j
 * wherever it lives), every rank sums x'Lx, the edge energy and the sum of squares over the rows of its block (a pair once, on
 * the rank that owns its smaller row), the 4 partial sums per signal are all-gathered and added in rank order; a second round of
 * one sum per signal runs only where G needs the literal shares (src/taumode.rs:611-639).  tau comes from the whole columns
 * on every rank.  lambdas_out[n_features] (host or device): the same values on every rank, equal to as_compute_taumode_signals
 * on one GPU up to the order of the sums.  Every rank must make the call. */
int as_compute_taumode_signals_sharded(as_ctx *ctx, as_comm *comm, const double *items_shard, uint64_t n_total, uint64_t n_features,
                                       const as_csr *L_block, int tau_mode, double tau_arg, double *lambdas_out);

/* ---- lambda-tau: replaces TauMode::compute_taumode_lambdas_parallel, src/taumode.rs:174-312 */
/* items: n_items x n_features.  tau is taken from each item's own values (select_tau,
 * src/taumode.rs:87-127), lambda from compute_synthetic_lambda_csr (src/taumode.rs:552-660).
 * Requires as_csr_n(L) <= n_features. */
int as_compute_taumode(as_ctx *ctx, const double *items, uint64_t n_items, uint64_t n_features,
                       const as_csr *L, int tau_mode, double tau_arg, double *lambdas_out);

/* The same lambdas for SIGNALS OVER THE ITEM GRAPH (SURVEY.md section 8(a'), item-graph mapping): L is the n x n Laplacian of
 * build_laplacian_matrix on the untransposed items (rows are nodes, src/energymaps.rs:270), a signal is a feature COLUMN of the
 * n_items x n_features item matrix -- one value per node ("feature row over item nodes"), n <= n_items.  lambdas_out[f] =
 * compute_synthetic_lambda_csr(column f, L, select_tau(column f)) (src/taumode.rs:552-660, 87-127): what
 * as_compute_taumode returns for the transposed matrix (n_features items of n_items values each), without the transpose:
 * the CSR is streamed once for all signals and each stored pair fetches the two item ROWS (algorithmic bytes nnz * 16 +
 * n_items * n_features * 8).  as_compute_taumode itself takes this route for signals of "signal_min_len" (9000) values or more.
 * Options: "signal_sym" (1: a bitwise symmetric L is streamed as its upper triangle), "signal_order" (1: rows visited in the
 * order attached to L); they take effect when the graph's record stream is next built (as_csr_set_order drops it).
 * Median / Percentile tau of signals of 131072 values or more: the order statistics come from a sampled bracket and one pass over
 * the columns, on a second stream of the context under the graph pass; the call reads one word back (did every bracket hold?), i.e.
 * it waits for the work queued before it on the context's stream. */
int as_compute_taumode_signals(as_ctx *ctx, const double *items, uint64_t n_items, uint64_t n_features, const as_csr *L, int tau_mode,
                               double tau_arg, double *lambdas_out);
/* An order of the rows of L in which neighbours sit together (order[position] = row, a permutation of 0 .. n - 1, host or
 * device; NULL removes it): the item-graph lambda kernels visit the rows in it so that the rows an entry fetches are still in
 * cache.  Any permutation gives the same lambdas (to rounding of the sums); a host that knows its clustering (ArrowSpace's
 * cluster_assignments, src/core.rs:373-377) passes the rows sorted by cluster.  Graphs built by as_build_laplacian carry the
 * order their neighbour search found. */
int as_csr_set_order(as_ctx *ctx, as_csr *m, const uint32_t *order);

/* Batched ArrowSpace::prepare_query_item, src/core.rs:533-549 (projection off): fails with
 * AS_ERR_NONFINITE if any query value is NaN/inf. */
int as_prepare_query_items(as_ctx *ctx, const double *queries, uint64_t nq, uint64_t n_features,
                           const as_csr *L, int tau_mode, double tau_arg, double *lambdas_out);

/* TauMode::select_tau for a batch of rows (src/taumode.rs:87-127); taus_out[n_rows] */
int as_select_tau(as_ctx *ctx, const double *rows, uint64_t n_rows, uint64_t len, int tau_mode,
                  double tau_arg, double *taus_out);

/* ---- search: replaces ArrowSpace::search_lambda_aware, src/core.rs:760-798 ---------------- */
/* An index may be searched from several contexts / threads at once (the reference's search takes &self, src/core.rs:760, and
 * ArrowSpace is Sync): every searching thread brings its own as_ctx (scratch and stream are the context's), the handle is
 * shared.  What a search builds lazily and keeps with the index (ordered / rotated images from the second search on, samples)
 * is built under a writer lock inside the handle, with the building stream synchronised before readers go on; a search holds
 * the reader side.  as_index_destroy must not run while a search is in flight. */
int as_index_create(as_ctx *ctx, const double *items, const double *lambdas, uint64_t n_items,
                    uint64_t n_features, as_index **out);
void as_index_destroy(as_index *index);
uint64_t as_index_nitems(const as_index *index);
/* Batched: for each query the first min(k, n_items) items by (score desc, index asc) with
 * score = alpha*cos + (1-alpha)*(1 - min(|lambda_q - lambda_i|, 1)) (src/core.rs:135-175).
 * idx_out / score_out are nq x k (unused tail: idx = UINT64_MAX).  Fails with
 * AS_ERR_ZERO_LAMBDA if any q_lambdas[j] == 0.0 and with AS_ERR_NAN_SCORE on a NaN score.
 * id_offset is added to every returned index (row-sharded indexes pass their shard start). */
int as_search_lambda_aware(as_ctx *ctx, const as_index *index, const double *queries,
                           const double *q_lambdas, uint64_t nq, uint64_t k, double alpha,
                           uint64_t id_offset, uint64_t *idx_out, double *score_out);

/* Batched ArrowSpace::search_lambda_aware_hybrid, src/core.rs:802-928: per query the union of
 * (a) every item with cosine > 0.9999 (score = cosine), (b) the lambda-aware top-k (blended score
 * unless already in (a)), (c) the best-cosine item; sorted by score descending (ties: lower index),
 * truncated to k.  No zero-lambda check (the reference has none here).  k <= 4096. */
int as_search_lambda_aware_hybrid(as_ctx *ctx, const as_index *index, const double *queries,
                                  const double *q_lambdas, uint64_t nq, uint64_t k, double alpha,
                                  uint64_t id_offset, uint64_t *idx_out, double *score_out);

/* ArrowSpace::range_search with a prepared query lambda, src/core.rs:944-976: every item with
 * lambda_q - lambda_i <= eps, in index order, with that signed difference.  Writes the first
 * min(cap, found) hits and the total count to *n_found (host pointer). */
int as_range_search(as_ctx *ctx, const as_index *index, double lambda_q, double eps, uint64_t id_offset,
                    uint64_t *idx_out, double *dist_out, uint64_t cap, uint64_t *n_found);

#ifdef __cplusplus
}
#endif
#endif /* ARROWSPACE_HIP_H */
