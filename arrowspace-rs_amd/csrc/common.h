// common.h -- context, device-buffer staging and launch helpers shared by the kernels of
// libarrowspace_hip.so (gfx950 only; no CPU path).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <atomic>
#include <functional>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

#include "../../include/arrowspace_hip.h"

namespace asx {

constexpr uint32_t kNoIdx = 0xFFFFFFFFu;
constexpr double kTauFloor = 1e-10;  // src/taumode.rs:84

// ---- one word of device memory that many waves update ------------------------------------------------------------------------
// Read-modify-writes on ONE address are carried out at the memory side one after the other: ~90 per microsecond (measured, cone
// bound of round 4), whatever the rest of the kernel does -- 32 768 waves adding to one counter cost the pruning kernel 0.36 ms
// at any n, one atomicMax per wave of a packing pass 0.17 ms per million rows.
// A maximum only grows, so a wave LOOKS first (a coherent load; loads do not queue behind each other) and sends its value only
// where it would raise the word: a stale look costs one atomic, never a result.
__device__ __forceinline__ void raise_max_u32(unsigned int *word, unsigned int v) {
  if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) atomicMax(word, v);
}
__device__ __forceinline__ void raise_max_u64(unsigned long long *word, unsigned long long v) {
  if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) atomicMax(word, v);
}
__device__ __forceinline__ void lower_min_u64(unsigned long long *word, unsigned long long v) {
  if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > v) atomicMin(word, v);
}
// A sum has no such shortcut: the waves of a block add up in LDS first (every thread of the block must call this, once, at the
// end of the kernel), and the blocks spread over `slots` words (a power of two; 1: one word) that the reader adds up.
__device__ __forceinline__ void block_add_u64(unsigned long long *words, unsigned int slots, unsigned long long wave_total) {
  __shared__ unsigned long long s_block_total;
  if (threadIdx.x == 0) s_block_total = 0ull;
  __syncthreads();
  if ((threadIdx.x & 63) == 0 && wave_total) atomicAdd(&s_block_total, wave_total);
  __syncthreads();
  if (threadIdx.x == 0 && s_block_total) atomicAdd(words + (blockIdx.x & (slots - 1u)), s_block_total);
}

struct ProfEntry {
  double ms = 0.0;
  uint64_t launches = 0;
};

struct PendingEvent {
  std::string name;
  hipEvent_t start, stop;
};

}  // namespace asx

namespace asx {
struct DeferredStat;  // (below DevBuf)
}
struct as_ctx {
  int device = 0;
  const char *cur_scope = nullptr;  // innermost open ProfScope (profile on)
  hipStream_t stream = nullptr;
  hipStream_t side = nullptr;             // a second stream for work that runs under a pass of the main one (ctx_side_stream)
  hipEvent_t side_start = nullptr, side_done = nullptr;
  int *side_word = nullptr;               // pinned: what the side stream reads back
  bool own_stream = false;
  std::string err;
  void *pinned = nullptr;  // staging buffer for host <-> device transfers
  std::multimap<size_t, void *> free_blocks;  // workspace cache (capacity -> block)
  uint64_t ws_malloc_us = 0, ws_mallocs = 0;  // time this context spent in hipMalloc for workspace blocks (statistics)
  // Options (as_ctx_set_option).  include/arrowspace_hip.h documents the 35 SUPPORTED ones; the others here are switches for the tests
  // (forms of one stage that must agree bit for bit: lambda_v2 / lambda_v3 / lambda_wide, join_bf16, second_pass_f16, search_scan_f16,
  // assembly_lanes, nearest_mfma, cluster_on_device, join_theta_parts, join_unit_list, sharded_own_rows, search_stream_quick), failure
  // injection and capacity hooks (sharded_inject_failure, rerank_stream_cap), and sizes kept adjustable for measurements (cand_slack,
  // join_block, join_basis_iters, join_theta_rate, join_theta_grouped_bad_e3, join_cone_tiles_max_e3, join_theta_band_second,
  // search_scan_sample / _rank / _cap, search_rot_tail_e3, search_staged_min_batch, signal_min_len): accepted, but no interface.
  int64_t opt_knn_mfma = 1;       // 1: MFMA candidates + exact re-rank, 0: exact scan only
  int64_t opt_search_mfma = 1;    // same for the search scan
  int64_t opt_cand_slack = 8;     // extra candidates kept by the MFMA pass beyond topk+1
  int64_t opt_lambda_v2 = 1;      // 1: register sorting-network tau + 2-wave tile kernel when n_features <= 128
  int64_t opt_lambda_v3 = 1;      // 1: streaming 2-lanes-per-item kernel when n_features == 128 (taumode_v3.hip)
  int64_t opt_knn_join = 1;       // item graph over the whole matrix: symmetric threshold join (mfma_join_f32.hip)
  // Thresholds of the join: the m-th best product of every row with a 1-in-N sample of the rows, and the list slots per
  // row.  0 = automatic: 32 / 4 / 512 (also for the sharded sequence as_knn_join_*: its lists cross the links between
  // the GPUs in compact form, as_knn_join_pack).  1 M x 128 on one GPU, theta + join + second pass +
  // re-rank: 12/6/192 387 ms, 16/5/256 378, 24/4/384 374, 32/4/512 370, 40/4/640 395 (the second pass starts to
  // overflow), 48/4/768 395; a sparser sample costs candidates (74 -> 132 per row), which are cheap now that the
  // re-rank prunes by the stored scores before it fetches anything.
  int64_t opt_join_sample = 0;
  int64_t opt_join_rank = 0;
  int64_t opt_join_cap = 0;
  int64_t opt_join_min_rows = 32768;
  int64_t opt_join_block = 64;    // column super tiles per work unit of the join passes, at most 64 (the masked pass lists a unit's <= 128 stages in LDS); 1 M x 128: 32 -> 45.0 ms, 64 -> 44.4, 16 -> 46.1, 8 -> 48.2 (r03n); the dense bf16 join of round 1: 4 -> 304 ms, 8 .. 64 ->
                                  // 291-293: longer units amortise the unit prologue; devices differ by more than that)
  int64_t opt_join_pipe = 16;     // contraction of the join: 16 = one f16 plane (dim <= 128; mfma_join_f16.hip), 48 = split bf16 in
                                  // two phases (dim <= 128; mfma_join_bf16.hip), 32 = f32 MFMA
  int64_t opt_search_scan = 1;    // search: threshold scan on the bf16 pipe before the top-k passes (dim <= 128, alpha > 0)
  int64_t opt_search_scan_sample = 32;   // its score guess: the m-th best of a 1-in-N sample of the items
  int64_t opt_search_scan_rank = 3;
  int64_t opt_search_scan_cap = 512;     // candidate slots per query
  int64_t opt_join_bf16 = 1;      // thresholds from bf16 products (any threshold is correct; bf16 is 16x cheaper)
  int64_t opt_stream_scan = 1;    // small batches / leftover rows: one exact pass over the matrix (stream_scan.hip)
  int64_t opt_search_scan_f16 = 1; // threshold scan of the search (items up to 128 wide) on one f16 plane; 0: split bf16
  int64_t opt_search_scan_order = 1;  // search scan, dim <= 128: items (from 65,536) and large query batches grouped by direction:
                                      // 1 = from the second scan of an index on, 2 = at once, 0 = never
  int64_t opt_search_lambda_order = 1;  // search scan on items in their own order: batches of 1024 queries or more go in the order of their lambdas
  int64_t opt_search_prefilter = 1;   // ... and the int8 first pass in front of the scan
  int64_t opt_assembly_lanes = 2;        // graph assembly (K <= 32): the lane-parallel kernels (graph_assembly.hip); 0 = a thread per row (what deeper tables take)
  int64_t opt_search_rot_tail_e3 = 100;  // a re-searched index keeps rotated int8 images when the tail of its items is at most this (x 1e-3)
  int64_t opt_sharded_agreement = 1;     // sharded build: a one-word all-reduce of "has anyone failed" in front of every data collective (sharded.hip)
  int64_t opt_sharded_inject_failure = 0;  // test hook: this rank fails at checkpoint 1 (before the row all-gather), 2 (before the list exchange), 3 (before the nnz all-reduce); 4 statistics, 5 search; item-graph lambda: 6 (first stage), 7 (after the sums exchange), 8 (its rank-local argument checks)
  int64_t opt_join_basis_iters = 4;      // rotated coordinates: rounds of subspace iteration on the 16 k-row sample (each 0.15 ms: three small kernels, two copies, a host Gram-Schmidt; hard mixture 98.2 / 94.7 / 93.2 / 93.9 ms at 2 / 3 / 4 / 5 rounds, the bench data indifferent from 3 on)
  int64_t opt_join_order_sub = 4;        // row order of the join: inside a pivot's group, by the nearest pivot of a second set of np / this pivots (0 = off; 0.45 ms per million rows; pays wherever a group holds several clusters)
  int64_t opt_join_cone_tiles = 1;       // rotated first pass on an ordered image: behind a cone bound on pairs of 128-position tiles (cone_tiles.hip)
  int64_t opt_join_cone_tiles_max_e3 = 600;  // ... taken while it keeps at most this share (x 1e-3) of the tile pairs; beyond, the pass sweeps everything
  int64_t opt_join_unit_list = 1;        // masked f16 pass: the units that hold a flagged stage listed first (0: every unit drawn and looked at)
  int64_t opt_join_cone_second = 1;      // second pass of the join (f16 plane, 128 columns): behind the one-sided cone bound, the open rows in image order
  int64_t opt_join_theta_band_slack = 4; // thresholds of the own neighbourhood: taken instead of a sampled threshold that is higher by at most this many deltas (certified by construction)
  int64_t opt_join_order_cover = 150;    // row order of the join: where more than this many per mille of the rows lie far from their pivot (cosine below 0.8 of the 90 % point), further pivots are drawn from those rows, 1 024 a round up to 4 096 (0 = never; row_order.hip)
  int64_t opt_join_theta_dip_e3 = 50;    // ... and of the rows whose threshold lies this far (per mille) below the median of their 64-position block (0 = off)
  int64_t opt_join_theta_far = 1;        // thresholds of the rows far from their pivot (row_order.hip) from the whole sample, not their group's (0 = off)
  int64_t opt_debug_label_mod = 0;       // diagnostic: rows carry the label (row mod this); statistics of the labels per image tile (row_order.hip)
  int64_t opt_join_cone_split_above_e3 = 80;  // cone bound on tile pairs: where it keeps more than this many per mille it is taken again with two cones per tile, split where a pivot group ends (cone_tiles.hip; -1 = never)
  int64_t opt_join_cone_groups = 1;      // cone bound on tile pairs: a coarse level first, every tile's rows against the cones around 32 tiles (from 2048 tiles on; 0: one level)
  int64_t opt_join_open_order = 1;       // the rows the first re-rank leaves open go on in the order of the join's image (0: as they were appended)
  int64_t opt_rerank_image_order = 1;    // re-rank of rows wider than 128: records in the order of the join's image (0: in row order)
  int64_t opt_keep_open_rows = 0;  // diagnostic: the ids of the rows the first re-rank leaves open are kept as stats (mfma_drivers.hip)
  int64_t opt_join_basis_sample = 16384; // rotated coordinates: rows of the strided sample the directions are found on
  int64_t opt_join_basis_fused = 1;    // rows of 128 values: a round of the basis' subspace iteration in one kernel (0: coordinates, then Y = S^T Z, two kernels)
  int64_t opt_keep_join_images = 0;    // an item-graph build keeps its order and images for an index of the same rows (as_ctx::JoinKeep; ~300 MB per million rows)
  int64_t opt_keep_join_images_min_rows = 262144;  // ... from this many rows
  int64_t opt_index_adopt_join_images = 1;  // as_index_create takes them over when the items are those rows (same device pointer, shape and norms)
  int64_t opt_join_second_slices = 1;  // the second pass of an item graph (its open rows in the order of the image) on the scan by slices too: 0.70 -> 0.5 ms at 1 M rows (0: a workgroup per 512 rows)
  int64_t opt_search_scan_run = 1;     // ... units a wave draws at a time (consecutive ones: of one slice)
  int64_t opt_search_scan_slices = 2;  // the one-sided f16 scan behind a first pass: a wave per (64 queries, this many stages of 256 columns), nothing shared (0: a workgroup per 512 queries, stages through LDS)
  int64_t opt_join_masked_block = 0;  // column super tiles per work unit of the masked f16 pass alone (0: "join_block")
  int64_t opt_rerank_prune_halves = 1;  // the re-rank's pruning: a row per half wave with its list in registers where the list has at most 128 entries (0: every row through LDS, a wave each)
  int64_t opt_sharded_split_assembly = 1;  // a build sharded by rows: degrees, weights and the pruning of a row on its owner, the adjacency exchanged (0: the tables exchanged, every rank ranks all rows)
  int64_t opt_sharded_order_assign = 1;  // a build sharded by rows: the row order's pivot assignment on a rank's own rows, the assignments all-gathered (0: every rank sweeps all rows)
  int64_t opt_join_one_sided_block = 2;  // column super tiles per unit of the one-sided f16 pass behind a mask or on an ordered image (second pass, search scan)
  int64_t opt_join_theta_rows64 = 1;     // threshold sweeps: four waves of 64 rows per 256-row unit (two MFMAs per fragment read; mfma_theta_f16.hip), 0: eight waves of 32
  int64_t opt_join_fused_rows_pass = 1;  // rows of exactly 128 values: rotated coordinates and f16 image from ONE pass over the rows (rotate_wide.hip; 0: two passes)
  int64_t opt_join_pack_wide_tiles = 1;  // f16 image of rows of a multiple of 128 columns in one coalesced pass (mfma_join_f16.hip; 0: per fragment lane + a second read for the error)
  int64_t opt_join_zproj_mfma = 1;       // rotated coordinates of rows of 256 .. 1024 columns on the matrix pipe (rotate_wide.hip; 0: a wave per row, shuffle reductions)
  int64_t opt_join_second_prefilter = 1; // rows wider than 128: the rectangular int8 first pass in front of the one-sided second pass of the join (0 = off)
  int64_t opt_join_theta_band_auto = 1;  // a band of 2 (the default) grows to 3 or 4 super tiles on either side where the clusters found by the row order hold more rows than it covers (0: as set)
  int64_t opt_join_order_chain = 1;      // row order of the join: the pivot groups laid out along a greedy chain of their pivots (1), the second pivot set too (2); 0: in the order the pivots were drawn
  int64_t opt_rerank_share = 1;          // exact re-rank of the join (rerank_pairs.h): a pair two rows both keep is scored once; 0: every row scores all of its survivors
  int64_t opt_rerank_stream_cap = 0;     // ... test hook: batches the slot stream gets room for at the first attempt (0: three per row)
  int64_t opt_nearest_mfma = 1;          // nearest-centroid search of a batch: candidates from the f32 matrix pipe + exact re-rank (2: also behind as_nearest_centroids)
  int64_t opt_cluster_on_device = 1;     // incremental clustering: batches after the last centroid is created stay on the device
  int64_t opt_search_stream_quick = 1;   // up to 8 rows without an int8 image: quick f64 pass over the items + exact chains of what it leaves
  int64_t opt_search_stream_filter = 1;  // small batches on an index searched before: int8 filter in front of the streaming scan
  int64_t opt_search_staged_min_batch = 1 << 30;  // batches from this size take the scan with the first pass instead of streaming all
                                                  // rows (measured: 1.5 ms of fixed cost against 0.46 - 1.0 ms of streaming: off)
  int64_t opt_rerank_pair_bound = 1;       // the pruning in front of the exact re-rank bounds each pair by its two rows' own plane errors
  int64_t opt_join_theta_parts = 1;        // threshold passes with few rows split the sample over several units (0: one sweep per 256 rows)
  int64_t opt_join_theta_band_second = 1;  // the second pass never cuts below what the band certified
  int64_t opt_join_theta_band = 2;       // thresholds also from a row's own neighbourhood in the image: this many 512-position super tiles on either side (0: off)
  int64_t opt_join_theta_rate = 32;      // grouped thresholds: one image position in this many is sampled (a power of two, 2 .. 32)
  int64_t opt_join_theta_grouped_bad_e3 = 10;  // ... kept while at most this share (x 1e-3, times min(4, n / 1 M)) of 4 096 checked rows sit 5e-3 below the plain sample's
                                              // (measured at 1 M x 128: 0.7 % -> 5.5 ms better than the plain pass, 2.9 % -> 7 ms worse; break-even ~1.4 %)
  int64_t opt_join_theta_grouped = 1;    // own join ordered by pivots of the rotated coordinates: thresholds against the sample of the nearest
                                         // groups (8.2 -> 1.7 ms at 1 M x 128), checked against the whole sample on 4096 rows
  int64_t opt_join_order_pivots_auto = 1;  // ... 2048 where the default of 1024 stands and the matrix has more than 3 M rows (0: as set)
  int64_t opt_join_order_pivots = 1024;  // pivots of the row order found on the rotated coordinates (256 .. 4096; beyond 1024 a pivot may stand for 240 rows)
  int64_t opt_join_prefilter = 1;  // int8 first pass in front of the f16 join (dim <= 128; mfma_prefilter_i8.hip): 1 = on, the
                                   // rotated 32-wide image where it pays, 2 = full width always, 3 = rotated always, 0 = off
  int64_t opt_join_prefilter_min_rows = 65536;
  int64_t opt_join_row_order = 1;  // own join, dim <= 128: operand image packed with the rows grouped by direction (row_order.hip)
  int64_t opt_second_pass_f16 = 1; // second pass over the rows the f16 join leaves open on the f16 plane too; 0: split bf16
  int64_t opt_lambda_wide = 1;    // lambda-tau for items wider than 128 features: one wave per item (taumode_wide.hip); 0: the generic kernel
  int64_t opt_sharded_own_rows = 1;      // build sharded by rows: the join is prepared from the ranks' own rows and the all-gather of the f64 rows runs under it (0: rows first)
  int64_t opt_signal_sym = 1;            // item-graph lambda (signal_lambda.hip): a bitwise symmetric graph is streamed as its upper triangle
  int64_t opt_signal_order = 1;          // ... in the order attached to the graph (as_csr_set_order / the join's own); 0: the rows as stored
  int64_t opt_signal_tau_bracket = 1;    // Median / Percentile tau of columns of 131 072 values or more from a sampled bracket and ONE pass over the column (0: the digit passes, three reads)
  int64_t opt_signal_tau_overlap = 1;    // ... on a side stream under the sums pass of the item-graph lambda (0: in front of it)
  int64_t opt_signal_min_len = 9000;     // as_compute_taumode: signals this long or longer go through the item-graph kernels
  int64_t opt_debug = 0;          // diagnostics only (timing experiments); 0 in production
  int64_t opt_cand_precision = 32;  // first candidate stage: 32 = f32 MFMA (then f64), 64 = f64 MFMA only
  // Sharded join sequence (as_knn_join_thresholds -> _pairs -> _finish over the same `rows`): norms and the packed
  // f32 image of all rows, built by the first call, reused by the other two, released by the last.
  struct JoinSession {
    const double *rows = nullptr;
    uint64_t n = 0, d = 0;
    double *norms = nullptr;
    float *packed = nullptr;
    void *packed16 = nullptr;   // split-bf16 image of all rows (bf16 pipe of the join; the one-sided second pass)
    void *packedh = nullptr;    // f16 image of all rows (f16 pipe of the join)
    float *theta2 = nullptr;    // second (looser) thresholds of the rows of the last as_knn_join_thresholds call
    uint64_t theta2_begin = 0, theta2_rows = 0;
    bool theta2_valid = false;
    size_t norms_cap = 0, packed_cap = 0, packed16_cap = 0, packedh_cap = 0, theta2_cap = 0;
  } join;
  double f16_err = -1.0;  // measured error of the f16 plane of the rows packed last (pack_f16_all), -1: none
  const void *rowmajor_image = nullptr;  // while own_join_prepare runs: the buffer that holds the rows' f16 form ROW-MAJOR (pack_f16_from_image reads it as such)
  // the row order the last whole-matrix join found (position -> row; row_order.hip): as_build_laplacian attaches a copy to the
  // graph it returns, where the item-graph lambda kernels visit the rows in it (signal_lambda.hip)
  uint32_t *join_order = nullptr;
  uint64_t join_order_n = 0, join_order_cap = 0, join_order_seq = 0;
  // "keep_join_images": what the last item-graph build (one GPU, rows of 128 values, the join on an image in its own order) leaves
  // for an index of the SAME rows -- the order, the f16 image in it, the pivots of the order, the 31 directions, the rotated int8
  // image by position, the norms (to tell the rows apart from others) -- so that the index's first search runs on the ordered
  // image behind the int8 first pass instead of finding all of that again (as_index_create adopts them: "index_adopt_join_images")
  struct JoinKeep {
    const void *rows = nullptr;
    uint64_t n = 0, d = 0;
    void *packedh = nullptr, *packed8 = nullptr;
    uint32_t *perm = nullptr;
    float *piv = nullptr, *rot_qw = nullptr;
    double *norms = nullptr;
    uint32_t np = 0;
    double rot_err = -1.0, f16_err = -1.0;
    bool complete = false;
  } join_keep;
  void *keep_p8_dst = nullptr;   // where the first pass of the build in progress leaves a copy of its int8 image (or null)
  double keep_p8_err = -1.0;
  bool keep_p8_done = false;
  // stats of the last call
  std::map<std::string, int64_t> stats;
  // counters of kernels that nobody decides anything by (blocks contracted / flagged / listed): read back with the NEXT read-back
  // that has to wait for the stream anyway instead of a round trip of their own (deferred_stats_*, below DevBuf)
  std::vector<std::unique_ptr<asx::DeferredStat>> deferred;
  // profiling
  bool prof_on = false;
  std::map<std::string, asx::ProfEntry> prof;
  std::vector<asx::PendingEvent> pending;
};

struct as_csr {
  int device = 0;
  uint64_t n = 0, nnz = 0;      // rows, stored entries
  uint64_t n_cols = 0;          // columns (= n unless the handle is a row block of a sharded graph: then the node count)
  uint64_t row_begin = 0;       // first global row of a row block (0 for a whole matrix)
  uint64_t *indptr = nullptr;   // n+1 (device)
  uint64_t *indices = nullptr;  // nnz (device)
  double *data = nullptr;       // nnz (device)
  void *slab = nullptr;         // when set: ONE allocation that holds indptr, indices and data (graph assembly: a hipMalloc / hipFree costs ~0.2 ms each)
  // packed (row, col, value) entries for the lambda kernel, built on first use
  void *packed = nullptr;
  // upper-triangle stream (j >= i), present only when the matrix is bitwise symmetric
  void *packed_upper = nullptr;
  uint64_t nnz_upper = 0;
  // entry streams of the n_features == 128 streaming kernel (taumode_v3.hip), built on first use
  int v3_state = 0;  // 0: not prepared, 1: ready, -1: outside that kernel's domain
  void *v3_ent = nullptr, *v3_diag = nullptr;
  uint32_t v3_steps = 0;
  double v3_scale = 1.0;
  // record stream of the item-graph lambda kernels (signal_lambda.hip), built on first use
  int sig_state = 0;            // 0: not prepared, 1: ready
  void *sig_rec = nullptr;      // 16-byte records: per visited row a header and its kept off-diagonal entries
  uint64_t *sig_off = nullptr;  // record offsets by stream slot (slots + 1)
  uint64_t sig_nrec = 0;
  uint32_t sig_ch = 0, sig_wx = 0, sig_k = 0, sig_c8 = 0;  // its geometry: rows per chunk, waves per XCD group, chunks per wave, chunks per group
  int sig_sym = 0;              // the stream holds the upper triangle only (bitwise symmetric matrix)
  int known_symmetric = 0;      // symmetric by construction (the library's own graphs and their row blocks)
  uint32_t *order = nullptr;    // position -> row: an order of the rows in which neighbours sit together (device), or nullptr
  int order_state = 0;          // 0: none, 1: given by the caller or the build
};

struct as_index {
  // Concurrency (the reference's search takes &self, src/core.rs:760, and ArrowSpace is Sync): any number of contexts / threads may
  // search one index at once.  A search holds `rw` shared while it reads the state below; whatever is built lazily (images,
  // orders, samples) is built under `rw` held exclusively, by whichever search finds it missing, and the building stream is
  // synchronised before the lock is released -- so a reader never sees a half-built image, and an image is never freed under one.
  mutable std::shared_mutex rw;
  int device = 0;
  uint64_t n = 0, d = 0;
  const double *items = nullptr;    // n x d (device); owned iff own_items
  const double *lambdas = nullptr;  // n (device); owned iff own_lambdas
  bool own_items = false, own_lambdas = false;
  double *norms = nullptr;          // n exact norms
  double *packed = nullptr;         // fragment-packed unit rows, f64 image (lazy)
  float *packed32 = nullptr;        // fragment-packed unit rows, f32 image (lazy)
  void *packed16 = nullptr;         // split-bf16 image (lazy; the threshold scan of the search)
  uint32_t *scan_sample = nullptr;  // sampled item ids of the threshold scan and their lambdas (lazy)
  float *scan_sample_lam = nullptr;
  float *lam32 = nullptr;           // f32 lambdas of all items, padded to whole tiles (lazy)
  void *packedh = nullptr;          // f16 image of the items (lazy; the threshold scan of the search for items wider than 128)
  double f16_err = -1.0;            // measured plane error of that image (mfma_join_f16.hip)
  // items up to 128 wide, 65,536 of them or more: the f16 image (and lam32) are kept in an order that groups the items by
  // direction (row_order.hip), with the pivots for ordering a batch of queries the same way, and a rotated int8 image for
  // the first pass of the scan (mfma_prefilter_i8.hip)
  void *packed8_nat = nullptr;      // rotated int8 image in the order of the items (the filter of the streaming scan)
  double rot_err_nat = -1.0;
  int weird_items = -1;             // items with a non-finite norm or lambda (-1: not counted yet): the filter is not used then
  uint32_t *scan_perm = nullptr;    // image position -> item (nullptr: the items as given)
  float *lam32_pos = nullptr;       // f32 lambdas by image position, padded
  float *scan_piv = nullptr;        // 256 x 128 pivots
  void *packed8 = nullptr;          // rotated int8 image, by image position
  float *rot_q = nullptr;           // its 31 directions (32 x 128)
  float *rot_qw = nullptr;          // items wider than 128: the 31 directions of rotate_wide.hip (31 x rotated_wide_pitch(d)), kept to
                                    // project a batch of queries the same way (the ordered images of wide items, round 5)
  double rot_err = -1.0;            // its largest quantisation error norm
  int rot_state = 0;                // 0 not tried, 1 usable, -1 the directions do not carry the items
  // adopted from the item-graph build of the same rows (as_ctx::JoinKeep): scan_perm / packedh / packed8 are that build's, the
  // directions are rot_qw (31 x 128, rotate_wide.hip) and scan_piv holds scan_np pivots of 32 rotated coordinates -- a batch of queries
  // is projected, ordered and packed the way the batches of wide items are
  bool z_space = false;
  uint32_t scan_np = 256;
  int nat_state = 0;                // z_space: the directions of the filter of the streaming scan (rot_q), 0 not tried, 1 usable, -1 not
  std::atomic<uint32_t> scan_calls{0};  // searches run on this index so far
  uint32_t scan_ns = 0;
};

namespace asx {

inline int fail(as_ctx *ctx, int code, const std::string &msg) {
  if (ctx) ctx->err = msg;
  return code;
}

#define AS_HIP(ctx, expr)                                                                  \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      return asx::fail(ctx, AS_ERR_HIP,                                                    \
                       std::string(#expr) + ": " + hipGetErrorString(_e) + " @" + __FILE__ + \
                           ":" + std::to_string(__LINE__));                                \
    }                                                                                      \
  } while (0)

#define AS_TRY(expr)            \
  do {                          \
    int _rc = (expr);           \
    if (_rc != AS_OK) return _rc; \
  } while (0)

inline bool is_device_ptr(const void *p) {
  if (!p) return false;
  hipPointerAttribute_t a;
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // clear: plain host memory is reported as an error
    return false;
  }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// Transfers with a HOST endpoint go through a pinned staging buffer owned by the context, as
// genuine stream operations (hipMemcpyAsync on pinned memory).  hipMemcpy / hipMemcpyAsync straight
// from pageable memory is staged by the runtime outside the context's stream order: kernels queued
// behind such a copy were observed reading the destination (recycled pool memory) before the bytes
// were visible.  Device-resident callers never come through here.
constexpr size_t kPinnedBytes = 64u << 20;
inline hipError_t ensure_pinned(as_ctx *ctx) {
  if (ctx->pinned) return hipSuccess;
  return hipHostMalloc(&ctx->pinned, kPinnedBytes, hipHostMallocDefault);
}
// Large transfers (the item matrix of a host caller: 1 GB at 1 M x 128) are pipelined through the staging buffer cut into four
// slots of 16 MB: a few host threads copy chunk c + 1 into its slot while the DMA of chunk c is in flight, an event per slot
// says when it may be overwritten.  One thread and one slot (the form before) moved 1 GB in 52 ms: the host memcpy and the
// DMA took turns.  (api.hip: the worker pool.)
constexpr size_t kStageSlot = 16u << 20;
constexpr int kStageSlots = 4;
static_assert(kStageSlot * kStageSlots == kPinnedBytes, "the slots tile the staging buffer");
void host_copy_parallel(void *dst, const void *src, size_t bytes);  // memcpy split over the library's copy threads
inline hipError_t copy_h2d(as_ctx *ctx, void *dst, const void *src, size_t bytes) {
  hipError_t e = ensure_pinned(ctx);
  if (bytes <= kStageSlot) {
    if (e == hipSuccess && bytes) {
      std::memcpy(ctx->pinned, src, bytes);
      e = hipMemcpyAsync(dst, ctx->pinned, bytes, hipMemcpyHostToDevice, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // the staging buffer is reused
    }
    return e;
  }
  hipEvent_t done[kStageSlots] = {};
  for (int k = 0; k < kStageSlots && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&done[k], hipEventDisableTiming);
  size_t c = 0;
  for (size_t off = 0; e == hipSuccess && off < bytes; off += kStageSlot, ++c) {
    const size_t n = std::min(kStageSlot, bytes - off);
    const int k = (int)(c % kStageSlots);
    char *slot = (char *)ctx->pinned + (size_t)k * kStageSlot;
    if (c >= (size_t)kStageSlots) e = hipEventSynchronize(done[k]);  // the DMA that read this slot last
    if (e != hipSuccess) break;
    host_copy_parallel(slot, (const char *)src + off, n);
    e = hipMemcpyAsync((char *)dst + off, slot, n, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipEventRecord(done[k], ctx->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  else (void)hipStreamSynchronize(ctx->stream);
  for (int k = 0; k < kStageSlots; ++k)
    if (done[k]) (void)hipEventDestroy(done[k]);
  return e;
}
inline hipError_t copy_d2h(as_ctx *ctx, void *dst, const void *src, size_t bytes) {
  // (with the profile on: how many times the host waited for the stream to read something back, by the stage it was in)
  if (ctx->prof_on) ++ctx->stats[std::string("host_reads_in_") + (ctx->cur_scope ? ctx->cur_scope : "no_stage")];
  hipError_t e = ensure_pinned(ctx);
  if (bytes <= kStageSlot) {
    if (e == hipSuccess && bytes) {
      // (with the profile on: the longest the host spent in the two calls of a read-back -- stats "host_max_us_readback_copy" /
      // "..._wait": a wait far longer than the stage's GPU time is the host being held up, not the GPU)
      const auto t0 = std::chrono::steady_clock::now();
      e = hipMemcpyAsync(ctx->pinned, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
      const auto t1 = std::chrono::steady_clock::now();
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (ctx->prof_on) {
        const auto t2 = std::chrono::steady_clock::now();
        int64_t &mc = ctx->stats["host_max_us_readback_copy"], &mw = ctx->stats["host_max_us_readback_wait"];
        mc = std::max<int64_t>(mc, std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count());
        mw = std::max<int64_t>(mw, std::chrono::duration_cast<std::chrono::microseconds>(t2 - t1).count());
      }
      if (e == hipSuccess) std::memcpy(dst, ctx->pinned, bytes);
    }
    return e;
  }
  // the same pipeline the other way: the DMA of chunk c + 1 .. c + 3 runs while chunk c leaves its slot
  hipEvent_t done[kStageSlots] = {};
  for (int k = 0; k < kStageSlots && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&done[k], hipEventDisableTiming);
  const size_t chunks = (bytes + kStageSlot - 1) / kStageSlot;
  auto issue = [&](size_t c) -> hipError_t {
    const size_t off = c * kStageSlot, n = std::min(kStageSlot, bytes - off);
    const int k = (int)(c % kStageSlots);
    hipError_t r = hipMemcpyAsync((char *)ctx->pinned + (size_t)k * kStageSlot, (const char *)src + off, n, hipMemcpyDeviceToHost, ctx->stream);
    return r == hipSuccess ? hipEventRecord(done[k], ctx->stream) : r;
  };
  for (size_t c = 0; c < std::min<size_t>(chunks, kStageSlots) && e == hipSuccess; ++c) e = issue(c);
  for (size_t c = 0; c < chunks && e == hipSuccess; ++c) {
    const size_t off = c * kStageSlot, n = std::min(kStageSlot, bytes - off);
    const int k = (int)(c % kStageSlots);
    e = hipEventSynchronize(done[k]);
    if (e != hipSuccess) break;
    host_copy_parallel((char *)dst + off, (const char *)ctx->pinned + (size_t)k * kStageSlot, n);
    if (c + kStageSlots < chunks) e = issue(c + kStageSlots);
  }
  (void)hipStreamSynchronize(ctx->stream);
  for (int k = 0; k < kStageSlots; ++k)
    if (done[k]) (void)hipEventDestroy(done[k]);
  return e;
}

// Workspace allocations.  The context keeps a cache of hipMalloc'd blocks; a DevBuf takes a block
// for the duration of a call and hands it back.  All work of a context runs on ONE stream, so a
// block handed back while kernels that use it are still queued can only be reused by later work on
// the same stream, i.e. after those kernels.  (The runtime's stream-ordered pool, hipMallocAsync,
// was dropped: mixed with the blocking hipMalloc/hipFree of the long-lived handles it produced
// intermittent stale reads on this stack.)
void handle_cache_release();  // (below: the cache of handle memory) every idle block back to the device
inline int ws_alloc(as_ctx *ctx, size_t bytes, void **out, size_t *cap) {
  if (bytes == 0) bytes = 1;
  bytes = (bytes + 255) & ~(size_t)255;
  auto it = ctx->free_blocks.lower_bound(bytes);
  if (it != ctx->free_blocks.end() && it->first <= bytes + bytes / 2 + 4096) {
    *out = it->second;
    *cap = it->first;
    ctx->free_blocks.erase(it);
    return AS_OK;
  }
  const auto t0_ws = std::chrono::steady_clock::now();
  hipError_t e = hipMalloc(out, bytes);
  ctx->ws_malloc_us += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0_ws).count();
  ++ctx->ws_mallocs;
  if (e != hipSuccess) {  // drop the cache and retry once
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->free_blocks) (void)hipFree(kv.second);
    ctx->free_blocks.clear();
    handle_cache_release();  // ... and the blocks destroyed handles left behind
    e = hipMalloc(out, bytes);
  }
  if (e != hipSuccess) {
    ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e);
    return AS_ERR_HIP;
  }
  *cap = bytes;
  return AS_OK;
}
inline void ws_free(as_ctx *ctx, void *p, size_t cap) { ctx->free_blocks.emplace(cap, p); }

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t count = 0, cap = 0;
  as_ctx *owner = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  int alloc(as_ctx *ctx, size_t n) {
    release();
    owner = ctx;
    count = n;
    void *q = nullptr;
    AS_TRY(ws_alloc(ctx, n * sizeof(T), &q, &cap));
    p = (T *)q;
    return AS_OK;
  }
  void release() {
    if (p) ws_free(owner, p, cap);
    p = nullptr;
  }
};

// ---- statistics read back later (round 6) ------------------------------------------------------------------------------------------
// A kernel's counters that are statistics only used to cost a copy and a stream synchronisation of their own (~25 us with the GPU
// idle, 22 of them in a build).  The buffer is parked here with the stats it feeds; whoever next has to wait for the stream
// (read_u32 in the drivers, as_ctx_get_stat) issues the copies in front of its own wait and applies them behind it.
struct DeferredStat {
  DevBuf<unsigned long long> buf;
  std::vector<unsigned long long> host;
  std::vector<std::pair<std::string, int>> keys;  // stats[name] = host[index]
};
inline int deferred_stats_issue(as_ctx *ctx) {
  for (auto &d : ctx->deferred)
    if (hipMemcpyAsync(d->host.data(), d->buf.p, d->host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return AS_ERR_HIP;
  return AS_OK;
}
inline void deferred_stats_apply(as_ctx *ctx) {  // (behind a synchronisation of the stream that followed deferred_stats_issue)
  for (auto &d : ctx->deferred)
    for (auto &k : d->keys) ctx->stats[k.first] = (int64_t)d->host[(size_t)k.second];
  ctx->deferred.clear();
}
inline int deferred_stats_flush(as_ctx *ctx) {
  if (ctx->deferred.empty()) return AS_OK;
  AS_TRY(deferred_stats_issue(ctx));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  deferred_stats_apply(ctx);
  return AS_OK;
}

// Input that may live on the host: staged to HBM when needed.
template <typename T>
struct DevIn {
  const T *p = nullptr;
  DevBuf<T> tmp;
  int bind(as_ctx *ctx, const T *src, size_t n) {
    if (n == 0 || is_device_ptr(src)) {
      p = src;
      return AS_OK;
    }
    AS_TRY(tmp.alloc(ctx, n));
    AS_HIP(ctx, copy_h2d(ctx, tmp.p, src, n * sizeof(T)));
    p = tmp.p;
    return AS_OK;
  }
};

// Output that may live on the host: computed in HBM, copied back by finish().
template <typename T>
struct DevOut {
  T *p = nullptr;
  T *host = nullptr;
  size_t n = 0;
  DevBuf<T> tmp;
  int bind(as_ctx *ctx, T *dst, size_t count) {
    n = count;
    if (count == 0 || is_device_ptr(dst)) {
      p = dst;
      return AS_OK;
    }
    host = dst;
    AS_TRY(tmp.alloc(ctx, count));
    p = tmp.p;
    return AS_OK;
  }
  // Host destination: D2H + stream sync so the buffer is valid on return.  Device destination:
  // synchronous too while the context runs on its own private stream; once the caller has
  // installed a stream (as_ctx_set_stream) results are ordered on that stream instead.
  int finish(as_ctx *ctx) {
    if (host && n) {
      AS_HIP(ctx, copy_d2h(ctx, host, p, n * sizeof(T)));
    } else if (ctx->own_stream) {
      AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return AS_OK;
  }
};

// Scoped kernel-group timer: hipEvents on the context's stream, resolved lazily.
struct ProfScope {
  as_ctx *ctx;
  PendingEvent ev;
  bool active;
  const char *outer = nullptr;
  std::chrono::steady_clock::time_point host_t0;  // (with the profile on: the HOST's time inside the stage too, stat "host_us_in_<stage>")
  ProfScope(as_ctx *c, const char *name) : ctx(c), active(c->prof_on) {
    if (!active) return;
    host_t0 = std::chrono::steady_clock::now();
    ev.name = name;
    outer = ctx->cur_scope;
    ctx->cur_scope = name;
    if (hipEventCreate(&ev.start) != hipSuccess || hipEventCreate(&ev.stop) != hipSuccess) {
      active = false;
      return;
    }
    (void)hipEventRecord(ev.start, ctx->stream);
  }
  ~ProfScope() {
    if (!active) return;
    ctx->stats[std::string("host_us_in_") + ev.name] +=
        (int64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - host_t0).count();
    ctx->cur_scope = outer;
    (void)hipEventRecord(ev.stop, ctx->stream);
    ctx->pending.push_back(ev);
  }
};

inline void prof_resolve(as_ctx *ctx) {
  for (auto &e : ctx->pending) {
    float ms = 0.f;
    if (hipEventSynchronize(e.stop) == hipSuccess && hipEventElapsedTime(&ms, e.start, e.stop) == hipSuccess) {
      auto &p = ctx->prof[e.name];
      p.ms += ms;
      p.launches += 1;
    }
    (void)hipEventDestroy(e.start);
    (void)hipEventDestroy(e.stop);
  }
  ctx->pending.clear();
}

inline unsigned cdiv(uint64_t a, uint64_t b) { return (unsigned)((a + b - 1) / b); }

// ---- stage entry points implemented in the .hip files ---------------------------------------
// exact arithmetic row norms: norms[i] = sqrt(sum_c x[i][c]^2), sequential (src/core.rs:212-216)
int row_norms(as_ctx *ctx, const double *rows, uint64_t n, uint64_t d, double *norms);
// StandardScaler z-score of every column (normalise = true, src/laplacian.rs:146-151)
int standardise_rows(as_ctx *ctx, const double *rows, uint64_t n, uint64_t d, double *out);

struct KnnOut {
  uint32_t *idx;  // rows x K, kNoIdx when missing
  double *dist;   // rows x K, +inf when missing
};

// exact neighbour scan for the listed rows (row_ids == nullptr: rows row_begin..+row_count)
int knn_exact(as_ctx *ctx, const double *rows, const double *norms, uint64_t n, uint64_t d,
              const uint32_t *row_ids, uint64_t row_begin, uint64_t row_count, uint32_t K,
              int rectify, uint32_t *out_idx, double *out_dist, uint64_t out_stride_rows);

// self_in_results variant of the neighbour rule: post-process of a finished table (knn_exact.hip)
int apply_self_in_results(as_ctx *ctx, const double *rows, const double *norms, uint64_t d, uint64_t row_begin,
                          uint64_t row_count, uint32_t K, int rectify, uint32_t *idx, double *dist);

// Candidate lists of a threshold join that has already been run (row-sharded build: every rank contracts
// the tile pairs of its own super tiles, the lists are exchanged, each rank finishes its own rows).
// The join on an image in the library's own row order, prepared (own_join_prepare, mfma_drivers.hip): what the contraction
// of the tile pairs and every stage behind it need.  One per build; a build sharded by image positions (sharded.hip) prepares
// the same object on every rank.
// What the cone bound on tile pairs (cone_tiles.hip) keeps of the image's tiles for later one-sided uses (the second pass): the mean
// direction of every 128-position tile as an MFMA operand, the cosine and sine of its cone's half angle, the largest norm of a row.
struct ConeDirs {
  DevBuf<unsigned char> dirs_img;
  DevBuf<float> cos_a, sin_a;
  DevBuf<unsigned int> nmax;
  uint32_t ntiles = 0;
  bool valid = false;
  const uint32_t *group_start = nullptr;  // (set by the caller: where the pivot groups of the image begin, `groups` + 1 positions --
  uint32_t groups = 0;                    //  the tiles may be split there, cone_tiles.hip)
};
// Memory owned by handles (as_csr, as_index, as_jl, uploads): a block a destroyed handle gives back is kept (up to a cap) and handed
// to the next allocation of exactly its size on the same device, after the work queued on the library's streams when it was given back -- hipFree costs ~150 us whatever
// the size (it synchronises and unmaps), and a loop that rebuilds a graph or an index asks for the same sizes again.  The cap:
// environment ARROWSPACE_HANDLE_CACHE_MB or option "handle_cache_mb" (default 2048; 0 = off, and what is cached is released).
// Reuse waits for events recorded at the free on the streams the library's contexts run on (handle_cache_stream), not for the device.
void handle_cache_stream(int device, hipStream_t s, bool add);
hipError_t handle_malloc(void **p, size_t bytes);
hipError_t handle_free(void *p);
void handle_cache_limit(int64_t mb);
template <class T>
inline hipError_t handle_malloc(T **p, size_t bytes) {
  return handle_malloc(reinterpret_cast<void **>(p), bytes);
}
// what a build kept for an index of its rows (as_ctx::JoinKeep): released
inline void join_keep_clear(as_ctx *ctx) {
  as_ctx::JoinKeep &k = ctx->join_keep;
  if (k.packedh) (void)handle_free(k.packedh);
  if (k.packed8) (void)handle_free(k.packed8);
  if (k.perm) (void)handle_free(k.perm);
  if (k.piv) (void)handle_free(k.piv);
  if (k.rot_qw) (void)handle_free(k.rot_qw);
  if (k.norms) (void)handle_free(k.norms);
  k = as_ctx::JoinKeep{};
}

struct OwnJoin {
  DevBuf<unsigned char> P16own, Phown;                        // split-bf16 / f16 image of all rows (Phown in the order `perm`)
  DevBuf<unsigned char> Prow;                                 // sharded by rows with an order: the f16 image of all rows in ROW order (the all-gathered one)
  bool sharded_rows = false;                                  // the preparation ran on the ranks' own rows (ShardedRows below)
  ConeDirs cone;                                              // the tiles' cones, when the first pass ran behind the cone bound
  DevBuf<float> theta_own, theta2_own, theta_pos, theta2_pos; // thresholds (and the looser second ones) by row / by image position
  DevBuf<unsigned char> Hrow;           // one GPU, 128 columns: the rows' f16 form row-major in ROW order (rotate_wide.hip's fused pass), until the order is known
  DevBuf<float> err_by_row;             // ... and their plane errors by row
  DevBuf<float> f16_err_pos;            // plane error of every row of the f16 image by image position (the pruning's bound per pair)
  DevBuf<float> theta_band_own;         // by row: what the row's own neighbourhood certifies, less twice the bound (NaN: not computed); the second pass's floor
  DevBuf<uint32_t> perm_own, inv_own;                         // image position -> row; row -> image position (filled by whoever needs it)
  DevBuf<double> rot_z, rot_norm;                             // rotated coordinates of the rows (rotate_wide.hip), n x 32, and their norms
  DevBuf<float> rot_q;                                        // the 31 directions they were found on (31 x rotated_wide_pitch(d))
  DevBuf<float> order_piv;                                    // the pivots of the order found on them (np x 32), where the rows of each begin
  DevBuf<uint32_t> order_starts;
  DevBuf<uint32_t> order_near;                                // the 16 nearest pivot groups of every group (np x 16), where the chain of the order has them
  DevBuf<uint32_t> far_pos;                                   // image positions of the rows far from their pivot (row_order.hip), whole 512s, ~0 beyond far_count
  uint32_t far_count = 0;
  uint32_t order_np = 0, cap = 0;                             // pivots used; list slots per row
  bool wide_pre = false;       // the coordinates are in use (their directions carry the rows)
  bool reorder = false;        // the image is packed in the order perm_own (else: the rows as given)
  bool grouped_theta = false;  // thresholds from the sample of the nearest pivot groups
  bool have_theta2 = false;
  const uint32_t *perm = nullptr;       // perm_own.p or nullptr
  const float *theta_image = nullptr;   // thresholds in the order of the image: theta_pos.p or theta_own.p
};
// A build sharded by rows (csrc/sharded.hip): what the preparation of the join is handed instead of all rows.  Every rank holds
// the rows [lo, lo + count) (`own`; shard starts are multiples of 512) and the norms of ALL rows; per-row products of the
// preparation (coordinates, f16 image) are made from the own rows and completed by the gathers below, so the f64 rows themselves
// are not needed before the exact re-rank: their all-gather is issued behind the preparation (rows_begin) and awaited where they
// are first read (rows_all) -- under the first pass and the f16 pass instead of in front of everything.
// What the pass over all rows that computes the rotated coordinates may leave beside them (rotate_wide.hip: rotated_rows_wide,
// rows of exactly 128 values): the f16 form of the rows (the values the join's image stores) ROW-MAJOR in row order (n x 128 halves),
// the plane error of every row by row, the largest of them.  The caller provides the buffers; `done` says whether they were filled.
struct RotatedPacked {
  void *Hrow = nullptr;
  float *err_rows = nullptr;
  double f16_err = -1.0;
  bool done = false;
};

struct ShardedRows {
  const double *own = nullptr;
  uint32_t lo = 0, count = 0;
  int rank = 0, world = 1;
  std::vector<uint64_t> row_lo, row_hi;                        // the row range of every rank
  std::function<int(void *all, size_t elem_bytes)> gather;     // a buffer of n elements by row (or by image position): completed in place
  // pieces of any size: rank r contributes bytes[r] at offsets[r] of `all` (its own piece is in place already)
  std::function<int(void *all, const uint64_t *bytes, const uint64_t *offsets)> gather_v;
  std::function<int()> rows_begin;                             // start the all-gather of the f64 rows (once; returns at once)
  std::function<int(const double **rows)> rows_all;            // all rows, complete on the context's stream (starts the gather if nobody has)
};
int own_join_prepare(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, uint32_t K, OwnJoin &oj,
                     uint32_t pos_begin = 0, uint32_t pos_count = 0, const ShardedRows *sr = nullptr);
int own_join_pairs(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, uint32_t pos_begin, uint32_t pos_count,
                   OwnJoin &oj, uint32_t *cnt, uint32_t *cand, bool lists_by_pos);
uint32_t cone_tile_words(uint32_t n);
int cone_tile_mask(as_ctx *ctx, const double *Z, const uint32_t *perm, uint32_t n, const float *theta, uint32_t pos_begin, uint32_t pos_count,
                   uint32_t *tile_mask, double *kept_share, ConeDirs *keep = nullptr);
int cone_rect_mask(as_ctx *ctx, const double *Z, const uint32_t *ids, uint32_t nf, const float *theta_a, uint32_t pad, const ConeDirs &cone,
                   uint32_t n, unsigned char *mask, double *kept_share);

struct JoinSources {
  const float *packed;    // f32 image of all rows if the caller has one (else nullptr)
  const float *theta;     // thresholds of ALL rows (n)
  uint32_t n_src, cap;    // lists per row, slots per list
  const uint32_t *cnt;    // n_src x row_count
  const uint32_t *cand;   // n_src x row_count x cap
  const float *theta2 = nullptr;    // looser thresholds of the rows of the range (by position), or nullptr
  const void *packed16 = nullptr;   // split-bf16 image of all rows, or nullptr (the second pass packs its own then)
  const void *packedh = nullptr;    // f16 image of all rows (what the second pass over rows wider than 128 contracts)
  // a range of IMAGE POSITIONS [row_begin, row_begin + row_count) of a prepared join (build sharded by positions): the rows are
  // a_ids[r] = perm[row_begin + r], list r belongs to a_ids[r] and holds ROW ids; theta / theta2 are by global row,
  // theta_local[r] = theta[a_ids[r]]; oj->inv_own maps a row back to its position (list index = inv[g] - list_base)
  const uint32_t *a_ids = nullptr;
  const float *theta_local = nullptr;
  const OwnJoin *oj = nullptr;
  uint32_t list_base = 0;
};
// MFMA candidate pass + exact re-rank + certification + exact fall-back
int knn_mfma(as_ctx *ctx, const double *rows, const double *norms, uint64_t n, uint64_t d,
             uint64_t row_begin, uint64_t row_count, uint32_t K, int rectify, uint32_t *out_idx,
             double *out_dist, const JoinSources *js = nullptr);
// packed = f32 image of ALL rows (even tile count) when the caller already has it, else nullptr
// theta2_out (or nullptr): a second, looser threshold per row (bf16 pass only; left untouched otherwise, *has_theta2 = false)
int knn_join_thresholds(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d,
                        uint32_t row_begin, uint32_t row_count, float *theta_out, const float *packed = nullptr,
                        float *theta2_out = nullptr, bool *has_theta2 = nullptr, const void *packedh = nullptr);
// packed16 / packedh = split-bf16 / f16 image of ALL rows when the caller already has the one its pipe needs, else nullptr
int knn_join_pairs(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, uint32_t row_begin,
                   uint32_t row_count, const float *theta_all, uint32_t cap, uint32_t *cnt, uint32_t *cand,
                   const float *packed = nullptr, const void *packed16 = nullptr, const void *packedh = nullptr,
                   const uint32_t *perm = nullptr,         // image position -> row when packedh is packed in a permuted order
                   const uint32_t *image_perm = nullptr,   // ... when it is, but the lists are to be kept by image position
                   const double *rot_z = nullptr,          // rows wider than 128: their rotated coordinates (n x 32), for the first pass
                   ConeDirs *cone_keep = nullptr);
int join_session_open(as_ctx *ctx, const double *rows, uint64_t n, uint64_t d);   // (re)builds norms + image
bool join_session_matches(const as_ctx *ctx, const double *rows, uint64_t n, uint64_t d);
void join_session_close(as_ctx *ctx);
uint32_t knn_join_cap(const as_ctx *ctx, uint32_t K, bool sharded);
bool knn_join_applies(const as_ctx *ctx, uint32_t n, uint32_t d);  // for shard starts on super-tile boundaries
// batch nearest_centroid (src/clustering.rs:913-928), centroids.hip
int nearest_centroids(as_ctx *ctx, const double *rows, uint64_t n, uint64_t d, const double *cents, uint64_t n_cents,
                      uint64_t *out_idx, double *out_d2, double qmax2 = 0.0);  // qmax2 > 0: bound of |row|^2, |centroid|^2 (MFMA candidates)
int max_row_sqnorm(as_ctx *ctx, const double *rows, uint64_t n, uint64_t d, double *out);
int two_nn_mfma(as_ctx *ctx, const double *rows, uint64_t n, uint64_t d, const uint64_t *sample_ids_dev, uint64_t ns, double *out_dev,
                std::vector<uint32_t> *redo_samples);  // AS_ERR_UNSUPPORTED: keep the exact kernel; redo_samples: those it must still do  // upper bound; inf when a row is not finite

// rows [row_begin, row_begin + row_count) of the n x n Laplacian (the whole matrix: 0, n)
int laplacian_from_knn(as_ctx *ctx, const uint32_t *nbr_idx, const double *nbr_dist, uint64_t n,
                       uint32_t K, const as_graph_params *gp, as_csr **out, uint64_t row_begin, uint64_t row_count);
// the two halves of it apart (the row-sharded build: each rank ranks and prunes the edges of its own rows, the ranks exchange that
// adjacency, each assembles its row block from the adjacency of all nodes); K <= 32 only
struct GraphAdjacency {
  const uint32_t *adj_j;    // n x K kept neighbours of each node, in rank order
  const double *adj_w;      // n x K their weights
  const uint32_t *adj_len;  // n     how many
};
int graph_degrees_of_rows(as_ctx *ctx, const uint32_t *nbr_idx, const double *nbr_dist, const uint32_t *row_ids, uint64_t n_rows, uint32_t K,
                          const as_graph_params *gp, uint32_t *deg_out);
int graph_degrees_to_nodes(as_ctx *ctx, const uint32_t *deg_pos, const uint32_t *perm, uint64_t n, uint32_t *deg, unsigned long long *deg_sum);
int graph_adjacency_of_rows(as_ctx *ctx, const uint32_t *nbr_idx, const double *nbr_dist, const uint32_t *row_ids, uint64_t n_rows, uint64_t n_nodes,
                            uint32_t K, const as_graph_params *gp, const uint32_t *deg, const unsigned long long *deg_sum, uint32_t *adj_j, double *adj_w,
                            uint32_t *adj_len);
int graph_adjacency_to_nodes(as_ctx *ctx, const uint32_t *pj, const double *pw, const uint32_t *plen, const uint32_t *perm, uint64_t n, uint32_t K,
                             uint32_t *adj_j, double *adj_w, uint32_t *adj_len);
int laplacian_from_adjacency(as_ctx *ctx, const GraphAdjacency *adj, uint64_t n, uint32_t K, const as_graph_params *gp, as_csr **out, uint64_t row_begin,
                             uint64_t row_count);

int taumode_lambdas(as_ctx *ctx, const double *items, uint64_t n_items, uint64_t F, as_csr *L,
                    int mode, double arg, bool check_finite, double *lambdas_out);
int lambda_v3(as_ctx *ctx, const double *items, uint64_t n_items, uint64_t F, as_csr *L, int mode,
              double arg, bool check_finite, double *out, bool *handled);
int select_tau_rows(as_ctx *ctx, const double *rows, uint64_t n_rows, uint64_t len, int mode,
                    double arg, double *taus);
// item-graph lambda (signal_lambda.hip): the S columns of X (n_len x S) against L; signals stored as rows; the pieces of the sharded form
int signal_lambdas(as_ctx *ctx, const double *X, uint64_t n_len, uint32_t S, as_csr *L, int mode, double arg, double *out);
int signal_lambdas_rows(as_ctx *ctx, const double *items, uint64_t n_sig, uint64_t n_len, as_csr *L, int mode, double arg, double *out);
int signal_sums(as_ctx *ctx, const double *X, uint64_t n_len, uint32_t S, as_csr *L, uint64_t tail_begin, uint64_t tail_end, double *sums);
int signal_taus(as_ctx *ctx, const double *X, uint64_t n_len, uint32_t S, int mode, double arg, double *tau);
int ctx_side_stream(as_ctx *ctx);  // api.hip: creates ctx->side and its two events on first use
int signal_taus_rows(as_ctx *ctx, const double *rows, uint64_t n_rows, uint64_t len, int mode, double arg, double *taus);
int signal_lambda_finish(as_ctx *ctx, const double *X, uint32_t S, as_csr *L, const double *sums, const double *tau, double *out);
int signal_shares(as_ctx *ctx, const double *X, uint32_t S, as_csr *L, const double *sums, const int *need, double *g2);
int signal_rank_sum(as_ctx *ctx, const double *all, uint32_t world, uint32_t count, double *out);
uint32_t signal_pad(uint32_t S);
int signal_need_pass2(as_ctx *ctx, const double *sums, uint32_t S, int *need_dev);
int signal_lambda_from_sums(as_ctx *ctx, const double *sums, const double *g2, const int *need_dev, bool upper_only, uint32_t S, const double *tau, double *out);
int csr_set_order(as_ctx *ctx, as_csr *L, const uint32_t *order_dev);
int any_nonfinite(as_ctx *ctx, const double *p, uint64_t count, bool *out);
void signal_stream_release(as_csr *L);

int search_exact(as_ctx *ctx, const as_index *ix, const double *queries, const double *q_norms,
                 const double *q_lambdas, const uint32_t *q_ids, uint64_t nq_sel, uint64_t nq,
                 uint32_t k, double alpha, uint32_t *out_idx, double *out_score);
// the rotated int8 image of the items (natural order) and what the filter of the small-batch search needs next to it
struct StreamFilter {
  const void *P8;
  float err_items;
  const float *Qd;          // the 31 directions (32 x 128)
  const double *a_norms;    // norms and lambdas of the queries
  const double *a_lams;
  const double *x_norms, *x_lams;
  double alpha;
  // the item graph's leftover rows (GraphDistPolicy: key = 1 - cos): the image of the A rows is given (PQ8, with its error),
  // P8 is in the order of the join's image (perm: position -> row)
  const void *PQ8 = nullptr;
  float err_a = 0.f;
  const uint32_t *perm = nullptr;
};
int rotated_image_pack_async(as_ctx *ctx, const double *rows, const double *norms, const uint32_t *perm, uint32_t n, uint32_t d,
                             const float *Qd, void *P8, uint32_t tiles, unsigned int *err_bits_dev);
uint64_t rotated_image_bytes(uint32_t tiles);
int search_mfma(as_ctx *ctx, as_index *ix, const double *queries, const double *q_norms,
                const double *q_lambdas, uint64_t nq, uint32_t k, double alpha, uint32_t *out_idx,
                double *out_score);

// exact scan of a few rows (<= 32 per pass) against all rows in one pass at HBM speed (stream_scan.hip), dim 128
bool stream_scan_applies(uint64_t nA, uint64_t nB, uint64_t d, uint32_t K);

// device exclusive scan of uint32 -> uint64 (out has n+1 entries, out[n] = total)
int exclusive_scan_u32_u64(as_ctx *ctx, const uint32_t *in, uint64_t n, uint64_t *out);

}  // namespace asx
