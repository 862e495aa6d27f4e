// mfma_drivers.hip -- staged driver of the MFMA neighbour search (K2): preparation of the own join (order, images, thresholds), cone
// bound + int8 first pass + f16 join, pruning + pair-once exact re-rank + certification, second pass, f32 / f64 MFMA top-k passes and
// exact scans for the remainder.  See mfma_stages.h (shared with search_drivers.hip, the K9 driver) / DESIGN.md section 4.
#include "mfma_stages.h"
#include "rerank_split.h"
#include "rerank_pairs.h"
namespace asx {

// records (rerank_prune_kernel) -> every pair scored once -> the K best of every row + the rows that could not be certified
// (rerank_pairs.h).  records are by place p: row = order ? order[p] : p, its id a_ids ? a_ids[row] : row_begin + row;
// row2rec: row id -> record of the whole-matrix image order (the inverse of `order`), or nullptr (built here).
__global__ void gather_f64_kernel(const double *__restrict__ src, const uint32_t *__restrict__ idx, uint32_t n, double *__restrict__ dst) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}

struct RerankPairsState {  // what outlives rerank_pairs_launch until the caller's next read-back
  DevBuf<uint32_t> map, s_ids, meta, rowpos, seg_batches;
  DevBuf<uint64_t> seg_off;
  DevBuf<rp_u4> s_desc;
  DevBuf<unsigned long long> counters;
  uint32_t cap_batches = 0;
};
template <int SLOTS>
static int rerank_pairs_launch(as_ctx *ctx, RerankPairsState &st, uint64_t cap_batches, const double *rows, uint32_t d, uint64_t n_total,
                               uint32_t row_begin, uint32_t row_count, uint32_t *records, uint32_t K, int rectify, GraphBound bnd, double *out_dist,
                               uint32_t *out_idx, uint32_t *fail_rows, uint32_t *fail_count, const uint32_t *a_ids, const uint32_t *order,
                               const uint32_t *row2rec) {
  constexpr uint32_t S = 64u * SLOTS;
  const uint32_t n_seg = (row_count + kRpSeg - 1) / kRpSeg;
  const int share = ctx->opt_rerank_share && row_count < kRpShareMax;
  if (share && !row2rec) {
    if (!st.map.p) {
      AS_TRY(st.map.alloc(ctx, n_total));
      AS_HIP(ctx, hipMemsetAsync(st.map.p, 0xFF, sizeof(uint32_t) * n_total, ctx->stream));
      hipLaunchKernelGGL(rerank_row2rec_kernel, dim3(cdiv(row_count, 256)), dim3(256), 0, ctx->stream, row_begin, row_count, a_ids, order, st.map.p);
    }
    row2rec = st.map.p;
  }
  if (cap_batches >= 0xFFFFFFF0ull / 16) return fail(ctx, AS_ERR_UNSUPPORTED, "re-rank: too many pairs for one call");
  st.cap_batches = (uint32_t)cap_batches;
  AS_TRY(st.s_ids.alloc(ctx, (size_t)cap_batches * 16));
  AS_TRY(st.s_desc.alloc(ctx, (size_t)cap_batches * 16));
  if (!st.counters.p) AS_TRY(st.counters.alloc(ctx, kRpCounters));
  AS_HIP(ctx, hipMemsetAsync(st.counters.p, 0, sizeof(unsigned long long) * kRpCounters, ctx->stream));
  AS_TRY(st.meta.alloc(ctx, (size_t)row_count * S));
  AS_TRY(st.rowpos.alloc(ctx, row_count));
  AS_TRY(st.seg_batches.alloc(ctx, n_seg));
  AS_TRY(st.seg_off.alloc(ctx, (size_t)n_seg + 1));
  {
    ProfScope ps(ctx, "knn_rerank_link");
    // (a workgroup per segment, a multiple of 8 of them: the kernel deals the segments by XCD)
    const uint32_t link_grid = (n_seg + 7u) / 8u * 8u;
    hipLaunchKernelGGL((rerank_link_kernel<SLOTS>), dim3(link_grid), dim3(256), 0, ctx->stream, row_begin, row_count,
                       records, a_ids, order, row2rec, share, st.meta.p, st.rowpos.p, st.seg_batches.p, st.counters.p);
    AS_HIP(ctx, hipGetLastError());
    AS_TRY(exclusive_scan_u32_u64(ctx, st.seg_batches.p, n_seg, st.seg_off.p));
  }
  {
    ProfScope ps(ctx, "knn_rerank_fill");
    hipLaunchKernelGGL((rerank_fill_kernel<SLOTS>), dim3(std::min<uint32_t>(n_seg, 256u * 32u)), dim3(256), 0, ctx->stream, row_begin, row_count,
                       (const uint32_t *)records, a_ids, order, (const uint32_t *)st.meta.p, (const uint32_t *)st.rowpos.p,
                       (const uint32_t *)st.seg_batches.p, (const uint64_t *)st.seg_off.p, st.s_ids.p, reinterpret_cast<uint4 *>(st.s_desc.p),
                       st.cap_batches, st.counters.p);
    AS_HIP(ctx, hipGetLastError());
  }
  {
    ProfScope ps(ctx, "knn_rerank_score");
    const uint32_t rs_grid = 256u * 7u * 8u;  // workgroups (one wave each)
    // (the stream's length is known on the device only: a grid sized for the chip; waves past the stream's end leave at once)
    const uint32_t grid = std::min<uint64_t>(rs_grid, std::max<uint64_t>(8, cap_batches / 8 * 8));
    const uint32_t dbg = (ctx->opt_debug & 64) ? (uint32_t)(ctx->opt_debug >> 8) : 0u;
    const size_t lds = (size_t)kRsRows * kRsPitch;
#define AS_SCORE(PC)                                                                                                                              \
  case PC:                                                                                                                                        \
    hipLaunchKernelGGL((rerank_pairs_score_kernel<PC, SLOTS>), dim3(grid), dim3(64), lds, ctx->stream, rows, (const uint32_t *)st.s_ids.p,         \
                       (const rp_u4 *)st.s_desc.p, (const unsigned long long *)st.counters.p, rectify, records, dbg);                             \
    break;
    switch (d / 128) {
      AS_SCORE(1) AS_SCORE(2) AS_SCORE(3) AS_SCORE(4) AS_SCORE(5) AS_SCORE(6) AS_SCORE(7) AS_SCORE(8)
      default: return fail(ctx, AS_ERR_UNSUPPORTED, "re-rank: row width not covered");
    }
#undef AS_SCORE
    AS_HIP(ctx, hipGetLastError());
  }
  ProfScope ps(ctx, "knn_rerank_rank");
  hipLaunchKernelGGL((rerank_rank_kernel<SLOTS>), dim3(std::min<uint64_t>(cdiv(row_count, 8), 256 * 32)), dim3(256), 0, ctx->stream, row_begin, row_count,
                     (const uint32_t *)records, K, bnd, out_dist, out_idx, fail_rows, fail_count, a_ids, order, st.counters.p);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

// records (rerank_prune_kernel) -> every pair scored once -> the K best of every row + the rows that could not be certified
// (rerank_pairs.h).  records are by place p: row = order ? order[p] : p, its id a_ids ? a_ids[row] : row_begin + row;
// row2rec: row id -> record of the whole-matrix image order (the inverse of `order`), or nullptr (built here).
// No read-back in between: the slot stream gets room for 3 batches per row (half of that is used on the bench data); a call that
// needs more finds out at the caller's read-back of fail_count (*n_fail) and is repeated here with the size the device counted.
template <int SLOTS>
static int rerank_pairs_run(as_ctx *ctx, const double *rows, uint32_t d, uint64_t n_total, uint32_t row_begin, uint32_t row_count, uint32_t *records,
                            uint32_t K, int rectify, GraphBound bnd, double *out_dist, uint32_t *out_idx, uint32_t *fail_rows, uint32_t *fail_count,
                            const uint32_t *a_ids, const uint32_t *order, const uint32_t *row2rec) {
  RerankPairsState st;
  uint64_t cap = (uint64_t)row_count * std::max<uint64_t>(3, (3 * (uint64_t)K + 15) / 16) + 64;
  if (ctx->opt_rerank_stream_cap > 0) cap = (uint64_t)ctx->opt_rerank_stream_cap;  // (test hook: a first attempt that runs out of room)
  for (int attempt = 0; attempt < 2; ++attempt) {
    AS_TRY(rerank_pairs_launch<SLOTS>(ctx, st, cap, rows, d, n_total, row_begin, row_count, records, K, rectify, bnd, out_dist, out_idx, fail_rows,
                                      fail_count, a_ids, order, row2rec));
    unsigned long long h[kRpCounters] = {};
    AS_HIP(ctx, copy_d2h(ctx, h, st.counters.p, sizeof(h)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stats["knn_rerank_batches"] = (int64_t)h[0];
    for (int s = 0; s < kRpPairSlots; ++s) h[1] += h[kRpPairsAt + s];
    ctx->stats["knn_rerank_pairs_scored"] = (int64_t)h[1];
    ctx->stats["knn_rerank_stream_retries"] = attempt;
    ctx->stats["knn_rerank_open_list_ran_over"] = (int64_t)h[4];
    ctx->stats["knn_rerank_open_forced"] = (int64_t)h[5];
    ctx->stats["knn_rerank_open_undelivered"] = (int64_t)h[6];
    ctx->stats["knn_rerank_open_too_few"] = (int64_t)h[7];
    ctx->stats["knn_rerank_open_kth_not_clear"] = (int64_t)h[8];
    if (h[2] == 0) return AS_OK;
    cap = h[0] + 64;
    AS_HIP(ctx, hipMemsetAsync(fail_count, 0, sizeof(uint32_t), ctx->stream));  // (the ranking ran on keys nobody wrote)
  }
  return fail(ctx, AS_ERR_HIP, "re-rank: the slot stream overflowed twice");
}

// Exact re-rank of the joined candidates of rows [row_begin, row_begin + row_count) (one wave per row),
// certification against theta.  The candidates of a row come from n_src lists (one per rank that contracted
// tile pairs touching it): cnt_src[s * row_count + r], cand_src[(s * row_count + r) * cap ..].
bool knn_join_applies(const as_ctx *ctx, uint32_t n, uint32_t d) { return join_supported(ctx, n, d, 0); }

// theta_out[r] for r in [row_begin, row_begin + row_count): the m-th best approximate cosine of the row against a
// hashed 1-in-`rate` sample of ALL rows (the existing top-k pass with a short list).
int knn_join_thresholds(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, uint32_t row_begin,
                        uint32_t row_count, float *theta_out, const float *packed, float *theta2_out, bool *has_theta2,
                        const void *packedh) {
  if (has_theta2) *has_theta2 = false;
  if (!join_supported(ctx, n, d, row_begin)) return fail(ctx, AS_ERR_UNSUPPORTED, "threshold join not applicable");
  const int64_t rank_opt = ctx->opt_join_rank > 0 ? ctx->opt_join_rank : 4;
  const int64_t sample_opt = ctx->opt_join_sample > 0 ? ctx->opt_join_sample : 32;
  const uint32_t m = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(rank_opt, 64));
  const uint32_t a_tiles = (row_count + kTileRows - 1) / kTileRows, a_alloc = (a_tiles + 1) & ~1u;
  // sample size: whole pairs of 128-row tiles (the bf16 pass bins the sample by tile parity and has no tail
  // handling), about n / join_sample of them; n >= 512 here
  const uint32_t ns = std::max(256u, (n / (uint32_t)std::max<int64_t>(2, sample_opt)) & ~255u);
  const uint32_t rate = n / ns;
  const uint32_t s_tiles = (ns + kTileRows - 1) / kTileRows, s_alloc = (s_tiles + 1) & ~1u;
  const uint32_t KC0 = std::max(8u, (m + 7) / 8 * 8);
  DevBuf<uint32_t> sid, cidx;
  DevBuf<float> S, PA;
  DevBuf<double> ckey;
  AS_TRY(sid.alloc(ctx, ns));
  hipLaunchKernelGGL(sample_ids_kernel, dim3(cdiv(ns, 256)), dim3(256), 0, ctx->stream, ns, rate, sample_half_bits(ns), sid.p);
  if (ctx->opt_join_bf16) {  // thresholds on a 16-bit pipe (mfma_theta_{f16,bf16}.hip); dims they do not cover take the f32 pass
    // second threshold: rank 2 m + 6 (about three times the candidates), for the second pass over what fails
    const uint32_t m2 = std::min<uint32_t>(48, 2 * m + 6);
    // the f16 image of all rows, when the join runs on it, serves as the row operand: one MFMA per k step
    const bool on_f16 = packedh && join_on_f16(ctx, d) && !(ctx->opt_debug & 32);
    const int rc = on_f16 && d > 128
                       ? join_thresholds_f16_wide(ctx, rows, norms, d, packedh, row_begin, row_count, sid.p, ns, m, theta_out, m2,
                                                  theta2_out)
                   : on_f16
                       ? join_thresholds_f16(ctx, rows, norms, d, packedh, row_begin, row_count, sid.p, ns, m, theta_out, m2,
                                             theta2_out)
                       : join_thresholds_bf16(ctx, rows, norms, d, row_begin, row_count, sid.p, ns, m, theta_out, m2, theta2_out);
    if (rc == AS_OK && theta2_out && has_theta2) *has_theta2 = true;
    if (rc != AS_ERR_UNSUPPORTED) return rc;
  }
  const float *pa = packed ? packed + (size_t)(row_begin / kTileRows) * packed_elems(1, d) : nullptr;
  if (!pa) {
    AS_TRY(PA.alloc(ctx, packed_elems(a_alloc, d)));
    AS_TRY(pack_f32(ctx, rows, norms, nullptr, row_begin, row_count, d, PA.p, a_alloc));
    pa = PA.p;
  }

  AS_TRY(S.alloc(ctx, packed_elems(s_alloc, d)));
  AS_TRY(pack_f32(ctx, rows, norms, sid.p, 0, ns, d, S.p, s_alloc));
  AS_TRY(ckey.alloc(ctx, (size_t)row_count * KC0));
  AS_TRY(cidx.alloc(ctx, (size_t)row_count * KC0));
  CandProblem pb0{};
  pb0.mode = 0;
  pb0.A = rows;
  pb0.a_norms = norms;
  pb0.arows = ARows{nullptr, row_begin};
  pb0.nA = row_count;
  pb0.B = rows;
  pb0.b_norms = norms;
  pb0.nB = ns;
  pb0.d = d;
  pb0.KC = KC0;
  pb0.splits = 1;
  pb0.cand_key = ckey.p;
  pb0.cand_idx = cidx.p;
  AS_TRY(candidates_f32(ctx, pb0, pa, a_tiles, S.p, s_tiles, "knn_join_thresholds"));
  hipLaunchKernelGGL(join_theta_kernel, dim3(cdiv(row_count, 256)), dim3(256), 0, ctx->stream, ckey.p, cidx.p, row_count,
                     KC0, m, row_count, theta_out);
  AS_HIP(ctx, hipGetLastError());
  // (no synchronisation to hand the workspace back: a block returned to the cache is reused by later work on this same stream only)
  return AS_OK;
}

// Contracts the tile pairs owned by the super tiles of [row_begin, row_begin + row_count) and appends to
// cnt / cand (indexed by GLOBAL row, n x cap; zeroed here) every pair that reaches a threshold.
int knn_join_pairs(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, uint32_t row_begin,
                   uint32_t row_count, const float *theta_all, uint32_t cap, uint32_t *cnt, uint32_t *cand,
                   const float *packed, const void *packed16, const void *packedh, const uint32_t *perm, const uint32_t *image_perm,
                   const double *rot_z, ConeDirs *cone_keep) {
  if (!join_supported(ctx, n, d, row_begin)) return fail(ctx, AS_ERR_UNSUPPORTED, "threshold join not applicable");
  const uint32_t b_tiles = (n + kTileRows - 1) / kTileRows, b_alloc = (b_tiles + 1) & ~1u;
  const bool f16 = join_on_f16(ctx, d), bf16 = join_on_bf16(ctx, d);
  const uint32_t super = f16 ? 512 : 256;
  const uint32_t T2 = (n + super - 1) / super;
  const uint32_t I_begin = row_begin / super;
  const uint32_t I_end = (uint32_t)(((uint64_t)row_begin + row_count + super - 1) / super);
  DevBuf<float> P, theta;
  DevBuf<unsigned char> P16;
  DevBuf<uint32_t> unitc;
  if (f16) {
    if (!packedh) {
      const uint32_t tiles = packed_tiles_f16(n);
      AS_TRY(P16.alloc(ctx, packed_vecs_f16(tiles, d) * 16));
      AS_TRY(pack_f16_all(ctx, rows, norms, n, d, P16.p, tiles));
      packedh = P16.p;
    }
  } else if (bf16) {
    if (!packed16) {
      AS_TRY(P16.alloc(ctx, packed_vecs_bf16(b_alloc, d) * 16));
      AS_TRY(pack_bf16(ctx, rows, norms, nullptr, 0, n, d, P16.p, b_alloc));
      packed16 = P16.p;
    }
  } else if (!packed) {
    AS_TRY(P.alloc(ctx, packed_elems(b_alloc, d)));
    AS_TRY(pack_f32(ctx, rows, norms, nullptr, 0, n, d, P.p, b_alloc));
    packed = P.p;
  }
  AS_TRY(theta.alloc(ctx, (size_t)T2 * super));
  hipLaunchKernelGGL(pad_theta_kernel, dim3(cdiv((uint64_t)T2 * super, 256)), dim3(256), 0, ctx->stream, theta_all, n, T2 * super,
                     theta.p);
  AS_TRY(unitc.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(cnt, 0, sizeof(uint32_t) * n, ctx->stream));
  AS_HIP(ctx, hipMemsetAsync(unitc.p, 0, sizeof(uint32_t), ctx->stream));
  if (f16) {
    if (d <= 128) {
      // int8 first pass (mfma_prefilter_i8.hip): the f16 pass then visits only the ~2 % of the blocks that can list something
      DevBuf<unsigned char> mask;
      // (only on an image whose rows are grouped by direction, row_order.hip: in the order of a caller's rows -- the bench
      // generator deals its components round-robin -- every block holds some pair of neighbours and the pass rules out nothing;
      // join_prefilter_min_rows = 0 forces it, for the tests)
      const bool pre = ctx->opt_join_prefilter && n >= (uint64_t)std::max<int64_t>(0, ctx->opt_join_prefilter_min_rows) &&
                       (perm || image_perm || ctx->opt_join_prefilter_min_rows == 0);
      if (pre) {
        AS_TRY(mask.alloc(ctx, prefilter_mask_bytes(n, I_end - I_begin)));
        if (rot_z) {  // the caller has the rotated coordinates of the rows (own join: they also gave the row order)
          DevBuf<unsigned char> P8;
          const uint32_t tiles = packed_tiles_f16(n);
          double err = 0.0;
          AS_TRY(P8.alloc(ctx, (size_t)tiles * 4 * 64 * 16));
          {
            ProfScope ps(ctx, "knn_join_prefilter_pack");
            AS_TRY(pack_rotated_from_z(ctx, rot_z, image_perm ? image_perm : perm, n, tiles, P8.p, &err));
          }
          if (ctx->keep_p8_dst && I_begin == 0 && I_end == T2) {  // ("keep_join_images": a copy for an index of these rows)
            AS_HIP(ctx, hipMemcpyAsync(ctx->keep_p8_dst, P8.p, rotated_image_bytes(tiles), hipMemcpyDeviceToDevice, ctx->stream));
            ctx->keep_p8_err = err;
            ctx->keep_p8_done = true;
          }
          // in front of it, on an image ordered by direction: which pairs of 128-position tiles can hold a pair at all (a cone
          // around each tile's direction against the exact products of the other tile's rows with it: cone_tiles.hip).  Where the
          // bound keeps most pairs (data without directions to tell apart) the pass sweeps everything as before.
          DevBuf<uint32_t> tile_mask;
          const uint32_t tm_words = cone_tile_words(n);
          bool tiled = ctx->opt_join_cone_tiles != 0 && (perm || image_perm) && n >= 65536 && !(ctx->opt_debug & 2048);
          if (tiled) {
            double kept = 1.0;
            AS_TRY(tile_mask.alloc(ctx, (size_t)(I_end - I_begin) * 4 * tm_words));
            AS_TRY(cone_tile_mask(ctx, rot_z, image_perm ? image_perm : perm, n, theta.p, I_begin * 512u, (I_end - I_begin) * 512u, tile_mask.p, &kept,
                                  cone_keep));
            tiled = kept <= (double)ctx->opt_join_cone_tiles_max_e3 * 1e-3;
          }
          ctx->stats["knn_join_cone_tiles"] = tiled ? 1 : 0;
          AS_TRY(prefilter_i8_rotated(ctx, P8.p, err, n, I_begin, I_end - I_begin, theta.p, mask.p, tiled ? tile_mask.p : nullptr, tm_words));
        } else {
          AS_TRY(prefilter_i8(ctx, packedh, rows, norms, n, d, I_begin, I_end - I_begin, theta.p, mask.p, image_perm ? image_perm : perm));
        }
      }
      ctx->stats["knn_join_prefilter"] = pre ? 1 : 0;
      AS_TRY(join_f16(ctx, packedh, n, d, I_begin, I_end - I_begin, theta.p, cap, cnt, cand, unitc.p, "knn_mfma_join_f16", perm,
                      pre ? mask.p : nullptr));
    } else {
      // rows wider than 128: the same first pass on the rotated coordinates the caller has (rotate_wide.hip), reduced to "which
      // pairs of 256-row super tiles hold anything" for the panelled kernel
      DevBuf<unsigned char> mask, P8;
      // (on any range of row super tiles: a rank of a sharded build sweeps its own positions against every column)
      const bool pre = rot_z != nullptr && I_end > I_begin;
      if (pre) {
        const uint32_t tiles = packed_tiles_f16(n);
        double err = 0.0;
        AS_TRY(P8.alloc(ctx, (size_t)tiles * 4 * 64 * 16));
        AS_TRY(mask.alloc(ctx, prefilter_mask_bytes(n, I_end - I_begin)));
        AS_HIP(ctx, hipMemsetAsync(mask.p, 0, prefilter_mask_bytes(n, I_end - I_begin), ctx->stream));  // (both orientations of a pair are looked up)
        {
          ProfScope ps(ctx, "knn_join_prefilter_pack");
          AS_TRY(pack_rotated_from_z(ctx, rot_z, image_perm ? image_perm : perm, n, tiles, P8.p, &err));
        }
        // (behind the same cone bound on tile pairs as the narrow rows' pass)
        DevBuf<uint32_t> tile_mask;
        const uint32_t tm_words = cone_tile_words(n);
        bool tiled = ctx->opt_join_cone_tiles != 0 && (perm || image_perm) && n >= 65536 && !(ctx->opt_debug & 2048);
        if (tiled) {
          double kept = 1.0;
          AS_TRY(tile_mask.alloc(ctx, (size_t)(I_end - I_begin) * 4 * tm_words));
          AS_TRY(cone_tile_mask(ctx, rot_z, image_perm ? image_perm : perm, n, theta.p, I_begin * 512u, (I_end - I_begin) * 512u, tile_mask.p, &kept));
          tiled = kept <= (double)ctx->opt_join_cone_tiles_max_e3 * 1e-3;
        }
        ctx->stats["knn_join_cone_tiles"] = tiled ? 1 : 0;
        AS_TRY(prefilter_i8_rotated(ctx, P8.p, err, n, I_begin, I_end - I_begin, theta.p, mask.p, tiled ? tile_mask.p : nullptr, tm_words));
      }
      ctx->stats["knn_join_prefilter"] = pre ? 1 : 0;
      AS_TRY(join_f16_wide(ctx, packedh, n, d, I_begin, I_end - I_begin, theta.p, cap, cnt, cand, "knn_mfma_join_f16", perm,
                           pre ? mask.p : nullptr));
    }
    ctx->stats["knn_join_blocks"] = ctx->stats["knn_mfma_join_f16_blocks"];
    ctx->stats["knn_join_pipe"] = 16;
  } else if (bf16) {
    AS_TRY(join_bf16(ctx, packed16, n, d, I_begin, I_end - I_begin, theta.p, cap, cnt, cand, unitc.p, "knn_mfma_join_bf16"));
    ctx->stats["knn_join_blocks"] = ctx->stats["knn_mfma_join_bf16_blocks"];
    ctx->stats["knn_join_pipe"] = 48;
  } else {
    AS_TRY(join_f32(ctx, packed, n, d, I_begin, I_end - I_begin, theta.p, cap, cnt, cand, unitc.p, "knn_mfma_join_f32"));
    ctx->stats["knn_join_blocks"] = (int64_t)(I_end - I_begin) * 64;  // at least the diagonal super tiles
    ctx->stats["knn_join_pipe"] = 32;
  }
  return AS_OK;
}

int join_session_open(as_ctx *ctx, const double *rows, uint64_t n, uint64_t d) {
  join_session_close(ctx);
  ctx->f16_err = -1.0;
  auto &js = ctx->join;
  const uint32_t b_tiles = (uint32_t)((n + kTileRows - 1) / kTileRows), b_alloc = (b_tiles + 1) & ~1u;
  void *p = nullptr;
  AS_TRY(ws_alloc(ctx, sizeof(double) * n, &p, &js.norms_cap));
  js.norms = (double *)p;
  AS_TRY(row_norms(ctx, rows, n, d, js.norms));
  if (join_on_f16(ctx, (uint32_t)d)) {
    const uint32_t tiles = packed_tiles_f16(n);
    AS_TRY(ws_alloc(ctx, packed_vecs_f16(tiles, (uint32_t)d) * 16, &p, &js.packedh_cap));
    js.packedh = p;
    AS_TRY(pack_f16_all(ctx, rows, js.norms, n, (uint32_t)d, js.packedh, tiles));
  } else if (!join_on_bf16(ctx, (uint32_t)d)) {  // the f32 image: only the f32 pipe needs it up front (a later f32 stage packs its own)
    AS_TRY(ws_alloc(ctx, sizeof(float) * packed_elems(b_alloc, (uint32_t)d), &p, &js.packed_cap));
    js.packed = (float *)p;
    AS_TRY(pack_f32(ctx, rows, js.norms, nullptr, 0, n, (uint32_t)d, js.packed, b_alloc));
  } else {
    AS_TRY(ws_alloc(ctx, packed_vecs_bf16(b_alloc, (uint32_t)d) * 16, &p, &js.packed16_cap));
    js.packed16 = p;
    AS_TRY(pack_bf16(ctx, rows, js.norms, nullptr, 0, n, (uint32_t)d, js.packed16, b_alloc));
  }
  js.rows = rows;
  js.n = n;
  js.d = d;
  return AS_OK;
}
bool join_session_matches(const as_ctx *ctx, const double *rows, uint64_t n, uint64_t d) {
  return ctx->join.rows == rows && ctx->join.n == n && ctx->join.d == d && ctx->join.norms &&
         (ctx->join.packed || ctx->join.packed16 || ctx->join.packedh);
}
void join_session_close(as_ctx *ctx) {
  auto &js = ctx->join;
  if (js.norms) ws_free(ctx, js.norms, js.norms_cap);
  if (js.packed) ws_free(ctx, js.packed, js.packed_cap);
  if (js.packed16) ws_free(ctx, js.packed16, js.packed16_cap);
  if (js.packedh) ws_free(ctx, js.packedh, js.packedh_cap);
  if (js.theta2) ws_free(ctx, js.theta2, js.theta2_cap);
  js = as_ctx::JoinSession{};
}

uint32_t knn_join_cap(const as_ctx *ctx, uint32_t K, bool sharded) {
  (void)sharded;
  const int64_t cap_opt = ctx->opt_join_cap > 0 ? ctx->opt_join_cap : 512;
  return (uint32_t)std::max<int64_t>(K + 1, std::min<int64_t>(cap_opt, 1024));
}

// ---- the join on an image in the library's own row order: preparation ------------------------------------------------------
// Everything in front of the contraction of the tile pairs: the rotated coordinates of the rows (rotate_wide.hip), the row
// order found on them (row_order.hip), the f16 image packed in that order, the thresholds (grouped sample + check, or the
// plain sample).  All of it is a function of the rows and the options alone and every sum on the way has a fixed order, so
// every rank of a sharded build that holds all rows computes the SAME order, image and thresholds (csrc/sharded.hip calls
// this on every rank and then contracts only the tile pairs of its own range of image positions).
// pos_begin / pos_count / gather (a sharded build): the thresholds -- the one step of the preparation that is a sweep of MFMAs
// per row rather than a pass over the row -- are computed for that range of image positions (rows, when there is no order)
// only, and gather(buffer of n elements, element bytes) completes the buffer on every rank (in place: the caller's own part
// is where it belongs).  Everything else stays replicated.
constexpr double kRotatedTailMax = 0.95;  // share of the rows' energy outside the 31 directions up to which the rotated first pass is taken
int own_join_prepare(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, uint32_t K, OwnJoin &oj,
                     uint32_t pos_begin, uint32_t pos_count, const ShardedRows *sr) {
  const std::function<int(void *, size_t)> *gather = sr ? &sr->gather : nullptr;
  if (!gather) {
    pos_begin = 0;
    pos_count = n;
  }
  // the figures of stages that may not run in this call: a call that skips a stage must not report the call before's
  for (const char *k : {"knn_join_order_cover_rounds", "knn_join_order_pivots", "knn_join_order_cover_low_e3", "knn_join_order_chain",
                        "knn_join_theta_far_rows", "knn_join_theta_dip_rows"})
    ctx->stats[k] = 0;
  // sr (ShardedRows, common.h): `rows` is null until somebody needs the f64 rows of OTHER ranks (need_rows: the steps that have no
  // form on the rank's own rows -- the plain row order, the plain thresholds, widths the sharded packing does not cover); on the
  // path the benchmark shapes take nothing in here does, and the caller starts their all-gather behind this function
  auto need_rows = [&]() -> int {
    if (!rows) AS_TRY(sr->rows_all(&rows));
    return AS_OK;
  };
  if (!sr) oj.sharded_rows = false;
  const uint32_t b_tiles = (n + kTileRows - 1) / kTileRows;
  const uint32_t b_alloc = (b_tiles + 1) & ~1u;
  ctx->f16_err = -1.0;
  oj.cap = knn_join_cap(ctx, K, false);
  AS_TRY(oj.theta_own.alloc(ctx, n));
  AS_TRY(oj.theta2_own.alloc(ctx, n));
  // rows grouped by direction in the operand image (row_order.hip): the thresholds come out by image position
  const bool reorder_narrow = join_on_f16(ctx, d) && d <= 128 && ctx->opt_join_row_order && n >= 65536 && ctx->opt_join_bf16 &&
                              !(ctx->opt_debug & 32);  // (the f16 threshold pass is what reads the image)
  bool &reorder = oj.reorder;
  reorder = reorder_narrow;
  // rows wider than 128 (up to 1024): their rotated coordinates (31 dominant directions + tail norm, rotate_wide.hip) serve
  // the int8 first pass and, being 32 wide, the row order as well
  oj.wide_pre = false;
  RotatedPacked rp;
  struct RowMajorScope {  // (ctx->rowmajor_image lives as long as this function, whatever way it is left: its buffer belongs to oj)
    as_ctx *c;
    ~RowMajorScope() { c->rowmajor_image = nullptr; }
  } rowmajor_scope{ctx};
  ctx->rowmajor_image = nullptr;
  if (join_on_f16(ctx, d) && rotated_wide_supported(d) && ctx->opt_join_prefilter && ctx->opt_join_prefilter != 2 &&
      n >= (uint64_t)std::max<int64_t>(0, ctx->opt_join_prefilter_min_rows) && ctx->opt_join_bf16 && !(ctx->opt_debug & 32)) {
    ProfScope ps(ctx, "knn_join_prefilter_basis");
    double tail = 1.0;
    AS_TRY(oj.rot_z.alloc(ctx, (size_t)n * 32));
    AS_TRY(oj.rot_norm.alloc(ctx, n));
    if (sr && d > 128) AS_TRY(need_rows());
    // rows of exactly 128 values on one GPU: the same pass leaves the f16 form of the rows row-major in ROW order and every row's plane
    // error (one read of the rows instead of two; the image in the order of the join is re-arranged from it below) -- "join_fused_rows_pass" (1)
    // ... and on every rank of a build sharded by rows: its own rows into their place, the rest arrives with an all-gather of 256 B a
    // row (what the fragment image of the own rows used to be gathered as), and the re-arrangement reads whole rows
    const double *pass_rows = (sr && !rows) ? sr->own : rows;
    if (pass_rows && (!sr || !rows) && d == 128 && ((uintptr_t)pass_rows & 15) == 0 && ctx->opt_join_fused_rows_pass && join_on_f16(ctx, d) &&
        ctx->opt_join_row_order && n >= 65536 && !(ctx->opt_debug & 2048)) {
      AS_TRY(oj.Hrow.alloc(ctx, (size_t)n * 128 * 2));
      AS_TRY(oj.err_by_row.alloc(ctx, n));
      rp.Hrow = oj.Hrow.p;
      rp.err_rows = oj.err_by_row.p;
    }
    AS_TRY(oj.rot_q.alloc(ctx, (size_t)31 * rotated_wide_pitch(d)));  // (kept: an index of these rows projects its query batches on them)
    AS_TRY(rotated_rows_wide(ctx, rows, norms, n, d, oj.rot_z.p, oj.rot_norm.p, &tail, rows ? nullptr : sr, oj.rot_q.p, &rp));
    if (!rp.done) oj.Hrow.release(), oj.err_by_row.release();
    ctx->stats["knn_join_prefilter_tail_e6"] = (int64_t)(tail * 1e6);
    // Measured at 1 M x 128 (scripts/time_hard.py: mixtures of rank 16 ... 128 with 4 ... 50 000 clusters, duplicated rows, plain
    // Gaussian rows): the rotated 32-wide pass was never slower than the full-width one (within 2 %) and usually twice as fast
    // (rank 64 / 2 048 clusters, tail 0.28: 56 against 122 ms; rank 96 / 1 024 clusters, tail 0.37: 50 against 114) -- a loose
    // tail term still separates clusters -- and where neither pass can rule anything out (Gaussian rows, tail 0.76: 204 against
    // 248 ms) it simply costs a quarter of the other.  Rows wider than 128 have no full-width pass at all, only the dense
    // panelled join, d / 32 times the cost of the pass (400 k x 768, rank 128, tail 0.69: 87 against 173 ms).  So the cut only
    // keeps degenerate directions out.  (Until round 3 it was 0.10; join_prefilter = 2 still forces the full-width image.)
    oj.wide_pre = ctx->opt_join_prefilter == 3 || tail <= kRotatedTailMax;
  }
  // (with the coordinates at hand the order is found on them: 32 columns instead of dim, a quarter of the pivot pass)
  const bool reorder_wide = oj.wide_pre && ctx->opt_join_row_order && n >= 65536;
  if (reorder || reorder_wide) {
    AS_TRY(oj.perm_own.alloc(ctx, n));
    if (!reorder_wide) {
      AS_TRY(need_rows());
      AS_TRY(cluster_order(ctx, rows, norms, n, d, oj.perm_own.p));
    } else {
      AS_TRY(oj.order_piv.alloc(ctx, (size_t)4096 * 128));
      AS_TRY(oj.order_starts.alloc(ctx, 4097));
      // The first pivots: 1 024, twice that beyond 3 M rows ("join_order_pivots_auto").  A pivot group of 4 k rows is 30 tiles of the
      // image: at 4 M x 128 (256 clusters) 2 048 pivots leave the cone bound 2.7 % of the tile pairs instead of 3.4 %, no far rows
      // (34 k), a quarter of the rows whose lists run over, and the grouped sample is half as large per row: 131.9 -> 118.6 ms
      // (1 536: 118.8, 3 072: 120.6, 4 096: 130.9); at 1 M and 2 M rows 1 024 is the faster (18.4 / 19.2 ms, 42.7 / 44.0 ms).
      uint32_t first_pivots = (uint32_t)std::max<int64_t>(32, ctx->opt_join_order_pivots);
      if (ctx->opt_join_order_pivots_auto && first_pivots == 1024 && n > 3000000u) first_pivots = 2048;
      // (a build sharded by rows: every rank holds all coordinates by now; the pivots of its own rows are found here and the
      // assignments exchanged -- "sharded_order_assign", row_order.hip)
      AS_TRY(cluster_order(ctx, oj.rot_z.p, oj.rot_norm.p, n, 32, oj.perm_own.p, oj.order_piv.p,
                           first_pivots, oj.order_starts.p, &oj.order_np, 4096,
                           &oj.far_pos, &oj.far_count, &oj.order_near, (sr && !rows && ctx->opt_sharded_order_assign) ? sr : nullptr));
    }
    AS_TRY(oj.theta_pos.alloc(ctx, n));
    AS_TRY(oj.theta2_pos.alloc(ctx, n));
    reorder = true;
  }
  ctx->stats["knn_join_row_order"] = reorder ? 1 : 0;
  if (!join_on_f16(ctx, d)) oj.f16_err_pos.release();
  bool rows_packed_once = false;
  if (join_on_f16(ctx, d)) {  // the f16 image serves the thresholds and the join
    const uint32_t tiles = packed_tiles_f16(n);
    AS_TRY(oj.Phown.alloc(ctx, packed_vecs_f16(tiles, d) * 16));
    if (sr && !rows && d <= 128 && rp.done) {
      // sharded by rows, the own rows packed row-major by the pass that computed their coordinates: all-gather 256 B a row, the
      // largest plane error over the ranks, and the image in the order of the join re-arranged from whole rows
      AS_TRY(sr->gather(oj.Hrow.p, (size_t)128 * 2));
      AS_TRY(sharded_max_plane_error(ctx, *sr, rp.f16_err));
      ctx->rowmajor_image = oj.Hrow.p;
      AS_TRY(pack_f16_from_rows(ctx, oj.Hrow.p, n, reorder ? oj.perm_own.p : nullptr, n, oj.Phown.p, tiles));
      // every row's own plane error crosses the links too (4 B a row): the pruning of the re-rank bounds each pair by its two rows'
      // errors as the one-GPU build does (until round 6 the ranks exchanged the largest error only -- one bound for all pairs, a
      // third more survivors and six times the rows left open)
      // (kept BY ROW: the lists of a sharded build hold row ids, and the pruning kernel reads a row's own error by its id)
      AS_TRY(sr->gather(oj.err_by_row.p, sizeof(float)));
      oj.f16_err_pos.release();
      oj.sharded_rows = true;
      rows_packed_once = true;
    } else if (sr && !rows && d <= 128) {
      // sharded by rows: every rank packs its own rows, the image of all rows in ROW order is all-gathered (256 B per row at 128
      // columns), and the image in the order of the join is that one re-arranged -- the same bytes pack_f16_all stores
      oj.Hrow.release(), oj.err_by_row.release();
      if (reorder) {
        AS_TRY(oj.Prow.alloc(ctx, packed_vecs_f16(tiles, d) * 16));
        AS_TRY(pack_f16_rows_sharded(ctx, *sr, norms, n, d, oj.Prow.p, tiles));
        AS_TRY(pack_f16_from_image(ctx, oj.Prow.p, n, oj.perm_own.p, n, d, oj.Phown.p, tiles));
      } else {
        AS_TRY(pack_f16_rows_sharded(ctx, *sr, norms, n, d, oj.Phown.p, tiles));
      }
      oj.sharded_rows = true;
      oj.f16_err_pos.release();  // (the ranks exchange the largest plane error only: one bound for all pairs)
    } else if (rp.done && reorder) {
      // the rows were packed in ROW order by the pass that computed their coordinates: the image in the order of the join is that
      // one re-arranged (the bytes pack_f16_all stores), the plane errors follow the same permutation
      AS_TRY(oj.f16_err_pos.alloc(ctx, n));
      AS_TRY(pack_f16_from_rows(ctx, oj.Hrow.p, n, oj.perm_own.p, n, oj.Phown.p, tiles));
      AS_TRY(gather_f32_by_ids(ctx, oj.err_by_row.p, oj.perm_own.p, n, oj.f16_err_pos.p));
      ctx->f16_err = rp.f16_err;
      oj.err_by_row.release();
      ctx->rowmajor_image = oj.Hrow.p;  // (the samples of the thresholds below are copied from it: 256 B a row instead of 1 KB)
      rows_packed_once = true;
    } else {
      if (sr) AS_TRY(need_rows());
      oj.Hrow.release(), oj.err_by_row.release();
      AS_TRY(oj.f16_err_pos.alloc(ctx, n));  // the rows' plane errors by image position: the pruning's bound per pair
      AS_TRY(pack_f16_all(ctx, rows, norms, n, d, oj.Phown.p, tiles, reorder ? oj.perm_own.p : nullptr, oj.f16_err_pos.p));
    }
  }
  ctx->stats["knn_join_fused_rows_pass"] = rows_packed_once ? 1 : 0;
  ctx->stats["sharded_prepare_by_rows"] = oj.sharded_rows ? 1 : 0;
  // the f16 form of any row by id: the rows row-major (the fused pass; pack_f16_from_image knows by ctx->rowmajor_image), or the
  // all-gathered fragment image of a build sharded by rows
  const void *row_image = ctx->rowmajor_image ? ctx->rowmajor_image : (oj.sharded_rows ? (reorder ? oj.Prow.p : oj.Phown.p) : nullptr);

  // thresholds: with the order by 1024 pivots of the coordinates, against the sampled rows of the 16 groups nearest to a row's
  // own (1.7 % of the products of the plain sample, the same thresholds for 99.5 % of the rows; mfma_theta_f16.hip)
  const bool wide_grouped = d > 128 && join_on_f16(ctx, d) && pos_begin == 0 && (pos_count == 0 || pos_count == n) && !gather;  // (rows wider than 128: one GPU)
  // Whatever pass the sampled thresholds come from (grouped sample, plain sample), in the order of the image they are raised to what
  // the row's own neighbourhood certifies: K other rows of its super tile and the ones beside it reach it, so the row is certified
  // by construction and its list is cut near its K-th best cosine instead of its 128th (mfma_theta_f16.hip,
  // join_thresholds_f16_band); the sampled threshold stays for rows whose neighbours live elsewhere.  Computed first: the check of
  // the grouped thresholds below looks at the thresholds the join will use.
  // ("join_theta_band" 2, the default, also means: 3 or 4 where the clusters the row order found hold more rows than 2 super tiles
  // on either side cover -- row_order.hip, knn_join_order_cluster_rows)
  uint32_t band_tiles = (uint32_t)std::max<int64_t>(0, ctx->opt_join_theta_band);
  if (band_tiles == 2 && ctx->opt_join_theta_band_auto && reorder_wide) {
    const int64_t cr = ctx->stats.count("knn_join_order_cluster_rows") ? ctx->stats["knn_join_order_cluster_rows"] : 0;
    // (the estimate runs similar clusters together along the chain: 8.8 k for the bench mixture's 3.9 k-row clusters, 38 k for 15.6 k.
    // Measured optima: 2 below ~4 k estimated rows -- 400 k x 384, the mixtures of 1 024 clusters and more --, 3 at 9 k -- 1 M x 128
    // and x 768 --, 4 at 17 k and beyond -- 2 M and 4 M x 128.)
    band_tiles = cr >= 16000 ? 4u : (cr >= 6000 ? 3u : 2u);
  }
  ctx->stats["knn_join_theta_band_tiles"] = band_tiles;
  const bool have_band = reorder && oj.Phown.p && band_tiles > 0 && (d <= 128 || wide_grouped) && K + 1 <= 32;
  DevBuf<float> tband;
  if (have_band) {
    AS_TRY(tband.alloc(ctx, n));
    AS_HIP(ctx, hipMemsetAsync(tband.p, 0xFF, sizeof(float) * n, ctx->stream));  // (NaN where not computed: never the larger one)
    if (d > 128) AS_TRY(join_thresholds_f16_wide_band(ctx, n, d, oj.Phown.p, band_tiles, K + 1, tband.p));
    else AS_TRY(join_thresholds_f16_band(ctx, n, d, oj.Phown.p, band_tiles, K + 1, tband.p, pos_begin, pos_count));
  }
  auto merge_band = [&]() -> int {  // (own positions; the callers gather afterwards)
    if (have_band && pos_count)
      hipLaunchKernelGGL(theta_band_merge_kernel, dim3(cdiv(pos_count, 256)), dim3(256), 0, ctx->stream, oj.theta_pos.p + pos_begin,
                         (const float *)tband.p + pos_begin, pos_count, (float)(2.0 * join_delta(ctx, d) + 2e-6),
                         (float)((double)std::max<int64_t>(0, ctx->opt_join_theta_band_slack) * join_delta(ctx, d)));
    AS_HIP(ctx, hipGetLastError());
    return AS_OK;
  };
  if (reorder_wide && (d <= 128 || wide_grouped) && oj.order_np >= 256 && ctx->opt_join_theta_grouped && ctx->opt_join_bf16 &&
      !(ctx->opt_debug & 32) && ctx->opt_join_sample <= 0) {
    const int64_t rank_opt = ctx->opt_join_rank > 0 ? ctx->opt_join_rank : 4;
    const uint32_t m = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(rank_opt, 32)), m2 = std::min<uint32_t>(48, 2 * m + 6);
    if (d > 128)
      AS_TRY(join_thresholds_f16_wide_grouped(ctx, rows, norms, n, d, oj.Phown.p, oj.perm_own.p, oj.order_starts.p, oj.order_np, oj.order_piv.p, 32,
                                              32, m, oj.theta_pos.p, m2, oj.theta2_pos.p, oj.order_near.p));
    else
    AS_TRY(join_thresholds_f16_grouped(ctx, rows, norms, n, d, oj.Phown.p, oj.perm_own.p, oj.order_starts.p, oj.order_np, oj.order_piv.p, 32, 32, m,
                                       oj.theta_pos.p, m2, oj.theta2_pos.p, pos_begin, pos_count, row_image, oj.order_near.p));
    AS_TRY(merge_band());
    if (gather) {
      AS_TRY((*gather)(oj.theta_pos.p, sizeof(float)));
      AS_TRY((*gather)(oj.theta2_pos.p, sizeof(float)));
    }
    // Rows far from their pivot (row_order.hip: rows of a cluster that got no pivot, scattered over the groups of other clusters):
    // neither their group's sample nor their band holds their neighbours, their grouped threshold is that of a far cluster and each
    // of them lists thousands of pairs (rank-128 mixture of 4 096 clusters: 0.4 % of the rows, and the f16 pass of the join visits
    // five times the blocks).  They take the larger of it and the threshold against the WHOLE sample, like the rows of the check below.
    // pos: image positions, whole 512s, ~0 where unused.
    auto raise_from_whole_sample = [&](const uint32_t *pos, uint32_t room) -> int {
      const uint32_t ns_far = std::max(256u, (n / 32u) & ~255u);
      DevBuf<uint32_t> far_ids, far_sid;
      DevBuf<float> far_t1, far_t2;
      DevBuf<unsigned char> far_img;
      AS_TRY(far_ids.alloc(ctx, room));
      AS_TRY(far_sid.alloc(ctx, ns_far));
      AS_TRY(far_t1.alloc(ctx, room));
      AS_TRY(far_t2.alloc(ctx, room));
      AS_TRY(far_img.alloc(ctx, packed_vecs_f16(room / kTileRows, d) * 16));
      hipLaunchKernelGGL(rows_of_positions_kernel, dim3(cdiv(room, 256)), dim3(256), 0, ctx->stream, pos, (const uint32_t *)oj.perm_own.p, room, far_ids.p);
      hipLaunchKernelGGL(theta_check_sample_kernel, dim3(cdiv(ns_far, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)oj.perm_own.p, n, ns_far, far_sid.p);
      if (row_image) AS_TRY(pack_f16_from_image(ctx, row_image, n, far_ids.p, room, d, far_img.p, room / kTileRows));
      else AS_TRY(pack_f16(ctx, rows, norms, far_ids.p, 0, room, d, far_img.p, room / kTileRows));
      if (d > 128) AS_TRY(join_thresholds_f16_wide(ctx, rows, norms, d, far_img.p, 0, room, far_sid.p, ns_far, m, far_t1.p, m2, far_t2.p));
      else AS_TRY(join_thresholds_f16(ctx, rows, norms, d, far_img.p, 0, room, far_sid.p, ns_far, m, far_t1.p, m2, far_t2.p, row_image, n));
      hipLaunchKernelGGL(theta_raise_kernel, dim3(cdiv(room, 256)), dim3(256), 0, ctx->stream, oj.theta_pos.p, pos, (const float *)far_t1.p, room);
      hipLaunchKernelGGL(theta_raise_kernel, dim3(cdiv(room, 256)), dim3(256), 0, ctx->stream, oj.theta2_pos.p, pos, (const float *)far_t2.p, room);
      AS_HIP(ctx, hipGetLastError());
      return AS_OK;
    };
    if (oj.far_count && ctx->opt_join_theta_far) {
      AS_TRY(raise_from_whole_sample(oj.far_pos.p, (oj.far_count + 511u) / 512u * 512u));
      ctx->stats["knn_join_theta_far_rows"] = oj.far_count;
    } else {
      ctx->stats["knn_join_theta_far_rows"] = 0;
    }
    // ... and so do the rows whose threshold lies far below those around them: in an image ordered by direction the 64 positions of
    // a block are rows of one neighbourhood with thresholds alike, and one that sits "join_theta_dip_e3" (50: 0.05) below the
    // block's median is a row whose neighbours the sample of its group missed -- whatever the reason.  A row flagged without need
    // costs its share of one more pass and keeps its threshold.
    ctx->stats["knn_join_theta_dip_rows"] = 0;
    // (rows of 128 columns or fewer: the panelled pass of wider rows takes 3 ms for the 700 rows this finds at 1 M x 768 and the lists
    // of those rows stay as long)
    if (ctx->opt_join_theta_dip_e3 > 0 && ctx->opt_join_theta_far && d <= 128) {
      DevBuf<uint32_t> dflag, dpos;
      DevBuf<uint64_t> drank;
      AS_TRY(dflag.alloc(ctx, n));
      AS_TRY(drank.alloc(ctx, (size_t)n + 1));
      hipLaunchKernelGGL(theta_dip_flag_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, (const float *)oj.theta_pos.p, n,
                         (float)ctx->opt_join_theta_dip_e3 * 1e-3f, dflag.p);
      AS_TRY(exclusive_scan_u32_u64(ctx, dflag.p, n, drank.p));
      uint64_t h_dip = 0;
      AS_HIP(ctx, copy_d2h(ctx, &h_dip, drank.p + n, sizeof(h_dip)));
      ctx->stats["knn_join_theta_dip_rows"] = (int64_t)h_dip;
      if (h_dip > 0 && h_dip * 100 <= (uint64_t)n * 8) {
        const uint32_t room = (uint32_t)((h_dip + 511) / 512 * 512);
        AS_TRY(dpos.alloc(ctx, room));
        AS_HIP(ctx, hipMemsetAsync(dpos.p, 0xFF, sizeof(uint32_t) * room, ctx->stream));
        hipLaunchKernelGGL(list_flagged_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)dflag.p, (const uint64_t *)drank.p, n, room, dpos.p);
        AS_TRY(raise_from_whole_sample(dpos.p, room));
      }
    }
    if (ctx->opt_debug & 65536) {  // diagnostic: a checksum of the thresholds (are they the same from run to run?)
      DevBuf<unsigned long long> cs;
      AS_TRY(cs.alloc(ctx, 1));
      AS_HIP(ctx, hipMemsetAsync(cs.p, 0, sizeof(unsigned long long), ctx->stream));
      hipLaunchKernelGGL(theta_checksum_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, (const float *)oj.theta_pos.p, n, cs.p);
      unsigned long long h = 0;
      AS_HIP(ctx, copy_d2h(ctx, &h, cs.p, sizeof(h)));
      ctx->stats["knn_join_theta_checksum"] = (int64_t)(h & 0x7FFFFFFFFFFFFFFFull);
    }
    // How good they are depends on the data (do the pivots cover its clusters?), so they are CHECKED: 4096 rows spread over
    // the image get their threshold against the whole sample as well (0.1 ms), and if more than 1.0 % of them (0.6 % until round 4) sit more than
    // 5e-3 below it (rows whose neighbours live in groups the sample of their own group does not reach: every one of them
    // lists thousands of pairs) the grouped thresholds are dropped for the plain pass.  (1 M x 128 bench mixture, 1024
    // pivots: 0.24 %; 700 k rows: 0.27 %; 400 k rows, 768 pivots: 2.9 %; 200 k rows: 4.1 %.)
    const uint32_t n_chk = 4096, stride = n / n_chk;
    DevBuf<uint32_t> chk_ids, chk_bad, chk_sid;
    DevBuf<float> chk_t1, chk_t2;
    DevBuf<unsigned char> chk_img;
    const uint32_t ns_all = std::max(256u, (n / 32u) & ~255u);
    AS_TRY(chk_ids.alloc(ctx, n_chk));
    AS_TRY(chk_bad.alloc(ctx, 1));
    AS_TRY(chk_sid.alloc(ctx, ns_all));
    AS_TRY(chk_t1.alloc(ctx, n_chk));
    AS_TRY(chk_t2.alloc(ctx, n_chk));
    AS_TRY(chk_img.alloc(ctx, packed_vecs_f16(n_chk / kTileRows, d) * 16));
    AS_HIP(ctx, hipMemsetAsync(chk_bad.p, 0, sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(theta_check_ids_kernel, dim3(cdiv(n_chk, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)oj.perm_own.p, stride,
                       n_chk, chk_ids.p);
    hipLaunchKernelGGL(theta_check_sample_kernel, dim3(cdiv(ns_all, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)oj.perm_own.p, n,
                       ns_all, chk_sid.p);  // (the same sampled positions, all of them: what differs is the restriction alone)
    if (row_image) AS_TRY(pack_f16_from_image(ctx, row_image, n, chk_ids.p, n_chk, d, chk_img.p, n_chk / kTileRows));
    else AS_TRY(pack_f16(ctx, rows, norms, chk_ids.p, 0, n_chk, d, chk_img.p, n_chk / kTileRows));
    if (d > 128) AS_TRY(join_thresholds_f16_wide(ctx, rows, norms, d, chk_img.p, 0, n_chk, chk_sid.p, ns_all, m, chk_t1.p, m2, chk_t2.p));
    else AS_TRY(join_thresholds_f16(ctx, rows, norms, d, chk_img.p, 0, n_chk, chk_sid.p, ns_all, m, chk_t1.p, m2, chk_t2.p, row_image, n));
    hipLaunchKernelGGL(theta_check_count_kernel, dim3(cdiv(n_chk, 256)), dim3(256), 0, ctx->stream, (const float *)oj.theta_pos.p, stride,
                       (const float *)chk_t1.p, n_chk, 5e-3f, chk_bad.p);
    uint32_t h_bad = 0;
    AS_TRY(read_u32(ctx, chk_bad.p, &h_bad));
    ctx->stats["knn_join_theta_grouped_bad"] = h_bad;
    // (the plain pass grows with the square of the table and what the rows below it add with the table: the share allowed grows
    // with it -- 4 M rows: 0.66 % of the checked rows below, the plain pass 116 ms of 530)
    const uint32_t bad_per_mille = (uint32_t)std::max<int64_t>(0, ctx->opt_join_theta_grouped_bad_e3) * std::min(4u, std::max(1u, n / 1000000u));
    if (h_bad * 1000u <= bad_per_mille * n_chk) {
      oj.have_theta2 = true;
      oj.grouped_theta = true;
    } else {
      if (sr) AS_TRY(need_rows());  // (every rank takes this branch together: the check reads the same thresholds everywhere)
      if (pos_count)
        AS_TRY(knn_join_thresholds(ctx, rows, norms, n, d, pos_begin, pos_count, oj.theta_pos.p + pos_begin, nullptr, oj.theta2_pos.p + pos_begin,
                                   &oj.have_theta2, oj.Phown.p));
      else oj.have_theta2 = true;  // (a rank without positions: what the ranks with positions find, the 16-bit pass always has them)
      AS_TRY(merge_band());
      if (gather) {
        AS_TRY((*gather)(oj.theta_pos.p, sizeof(float)));
        AS_TRY((*gather)(oj.theta2_pos.p, sizeof(float)));
      }
    }
    ctx->stats["knn_join_theta_grouped"] = oj.grouped_theta ? 1 : 0;
  } else {
    ctx->stats["knn_join_theta_grouped"] = 0;
    float *t1 = reorder ? oj.theta_pos.p : oj.theta_own.p, *t2 = reorder ? oj.theta2_pos.p : oj.theta2_own.p;
    if (sr) AS_TRY(need_rows());
    if (pos_count)
      AS_TRY(knn_join_thresholds(ctx, rows, norms, n, d, pos_begin, pos_count, t1 + pos_begin, nullptr, t2 + pos_begin, &oj.have_theta2, oj.Phown.p));
    AS_TRY(merge_band());  // (have_band implies reorder: t1 is theta_pos)
    if (gather) {
      // (whether the pass produced second thresholds depends on the options and the width alone: every rank agrees; a rank
      // without positions takes the answer of the 16-bit passes, which always have them)
      if (!pos_count) oj.have_theta2 = ctx->opt_join_bf16 != 0;
      AS_TRY((*gather)(t1, sizeof(float)));
      if (oj.have_theta2) AS_TRY((*gather)(t2, sizeof(float)));
    }
  }
  if (have_band) {
    // kept by row for the second pass: a row the sampled threshold left open has K rows above this one by construction
    AS_TRY(oj.theta_band_own.alloc(ctx, n));
    AS_TRY(scatter_by_perm(ctx, tband.p, oj.perm_own.p, n, oj.theta_band_own.p));
  } else {
    oj.theta_band_own.release();
  }
  if (reorder) {  // ... and go back to row order for everything behind the join
    AS_TRY(scatter_by_perm(ctx, oj.theta_pos.p, oj.perm_own.p, n, oj.theta_own.p));
    if (oj.have_theta2) AS_TRY(scatter_by_perm(ctx, oj.theta2_pos.p, oj.perm_own.p, n, oj.theta2_own.p));
  }
  if (join_on_bf16(ctx, d)) {
    if (sr) AS_TRY(need_rows());
    AS_TRY(oj.P16own.alloc(ctx, packed_vecs_bf16(b_alloc, d) * 16));
    AS_TRY(pack_bf16(ctx, rows, norms, nullptr, 0, n, d, oj.P16own.p, b_alloc));
  }
  oj.perm = oj.reorder ? oj.perm_own.p : nullptr;
  oj.theta_image = oj.reorder ? oj.theta_pos.p : oj.theta_own.p;
  return AS_OK;
}

// Contracts the tile pairs owned by the 512-position super tiles of [pos_begin, pos_begin + pos_count) of the prepared image.
// cnt (n) / cand (n x cap slots of two words): lists_by_pos = indexed by image position and holding positions, else indexed
// by row and holding rows (the join kernel translates where it appends).
int own_join_pairs(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, uint32_t pos_begin, uint32_t pos_count,
                   OwnJoin &oj, uint32_t *cnt, uint32_t *cand, bool lists_by_pos) {
  // (the directions were tried above and do not carry the rows: the first pass, if any, takes the full-width image without
  // looking for them again)
  const int64_t saved_mode = ctx->opt_join_prefilter;
  if (oj.rot_z.p && !oj.wide_pre && saved_mode == 1) ctx->opt_join_prefilter = 2;
  // (where the pivot groups of the image begin: the cone bound may split its tiles there)
  oj.cone.group_start = (oj.reorder && oj.order_np && oj.order_starts.p) ? oj.order_starts.p : nullptr;
  oj.cone.groups = oj.cone.group_start ? oj.order_np : 0;
  const int rc_pairs = knn_join_pairs(ctx, rows, norms, n, d, pos_begin, pos_count, oj.theta_image, oj.cap, cnt, cand, nullptr, oj.P16own.p,
                                      oj.Phown.p, (oj.reorder && !lists_by_pos) ? oj.perm_own.p : nullptr,
                                      oj.reorder ? oj.perm_own.p : nullptr, oj.wide_pre ? oj.rot_z.p : nullptr, &oj.cone);
  ctx->opt_join_prefilter = saved_mode;
  return rc_pairs;
}

// ---- K2 driver ----------------------------------------------------------------------------------
int knn_mfma(as_ctx *ctx, const double *rows, const double *norms, uint64_t n64, uint64_t d64, uint64_t row_begin,
             uint64_t row_count64, uint32_t K, int rectify, uint32_t *out_idx, double *out_dist, const JoinSources *js) {
  const uint32_t n = (uint32_t)n64, d = (uint32_t)d64, row_count = (uint32_t)row_count64;
  const int NC = (int)((d + 31) / 32);
  // one extra slot: a row's own index is the best candidate of its list and is dropped by the re-rank
  const uint32_t KC = cand_capacity(ctx, K + 1);
  const bool wide = NC > kMaxNC;  // dim > 128: only the f32 pass has the k-panel loop
  const bool use_f32 = (ctx->opt_cand_precision != 64 || wide) && fits_f32(KC);
  // deep tables (K beyond what the top-k kernels keep in LDS: KC > 40) still take the 16-bit join -- its lists hold 512
  // candidates whatever K is -- and hand what it leaves open straight to the exact scans
  const bool stages_fit = !(wide && !use_f32) && fits_f64(KC);
  const bool deep_join = !stages_fit && !wide && K <= 64 && ctx->opt_cand_precision != 64 &&
                         (join_on_f16(ctx, d) || join_on_bf16(ctx, d)) &&
                         (js != nullptr || (row_begin == 0 && row_count == n && join_supported(ctx, n, d, 0)));
  if ((!stages_fit && !deep_join) || n < 4 * kTileRows || row_count == 0 || !rerank_fits(d))  // small, deep or very wide problems: exact scan
    return knn_exact(ctx, rows, norms, n, d, (js && js->a_ids) ? js->a_ids : nullptr, (js && js->a_ids) ? 0 : row_begin, row_count, K, rectify,
                     out_idx, out_dist, 0);

  const uint32_t b_tiles = (n + kTileRows - 1) / kTileRows;
  const uint32_t b_alloc = (b_tiles + 1) & ~1u;  // even: the f32 pass reads A tiles in pairs

  // rows whose pairs may hit the 0 < |a||b| <= 1e-12 rule are always re-done exactly
  DevBuf<unsigned long long> minb;
  DevBuf<uint8_t> force;
  AS_TRY(minb.alloc(ctx, 1));
  AS_TRY(force.alloc(ctx, n));
  hipLaunchKernelGGL(set_u64_kernel, dim3(1), dim3(1), 0, ctx->stream, minb.p, 0x7FF0000000000000ull);  // +inf
  hipLaunchKernelGGL(min_norm_kernel, dim3(256), dim3(256), 0, ctx->stream, norms, n, minb.p);
  hipLaunchKernelGGL(flag_tiny_norm_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, norms, n, (const unsigned long long *)minb.p,
                     force.p);

  DevBuf<uint32_t> fail1, fail_count;
  DevBuf<int> err;
  AS_TRY(fail1.alloc(ctx, row_count));
  AS_TRY(fail_count.alloc(ctx, 1));
  AS_TRY(err.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(fail_count.p, 0, sizeof(uint32_t), ctx->stream));
  AS_HIP(ctx, hipMemsetAsync(err.p, 0, sizeof(int), ctx->stream));
  GraphDistPolicy pol{norms, rectify};

  CandProblem pb{};
  pb.mode = 0;
  pb.A = rows;
  pb.a_norms = norms;
  pb.arows = ARows{nullptr, (uint32_t)row_begin};
  pb.nA = row_count;
  pb.B = rows;
  pb.b_norms = norms;
  pb.nB = n;
  pb.d = d;
  pb.KC = KC;

  // ---- stage 1 over the contiguous row range ----
  // The bulk of the rows runs as whole rounds of 256 workgroups; a remainder that would leave most of
  // the chip idle for a full sweep is launched separately with its column range split (tail quantisation).
  PackedF32 B32, A32;
  PackedF64 B64, A64;
  uint32_t n_fail = 0;
  // dim <= 128: symmetric threshold join (half the contractions).  Either the caller has run the join and
  // exchanged the candidate lists (row-sharded build: js), or the range is the whole matrix and it runs here.
  const bool own_join = !js && (use_f32 || deep_join) && row_begin == 0 && row_count == n &&
                        join_supported(ctx, n, d, 0);
  const bool use_join = js != nullptr || own_join;
  ctx->stats["knn_join"] = use_join ? 1 : 0;
  if (!js) ctx->stats["knn_join_prefilter"] = 0;  // (not a leftover of the call before: only the 16-bit joins set it; with js the caller has just run the join)
  JoinSources local{};  // outlives the block below: the later stages still read js->packed
  // own join: the images, the order, the rotated coordinates and the thresholds (own_join_prepare) stay for the later stages;
  // a build sharded by image positions (csrc/sharded.hip) has prepared the same thing and hands it in with the lists
  OwnJoin oj_local;
  const OwnJoin *oj = js ? js->oj : nullptr;
  bool lists_by_pos = false;            // the lists of the own join are indexed by (and hold) image positions
  bool rerank_order_rows = false;       // lists by row id, but the re-rank visits the rows in the order of the image (wide rows)
  DevBuf<uint32_t> cnt_own, cand_own;
  if (use_join) {
    if (own_join) {
      AS_TRY(own_join_prepare(ctx, rows, norms, n, d, K, oj_local));
      if (oj_local.reorder) {  // kept for the caller (as_build_laplacian hands it to the graph: common.h, join_order)
        if (ctx->join_order_cap < n) {
          if (ctx->join_order) (void)hipFree(ctx->join_order);
          ctx->join_order = nullptr;
          ctx->join_order_cap = 0;
          if (hipMalloc((void **)&ctx->join_order, sizeof(uint32_t) * n) == hipSuccess) ctx->join_order_cap = n;
          else (void)hipGetLastError();
        }
        if (ctx->join_order_cap >= n) {
          AS_HIP(ctx, hipMemcpyAsync(ctx->join_order, oj_local.perm_own.p, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, ctx->stream));
          ctx->join_order_n = n;
          ++ctx->join_order_seq;
        }
      }
      const uint32_t cap = oj_local.cap;
      AS_TRY(cnt_own.alloc(ctx, n));
      AS_TRY(cand_own.alloc(ctx, (size_t)n * cap * 2));  // two words per slot: row id, approximate cosine
      // 128 columns: the lists stay in image positions and the pruning kernel of the re-rank translates the ~30 survivors of
      // a row (rerank_split.h); other widths: the join kernel translates every entry where it appends it
      lists_by_pos = oj_local.reorder && d == 128 && !(ctx->opt_debug & 16) && ((uintptr_t)rows & 15) == 0 &&
                     ctx->opt_debug != 5;
      // other widths with the image in the library's own order: the lists hold row ids, but the re-rank still visits the rows in the
      // order of the image (rerank_order_rows below), for which it needs row -> place as well
      rerank_order_rows = oj_local.reorder && !lists_by_pos && d != 128 && ctx->opt_rerank_image_order && !(ctx->opt_debug & 16) && ctx->opt_debug != 5;
      if (lists_by_pos || rerank_order_rows) {
        AS_TRY(oj_local.inv_own.alloc(ctx, n));
        hipLaunchKernelGGL(invert_perm_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, oj_local.perm_own.p, n, oj_local.inv_own.p);
      }
      // "keep_join_images": the order and the images stay with the context for an index of these rows (common.h: as_ctx::JoinKeep)
      join_keep_clear(ctx);
      // (from 2^18 rows: at 100 k items the scan of 10 k queries over the image as it is takes 1.33 ms, behind the first pass 1.53)
      const bool keeping = ctx->opt_keep_join_images && n >= (uint32_t)std::max<int64_t>(65536, ctx->opt_keep_join_images_min_rows) && lists_by_pos &&
                           oj_local.wide_pre && oj_local.rot_q.p && oj_local.order_piv.p &&
                           oj_local.order_np && oj_local.Phown.p && is_device_ptr(rows);
      const uint32_t keep_tiles = packed_tiles_f16(n);
      if (keeping) {
        as_ctx::JoinKeep &jk = ctx->join_keep;
        AS_HIP(ctx, handle_malloc(&jk.packed8, rotated_image_bytes(keep_tiles)));
        ctx->keep_p8_dst = jk.packed8;
        ctx->keep_p8_done = false;
      }
      {
        const int rc_pairs = own_join_pairs(ctx, rows, norms, n, d, 0, n, oj_local, cnt_own.p, cand_own.p, lists_by_pos);
        ctx->keep_p8_dst = nullptr;
        if (rc_pairs != AS_OK) {
          join_keep_clear(ctx);
          return rc_pairs;
        }
      }
      if (keeping && ctx->keep_p8_done) {
        as_ctx::JoinKeep &jk = ctx->join_keep;
        const size_t img_bytes = (size_t)packed_vecs_f16(keep_tiles, d) * 16, piv_bytes = sizeof(float) * (size_t)oj_local.order_np * 32,
                     q_bytes = sizeof(float) * (size_t)31 * rotated_wide_pitch(d);
        bool ok = handle_malloc(&jk.packedh, img_bytes) == hipSuccess && handle_malloc((void **)&jk.perm, sizeof(uint32_t) * (size_t)n) == hipSuccess &&
                  handle_malloc((void **)&jk.piv, piv_bytes) == hipSuccess && handle_malloc((void **)&jk.rot_qw, q_bytes) == hipSuccess &&
                  handle_malloc((void **)&jk.norms, sizeof(double) * (size_t)n) == hipSuccess;
        ok = ok && hipMemcpyAsync(jk.packedh, oj_local.Phown.p, img_bytes, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess &&
             hipMemcpyAsync(jk.perm, oj_local.perm_own.p, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess &&
             hipMemcpyAsync(jk.piv, oj_local.order_piv.p, piv_bytes, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess &&
             hipMemcpyAsync(jk.rot_qw, oj_local.rot_q.p, q_bytes, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess &&
             hipMemcpyAsync(jk.norms, norms, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess;
        if (ok) {
          jk.rows = rows;
          jk.n = n;
          jk.d = d;
          jk.np = oj_local.order_np;
          jk.rot_err = ctx->keep_p8_err;
          jk.f16_err = ctx->f16_err;
          jk.complete = true;
        } else {
          (void)hipGetLastError();
          join_keep_clear(ctx);  // (no room: the index finds its own)
        }
      } else if (keeping) {
        join_keep_clear(ctx);
      }
      ctx->stats["knn_join_images_kept"] = ctx->join_keep.complete ? 1 : 0;
      if ((ctx->opt_debug & 0xC0FF & ~(int64_t)(8 | 16 | 32 | 64)) || (ctx->opt_debug & 0xE000000)) return AS_OK;  // timing experiment (ablation bits): the join only, outputs undefined
      local = JoinSources{nullptr, oj_local.theta_own.p, 1, cap, cnt_own.p, cand_own.p, oj_local.have_theta2 ? oj_local.theta2_own.p : nullptr,
                          oj_local.P16own.p, oj_local.Phown.p};
      js = &local;
      oj = &oj_local;
    }
    DevBuf<unsigned long long> scored;
    // the rows of the range: [row_begin, row_begin + row_count), or the list js->a_ids (a range of image positions)
    const uint32_t *a_ids = js->a_ids;
    const uint32_t *ent_perm = lists_by_pos ? oj->perm_own.p : nullptr;   // list entries are image positions: position -> row
    const uint32_t *list_inv = lists_by_pos ? oj->inv_own.p : nullptr;    // ... and the list of row a sits at inv[a]
    {
      const uint32_t cmax = std::min<uint32_t>(js->n_src * js->cap, std::max<uint32_t>(512, js->cap));
      GraphBound bnd{join_delta(ctx, d), rectify};
      // candidates whose approximate cosine (stored with them) cannot reach the top K are dropped before any row is
      // fetched: 74 -> ~20 rows of 1 KB per row of the table (debug 16: keep them all)
      const double prune_delta = (ctx->opt_debug & 16) ? -1.0 : join_delta(ctx, d);
      ProfScope ps(ctx, "knn_rerank");
      // rows per batch x buffers.  Measured at 1 M rows (re-rank + fall-back re-rank): per-lane walk 66 ms, 8x1 39, 8x2 46,
      // 16x1 37, 16x2 58: occupancy (LDS per wave) beats double buffering.  debug = 10 + {0: 8x1, 1: 8x2, 2: 16x1, 3: 16x2}.
      // with the pruning ~20 candidates of a row are scored exactly: 8 staged rows per batch and key slots for 128
      // candidates keep three workgroups per CU whatever the list capacity (without it: 16 rows, cmax slots)
      const bool pruning = prune_delta >= 0.0;
      const int rr_rows = pruning ? 8 : 16, rr_nbuf = 1;
      const uint32_t kmax = pruning ? std::min<uint32_t>(cmax, std::max<uint32_t>(128, 2 * K)) : cmax;
      const size_t lds128 = (size_t)4 * (1024 + (size_t)rr_nbuf * rr_rows * kRrPitch + rr_key_bytes(cmax, kmax) + (size_t)cmax * 4);
      DevBuf<uint32_t> records;
      DevBuf<double> norms_pos;
      if (d % 128 == 0 && d <= 1024 && pruning && ((uintptr_t)rows & 15) == 0 && (size_t)4 * cmax * 12 <= 64 * 1024 &&
          ctx->opt_debug != 5) {
        // two kernels (rerank_split.h): lists -> pruned records, then one stream of row batches per wave
        const int slots = d == 128 ? 2 : 4;  // 124 / 252 survivors per record
        AS_TRY(records.alloc(ctx, (size_t)row_count * rs_words(slots) + kRsRows));
        AS_TRY(scored.alloc(ctx, kRsScoredSlots));
        AS_HIP(ctx, hipMemsetAsync(scored.p, 0, sizeof(unsigned long long) * kRsScoredSlots, ctx->stream));
        // the rows in the order of the image (a list of image positions is in that order already): neighbours share candidates
        // (rows wider than 128, round 5: the records used to lie in ROW order -- a row's partners anywhere among them: every lookup of
        // the link kernel missed the L2, 106 M lines for 52 M lookups at 1 M x 768, and the scoring kernel's waves shared no rows)
        const uint32_t *score_order = ((lists_by_pos || rerank_order_rows) && !a_ids && row_begin == 0 && row_count == n) ? oj->perm_own.p : nullptr;
        if (score_order || !a_ids) ctx->stats["knn_rerank_image_order"] = score_order ? 1 : 0;  // (the re-rank of a list of open rows says nothing)
        {
          // the bound per pair from the rows' own plane errors, where the lists hold image positions and the image was packed here
          const float *err_pos = (lists_by_pos && ent_perm && join_on_f16(ctx, d) && oj->f16_err_pos.p && ctx->opt_rerank_pair_bound) ? oj->f16_err_pos.p : nullptr;
          // lists that hold ROW ids (rows wider than 128 columns, images in row order): the same bound from the errors by row
          DevBuf<float> err_row;
          if (!err_pos && !lists_by_pos && oj && own_join && oj->f16_err_pos.p && join_on_f16(ctx, d) && ctx->opt_rerank_pair_bound && !a_ids &&
              js->n_src == 1) {
            if (oj->reorder) {
              AS_TRY(err_row.alloc(ctx, n));
              AS_TRY(scatter_by_perm(ctx, oj->f16_err_pos.p, oj->perm_own.p, n, err_row.p));
              err_pos = err_row.p;
            } else {
              err_pos = oj->f16_err_pos.p;
            }
          }
          // a build sharded by rows: the lists hold row ids, the rows are a list (a_ids), the errors were all-gathered by row
          const float *err_self = nullptr;
          if (!err_pos && !lists_by_pos && a_ids && oj && oj->err_by_row.p && join_on_f16(ctx, d) && ctx->opt_rerank_pair_bound)
            err_pos = err_self = oj->err_by_row.p;
          ctx->stats["knn_rerank_pair_bound"] = err_pos ? 1 : 0;
          if (ent_perm) {  // the norms by image position: a survivor's norm is asked for together with its row id
            AS_TRY(norms_pos.alloc(ctx, n));
            hipLaunchKernelGGL(gather_f64_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, norms, ent_perm, n, norms_pos.p);
          }
          // one list of candidates with a bound per pair (every item graph of 128-wide rows): a row per half wave with the list in
          // registers, the lists longer than 128 entries left to the LDS form (rerank_split.h)
          const bool halves = js->n_src == 1 && err_pos && ctx->opt_rerank_prune_halves;
          DevBuf<uint32_t> long_rows;
          auto parked = std::make_unique<DeferredStat>();
          if (halves) {
            AS_TRY(long_rows.alloc(ctx, (size_t)row_count));
            AS_TRY(parked->buf.alloc(ctx, 1));
            AS_HIP(ctx, hipMemsetAsync(parked->buf.p, 0, sizeof(unsigned long long), ctx->stream));
          }
          const float accum_f = (float)(cos_delta_f16_accum(d) * (1.0 + 1e-6)), floor_f = (float)(rect_delta(d) * (1.0 + 1e-6));
          ProfScope ps_prune(ctx, "knn_rerank_prune");
          if (halves)
            hipLaunchKernelGGL(rerank_prune_halves_kernel, dim3(std::min<uint64_t>(cdiv(row_count, 8), 256 * 32)), dim3(256), 0, ctx->stream,
                               (uint32_t)row_begin, row_count, js->theta, norms, js->cnt, js->cand, js->cap, K, force.p, rectify, records.p, slots, ent_perm,
                               list_inv, scored.p, a_ids, score_order, err_pos, accum_f, floor_f, (const double *)norms_pos.p, err_self, long_rows.p,
                               reinterpret_cast<uint32_t *>(parked->buf.p));
          hipLaunchKernelGGL(rerank_prune_kernel, dim3(std::min<uint64_t>(cdiv(row_count, 4), halves ? 1024 : 256 * 32)), dim3(256), (size_t)4 * cmax * 12,
                             ctx->stream, (uint32_t)row_begin, row_count, js->theta, norms, js->n_src, js->cnt, js->cand, js->cap, cmax, K,
                             force.p, prune_delta, rectify, records.p, slots, ent_perm, list_inv, scored.p, a_ids, score_order, err_pos,
                             accum_f, floor_f, (const double *)norms_pos.p, err_self, halves ? (const uint32_t *)long_rows.p : nullptr,
                             halves ? reinterpret_cast<const uint32_t *>(parked->buf.p) : nullptr);
          if (halves && !a_ids) {  // (statistics: read back with the next wait for the stream)
            parked->host.resize(1);
            parked->keys = {{"knn_rerank_long_lists", 0}};
            ctx->deferred.push_back(std::move(parked));
          }
        }
        AS_HIP(ctx, hipGetLastError());
        {
          // every unordered pair scored once, the ranking in a kernel of its own (rerank_pairs.h)
          const uint32_t *row2rec = score_order ? (const uint32_t *)oj->inv_own.p : nullptr;
          if (d == 128) AS_TRY(rerank_pairs_run<2>(ctx, rows, d, n, (uint32_t)row_begin, row_count, records.p, K, rectify, bnd, out_dist, out_idx, fail1.p, fail_count.p, a_ids, score_order, row2rec));
          else AS_TRY(rerank_pairs_run<4>(ctx, rows, d, n, (uint32_t)row_begin, row_count, records.p, K, rectify, bnd, out_dist, out_idx, fail1.p, fail_count.p, a_ids, score_order, row2rec));
        }
      } else if (lists_by_pos) {
        return fail(ctx, AS_ERR_HIP, "internal: lists by image position need the two-kernel re-rank");
      } else if (d == 128 && lds128 <= 160 * 1024 && ((uintptr_t)rows & 15) == 0 && ctx->opt_debug != 5) {
#define AS_RR(R, NB)                                                                                                       \
  AS_HIP(ctx, hipFuncSetAttribute((const void *)rerank_join_d128_kernel<GraphDistPolicy, GraphBound, R, NB>,               \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds128));                               \
  hipLaunchKernelGGL((rerank_join_d128_kernel<GraphDistPolicy, GraphBound, R, NB>),                                        \
                     dim3(std::min<uint64_t>(cdiv(row_count, 4), 256 * 16)), dim3(256), lds128, ctx->stream, rows, rows,   \
                     a_ids, (uint32_t)row_begin, row_count, a_ids ? js->theta_local : js->theta, js->n_src, js->cnt, js->cand, js->cap, cmax, K, force.p, pol, \
                     bnd, prune_delta, rectify, kmax, out_dist, out_idx, fail1.p, fail_count.p, err.p)
        if (rr_rows == 8) { AS_RR(8, 1); }
        else { AS_RR(16, 1); }
#undef AS_RR
      } else {
        const size_t lds = (size_t)4 * (d + cmax) * sizeof(double) + (size_t)4 * cmax * sizeof(uint32_t);
        AS_HIP(ctx, hipFuncSetAttribute((const void *)rerank_join_kernel<GraphDistPolicy, GraphBound>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((rerank_join_kernel<GraphDistPolicy, GraphBound>), dim3(std::min<uint64_t>(cdiv(row_count, 4), 8192)),
                           dim3(256), lds, ctx->stream, rows, rows, a_ids, d, (uint32_t)row_begin, row_count,
                           a_ids ? js->theta_local : js->theta, js->n_src, js->cnt,
                           js->cand, js->cap, cmax, K, force.p, pol, bnd, prune_delta, rectify, out_dist, out_idx, fail1.p,
                           fail_count.p, err.p);
      }
      AS_HIP(ctx, hipGetLastError());
    }
    {  // statistics: candidates per row over all sources -- and, in the same read-back (each one is a wait for the stream), the
      // survivors of the pruning and the number of rows left open
      DevBuf<unsigned long long> st;
      AS_TRY(st.alloc(ctx, 5));
      AS_HIP(ctx, hipMemsetAsync(st.p, 0, 5 * sizeof(unsigned long long), ctx->stream));
      hipLaunchKernelGGL(join_count_stats_kernel, dim3(256), dim3(256), 0, ctx->stream, js->cnt, js->n_src * row_count,
                           js->cap, st.p);
      hipLaunchKernelGGL(join_tail_words_kernel, dim3(1), dim3(64), 0, ctx->stream, (const unsigned long long *)scored.p, kRsScoredSlots,
                         (const uint32_t *)fail_count.p, st.p + 3);
      unsigned long long h[5] = {0, 0, 0, 0, 0};
      AS_HIP(ctx, copy_d2h(ctx, h, st.p, sizeof(h)));
      ctx->stats["knn_join_cap"] = js->cap;
      ctx->stats["knn_join_candidates"] = (int64_t)h[0];
      ctx->stats["knn_join_max_per_row"] = (int64_t)h[1];
      ctx->stats["knn_join_overflow_rows"] = (int64_t)h[2];
      if (scored.p) ctx->stats["knn_rerank_scored"] = (int64_t)h[3];  // candidates that survived the pruning: the rows the scoring kernel fetched
      n_fail = (uint32_t)h[4];
    }
  } else {
    const uint32_t item_rows = use_f32 ? 2 * kTileRows : kTileRows;  // A rows per work item
    const uint32_t a_tiles_all = (row_count + kTileRows - 1) / kTileRows;
    const float *PA32 = nullptr;
    const double *PA64 = nullptr;
    if (use_f32) {
      AS_TRY(B32.buf.alloc(ctx, packed_elems(b_alloc, d)));
      AS_TRY(pack_f32(ctx, rows, norms, nullptr, 0, n, d, B32.buf.p, b_alloc));
      B32.p = B32.buf.p;
      PA32 = B32.p + (row_begin / kTileRows) * packed_elems(1, d);
      if (row_begin % item_rows != 0) {
        const uint32_t at = (a_tiles_all + 1) & ~1u;
        AS_TRY(A32.buf.alloc(ctx, packed_elems(at, d)));
        AS_TRY(pack_f32(ctx, rows, norms, nullptr, row_begin, row_count, d, A32.buf.p, at));
        PA32 = A32.buf.p;
      }
    } else {
      AS_TRY(B64.buf.alloc(ctx, packed_elems(b_alloc, d)));
      AS_TRY(pack_f64(ctx, rows, norms, nullptr, 0, n, d, B64.buf.p, b_alloc));
      B64.p = B64.buf.p;
      PA64 = B64.p + (row_begin / kTileRows) * packed_elems(1, d);
      if (row_begin % item_rows != 0) {
        AS_TRY(A64.buf.alloc(ctx, packed_elems(a_tiles_all, d)));
        AS_TRY(pack_f64(ctx, rows, norms, nullptr, row_begin, row_count, d, A64.buf.p, a_tiles_all));
        PA64 = A64.buf.p;
      }
    }
    const uint32_t items_all = (row_count + item_rows - 1) / item_rows;
    uint32_t bulk_items = items_all;
    if (items_all >= 512 && items_all % 256 != 0) bulk_items = items_all / 256 * 256;
    uint32_t splits_used = 1;
    for (int part = 0; part < 2; ++part) {
      const uint32_t it0 = part == 0 ? 0 : bulk_items;
      const uint32_t it1 = part == 0 ? bulk_items : items_all;
      if (it0 >= it1) continue;
      const uint32_t r0 = it0 * item_rows;                         // first row of this part (relative)
      const uint32_t rc = std::min(row_count, it1 * item_rows) - r0;  // its row count
      const uint32_t a_tiles = (rc + kTileRows - 1) / kTileRows;
      pb.arows = ARows{nullptr, (uint32_t)row_begin + r0};
      pb.nA = rc;
      pb.splits = clamp_splits(pick_splits(it1 - it0, b_tiles), d, KC);
      if (part == 1 && it1 - it0 < 256)  // remainder round: one work unit per workgroup, as short as possible
        pb.splits = clamp_splits(std::max(1u, std::min({16u, 256u / (it1 - it0), b_tiles})), d, KC);
      splits_used = std::max(splits_used, pb.splits);
      DevBuf<double> ckey;
      DevBuf<uint32_t> cidx;
      AS_TRY(ckey.alloc(ctx, (size_t)rc * pb.splits * KC));
      AS_TRY(cidx.alloc(ctx, (size_t)rc * pb.splits * KC));
      pb.cand_key = ckey.p;
      pb.cand_idx = cidx.p;
      if (use_f32) {
        AS_TRY(candidates_f32(ctx, pb, PA32 + (size_t)(r0 / kTileRows) * packed_elems(1, d), a_tiles, B32.p, b_tiles,
                              "knn_mfma_candidates_f32"));
        if (ctx->opt_debug) continue;  // timing experiment: candidate pass only, outputs undefined
        GraphBound bnd{cos_delta_f32(d), rectify};
        AS_TRY(launch_rerank(ctx, "knn_rerank", rows, rows, d, nullptr, (uint32_t)row_begin + r0, rc, r0, K, KC,
                             pb.splits, ckey.p, cidx.p, force.p, pol, bnd, out_dist, out_idx, fail1.p, fail_count.p,
                             err.p));
      } else {
        AS_TRY(candidates_f64(ctx, pb, PA64 + (size_t)(r0 / kTileRows) * packed_elems(1, d), a_tiles, B64.p, b_tiles,
                              "knn_mfma_candidates"));
        GraphBound bnd{cos_delta_f64(d), rectify};
        AS_TRY(launch_rerank(ctx, "knn_rerank", rows, rows, d, nullptr, (uint32_t)row_begin + r0, rc, r0, K, KC,
                             pb.splits, ckey.p, cidx.p, force.p, pol, bnd, out_dist, out_idx, fail1.p, fail_count.p,
                             err.p));
      }
      // (no synchronisation to hand the workspace back: a block returned to the cache is reused by later work on this same stream only)
    }
    if (ctx->opt_debug) return AS_OK;
    pb.splits = splits_used;
    pb.arows = ARows{nullptr, (uint32_t)row_begin};
    pb.nA = row_count;
    AS_TRY(read_u32(ctx, fail_count.p, &n_fail));
  }
  ctx->stats["knn_uncertified_rows"] = n_fail;
  ctx->stats["knn_join_delta_e9"] = use_join ? (int64_t)(join_delta(ctx, d) * 1e9) : 0;
  ctx->stats["knn_uncertified_rows_stage1a"] = 0;
  ctx->stats["knn_stream_scan_rows"] = 0;
  ctx->stats["knn_uncertified_rows_stage1b"] = 0;
  ctx->stats["knn_uncertified_rows_stage2"] = 0;
  ctx->stats["knn_cand_capacity"] = KC;
  ctx->stats["knn_col_splits"] = pb.splits;
  if (n_fail == 0) return AS_OK;

  // what the later stages need to know about the join behind the lists
  const bool grouped_theta = oj && oj->grouped_theta;                                // thresholds from the sample of the nearest pivot groups
  const bool wide_pre = oj && oj->wide_pre;                                          // the rotated coordinates of the rows are in use
  const double *rot_z = oj ? oj->rot_z.p : nullptr;                                  // ... n x 32
  const uint32_t *perm_used = (oj && oj->reorder) ? oj->perm_own.p : nullptr;        // image position -> row of the f16 image js->packedh
  // the lists of row g: index (lists_inv ? lists_inv[g] : g) - list_base (second_theta_kernel); its looser threshold: theta2[g - t2_base]
  const uint32_t *lists_inv = (lists_by_pos || (js && js->a_ids)) ? oj->inv_own.p : nullptr;
  const uint32_t list_base = (js && js->a_ids) ? js->list_base : (lists_by_pos ? 0u : (uint32_t)row_begin);
  const uint32_t t2_base = (js && js->a_ids) ? 0u : (uint32_t)row_begin;
  // Failed rows go through the stages that follow, each on the gathered list of what the one before
  // could not certify: (after the join only) the f32 top-k pass, the f64 MFMA pass, the exact scan.
  // cur_ids = global ids of the rows still open, cur_pos = their positions in the output range.
  DevBuf<uint32_t> gids;
  AS_TRY(gids.alloc(ctx, n_fail));
  {
    const uint32_t *a_ids_f = js ? js->a_ids : nullptr;  // (a range of image positions: the rows are a list)
    hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(n_fail, 256)), dim3(256), 0, ctx->stream, a_ids_f, fail1.p, n_fail,
                       a_ids_f ? 0u : (uint32_t)row_begin, gids.p);
  }
  ctx->stats["knn_open_rows_kept"] = 0;
  if (ctx->opt_keep_open_rows) {
    // diagnostic ("keep_open_rows"): the ids of the rows the first re-rank left open -- every row that goes through the second pass,
    // the f32 / f64 top-k passes or an exact scan is one of them -- readable as the stats "knn_open_row_<i>", so that a test can
    // put exactly these rows in front of the oracle (tests/test_gpu_fullsize.py)
    std::vector<uint32_t> h(n_fail);
    AS_HIP(ctx, copy_d2h(ctx, h.data(), gids.p, sizeof(uint32_t) * n_fail));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (uint32_t i = 0; i < n_fail; ++i) ctx->stats["knn_open_row_" + std::to_string(i)] = (int64_t)h[i];
    ctx->stats["knn_open_rows_kept"] = n_fail;
  }
  const uint32_t *cur_ids = gids.p;
  const uint32_t *cur_pos = fail1.p;
  uint32_t cur_n = n_fail;
  // The open rows in the order of the join's image (round 5).  The ranking kernel appends them as its waves finish; in the order
  // of the image the open rows of one neighbourhood follow each other, and the waves of the later stages that run side by side
  // fetch the same candidate rows (1 M x 768: the second re-rank moves 4.8 M pairs x 6 KB).  Flags by position, a scan, the
  // flagged positions in order: the same list whatever order the rows were appended in.
  DevBuf<uint32_t> ord_flag, ord_pos, ord_ids, ord_rows;
  DevBuf<uint64_t> ord_rank;
  if (ctx->opt_join_open_order && oj && oj->reorder && oj->inv_own.p && oj->perm_own.p && !(js && js->a_ids) && row_begin == 0 && row_count == n &&
      n_fail > 64) {
    AS_TRY(ord_flag.alloc(ctx, n));
    AS_TRY(ord_rank.alloc(ctx, (size_t)n + 1));
    AS_TRY(ord_pos.alloc(ctx, n_fail));
    AS_TRY(ord_ids.alloc(ctx, n_fail));
    AS_HIP(ctx, hipMemsetAsync(ord_flag.p, 0, sizeof(uint32_t) * n, ctx->stream));
    hipLaunchKernelGGL(flag_positions_kernel, dim3(cdiv(n_fail, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)gids.p, (const uint32_t *)oj->inv_own.p,
                       n_fail, ord_flag.p);
    AS_TRY(exclusive_scan_u32_u64(ctx, ord_flag.p, n, ord_rank.p));
    hipLaunchKernelGGL(list_flagged_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)ord_flag.p, (const uint64_t *)ord_rank.p, n,
                       n_fail, ord_pos.p);
    hipLaunchKernelGGL(rows_of_positions_kernel, dim3(cdiv(n_fail, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)ord_pos.p,
                       (const uint32_t *)oj->perm_own.p, n_fail, ord_ids.p);
    AS_HIP(ctx, hipGetLastError());
    cur_ids = ord_ids.p;  // (the whole matrix from row 0: a row's place in the output is its id)
    cur_pos = ord_ids.p;
    ctx->stats["knn_join_open_order"] = 1;
  } else {
    ctx->stats["knn_join_open_order"] = 0;
  }
  DevBuf<uint32_t> ids_a, pos_a, ids_b, pos_b;

  auto mfma_stage = [&](bool f32, DevBuf<uint32_t> &ids_out, DevBuf<uint32_t> &pos_out, const char *stat) -> int {
    const uint32_t a_tiles = (cur_n + kTileRows - 1) / kTileRows;
    pb.arows = ARows{cur_ids, 0};
    pb.nA = cur_n;
    pb.splits = clamp_splits(pick_splits(f32 ? (a_tiles + 1) / 2 : a_tiles, b_tiles), d, KC);
    DevBuf<double> ckey, ok, A64g;
    DevBuf<uint32_t> cidx, oi, failx;
    DevBuf<float> A32g;
    AS_TRY(ckey.alloc(ctx, (size_t)cur_n * pb.splits * KC));
    AS_TRY(cidx.alloc(ctx, (size_t)cur_n * pb.splits * KC));
    pb.cand_key = ckey.p;
    pb.cand_idx = cidx.p;
    if (f32) {
      if (!B32.p && js && js->packed) B32.p = js->packed;
      if (!B32.p) {
        AS_TRY(B32.buf.alloc(ctx, packed_elems(b_alloc, d)));
        AS_TRY(pack_f32(ctx, rows, norms, nullptr, 0, n, d, B32.buf.p, b_alloc));
        B32.p = B32.buf.p;
      }
      const uint32_t at = (a_tiles + 1) & ~1u;
      AS_TRY(A32g.alloc(ctx, packed_elems(at, d)));
      AS_TRY(pack_f32(ctx, rows, norms, cur_ids, 0, cur_n, d, A32g.p, at));
      AS_TRY(candidates_f32(ctx, pb, A32g.p, a_tiles, B32.p, b_tiles, "knn_mfma_candidates_f32"));
    } else {
      if (!B64.p) {
        AS_TRY(B64.buf.alloc(ctx, packed_elems(b_alloc, d)));
        AS_TRY(pack_f64(ctx, rows, norms, nullptr, 0, n, d, B64.buf.p, b_alloc));
        B64.p = B64.buf.p;
      }
      AS_TRY(A64g.alloc(ctx, packed_elems(a_tiles, d)));
      AS_TRY(pack_f64(ctx, rows, norms, cur_ids, 0, cur_n, d, A64g.p, a_tiles));
      AS_TRY(candidates_f64(ctx, pb, A64g.p, a_tiles, B64.p, b_tiles, "knn_mfma_candidates"));
    }
    AS_TRY(ok.alloc(ctx, (size_t)cur_n * K));
    AS_TRY(oi.alloc(ctx, (size_t)cur_n * K));
    AS_TRY(failx.alloc(ctx, cur_n));
    AS_HIP(ctx, hipMemsetAsync(fail_count.p, 0, sizeof(uint32_t), ctx->stream));
    GraphBound bnd{f32 ? cos_delta_f32(d) : cos_delta_f64(d), rectify};
    AS_TRY(launch_rerank(ctx, "knn_rerank", rows, rows, d, cur_ids, 0u, cur_n, 0u, K, KC, pb.splits, ckey.p, cidx.p,
                         force.p, pol, bnd, ok.p, oi.p, failx.p, fail_count.p, err.p));
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv((uint64_t)cur_n * K, 256)), dim3(256), 0, ctx->stream, cur_pos,
                       cur_n, K, ok.p, oi.p, out_dist, out_idx);
    uint32_t n_next = 0;
    AS_TRY(read_u32(ctx, fail_count.p, &n_next));
    ctx->stats[stat] = n_next;
    if (n_next) {
      AS_TRY(ids_out.alloc(ctx, n_next));
      AS_TRY(pos_out.alloc(ctx, n_next));
      hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(n_next, 256)), dim3(256), 0, ctx->stream, cur_ids, failx.p, n_next,
                         0u, ids_out.p);
      hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(n_next, 256)), dim3(256), 0, ctx->stream, cur_pos, failx.p, n_next,
                         0u, pos_out.p);
      // (no synchronisation to hand the workspace back: a block returned to the cache is reused by later work on this same stream only)
      cur_ids = ids_out.p;
      cur_pos = pos_out.p;
    }
    cur_n = n_next;
    return AS_OK;
  };

  // ---- stage 1a (after a join on the bf16 pipe): the one-sided join over the gathered rows ----
  // A row fails the first pass because its list overflowed (threshold too low for its neighbourhood: same threshold,
  // longer list) or because its K-th neighbour is not clear of the threshold (too high: the looser second threshold).
  // ~1 % of the rows; as a one-sided sweep of the split image this is a few ms, the f32 top-k pass over them 11-22.
  DevBuf<uint32_t> ids_r, pos_r, ids_w, pos_w;
  DevBuf<unsigned char> P16late;
  // rows wider than 128 after the f16 join: the same pass on the f16 plane (one-sided form of mfma_join_f16_wide.hip);
  // the looser thresholds carry it, the bound stays the f16 one
  const bool second_wide = join_on_f16(ctx, d) && d > 128 && js && js->packedh;
  // up to 128 wide after the f16 join: the pass on the f16 plane as well -- the search mode of pair_join_f16_kernel with
  // alpha = 1, beta = 0 and zero lambdas is exactly a one-sided threshold pass (a row lists itself; the re-rank drops it) --
  // which saves building the split-bf16 image of all rows for half a percent of them (opt second_pass_f16 = 0: as before)
  // (where the streaming exact scan takes what is left: the f16 bound leaves a few more rows open than the split form)
  const bool second_h = join_on_f16(ctx, d) && d == 128 && js && js->packedh && ctx->opt_second_pass_f16 && ctx->opt_stream_scan &&
                        stream_scan_applies(1, n, d, K);
  if (use_join && js->theta2 && (second_pass_bf16(ctx, d) || second_wide) && cur_n > 0 && cur_n <= n / 4) {
    const uint32_t nf = cur_n, cap2 = 1024;
    const void *split_image = (second_wide || second_h) ? js->packedh : js->packed16;
    if (!split_image) {  // the join ran on the f16 plane: the split-bf16 image of all rows is only built when rows are left
      AS_TRY(P16late.alloc(ctx, packed_vecs_bf16(b_alloc, d) * 16));
      AS_TRY(pack_bf16(ctx, rows, norms, nullptr, 0, n, d, P16late.p, b_alloc));
      split_image = P16late.p;
    }
    const uint32_t bt = (nf + kTileRows - 1) / kTileRows, b_al = (second_h || second_wide) ? (bt + 3) & ~3u : (bt + 1) & ~1u, pad = b_al * kTileRows;
    // The f16 pass behind a ONE-SIDED cone bound (cone_tiles.hip: the cones of the image's tiles are kept from the first pass): the
    // open rows are put in the order of the image first -- a wave's 64 rows are then neighbours, and most column tiles lie outside
    // what any of them can reach.
    DevBuf<uint32_t> ids_s, pos_s;
    const bool cone_second = second_h && oj && oj->cone.valid && oj->rot_z.p && oj->inv_own.p && ctx->opt_join_cone_second && nf >= 512 &&
                             !(ctx->opt_debug & 2048);
    if (cone_second) {
      DevBuf<double> keys;
      DevBuf<uint32_t> ord;
      AS_TRY(keys.alloc(ctx, nf));
      AS_TRY(ord.alloc(ctx, nf));
      AS_TRY(ids_s.alloc(ctx, nf));
      AS_TRY(pos_s.alloc(ctx, nf));
      hipLaunchKernelGGL(position_keys_kernel, dim3(cdiv(nf, 256)), dim3(256), 0, ctx->stream, cur_ids, (const uint32_t *)oj->inv_own.p, nf, keys.p);
      AS_TRY(order_by_value(ctx, keys.p, nf, ord.p));
      hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(nf, 256)), dim3(256), 0, ctx->stream, cur_ids, (const uint32_t *)ord.p, nf, 0u, ids_s.p);
      hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(nf, 256)), dim3(256), 0, ctx->stream, cur_pos, (const uint32_t *)ord.p, nf, 0u, pos_s.p);
      AS_HIP(ctx, hipGetLastError());
      cur_ids = ids_s.p;
      cur_pos = pos_s.p;
    }
    DevBuf<unsigned char> PA;
    DevBuf<float> theta_a, zeros;
    DevBuf<uint32_t> cnt_b, cand_b, unitc, oi, failx;
    DevBuf<double> ok;
    AS_TRY(PA.alloc(ctx, ((second_wide || second_h) ? packed_vecs_f16(b_al, d) : packed_vecs_bf16(b_al, d)) * 16));
    AS_TRY(theta_a.alloc(ctx, pad));
    AS_TRY(cnt_b.alloc(ctx, pad));
    AS_TRY(cand_b.alloc(ctx, (size_t)pad * cap2 * 2));
    AS_TRY(unitc.alloc(ctx, 1));
    AS_TRY(ok.alloc(ctx, (size_t)nf * K));
    AS_TRY(oi.alloc(ctx, (size_t)nf * K));
    AS_TRY(failx.alloc(ctx, nf));
    if (second_wide || second_h) AS_TRY(pack_f16(ctx, rows, norms, cur_ids, 0, nf, d, PA.p, b_al));
    else AS_TRY(pack_bf16(ctx, rows, norms, cur_ids, 0, nf, d, PA.p, b_al));
    const float *band_own = ((second_h || second_wide) && oj && oj->theta_band_own.p && ctx->opt_join_theta_band_second) ? oj->theta_band_own.p : nullptr;
    const float band_by = (float)(2.0 * join_delta(ctx, d) + 2e-6);
    if (grouped_theta && second_h && !(ctx->opt_debug & 131072)) {  // (debug bit 131072: the first pass's own second thresholds)
      // the thresholds of the first pass came from the sampled rows of the nearest pivot groups only; a row that is open after it
      // may be one whose neighbours live elsewhere (a cluster without a pivot of its own), so the second pass takes thresholds
      // against the whole sample for the few thousand rows it sweeps (PA is their f16 image)
      const int64_t rank_opt = ctx->opt_join_rank > 0 ? ctx->opt_join_rank : 4;
      const uint32_t m = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(rank_opt, 32)), m2 = std::min<uint32_t>(48, 2 * m + 6);
      const uint32_t ns = std::max(256u, (n / 32u) & ~255u);
      DevBuf<uint32_t> sid;
      DevBuf<float> t1, t2;
      AS_TRY(sid.alloc(ctx, ns));
      AS_TRY(t1.alloc(ctx, pad));
      AS_TRY(t2.alloc(ctx, pad));
      hipLaunchKernelGGL(sample_ids_kernel, dim3(cdiv(ns, 256)), dim3(256), 0, ctx->stream, ns, n / ns, sample_half_bits(ns), sid.p);
      AS_TRY(join_thresholds_f16(ctx, rows, norms, d, PA.p, 0, nf, sid.p, ns, m, t1.p, m2, t2.p));
      hipLaunchKernelGGL(second_theta_fresh_kernel, dim3(cdiv(pad, 256)), dim3(256), 0, ctx->stream, (const float *)t1.p, (const float *)t2.p,
                         js->theta, js->cnt, js->n_src, list_base, row_count, js->cap, cur_ids, nf, pad, theta_a.p, lists_inv, band_own, band_by);
    } else {
      hipLaunchKernelGGL(second_theta_kernel, dim3(cdiv(pad, 256)), dim3(256), 0, ctx->stream, js->theta, js->theta2, js->cnt,
                         js->n_src, t2_base, list_base, row_count, js->cap, cur_ids, nf, pad, theta_a.p, lists_inv, band_own, band_by);
    }
    AS_HIP(ctx, hipMemsetAsync(cnt_b.p, 0, sizeof(uint32_t) * pad, ctx->stream));
    AS_HIP(ctx, hipMemsetAsync(unitc.p, 0, sizeof(uint32_t), ctx->stream));
    AS_HIP(ctx, hipMemsetAsync(fail_count.p, 0, sizeof(uint32_t), ctx->stream));
    if (second_h) {
      const size_t zn = std::max<size_t>(((size_t)n + 511) / 512 * 512, pad);
      AS_TRY(zeros.alloc(ctx, zn));
      AS_HIP(ctx, hipMemsetAsync(zeros.p, 0, sizeof(float) * zn, ctx->stream));
      DevBuf<unsigned char> mask2;
      bool masked = false;
      if (cone_second) {
        double kept = 1.0;
        AS_TRY(mask2.alloc(ctx, (size_t)(pad / 512u) * 2 * ((n + 511) / 512) * 8));
        AS_TRY(cone_rect_mask(ctx, oj->rot_z.p, cur_ids, nf, theta_a.p, pad, oj->cone, n, mask2.p, &kept));
        masked = kept <= (double)ctx->opt_join_cone_tiles_max_e3 * 1e-3;
      }
      ctx->stats["knn_join_cone_second"] = masked ? 1 : 0;
      AS_TRY(join_f16_search(ctx, split_image, n, d, PA.p, nf, theta_a.p, theta_a.p, zeros.p, zeros.p, 1.f, 0.f, cap2, cnt_b.p, cand_b.p,
                             unitc.p, "knn_mfma_join_f16_second", perm_used, nullptr, masked ? mask2.p : nullptr, cur_ids));
    } else if (second_wide) {
      // rows wider than 128: the rectangular int8 first pass of the search (prefilter_i8_rect: the open rows' rotated image against
      // the rotated image of all rows) in front of the panelled one-sided pass, which then contracts only the pairs of super tiles
      // it leaves -- 6 k open rows against 1 M x 768 were 11.7 ms of dense products
      DevBuf<unsigned char> P8all, PA8, mask2;
      bool masked = false;
      if (oj && oj->wide_pre && oj->rot_z.p && ctx->opt_join_second_prefilter && pad % 512u == 0) {
        const uint32_t tiles_all = packed_tiles_f16(n), a_tiles = pad / 128u;
        double e_all = 0.0, e_a = 0.0;
        AS_TRY(P8all.alloc(ctx, (size_t)tiles_all * 4 * 64 * 16));
        AS_TRY(PA8.alloc(ctx, (size_t)a_tiles * 4 * 64 * 16));
        AS_TRY(pack_rotated_from_z(ctx, oj->rot_z.p, perm_used, n, tiles_all, P8all.p, &e_all));
        AS_TRY(pack_rotated_from_z(ctx, oj->rot_z.p, cur_ids, nf, a_tiles, PA8.p, &e_a));
        const double E = std::max(e_all, e_a), delta8 = (2.0 * E + E * E) * (1.0 + 1e-6) + 1e-4;
        const uint32_t T2i = (n + 511u) / 512u, a_supers = pad / 512u;
        AS_TRY(mask2.alloc(ctx, (size_t)a_supers * 2 * T2i * 8));
        AS_TRY(prefilter_i8_rect(ctx, P8all.p, n, PA8.p, a_supers, theta_a.p, delta8, mask2.p, "knn_join_second_prefilter"));
        masked = true;
      }
      ctx->stats["knn_join_second_prefilter"] = masked ? 1 : 0;
      AS_TRY(join_f16_wide_rect(ctx, split_image, n, d, PA.p, nf, cur_ids, theta_a.p, cap2, cnt_b.p, cand_b.p, "knn_mfma_join_f16_second",
                                perm_used, masked ? mask2.p : nullptr));
    }
    else
      AS_TRY(join_bf16_rect(ctx, split_image, n, d, PA.p, nf, cur_ids, theta_a.p, cap2, cnt_b.p, cand_b.p, unitc.p,
                            "knn_mfma_join_bf16_second"));
    const double delta2 = (second_wide || second_h) ? join_delta(ctx, d) : rect_delta(d);
    {
      GraphBound bnd{delta2, rectify};
      ProfScope ps(ctx, "knn_rerank");
      const size_t lds = (size_t)4 * (d + cap2) * sizeof(double) + (size_t)4 * cap2 * sizeof(uint32_t);
      AS_HIP(ctx, hipFuncSetAttribute((const void *)rerank_join_kernel<GraphDistPolicy, GraphBound>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL((rerank_join_kernel<GraphDistPolicy, GraphBound>), dim3(std::min<uint64_t>(cdiv(nf, 4), 8192)),
                         dim3(256), lds, ctx->stream, rows, rows, cur_ids, d, 0u, nf, theta_a.p, 1u, cnt_b.p, cand_b.p, cap2, cap2,
                         K, force.p, pol, bnd, (ctx->opt_debug & 16) ? -1.0 : delta2, rectify, ok.p, oi.p, failx.p,
                         fail_count.p, err.p);
      AS_HIP(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv((uint64_t)nf * K, 256)), dim3(256), 0, ctx->stream, cur_pos, nf, K, ok.p,
                       oi.p, out_dist, out_idx);
    uint32_t n_next = 0;
    AS_TRY(read_u32(ctx, fail_count.p, &n_next));
    ctx->stats["knn_uncertified_rows_stage1a"] = n_next;
    if (n_next) {
      AS_TRY(ids_r.alloc(ctx, n_next));
      AS_TRY(pos_r.alloc(ctx, n_next));
      hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(n_next, 256)), dim3(256), 0, ctx->stream, cur_ids, failx.p, n_next, 0u,
                         ids_r.p);
      hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(n_next, 256)), dim3(256), 0, ctx->stream, cur_pos, failx.p, n_next, 0u,
                         pos_r.p);
      // (no synchronisation to hand the workspace back: a block returned to the cache is reused by later work on this same stream only)
      cur_ids = ids_r.p;
      cur_pos = pos_r.p;
    }
    cur_n = n_next;
    if (cur_n == 0) return AS_OK;
  }

  // ---- a handful of rows left: one exact pass over the matrix per 32 of them (stream_scan.hip) instead of MFMA sweeps ----
  if (ctx->opt_stream_scan && cur_n <= kStreamRowsMax && stream_scan_applies(std::min<uint32_t>(cur_n, 32), n, d, K)) {
    DevBuf<double> fk;
    DevBuf<uint32_t> fi, redo_ids, redo_pos_d;
    AS_TRY(fk.alloc(ctx, (size_t)cur_n * K));
    AS_TRY(fi.alloc(ctx, (size_t)cur_n * K));
    std::vector<uint32_t> redo_all;
    // with the rotated coordinates of the own join at hand the list pass of the scan is the int8 filter + the pairs it leaves
    // (stream_scan.hip): the image of all rows and of each batch are packed from the coordinates
    DevBuf<unsigned char> f_p8, f_q8;
    StreamFilter flt{};
    bool use_flt = false;
    if (wide_pre && rot_z && d == 128 && ctx->opt_search_stream_filter && n < (1u << 27)) {
      const uint32_t tiles = packed_tiles_f16(n);
      double e_items = 0.0;
      AS_TRY(f_p8.alloc(ctx, (size_t)tiles * 4 * 64 * 16));
      AS_TRY(f_q8.alloc(ctx, (size_t)4 * 64 * 16));
      AS_TRY(pack_rotated_from_z(ctx, rot_z, nullptr, n, tiles, f_p8.p, &e_items));
      flt.P8 = f_p8.p;
      flt.err_items = (float)(e_items * (1.0 + 1e-6));
      use_flt = true;
    }
    for (uint32_t c0 = 0; c0 < cur_n; c0 += 32) {
      const uint32_t cn = std::min<uint32_t>(32, cur_n - c0);
      std::vector<uint32_t> redo;
      if (use_flt) {
        double e_a = 0.0;
        AS_TRY(pack_rotated_from_z(ctx, rot_z, cur_ids + c0, cn, 1, f_q8.p, &e_a));
        flt.PQ8 = f_q8.p;
        flt.err_a = (float)(e_a * (1.0 + 1e-6));
      }
      AS_TRY(stream_scan_graph(ctx, rows, cur_ids + c0, cn, n, K, pol, fk.p + (size_t)c0 * K, fi.p + (size_t)c0 * K, err.p, &redo,
                               use_flt ? &flt : nullptr));
      for (uint32_t r : redo) redo_all.push_back(c0 + r);
    }
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv((uint64_t)cur_n * K, 256)), dim3(256), 0, ctx->stream, cur_pos, cur_n, K, fk.p,
                       fi.p, out_dist, out_idx);
    AS_HIP(ctx, hipGetLastError());
    ctx->stats["knn_stream_scan_rows"] = cur_n;
    if (redo_all.empty()) return AS_OK;
    // rows whose candidate list overflowed (a sample that misses their neighbourhood): the exact scan, below
    AS_TRY(redo_pos_d.alloc(ctx, redo_all.size()));
    AS_HIP(ctx, copy_h2d(ctx, redo_pos_d.p, redo_all.data(), sizeof(uint32_t) * redo_all.size()));
    const uint32_t nr = (uint32_t)redo_all.size();
    AS_TRY(ids_r.alloc(ctx, nr));
    AS_TRY(pos_r.alloc(ctx, nr));
    hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(nr, 256)), dim3(256), 0, ctx->stream, cur_ids, redo_pos_d.p, nr, 0u, ids_r.p);
    hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(nr, 256)), dim3(256), 0, ctx->stream, cur_pos, redo_pos_d.p, nr, 0u, pos_r.p);
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    cur_ids = ids_r.p;
    cur_pos = pos_r.p;
    cur_n = nr;
    DevBuf<double> fk2;
    DevBuf<uint32_t> fi2;
    AS_TRY(fk2.alloc(ctx, (size_t)cur_n * K));
    AS_TRY(fi2.alloc(ctx, (size_t)cur_n * K));
    AS_TRY(knn_exact(ctx, rows, norms, n, d, cur_ids, 0, cur_n, K, rectify, fi2.p, fk2.p, 0));
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv((uint64_t)cur_n * K, 256)), dim3(256), 0, ctx->stream, cur_pos, cur_n, K, fk2.p,
                       fi2.p, out_dist, out_idx);
    AS_HIP(ctx, hipGetLastError());
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return AS_OK;
  }

  // ---- rows of 256 .. 1024 values, a handful left: the quick pass over all rows + exact chains (stream_scan.hip) instead of an
  // f32 top-k sweep for them (47 ms for ten rows at 1 M x 768) ----
  if (ctx->opt_stream_scan && ctx->opt_search_stream_quick && cur_n <= kStreamRowsMax &&
      stream_quick_wide_applies(std::min<uint32_t>(cur_n, 32), n, d, K)) {
    DevBuf<double> fk;
    DevBuf<uint32_t> fi, redo_pos_d;
    AS_TRY(fk.alloc(ctx, (size_t)cur_n * K));
    AS_TRY(fi.alloc(ctx, (size_t)cur_n * K));
    std::vector<uint32_t> redo_all;
    for (uint32_t c0 = 0; c0 < cur_n; c0 += 32) {
      const uint32_t cn = std::min<uint32_t>(32, cur_n - c0);
      std::vector<uint32_t> redo;
      AS_TRY(stream_quick_wide_graph(ctx, rows, cur_ids + c0, cn, n, d, K, pol, fk.p + (size_t)c0 * K, fi.p + (size_t)c0 * K, err.p, &redo));
      for (uint32_t r : redo) redo_all.push_back(c0 + r);
    }
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv((uint64_t)cur_n * K, 256)), dim3(256), 0, ctx->stream, cur_pos, cur_n, K, fk.p, fi.p,
                       out_dist, out_idx);
    AS_HIP(ctx, hipGetLastError());
    ctx->stats["knn_stream_scan_rows"] = cur_n;
    if (redo_all.empty()) {
      AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
      return AS_OK;
    }
    const uint32_t nr = (uint32_t)redo_all.size();
    AS_TRY(redo_pos_d.alloc(ctx, nr));
    AS_HIP(ctx, copy_h2d(ctx, redo_pos_d.p, redo_all.data(), sizeof(uint32_t) * nr));
    AS_TRY(ids_w.alloc(ctx, nr));  // (own buffers: cur_ids / cur_pos may live in ids_r / pos_r)
    AS_TRY(pos_w.alloc(ctx, nr));
    hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(nr, 256)), dim3(256), 0, ctx->stream, cur_ids, redo_pos_d.p, nr, 0u, ids_w.p);
    hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(nr, 256)), dim3(256), 0, ctx->stream, cur_pos, redo_pos_d.p, nr, 0u, pos_w.p);
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    cur_ids = ids_w.p;
    cur_pos = pos_w.p;
    cur_n = nr;
  }

  // ---- stage 1b (after the join): f32 top-k pass over the gathered rows ----
  if (use_join && stages_fit && !exact_direct(cur_n, n)) {
    AS_TRY(mfma_stage(true, ids_a, pos_a, "knn_uncertified_rows_stage1b"));
    if (cur_n == 0) return AS_OK;
  }
  // ---- stage 2: f64 MFMA over the gathered rows ----
  if (use_f32 && stages_fit && !wide && !exact_direct(cur_n, n)) {
    AS_TRY(mfma_stage(false, ids_b, pos_b, "knn_uncertified_rows_stage2"));
    if (cur_n == 0) return AS_OK;
  }

  // ---- last stage: exact scan ----
  DevBuf<double> fk;
  DevBuf<uint32_t> fi;
  AS_TRY(fk.alloc(ctx, (size_t)cur_n * K));
  AS_TRY(fi.alloc(ctx, (size_t)cur_n * K));
  AS_TRY(knn_exact(ctx, rows, norms, n, d, cur_ids, 0, cur_n, K, rectify, fi.p, fk.p, 0));
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv((uint64_t)cur_n * K, 256)), dim3(256), 0, ctx->stream, cur_pos,
                     cur_n, K, fk.p, fi.p, out_dist, out_idx);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}


}  // namespace asx
