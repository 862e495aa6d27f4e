// search_drivers.hip -- staged driver of the lambda-aware search scan (K9; src/core.rs:760-798): score guesses -> (rectangular int8
// first pass on an index searched before) -> threshold scan on the 16-bit matrix pipe -> exact re-rank + certification -> top-k
// passes / exact scans for what is left.  Split from mfma_drivers.hip in round 5 (the shared pieces: mfma_stages.h).
#include "mfma_stages.h"

namespace asx {

// ---- K9 driver ----------------------------------------------------------------------------------
// max |v| over a vector as the bits of a non-negative double (atomicMax on the integer image is the
// max on the value); a NaN counts as +inf so that the caller takes the f64 epilogue.
__global__ void absmax_kernel(const double *__restrict__ v, uint64_t n, unsigned long long *__restrict__ out) {
  double m = 0.0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const double a = fabs(v[i]);
    m = a != a ? __builtin_inf() : fmax(m, a);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) raise_max_u64(out, (unsigned long long)__double_as_longlong(m));
}

// ---- threshold scan of the search (stage 0 of K9) --------------------------------------------------
// score(q, i) = alpha cos(q, i) + beta lam(q, i), lam = 1 - min(|lambda_q - lambda_i|, 1) (src/core.rs:229-239).
// Every query gets a guess g_q for its K-th best score (the m-th best approximate score over a sample of the items,
// mfma_theta_bf16.hip); the search form of the one-sided bf16 join lists every item whose approximate score
// reaches g_q; the exact re-rank scores the listed items in f64, and a query is certified when its K-th best exact
// score is clear of what an unlisted item can reach, g_q + delta (delta: the cosine bound of the split-bf16
// products scaled by alpha, the f32 evaluation of the lambda term, the f32 roundings of the sum).  Any guess is
// correct; a guess above the true K-th score fails the certification and the query goes on to the MFMA top-k /
// exact stages.
__global__ void gather_lambda_f32_kernel(const double *__restrict__ lam, const uint32_t *__restrict__ ids, uint32_t count,
                                         uint32_t padded, float *__restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < padded) out[i] = i < count ? (float)lam[ids ? ids[i] : i] : 0.f;
}
// guesses padded to whole super tiles: +inf on padding rows (never a candidate), -inf for a NaN guess (everything
// is listed, the list overflows, the query falls through)
__global__ void scan_guess_kernel(const float *__restrict__ guess, const uint32_t *__restrict__ ids, uint32_t nq,
                                  uint32_t padded, float *__restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= padded) return;
  float g = __builtin_inff();
  if (i < nq) {
    g = guess[ids ? ids[i] : i];
    if (!(g == g)) g = -__builtin_inff();
  }
  out[i] = g;
}

// the same for the wide form of the scan (mfma_join_f16_wide.hip): what the cosine alone must reach, (g - beta) / alpha,
// and g - beta
__global__ void scan_guess_wide_kernel(const float *__restrict__ guess, const uint32_t *__restrict__ ids, uint32_t nq, uint32_t padded,
                                       float alpha, float beta, float *__restrict__ cos_floor, float *__restrict__ g_minus_beta) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= padded) return;
  float g = __builtin_inff();
  if (i < nq) {
    g = guess[ids ? ids[i] : i];
    if (!(g == g)) g = -__builtin_inff();
  }
  g_minus_beta[i] = g - beta;
  cos_floor[i] = (g - beta) / alpha;
}

int search_mfma(as_ctx *ctx, as_index *ix, const double *queries, const double *q_norms, const double *q_lambdas,
                uint64_t nq64, uint32_t k, double alpha, uint32_t *out_idx, double *out_key) {
  const uint32_t n = (uint32_t)ix->n, d = (uint32_t)ix->d, nq = (uint32_t)nq64;
  const int NC = (int)((d + 31) / 32);
  const uint32_t K = std::min<uint32_t>(k, n);
  const uint32_t KC = cand_capacity(ctx, K);
  const bool wide = NC > kMaxNC;
  const bool use_f32 = (ctx->opt_cand_precision != 64 || wide) && fits_f32(KC);
  // deep lists (large k) and small matrices do not fit the MFMA top-k passes; the threshold scan (stage 0) has no
  // such limit, so what it leaves open then goes straight to the exact scan
  const bool mfma_ok = !((wide && !use_f32) || !fits_f64(KC) || n < 4 * kTileRows);
  if (nq == 0 || n < 4 * kTileRows || !rerank_fits(d))
    return search_exact(ctx, ix, queries, q_norms, q_lambdas, nullptr, nq, nq, k, alpha, out_idx, out_key);
  // an index that is searched again gets its items ordered by direction and a rotated int8 image (below); with those a small
  // batch reads 32 MB + the few blocks its queries can reach instead of streaming the 1 GB of rows
  const uint32_t prior_calls = ix->scan_calls.fetch_add(1);
  // shared for the whole search; a lazy build drops it, takes the index exclusively, builds (re-checking), synchronises the stream
  // and comes back (as_index, common.h).  Pointers of the index are read where they are used, after the builds.
  std::shared_lock<std::shared_mutex> reading(ix->rw);
#define AS_IX_BUILD(cond, ...)                                      \
  if (cond) {                                                       \
    reading.unlock();                                               \
    {                                                               \
      std::unique_lock<std::shared_mutex> building(ix->rw);         \
      if (cond) {                                                   \
        __VA_ARGS__                                                 \
        AS_HIP(ctx, hipStreamSynchronize(ctx->stream));             \
      }                                                             \
    }                                                               \
    reading.lock();                                                 \
  }
  const bool ordered_ok = !wide && n >= 65536 && ctx->opt_join_pipe == 16 && ctx->opt_search_scan_f16 && ctx->opt_search_scan &&
                          (ctx->opt_search_scan_order == 2 || (ctx->opt_search_scan_order == 1 && prior_calls >= 1));
  // items wider than 128 (round 5): the same two things -- an image in an order that groups the items by direction, found on their
  // rotated coordinates (rotate_wide.hip), and the rotated int8 image for a rectangular first pass in front of the panelled scan
  const bool ordered_ok_w = wide && n >= 65536 && rotated_wide_supported((uint32_t)d) && ctx->opt_join_pipe == 16 && ctx->opt_search_scan &&
                            ctx->opt_search_prefilter &&
                            // (a batch of 8 192 queries or more pays for the images -- ~30 ms per million items of 768 columns -- within
                            // its own scan, 1.5 ms per thousand queries there: such a batch gets them at the first search already)
                            (ctx->opt_search_scan_order == 2 || (ctx->opt_search_scan_order == 1 && (prior_calls >= 1 || nq >= 8192)));
  const bool prefer_staged = ordered_ok && ctx->opt_search_prefilter && d > 32 && ix->rot_state >= 0 && alpha > 0.0 && alpha <= 1.0 &&
                             nq >= (uint32_t)std::max<int64_t>(1, ctx->opt_search_staged_min_batch);
  // one query or a small batch: the bandwidth-bound regime.  One exact pass over the items (stream_scan.hip), no
  // candidate stages, nothing to certify.
  if (ctx->opt_stream_scan && stream_scan_applies(nq, n, d, K) && K == k && !prefer_staged) {
    DevBuf<int> err0;
    AS_TRY(err0.alloc(ctx, 1));
    AS_HIP(ctx, hipMemsetAsync(err0.p, 0, sizeof(int), ctx->stream));
    SearchScorePolicy pol0{q_norms, q_lambdas, ix->norms, ix->lambdas, alpha};
    std::vector<uint32_t> redo;
    // an index that is searched again keeps a rotated int8 image of its items (32 bytes each, in their own order): the list
    // pass then streams the ~1-3 % of the items that can reach some query's cut instead of all of them
    StreamFilter flt{};
    const StreamFilter *use_filter = nullptr;
    // (an index that adopted the images of its item-graph build -- z_space -- has that build's directions, in another form: the
    // filter's own, for the image in the order of the items, are found here like on any index, their state kept apart in nat_state)
    auto nat_state = [&]() -> int { return ix->z_space ? ix->nat_state : ix->rot_state; };
    if (ctx->opt_search_stream_filter && prior_calls >= 1 && alpha > 0.0 && alpha <= 1.0 && nat_state() >= 0 && ix->weird_items != 1) {
      AS_IX_BUILD(ix->weird_items < 0 || (ix->weird_items == 0 && nat_state() == 0) || (ix->weird_items == 0 && nat_state() == 1 && !ix->packed8_nat),
      if (ix->weird_items < 0) {  // (a non-finite norm or lambda must reach the exact arithmetic: no filter for such an index)
        DevBuf<int> wc;
        AS_TRY(wc.alloc(ctx, 1));
        AS_HIP(ctx, hipMemsetAsync(wc.p, 0, sizeof(int), ctx->stream));
        hipLaunchKernelGGL(count_weird_items_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, ix->norms, ix->lambdas, n, wc.p);
        int h = 0;
        AS_HIP(ctx, copy_d2h(ctx, &h, wc.p, sizeof(h)));
        AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ix->weird_items = h ? 1 : 0;
      }
      if (ix->weird_items == 0 && nat_state() == 0) {
        double tail = 1.0;
        if (!ix->rot_q) AS_HIP(ctx, handle_malloc((void **)&ix->rot_q, sizeof(float) * 32 * 128));  // (kept on failure: the next call reuses it)
        AS_TRY(rotated_basis(ctx, ix->items, ix->norms, n, d, ix->rot_q, &tail));
        (ix->z_space ? ix->nat_state : ix->rot_state) = tail <= (double)ctx->opt_search_rot_tail_e3 * 1e-3 ? 1 : -1;
      }
      if (ix->weird_items == 0 && nat_state() == 1 && !ix->packed8_nat) {
        const uint32_t tiles = packed_tiles_f16(n);
        AS_HIP(ctx, handle_malloc(&ix->packed8_nat, rotated_image_bytes(tiles)));
        AS_TRY(rotated_image_pack(ctx, ix->items, ix->norms, nullptr, n, d, ix->rot_q, ix->packed8_nat, tiles, &ix->rot_err_nat));
      }
      )
      if (ix->weird_items == 0 && nat_state() == 1 && ix->packed8_nat) {
        flt = StreamFilter{ix->packed8_nat, (float)(ix->rot_err_nat * (1.0 + 1e-6)), ix->rot_q, q_norms, q_lambdas, ix->norms, ix->lambdas, alpha};
        use_filter = &flt;
      }
    }
    AS_TRY(stream_scan_search(ctx, queries, nullptr, nq, ix->items, n, K, pol0, out_key, out_idx, err0.p, &redo, use_filter));
    int h_err0 = 0;
    AS_HIP(ctx, copy_d2h(ctx, &h_err0, err0.p, sizeof(int)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_err0) return fail(ctx, AS_ERR_NAN_SCORE, as_status_message(AS_ERR_NAN_SCORE));
    ctx->stats["search_scan"] = 0;
    ctx->stats["search_stream_scan"] = 1;
    if (redo.empty()) return AS_OK;
    // a candidate list overflowed (a sample that misses the query's neighbourhood): the whole batch takes the staged path
  } else if (ctx->opt_stream_scan && ctx->opt_search_stream_quick && stream_quick_wide_applies(nq, n, d, K) && K == k) {
    // items of 256 .. 1024 values, a small batch: the quick pass over the items (sample, then all of them) and the reference's
    // chain for the pairs it leaves (stream_scan.hip): ~1 ms per 6 GB instead of the staged path's 22 ms at one query
    DevBuf<int> err0;
    AS_TRY(err0.alloc(ctx, 1));
    AS_HIP(ctx, hipMemsetAsync(err0.p, 0, sizeof(int), ctx->stream));
    SearchScorePolicy pol0{q_norms, q_lambdas, ix->norms, ix->lambdas, alpha};
    std::vector<uint32_t> redo;
    AS_TRY(stream_quick_wide_search(ctx, queries, nq, ix->items, n, d, K, pol0, out_key, out_idx, err0.p, &redo));
    int h_err0 = 0;
    AS_HIP(ctx, copy_d2h(ctx, &h_err0, err0.p, sizeof(int)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_err0) return fail(ctx, AS_ERR_NAN_SCORE, as_status_message(AS_ERR_NAN_SCORE));
    ctx->stats["search_scan"] = 0;
    ctx->stats["search_stream_scan"] = 1;
    if (redo.empty()) return AS_OK;
  }
  else ctx->stats["search_stream_scan"] = 0;
  const uint32_t b_tiles = (n + kTileRows - 1) / kTileRows;
  const uint32_t b_alloc = (b_tiles + 1) & ~1u;
  auto ensure64 = [&]() -> int {
    AS_IX_BUILD(!ix->packed,
      AS_HIP(ctx, handle_malloc((void **)&ix->packed, sizeof(double) * packed_elems(b_alloc, d)));
      AS_TRY(pack_f64(ctx, ix->items, ix->norms, nullptr, 0, n, d, ix->packed, b_alloc));
    )
    return AS_OK;
  };
  auto ensure32 = [&]() -> int {
    AS_IX_BUILD(!ix->packed32,
      AS_HIP(ctx, handle_malloc((void **)&ix->packed32, sizeof(float) * packed_elems(b_alloc, d)));
      AS_TRY(pack_f32(ctx, ix->items, ix->norms, nullptr, 0, n, d, ix->packed32, b_alloc));
    )
    return AS_OK;
  };

  DevBuf<uint32_t> fail1, fail_count;
  DevBuf<int> err;
  AS_TRY(fail1.alloc(ctx, nq));
  AS_TRY(fail_count.alloc(ctx, 1));
  AS_TRY(err.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(fail_count.p, 0, sizeof(uint32_t), ctx->stream));
  AS_HIP(ctx, hipMemsetAsync(err.p, 0, sizeof(int), ctx->stream));
  SearchScorePolicy pol{q_norms, q_lambdas, ix->norms, ix->lambdas, alpha};

  CandProblem pb{};
  pb.mode = 1;
  pb.A = queries;
  pb.a_norms = q_norms;
  pb.arows = ARows{nullptr, 0};
  pb.nA = nq;
  pb.B = ix->items;
  pb.b_norms = ix->norms;
  pb.nB = n;
  pb.d = d;
  pb.q_lambdas = q_lambdas;
  pb.x_lambdas = ix->lambdas;
  pb.alpha = alpha;
  pb.KC = KC;

  uint32_t n_fail = 0;
  // ---- stage 0: threshold scan on the bf16 pipe (see the note above gather_lambda_f32_kernel) ----
  // (items wider than 128: the same scan on the panelled f16 kernel, mfma_join_f16_wide.hip)
  const bool scan_wide = wide && d <= 2048 && d % 2 == 0 && ctx->opt_join_pipe == 16;
  // up to 128 wide: the scan itself on one f16 plane (MODE 1 of pair_join_f16_kernel), guesses still from split-bf16 products
  const bool scan_h = !wide && ctx->opt_join_pipe == 16 && ctx->opt_search_scan_f16;
  bool scan = ctx->opt_search_scan && ctx->opt_cand_precision != 64 && (!wide || scan_wide) && alpha > 0.0 && alpha <= 1.0 && K < 1024 &&
              n >= (uint32_t)std::max<int64_t>(4 * kTileRows, ctx->opt_join_min_rows);
  double lsum = 0.0;
  if (scan) {  // the f32 lambda term needs lambdas of moderate size (as in the f32 top-k pass below)
    DevBuf<unsigned long long> amax;
    AS_TRY(amax.alloc(ctx, 2));
    AS_HIP(ctx, hipMemsetAsync(amax.p, 0, 2 * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(absmax_kernel, dim3(std::min<uint64_t>(1024, cdiv(nq, 256))), dim3(256), 0, ctx->stream, q_lambdas,
                       (uint64_t)nq, amax.p);
    hipLaunchKernelGGL(absmax_kernel, dim3(std::min<uint64_t>(1024, cdiv(n, 256))), dim3(256), 0, ctx->stream, ix->lambdas,
                       (uint64_t)n, amax.p + 1);
    AS_HIP(ctx, hipGetLastError());
    double hmax[2] = {0.0, 0.0};
    AS_HIP(ctx, copy_d2h(ctx, hmax, amax.p, sizeof(hmax)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    lsum = hmax[0] + hmax[1];
    scan = lsum <= 64.0;  // NaN lambdas fail this too
  }
  ctx->stats["search_scan"] = scan ? 1 : 0;
  if (!scan && !mfma_ok)
    return search_exact(ctx, ix, queries, q_norms, q_lambdas, nullptr, nq, nq, k, alpha, out_idx, out_key);
  if (scan) {
    const double beta = 1.0 - alpha;
    // up to 128 wide: the image in an order that groups the items by direction (the listing of the scan then meets its pairs in
    // few blocks, and the first pass below can rule the others out); the scan translates positions back.  Ordering the items,
    // finding their directions and the second image cost ~5 ms at 1 M items, so an index gets them at its SECOND scan
    // (search_scan_order = 2: at the first): one-shot use pays nothing, a serving index pays once.
    const bool want_order = scan_h && ordered_ok;
    const uint32_t ns_want = std::max(256u, (n / (uint32_t)std::max<int64_t>(2, ctx->opt_search_scan_sample)) & ~255u);
    const bool want_order_w = scan_wide && ordered_ok_w && ix->rot_state >= 0;
    auto scan_state_missing = [&]() -> bool {
      return (want_order && !ix->scan_perm && ix->packedh) || ((scan_wide || scan_h) && !ix->packedh) ||
             (want_order_w && ix->rot_state == 0) || (want_order_w && ix->rot_state == 1 && (!ix->scan_perm || !ix->packed8 || !ix->lam32_pos)) ||
             (scan_h && ctx->opt_search_prefilter && d > 32 && (want_order || ix->scan_perm) && (ix->rot_state == 0 || (ix->rot_state == 1 && !ix->packed8))) ||
             (!scan_wide && !scan_h && !ix->packed16) ||
             !ix->lam32 || (scan_h && ix->scan_perm && !ix->lam32_pos) || !ix->scan_sample || ix->scan_ns != ns_want;
    };
    AS_IX_BUILD(scan_state_missing(),
    if (want_order && !ix->scan_perm && ix->packedh) {  // (the image of the first scan was in the order of the items)
      (void)handle_free(ix->packedh);
      ix->packedh = nullptr;
    }
    if (want_order_w && (ix->rot_state == 0 || (ix->rot_state == 1 && (!ix->scan_perm || !ix->packed8)))) {
      // wide items: rotated coordinates -> order, ordered f16 image, rotated int8 image by position; the directions stay with the index
      const uint32_t tiles = packed_tiles_f16(n);
      DevBuf<double> Zi, zn;
      double tail = 1.0;
      AS_TRY(Zi.alloc(ctx, (size_t)n * 32));
      AS_TRY(zn.alloc(ctx, n));
      if (!ix->rot_qw) AS_HIP(ctx, handle_malloc((void **)&ix->rot_qw, sizeof(float) * 31 * (size_t)rotated_wide_pitch(d)));
      AS_TRY(rotated_rows_wide(ctx, ix->items, ix->norms, n, d, Zi.p, zn.p, &tail, nullptr, ix->rot_qw));
      // (the cut of the narrow items -- a tenth of the energy outside the directions -- weighs a 32-wide pass against a 128-wide scan;
      // here the scan is d / 32 times the pass, as for the join of wide rows: the images are kept unless the directions carry nothing)
      ix->rot_state = tail <= std::max(0.5, (double)ctx->opt_search_rot_tail_e3 * 1e-3) ? 1 : -1;
      ctx->stats["search_rot_tail_e6"] = (int64_t)(tail * 1e6);
      if (ix->rot_state == 1) {
        if (ix->packedh) {  // (the image of the first scan was in the order of the items)
          (void)handle_free(ix->packedh);
          ix->packedh = nullptr;
        }
        if (!ix->scan_perm) AS_HIP(ctx, handle_malloc((void **)&ix->scan_perm, sizeof(uint32_t) * (size_t)n));
        if (!ix->scan_piv) AS_HIP(ctx, handle_malloc((void **)&ix->scan_piv, sizeof(float) * 256 * 128));
        AS_TRY(cluster_order(ctx, Zi.p, zn.p, n, 32, ix->scan_perm, ix->scan_piv));
        AS_HIP(ctx, handle_malloc(&ix->packedh, packed_vecs_f16(tiles, d) * 16));
        AS_TRY(pack_f16_all(ctx, ix->items, ix->norms, n, d, ix->packedh, tiles, ix->scan_perm));
        ix->f16_err = ctx->f16_err;
        if (!ix->packed8) AS_HIP(ctx, handle_malloc(&ix->packed8, rotated_image_bytes(tiles)));
        AS_TRY(pack_rotated_from_z(ctx, Zi.p, ix->scan_perm, n, tiles, ix->packed8, &ix->rot_err));
      }
    }
    if ((scan_wide || scan_h) && !ix->packedh) {
      const uint32_t tiles = packed_tiles_f16(n);
      if (want_order) {
        AS_HIP(ctx, handle_malloc((void **)&ix->scan_perm, sizeof(uint32_t) * (size_t)n));
        AS_HIP(ctx, handle_malloc((void **)&ix->scan_piv, sizeof(float) * 256 * 128));
        AS_TRY(cluster_order(ctx, ix->items, ix->norms, n, d, ix->scan_perm, ix->scan_piv));
      }
      AS_HIP(ctx, handle_malloc(&ix->packedh, packed_vecs_f16(tiles, d) * 16));
      AS_TRY(pack_f16_all(ctx, ix->items, ix->norms, n, d, ix->packedh, tiles, ix->scan_perm));
      ix->f16_err = ctx->f16_err;
    }
    if (scan_h && ix->scan_perm && ctx->opt_search_prefilter && d > 32) {
      const uint32_t tiles = packed_tiles_f16(n);
      if (ix->rot_state == 0) {
        double tail = 1.0;
        if (!ix->rot_q) AS_HIP(ctx, handle_malloc((void **)&ix->rot_q, sizeof(float) * 32 * 128));
        AS_TRY(rotated_basis(ctx, ix->items, ix->norms, n, d, ix->rot_q, &tail));
        ix->rot_state = tail <= (double)ctx->opt_search_rot_tail_e3 * 1e-3 ? 1 : -1;
      }
      // (the directions may have been found by a small batch before, for its own image in the order of the items: the image by
      // position is this path's to build -- it used to be built only together with the directions, and a large batch that came
      // after a small one then handed the first pass a null image)
      if (ix->rot_state == 1 && !ix->packed8) {
        AS_HIP(ctx, handle_malloc(&ix->packed8, rotated_image_bytes(tiles)));
        AS_TRY(rotated_image_pack(ctx, ix->items, ix->norms, ix->scan_perm, n, d, ix->rot_q, ix->packed8, tiles, &ix->rot_err));
      }
    }
    if (!scan_wide && !scan_h && !ix->packed16) {
      AS_HIP(ctx, handle_malloc(&ix->packed16, packed_vecs_bf16(b_alloc, d) * 16));
      AS_TRY(pack_bf16(ctx, ix->items, ix->norms, nullptr, 0, n, d, ix->packed16, b_alloc));
    }
    if (!ix->lam32) {  // padded to whole 512-row super tiles (the images of both forms of the scan end there)
      const uint32_t lpad = (n + 511) / 512 * 512;
      AS_HIP(ctx, handle_malloc((void **)&ix->lam32, sizeof(float) * (size_t)lpad));
      hipLaunchKernelGGL(gather_lambda_f32_kernel, dim3(cdiv((uint64_t)lpad, 256)), dim3(256), 0, ctx->stream,
                         ix->lambdas, (const uint32_t *)nullptr, n, lpad, ix->lam32);
    }
    if ((scan_h || scan_wide) && ix->scan_perm && !ix->lam32_pos) {  // ... and by position in the ordered image
      const uint32_t lpad = (n + 511) / 512 * 512;
      AS_HIP(ctx, handle_malloc((void **)&ix->lam32_pos, sizeof(float) * (size_t)lpad));
      hipLaunchKernelGGL(gather_lambda_f32_kernel, dim3(cdiv((uint64_t)lpad, 256)), dim3(256), 0, ctx->stream, ix->lambdas,
                         (const uint32_t *)ix->scan_perm, n, lpad, ix->lam32_pos);
    }
    if (ix->scan_sample && ix->scan_ns != ns_want) {  // the option changed since the sample was drawn
      (void)handle_free(ix->scan_sample);
      (void)handle_free(ix->scan_sample_lam);
      ix->scan_sample = nullptr;
      ix->scan_sample_lam = nullptr;
    }
    if (!ix->scan_sample) {
      const uint32_t ns = ns_want;
      AS_HIP(ctx, handle_malloc((void **)&ix->scan_sample, sizeof(uint32_t) * ns));
      AS_HIP(ctx, handle_malloc((void **)&ix->scan_sample_lam, sizeof(float) * ns));
      hipLaunchKernelGGL(sample_ids_kernel, dim3(cdiv(ns, 256)), dim3(256), 0, ctx->stream, ns, n / ns, sample_half_bits(ns),
                         ix->scan_sample);
      hipLaunchKernelGGL(gather_lambda_f32_kernel, dim3(cdiv(ns, 256)), dim3(256), 0, ctx->stream, ix->lambdas,
                         ix->scan_sample, ns, ns, ix->scan_sample_lam);
      AS_HIP(ctx, hipGetLastError());
      ix->scan_ns = ns;
    }
    )
    const float *x_lam_h = ((scan_h || scan_wide) && ix->scan_perm) ? ix->lam32_pos : ix->lam32;  // what the f16 scan reads: by image position
    const uint32_t ns = ix->scan_ns;
    const uint32_t m = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(ctx->opt_search_scan_rank, 32));
    const uint32_t a_tiles = (nq + kTileRows - 1) / kTileRows, a_alloc = scan_h ? (a_tiles + 3) & ~3u : (a_tiles + 1) & ~1u;
    const uint32_t padded = a_alloc * kTileRows;  // whole super tiles of the kernel that scans (512 rows on the f16 plane, else 256)
    const uint32_t cap = (uint32_t)std::max<int64_t>(K + 1, std::min<int64_t>(ctx->opt_search_scan_cap, 1024));
    DevBuf<unsigned char> PQ, PQh;
    DevBuf<float> qlam, guess, guess2, guess_pad;
    // the looser guess of the second pass over what fails: rank 2 m + 6, as far as its ~rank x rate candidates fit the lists
    const uint32_t m2 = std::max(m, std::min<uint32_t>(std::min<uint32_t>(48, 2 * m + 6), 512u / std::max(1u, n / ns)));
    DevBuf<uint32_t> cnt, cand, unitc;
    AS_TRY(PQ.alloc(ctx, (scan_wide ? packed_vecs_f16(a_alloc, d) : packed_vecs_bf16(a_alloc, d)) * 16));
    AS_TRY(qlam.alloc(ctx, padded));
    AS_TRY(guess.alloc(ctx, nq));
    AS_TRY(guess2.alloc(ctx, nq));
    AS_TRY(guess_pad.alloc(ctx, padded));
    AS_TRY(cnt.alloc(ctx, padded));
    AS_TRY(cand.alloc(ctx, (size_t)padded * cap * 2));
    AS_TRY(unitc.alloc(ctx, 1));
    double cos_delta = cos_delta_bf16x3(d);  // bound between a listed product and the exact cosine
    DevBuf<float> floor_pad, gmb_pad;
    // a large batch is put in the order of the items' image too (the same pivots): positions in the query image -> queries
    DevBuf<uint32_t> qperm;
    // ... and on an index whose items are in their own order a batch goes in the order of its LAMBDAS: a row group of the scan
    // then spans a narrow lambda range, and a lane can rule its column out by the lambda term from two scalars (the cosine alone
    // reaching its floor sent half of all blocks of a 10 k-query batch through the walk; mfma_join_f16.hip)
    const bool wide_pre = scan_wide && ix->scan_perm && ix->rot_state == 1 && ix->packed8 && ix->rot_qw && ctx->opt_search_prefilter;
    DevBuf<double> Zq, zqn;  // wide items behind the first pass: the batch's rotated coordinates
    if (wide_pre) {
      AS_TRY(Zq.alloc(ctx, (size_t)nq * 32));
      AS_TRY(zqn.alloc(ctx, nq));
      AS_TRY(rotated_project_wide(ctx, queries, q_norms, nq, d, ix->rot_qw, Zq.p, zqn.p));
    }
    // an index with the images of its item-graph build (z_space, up to 128 columns): a batch is projected on that build's directions
    // like a batch for wide items -- its order from the build's pivots, its int8 image from the coordinates
    const bool z_pre = scan_h && ix->z_space && ix->scan_perm && ix->rot_state == 1 && ix->packed8 && ix->rot_qw && ctx->opt_search_prefilter;
    if (z_pre) {
      AS_TRY(Zq.alloc(ctx, (size_t)nq * 32));
      AS_TRY(zqn.alloc(ctx, nq));
      AS_TRY(rotated_project_wide(ctx, queries, q_norms, nq, d, ix->rot_qw, Zq.p, zqn.p));
    }
    const bool q_by_pivots = (scan_h && ix->scan_perm && nq >= 2048 && (!ix->z_space || z_pre)) || (wide_pre && nq >= 2048);
    const bool q_by_lambda = scan_h && !q_by_pivots && nq >= 1024 && ctx->opt_search_lambda_order && is_device_ptr(q_lambdas);
    const bool q_order = q_by_pivots || q_by_lambda;
    if (q_order) {
      AS_TRY(qperm.alloc(ctx, nq));
      if (q_by_pivots && z_pre) AS_TRY(order_by_pivots(ctx, Zq.p, nq, 32, ix->scan_piv, qperm.p, ix->scan_np));
      else if (q_by_pivots && wide_pre) AS_TRY(order_by_pivots(ctx, Zq.p, nq, 32, ix->scan_piv, qperm.p));
      else if (q_by_pivots) AS_TRY(order_by_pivots(ctx, queries, nq, d, ix->scan_piv, qperm.p));
      else AS_TRY(order_by_value(ctx, q_lambdas, nq, qperm.p));
    }
    ctx->stats["search_scan_lambda_order"] = q_by_lambda ? 1 : 0;
    if (scan_h) {
      AS_TRY(PQh.alloc(ctx, packed_vecs_f16(a_alloc, d) * 16));
      AS_TRY(pack_f16_all(ctx, queries, q_norms, nq, d, PQh.p, a_alloc, q_order ? qperm.p : nullptr));
      const double e = std::max(ctx->f16_err, ix->f16_err);
      cos_delta = (ctx->f16_err >= 0.0 && ix->f16_err >= 0.0) ? std::min(cos_delta_f16(d), 2.0 * e + e * e + cos_delta_f16_accum(d))
                                                                : cos_delta_f16(d);
      AS_TRY(floor_pad.alloc(ctx, padded));
      AS_TRY(gmb_pad.alloc(ctx, padded));
    }
    if (scan_wide) {  // one f16 plane: the measured plane errors of the two images (mfma_join_f16.hip)
      AS_TRY(pack_f16_all(ctx, queries, q_norms, nq, d, PQ.p, a_alloc, q_order ? qperm.p : nullptr));
      const double e = std::max(ctx->f16_err, ix->f16_err);
      cos_delta = (ctx->f16_err >= 0.0 && ix->f16_err >= 0.0) ? std::min(cos_delta_f16(d), 2.0 * e + e * e + cos_delta_f16_accum(d))
                                                                : cos_delta_f16(d);
      AS_TRY(floor_pad.alloc(ctx, padded));
      AS_TRY(gmb_pad.alloc(ctx, padded));
    } else {
      AS_TRY(pack_bf16(ctx, queries, q_norms, nullptr, 0, nq, d, PQ.p, a_alloc));
    }
    hipLaunchKernelGGL(gather_lambda_f32_kernel, dim3(cdiv(padded, 256)), dim3(256), 0, ctx->stream, q_lambdas,
                       (const uint32_t *)nullptr, nq, padded, qlam.p);
    DevBuf<float> qlam_pos;  // ... by position in the query image, for the f16 scan
    if (q_order) {
      AS_TRY(qlam_pos.alloc(ctx, padded));
      hipLaunchKernelGGL(gather_lambda_f32_kernel, dim3(cdiv(padded, 256)), dim3(256), 0, ctx->stream, q_lambdas,
                         (const uint32_t *)qperm.p, nq, padded, qlam_pos.p);
    }
    if (scan_wide && q_order) {  // (the image is in the batch's own order: the guesses come out by position and go back to the queries)
      DevBuf<float> g1, g2;
      AS_TRY(g1.alloc(ctx, padded));
      AS_TRY(g2.alloc(ctx, padded));
      AS_TRY(search_thresholds_f16_wide(ctx, PQ.p, nq, d, ix->items, ix->norms, ix->scan_sample, ns, m, qlam_pos.p, ix->scan_sample_lam,
                                        (float)alpha, (float)beta, g1.p, m2, g2.p));
      AS_TRY(scatter_by_perm(ctx, g1.p, qperm.p, nq, guess.p));
      AS_TRY(scatter_by_perm(ctx, g2.p, qperm.p, nq, guess2.p));
    } else if (scan_wide)
      AS_TRY(search_thresholds_f16_wide(ctx, PQ.p, nq, d, ix->items, ix->norms, ix->scan_sample, ns, m, qlam.p, ix->scan_sample_lam,
                                        (float)alpha, (float)beta, guess.p, m2, guess2.p));
    else
      AS_TRY(search_thresholds_bf16(ctx, PQ.p, nq, d, ix->items, ix->norms, ix->scan_sample, ns, m, qlam.p, ix->scan_sample_lam,
                                    (float)alpha, (float)beta, guess.p, m2, guess2.p));
    hipLaunchKernelGGL(scan_guess_kernel, dim3(cdiv(padded, 256)), dim3(256), 0, ctx->stream, guess.p, (const uint32_t *)nullptr,
                       nq, padded, guess_pad.p);
    AS_HIP(ctx, hipMemsetAsync(cnt.p, 0, sizeof(uint32_t) * padded, ctx->stream));
    AS_HIP(ctx, hipMemsetAsync(unitc.p, 0, sizeof(uint32_t), ctx->stream));
    if (scan_h) {
      hipLaunchKernelGGL(scan_guess_wide_kernel, dim3(cdiv(padded, 256)), dim3(256), 0, ctx->stream, guess.p,
                         (const uint32_t *)(q_order ? qperm.p : nullptr), nq, padded, (float)alpha, (float)beta, floor_pad.p, gmb_pad.p);
      // int8 first pass on the rotated images of the items (kept with the index) and of this batch: which blocks can hold a
      // pair whose cosine alone reaches what the query's guess asks for
      DevBuf<unsigned char> PQ8, mask;
      const bool pre = ix->scan_perm && ix->rot_state == 1 && ctx->opt_search_prefilter && (!ix->z_space || z_pre);
      if (pre) {
        double eq = 0.0;
        AS_TRY(PQ8.alloc(ctx, rotated_image_bytes(a_alloc)));
        if (z_pre) AS_TRY(pack_rotated_from_z(ctx, Zq.p, q_order ? qperm.p : nullptr, nq, a_alloc, PQ8.p, &eq));
        else AS_TRY(rotated_image_pack(ctx, queries, q_norms, q_order ? qperm.p : nullptr, nq, d, ix->rot_q, PQ8.p, a_alloc, &eq));
        const double E = std::max(eq, ix->rot_err), delta8 = (2.0 * E + E * E) * (1.0 + 1e-6) + 1e-4;
        const uint32_t T2i = (n + 511) / 512, a_supers = a_alloc / 4;
        AS_TRY(mask.alloc(ctx, (size_t)a_supers * 2 * T2i * 8));
        AS_TRY(prefilter_i8_rect(ctx, ix->packed8, n, PQ8.p, a_supers, floor_pad.p, delta8, mask.p, "search_scan_prefilter"));
      }
      ctx->stats["search_scan_prefilter"] = pre ? 1 : 0;
      ctx->stats["search_scan_order"] = q_by_pivots ? 1 : 0;
      AS_TRY(join_f16_search(ctx, ix->packedh, n, d, PQh.p, nq, floor_pad.p, gmb_pad.p, q_order ? qlam_pos.p : qlam.p, x_lam_h,
                             (float)alpha, (float)beta, cap, cnt.p, cand.p, unitc.p, "search_scan_f16", ix->scan_perm,
                             q_order ? qperm.p : nullptr, pre ? mask.p : nullptr));
    } else if (scan_wide) {
      // (floor / g - beta by position of the query image; padded to whole 512-position super tiles of the first pass: +inf beyond nq)
      const uint32_t a_alloc4 = (a_alloc + 3) & ~3u, padded4 = a_alloc4 * kTileRows;
      DevBuf<float> floor4, gmb4;
      AS_TRY(floor4.alloc(ctx, padded4));
      AS_TRY(gmb4.alloc(ctx, padded4));
      hipLaunchKernelGGL(scan_guess_wide_kernel, dim3(cdiv(padded4, 256)), dim3(256), 0, ctx->stream, guess.p,
                         (const uint32_t *)(q_order ? qperm.p : nullptr), nq, padded4, (float)alpha, (float)beta, floor4.p, gmb4.p);
      DevBuf<unsigned char> PQ8, mask;
      if (wide_pre) {
        double eq = 0.0;
        AS_TRY(PQ8.alloc(ctx, rotated_image_bytes(a_alloc4)));
        AS_TRY(pack_rotated_from_z(ctx, Zq.p, q_order ? qperm.p : nullptr, nq, a_alloc4, PQ8.p, &eq));
        const double E = std::max(eq, ix->rot_err), delta8 = (2.0 * E + E * E) * (1.0 + 1e-6) + 1e-4;
        const uint32_t T2i = (n + 511) / 512, a_supers = a_alloc4 / 4;
        AS_TRY(mask.alloc(ctx, (size_t)a_supers * 2 * T2i * 8));
        AS_TRY(prefilter_i8_rect(ctx, ix->packed8, n, PQ8.p, a_supers, floor4.p, delta8, mask.p, "search_scan_prefilter"));
      }
      ctx->stats["search_scan_prefilter"] = wide_pre ? 1 : 0;
      ctx->stats["search_scan_order"] = (q_by_pivots && wide_pre) ? 1 : 0;
      AS_TRY(join_f16_wide_search(ctx, ix->packedh, n, d, PQ.p, nq, floor4.p, gmb4.p, q_order ? qlam_pos.p : qlam.p, x_lam_h, (float)alpha,
                                  (float)beta, cap, cnt.p, cand.p, "search_scan_f16", ix->scan_perm, q_order ? qperm.p : nullptr,
                                  wide_pre ? mask.p : nullptr));
    } else {
      AS_TRY(join_bf16_search(ctx, ix->packed16, n, d, PQ.p, nq, guess_pad.p, qlam.p, ix->lam32, (float)alpha, (float)beta, cap,
                              cnt.p, cand.p, unitc.p, "search_scan_bf16"));
    }
    {
      // unlisted items: approximate score < g_q.  Approximate vs exact score: alpha x the cosine bound of the pass,
      // the lambda term in f32 with beta folded into its operands (four roundings on numbers of size <= lsum),
      // the f32 roundings of alpha, beta and of the two fused multiply-adds.
      const double lam_err = std::fabs(beta) * (lsum * 0x1p-22 + 0x1p-21);
      const SearchBound bnd{std::fabs(alpha) * cos_delta + lam_err + (std::fabs(alpha) + std::fabs(beta)) * 5.2e-7 + 1e-12};
      ProfScope ps(ctx, "search_rerank");
      // the lists carry the approximate score minus beta: candidates that cannot reach the top K are dropped before
      // their rows are fetched, as in the item graph (prune_candidates)
      const uint32_t kmax = std::min<uint32_t>(cap, std::max<uint32_t>(128, 2 * K));
      const size_t lds128 = (size_t)4 * (1024 + (size_t)8 * kRrPitch + rr_key_bytes(cap, kmax) + (size_t)cap * 4);
      if (d == 128 && lds128 <= 160 * 1024 && ((uintptr_t)ix->items & 15) == 0) {
        AS_HIP(ctx, hipFuncSetAttribute((const void *)rerank_join_d128_kernel<SearchScorePolicy, SearchBound, 8, 1>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds128));
        hipLaunchKernelGGL((rerank_join_d128_kernel<SearchScorePolicy, SearchBound, 8, 1>),
                           dim3(std::min<uint64_t>(cdiv(nq, 4), 256 * 16)), dim3(256), lds128, ctx->stream, ix->items, queries,
                           (const uint32_t *)nullptr, 0u, nq, guess_pad.p, 1u, cnt.p, cand.p, cap, cap, K,
                           (const uint8_t *)nullptr, pol, bnd, bnd.delta, 0, kmax, out_key, out_idx, fail1.p, fail_count.p,
                           err.p);
      } else {
        const size_t lds = (size_t)4 * (d + cap) * sizeof(double) + (size_t)4 * cap * sizeof(uint32_t);
        AS_HIP(ctx, hipFuncSetAttribute((const void *)rerank_join_kernel<SearchScorePolicy, SearchBound>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((rerank_join_kernel<SearchScorePolicy, SearchBound>), dim3(std::min<uint64_t>(cdiv(nq, 4), 8192)),
                           dim3(256), lds, ctx->stream, ix->items, queries, (const uint32_t *)nullptr, d, 0u, nq, guess_pad.p, 1u,
                           cnt.p, cand.p, cap, cap, K, (const uint8_t *)nullptr, pol, bnd, bnd.delta, 0, out_key, out_idx, fail1.p,
                           fail_count.p, err.p);
      }
      AS_HIP(ctx, hipGetLastError());
    }
    {  // statistics: candidates per query (read back with the count of open queries just below: one wait for the stream, not two)
      auto parked = std::make_unique<DeferredStat>();
      AS_TRY(parked->buf.alloc(ctx, 3));
      AS_HIP(ctx, hipMemsetAsync(parked->buf.p, 0, 3 * sizeof(unsigned long long), ctx->stream));
      hipLaunchKernelGGL(join_count_stats_kernel, dim3(64), dim3(256), 0, ctx->stream, cnt.p, nq, cap, parked->buf.p);
      parked->host.resize(3);
      parked->keys = {{"search_scan_candidates", 0}, {"search_scan_max_per_query", 1}, {"search_scan_overflow_queries", 2}};
      ctx->deferred.push_back(std::move(parked));
    }
    AS_TRY(read_u32(ctx, fail_count.p, &n_fail));
    ctx->stats["search_scan_uncertified_first"] = n_fail;
    // ---- stage 0b: the queries whose guess was too high, again with the looser one ----
    if (n_fail > 0 && n_fail <= nq / 4 + 16) {
      const uint32_t nf = n_fail;
      const uint32_t bt = (nf + kTileRows - 1) / kTileRows, b_al = scan_h ? (bt + 3) & ~3u : (bt + 1) & ~1u, pad_b = b_al * kTileRows;
      const uint32_t cap2 = 1024;
      DevBuf<unsigned char> PQb;
      DevBuf<float> qlam_b, guess_b;
      DevBuf<uint32_t> cnt_b, cand_b, oi, fail_b, ids_b;
      DevBuf<double> ok;
      AS_TRY(PQb.alloc(ctx, ((scan_wide || scan_h) ? packed_vecs_f16(b_al, d) : packed_vecs_bf16(b_al, d)) * 16));
      AS_TRY(qlam_b.alloc(ctx, pad_b));
      AS_TRY(guess_b.alloc(ctx, pad_b));
      AS_TRY(cnt_b.alloc(ctx, pad_b));
      AS_TRY(cand_b.alloc(ctx, (size_t)pad_b * cap2 * 2));
      AS_TRY(ok.alloc(ctx, (size_t)nf * K));
      AS_TRY(oi.alloc(ctx, (size_t)nf * K));
      AS_TRY(fail_b.alloc(ctx, nf));
      AS_TRY(ids_b.alloc(ctx, nf));
      if (scan_wide || scan_h) AS_TRY(pack_f16(ctx, queries, q_norms, fail1.p, 0, nf, d, PQb.p, b_al));
      else AS_TRY(pack_bf16(ctx, queries, q_norms, fail1.p, 0, nf, d, PQb.p, b_al));
      hipLaunchKernelGGL(gather_lambda_f32_kernel, dim3(cdiv(pad_b, 256)), dim3(256), 0, ctx->stream, q_lambdas, fail1.p, nf, pad_b,
                         qlam_b.p);
      hipLaunchKernelGGL(scan_guess_kernel, dim3(cdiv(pad_b, 256)), dim3(256), 0, ctx->stream, guess2.p, fail1.p, nf, pad_b,
                         guess_b.p);
      AS_HIP(ctx, hipMemsetAsync(cnt_b.p, 0, sizeof(uint32_t) * pad_b, ctx->stream));
      AS_HIP(ctx, hipMemsetAsync(unitc.p, 0, sizeof(uint32_t), ctx->stream));
      AS_HIP(ctx, hipMemsetAsync(fail_count.p, 0, sizeof(uint32_t), ctx->stream));
      if (scan_wide || scan_h) {
        DevBuf<float> floor_b, gmb_b;
        AS_TRY(floor_b.alloc(ctx, pad_b));
        AS_TRY(gmb_b.alloc(ctx, pad_b));
        hipLaunchKernelGGL(scan_guess_wide_kernel, dim3(cdiv(pad_b, 256)), dim3(256), 0, ctx->stream, guess2.p, fail1.p, nf, pad_b,
                           (float)alpha, (float)beta, floor_b.p, gmb_b.p);
        if (scan_h)
          AS_TRY(join_f16_search(ctx, ix->packedh, n, d, PQb.p, nf, floor_b.p, gmb_b.p, qlam_b.p, x_lam_h, (float)alpha, (float)beta, cap2,
                                 cnt_b.p, cand_b.p, unitc.p, "search_scan_f16", ix->scan_perm));
        else
          AS_TRY(join_f16_wide_search(ctx, ix->packedh, n, d, PQb.p, nf, floor_b.p, gmb_b.p, qlam_b.p, x_lam_h, (float)alpha, (float)beta,
                                    cap2, cnt_b.p, cand_b.p, "search_scan_f16", ix->scan_perm));
      } else {
        AS_TRY(join_bf16_search(ctx, ix->packed16, n, d, PQb.p, nf, guess_b.p, qlam_b.p, ix->lam32, (float)alpha, (float)beta, cap2,
                                cnt_b.p, cand_b.p, unitc.p, "search_scan_bf16"));
      }
      {
        const double lam_err = std::fabs(beta) * (lsum * 0x1p-22 + 0x1p-21);
        const SearchBound bnd{std::fabs(alpha) * cos_delta + lam_err + (std::fabs(alpha) + std::fabs(beta)) * 5.2e-7 + 1e-12};
        ProfScope ps(ctx, "search_rerank");
        const size_t lds = (size_t)4 * (d + cap2) * sizeof(double) + (size_t)4 * cap2 * sizeof(uint32_t);
        AS_HIP(ctx, hipFuncSetAttribute((const void *)rerank_join_kernel<SearchScorePolicy, SearchBound>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((rerank_join_kernel<SearchScorePolicy, SearchBound>), dim3(std::min<uint64_t>(cdiv(nf, 4), 8192)),
                           dim3(256), lds, ctx->stream, ix->items, queries, (const uint32_t *)fail1.p, d, 0u, nf, guess_b.p, 1u,
                           cnt_b.p, cand_b.p, cap2, cap2, K, (const uint8_t *)nullptr, pol, bnd, bnd.delta, 0, ok.p, oi.p, fail_b.p,
                           fail_count.p, err.p);
        AS_HIP(ctx, hipGetLastError());
      }
      hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv((uint64_t)nf * K, 256)), dim3(256), 0, ctx->stream, fail1.p, nf, K, ok.p,
                         oi.p, out_key, out_idx);
      uint32_t nf2 = 0;
      AS_TRY(read_u32(ctx, fail_count.p, &nf2));
      if (nf2) {  // what is still open: positions in the gathered list -> query ids, back into fail1
        hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(nf2, 256)), dim3(256), 0, ctx->stream, fail1.p, fail_b.p, nf2, 0u, ids_b.p);
        AS_HIP(ctx, hipMemcpyAsync(fail1.p, ids_b.p, sizeof(uint32_t) * nf2, hipMemcpyDeviceToDevice, ctx->stream));
        AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
      }
      n_fail = nf2;
    }
  } else {
    const uint32_t a_tiles = (nq + kTileRows - 1) / kTileRows;
    const uint32_t a_items = use_f32 ? (a_tiles + 1) / 2 : a_tiles;
    pb.splits = clamp_splits(pick_splits_scan(a_items, b_tiles, n), d, KC);
    DevBuf<double> ckey;
    DevBuf<uint32_t> cidx;
    AS_TRY(ckey.alloc(ctx, (size_t)nq * pb.splits * KC));
    AS_TRY(cidx.alloc(ctx, (size_t)nq * pb.splits * KC));
    pb.cand_key = ckey.p;
    pb.cand_idx = cidx.p;
    if (use_f32) {
      AS_TRY(ensure32());
      const uint32_t at = (a_tiles + 1) & ~1u;
      DevBuf<float> PA;
      AS_TRY(PA.alloc(ctx, packed_elems(at, d)));
      AS_TRY(pack_f32(ctx, queries, q_norms, nullptr, 0, nq, d, PA.p, at));
      // lambda term of the epilogue: f32 when the lambdas are small enough for its error to stay far below
      // the cosine error (the usual case: lambda in [0, 1]), f64 otherwise (mfma_topk_f32.hip)
      DevBuf<unsigned long long> amax;
      AS_TRY(amax.alloc(ctx, 2));
      AS_HIP(ctx, hipMemsetAsync(amax.p, 0, 2 * sizeof(unsigned long long), ctx->stream));
      hipLaunchKernelGGL(absmax_kernel, dim3(std::min<uint64_t>(1024, cdiv(nq, 256))), dim3(256), 0, ctx->stream,
                         q_lambdas, (uint64_t)nq, amax.p);
      hipLaunchKernelGGL(absmax_kernel, dim3(std::min<uint64_t>(1024, cdiv(n, 256))), dim3(256), 0, ctx->stream,
                         ix->lambdas, (uint64_t)n, amax.p + 1);
      AS_HIP(ctx, hipGetLastError());
      double hmax[2] = {0.0, 0.0};
      AS_HIP(ctx, copy_d2h(ctx, hmax, amax.p, sizeof(hmax)));
      const double lsum = hmax[0] + hmax[1];
      pb.lam_f32 = lsum <= 64.0 ? 1 : 0;
      ctx->stats["search_lambda_term_f32"] = pb.lam_f32;
      AS_TRY(candidates_f32(ctx, pb, PA.p, a_tiles, ix->packed32, b_tiles, "search_mfma_candidates_f32"));
      // f32 pass: cosine error scaled by |alpha|, the lambda term's error scaled by |beta| (f32 form only),
      // plus the f32 roundings of a sum bounded by |alpha| + |beta| (product, fma)
      const double beta_abs = std::fabs(1.0 - alpha);
      const double lam_err = pb.lam_f32 ? beta_abs * (lsum * 0x1p-23 + 0x1p-22) : 0.0;
      SearchBound bnd{std::fabs(alpha) * cos_delta_f32(d) + lam_err + (std::fabs(alpha) + beta_abs) * 2.6e-7 + 1e-12};
      AS_TRY(launch_rerank(ctx, "search_rerank", queries, ix->items, d, nullptr, 0u, nq, 0u, K, KC, pb.splits, ckey.p,
                           cidx.p, (const uint8_t *)nullptr, pol, bnd, out_key, out_idx, fail1.p, fail_count.p, err.p));
    } else {
      AS_TRY(ensure64());
      DevBuf<double> PA;
      AS_TRY(PA.alloc(ctx, packed_elems(a_tiles, d)));
      AS_TRY(pack_f64(ctx, queries, q_norms, nullptr, 0, nq, d, PA.p, a_tiles));
      AS_TRY(candidates_f64(ctx, pb, PA.p, a_tiles, ix->packed, b_tiles, "search_mfma_candidates"));
      SearchBound bnd{std::fabs(alpha) * cos_delta_f64(d) + 1e-15};
      AS_TRY(launch_rerank(ctx, "search_rerank", queries, ix->items, d, nullptr, 0u, nq, 0u, K, KC, pb.splits, ckey.p,
                           cidx.p, (const uint8_t *)nullptr, pol, bnd, out_key, out_idx, fail1.p, fail_count.p, err.p));
    }
    AS_TRY(read_u32(ctx, fail_count.p, &n_fail));
  }
  ctx->stats["search_uncertified_rows"] = n_fail;
  ctx->stats["search_uncertified_rows_stage2"] = 0;
  ctx->stats["search_cand_capacity"] = KC;
  ctx->stats["search_col_splits"] = pb.splits;

  const uint32_t *exact_ids = fail1.p;
  uint32_t n_exact = n_fail;
  DevBuf<uint32_t> fail2, ids2;
  if (mfma_ok && (use_f32 || scan) && !wide && n_fail > 0 && !exact_direct(n_fail, n)) {
    const uint32_t a_tiles = (n_fail + kTileRows - 1) / kTileRows;
    pb.arows = ARows{fail1.p, 0};
    pb.nA = n_fail;
    pb.splits = clamp_splits(pick_splits(a_tiles, b_tiles), d, KC);
    DevBuf<double> ckey, PA, ok;
    DevBuf<uint32_t> cidx, oi;
    AS_TRY(ckey.alloc(ctx, (size_t)n_fail * pb.splits * KC));
    AS_TRY(cidx.alloc(ctx, (size_t)n_fail * pb.splits * KC));
    pb.cand_key = ckey.p;
    pb.cand_idx = cidx.p;
    AS_TRY(ensure64());
    AS_TRY(PA.alloc(ctx, packed_elems(a_tiles, d)));
    AS_TRY(pack_f64(ctx, queries, q_norms, fail1.p, 0, n_fail, d, PA.p, a_tiles));
    AS_TRY(candidates_f64(ctx, pb, PA.p, a_tiles, ix->packed, b_tiles, "search_mfma_candidates"));
    AS_TRY(ok.alloc(ctx, (size_t)n_fail * K));
    AS_TRY(oi.alloc(ctx, (size_t)n_fail * K));
    AS_TRY(fail2.alloc(ctx, n_fail));
    AS_HIP(ctx, hipMemsetAsync(fail_count.p, 0, sizeof(uint32_t), ctx->stream));
    SearchBound bnd{std::fabs(alpha) * cos_delta_f64(d) + 1e-15};
    AS_TRY(launch_rerank(ctx, "search_rerank", queries, ix->items, d, fail1.p, 0u, n_fail, 0u, K, KC, pb.splits, ckey.p,
                         cidx.p, (const uint8_t *)nullptr, pol, bnd, ok.p, oi.p, fail2.p, fail_count.p, err.p));
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv((uint64_t)n_fail * K, 256)), dim3(256), 0, ctx->stream, fail1.p,
                       n_fail, K, ok.p, oi.p, out_key, out_idx);
    uint32_t n_fail2 = 0;
    AS_TRY(read_u32(ctx, fail_count.p, &n_fail2));
    ctx->stats["search_uncertified_rows_stage2"] = n_fail2;
    n_exact = n_fail2;
    if (n_fail2) {
      AS_TRY(ids2.alloc(ctx, n_fail2));
      hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(n_fail2, 256)), dim3(256), 0, ctx->stream, fail1.p, fail2.p,
                         n_fail2, 0u, ids2.p);
      exact_ids = ids2.p;
    }
  }
  int h_err = 0;
  AS_HIP(ctx, copy_d2h(ctx, &h_err, err.p, sizeof(int)));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (h_err) return fail(ctx, AS_ERR_NAN_SCORE, as_status_message(AS_ERR_NAN_SCORE));
  // Items of 256 .. 1024 values and a handful of queries left open: the quick pass over the items (stream_scan.hip, what a small
  // batch takes from the start), 32 queries at a time -- search_exact puts ONE wave on a query's 6 GB: 0.9 s a query at 1 M x 768,
  // and two queries left of 100 000 made a 41 ms batch take 1.8 s (a draw of lambdas in 15: scripts/time_search.py, round 5).
  ctx->stats["search_leftover_stream_scan"] = 0;
  // (up to 4 096 of them: a pass of 32 queries streams the items once, ~1 ms per 6 GB, where search_exact's waves each take a second)
  if (n_exact && wide && n_exact <= 4096u && K == k && ctx->opt_stream_scan && ctx->opt_search_stream_quick &&
      stream_quick_wide_applies(std::min<uint32_t>(n_exact, 32), n, d, K)) {
    DevBuf<double> fk;
    DevBuf<uint32_t> fi, redo_d;
    AS_TRY(fk.alloc(ctx, (size_t)n_exact * K));
    AS_TRY(fi.alloc(ctx, (size_t)n_exact * K));
    AS_HIP(ctx, hipMemsetAsync(err.p, 0, sizeof(int), ctx->stream));
    std::vector<uint32_t> redo_all;
    for (uint32_t c0 = 0; c0 < n_exact; c0 += 32) {
      const uint32_t cn = std::min<uint32_t>(32, n_exact - c0);
      std::vector<uint32_t> redo;
      AS_TRY(stream_quick_wide_search(ctx, queries, cn, ix->items, n, d, K, pol, fk.p + (size_t)c0 * K, fi.p + (size_t)c0 * K, err.p, &redo, exact_ids + c0));
      for (uint32_t r : redo) redo_all.push_back(c0 + r);
    }
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv((uint64_t)n_exact * K, 256)), dim3(256), 0, ctx->stream, exact_ids, n_exact, K, fk.p, fi.p, out_key,
                       out_idx);
    AS_HIP(ctx, hipGetLastError());
    int h_err1 = 0;
    AS_HIP(ctx, copy_d2h(ctx, &h_err1, err.p, sizeof(int)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_err1) return fail(ctx, AS_ERR_NAN_SCORE, as_status_message(AS_ERR_NAN_SCORE));
    ctx->stats["search_leftover_stream_scan"] = n_exact;
    if (redo_all.empty()) return AS_OK;
    // (a query whose list ran over in the quick pass: the exact scan for those)
    const uint32_t nr = (uint32_t)redo_all.size();
    AS_TRY(redo_d.alloc(ctx, nr));
    AS_HIP(ctx, copy_h2d(ctx, redo_d.p, redo_all.data(), sizeof(uint32_t) * nr));
    AS_TRY(ids2.alloc(ctx, nr));
    hipLaunchKernelGGL(gather_u32_kernel, dim3(cdiv(nr, 256)), dim3(256), 0, ctx->stream, exact_ids, redo_d.p, nr, 0u, ids2.p);
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    exact_ids = ids2.p;
    n_exact = nr;
  }
  if (n_exact) {
    DevBuf<double> fk;
    DevBuf<uint32_t> fi;
    AS_TRY(fk.alloc(ctx, (size_t)n_exact * K));
    AS_TRY(fi.alloc(ctx, (size_t)n_exact * K));
    AS_TRY(search_exact(ctx, ix, queries, q_norms, q_lambdas, exact_ids, n_exact, nq, K, alpha, fi.p, fk.p));
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv((uint64_t)n_exact * K, 256)), dim3(256), 0, ctx->stream,
                       exact_ids, n_exact, K, fk.p, fi.p, out_key, out_idx);
    AS_HIP(ctx, hipGetLastError());
  }
  return AS_OK;
}
#undef AS_IX_BUILD

}  // namespace asx
