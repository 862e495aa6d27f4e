// rerank_split.h -- exact re-rank of the joined candidate lists of the item graph: the pruning kernel and the record layout.
// (included by mfma_drivers.hip behind prune_candidates / GraphBound; the scoring and ranking kernels: rerank_pairs.h)
//
// rerank_prune_kernel streams the lists (one wave per row), drops the candidates whose stored approximate cosine cannot reach the
// row's top K (per-pair bounds from the rows' own plane errors) and leaves one record per row: count, flags, theta, the row's
// norm, up to 124 / 252 survivor ids and their norms.  rerank_pairs.h turns the records into a stream of row batches, scores every
// pair once, writes each key over the norm of its slot, and ranks.
#pragma once

namespace asx {

// A record of SLOTS x 64 id slots (SLOTS = 2 for 128-wide rows, 4 for wider ones, whose lists prune less): u32 words
constexpr int kRsIdsAt = 4;            // word 0: count | overflow << 8 | force << 9; 1: theta; 2-3: norm of the row
__host__ __device__ constexpr int rs_max_ids(int slots) { return 64 * slots - kRsIdsAt; }  // survivor ids behind the header
__host__ __device__ constexpr int rs_norms_at(int slots) { return 64 * slots; }            // their norms (f64): what GraphDistPolicy::key reads per pair
__host__ __device__ constexpr int rs_words(int slots) { return 192 * slots; }              // 1.5 KB / 3 KB per row
constexpr unsigned int kRsScoredSlots = 256;  // words the pruning kernel's count of survivors is spread over
constexpr int kRsRows = 16;            // candidate rows per batch
constexpr int kRsPitch = 1040;         // LDS pitch of a staged row (bytes)

__global__ __launch_bounds__(256) void rerank_prune_kernel(uint32_t row_begin, uint32_t row_count, const float *__restrict__ theta,
                                                           const double *__restrict__ norms, uint32_t n_src,
                                                           const uint32_t *__restrict__ cnt_src, const uint32_t *__restrict__ cand_src,
                                                           uint32_t cap, uint32_t cmax, uint32_t K,
                                                           const uint8_t *__restrict__ force_exact, double prune_delta, int prune_rectify,
                                                           uint32_t *__restrict__ records, int slots,
                                                           const uint32_t *__restrict__ perm, const uint32_t *__restrict__ inv,
                                                           unsigned long long *__restrict__ scored_total,
                                                           const uint32_t *__restrict__ a_ids, const uint32_t *__restrict__ order,
                                                           const float *__restrict__ err_pos, float accum_delta, float floor_delta,
                                                           const double *__restrict__ norms_pos, const float *__restrict__ err_self = nullptr,
                                                           const uint32_t *__restrict__ row_list = nullptr, const uint32_t *__restrict__ row_list_count = nullptr) {
  // row_list (or null): the places to visit, *row_list_count of them (the long lists rerank_prune_halves_kernel left behind)
  // err_self (or null): the rows' own errors BY ROW ID, where the lists hold row ids and the rows are a list (a build sharded by rows)
  // records are written by PLACE r of the visiting order: the rows of the range in their own order, or (order: place -> row of
  // the whole matrix, row_begin = 0) in the order of the join's image, where the list of place r sits at position r
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t kRsMaxIds = (uint32_t)rs_max_ids(slots), kRsWords = (uint32_t)rs_words(slots), kRsNormsAt = (uint32_t)rs_norms_at(slots);
  float *es = reinterpret_cast<float *>(smem) + (size_t)wv * 3 * cmax;
  uint32_t *ei = reinterpret_cast<uint32_t *>(es + cmax);
  float *ed = es + 2 * cmax;  // the bound of each pair (err_pos: the rows' plane errors by image position = what the lists hold)
  unsigned long long scored = 0;  // candidates this wave hands to the scoring kernel (the rows it fetches: bench.py's algorithmic bytes)
  // (Tried in round 5: asking for the NEXT row's count and first 64 entries while this row is pruned -- two of the four dependent
  // round trips of a row a row ahead.  1 M rows: 0.96 -> 1.18 ms: the 64 entries read whatever the count are a third more bytes
  // than the lists hold and the registers they wait in cost occupancy.  Not kept.)
  const uint32_t n_visit = row_list ? *row_list_count : row_count;
  for (uint32_t rv = blockIdx.x * 4 + wv; rv < n_visit; rv += gridDim.x * 4) {
    const uint32_t r = row_list ? row_list[rv] : rv;
    // (a_ids: the rows of the range as a list -- a rank of a build sharded by image positions finishes the rows perm[pb .. pe))
    const uint32_t a = order ? order[r] : (a_ids ? a_ids[r] : row_begin + r);
    // lists kept by image POSITION (the own join on a permuted image, row_order.hip): row a's list sits at inv[a] and
    // holds positions; what leaves this kernel is row ids again
    // (order without inv: the rows in the order of the image, the lists by ROW -- rows wider than 128)
    const uint32_t lr = order ? (inv ? r : a - row_begin) : (inv ? inv[a] : r);
    // what the record's header needs, and the row's own error: asked for now, used at the end of the row
    const float th_a = theta[a];
    const double nrm_a = norms[a];
    const bool force = force_exact && force_exact[a];
    const float e_self = err_self ? err_self[a] : (err_pos ? err_pos[lr] : 0.f);
    uint32_t C = 0;
    bool overflow = false;
    for (uint32_t sidx = 0; sidx < n_src; ++sidx) {
      const uint32_t total = cnt_src[(size_t)sidx * row_count + lr];
      uint32_t take = min(total, cap);
      overflow = overflow || total > cap;
      if (C + take > cmax) {
        overflow = true;
        take = cmax - C;
      }
      const uint2 *src = reinterpret_cast<const uint2 *>(cand_src) + ((size_t)sidx * row_count + lr) * cap;
      if (err_pos) {  // the bound of each pair as its entry passes by (prune_candidates_pairwise: es = the LOWER end of the pair's cosine)
        // (four entries per lane in flight, then their four gathered errors: two dependent levels per 256 entries, not eight)
        for (uint32_t c0 = 0; c0 < take; c0 += 256) {
          uint2 e[4];
          float ej[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const uint32_t c = c0 + (uint32_t)u * 64u + (uint32_t)lane;
            e[u] = c < take ? src[c] : make_uint2(0u, 0u);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const uint32_t c = c0 + (uint32_t)u * 64u + (uint32_t)lane;
            ej[u] = c < take ? err_pos[e[u].x] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const uint32_t c = c0 + (uint32_t)u * 64u + (uint32_t)lane;
            if (c < take) {
              const float dc = pair_bound(e_self, ej[u], accum_delta, floor_delta);
              ei[C + c] = e[u].x;
              ed[C + c] = dc;
              es[C + c] = __uint_as_float(e[u].y) - dc;
            }
          }
        }
      } else {
        for (uint32_t c = lane; c < take; c += 64) {
          const uint2 e = src[c];
          ei[C + c] = e.x;
          es[C + c] = __uint_as_float(e.y);
        }
      }
      C += take;
    }
    wave_sync();
    if (err_pos) C = prune_candidates_pairwise(ei, es, ed, C, K, accum_delta + e_self, prune_rectify, lane);
    else C = prune_candidates(ei, es, C, K, prune_delta, prune_rectify, lane);
    if (C > kRsMaxIds) {  // more survivors than a record holds: the row goes on to the next stage
      overflow = true;
      C = kRsMaxIds;
    }
    scored += C;
    uint32_t *rec = records + (size_t)r * kRsWords;
    for (uint32_t c = lane; c < C; c += 64) {
      const uint32_t pj = ei[c];
      const uint32_t j = perm ? perm[pj] : pj;
      rec[kRsIdsAt + c] = j;
      // (norms_pos: the norms by image position, so that the norm does not wait for the row id)
      reinterpret_cast<double *>(rec + kRsNormsAt)[c] = (perm && norms_pos) ? norms_pos[pj] : norms[j];
    }
    if (lane == 0) {
      rec[0] = C | (overflow ? 0x100u : 0u) | (force ? 0x200u : 0u);
      rec[1] = __float_as_uint(th_a);
      const unsigned long long nb = (unsigned long long)__double_as_longlong(nrm_a);
      rec[2] = (uint32_t)nb;
      rec[3] = (uint32_t)(nb >> 32);
    }
    wave_sync();
  }
  if (scored_total) block_add_u64(scored_total, kRsScoredSlots, scored);  // (one word took 0.36 ms of adds: common.h)
}

// ---- the same pruning for the usual short list, a row per HALF wave, the list in registers (round 6) ---------------------------------
// rerank_prune_kernel keeps a row's list in 3 x cmax words of LDS per wave -- 24 KB a workgroup at 512 slots, five waves a SIMD --
// and a row is five dependent round trips (place -> count -> entries -> their errors -> row ids and norms of the survivors) with one
// row per wave in flight: 0.98 ms at 1 M rows, 0.11 of HBM.  A list of at most 128 entries (the mean is 46 - 61) fits four registers
// of a half wave: no LDS, two rows per wave, seven waves a SIMD -- 0.63 ms, and from there the kernel is bound by its vector
// instructions (0.78 of the issue cycles; neither more waves nor asking for the next row's entries ahead moved it), so a pair of
// rows whose lists both have at most 64 entries -- most pairs -- runs the two-register body.  Longer lists are left to
// rerank_prune_kernel (long_rows / long_count: their places; 0.4 % of the rows).  tau is the bisection of prune_tau on the same keys
// with the same stops, so the records are the same bits.
__device__ __forceinline__ uint32_t rs_half_ballot(bool p, int half) {
  const unsigned long long b = __ballot(p);
  return half ? (uint32_t)(b >> 32) : (uint32_t)b;
}
constexpr uint32_t kRsHalfList = 128;  // entries of a list the half-wave form takes
struct RsHalfRow {  // what the body needs of the half's row
  const uint2 *src;
  uint32_t *rec;
  uint32_t C0;
  float e_self;
};
// NU registers of 32 entries: returns the survivors written (the caller caps the count at the record's capacity first)
template <int NU>
__device__ __forceinline__ uint32_t rs_prune_half_row(const RsHalfRow &h, int half, int hl, uint32_t K, int prune_rectify, const float *__restrict__ err_pos,
                                                      float accum_delta, float floor_delta, const uint32_t *__restrict__ perm,
                                                      const double *__restrict__ norms_pos, const double *__restrict__ norms, uint32_t max_ids,
                                                      uint32_t norms_at, bool *overflow) {
  const uint32_t C0 = h.C0;
  uint2 e[NU];
  float dc[NU], lo[NU];
  {
    float ej[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const uint32_t c = (uint32_t)u * 32u + (uint32_t)hl;
      e[u] = c < C0 ? h.src[c] : make_uint2(0u, 0u);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const uint32_t c = (uint32_t)u * 32u + (uint32_t)hl;
      ej[u] = c < C0 ? err_pos[e[u].x] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      dc[u] = pair_bound(h.e_self, ej[u], accum_delta, floor_delta);
      lo[u] = __uint_as_float(e[u].y) - dc[u];
    }
  }
  // tau of the half's list: prune_tau's bisection, each half on its own bits of the ballots
  bool keep_all = C0 <= K;
  float tau = 0.f;
  {
    uint32_t key[NU];
    bool has_nan = false;
    uint32_t kmax = 0u, kmin = 0xFFFFFFFFu;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const uint32_t c = (uint32_t)u * 32u + (uint32_t)hl;
      key[u] = 0u;
      if (c < C0) {
        has_nan = has_nan || lo[u] != lo[u];
        const uint32_t b = __float_as_uint(lo[u]);
        key[u] = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
        kmax = max(kmax, key[u]);
        kmin = min(kmin, key[u]);
      }
    }
    if (rs_half_ballot(has_nan, half)) keep_all = true;  // NaN scores: leave the list alone
    for (int o = 16; o > 0; o >>= 1) {
      kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, o, 64));
      kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, o, 64));
    }
    const uint32_t diff = kmax ^ kmin;
    int top = diff ? 31 - __builtin_clz(diff) : -1;  // highest bit in which two scores differ
    uint32_t prefix = top >= 31 ? 0u : (kmax & ~((2u << (top < 0 ? 0 : top)) - 1u));
    if (top < 0) prefix = kmax;
    int stop = 0;
    const double delta = (double)(accum_delta + h.e_self);  // (prune_candidates_pairwise's step)
    if (delta > 0.0) {
      const float top_score = fabsf(__uint_as_float((kmax & 0x80000000u) ? (kmax & 0x7FFFFFFFu) : ~kmax));
      const uint32_t tb = __float_as_uint(top_score) >> 23;
      const int ex = tb ? (int)tb - 126 : 0;
      const unsigned long long db = (unsigned long long)__double_as_longlong(delta / 8.0) >> 52;
      const int lg = db ? (int)(db & 0x7FFu) - 1023 : -1023;
      stop = lg - (ex - 24);
      stop = stop < 0 ? 0 : stop;
    }
    if (keep_all) top = -1;  // (a half with nothing to decide takes no part in the rounds)
    const int top_o = __shfl_xor(top, 32, 64), stop_o = __shfl_xor(stop, 32, 64);
    const int hi_bit = __builtin_amdgcn_readfirstlane(top > top_o ? top : top_o);  // (the same in both halves: the rounds are the wave's)
    const int lo_bit = __builtin_amdgcn_readfirstlane(top < 0 ? stop_o : (top_o < 0 ? stop : (stop < stop_o ? stop : stop_o)));
    for (int bit = hi_bit; bit >= lo_bit && bit >= 0; --bit) {
      const uint32_t trial = prefix | (1u << bit);
      uint32_t cnt = 0;
#pragma unroll
      for (int u = 0; u < NU; ++u) cnt += (uint32_t)__popc(rs_half_ballot(key[u] >= trial, half));  // (slots past C0 hold key 0 < trial)
      if (bit <= top && bit >= stop && cnt >= K) prefix = trial;
    }
    tau = __uint_as_float((prefix & 0x80000000u) ? (prefix & 0x7FFFFFFFu) : ~prefix);
    if (prune_rectify && !(tau > 0.f)) keep_all = true;  // (cosines rectified to 0 tie: a dropped candidate could still rank by its id)
  }
  uint32_t pos[NU], C = 0;
  bool keep[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const uint32_t c = (uint32_t)u * 32u + (uint32_t)hl;
    keep[u] = c < C0 && (keep_all || !((double)lo[u] + 2.0 * (double)dc[u] < (double)tau));
    const uint32_t m = rs_half_ballot(keep[u], half);
    pos[u] = C + (uint32_t)__popc(m & ((1u << hl) - 1u));
    C += (uint32_t)__popc(m);
  }
  if (C > max_ids) {  // more survivors than a record holds: the row goes on to the next stage
    *overflow = true;
    C = max_ids;
  }
  uint32_t jj[NU];
  double nj[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    keep[u] = keep[u] && pos[u] < C;
    const uint32_t pj = keep[u] ? e[u].x : 0u;
    jj[u] = perm ? perm[pj] : pj;
    if (perm && norms_pos) nj[u] = norms_pos[pj];  // (the norms by image position, so that the norm does not wait for the row id)
  }
  if (!(perm && norms_pos)) {
#pragma unroll
    for (int u = 0; u < NU; ++u) nj[u] = norms[keep[u] ? jj[u] : 0u];
  }
#pragma unroll
  for (int u = 0; u < NU; ++u)
    if (keep[u]) {
      h.rec[kRsIdsAt + pos[u]] = jj[u];
      reinterpret_cast<double *>(h.rec + norms_at)[pos[u]] = nj[u];
    }
  return C;
}

__global__ __launch_bounds__(256) void rerank_prune_halves_kernel(uint32_t row_begin, uint32_t row_count, const float *__restrict__ theta,
                                                                  const double *__restrict__ norms, const uint32_t *__restrict__ cnt_src,
                                                                  const uint32_t *__restrict__ cand_src, uint32_t cap, uint32_t K,
                                                                  const uint8_t *__restrict__ force_exact, int prune_rectify,
                                                                  uint32_t *__restrict__ records, int slots, const uint32_t *__restrict__ perm,
                                                                  const uint32_t *__restrict__ inv, unsigned long long *__restrict__ scored_total,
                                                                  const uint32_t *__restrict__ a_ids, const uint32_t *__restrict__ order,
                                                                  const float *__restrict__ err_pos, float accum_delta, float floor_delta,
                                                                  const double *__restrict__ norms_pos, const float *__restrict__ err_self,
                                                                  uint32_t *__restrict__ long_rows, uint32_t *__restrict__ long_count) {
  const int lane = threadIdx.x & 63, half = lane >> 5, hl = lane & 31;
  const uint32_t kRsMaxIds = (uint32_t)rs_max_ids(slots), kRsWords = (uint32_t)rs_words(slots), kRsNormsAt = (uint32_t)rs_norms_at(slots);
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = gridDim.x * (blockDim.x >> 6);
  unsigned long long scored = 0;
  for (uint32_t r2 = wave * 2; r2 < row_count; r2 += n_waves * 2) {
    const uint32_t r = r2 + (uint32_t)half;
    const bool live = r < row_count;
    const uint32_t rs = live ? r : r2;
    const uint32_t a = order ? order[rs] : (a_ids ? a_ids[rs] : row_begin + rs);
    const uint32_t lr = order ? (inv ? rs : a - row_begin) : (inv ? inv[a] : rs);
    const uint32_t total = cnt_src[lr];
    const float th_a = theta[a];
    const double nrm_a = norms[a];
    const bool force = force_exact && force_exact[a];
    const uint32_t take = min(total, cap);
    bool overflow = total > cap;
    const bool is_long = take > kRsHalfList;
    if (live && is_long && hl == 0) long_rows[atomicAdd(long_count, 1u)] = r;
    RsHalfRow h;
    h.src = reinterpret_cast<const uint2 *>(cand_src) + (size_t)lr * cap;
    h.rec = records + (size_t)rs * kRsWords;
    h.C0 = (live && !is_long) ? take : 0u;
    h.e_self = err_self ? err_self[a] : err_pos[lr];
    uint32_t C;
    if (__any(h.C0 > 64u))
      C = rs_prune_half_row<4>(h, half, hl, K, prune_rectify, err_pos, accum_delta, floor_delta, perm, norms_pos, norms, kRsMaxIds, kRsNormsAt, &overflow);
    else
      C = rs_prune_half_row<2>(h, half, hl, K, prune_rectify, err_pos, accum_delta, floor_delta, perm, norms_pos, norms, kRsMaxIds, kRsNormsAt, &overflow);
    if (hl == 0) scored += C;
    if (live && !is_long && hl == 0) {
      h.rec[0] = C | (overflow ? 0x100u : 0u) | (force ? 0x200u : 0u);
      h.rec[1] = __float_as_uint(th_a);
      const unsigned long long nb = (unsigned long long)__double_as_longlong(nrm_a);
      h.rec[2] = (uint32_t)nb;
      h.rec[3] = (uint32_t)(nb >> 32);
    }
  }
  for (int o = 32; o > 0; o >>= 1) scored += __shfl_xor(scored, o, 64);
  if (scored_total) block_add_u64(scored_total, kRsScoredSlots, scored);
}

typedef double rs_d2 __attribute__((ext_vector_type(2)));  // (a native vector: asm operands of struct type go through memory)
// row j of X -> one register per lane (lane l: bytes [16 l, 16 l + 16) of the row).  NOT valid before the set has landed (rerank_pairs.h).
__device__ __forceinline__ void rs_gload16(rs_d2 &dst, uint32_t lane_off, const double *row) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(lane_off), "s"(row) : "memory");
}

__device__ __forceinline__ double rs_readlane(double v, uint32_t l) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, (int)l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), (int)l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

}  // namespace asx
