// rerank_pairs.h -- exact re-rank of the joined candidate lists, every unordered pair scored ONCE (round 5).
// (included by mfma_drivers.hip behind rerank_split.h: the pruning kernel and the record layout are rerank_split.h's)
//
// rerank_split.h's scoring kernel fetched, per row, every survivor of the row's pruned list: a pair {i, j} that both rows keep
// was gathered and chained twice (the cosine is one number: x_i . x_j in the reference's order is the same bits from either
// side, and so is the product of the two norms), K = 17 cost two batches of 16 rows for most rows, and the sixteen chains of
// a batch ran behind scalar-cache round trips for ids, norms and headers.  Here:
//   1. rerank_link_kernel + rerank_fill_kernel   per record: for every survivor j of row i, is i a survivor of j (a search in j's record,
//                             L2-resident: neighbours sit together in the visiting order)?  A mutual pair belongs to one of
//                             its two rows by a parity rule both sides evaluate alike (so every row owns about half of its
//                             mutual pairs); the other row's slot will be DELIVERED.  Then the layout of a segment of 64 rows
//                             into batches of 16 slots (an X slot = the row itself, followed by its J slots = the partners
//                             it owns; a run that crosses a batch boundary repeats its X), and the slot stream itself: row id
//                             to load, and per J slot (norm product, own destination, partner's destination).
//   2. rerank_pairs_score_kernel  one autonomous wave per workgroup walks batches of the stream: 16 row loads + the batch's
//                             descriptors + the ids of the batch two steps ahead, ALL as counted vector loads into rotating
//                             register sets (no scalar-cache round trip in the loop); a landed set is multiplied by its X row
//                             IN REGISTERS by all 64 lanes (the products are the same bits whoever forms them) and the
//                             products go to the LDS tile; the 16 chain lanes then run 128 sequential additions each
//                             (src/core.rs:204-210: k ascending, multiply and add rounded separately) and the key is stored
//                             to the slot of BOTH rows (stores deferred to the next step, in front of its loads, so that the
//                             hand-counted waits stay exact).
//   3. rerank_rank_kernel     per record: (key, id) of its slots -> ranks, the K best, certification (what the tail of
//                             rerank_score_d128_kernel did), as a kernel of its own: the gather kernel keeps no ranking state.
// Outputs are bit-identical to rerank_score_d128_kernel's (same chains, same key formula, same ranking and certification).
#pragma once

namespace asx {

constexpr uint32_t kRpSelf = 0xFFFFFFFFu;       // meta of a slot: the row itself (never scored)
constexpr uint32_t kRpDelivered = 0xFFFFFFFEu;  // ... the partner owns the pair and writes this slot's key
constexpr uint32_t kRpSolo = 0xFFFFFFFDu;       // ... scored by this row for itself alone (the partner does not list it)
                                                // anything else: (partner's record << 8) | slot there, owned by this row
constexpr uint32_t kRpX = 0xFFFFFFFFu, kRpPad = 0xFFFFFFFEu;  // desc.z of a stream slot that is no pair
constexpr unsigned long long kRpUndelivered = 0x7FF8DEADBEEF0001ull;  // key of a slot nobody has written (a NaN no arithmetic produces)
constexpr int kRpSeg = 32;                      // records per layout segment (a workgroup of the link + fill kernel: 16 KB of LDS, 8 per CU)
constexpr uint32_t kRpShareMax = 1u << 24;      // records beyond this cannot be named in 24 bits: no sharing

// one row's run through the slot stream: X, its J slots, X again after every batch boundary; a batch's last slot never holds an X
struct RpLayout {
  uint32_t pos;
  __device__ __forceinline__ void skip_row(uint32_t A) {
    while (A > 0) {
      const uint32_t rem = 16u - (pos & 15u);
      if (rem < 2u) {
        pos += rem;
        continue;
      }
      const uint32_t take = min(A, rem - 1u);
      pos += 1u + take;
      A -= take;
    }
  }
};

// counters of a call (device): [0] batches of the slot stream (allocated by the segments' atomics), [1] pairs scored, [2] set when the
// stream ran out of room (nothing of it is then valid: the scoring kernel does not run and the host repeats the call with [0] batches)
constexpr int kRpPairSlots = 64;  // words the count of scored pairs is spread over (one word: ~90 adds per microsecond, common.h)
constexpr int kRpPairsAt = 16;
constexpr int kRpCounters = kRpPairsAt + kRpPairSlots;  // ([4 .. 8]: why rows were left open, rerank_rank_kernel; [16 .. 80): pairs scored, by block)

// Link + layout (rerank_link_kernel), then -- behind an exclusive scan of the segments' batch counts, so that the stream is in the
// order of the records whatever the order the workgroups run in: the scoring kernel's waves work side by side on one L2 -- the
// slots themselves (rerank_fill_kernel).  One segment of 32 records per workgroup of 4 waves; each HALF of a wave takes a record at
// a time (lane hl of the half = slot hl of the record; longer records loop), so a wave walks its 8 records in 4 steps, and the loads
// of a step -- header and ids, then the partners' places, then the partners' id blocks -- are three dependent levels issued for all
// lanes at once.
// The slot stream: ids[slot] = row to load, desc[slot] = X: (record, -, kRpX, -); J: (norm product lo, hi, slot in the record,
// partner's (record << 8 | slot) or kNoIdx); padding: (-, -, kRpPad, -) with a loadable id.
__device__ __forceinline__ uint32_t rp_half_mask(unsigned long long bal, int half) { return (uint32_t)(half ? (bal >> 32) : bal); }

template <int SLOTS>
__global__ __launch_bounds__(256, 8) void rerank_link_kernel(uint32_t row_begin, uint32_t row_count, uint32_t *__restrict__ records,
                                                          const uint32_t *__restrict__ a_ids, const uint32_t *__restrict__ order,
                                                          const uint32_t *__restrict__ row2rec, int share, uint32_t *__restrict__ meta,
                                                          uint32_t *__restrict__ rowpos, uint32_t *__restrict__ seg_batches,
                                                          unsigned long long *__restrict__ counters) {
  constexpr uint32_t W = (uint32_t)rs_words(SLOTS), S = 64u * SLOTS, NA = (uint32_t)rs_norms_at(SLOTS);
  constexpr uint32_t kRowsPerWave = kRpSeg / 4;
  __shared__ uint32_t s_act[kRpSeg];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, hl = lane & 31;
  const uint32_t n_seg = (row_count + kRpSeg - 1) / kRpSeg;
  unsigned long long total = 0;
  // Which segments this workgroup takes: XCD x = blockIdx.x % 8 walks the x-th eighth of the records with its workgroups side by
  // side (as the scoring kernel does with the stream), so that the workgroups that run at the same time on one L2 look up partners
  // among the same few thousand records -- the records are in the order of the image, a row's partners its neighbours there.
  // With the segments dealt round robin over the whole grid every lookup missed the L2 (1 M x 768: 106 M missed lines for 52 M
  // lookups of two lines, 8.0 ms).
  uint32_t seg_first = blockIdx.x, seg_last = n_seg, seg_stride = gridDim.x;
  if (gridDim.x % 8u == 0u) {
    const uint32_t x = blockIdx.x & 7u, chunk = (n_seg + 7u) / 8u;
    seg_stride = gridDim.x / 8u;
    seg_first = x * chunk + (blockIdx.x >> 3);
    seg_last = min((x + 1u) * chunk, n_seg);
  }
  for (uint32_t seg = seg_first; seg < seg_last; seg += seg_stride) {
#pragma unroll 1
    for (uint32_t i = 0; i < kRowsPerWave / 2; ++i) {
      const uint32_t l = (uint32_t)wv * kRowsPerWave + 2u * i + (uint32_t)half, r = seg * kRpSeg + l;
      const bool live = r < row_count;
      uint32_t *rec = records + (size_t)(live ? r : 0u) * W;
      const uint32_t C = live ? rec[0] & 0xFFu : 0u;
      const uint32_t ro = live ? (order ? order[r] : r) : 0u;
      const uint32_t a = a_ids ? a_ids[ro] : row_begin + ro;
      const uint32_t Cmax = max((uint32_t)__builtin_amdgcn_readlane((int)C, 0), (uint32_t)__builtin_amdgcn_readlane((int)C, 32));
      uint32_t A = 0;
      for (uint32_t base = 0; base < Cmax; base += 32) {
        const uint32_t c = base + (uint32_t)hl;
        uint32_t m = kRpSolo;
        if (c < C) {
          const uint32_t j = rec[kRsIdsAt + c];
          if (j == a) {
            m = kRpSelf;
          } else if (share) {
            const uint32_t q = row2rec[j];
            if (q < row_count && q != r) {
              // the partner's header and its first 28 ids in one go (a record's ids start 16 bytes in); longer lists go on from there
              // (records of wide rows -- SLOTS 4, 52 survivors a row at 768 columns -- : header + 60 ids at once: the loads that went
              // on from 28 were a chain of six dependent round trips per lookup)
              constexpr int kPre = SLOTS >= 4 ? 16 : 8;
              const uint4 *bq = reinterpret_cast<const uint4 *>(records + (size_t)q * W);
              uint4 v[kPre];
#pragma unroll
              for (int u = 0; u < kPre; ++u) v[u] = bq[u];
              const uint32_t Cq = v[0].x & 0xFFu;
              uint32_t t = kNoIdx;
#pragma unroll
              for (int u = 1; u < kPre; ++u) {
                const uint32_t t0 = (uint32_t)(u - 1) * 4u;
                if (t0 < Cq && v[u].x == a) t = t0;
                if (t0 + 1 < Cq && v[u].y == a) t = t0 + 1;
                if (t0 + 2 < Cq && v[u].z == a) t = t0 + 2;
                if (t0 + 3 < Cq && v[u].w == a) t = t0 + 3;
              }
              for (uint32_t t0 = (uint32_t)(kPre - 1) * 4u; t0 < Cq; t0 += 4) {
                const uint4 x = bq[1 + t0 / 4];
                if (x.x == a) t = t0;
                if (t0 + 1 < Cq && x.y == a) t = t0 + 1;
                if (t0 + 2 < Cq && x.z == a) t = t0 + 2;
                if (t0 + 3 < Cq && x.w == a) t = t0 + 3;
              }
              if (t != kNoIdx) {
                const uint32_t owner = ((r + q) & 1u) ? max(r, q) : min(r, q);
                m = owner == r ? ((q << 8) | t) : kRpDelivered;
              }
            }
          }
          meta[(size_t)r * S + c] = m;
          if (m == kRpDelivered) reinterpret_cast<unsigned long long *>(rec + NA)[c] = kRpUndelivered;
        }
        const bool active = c < C && m != kRpSelf && m != kRpDelivered;
        A += (uint32_t)__builtin_popcount(rp_half_mask(__ballot(active), half));
      }
      if (hl == 0) s_act[l] = A;
      if (hl == 0) total += A;
    }
    __syncthreads();
    if (wv == 0) {  // the layout of the segment
      const uint32_t mine = lane < kRpSeg ? s_act[lane] : 0u;
      RpLayout lay{0u};
      uint32_t start = 0;
      for (int l = 0; l < kRpSeg; ++l) {
        const uint32_t A = (uint32_t)__builtin_amdgcn_readlane((int)mine, l);
        if (lane == l) start = lay.pos;
        lay.skip_row(A);
      }
      const uint32_t r = seg * kRpSeg + (uint32_t)lane;
      if (lane < kRpSeg && r < row_count) rowpos[r] = start | (mine ? 0u : 0x80000000u);  // (top bit: the row scores nothing)
      if (lane == 0) seg_batches[seg] = (lay.pos + 15u) / 16u;
    }
    __syncthreads();
  }
  for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, 64);
  block_add_u64(counters + kRpPairsAt, (unsigned int)kRpPairSlots, total);
}

// seg_off: exclusive scan of seg_batches (seg_off[n_seg] = the stream's length, which the scoring kernel reads from there)
template <int SLOTS>
__global__ __launch_bounds__(256) void rerank_fill_kernel(uint32_t row_begin, uint32_t row_count, const uint32_t *__restrict__ records,
                                                          const uint32_t *__restrict__ a_ids, const uint32_t *__restrict__ order,
                                                          const uint32_t *__restrict__ meta, const uint32_t *__restrict__ rowpos,
                                                          const uint32_t *__restrict__ seg_batches, const uint64_t *__restrict__ seg_off,
                                                          uint32_t *__restrict__ s_ids, uint4 *__restrict__ s_desc, uint32_t cap_batches,
                                                          unsigned long long *__restrict__ counters) {
  constexpr uint32_t W = (uint32_t)rs_words(SLOTS), S = 64u * SLOTS, NA = (uint32_t)rs_norms_at(SLOTS);
  constexpr uint32_t kRowsPerWave = kRpSeg / 4;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, hl = lane & 31;
  const uint32_t n_seg = (row_count + kRpSeg - 1) / kRpSeg;
  if (blockIdx.x == 0 && threadIdx.x == 0) counters[0] = seg_off[n_seg];
  if (seg_off[n_seg] > (uint64_t)cap_batches) {  // no room for the stream: the host repeats the call with seg_off[n_seg] batches
    if (blockIdx.x == 0 && threadIdx.x == 0) counters[2] = 1ull;
    return;
  }
  for (uint32_t seg = blockIdx.x; seg < n_seg; seg += gridDim.x) {
    const uint64_t base_slot = seg_off[seg] * 16u;
    uint32_t last_end = 0;  // where the segment's last scoring row stops (its tail is padded from there)
#pragma unroll 1
    for (uint32_t i = 0; i < kRowsPerWave / 2; ++i) {
      const uint32_t l = (uint32_t)wv * kRowsPerWave + 2u * i + (uint32_t)half, r = seg * kRpSeg + l;
      const bool live = r < row_count;
      const uint32_t rp = live ? rowpos[r] : 0x80000000u;
      const bool scores = (rp & 0x80000000u) == 0u;
      const uint32_t start = rp & 0x7FFFFFFFu;
      const uint32_t *rec = records + (size_t)(live ? r : 0u) * W;
      const uint32_t C = scores ? rec[0] & 0xFFu : 0u;
      const uint32_t ro = live ? (order ? order[r] : r) : 0u;
      const uint32_t a = a_ids ? a_ids[ro] : row_begin + ro;
      const double na = __longlong_as_double((long long)(((unsigned long long)rec[3] << 32) | rec[2]));
      const uint32_t Cmax = max((uint32_t)__builtin_amdgcn_readlane((int)C, 0), (uint32_t)__builtin_amdgcn_readlane((int)C, 32));
      uint32_t before = 0;  // active slots of the record in front of this pass of 32
      for (uint32_t base = 0; base < Cmax; base += 32) {
        const uint32_t c = base + (uint32_t)hl;
        const uint32_t m = c < C ? meta[(size_t)r * S + c] : kRpSelf;
        const bool active = m != kRpSelf && m != kRpDelivered;
        const uint32_t hm = rp_half_mask(__ballot(active), half);
        if (active) {
          // item e of the row -> its slot: RpLayout::skip_row walked by the item itself (a row's run crosses few batch boundaries)
          uint32_t e = before + (uint32_t)__builtin_popcount(hm & ((1u << hl) - 1u)), pos = start;
          for (;;) {
            const uint32_t rem = 16u - (pos & 15u);
            if (rem < 2u) {
              pos += rem;
              continue;
            }
            if (e < rem - 1u) break;
            pos += rem;
            e -= rem - 1u;
          }
          const uint64_t slot = base_slot + pos + 1u + e;
          const double nj = reinterpret_cast<const double *>(rec + NA)[c];
          const unsigned long long db = (unsigned long long)__double_as_longlong(__dmul_rn(na, nj));  // GraphDistPolicy::key's denominator
          s_ids[slot] = rec[kRsIdsAt + c];
          s_desc[slot] = make_uint4((uint32_t)db, (uint32_t)(db >> 32), c, m == kRpSolo ? kNoIdx : m);
        }
        before += (uint32_t)__builtin_popcount(hm);
      }
      if (hl == 0 && before) {  // the row's X slots and the single slots left in front of a batch boundary
        uint32_t pos = start, left = before;
        while (left > 0) {
          const uint32_t rem = 16u - (pos & 15u);
          if (rem < 2u) {
            s_ids[base_slot + pos] = a;
            s_desc[base_slot + pos] = make_uint4(0u, 0u, kRpPad, kNoIdx);
            pos += rem;
            continue;
          }
          const uint32_t take = min(left, rem - 1u);
          s_ids[base_slot + pos] = a;
          s_desc[base_slot + pos] = make_uint4(r, 0u, kRpX, kNoIdx);
          pos += 1u + take;
          left -= take;
        }
        last_end = pos;
      }
    }
    // the tail of the segment's last batch: behind the end of the last row that scores anything (rows are laid out in order)
    last_end = max(last_end, __shfl_xor(last_end, 32, 64));
    __shared__ uint32_t s_end[4];
    if (lane == 0) s_end[wv] = last_end;
    __syncthreads();
    const uint32_t used = max(max(s_end[0], s_end[1]), max(s_end[2], s_end[3])), end = seg_batches[seg] * 16u;
    if (threadIdx.x < 16u && used + threadIdx.x < end) {
      const uint32_t first_ro = order ? order[seg * kRpSeg] : seg * kRpSeg;
      s_ids[base_slot + used + threadIdx.x] = a_ids ? a_ids[first_ro] : row_begin + first_ro;  // a row that exists
      s_desc[base_slot + used + threadIdx.x] = make_uint4(0u, 0u, kRpPad, kNoIdx);
    }
    __syncthreads();
  }
}

typedef uint32_t rp_u4 __attribute__((ext_vector_type(4)));  // (a native vector: asm operands of struct type go through memory)
struct RpSet {
  rs_d2 v[kRsRows];  // this lane's 16 B of the piece of each of the batch's 16 rows
  rp_u4 desc;          // lane l: descriptor of slot l & 15 of the batch
  uint32_t ids2;       // lane l: row id of slot l & 15 of the batch TWO steps ahead (what this register set is loaded with next)
};
constexpr int kRpLoads = kRsRows + 2;
__device__ __forceinline__ void rp_gload_desc(rp_u4 &dst, const rp_u4 *p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void rp_gload_u32(uint32_t &dst, const uint32_t *p) {
  asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void rp_gstore_f64(double *p, double v) {
  asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
template <int N>
__device__ __forceinline__ void rp_landed(RpSet &s) {
  static_assert(kRsRows == 16, "operand lists below");
  asm volatile("s_waitcnt vmcnt(%9)"
               : "+v"(s.v[0]), "+v"(s.v[1]), "+v"(s.v[2]), "+v"(s.v[3]), "+v"(s.v[4]), "+v"(s.v[5]), "+v"(s.v[6]), "+v"(s.v[7]), "+v"(s.desc)
               : "n"(N));
  asm volatile("" : "+v"(s.v[8]), "+v"(s.v[9]), "+v"(s.v[10]), "+v"(s.v[11]), "+v"(s.v[12]), "+v"(s.v[13]), "+v"(s.v[14]), "+v"(s.v[15]), "+v"(s.ids2));
}

// Rows of d = 128 x PIECES values: a step stages one 128-wide piece of a batch's 16 rows; the chains of a batch run on across
// its pieces.  Steps of a wave: (its batch m, piece q), m = 0, 1, .. over the batches first + m * stride < last.
template <uint32_t PIECES, int SLOTS>
__global__ __launch_bounds__(64) void rerank_pairs_score_kernel(const double *__restrict__ X, const uint32_t *__restrict__ s_ids,
                                                                const rp_u4 *__restrict__ s_desc, const unsigned long long *__restrict__ counters,
                                                                int rectify, uint32_t *__restrict__ records, uint32_t dbg) {
  constexpr uint32_t d = 128u * PIECES;
  constexpr uint32_t W = (uint32_t)rs_words(SLOTS), NA = (uint32_t)rs_norms_at(SLOTS);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *tile = smem;  // the products of the batch whose chains run: 16 rows of 128
  const int lane = threadIdx.x;
  const uint32_t lane_off = (uint32_t)lane * 16u;
  const uint32_t chain_len = (dbg & 1) ? 16u : 128u;  // dbg: timing experiments only (outputs undefined)

  // Which batches this wave takes: XCD x = blockIdx.x % 8 walks the x-th eighth of the stream with its waves side by side, so
  // that the waves that run at the same time on one L2 work on neighbouring rows, whose partners are largely the same rows.
  if (counters[2] != 0ull) return;  // the stream ran out of room: the host repeats the call
  const uint32_t n_batches = (uint32_t)counters[0];
  uint32_t stride = gridDim.x, first = blockIdx.x, last = n_batches;
  if (gridDim.x % 8u == 0u) {
    const uint32_t x = blockIdx.x & 7u, chunk = (n_batches + 7u) / 8u;
    stride = gridDim.x / 8u;
    first = x * chunk + (blockIdx.x >> 3);
    last = min((x + 1u) * chunk, n_batches);
  }
  if (first >= last) return;
  const uint32_t n_mine = (last - first + stride - 1u) / stride, n_steps = n_mine * PIECES;
  // batch and piece of step u (past the end: the wave's first batch again -- loads of rows that exist, never staged)
  auto batch_of = [&](uint32_t u) __attribute__((always_inline)) -> uint32_t { return u < n_steps ? first + (u / PIECES) * stride : first; };

  RpSet sA, sB;
  // the ids of a set's rows sit in the OTHER lanes of a register: lane t -> SGPR, one load each
#define AS_GL(S, T) rs_gload16(S.v[T], lane_off, X + (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)ids_, T) * d + q_ * 128u);
#define AS_ISSUE(S, U)                                                                                            \
  {                                                                                                               \
    const uint32_t u_ = (U), k_ = batch_of(u_), q_ = u_ < n_steps ? u_ % PIECES : 0u;                            \
    AS_GL(S, 0) AS_GL(S, 1) AS_GL(S, 2) AS_GL(S, 3) AS_GL(S, 4) AS_GL(S, 5) AS_GL(S, 6) AS_GL(S, 7)               \
    AS_GL(S, 8) AS_GL(S, 9) AS_GL(S, 10) AS_GL(S, 11) AS_GL(S, 12) AS_GL(S, 13) AS_GL(S, 14) AS_GL(S, 15)         \
    (void)k_;                                                                                                     \
    rp_gload_u32(S.ids2, s_ids + (size_t)batch_of(u_ + 2u) * 16u + (lane & 15));                                  \
  }
  // (the descriptors of a set are loaded at the END of the step that reloads its rows: the landed ones are read until then, and a
  // register that is live while its successor is in flight makes the compiler copy it -- in front of the wait, i.e. before it landed)
#define AS_ISSUE_DESC(S, U) rp_gload_desc(S.desc, s_desc + (size_t)batch_of(U) * 16u + (lane & 15));
  {
    uint32_t ids_ = s_ids[(size_t)batch_of(0) * 16u + (lane & 15)];
    AS_ISSUE(sA, 0u)
    AS_ISSUE_DESC(sA, 0u)
    ids_ = s_ids[(size_t)batch_of(1) * 16u + (lane & 15)];
    AS_ISSUE(sB, 1u)
    AS_ISSUE_DESC(sB, 1u)
  }

  // the key of the step before, stored in front of this step's loads
  double pend_key = 0.0, *pend_own = nullptr, *pend_mut = nullptr;
  auto flush = [&]() __attribute__((always_inline)) {
    if (pend_own) rp_gstore_f64(pend_own, pend_key);
    if (pend_mut) rp_gstore_f64(pend_mut, pend_key);
    pend_own = nullptr;
    pend_mut = nullptr;
  };
  double acc = 0.0;  // this lane's chain, carried across the pieces of a batch

#define AS_PR(S, T)                                                                             \
  {                                                                                             \
    if ((xmask >> T) & 1u) xcur = S.v[T];                                                       \
    rs_d2 pr;                                                                                   \
    pr.x = __dmul_rn(S.v[T].x, xcur.x);                                                         \
    pr.y = __dmul_rn(S.v[T].y, xcur.y);                                                         \
    *reinterpret_cast<rs_d2 *>(tile + T * kRsPitch + lane_off) = pr;                            \
  }
#define AS_STEP(P, U)                                                                                                       \
  {                                                                                                                         \
    const uint32_t us_ = (U);                                                                                                \
    rp_landed<kRpLoads>(P);                                                                                                 \
    const rp_u4 dsc = P.desc;                                                                                               \
    const uint32_t ids_ = P.ids2;                                                                                           \
    const uint32_t xmask = (uint32_t)__ballot(lane < kRsRows && dsc.z == kRpX);                                             \
    {                                                                                                                       \
      rs_d2 xcur = P.v[0];                                                                                                  \
      AS_PR(P, 0) AS_PR(P, 1) AS_PR(P, 2) AS_PR(P, 3) AS_PR(P, 4) AS_PR(P, 5) AS_PR(P, 6) AS_PR(P, 7)                       \
      AS_PR(P, 8) AS_PR(P, 9) AS_PR(P, 10) AS_PR(P, 11) AS_PR(P, 12) AS_PR(P, 13) AS_PR(P, 14) AS_PR(P, 15)                 \
    }                                                                                                                       \
    flush();                                                                                                                \
    AS_ISSUE(P, us_ + 2u)                                                                                                     \
    process(us_, dsc, xmask);                                                                                                 \
    AS_ISSUE_DESC(P, us_ + 2u)                                                                                                \
    if (us_ + 1u == n_steps) break;                                                                                           \
  }

  auto process = [&](uint32_t u, const rp_u4 dsc, uint32_t xmask) __attribute__((always_inline)) {
    const uint32_t q = u % PIECES;
    wave_sync();
    const bool is_pair = lane < kRsRows && dsc.z < kRpPad;
    if (!(dbg & 4) && is_pair) {
      // the reference's chain (src/core.rs:204-210): k ascending, multiply and add rounded separately (the products were
      // rounded when they were formed)
      const rs_d2 *pj = reinterpret_cast<const rs_d2 *>(tile + (size_t)lane * kRsPitch);
      if (q == 0) acc = 0.0;
      rs_d2 pa[4], pb[4];
#pragma unroll
      for (int h = 0; h < 4; ++h) pa[h] = pj[h];
#pragma unroll 1
      for (uint32_t g = 0; g < chain_len / 16; ++g) {
#pragma unroll
        for (int h = 0; h < 4; ++h) pb[h] = pj[g * 8 + 4 + h];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          acc = __dadd_rn(acc, pa[h].x);
          acc = __dadd_rn(acc, pa[h].y);
        }
        if (g + 1 < chain_len / 16) {
#pragma unroll
          for (int h = 0; h < 4; ++h) pa[h] = pj[g * 8 + 8 + h];
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          acc = __dadd_rn(acc, pb[h].x);
          acc = __dadd_rn(acc, pb[h].y);
        }
      }
    }
    if (q + 1 == PIECES) {
      // the record of a J slot's row: the X slot in front of it
      const uint32_t below = xmask & ((2u << (lane & 15)) - 1u);
      const int xl = 31 - __builtin_clz(below | 1u);
      const uint32_t r = (uint32_t)__builtin_amdgcn_ds_bpermute(xl * 4, (int)dsc.x);
      if (is_pair) {  // GraphDistPolicy::key with the operands at hand
        const double denom = __longlong_as_double((long long)(((unsigned long long)dsc.y << 32) | dsc.x));
        double cs = denom > 1e-12 ? __ddiv_rn(acc, denom) : 0.0;
        if (rectify) {
          cs = fmin(fmax(cs, -1.0), 1.0);
          cs = fmax(cs, 0.0);
        }
        pend_key = __dsub_rn(1.0, cs);
        pend_own = reinterpret_cast<double *>(records + (size_t)r * W + NA) + dsc.z;
        pend_mut = dsc.w == kNoIdx ? nullptr : reinterpret_cast<double *>(records + (size_t)(dsc.w >> 8) * W + NA) + (dsc.w & 0xFFu);
      }
    }
    wave_sync();
  };

  for (uint32_t u = 0;; u += 2) {
    AS_STEP(sA, u)
    AS_STEP(sB, u + 1u)
  }
#undef AS_STEP
#undef AS_PR
#undef AS_ISSUE
#undef AS_ISSUE_DESC
#undef AS_GL
  flush();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the batches issued past the end, the last keys
}

// (key, id) of a record's slots -> its K best and the certification of the row (the tail of rerank_score_d128_kernel).
// A wave takes two records: when both have at most 32 slots (nearly all) each half of the wave ranks one of them -- a candidate's
// (key, id) reaches the lanes of its half through ds_bpermute, half the instructions per row --; otherwise one after the other.
template <int SLOTS>
__device__ __forceinline__ void rp_rank_full(uint32_t p, uint32_t row_begin, const uint32_t *__restrict__ records, uint32_t K, const GraphBound &bnd,
                                             double *__restrict__ out_key, uint32_t *__restrict__ out_idx, uint32_t *__restrict__ fail_rows,
                                             uint32_t *__restrict__ fail_count, const uint32_t *__restrict__ a_ids, const uint32_t *__restrict__ order,
                                             int lane, unsigned long long *__restrict__ why) {
  constexpr uint32_t W = (uint32_t)rs_words(SLOTS), NA = (uint32_t)rs_norms_at(SLOTS);
  const uint32_t *rec = records + (size_t)p * W;
  const uint4 h = *reinterpret_cast<const uint4 *>(rec);
  const uint32_t C = h.x & 0xFFu;
  const bool p_overflow = (h.x & 0x100u) != 0, p_force = (h.x & 0x200u) != 0;
  const float p_theta = __uint_as_float(h.y);
  const uint32_t r = order ? order[p] : p;
  const uint32_t a = a_ids ? a_ids[r] : row_begin + r;
  double kv[SLOTS];
  uint32_t iv[SLOTS], rank[SLOTS];
  bool undelivered = false;
#pragma unroll
  for (int q = 0; q < SLOTS; ++q) {
    kv[q] = __builtin_inf();
    iv[q] = kNoIdx;
    rank[q] = 0;
    const uint32_t c = 64u * q + (uint32_t)lane;
    if (c < C) {
      const uint32_t id = rec[kRsIdsAt + c];
      const double key = reinterpret_cast<const double *>(rec + NA)[c];
      undelivered = undelivered || (id != a && (unsigned long long)__double_as_longlong(key) == kRpUndelivered);
      if (id != a && !(key != key)) {
        kv[q] = key;
        iv[q] = id;
      }
    }
  }
#pragma unroll
  for (int q2 = 0; q2 < SLOTS; ++q2) {
    const uint32_t c_hi = C < 64u * (q2 + 1) ? C : 64u * (q2 + 1);
    for (uint32_t c2 = 64u * q2; c2 < c_hi; ++c2) {  // (no iterations when C <= 64 q2)
      const double kc = rs_readlane(kv[q2], c2 - 64u * q2);
      const uint32_t ic = (uint32_t)__builtin_amdgcn_readlane((int)iv[q2], (int)(c2 - 64u * q2));
#pragma unroll
      for (int q = 0; q < SLOTS; ++q) rank[q] += lex_less(kc, ic, kv[q], iv[q]) ? 1u : 0u;
    }
  }
  double kth = -__builtin_inf();
  uint32_t n_valid = 0;
#pragma unroll
  for (int q = 0; q < SLOTS; ++q)
    if (iv[q] != kNoIdx) {
      ++n_valid;
      if (rank[q] < K) {
        out_key[(uint64_t)r * K + rank[q]] = kv[q];
        out_idx[(uint64_t)r * K + rank[q]] = iv[q];
        kth = fmax(kth, kv[q]);
      }
    }
  for (int o = 32; o > 0; o >>= 1) {
    kth = fmax(kth, __shfl_xor(kth, o, 64));
    n_valid += __shfl_xor(n_valid, o, 64);
  }
  for (uint32_t s = n_valid + lane; s < K; s += 64) {
    out_key[(uint64_t)r * K + s] = __builtin_inf();
    out_idx[(uint64_t)r * K + s] = kNoIdx;
  }
  // every pair that is not in the lists has an approximate cosine below theta; a slot nobody wrote (never expected: the
  // link rule is symmetric) sends the row to the next stage like any other uncertified row
  const bool und = __ballot(undelivered) != 0ull;
  const bool ok = !p_overflow && !p_force && !und && n_valid >= K && kth < bnd.lower_bound(-(double)p_theta);
  if (!ok && lane == 0) {
    fail_rows[atomicAdd(fail_count, 1u)] = r;
    // why (counters 4 ..: list ran over, forced, a slot nobody wrote, fewer than K candidates, K-th not clear of the threshold)
    atomicAdd(&why[p_overflow ? 4 : (p_force ? 5 : (und ? 6 : (n_valid < K ? 7 : 8)))], 1ull);
  }
}

template <int SLOTS>
__global__ __launch_bounds__(256) void rerank_rank_kernel(uint32_t row_begin, uint32_t row_count, const uint32_t *__restrict__ records, uint32_t K,
                                                          GraphBound bnd, double *__restrict__ out_key, uint32_t *__restrict__ out_idx,
                                                          uint32_t *__restrict__ fail_rows, uint32_t *__restrict__ fail_count,
                                                          const uint32_t *__restrict__ a_ids, const uint32_t *__restrict__ order,
                                                          unsigned long long *__restrict__ why) {
  constexpr uint32_t W = (uint32_t)rs_words(SLOTS), NA = (uint32_t)rs_norms_at(SLOTS);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int half = lane >> 5, hl = lane & 31;
  for (uint32_t p0 = (blockIdx.x * 4 + wv) * 2u; p0 < row_count; p0 += gridDim.x * 8u) {
    const uint32_t C0 = records[(size_t)p0 * W] & 0xFFu;
    const uint32_t C1 = p0 + 1 < row_count ? (records[(size_t)(p0 + 1) * W] & 0xFFu) : 0u;
    if (C0 > 32u || C1 > 32u || p0 + 1 >= row_count) {
      rp_rank_full<SLOTS>(p0, row_begin, records, K, bnd, out_key, out_idx, fail_rows, fail_count, a_ids, order, lane, why);
      if (p0 + 1 < row_count) rp_rank_full<SLOTS>(p0 + 1, row_begin, records, K, bnd, out_key, out_idx, fail_rows, fail_count, a_ids, order, lane, why);
      continue;
    }
    // each half of the wave one record: lane hl of the half holds slot hl.  Keys compare as integers -- the order-preserving image of a
    // double: sign bit set for non-negative values, all bits flipped for negative ones (a cosine a rounding above 1 gives a distance just
    // below 0 where nothing is rectified); -0 and NaN do not occur: 1 - c is +0 at c = 1, NaN keys were replaced by +inf.
    const uint32_t p = p0 + (uint32_t)half;
    const uint32_t *rec = records + (size_t)p * W;
    const uint4 h = *reinterpret_cast<const uint4 *>(rec);
    // (slot hl < 32 exists in every record: loaded with the header, not behind its count)
    const uint32_t id_l = rec[kRsIdsAt + hl];
    const double key_l = reinterpret_cast<const double *>(rec + NA)[hl];
    const uint32_t C = h.x & 0xFFu;
    const bool p_overflow = (h.x & 0x100u) != 0, p_force = (h.x & 0x200u) != 0;
    const float p_theta = __uint_as_float(h.y);
    const uint32_t r = order ? order[p] : p;
    const uint32_t a = a_ids ? a_ids[r] : row_begin + r;
    double kv = __builtin_inf();
    uint32_t iv = kNoIdx, rank = 0;
    bool undelivered = false;
    if ((uint32_t)hl < C) {
      const uint32_t id = id_l;
      const double key = key_l;
      undelivered = id != a && (unsigned long long)__double_as_longlong(key) == kRpUndelivered;
      if (id != a && !(key != key)) {
        kv = key;
        iv = id;
      }
    }
    const unsigned long long kraw = (unsigned long long)__double_as_longlong(kv);
    const unsigned long long kb = (kraw >> 63) ? ~kraw : (kraw | 0x8000000000000000ull);
    const uint32_t Cm = max(C0, C1);
    for (uint32_t c2 = 0; c2 < Cm; ++c2) {
      const int src = (int)((c2 + 32u * (uint32_t)half) * 4u);
      const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)kb);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)(kb >> 32));
      const uint32_t ic = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)iv);
      const unsigned long long kc = ((unsigned long long)hi << 32) | lo;
      // (slots past a record's count hold (+inf, kNoIdx): they rank behind everything and count for nobody in front of a valid slot)
      rank += (kc < kb || (kc == kb && ic < iv)) ? 1u : 0u;
    }
    double kth = -__builtin_inf();
    uint32_t n_valid = 0;
    if (iv != kNoIdx) {
      n_valid = 1;
      if (rank < K) {
        out_key[(uint64_t)r * K + rank] = kv;
        out_idx[(uint64_t)r * K + rank] = iv;
        kth = kv;
      }
    }
    for (int o = 16; o > 0; o >>= 1) {  // within the half
      kth = fmax(kth, __shfl_xor(kth, o, 64));
      n_valid += __shfl_xor(n_valid, o, 64);
    }
    for (uint32_t s = n_valid + hl; s < K; s += 32) {
      out_key[(uint64_t)r * K + s] = __builtin_inf();
      out_idx[(uint64_t)r * K + s] = kNoIdx;
    }
    const unsigned long long und = __ballot(undelivered);
    const bool und_half = ((half ? (und >> 32) : und) & 0xFFFFFFFFull) != 0ull;
    const bool ok = !p_overflow && !p_force && !und_half && n_valid >= K && kth < bnd.lower_bound(-(double)p_theta);
    if (!ok && hl == 0) {
      fail_rows[atomicAdd(fail_count, 1u)] = r;
      atomicAdd(&why[p_overflow ? 4 : (p_force ? 5 : (und_half ? 6 : (n_valid < K ? 7 : 8)))], 1ull);
    }
  }
}

__global__ void rerank_row2rec_kernel(uint32_t row_begin, uint32_t row_count, const uint32_t *__restrict__ a_ids, const uint32_t *__restrict__ order,
                                      uint32_t *__restrict__ row2rec) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= row_count) return;
  const uint32_t r = order ? order[p] : p;
  row2rec[a_ids ? a_ids[r] : row_begin + r] = p;
}

}  // namespace asx
