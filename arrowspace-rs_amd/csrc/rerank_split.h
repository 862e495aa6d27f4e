// rerank_split.h -- exact re-rank of the joined candidate lists of the item graph (128-wide rows) as two kernels.
// (included by mfma_drivers.hip behind prune_candidates / GraphBound)
//
// The one-kernel form (rerank_join_d128_kernel) serialises, per row and wave, list -> pruning -> four batches of
// eight row fetches with their arithmetic in between: 21 ms at 1 M rows for 31 GB, 1.5 TB/s, latency-bound.  Here
//   1. rerank_prune_kernel streams the lists (cheap LDS footprint: 32 waves per CU), prunes them by their stored
//      scores and leaves one 512-byte record per row: count, flags, theta, the row's norm, up to 124 survivor ids;
//   2. rerank_score_d128_kernel (one autonomous wave per workgroup) turns the rows of a wave into ONE stream of
//      batches of eight candidate rows and keeps TWO batches in flight behind the one whose sequential f64 chains run.
//      A 1 KB row gather takes ~6.5 us to land under load, so what sets the rate is the bytes in flight per CU; LDS
//      (160 KB) holds 56 KB of them as LDS-DMA targets next to the rest of a wave's state (measured: 16 ms, no better
//      than the one-kernel form), the VGPR file (512 KB) holds 192 KB: the rows are fetched INTO REGISTERS (one
//      global_load_dwordx4 per row: the wave's 64 lanes take 16 B each), three register sets rotate, and a set is
//      written to a single LDS tile only once it has landed.  The loads are asm so that their waits can be counted
//      by hand (vmcnt retires in order: "at most 16 younger" = the set is there); everything data-dependent (counts,
//      ids, theta, the norms of the row and of its candidates, gathered by the pruning kernel) comes through the
//      scalar cache, which has its own counter -- a compiler-generated vector load in the loop would be waited for with
//      vmcnt(0) and drain the pipeline.
#pragma once

namespace asx {

// A record of SLOTS x 64 id slots (SLOTS = 2 for 128-wide rows, 4 for wider ones, whose lists prune less): u32 words
constexpr int kRsIdsAt = 4;            // word 0: count | overflow << 8 | force << 9; 1: theta; 2-3: norm of the row
__host__ __device__ constexpr int rs_max_ids(int slots) { return 64 * slots - kRsIdsAt; }  // survivor ids behind the header
__host__ __device__ constexpr int rs_norms_at(int slots) { return 64 * slots; }            // their norms (f64): what GraphDistPolicy::key reads per pair
__host__ __device__ constexpr int rs_words(int slots) { return 192 * slots; }              // 1.5 KB / 3 KB per row
constexpr int kRsRows = 16;            // candidate rows per batch
constexpr int kRsPitch = 1040;         // LDS pitch of a staged row (bytes)
__host__ __device__ constexpr size_t rs_lds(int slots) { return 1024 + kRsRows * kRsPitch + (size_t)64 * slots * 12; }

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
                                                           const double *__restrict__ norms_pos) {
  // records are written by PLACE r of the visiting order: the rows of the range in their own order, or (order: place -> row of
  // the whole matrix, row_begin = 0) in the order of the join's image, where the list of place r sits at position r
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t kRsMaxIds = (uint32_t)rs_max_ids(slots), kRsWords = (uint32_t)rs_words(slots), kRsNormsAt = (uint32_t)rs_norms_at(slots);
  float *es = reinterpret_cast<float *>(smem) + (size_t)wv * 3 * cmax;
  uint32_t *ei = reinterpret_cast<uint32_t *>(es + cmax);
  float *ed = es + 2 * cmax;  // the bound of each pair (err_pos: the rows' plane errors by image position = what the lists hold)
  unsigned long long scored = 0;  // candidates this wave hands to the scoring kernel (the rows it fetches: bench.py's algorithmic bytes)
  for (uint32_t r = blockIdx.x * 4 + wv; r < row_count; r += gridDim.x * 4) {
    // (a_ids: the rows of the range as a list -- a rank of a build sharded by image positions finishes the rows perm[pb .. pe))
    const uint32_t a = order ? order[r] : (a_ids ? a_ids[r] : row_begin + r);
    // lists kept by image POSITION (the own join on a permuted image, row_order.hip): row a's list sits at inv[a] and
    // holds positions; what leaves this kernel is row ids again
    const uint32_t lr = order ? r : (inv ? inv[a] : r);
    // what the record's header needs, and the row's own error: asked for now, used at the end of the row
    const float th_a = theta[a];
    const double nrm_a = norms[a];
    const bool force = force_exact && force_exact[a];
    const float e_self = err_pos ? err_pos[lr] : 0.f;
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
  if (scored_total && lane == 0 && scored) atomicAdd(scored_total, scored);
}

// The same pruning from lists kept per GROUP of 32 image positions (mfma_join_f16.hip, GROUPED): cnt[g] entries in the slots
// [g * 32 * cap, ..) of cand, an entry = (other position | local row << 27, score).  A workgroup takes a group: per half of 16
// rows it counts the half's entries per row, lays them out densely in LDS (a row keeps at most `cap`, what it kept in the per-row
// form; 16 * cap entries always fit), and each wave then prunes four rows exactly as above.  The group's entries are read four
// times, from L2 after the first.  Records go out by place = image position (`order`: position -> row).
constexpr uint32_t kGrpRows = 32, kGrpShift = 27, kGrpHalf = 16;
__global__ __launch_bounds__(256) void rerank_prune_grouped_kernel(uint32_t n, const float *__restrict__ theta, const double *__restrict__ norms,
                                                                   const uint32_t *__restrict__ gcnt, const uint32_t *__restrict__ cand, uint32_t cap,
                                                                   uint32_t K, const uint8_t *__restrict__ force_exact, double prune_delta,
                                                                   int prune_rectify, uint32_t *__restrict__ records, int slots,
                                                                   const uint32_t *__restrict__ order, unsigned long long *__restrict__ scored_total) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ uint32_t s_cnt[kGrpHalf], s_off[kGrpHalf + 1], s_fill[kGrpHalf];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t kRsMaxIds = (uint32_t)rs_max_ids(slots), kRsWords = (uint32_t)rs_words(slots), kRsNormsAt = (uint32_t)rs_norms_at(slots);
  float *es_all = reinterpret_cast<float *>(smem);                                   // [16 * cap]
  uint32_t *ei_all = reinterpret_cast<uint32_t *>(smem) + (size_t)kGrpHalf * cap;    // [16 * cap]
  const uint32_t n_groups = (n + kGrpRows - 1) / kGrpRows;
  unsigned long long scored = 0;
  for (uint32_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const uint32_t rows_in = min(kGrpRows, n - g * kGrpRows), gcap = rows_in * cap;
    const uint32_t total = gcnt[g], E = min(total, gcap);
    const bool group_overflow = total > gcap;  // entries were dropped: no row of the group can be certified from its list
    const uint2 *src = reinterpret_cast<const uint2 *>(cand) + (size_t)g * kGrpRows * cap;
    for (uint32_t h = 0; h < kGrpRows / kGrpHalf; ++h) {
      if (h * kGrpHalf >= rows_in) break;
      __syncthreads();
      if (threadIdx.x < kGrpHalf) {
        s_cnt[threadIdx.x] = 0;
        s_fill[threadIdx.x] = 0;
      }
      __syncthreads();
      for (uint32_t e = threadIdx.x; e < E; e += 256) {
        const uint32_t l = src[e].x >> kGrpShift;
        if ((l / kGrpHalf) == h) atomicAdd(&s_cnt[l % kGrpHalf], 1u);
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t l = 0; l < kGrpHalf; ++l) {
          s_off[l] = run;
          run += min(s_cnt[l], cap);
        }
        s_off[kGrpHalf] = run;
      }
      __syncthreads();
      for (uint32_t e = threadIdx.x; e < E; e += 256) {
        const uint2 v = src[e];
        const uint32_t l = v.x >> kGrpShift;
        if ((l / kGrpHalf) == h) {
          const uint32_t slot = atomicAdd(&s_fill[l % kGrpHalf], 1u);
          if (slot < cap) {
            ei_all[s_off[l % kGrpHalf] + slot] = v.x & ((1u << kGrpShift) - 1u);
            es_all[s_off[l % kGrpHalf] + slot] = __uint_as_float(v.y);
          }
        }
      }
      __syncthreads();
      for (uint32_t l = wv; l < kGrpHalf; l += 4) {
        const uint32_t r = g * kGrpRows + h * kGrpHalf + l;  // place = image position
        if (r >= n) break;
        const uint32_t a = order[r];
        uint32_t *ei = ei_all + s_off[l];
        float *es = es_all + s_off[l];
        uint32_t C = min(s_cnt[l], cap);
        bool overflow = group_overflow || s_cnt[l] > cap;
        C = prune_candidates(ei, es, C, K, prune_delta, prune_rectify, lane);
        if (C > kRsMaxIds) {  // more survivors than a record holds: the row goes on to the next stage
          overflow = true;
          C = kRsMaxIds;
        }
        scored += C;
        uint32_t *rec = records + (size_t)r * kRsWords;
        for (uint32_t c = lane; c < C; c += 64) {
          const uint32_t j = order[ei[c]];
          rec[kRsIdsAt + c] = j;
          reinterpret_cast<double *>(rec + kRsNormsAt)[c] = norms[j];
        }
        if (lane == 0) {
          const bool force = force_exact && force_exact[a];
          rec[0] = C | (overflow ? 0x100u : 0u) | (force ? 0x200u : 0u);
          rec[1] = __float_as_uint(theta[a]);
          const unsigned long long nb = (unsigned long long)__double_as_longlong(norms[a]);
          rec[2] = (uint32_t)nb;
          rec[3] = (uint32_t)(nb >> 32);
        }
        wave_sync();
      }
    }
  }
  if (scored_total && lane == 0 && scored) atomicAdd(scored_total, scored);
}

// candidates listed in all / groups that ran over, for the statistics of a build with lists by group
__global__ void join_group_stats_kernel(const uint32_t *__restrict__ gcnt, uint32_t n, uint32_t cap, unsigned long long *__restrict__ st) {
  const uint32_t n_groups = (n + kGrpRows - 1) / kGrpRows;
  unsigned long long sum = 0, over = 0, mx = 0;
  for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += gridDim.x * blockDim.x) {
    const uint32_t gcap = min(kGrpRows, n - g * kGrpRows) * cap, c = gcnt[g];
    sum += min(c, gcap);
    over += c > gcap ? min(kGrpRows, n - g * kGrpRows) : 0u;
    mx = max(mx, (unsigned long long)c);
  }
  for (int o = 32; o > 0; o >>= 1) {
    sum += __shfl_xor(sum, o, 64);
    over += __shfl_xor(over, o, 64);
    mx = max(mx, (unsigned long long)__shfl_xor(mx, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(st, sum);
    atomicMax(st + 1, mx);
    atomicAdd(st + 2, over);
  }
}

// A failed row's slot in the list: the returning atomic and its wait by hand, INSIDE the branch.  Left to the compiler the wait for
// the returned value (s_waitcnt vmcnt(0)) lands behind the branch's join, on the path of every row, and drains the two batches of
// row gathers in flight at every row end.
__device__ __forceinline__ uint32_t rs_take_fail_slot(uint32_t *fail_count) {
  uint32_t slot;
  const uint32_t zero = 0u, one = 1u;
  // (s_nop: the pointer may have just come back from a spill lane by v_readlane -- a VALU write of an SGPR that a VMEM instruction
  // reads needs five wait states, and the compiler does not look inside the statement)
  asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(slot) : "v"(zero), "v"(one), "s"(fail_count) : "memory");
  return slot;
}

typedef double rs_d2 __attribute__((ext_vector_type(2)));  // (a native vector: asm operands of struct type go through memory)
struct RsSet {
  rs_d2 v[kRsRows];  // this lane's 16 B of each candidate row of a batch
  rs_d2 x;             // ... and of the row the batch belongs to (stored at the row's first batch; otherwise an L2 hit
                       // that keeps every batch at the same number of loads)
};
constexpr int kRsLoads = kRsRows + 1;
// row j of X -> one register per lane (lane l: bytes [16 l, 16 l + 16) of the row).  NOT valid before rs_landed.
__device__ __forceinline__ void rs_gload16(rs_d2 &dst, uint32_t lane_off, const double *row) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(lane_off), "s"(row) : "memory");
}
// at most N vector-memory operations younger than the set are pending: it has landed (vmcnt retires in order)
template <int N>
__device__ __forceinline__ void rs_landed(RsSet &s) {
  static_assert(kRsRows == 16, "operand lists below");
  asm volatile("s_waitcnt vmcnt(%9)"
               : "+v"(s.v[0]), "+v"(s.v[1]), "+v"(s.v[2]), "+v"(s.v[3]), "+v"(s.v[4]), "+v"(s.v[5]), "+v"(s.v[6]), "+v"(s.v[7]), "+v"(s.x)
               : "n"(N));
  asm volatile("" : "+v"(s.v[8]), "+v"(s.v[9]), "+v"(s.v[10]), "+v"(s.v[11]), "+v"(s.v[12]), "+v"(s.v[13]), "+v"(s.v[14]), "+v"(s.v[15]));
}

__device__ __forceinline__ double rs_readlane(double v, uint32_t l) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, (int)l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), (int)l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// the rows of a wave as a stream of steps: (row, batch in row, 128-wide piece of the rows); a row without survivors
// still has one (empty) batch
struct RsCursor {
  uint32_t p, r, b, q, C, nb;  // p: place in the visiting order; r: the row (index into records / outputs) found there
  bool done;
  uint32_t words;   // of a record
  uint32_t p_end;   // end of this wave's share of the places
  const uint32_t *order;  // place -> row, or nullptr (the rows in their own order)
  __device__ __forceinline__ void enter(const uint32_t *__restrict__ records, uint32_t place) {
    p = place;
    b = 0;
    q = 0;
    done = place >= p_end;
    r = done ? 0u : (order ? order[place] : place);
    C = done ? 0u : (records[(size_t)place * words] & 0xFFu);  // (records are kept by place)
    nb = C ? (C + kRsRows - 1) / kRsRows : 1u;
  }
  // next step: the next piece of the batch, the next batch, the wave's next row
  __device__ __forceinline__ void advance(const uint32_t *__restrict__ records, uint32_t pieces, uint32_t stride) {
    if (done) return;
    if (++q < pieces) return;
    q = 0;
    if (++b == nb) enter(records, p + stride);
  }
};

// Rows of d = 128 x pieces values: a step stages one 128-wide piece of the batch's 16 candidate rows (and of the row itself);
// the chains of a batch run on across its pieces.
template <uint32_t PIECES, int SLOTS>
__global__ __launch_bounds__(64) void rerank_score_d128_kernel(const double *__restrict__ X, uint32_t row_begin, uint32_t row_count,
                                                               const uint32_t *__restrict__ records, uint32_t K, int rectify,
                                                               GraphBound bnd, double *__restrict__ out_key,
                                                               uint32_t *__restrict__ out_idx, uint32_t *__restrict__ fail_rows,
                                                               uint32_t *__restrict__ fail_count, uint32_t dbg,
                                                               const uint32_t *__restrict__ a_ids, const uint32_t *__restrict__ order) {
  constexpr uint32_t pieces = PIECES, d = 128u * PIECES;
  // row of the range at index r (wave-uniform: a_ids comes through the scalar cache)
  auto arow = [&](uint32_t r) __attribute__((always_inline)) { return a_ids ? a_ids[r] : row_begin + r; };
  constexpr uint32_t kRsWords = (uint32_t)rs_words(SLOTS), kRsNormsAt = (uint32_t)rs_norms_at(SLOTS);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  double *xrow = reinterpret_cast<double *>(smem);                       // the row itself
  unsigned char *tile = smem + 1024;                                      // the batch whose chains run
  double *ek = reinterpret_cast<double *>(tile + kRsRows * kRsPitch);     // keys and ids of the scored candidates
  uint32_t *ei = reinterpret_cast<uint32_t *>(ek + 64 * SLOTS);
  const uint32_t lane_off = (uint32_t)lane * 16u;
  const uint32_t chain_len = (dbg & 1) ? 16u : 128u;  // dbg: timing experiments only (outputs undefined)

  // Which rows this wave scores.  `order` (the row order of the join's image: rows of one direction group side by side) makes
  // the waves that run at the same time work on neighbouring rows, whose candidates are largely the SAME rows: they are then
  // fetched from HBM once and found in L2 by the others.  L2 is per XCD and workgroups are dealt to the eight XCDs in turn, so
  // XCD x = blockIdx.x % 8 takes the x-th eighth of the places and its waves walk it side by side.
  uint32_t stride = gridDim.x, first = blockIdx.x, last = row_count;
  if (gridDim.x % 8u == 0u) {
    const uint32_t x = blockIdx.x & 7u, chunk = (row_count + 7u) / 8u;
    stride = gridDim.x / 8u;
    first = x * chunk + (blockIdx.x >> 3);
    last = min((x + 1u) * chunk, row_count);
  }
  RsCursor is, ps;  // issue side (two batches ahead), process side
  is.words = kRsWords;
  is.p_end = last;
  is.order = order;
  is.enter(records, first);
  ps = is;
  if (ps.done) return;

  // the ids of a batch come through the scalar cache (uniform addresses); slots past the count fetch the row itself
  // (L2-hot), so that every batch is exactly kRsLoads loads and the waits can be counted
  auto batch_ids = [&](const RsCursor &cu, uint32_t (&ids)[kRsRows]) __attribute__((always_inline)) {
    const uint32_t row = cu.done ? 0u : cu.r;
    const uint32_t *rec = records + (size_t)(cu.done ? 0u : cu.p) * kRsWords + kRsIdsAt + (cu.done ? 0u : cu.b * kRsRows);
    const uint4 q0 = *reinterpret_cast<const uint4 *>(rec), q1 = *reinterpret_cast<const uint4 *>(rec + 4);
    const uint4 q2 = *reinterpret_cast<const uint4 *>(rec + 8), q3 = *reinterpret_cast<const uint4 *>(rec + 12);
    const uint32_t raw[kRsRows] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
    const uint32_t self = arow(row);
#pragma unroll
    for (int t = 0; t < kRsRows; ++t) ids[t] = (!cu.done && cu.b * kRsRows + t < cu.C) ? raw[t] : self;
  };
  // (slots past the count fetch the row itself; all lanes taking the SAME 16 bytes of it instead was measured: +0.3 ms)
#define AS_GL(S, T) rs_gload16(S.v[T], lane_off, X + (uint64_t)__builtin_amdgcn_readfirstlane((int)ids_[T]) * d + is.q * 128u);
#define AS_ISSUE(S)                                                                                                         \
  {                                                                                                                         \
    uint32_t ids_[kRsRows];                                                                                                 \
    batch_ids(is, ids_);                                                                                                    \
    AS_GL(S, 0) AS_GL(S, 1) AS_GL(S, 2) AS_GL(S, 3) AS_GL(S, 4) AS_GL(S, 5) AS_GL(S, 6) AS_GL(S, 7)                         \
    AS_GL(S, 8) AS_GL(S, 9) AS_GL(S, 10) AS_GL(S, 11) AS_GL(S, 12) AS_GL(S, 13) AS_GL(S, 14) AS_GL(S, 15)                   \
    rs_gload16(S.x, lane_off,                                                                                               \
               X + (uint64_t)__builtin_amdgcn_readfirstlane((int)arow(is.done ? 0u : is.r)) * d + is.q * 128u);              \
    is.advance(records, pieces, stride);                                                                                    \
  }

  bool p_overflow = false, p_force = false;  // header of the process side's row
  float p_theta = 0.f;
  double p_na = 0.0;

  // a landed set -> the LDS tile and the row's own piece
#define AS_ST(S, T) *reinterpret_cast<rs_d2 *>(tile + T * kRsPitch + lane_off) = S.v[T];
#define AS_STAGE(S)                                                                                                 \
  {                                                                                                                 \
    *reinterpret_cast<rs_d2 *>(reinterpret_cast<unsigned char *>(xrow) + lane_off) = S.x; /* piece q of the row */   \
    AS_ST(S, 0) AS_ST(S, 1) AS_ST(S, 2) AS_ST(S, 3) AS_ST(S, 4) AS_ST(S, 5) AS_ST(S, 6) AS_ST(S, 7)                 \
    AS_ST(S, 8) AS_ST(S, 9) AS_ST(S, 10) AS_ST(S, 11) AS_ST(S, 12) AS_ST(S, 13) AS_ST(S, 14) AS_ST(S, 15)           \
  }
  double acc = 0.0;  // this lane's chain, carried across the pieces of a batch
  auto process = [&]() {
    uint32_t ids[kRsRows];
    batch_ids(ps, ids);
    double nrm[kRsRows];  // norms of the batch's candidates (slots past the count: whatever follows, unused)
    {
      const double *np = reinterpret_cast<const double *>(records + (size_t)ps.p * kRsWords + kRsNormsAt) + ps.b * kRsRows;
#pragma unroll
      for (int t = 0; t < kRsRows; ++t) nrm[t] = np[t];
    }
    if (ps.b == 0 && ps.q == 0) {  // row entry
      const uint4 h = *reinterpret_cast<const uint4 *>(records + (size_t)ps.p * kRsWords);
      p_overflow = (h.x & 0x100u) != 0;
      p_force = (h.x & 0x200u) != 0;
      p_theta = __uint_as_float(h.y);
      p_na = __longlong_as_double((long long)(((unsigned long long)h.w << 32) | h.z));
    }
    wave_sync();
    const uint32_t c = ps.b * kRsRows + (uint32_t)lane;
    if (!(dbg & 4) && lane < kRsRows && c < ps.C) {
      // the reference's chain (src/core.rs:204-210): k ascending, multiply and add rounded separately
      const rs_d2 *bj = reinterpret_cast<const rs_d2 *>(tile + (size_t)lane * kRsPitch);
      const rs_d2 *xj = reinterpret_cast<const rs_d2 *>(xrow);
      if (ps.q == 0) acc = 0.0;
      // groups of 8 k: the next group's LDS reads are issued before this group's chain
      rs_d2 xa[4], ba[4], xb[4], bb[4];
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        xa[h] = xj[h];
        ba[h] = bj[h];
      }
#pragma unroll 1
      for (uint32_t g = 0; g < chain_len / 16; ++g) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          xb[h] = xj[g * 8 + 4 + h];
          bb[h] = bj[g * 8 + 4 + h];
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          acc = __dadd_rn(acc, __dmul_rn(xa[h].x, ba[h].x));
          acc = __dadd_rn(acc, __dmul_rn(xa[h].y, ba[h].y));
        }
        if (g + 1 < chain_len / 16) {
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            xa[h] = xj[g * 8 + 8 + h];
            ba[h] = bj[g * 8 + 8 + h];
          }
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          acc = __dadd_rn(acc, __dmul_rn(xb[h].x, bb[h].x));
          acc = __dadd_rn(acc, __dmul_rn(xb[h].y, bb[h].y));
        }
      }
      if (ps.q + 1 == pieces) {
        double nj = nrm[0];
#pragma unroll
        for (int t = 1; t < kRsRows; ++t) nj = lane == t ? nrm[t] : nj;
        uint32_t jj = ids[0];
#pragma unroll
        for (int t = 1; t < kRsRows; ++t) jj = lane == t ? ids[t] : jj;
        const uint32_t a = arow(ps.r);
        double key = __builtin_inf();
        uint32_t id = kNoIdx;
        if (a != jj) {  // GraphDistPolicy::key with the operands at hand
          const double denom = __dmul_rn(p_na, nj);
          double cs = denom > 1e-12 ? __ddiv_rn(acc, denom) : 0.0;
          if (rectify) {
            cs = fmin(fmax(cs, -1.0), 1.0);
            cs = fmax(cs, 0.0);
          }
          const double kx = __dsub_rn(1.0, cs);
          if (!(kx != kx)) {
            key = kx;
            id = jj;
          }
        }
        ek[c] = key;
        ei[c] = id;
      }
    }
    wave_sync();
    {
      const bool row_done = ps.q + 1 == pieces && ps.b + 1 == ps.nb;
      if (!row_done) {
        ps.advance(records, pieces, stride);
        return;
      }
    }
    // ---- the row is scored: rank, write its K best, certify ----
    const uint32_t r = ps.r, C = ps.C;
    double kth = -__builtin_inf();
    uint32_t n_valid = 0;
    if (!(dbg & 2)) {
      // lane l holds candidates l and 64 + l; every candidate's (key, id) passes by as a wave-uniform value (readlane:
      // no LDS round trip per comparison)
      double kv[SLOTS];
      uint32_t iv[SLOTS], rank[SLOTS];
#pragma unroll
      for (int q = 0; q < SLOTS; ++q) {
        kv[q] = __builtin_inf();
        iv[q] = kNoIdx;
        rank[q] = 0;
        if ((uint32_t)(64 * q + lane) < C) {
          kv[q] = ek[64 * q + lane];
          iv[q] = ei[64 * q + lane];
        }
      }
      if (C <= 64u) {
        // (nearly every row: one candidate per lane, and the keys compare as integers -- the order-preserving image of a double: sign
        // bit set for non-negative values, all bits flipped for negative ones (a cosine a rounding above 1 gives a distance just below
        // 0 where nothing is rectified); -0 and NaN do not occur: 1 - c is +0 at c = 1, NaN keys were replaced by +inf.  The general form
        // below spends as much again on the empty second slot: ranking was 1.3 of the kernel's 6 ms)
        const unsigned long long kraw = (unsigned long long)__double_as_longlong(kv[0]);
        const unsigned long long kb = (kraw >> 63) ? ~kraw : (kraw | 0x8000000000000000ull);
        for (uint32_t c2 = 0; c2 < C; ++c2) {
          const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)kb, (int)c2);
          const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(kb >> 32), (int)c2);
          const uint32_t ic = (uint32_t)__builtin_amdgcn_readlane((int)iv[0], (int)c2);
          const unsigned long long kc = ((unsigned long long)hi << 32) | lo;
          rank[0] += (kc < kb || (kc == kb && ic < iv[0])) ? 1u : 0u;
        }
      } else {
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
      }
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
    }
    for (int o = 32; o > 0; o >>= 1) {
      kth = fmax(kth, __shfl_xor(kth, o, 64));
      n_valid += __shfl_xor(n_valid, o, 64);
    }
    for (uint32_t s = n_valid + lane; s < K; s += 64) {
      out_key[(uint64_t)r * K + s] = __builtin_inf();
      out_idx[(uint64_t)r * K + s] = kNoIdx;
    }
    // every pair that is not in the lists has an approximate cosine below theta
    const bool ok = !p_overflow && !p_force && n_valid >= K && kth < bnd.lower_bound(-(double)p_theta);
    if (!ok && lane == 0 && !dbg) fail_rows[rs_take_fail_slot(fail_count)] = r;
    wave_sync();
    ps.enter(records, ps.p + stride);
  };

  RsSet sA, sB;
  AS_ISSUE(sA)
  AS_ISSUE(sB)
  for (;;) {
    // a step: the older set has landed (the other one's loads are younger) -> LDS tile; its registers take the batch two
    // ahead at once; then the chains run from the tile with two batches in flight
#define AS_STEP(P)             \
  {                            \
    rs_landed<kRsLoads>(P);    \
    AS_STAGE(P)                \
    AS_ISSUE(P)                \
    process();                 \
    if (ps.done) break;        \
  }
    AS_STEP(sA)
    AS_STEP(sB)
#undef AS_STEP
  }
#undef AS_ISSUE
#undef AS_STAGE
#undef AS_GL
#undef AS_ST
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the batches issued past the end
}


}  // namespace asx
