// mfma_join_f16_wide.hip -- the symmetric threshold join on one f16 plane (mfma_join_f16.hip: scheme, bound, meaning of the
// lists) for rows WIDER than 128: configs[4] of the baseline is 1 M x 768.
//
// The narrow kernel keeps a wave's A fragments in registers for a whole sweep of B and finishes a 64 x 32 block -- test
// included -- before it starts the next.  With 768 columns the fragments of 64 rows are 384 registers, and re-reading them
// per block from L2 costs 8 x the bytes the matrix pipe can be fed with.  So the wide form is shaped like a GEMM: the
// products of a pair of 256-row super tiles stay in the accumulators over the whole k range (a wave owns 128 x 128 of them:
// 16 blocks of 32 x 32 = 256 accumulator registers; four waves, one workgroup per CU so that a wave has 512 registers),
// A and B arrive as panels of 64 columns of k (32 KB each) through a 2-deep LDS-DMA ring with one barrier per panel, and
// every fragment read from LDS feeds four MFMAs.  Per panel a wave issues 64 MFMAs, 32 ds_read_b128 and 16 DMA pieces.
// The threshold test runs once per super-tile pair, on the finished accumulators: as in the narrow form the accumulators
// start at -theta_row (scaled), so "some product reaches its row's threshold" is max(acc) >= 0, the column side is tested
// conservatively on the same maximum, and only blocks that pass go through the element-wise test.
//
// Bytes: a pair of super tiles moves 2 x 256 x d x 2 B through L2 -> LDS for 2 x 256 x 256 x d flop: 7.8 B per kflop
// (the narrow form: 1.9).  The workgroups of an XCD (32 CUs, one L2) therefore walk the pairs in rectangles of
// 4 row super tiles x 8 column offsets, each XCD from its own counter: 15 distinct tiles per 32 pairs reach its L2
// instead of 64.
#include <hip/hip_fp16.h>

#include <type_traits>

#include "mfma_common.h"

namespace asx {

namespace {

typedef float v16f32 __attribute__((ext_vector_type(16)));
typedef float v4f32 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

constexpr int kWS = 256;                  // rows per super tile of the wide form (two packed 128-row tiles)
constexpr int kPK = 4;                    // 16-wide k steps per staged panel
constexpr int kPanelVecs = kPK * 4 * 64;  // 16-B vectors of one 128-row tile's share of a panel (16 KB)
constexpr uint32_t kQueueW = 256;         // queued appends per wave
constexpr float kProductScaleW = 16384.f;
constexpr int kSlotFloatsW = 36;
constexpr size_t kBufBytes = (size_t)4 * kPanelVecs * 16;                    // A (2 tiles) + B (2 tiles) of a panel: 64 KB
constexpr size_t kThrAt = 2 * kBufBytes;  // [2 units][row thresholds | column thresholds or lambdas | row lambdas | row guesses] x 256 floats
constexpr size_t kSlotBytes = 4 * 256 * sizeof(float);
constexpr size_t kQueueAt = kThrAt + 2 * kSlotBytes;
constexpr size_t kSlotsAt = kQueueAt + (size_t)4 * kQueueW * (sizeof(uint2) + sizeof(float));
constexpr size_t kWideLds = kSlotsAt + (size_t)4 * 4 * kSlotFloatsW * sizeof(float);

template <int B, int E, class F>
__device__ __forceinline__ void static_for_w(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for_w<B + 1, E>(f);
  }
}
__device__ __forceinline__ uint32_t lds_addr_w(const void *p) { return (uint32_t)(size_t)(lds_void *)p; }
template <int OFF>
__device__ __forceinline__ void lds_read16_w(f16x8 &dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read16f_w(v4f32 &dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_read4f_w(float &dst, uint32_t addr) {
  asm volatile("ds_read_b32 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
}
// at most N LDS reads younger than these eight fragments are pending
template <int N>
__device__ __forceinline__ void lds_wait8_w(f16x8 (&a)[4], f16x8 (&b)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N));
}
__device__ __forceinline__ void lds_write16f_w(uint32_t addr, const v4f32 &v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_write8_w(uint32_t addr, uint2 v) {
  const unsigned long long b = ((unsigned long long)v.y << 32) | v.x;
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(b) : "memory");
}
__device__ __forceinline__ void lds_write4f_w(uint32_t addr, float v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_wait_all_w() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ float thr_scaled_w(float t) {  // as thr_scaled of the narrow form
  if (!(t <= 1.5f)) return __builtin_inff();
  return __builtin_fmaxf(t, -1.5f) * kProductScaleW;
}
__device__ __forceinline__ float max16_w(const v16f32 &a) {
  float m = __builtin_fmaxf(__builtin_fmaxf(a[0], a[1]), a[2]);
  m = __builtin_fmaxf(__builtin_fmaxf(m, a[3]), a[4]);
  m = __builtin_fmaxf(__builtin_fmaxf(m, a[5]), a[6]);
  m = __builtin_fmaxf(__builtin_fmaxf(m, a[7]), a[8]);
  m = __builtin_fmaxf(__builtin_fmaxf(m, a[9]), a[10]);
  m = __builtin_fmaxf(__builtin_fmaxf(m, a[11]), a[12]);
  m = __builtin_fmaxf(__builtin_fmaxf(m, a[13]), a[14]);
  return __builtin_fmaxf(m, a[15]);
}
__device__ __forceinline__ void append_w(uint32_t *__restrict__ cnt, uint32_t *__restrict__ cand, uint32_t cap, uint32_t target,
                                         uint32_t other, float score) {
  const uint32_t pos = atomicAdd(cnt + target, 1u);
  if (pos < cap) reinterpret_cast<uint2 *>(cand)[(size_t)target * cap + pos] = make_uint2(other, __float_as_uint(score));
}

// MODE 0 behind the first pass: is the w-th pair of rectangle `sb` (4 row super tiles x 8 column offsets) a pair of the pattern that
// the first pass flagged (or did not look at)?  The first pass swept every unordered pair of 512-row super tiles ONCE, from the side
// its cyclic half pattern names (looked): rows of A against columns of B, or the other way round.  The mask holds the row super
// tiles of THIS call's range [A0, A1) (a rank of a sharded build: its own positions) against every column; a pair the first pass
// saw from a row super tile outside the range (a few pairs near the antipodal offset) is unknown here and contracted.
__device__ __forceinline__ bool wide_pair_wanted(uint32_t sb, uint32_t w, uint32_t T2, uint32_t I_begin, uint32_t I_count, uint32_t n_ob,
                                                 const unsigned long long *__restrict__ mask, uint32_t mask_pitch, uint32_t *I_out, uint32_t *J_out,
                                                 bool one_sided = false) {
  const uint32_t n_norm = (T2 - 1) / 2;
  const uint32_t ib = sb / n_ob, ob = sb - ib * n_ob;
  const uint32_t di = w & 3u, off = ob * 8u + (w >> 2);
  if (ib * 4u + di >= I_count) return false;
  const uint32_t I = I_begin + ib * 4u + di;
  if (one_sided) {  // MODE 4 behind the rectangular first pass (prefilter_i8_rect): row super tile I of the batch against column super tile `off`
    if (off >= T2) return false;
    *I_out = I;
    *J_out = off;
    return !mask || (uint32_t)(mask[(size_t)(I >> 1) * mask_pitch + off] >> (32u * (I & 1u))) != 0u;
  }
  const bool antipodal = (T2 % 2 == 0) && (I < T2 / 2) && T2 > 1;
  if (off >= 1 + n_norm + (antipodal ? 1u : 0u)) return false;
  uint32_t J = I + off;
  J = J >= T2 ? J - T2 : J;
  *I_out = I;
  *J_out = J;
  if (!mask) return true;
  const uint32_t A = I >> 1, B = J >> 1, T5 = T2 >> 1, nn5 = (T5 - 1) / 2;
  const uint32_t A0 = I_begin >> 1, A1 = (I_begin + I_count + 1) >> 1;
  auto looked = [&](uint32_t from, uint32_t to) -> bool {
    const uint32_t dl = to >= from ? to - from : to + T5 - from;
    return dl <= nn5 || ((T5 % 2 == 0) && T5 > 1 && from < T5 / 2 && dl == nn5 + 1);
  };
  bool known = false;
  unsigned long long bits = 0;
  if (looked(A, B)) {
    known = true;
    bits |= mask[(size_t)(A - A0) * mask_pitch + J] >> (32u * (I & 1u));
  }
  if (looked(B, A) && B >= A0 && B < A1) {
    known = true;
    bits |= mask[(size_t)(B - A0) * mask_pitch + I] >> (32u * (J & 1u));
  }
  return !(known && (uint32_t)bits == 0u);
}
// The wanted pairs as a list (round 5): rectangle by rectangle in the order of the walk, so that consecutive entries share super
// tiles.  Until then thread 0 of every workgroup looked for its next pair by drawing candidates from its XCD's counter one returning
// atomic and one mask read at a time: at 1 M x 768 that search -- 7.6 M candidates for 0.4 M wanted pairs, 30 k serial round trips per
// workgroup -- took the same 89 ms whether the first pass left 0.62 M pairs or 0.44 M: it, not the contraction, set the pass's time.
__global__ __launch_bounds__(256) void wide_unit_list_kernel(uint32_t T2, uint32_t I_begin, uint32_t I_count, uint32_t n_sb, uint32_t n_ob,
                                                             const unsigned long long *__restrict__ mask, uint32_t mask_pitch,
                                                             uint32_t *__restrict__ counts, const uint64_t *__restrict__ offsets,
                                                             uint2 *__restrict__ list, int one_sided) {
  // counts given (offsets null): the number of wanted pairs of every rectangle; offsets given: the pairs themselves
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (uint32_t sb0 = (blockIdx.x * 4 + wv) * 2; sb0 < n_sb; sb0 += gridDim.x * 8) {
    const uint32_t sb = sb0 + (lane >> 5), w = lane & 31;  // two rectangles per wave
    uint32_t I = 0, J = 0;
    const bool want = sb < n_sb && wide_pair_wanted(sb, w, T2, I_begin, I_count, n_ob, mask, mask_pitch, &I, &J, one_sided != 0);
    const unsigned long long bal = __ballot(want);
    const uint32_t half = (uint32_t)((lane >> 5) ? (bal >> 32) : bal);
    if (offsets) {
      if (want) list[offsets[sb] + (uint32_t)__builtin_popcount(half & ((1u << w) - 1u))] = make_uint2(I, J);
    } else if (w == 0 && sb < n_sb) {
      counts[sb] = (uint32_t)__builtin_popcount(half);
    }
  }
}

// P: f16 image of ALL rows with KS = a multiple of kPK k steps per row (zero padded), T2 super tiles of 256 rows (the image
// is padded to whole 512-row super tiles: an even number of them).
// MODE 0: the symmetric join.  theta: T2 * 256 thresholds, +inf on padding rows.  Super tile I meets the column super
//   tiles I .. I + (T2 - 1) / 2 (mod T2), the antipodal one only from the smaller index.
// MODE 1: one-sided (the second pass over the rows the join left open).  PA: image of the nA gathered rows (I_count super
//   tiles of 256, zero padded), a_ids their global ids, theta their thresholds by POSITION (+inf on padding); every PA super
//   tile meets every super tile of P; lists (cnt / cand) are indexed by position.
// MODE 2: thresholds.  PA = the image of the sample (I_count = its super tiles, no padding rows), rows = super tiles
//   [I_begin, I_begin + T2) of P; a workgroup sweeps the whole sample for one row super tile at a time and leaves, per row,
//   64 maxima over 64 disjoint sub-samples (column position mod 32 x parity of the 128-row tile, as mfma_theta_bf16.hip)
//   in maxima[row - 256 I_begin][64]; a second kernel picks the order statistics.
// MODE 3: MODE 2 for the search: P = image of the queries, PA of the sampled items; what is reduced is the approximate SCORE
//   alpha cos + beta (1 - min(|lambda_q - lambda_x|, 1)) (src/core.rs:229-239): row_lam / col_lam = f32 lambdas of the queries
//   (padded) and of the sample.
// MODE 4: MODE 1 for the search: PA = image of the queries, theta = (g_q - beta) / alpha (what the cosine alone must reach:
//   the coarse test), row_g = g_q - beta, row_lam / col_lam = f32 lambdas of the queries / of ALL items (padded); a pair is
//   listed when alpha s - min(|beta lambda_q - beta lambda_x|, beta) >= g_q - beta, with that value as its score.
template <int MODE>
__global__ __launch_bounds__(256, 1) void pair_join_f16_wide_kernel(const f16x8 *__restrict__ P, const f16x8 *__restrict__ PA,
                                                                    const uint32_t *__restrict__ a_ids, uint32_t KS, uint32_t T2,
                                                                    uint32_t I_begin, uint32_t I_count, uint32_t n_rows, uint32_t n,
                                                                    const float *__restrict__ theta, uint32_t cap,
                                                                    uint32_t *__restrict__ cnt, uint32_t *__restrict__ cand,
                                                                    float *__restrict__ maxima, uint32_t *__restrict__ unit_counters,
                                                                    int dbg, unsigned long long *__restrict__ stamps,
                                                                    const float *__restrict__ col_lam, const float *__restrict__ row_lam,
                                                                    const float *__restrict__ row_g, float alpha, float beta,
                                                                    const uint32_t *__restrict__ perm,
                                                                    const unsigned long long *__restrict__ mask, uint32_t mask_pitch,
                                                                    const uint32_t *__restrict__ grp_start = nullptr, uint32_t grp_supers = 0,
                                                                    const uint2 *__restrict__ unit_list = nullptr, uint32_t n_list = 0) {
  // perm (MODE 0 / 1): P is packed in a permuted order (image position -> row): list entries are translated where they are
  // appended.  mask (MODE 0): what the int8 first pass found (mfma_prefilter_i8.hip; zeroed where it did not look): a word per
  // (512-row super tile, 256-column stage), a byte per 64-row wave; a pair of 256-row super tiles none of whose bits is set in
  // either orientation holds no pair of rows that reaches a threshold and is skipped.
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ uint32_t s_next[2];  // (I, J) of the unit after the current one; I = ~0: none
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lh = lane >> 5, lc = lane & 31;
  const int wr = wv >> 1, wc = wv & 1;  // the wave's 128 x 128 share: tile wr of the A super tile, tile wc of the B one
  uint2 *queue = reinterpret_cast<uint2 *>(smem + kQueueAt) + (size_t)wv * kQueueW;
  float *queue_s = reinterpret_cast<float *>(smem + kQueueAt + (size_t)4 * kQueueW * sizeof(uint2)) + (size_t)wv * kQueueW;
  float *ex = reinterpret_cast<float *>(smem + kSlotsAt) + (size_t)wv * 4 * kSlotFloatsW;
  uint32_t qn = 0;  // wave-uniform
  auto lanes_below = [&](unsigned long long m) -> uint32_t {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  };
  auto flush_queue = [&]() {
    wave_sync();
    for (uint32_t base = 0; base < qn; base += 64) {
      const uint32_t e = base + lane;
      if (e < qn) {
        const uint2 v = queue[e];
        if (perm && MODE == 0) append_w(cnt, cand, cap, perm[v.x], perm[v.y], queue_s[e]);
        else if (perm && MODE == 1) append_w(cnt, cand, cap, v.x, perm[v.y], queue_s[e]);
        // (MODE 4: both images may be in an order of their own -- a_ids: position in the query image -> query, perm: item position -> item)
        else if (MODE == 4) append_w(cnt, cand, cap, a_ids ? a_ids[v.x] : v.x, perm ? perm[v.y] : v.y, queue_s[e]);
        else append_w(cnt, cand, cap, v.x, v.y, queue_s[e]);
      }
    }
    wave_sync();
    qn = 0;
  };

  // ---- MODE 0 / 1: the pairs of this workgroup's XCD, rectangles of 4 row super tiles x 8 column offsets, from the XCD's
  // own counter.  MODE 2: a row super tile from the one counter, then the sample super tiles in order.
  const uint32_t n_norm = (T2 - 1) / 2;
  constexpr bool kSweep = MODE == 2 || MODE == 3;   // a row super tile at a time against every column super tile of PA
  constexpr bool kOneSided = MODE == 1 || MODE == 4;
  const uint32_t l_max = MODE == 0 ? 1 + n_norm + 1 : T2;
  const uint32_t n_ib = (I_count + 3) / 4, n_ob = (l_max + 7) / 8;
  const uint32_t xcd = blockIdx.x & 7u;
  uint32_t t_row = 0xFFFFFFFFu, t_col = 0, t_end = 0;  // MODE 2, thread 0: where the sweep stands
  auto fetch_unit = [&]() {  // thread 0: the next valid pair into s_next
    if (kSweep) {
      // (MODE 2 with a_ids: the GROUPED sample -- a_ids[2 t], a_ids[2 t + 1] = the range of sample super tiles row super tile t
      // sweeps: those of the pivot groups its rows belong to, join_thresholds_f16_wide_grouped)
      if (t_row == 0xFFFFFFFFu || ++t_col == t_end) {
        const uint32_t v = atomicAdd(unit_counters, 1u);
        t_row = v < T2 ? v : 0xFFFFFFFFu;
        t_col = 0;
        t_end = I_count;
        if (v >= T2) {
          s_next[0] = 0xFFFFFFFFu;
          return;
        }
        if (MODE == 2 && a_ids) {
          t_col = a_ids[2 * v];
          t_end = min(a_ids[2 * v + 1], I_count);
          if (t_end <= t_col) t_end = t_col + 1;
        }
      }
      s_next[0] = I_begin + t_row;
      s_next[1] = t_col;
      return;
    }
    for (;;) {
      const uint32_t v = atomicAdd(unit_counters + xcd, 1u);
      const uint32_t sb = xcd + 8u * (v >> 5), w = v & 31u;
      if ((MODE == 0 || MODE == 4 || MODE == 1) && unit_list) {  // the wanted pairs are listed: XCD x takes the x-th of every eight runs of 32 entries
        const uint32_t e = sb * 32u + w;
        if (sb * 32u >= n_list) {
          s_next[0] = 0xFFFFFFFFu;
          return;
        }
        if (e >= n_list) continue;
        const uint2 pr = unit_list[e];
        s_next[0] = pr.x;
        s_next[1] = pr.y;
        return;
      }
      if (sb >= n_ib * n_ob) {
        s_next[0] = 0xFFFFFFFFu;
        return;
      }
      const uint32_t ib = sb / n_ob, ob = sb - ib * n_ob;
      const uint32_t di = w & 3u, off = ob * 8u + (w >> 2);
      if (ib * 4u + di >= I_count) continue;
      const uint32_t I = I_begin + ib * 4u + di;
      if (kOneSided) {
        if (off >= T2) continue;
        s_next[0] = I;
        s_next[1] = off;
        return;
      }
      uint32_t J = 0, I2 = 0;
      if (!wide_pair_wanted(sb, w, T2, I_begin, I_count, n_ob, MODE == 0 ? mask : nullptr, mask_pitch, &I2, &J)) continue;
      s_next[0] = I;
      s_next[1] = J;
      return;
    }
  };

  // DMA of panel p of the pair (I, J) into ring buffer `buf`: 64 pieces of 1 KB, 16 per wave (wave 0 / 1: the two tiles of
  // A, 2 / 3: of B); pieces [q0, q0 + 4) of the wave's 16 with every k step of the panel before
  const uint32_t np = KS / kPK;
  auto dma_panel = [&](uint32_t I, uint32_t J, uint32_t p, int buf, int part) {
    // rows (waves 0, 1): the gathered rows' image in MODE 1; columns (waves 2, 3): the sample's in MODE 2
    const uint32_t S = (wv < 2 ? I : J), tile = 2u * S + (uint32_t)(wv & 1);
    const f16x8 *img = (kOneSided && wv < 2) || (kSweep && wv >= 2) ? PA : P;
    const f16x8 *src = img + ((size_t)tile * KS + (size_t)p * kPK) * 256 + lane;
    unsigned char *dst = smem + (size_t)buf * kBufBytes + (size_t)wv * kPanelVecs * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = part * 4 + i;
      __builtin_amdgcn_global_load_lds((gbl_void *)(src + q * 64), (lds_void *)(dst + q * 1024), 16, 0, 0);
    }
  };
  // thresholds of the pair's rows and columns -> LDS (slot = parity of the unit): 1 KB each, waves 0 and 2
  auto dma_thr = [&](uint32_t I, uint32_t J, int slot) {
    unsigned char *dst = smem + kThrAt + (size_t)slot * kSlotBytes;
    const float *src = nullptr;
    if (wv == 0 && MODE != 2 && MODE != 3) src = theta + (size_t)I * kWS;                 // row thresholds
    if (wv == 1 && MODE == 0) src = theta + (size_t)J * kWS;                               // column thresholds
    if (wv == 1 && MODE == 3) src = col_lam + (size_t)J * kWS;                             // lambdas of the sample
    if (wv == 1 && MODE == 4) src = col_lam + (size_t)J * kWS;                             // lambdas of the items
    if (wv == 2 && (MODE == 3 || MODE == 4)) src = row_lam + (size_t)(I - (MODE == 3 ? I_begin : 0u)) * kWS;
    if (wv == 3 && MODE == 4) src = row_g + (size_t)I * kWS;
    if (src) __builtin_amdgcn_global_load_lds((gbl_void *)(src + lane * 4), (lds_void *)(dst + (size_t)wv * 1024), 16, 0, 0);
  };

  unsigned long long n_blk = 0, n_pass = 0;
  if (tid == 0) fetch_unit();
  __syncthreads();
  uint32_t I = s_next[0], J = s_next[1];
  if (I == 0xFFFFFFFFu) return;
  __syncthreads();
  int buf = 0, slot = 0;
  static_for_w<0, 4>([&](auto part) { dma_panel(I, J, 0, 0, decltype(part)::value); });
  dma_thr(I, J, 0);

  v16f32 acc[4][4];
  v16f32 rmax[kSweep ? 4 : 1];  // MODE 2: running maxima of the wave's rows over the columns = lc (mod 32) of its half
  if (kSweep) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) rmax[kSweep ? q : 0][r] = -__builtin_inff();
  }
  for (;;) {
    if (tid == 0) fetch_unit();  // visible after the first barrier of the panel loop
    uint32_t In = 0xFFFFFFFFu, Jn = 0;
    f16x8 fa[2][4], fb[2][4];
#define AS_FRAGS(SET, KSTEP, A_ADDR, B_ADDR)                        \
  {                                                                 \
    lds_read16_w<((KSTEP)*4 + 0) * 1024>(fa[SET][0], A_ADDR);       \
    lds_read16_w<((KSTEP)*4 + 1) * 1024>(fa[SET][1], A_ADDR);       \
    lds_read16_w<((KSTEP)*4 + 2) * 1024>(fa[SET][2], A_ADDR);       \
    lds_read16_w<((KSTEP)*4 + 3) * 1024>(fa[SET][3], A_ADDR);       \
    lds_read16_w<((KSTEP)*4 + 0) * 1024>(fb[SET][0], B_ADDR);       \
    lds_read16_w<((KSTEP)*4 + 1) * 1024>(fb[SET][1], B_ADDR);       \
    lds_read16_w<((KSTEP)*4 + 2) * 1024>(fb[SET][2], B_ADDR);       \
    lds_read16_w<((KSTEP)*4 + 3) * 1024>(fb[SET][3], B_ADDR);       \
  }
    for (uint32_t p = 0; p < np; ++p) {
      // fragments: A (ks, sa) of the wave's A tile, B (ks, sb) of its B tile
      const uint32_t a_addr = lds_addr_w(smem + (size_t)buf * kBufBytes + (size_t)wr * kPanelVecs * 16) + (uint32_t)lane * 16u;
      const uint32_t b_addr = lds_addr_w(smem + (size_t)buf * kBufBytes + (size_t)(2 + wc) * kPanelVecs * 16) + (uint32_t)lane * 16u;
      const uint32_t a_next = lds_addr_w(smem + (size_t)(buf ^ 1) * kBufBytes + (size_t)wr * kPanelVecs * 16) + (uint32_t)lane * 16u;
      const uint32_t b_next = lds_addr_w(smem + (size_t)(buf ^ 1) * kBufBytes + (size_t)(2 + wc) * kPanelVecs * 16) + (uint32_t)lane * 16u;
      if (p == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // panel 0 and the thresholds of the pair have landed for everyone
        In = s_next[0];
        Jn = s_next[1];
        AS_FRAGS(0, 0, a_addr, b_addr)
      }
      const bool last = p + 1 == np;
      const bool more = !last || In != 0xFFFFFFFFu;
      const uint32_t Id = last ? In : I, Jd = last ? Jn : J, pd = last ? 0u : p + 1;
      if (p == 0 && kSweep) {
        static_for_w<0, 16>([&](auto e_c) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[decltype(e_c)::value >> 2][decltype(e_c)::value & 3][r] = 0.f;
        });
      }
      if (p == 0 && !kSweep) {  // the accumulators start at minus the scaled row thresholds
        const uint32_t ra = lds_addr_w(smem + kThrAt + (size_t)slot * kSlotBytes) + (uint32_t)(wr * 128 + 4 * lh) * 4u;
        static_for_w<0, 4>([&](auto sa_c) {
          constexpr int sa = decltype(sa_c)::value;
          v4f32 t[4];
          lds_read16f_w<(sa * 32 + 0) * 4>(t[0], ra);
          lds_read16f_w<(sa * 32 + 8) * 4>(t[1], ra);
          lds_read16f_w<(sa * 32 + 16) * 4>(t[2], ra);
          lds_read16f_w<(sa * 32 + 24) * 4>(t[3], ra);
          lds_wait_all_w();
          v16f32 nth;
#pragma unroll
          for (int r = 0; r < 16; ++r) nth[r] = -thr_scaled_w(t[r >> 2][r & 3]);
          acc[sa][0] = nth;
          acc[sa][1] = nth;
          acc[sa][2] = nth;
          acc[sa][3] = nth;
        });
      }
      // The set of k step 0 has been requested (behind the barrier that made the panel visible).  Every k step requests
      // the next one's fragments before its own MFMAs; the last one instead waits for the next panel's DMA and meets the
      // barrier while its 16 MFMAs are still in the pipe, and requests the first fragments of the next panel behind it:
      // at that point every wave holds all its fragments of this panel in registers, so its buffer may be overwritten.
      static_for_w<0, kPK>([&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value, cur = ks & 1;
        if (more && !(dbg & 4)) dma_panel(Id, Jd, pd, buf ^ 1, ks);
        if constexpr (ks + 1 < kPK) {
          if constexpr (cur == 0) AS_FRAGS(1, ks + 1, a_addr, b_addr) else AS_FRAGS(0, ks + 1, a_addr, b_addr)
          lds_wait8_w<8>(fa[cur], fb[cur]);
        } else {
          if (last && more) dma_thr(Id, Jd, slot ^ 1);
          lds_wait8_w<0>(fa[cur], fb[cur]);
        }
        static_for_w<0, 16>([&](auto e_c) {
          constexpr int sa = decltype(e_c)::value >> 2, sb = decltype(e_c)::value & 3;
          acc[sa][sb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][sa], fb[cur][sb], acc[sa][sb], 0, 0, 0);
        });
        if constexpr (ks + 1 == kPK) {
          if (!last) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            AS_FRAGS(0, 0, a_next, b_next)
          }
        }
      });
      buf ^= 1;
    }
#undef AS_FRAGS
    // ---- the pair is contracted: coarse test per 32 x 32 block, element-wise test and listing for the few that pass ----
    n_blk += 16;
    if (kSweep) {
      // MODE 2 with the grouped sample (grp_start: first image position of every pivot group; grp_supers: sample super tiles per
      // group): a row takes the maxima of ITS group's sample alone -- the samples of neighbouring groups hold largely the same rows at
      // other slots (other bins) where the groups lie along a chain of similar pivots, and the union counted them twice
      uint32_t g_lo_pos = 0u, g_hi_pos = 0xFFFFFFFFu;
      if (MODE == 2 && grp_start && grp_supers) {
        const uint32_t g = J / grp_supers;
        g_lo_pos = grp_start[g];
        g_hi_pos = grp_start[g + 1];
      }
      float lx[4] = {0.f, 0.f, 0.f, 0.f};  // MODE 3: lambdas of the lane's columns
      const uint32_t la = lds_addr_w(smem + kThrAt + (size_t)slot * kSlotBytes + 2048) + (uint32_t)(wr * 128 + 4 * lh) * 4u;
      if (MODE == 3) {
        const uint32_t ca = lds_addr_w(smem + kThrAt + (size_t)slot * kSlotBytes + 1024) + (uint32_t)(wc * 128 + lc) * 4u;
        lds_read4f_w(lx[0], ca);
        lds_read4f_w(lx[1], ca + 32 * 4);
        lds_read4f_w(lx[2], ca + 64 * 4);
        lds_read4f_w(lx[3], ca + 96 * 4);
        lds_wait_all_w();
      }
      static_for_w<0, 4>([&](auto sa_c) {
        constexpr int sa = decltype(sa_c)::value;
        float lq[16];  // MODE 3: lambdas of the lane's rows of this block row
        if (MODE == 3) {
          v4f32 t[4];
          lds_read16f_w<(sa * 32 + 0) * 4>(t[0], la);
          lds_read16f_w<(sa * 32 + 8) * 4>(t[1], la);
          lds_read16f_w<(sa * 32 + 16) * 4>(t[2], la);
          lds_read16f_w<(sa * 32 + 24) * 4>(t[3], la);
          lds_wait_all_w();
#pragma unroll
          for (int r = 0; r < 16; ++r) lq[r] = t[r >> 2][r & 3];
        }
        static_for_w<0, 4>([&](auto sb_c) {
          constexpr int sb = decltype(sb_c)::value;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float val = acc[sa][sb][r];
            if (MODE == 3) {  // the approximate score: alpha s + beta (1 - min(|lambda_q - lambda_x|, 1))
              const float tt = __builtin_fminf(__builtin_fabsf(lq[r] - lx[sb]), 1.f);
              val = __builtin_fmaf(-beta, tt, __builtin_fmaf(alpha * (1.f / kProductScaleW), val, beta));
            }
            if (MODE == 2) {
              const uint32_t pos = I * (uint32_t)kWS + (uint32_t)(wr * 128 + sa * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh);
              val = (pos >= g_lo_pos && pos < g_hi_pos) ? val : -__builtin_inff();
            }
            rmax[kSweep ? sa : 0][r] = __builtin_fmaxf(rmax[kSweep ? sa : 0][r], val);
          }
        });
      });
      if (In != I) {  // the sweep of this row super tile is over: its maxima go out (32 consecutive floats per row and half wave)
        static_for_w<0, 4>([&](auto sa_c) {
          constexpr int sa = decltype(sa_c)::value;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const uint32_t row = (I - I_begin) * kWS + wr * 128 + sa * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
            maxima[(size_t)row * 64 + wc * 32 + lc] = rmax[kSweep ? sa : 0][r] * (MODE == 3 ? 1.f : 1.f / kProductScaleW);
            rmax[kSweep ? sa : 0][r] = -__builtin_inff();
          }
        });
      }
    } else if (!(dbg & 128)) {
      const bool diag = MODE == 0 && I == J;
      const uint32_t ra = lds_addr_w(smem + kThrAt + (size_t)slot * kSlotBytes) + (uint32_t)(wr * 128 + 4 * lh) * 4u;
      const uint32_t ca = lds_addr_w(smem + kThrAt + (size_t)slot * kSlotBytes + 1024) + (uint32_t)(wc * 128 + lc) * 4u;
      float tcs[4] = {0.f, 0.f, 0.f, 0.f};
      if (MODE == 0) {
        lds_read4f_w(tcs[0], ca);
        lds_read4f_w(tcs[1], ca + 32 * 4);
        lds_read4f_w(tcs[2], ca + 64 * 4);
        lds_read4f_w(tcs[3], ca + 96 * 4);
        lds_wait_all_w();
      }
      // one copy of the listing code: a rolled loop over the 16 blocks (unrolled, its rare path costs hundreds of
      // registers in live ranges); the block's accumulator is picked by a wave-uniform switch
      v16f32 nth;
      float tmax = -__builtin_inff();
#pragma nounroll
      for (int e = 0; e < 16; ++e) {
        const int sa = e >> 2, sb = e & 3;
        if (sb == 0) {  // thresholds of the 32 rows of this block row
          v4f32 t[4];
          const uint32_t rb = ra + (uint32_t)sa * 128u;
          lds_read16f_w<0>(t[0], rb);
          lds_read16f_w<32>(t[1], rb);
          lds_read16f_w<64>(t[2], rb);
          lds_read16f_w<96>(t[3], rb);
          lds_wait_all_w();
          tmax = -__builtin_inff();
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float ts = thr_scaled_w(t[r >> 2][r & 3]);
            nth[r] = -ts;
            if (ts < __builtin_inff()) tmax = __builtin_fmaxf(tmax, ts);
          }
          tmax = __builtin_fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        }
        v16f32 x;
        switch (e) {
#define AS_PICK(E) case E: x = acc[(E) >> 2][(E)&3]; break;
          AS_PICK(0) AS_PICK(1) AS_PICK(2) AS_PICK(3) AS_PICK(4) AS_PICK(5) AS_PICK(6) AS_PICK(7)
          AS_PICK(8) AS_PICK(9) AS_PICK(10) AS_PICK(11) AS_PICK(12) AS_PICK(13) AS_PICK(14)
          default: x = acc[3][3]; break;
#undef AS_PICK
        }
        const uint32_t row_group0 = I * kWS + wr * 128 + sa * 32;
        const uint32_t colbase = J * kWS + wc * 128 + sb * 32;
        const float tcr = sb == 0 ? tcs[0] : (sb == 1 ? tcs[1] : (sb == 2 ? tcs[2] : tcs[3]));
        const float tc = (MODE != 0 || diag || colbase + lc >= n) ? __builtin_inff() : thr_scaled_w(tcr);
        const bool h = max16_w(x) >= __builtin_fminf(0.f, tc - tmax);
        if (__any(h) && !(dbg & 2)) {
          // exact per-lane tests, then the lanes that hold a pair go through the wave's LDS slots four at a time
          // (the listing of mfma_join_f16.hip)
          v16f32 v;
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = x[r] - nth[r];
          const bool hit = max16_w(x) >= 0.f || max16_w(v) >= tc;
          unsigned long long m = __ballot(hit);
          n_pass += m ? 1 : 0;
          // (LDS accesses by hand: the first panel of the next pair is in flight by LDS-DMA, and the compiler would wait
          // it out in front of every LDS access it generates itself)
          const uint32_t ex_addr = lds_addr_w(ex), q_addr = lds_addr_w(queue), qs_addr = lds_addr_w(queue_s);
          while (m) {
            const uint32_t rank = lanes_below(m);
            if (hit && ((m >> lane) & 1ull) && rank < 4) {
              const uint32_t sl = ex_addr + rank * (kSlotFloatsW * 4);
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                v4f32 w0, w1;
                w0[0] = x[4 * q]; w0[1] = x[4 * q + 1]; w0[2] = x[4 * q + 2]; w0[3] = x[4 * q + 3];
                w1[0] = v[4 * q]; w1[1] = v[4 * q + 1]; w1[2] = v[4 * q + 2]; w1[3] = v[4 * q + 3];
                lds_write16f_w(sl + q * 16, w0);
                lds_write16f_w(sl + 64 + q * 16, w1);
              }
              lds_write4f_w(sl + 32 * 4, __uint_as_float((uint32_t)lane));
              lds_write4f_w(sl + 33 * 4, tc);
            }
            const uint32_t filled = min(4u, (uint32_t)__builtin_popcountll(m));
            const uint32_t sl_i = (uint32_t)lane >> 4, r = (uint32_t)lane & 15u;
            const uint32_t sl = ex_addr + sl_i * (kSlotFloatsW * 4);
            const bool valid = sl_i < filled;
            float ap, vv, tcx, srcf;
            lds_read4f_w(ap, sl + r * 4);
            lds_read4f_w(vv, sl + (16 + r) * 4);
            lds_read4f_w(tcx, sl + 33 * 4);
            lds_read4f_w(srcf, sl + 32 * 4);
            lds_wait_all_w();
            const uint32_t src = __float_as_uint(srcf);
            const uint32_t row = row_group0 + 4 * (src >> 5) + (r & 3) + 8 * (r >> 2), col = colbase + (src & 31);
            bool inside = valid && col < n && row < n_rows;
            if (MODE == 1)  // lists by position, self by global id
              inside = inside && a_ids[row < n_rows ? row : 0] != (perm ? perm[col < n ? col : 0] : col);
            else if (MODE == 0) inside = inside && row != col;
            float sc = vv * (1.f / kProductScaleW);
            bool w1 = inside && ap >= 0.f;
            if (MODE == 4) {  // alpha s - min(|beta lambda_q - beta lambda_x|, beta) >= g_q - beta
              const uint32_t lbase = lds_addr_w(smem + kThrAt + (size_t)slot * kSlotBytes);
              const uint32_t rl = (uint32_t)(wr * 128 + sa * 32) + 4 * (src >> 5) + (r & 3) + 8 * (r >> 2);
              const uint32_t cl = (uint32_t)(wc * 128 + sb * 32) + (src & 31);
              float lqv, lxv, gq;
              lds_read4f_w(lqv, lbase + 2048 + rl * 4);
              lds_read4f_w(lxv, lbase + 1024 + cl * 4);
              lds_read4f_w(gq, lbase + 3072 + rl * 4);
              lds_wait_all_w();
              sc = __builtin_fmaf(alpha, sc, -__builtin_fminf(__builtin_fabsf(__builtin_fmaf(beta, lqv, -beta * lxv)), beta));
              w1 = inside && sc >= gq;
            }
            const bool w2 = MODE == 0 && inside && vv >= tcx;
            const unsigned long long m1 = __ballot(w1), m2 = __ballot(w2);
            const uint32_t n1 = (uint32_t)__builtin_popcountll(m1), n2 = (uint32_t)__builtin_popcountll(m2);
            if (qn + n1 + n2 > kQueueW) flush_queue();
            if (w1) {
              const uint32_t at = qn + lanes_below(m1);
              lds_write8_w(q_addr + at * 8, make_uint2(row, col));
              lds_write4f_w(qs_addr + at * 4, sc);
            }
            if (w2) {
              const uint32_t at = qn + n1 + lanes_below(m2);
              lds_write8_w(q_addr + at * 8, make_uint2(col, row));
              lds_write4f_w(qs_addr + at * 4, sc);
            }
            qn += n1 + n2;
            m &= m - 1;  // the four lowest lanes are done
            m &= m - 1;
            m &= m - 1;
            m &= m - 1;
          }
        }
      }
    }
    if (In == 0xFFFFFFFFu) break;
    __syncthreads();  // s_next has been read by everyone before thread 0 fetches again
    I = In;
    J = Jn;
    slot ^= 1;
  }
  flush_queue();
  if (lane == 0) {
    atomicAdd(stamps + 5, n_blk);
    atomicAdd(stamps + 7, n_pass);
  }
}

}  // namespace

// k steps (16 wide) per row of the f16 image: rows wider than 128 are padded to whole 64-wide panels
uint32_t f16_ksteps(uint32_t d) {
  const uint32_t ks = (d + 15) / 16;
  return ks <= 8 ? ks : (ks + kPK - 1) / kPK * kPK;
}

// m-th (and m2-th) largest of the 64 sub-sample maxima of every row (mfma_theta_bf16.hip: what theta means)
__global__ void theta_select_wide_kernel(const float *__restrict__ maxima, uint32_t row_count, uint32_t m, uint32_t m2,
                                         float *__restrict__ theta_out, float *__restrict__ theta2_out) {
  const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= row_count) return;
  float vals[64];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const float4 q = reinterpret_cast<const float4 *>(maxima + (size_t)row * 64)[c];
    vals[4 * c] = q.x;
    vals[4 * c + 1] = q.y;
    vals[4 * c + 2] = q.z;
    vals[4 * c + 3] = q.w;
  }
  float t = -__builtin_inff(), t1 = -__builtin_inff();
  const uint32_t mm = theta2_out ? max(m, m2) : m;
  for (uint32_t k = 0; k < mm; ++k) {  // remove the current maximum mm - 1 times
    float best = -__builtin_inff();
    int at = 0;
#pragma unroll
    for (int c = 0; c < 64; ++c) {
      const bool gt = vals[c] > best;
      best = gt ? vals[c] : best;
      at = gt ? c : at;
    }
    t = best;
    if (k + 1 == m) t1 = best;
#pragma unroll
    for (int c = 0; c < 64; ++c) vals[c] = c == at ? -__builtin_inff() : vals[c];
  }
  if (theta2_out) theta2_out[row] = t;
  theta_out[row] = t1;
}

template <int MODE>
static int launch_wide(as_ctx *ctx, const char *prof_name, const void *P, const void *PA, const uint32_t *a_ids, uint32_t KS,
                       uint32_t T2, uint32_t I_begin, uint32_t I_count, uint32_t n_rows, uint32_t n, const float *theta,
                       uint32_t cap, uint32_t *cnt, uint32_t *cand, float *maxima, const float *col_lam = nullptr,
                       const float *row_lam = nullptr, const float *row_g = nullptr, float alpha = 0.f, float beta = 0.f,
                       const uint32_t *perm = nullptr, const unsigned char *mask = nullptr, uint32_t mask_pitch = 0,
                       const uint32_t *grp_start = nullptr, uint32_t grp_supers = 0) {
  DevBuf<unsigned long long> stamps;
  DevBuf<uint32_t> counters;
  AS_TRY(stamps.alloc(ctx, 8));
  AS_TRY(counters.alloc(ctx, 8));
  AS_HIP(ctx, hipMemsetAsync(stamps.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
  AS_HIP(ctx, hipMemsetAsync(counters.p, 0, 8 * sizeof(uint32_t), ctx->stream));
  DevBuf<uint2> unit_list;
  uint32_t n_list = 0;
  if ((MODE == 0 || MODE == 4 || MODE == 1) && mask && ctx->opt_join_unit_list) {  // the wanted pairs, listed (wide_unit_list_kernel)
    const int one_sided = MODE != 0;
    const uint32_t n_norm = (T2 - 1) / 2, l_max = one_sided ? T2 : 1 + n_norm + 1, n_ib = (I_count + 3) / 4, n_ob = (l_max + 7) / 8, n_sb = n_ib * n_ob;
    DevBuf<uint32_t> counts;
    DevBuf<uint64_t> offsets;
    AS_TRY(counts.alloc(ctx, n_sb));
    AS_TRY(offsets.alloc(ctx, (size_t)n_sb + 1));
    const unsigned grid = std::min<uint32_t>(cdiv(n_sb, 8), 256u * 8u);
    hipLaunchKernelGGL(wide_unit_list_kernel, dim3(grid), dim3(256), 0, ctx->stream, T2, I_begin, I_count, n_sb, n_ob, (const unsigned long long *)mask,
                       mask_pitch, counts.p, (const uint64_t *)nullptr, (uint2 *)nullptr, one_sided);
    AS_HIP(ctx, hipGetLastError());
    AS_TRY(exclusive_scan_u32_u64(ctx, counts.p, n_sb, offsets.p));
    uint64_t total = 0;
    AS_HIP(ctx, copy_d2h(ctx, &total, offsets.p + n_sb, sizeof(total)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (total && total < 0xFFFFFFFFull) {
      AS_TRY(unit_list.alloc(ctx, total));
      hipLaunchKernelGGL(wide_unit_list_kernel, dim3(grid), dim3(256), 0, ctx->stream, T2, I_begin, I_count, n_sb, n_ob, (const unsigned long long *)mask,
                         mask_pitch, (uint32_t *)nullptr, (const uint64_t *)offsets.p, unit_list.p, one_sided);
      AS_HIP(ctx, hipGetLastError());
      n_list = (uint32_t)total;
    }
    ctx->stats[std::string(prof_name) + "_pairs_listed"] = (int64_t)total;
    if (total == 0) return AS_OK;  // (nothing the first pass left: no pair to contract)
  }
  {
    ProfScope ps(ctx, prof_name);
    AS_HIP(ctx, hipFuncSetAttribute((const void *)pair_join_f16_wide_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)kWideLds));
    hipLaunchKernelGGL((pair_join_f16_wide_kernel<MODE>), dim3(256), dim3(256), kWideLds, ctx->stream, (const f16x8 *)P,
                       (const f16x8 *)PA, a_ids, KS, T2, I_begin, I_count, n_rows, n, theta, cap, cnt, cand, maxima, counters.p,
                       (int)ctx->opt_debug, stamps.p, col_lam, row_lam, row_g, alpha, beta, perm, (const unsigned long long *)mask,
                       mask_pitch, grp_start, grp_supers, (const uint2 *)unit_list.p, n_list);
    AS_HIP(ctx, hipGetLastError());
  }
  unsigned long long h[8];
  AS_HIP(ctx, copy_d2h(ctx, h, stamps.p, sizeof(h)));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stats[std::string(prof_name) + "_blocks"] = (int64_t)h[5];         // 32x32 blocks contracted (KS MFMAs each)
  ctx->stats[std::string(prof_name) + "_blocks_listed"] = (int64_t)h[7];  // ... that listed at least one pair
  return AS_OK;
}

// The wide form of join_f16 (same contract; I_begin / I_count in 512-row super tiles as there): 128 < dim.
int join_f16_wide(as_ctx *ctx, const void *P, uint32_t n, uint32_t d, uint32_t I_begin, uint32_t I_count, const float *theta,
                  uint32_t cap, uint32_t *cnt, uint32_t *cand, const char *prof_name, const uint32_t *perm, const unsigned char *mask) {
  const uint32_t KS = f16_ksteps(d);
  if (KS <= 8 || KS % kPK != 0) return fail(ctx, AS_ERR_UNSUPPORTED, "wide f16 join: dim not covered");
  const uint32_t T2 = (uint32_t)(((uint64_t)n + 511) / 512) * 2;  // 256-row super tiles of the image (whole 512-row ones)
  return launch_wide<0>(ctx, prof_name, P, nullptr, nullptr, KS, T2, 2 * I_begin, std::min(2 * I_count, T2 - 2 * I_begin), n, n, theta,
                        cap, cnt, cand, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, perm, mask, T2);
}

// One-sided pass: the nA gathered rows (image PA: whole 256-row super tiles, zero padded; a_ids their global ids; theta_a
// their thresholds by position, +inf on padding) against all n rows of P.  cnt / cand (by position; cnt zeroed by the caller).
// mask (optional): what prefilter_i8_rect found for these rows (a word per 512-row super tile of the list and 256-column stage).
int join_f16_wide_rect(as_ctx *ctx, const void *P, uint32_t n, uint32_t d, const void *PA, uint32_t nA, const uint32_t *a_ids,
                       const float *theta_a, uint32_t cap, uint32_t *cnt, uint32_t *cand, const char *prof_name, const uint32_t *perm,
                       const unsigned char *mask) {
  const uint32_t KS = f16_ksteps(d);
  if (KS <= 8 || KS % kPK != 0) return fail(ctx, AS_ERR_UNSUPPORTED, "wide f16 join: dim not covered");
  const uint32_t T2 = (uint32_t)(((uint64_t)n + 511) / 512) * 2;
  return launch_wide<1>(ctx, prof_name, P, PA, a_ids, KS, T2, 0, (nA + kWS - 1) / kWS, nA, n, theta_a, cap, cnt, cand, nullptr, nullptr,
                        nullptr, nullptr, 0.f, 0.f, perm, mask, T2);
}

int pack_f16(as_ctx *ctx, const double *R, const double *norms, const uint32_t *row_ids, uint64_t row0, uint64_t n, uint32_t d,
             void *P, uint32_t tiles);
uint64_t packed_vecs_f16(uint32_t tiles, uint32_t d);
int pack_f16_from_image(as_ctx *ctx, const void *row_image, uint64_t n_src, const uint32_t *row_ids, uint64_t n, uint32_t d, void *P, uint32_t tiles);

// Thresholds of rows [row_begin, row_begin + row_count) (row_begin a multiple of 256) against the sample (ns ids, ns a
// multiple of 256) for rows wider than 128: the contract of join_thresholds_f16 (mfma_theta_f16.hip).
int join_thresholds_f16_wide(as_ctx *ctx, const double *rows, const double *norms, uint32_t d, const void *P_all, uint32_t row_begin,
                             uint32_t row_count, const uint32_t *sample_ids, uint32_t ns, uint32_t m, float *theta_out, uint32_t m2,
                             float *theta2_out) {
  const uint32_t KS = f16_ksteps(d);
  if (KS <= 8 || KS % kPK != 0) return fail(ctx, AS_ERR_UNSUPPORTED, "wide f16 threshold pass: dim not covered");
  if (m < 1 || m > 32 || m2 > 48 || ns < 256 || ns % 256 != 0 || row_begin % 256 != 0)
    return fail(ctx, AS_ERR_UNSUPPORTED, "wide f16 threshold pass: the sample must be a multiple of 256 rows");
  const uint32_t s_tiles = ns / kTileRows, row_supers = (row_count + kWS - 1) / kWS;
  DevBuf<unsigned char> PS;
  DevBuf<float> maxima;
  AS_TRY(PS.alloc(ctx, packed_vecs_f16(s_tiles, d) * 16));
  AS_TRY(pack_f16(ctx, rows, norms, sample_ids, 0, ns, d, PS.p, s_tiles));
  AS_TRY(maxima.alloc(ctx, (size_t)row_supers * kWS * 64));
  AS_TRY(launch_wide<2>(ctx, "knn_join_thresholds", P_all, PS.p, nullptr, KS, row_supers, row_begin / kWS, ns / kWS, row_count, ns,
                        nullptr, 0, nullptr, nullptr, maxima.p));
  hipLaunchKernelGGL(theta_select_wide_kernel, dim3(cdiv(row_count, 64)), dim3(64), 0, ctx->stream, maxima.p, row_count, m, m2, theta_out,
                     theta2_out);
  AS_HIP(ctx, hipGetLastError());
  // (no synchronisation to hand the workspace back: a block returned to the cache is reused by later work on this same stream only)
  return AS_OK;
}

// The same thresholds from the GROUPED sample (mfma_theta_f16.hip: with the rows ordered by 1 024 pivots of their rotated
// coordinates, a row super tile meets only the sampled rows of the groups nearest to its own: 2 - 4 sample super tiles instead of
// n / 32 / 256 of them).  P_all: the image of all rows in that order; thresholds come out by image position.
uint32_t grouped_sample_rows_per_group(uint32_t n, uint32_t np);
int grouped_sample_plan(as_ctx *ctx, const uint32_t *perm, const uint32_t *group_start, uint32_t np, const float *piv, uint32_t piv_pitch,
                        uint32_t piv_dim, uint32_t n, uint32_t *ids, uint32_t *unit_ranges, const uint32_t *near_pre);
int join_thresholds_f16_wide_grouped(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, const void *P_all,
                                     const uint32_t *perm, const uint32_t *group_start, uint32_t np, const float *piv, uint32_t piv_pitch,
                                     uint32_t piv_dim, uint32_t m, float *theta_out, uint32_t m2, float *theta2_out, const uint32_t *near_pre) {
  const uint32_t KS = f16_ksteps(d);
  if (KS <= 8 || KS % kPK != 0) return fail(ctx, AS_ERR_UNSUPPORTED, "wide f16 threshold pass: dim not covered");
  if (m < 1 || m > 32 || m2 > 48 || np == 0 || np > 4096) return fail(ctx, AS_ERR_UNSUPPORTED, "wide f16 threshold pass: parameters not covered");
  const uint32_t per = grouped_sample_rows_per_group(n, np), ns = np * per, row_supers = (n + kWS - 1) / kWS;
  if (per % kWS != 0) return fail(ctx, AS_ERR_UNSUPPORTED, "wide f16 threshold pass: group sample size");
  DevBuf<uint32_t> ids, ranges;
  DevBuf<unsigned char> PS;
  DevBuf<float> maxima;
  AS_TRY(ids.alloc(ctx, ns));
  AS_TRY(ranges.alloc(ctx, (size_t)2 * row_supers));
  // (the sampled rows as image POSITIONS, gathered from P_all itself -- 2 d bytes a row and no conversion, where the f64 rows were
  // 8 d bytes and the division again: half a million sampled rows of 768 columns, 3 GB, at 1 M x 768; mfma_theta_f16.hip does the same)
  AS_TRY(grouped_sample_plan(ctx, perm ? nullptr : perm, group_start, np, piv, piv_pitch, piv_dim, n, ids.p, ranges.p, near_pre));
  AS_TRY(PS.alloc(ctx, packed_vecs_f16(ns / kTileRows, d) * 16));
  if (perm) AS_TRY(pack_f16_from_image(ctx, P_all, n, ids.p, ns, d, PS.p, ns / kTileRows));
  else AS_TRY(pack_f16(ctx, rows, norms, ids.p, 0, ns, d, PS.p, ns / kTileRows));
  AS_TRY(maxima.alloc(ctx, (size_t)row_supers * kWS * 64));
  AS_TRY(launch_wide<2>(ctx, "knn_join_thresholds", P_all, PS.p, ranges.p, KS, row_supers, 0, ns / kWS, n, ns, nullptr, 0, nullptr, nullptr,
                        maxima.p, nullptr, nullptr, nullptr, 0.f, 0.f, nullptr, nullptr, 0, group_start, per / kWS));
  hipLaunchKernelGGL(theta_select_wide_kernel, dim3(cdiv(n, 64)), dim3(64), 0, ctx->stream, maxima.p, n, m, m2, theta_out, theta2_out);
  AS_HIP(ctx, hipGetLastError());
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return AS_OK;
}

// Thresholds from a row's OWN NEIGHBOURHOOD in the ordered image for rows wider than 128 (mfma_theta_f16.hip,
// join_thresholds_f16_band: what the value certifies): the sweep kernel with the image as its own sample, every row super tile
// against the 512-position super tiles `band` on either side of its own.  theta_out[p] = the m-th largest of the row's 64 bin maxima.
namespace {
__global__ void band_ranges_wide_kernel(uint32_t row_supers, uint32_t t_img, uint32_t band, uint32_t *__restrict__ ranges) {
  const uint32_t I = blockIdx.x * blockDim.x + threadIdx.x;
  if (I >= row_supers) return;
  const uint32_t I5 = I >> 1;
  ranges[2 * I] = 2u * (I5 > band ? I5 - band : 0u);
  ranges[2 * I + 1] = min(t_img, 2u * (I5 + band) + 2u);
}
}  // namespace
int join_thresholds_f16_wide_band(as_ctx *ctx, uint32_t n, uint32_t d, const void *P_all, uint32_t band, uint32_t m, float *theta_out) {
  const uint32_t KS = f16_ksteps(d);
  if (KS <= 8 || KS % kPK != 0) return fail(ctx, AS_ERR_UNSUPPORTED, "wide f16 threshold pass: dim not covered");
  if (m < 1 || m > 32) return fail(ctx, AS_ERR_UNSUPPORTED, "wide f16 threshold pass: parameters not covered");
  const uint32_t row_supers = (n + kWS - 1) / kWS, t_img = (uint32_t)(((uint64_t)n + 511) / 512) * 2;
  DevBuf<uint32_t> ranges;
  DevBuf<float> maxima;
  AS_TRY(ranges.alloc(ctx, (size_t)2 * row_supers));
  AS_TRY(maxima.alloc(ctx, (size_t)row_supers * kWS * 64));
  hipLaunchKernelGGL(band_ranges_wide_kernel, dim3(cdiv(row_supers, 256)), dim3(256), 0, ctx->stream, row_supers, t_img, band, ranges.p);
  AS_TRY(launch_wide<2>(ctx, "knn_join_thresholds_band", P_all, P_all, ranges.p, KS, row_supers, 0, t_img, n, n, nullptr, 0, nullptr, nullptr,
                        maxima.p));
  hipLaunchKernelGGL(theta_select_wide_kernel, dim3(cdiv(n, 64)), dim3(64), 0, ctx->stream, maxima.p, n, m, 0u, theta_out, (float *)nullptr);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

// ---- the search on the same kernel (items wider than 128): score guesses and the one-sided scan ----
// PQ: f16 image of the nq queries (whole 256-row super tiles, zero padded).  guesses of every query = the m-th (and m2-th)
// largest approximate score over the sample (ns item ids, a multiple of 256; s_lam their f32 lambdas); q_lam: f32 lambdas
// of the queries, padded to the image.  The contract of search_thresholds_bf16 (mfma_theta_bf16.hip).
int search_thresholds_f16_wide(as_ctx *ctx, const void *PQ, uint32_t nq, uint32_t d, const double *items, const double *norms,
                               const uint32_t *sample_ids, uint32_t ns, uint32_t m, const float *q_lam, const float *s_lam, float alpha,
                               float beta, float *score_out, uint32_t m2, float *score2_out) {
  const uint32_t KS = f16_ksteps(d);
  if (KS <= 8 || KS % kPK != 0) return fail(ctx, AS_ERR_UNSUPPORTED, "wide f16 search thresholds: dim not covered");
  if (m < 1 || m > 32 || m2 > 48 || ns < 256 || ns % 256 != 0) return fail(ctx, AS_ERR_UNSUPPORTED, "wide f16 search thresholds: sample size");
  const uint32_t s_tiles = ns / kTileRows, q_supers = (nq + kWS - 1) / kWS;
  DevBuf<unsigned char> PS;
  DevBuf<float> maxima;
  AS_TRY(PS.alloc(ctx, packed_vecs_f16(s_tiles, d) * 16));
  AS_TRY(pack_f16(ctx, items, norms, sample_ids, 0, ns, d, PS.p, s_tiles));
  AS_TRY(maxima.alloc(ctx, (size_t)q_supers * kWS * 64));
  AS_TRY(launch_wide<3>(ctx, "search_thresholds", PQ, PS.p, nullptr, KS, q_supers, 0, ns / kWS, nq, ns, nullptr, 0, nullptr, nullptr, maxima.p,
                        s_lam, q_lam, nullptr, alpha, beta));
  hipLaunchKernelGGL(theta_select_wide_kernel, dim3(cdiv(nq, 64)), dim3(64), 0, ctx->stream, maxima.p, nq, m, m2, score_out, score2_out);
  AS_HIP(ctx, hipGetLastError());
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return AS_OK;
}

// Lists, per query (position), every item whose approximate score reaches its guess.  P: f16 image of the n items; PQ: of
// the nq queries; cos_floor = (g_q - beta) / alpha and g_minus_beta = g_q - beta by position (padded to the image: +inf);
// q_lam / x_lam: f32 lambdas of the queries / the items (padded).  The contract of join_bf16_search (mfma_join_bf16.hip).
// perm (item image position -> item), q_perm (query image position -> query): the images' orders, where they have one (lists come out
// by query and hold items); mask: what prefilter_i8_rect found (a word per 512-query super tile and 256-item stage), the pairs of
// super tiles none of whose bits is set are not contracted.
int join_f16_wide_search(as_ctx *ctx, const void *P, uint32_t n, uint32_t d, const void *PQ, uint32_t nq, const float *cos_floor,
                         const float *g_minus_beta, const float *q_lam, const float *x_lam, float alpha, float beta, uint32_t cap,
                         uint32_t *cnt, uint32_t *cand, const char *prof_name, const uint32_t *perm, const uint32_t *q_perm,
                         const unsigned char *mask) {
  const uint32_t KS = f16_ksteps(d);
  if (KS <= 8 || KS % kPK != 0) return fail(ctx, AS_ERR_UNSUPPORTED, "wide f16 search: dim not covered");
  const uint32_t T2 = (uint32_t)(((uint64_t)n + 511) / 512) * 2;
  return launch_wide<4>(ctx, prof_name, P, PQ, q_perm, KS, T2, 0, (nq + kWS - 1) / kWS, nq, n, cos_floor, cap, cnt, cand, nullptr, x_lam,
                        q_lam, g_minus_beta, alpha, beta, perm, mask, T2);
}

}  // namespace asx
