// mfma_theta_f16.hip -- candidate thresholds of the symmetric join from ONE f16 plane of the unit rows (the form the
// join itself contracts: mfma_join_f16.hip): the kernel of mfma_theta_bf16.hip (scheme, geometry and what theta means
// are described there) with one MFMA per k step instead of the three of the split-bf16 form.  theta only has to be a
// sensible cut (any threshold is correct), and an f16 plane is good to ~2.5e-4 typically, far inside the spacing of the
// order statistics the cut is taken from; the kernel is power-limited on the matrix pipe, so a third of the MFMAs is a
// third of the time.  The A rows are read straight from the f16 image of ALL rows that the join packs anyway.
#include <hip/hip_fp16.h>

#include <type_traits>

#include "mfma_common.h"
#include "sortnet.h"

namespace asx {

namespace {

typedef float v16f32 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(size_t)(lds_void *)p; }

// LDS fragment reads issued and waited for by hand (see mfma_join_bf16.hip)
template <int OFF>
__device__ __forceinline__ void lds_read16(f16x8 &dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int PENDING>
__device__ __forceinline__ void lds_wait(f16x8 &x) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(PENDING));
}

// vals sorted DESCENDING in place by Batcher's 64-network (543 compare-exchanges of two instructions each; sortnet.h)
template <size_t... Is>
__device__ __forceinline__ void sort64_desc(float (&v)[64], std::index_sequence<Is...>) {
  auto ce = [&](int a, int b) {
    const float hi = __builtin_fmaxf(v[a], v[b]), lo = __builtin_fminf(v[a], v[b]);
    v[a] = hi;
    v[b] = lo;
  };
  (ce(SortNet<64>::kA[Is], SortNet<64>::kB[Is]), ...);
}

__device__ __forceinline__ void pin16(float (&x)[16]) {
  asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]),
               "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
}

// KS = 16-wide k steps per row (d_pad / 16).  PA: image of the rows [row_begin, ...) whose thresholds are wanted
// (a_tiles 128-row tiles, even count); PS: image of the ns sample rows, ns a multiple of 256 (s_tiles even, no
// padded sample columns).  The sweep is the block pipeline of pair_join_bf16_kernel: per staged sample tile a wave
// contracts four 32x32 blocks, the fragment reads run one k step ahead of their MFMAs, the 16 v_max of a block ride
// behind the MFMAs of the next one (two accumulators alternate), the DMA pieces of the next tile are spread over
// the blocks.  The tile loop is unrolled by two so that the parity bin of a tile is static.
//
// SEARCH: the rows are queries, the sample rows items, and what is reduced is the approximate search score
// alpha * cos + beta * (1 - min(|lambda_q - lambda_x|, 1)) (src/core.rs:229-239) instead of the bare product: the
// m-th largest sampled SCORE of a query is the guess for its K-th best score that the threshold scan of the search
// works from (mfma_drivers.hip).  a_lam / s_lam: f32 lambdas of the rows (padded to whole super tiles) and of the
// sample rows.
template <int KS>
__global__ __launch_bounds__(512, 2) void theta_f16_kernel(const f16x8 *__restrict__ PA, uint32_t a_supers,
                                                           uint32_t row_count, const f16x8 *__restrict__ PS,
                                                           uint32_t s_tiles, uint32_t m, float *__restrict__ theta_out,
                                                           uint32_t m2, float *__restrict__ theta2_out,
                                                           uint32_t *__restrict__ unit_counter,
                                                           const uint32_t *__restrict__ unit_groups, uint32_t parts = 1,
                                                           float *__restrict__ bins_out = nullptr,
                                                           const uint32_t *__restrict__ group_start = nullptr) {
  // parts > 1 (plain sample only, bins_out given): a unit is (256 rows, one of `parts` even ranges of the sample's tiles) and
  // leaves its 64 sub-sample maxima per row in bins_out[(row * parts + part) * 64 ..] instead of selecting -- the maxima of the
  // whole sample are the maxima over the parts bin by bin (a tile keeps its parity), so theta_bins_select_kernel returns the very
  // thresholds of one sweep.  For the calls with few rows (the check of the grouped thresholds, the second pass): 16 or 24 units
  // sweeping the whole sample each left most of the chip idle for 0.37 ms.
  // unit_groups (2 per unit: first and last pivot group of its 256 image positions): the sample is not one image for all rows
  // but s_tiles tiles per GROUP (the sampled rows of the groups nearest to it: join_thresholds_f16_grouped), and a unit sweeps
  // those of the groups it covers.  group_start (the groups' first positions): a row takes the maxima of ITS group's sample alone.
  // Where the groups are laid out along a chain of similar pivots (row_order.hip, round 5) two neighbouring groups' samples hold
  // largely the SAME rows at different slots, i.e. in different bins: the union counted a row's best partners twice and the m-th
  // largest bin maximum came out too high for the rows of every unit that straddles a boundary (3.4 % of all rows left open
  // instead of 0.6 %).
  constexpr bool SEARCH = false;
  constexpr int kTileVecs = KS * 4 * 64;  // 16-B vectors of a 128-row tile (one plane)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f16x8 *bufB = reinterpret_cast<f16x8 *>(smem);                               // 2 x tile
  float *s_max = reinterpret_cast<float *>(bufB);  // [8 waves][32 rows][65], aliases the ring once the sweep is over
  __shared__ uint32_t s_unit;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lh = lane >> 5, lc = lane & 31;

  // 128-row sample tile J -> LDS: 4 KS pieces of 1 KB, piece p by wave p mod 8; part < 0: all of a wave's pieces, else
  // those that go with block `part` of the tile before
  const f16x8 *PSc = PS;  // the sample image of the current sweep
  auto stage = [&](uint32_t J, int buf, int part) {
    const f16x8 *src = PSc + (size_t)J * kTileVecs;
    f16x8 *dst = bufB + (size_t)buf * kTileVecs;
#pragma unroll
    for (int i = 0; i < (4 * KS + 7) / 8; ++i) {
      const int p = wv + 8 * i;
      if (p < 4 * KS && (part < 0 || (i & 3) == part))
        __builtin_amdgcn_global_load_lds((gbl_void *)(src + p * 64 + lane), (lds_void *)(dst + p * 64), 16, 0, 0);
    }
  };

  for (;;) {
    __syncthreads();
    if (tid == 0) s_unit = atomicAdd(unit_counter, 1u);
    __syncthreads();
    const uint32_t U = s_unit;
    if (U >= a_supers * parts) break;
    const uint32_t I = U / parts, part = U % parts;
    // this unit's tiles of the sample (whole pairs of tiles; all of them when parts == 1)
    const uint32_t per = ((s_tiles / 2 + parts - 1) / parts) * 2, t_lo = min(s_tiles, part * per), t_hi = min(s_tiles, t_lo + per);
    const uint32_t st = unit_groups ? s_tiles : t_hi - t_lo;

    f16x8 a_hi[KS];
    {
      const int row0 = wv * 32;
      const size_t T = (size_t)2 * I + (row0 >> 7);
      const int s0 = (row0 & 127) >> 5;
      const f16x8 *a_src = PA + T * kTileVecs + (size_t)s0 * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        a_hi[ks] = a_src[(size_t)ks * 256];
      }
    }
    float rmax[2][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rmax[0][r] = rmax[1][r] = -__builtin_inff();
    bool in_group[16];  // does accumulator row r of this lane belong to the group whose sample is being swept?
#pragma unroll
    for (int r = 0; r < 16; ++r) in_group[r] = true;
    auto value = [&](float sdot, int r) -> float { return in_group[r] ? sdot : -__builtin_inff(); };

    f16x8 b_hi[2];
#define AS_READ(SET, KSTEP, NN) lds_read16<((KSTEP)*4 + (NN)) * 1024>(b_hi[SET], b_addr);
    auto contract = [&](uint32_t b_addr, auto nn_c, auto next_c, auto k0_c, auto k1_c, v16f32 &acc) {
      constexpr int nn = decltype(nn_c)::value, next_nn = decltype(next_c)::value;
      constexpr int K0 = decltype(k0_c)::value, K1 = decltype(k1_c)::value;
      constexpr int s0 = (nn * KS) & 1;  // set holding ks = 0 of this block
      v16f32 z;
#pragma unroll
      for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
      for (int ks = K0; ks < K1; ++ks) {
        const int cur = (s0 + ks) & 1, nxt = cur ^ 1;
        bool more = true;
        if (ks + 1 < KS) {
          switch (ks + 1) {  // compile-time offsets
#define AS_CASE(K) case K: if (nxt) AS_READ(1, K, nn) else AS_READ(0, K, nn) break;
            AS_CASE(1) AS_CASE(2) AS_CASE(3) AS_CASE(4) AS_CASE(5) AS_CASE(6) AS_CASE(7)
#undef AS_CASE
          }
        } else if (next_nn >= 0) {
          if (nxt) AS_READ(1, 0, (next_nn < 0 ? 0 : next_nn)) else AS_READ(0, 0, (next_nn < 0 ? 0 : next_nn))
        } else {
          more = false;
        }
        if (more) {
          if (cur) lds_wait<1>(b_hi[1]); else lds_wait<1>(b_hi[0]);
        } else {
          if (cur) lds_wait<0>(b_hi[1]); else lds_wait<0>(b_hi[0]);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi[ks], b_hi[cur], ks == 0 ? z : acc, 0, 0, 0);
      }
    };

    v16f32 accA, accB;
#pragma unroll
    for (int r = 0; r < 16; ++r) accA[r] = accB[r] = -__builtin_inff();  // nothing pending

    const uint32_t g_lo = unit_groups ? unit_groups[2 * I] : 0u, g_hi = unit_groups ? unit_groups[2 * I + 1] : 0u;
    for (uint32_t g = g_lo; g <= g_hi && st > 0; ++g) {
    PSc = unit_groups ? PS + (size_t)g * s_tiles * kTileVecs : PS + (size_t)t_lo * kTileVecs;
    if (group_start && g_lo != g_hi) {
      const uint32_t p_lo = group_start[g], p_hi = group_start[g + 1];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t pos = I * 256u + (uint32_t)wv * 32u + (uint32_t)((r & 3) + 8 * (r >> 2) + 4 * lh);
        in_group[r] = pos >= p_lo && pos < p_hi;
      }
    }
    __syncthreads();  // the selection of the unit before is done with s_max (it aliases the ring); the sweep before with its buffers
    stage(0, 0, -1);
    __syncthreads();
    const uint32_t b_addr0 = lds_addr(bufB) + (uint32_t)lane * 16u;
    for (uint32_t J = 0; J < st; J += 2) {
#define AS_IC(V) std::integral_constant<int, (V)>{}
// block NN of the tile in buffer PAR (= its parity bin); PREV_PAR = bin of the block whose maxima are taken now
#define AS_BLOCK(PAR, NN, NEXT, CUR, PREV, PREV_PAR, MORE, JN)      \
  {                                                                 \
    constexpr int KH = (KS + 1) / 2;                                \
    constexpr int kValu = SEARCH ? 4 : 2;                           \
    const uint32_t b_addr = b_addr0 + (PAR) * kTileVecs * 16;       \
    if ((MORE) && wv < 4) stage(JN, (PAR) ^ 1, NN);                 \
    contract(b_addr, AS_IC(NN), AS_IC(NEXT), AS_IC(0), AS_IC(KH), CUR); \
    _Pragma("unroll") for (int r = 0; r < 16; ++r)                  \
        rmax[PREV_PAR][r] = __builtin_fmaxf(rmax[PREV_PAR][r], value(PREV[r], r)); \
    _Pragma("unroll") for (int i = 0; i < KH; ++i) {                \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            \
      __builtin_amdgcn_sched_group_barrier(0x002, kValu, 0);        \
    }                                                               \
    if ((MORE) && wv >= 4) stage(JN, (PAR) ^ 1, NN);                \
    contract(b_addr, AS_IC(NN), AS_IC(NEXT), AS_IC(KH), AS_IC(KS), CUR); \
    pin16(rmax[PREV_PAR]);  /* the maxima are taken by here: PREV's registers are free for the next block */ \
  }
      {  // even tile J (buffer 0, bin 0); tile J + 1 always exists
        const uint32_t b_addr = b_addr0;
        AS_READ(0, 0, 0)
        AS_BLOCK(0, 0, 1, accA, accB, 1, true, J + 1)
        AS_BLOCK(0, 1, 2, accB, accA, 0, true, J + 1)
        AS_BLOCK(0, 2, 3, accA, accB, 0, true, J + 1)
        AS_BLOCK(0, 3, -1, accB, accA, 0, true, J + 1)
      }
      __syncthreads();  // tile J + 1 has landed; everyone is done with buffer 0
      {  // odd tile J + 1 (buffer 1, bin 1)
        const bool more = J + 2 < st;
        const uint32_t b_addr = b_addr0 + kTileVecs * 16;
        AS_READ(0, 0, 0)
        AS_BLOCK(1, 0, 1, accA, accB, 0, more, J + 2)
        AS_BLOCK(1, 1, 2, accB, accA, 1, more, J + 2)
        AS_BLOCK(1, 2, 3, accA, accB, 1, more, J + 2)
        AS_BLOCK(1, 3, -1, accB, accA, 1, more, J + 2)
      }
#undef AS_BLOCK
#undef AS_IC
      __syncthreads();
    }
#undef AS_READ
    // the last block of the sweep (bin 1)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      rmax[1][r] = __builtin_fmaxf(rmax[1][r], value(accB[r], r));
      accA[r] = accB[r] = -__builtin_inff();  // nothing pending for the next sweep
    }
    }

    // 64 sub-sample maxima per row -> m-th largest.  s_max[wave][row][parity * 32 + lane column]
    float *wm = s_max + (size_t)wv * 32 * 65;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      wm[((r & 3) + 8 * (r >> 2) + 4 * lh) * 65 + lc] = rmax[0][r];
      wm[((r & 3) + 8 * (r >> 2) + 4 * lh) * 65 + 32 + lc] = rmax[1][r];
    }
    wave_sync();
    if (lane < 32) {
      const uint32_t row = I * 256 + wv * 32 + lane;
      float vals[64];
#pragma unroll
      for (int c = 0; c < 64; ++c) vals[c] = wm[lane * 65 + c];
      if (bins_out) {  // (one part of the sample: the selection follows in theta_bins_select_kernel)
        if (row < row_count) {
          float *dst = bins_out + ((size_t)row * parts + part) * 64;
#pragma unroll
          for (int c = 0; c < 64; ++c) dst[c] = vals[c];
        }
      } else {
      float t = -__builtin_inff(), t1 = -__builtin_inff();
      const uint32_t mm = theta2_out ? max(m, m2) : m;  // theta2_out: a second, looser order statistic (m2 >= m)
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
      // the image carries 2^7 per operand: products are 2^14 times the cosines
      if (row < row_count && theta2_out) theta2_out[row] = t * (1.f / 16384.f);
      t = t1;
      if (row < row_count) theta_out[row] = t * (1.f / 16384.f);
      }
    }
    wave_sync();
  }
}

// ---- the same sweep with TWO row groups per wave (round 6) -----------------------------------------------------------------------
// The kernel above gives a wave 32 rows: every 1 KB fragment it reads from LDS feeds ONE MFMA, and the eight waves of a workgroup
// each read the whole 32 KB stage -- 256 B per clock and CU at the MFMA peak, which is all the LDS array delivers: the sweep was
// bound by its fragment reads (matrix pipe busy 0.25).  Here a workgroup is FOUR waves of 64 rows (the same 256-row unit, the
// same thresholds bit for bit: a maximum does not depend on the order it is taken in): a fragment feeds two MFMAs, the stage is
// read four times instead of eight, and two workgroups share a CU (2 x 66.5 KB of LDS, 2 waves per SIMD as before) so that one's
// barrier is the other's time.  Same arguments, same unit geometry (unit_groups, parts, bins_out, group_start) as theta_f16_kernel.
template <int KS>
__global__ __launch_bounds__(256, 2) void theta_f16x2_kernel(const f16x8 *__restrict__ PA, uint32_t a_supers,
                                                             uint32_t row_count, const f16x8 *__restrict__ PS,
                                                             uint32_t s_tiles, uint32_t m, float *__restrict__ theta_out,
                                                             uint32_t m2, float *__restrict__ theta2_out,
                                                             uint32_t *__restrict__ unit_counter,
                                                             const uint32_t *__restrict__ unit_groups, uint32_t parts = 1,
                                                             float *__restrict__ bins_out = nullptr,
                                                             const uint32_t *__restrict__ group_start = nullptr) {
  constexpr int kTileVecs = KS * 4 * 64;  // 16-B vectors of a 128-row tile (one plane)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f16x8 *bufB = reinterpret_cast<f16x8 *>(smem);   // 2 x tile
  float *s_max = reinterpret_cast<float *>(bufB);  // [4 waves][64 rows][65], aliases the ring once the sweep is over
  __shared__ uint32_t s_unit;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lh = lane >> 5, lc = lane & 31;

  // 128-row sample tile J -> LDS: 4 KS pieces of 1 KB, piece p by wave p mod 4 (KS pieces a wave); part < 0: all of a wave's
  // pieces, else those that go with block `part` of the tile before
  const f16x8 *PSc = PS;  // the sample image of the current sweep
  auto stage = [&](uint32_t J, int buf, int part) {
    const f16x8 *src = PSc + (size_t)J * kTileVecs;
    f16x8 *dst = bufB + (size_t)buf * kTileVecs;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int p = wv + 4 * i;
      if (part < 0 || (i & 3) == part)
        __builtin_amdgcn_global_load_lds((gbl_void *)(src + p * 64 + lane), (lds_void *)(dst + p * 64), 16, 0, 0);
    }
  };

  for (;;) {
    __syncthreads();
    if (tid == 0) s_unit = atomicAdd(unit_counter, 1u);
    __syncthreads();
    const uint32_t U = s_unit;
    if (U >= a_supers * parts) break;
    const uint32_t I = U / parts, part = U % parts;
    const uint32_t per = ((s_tiles / 2 + parts - 1) / parts) * 2, t_lo = min(s_tiles, part * per), t_hi = min(s_tiles, t_lo + per);
    const uint32_t st = unit_groups ? s_tiles : t_hi - t_lo;

    // the wave's 64 rows: two groups of 32 (sub-tiles 2 (wv & 1) and + 1 of tile 2 I + wv / 2), A fragments in registers
    f16x8 a0[KS], a1[KS];
    {
      const size_t T = (size_t)2 * I + (wv >> 1);
      const int s0 = 2 * (wv & 1);
      const f16x8 *a_src = PA + T * kTileVecs + (size_t)s0 * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        a0[ks] = a_src[(size_t)ks * 256];
        a1[ks] = a_src[(size_t)ks * 256 + 64];
      }
    }
    float rmax0[2][16], rmax1[2][16];  // [parity bin][accumulator row] of the two row groups
#pragma unroll
    for (int r = 0; r < 16; ++r) rmax0[0][r] = rmax0[1][r] = rmax1[0][r] = rmax1[1][r] = -__builtin_inff();
    uint32_t in0 = 0xFFFFu, in1 = 0xFFFFu;  // bit r: does accumulator row r of the group belong to the pivot group whose sample is swept?
    auto value0 = [&](float sdot, int r) -> float { return ((in0 >> r) & 1u) ? sdot : -__builtin_inff(); };
    auto value1 = [&](float sdot, int r) -> float { return ((in1 >> r) & 1u) ? sdot : -__builtin_inff(); };

    f16x8 b_hi[2];
#define AS_READ(SET, KSTEP, NN) lds_read16<((KSTEP)*4 + (NN)) * 1024>(b_hi[SET], b_addr);
    auto contract = [&](uint32_t b_addr, auto nn_c, auto next_c, auto k0_c, auto k1_c, v16f32 &acc0, v16f32 &acc1) {
      constexpr int nn = decltype(nn_c)::value, next_nn = decltype(next_c)::value;
      constexpr int K0 = decltype(k0_c)::value, K1 = decltype(k1_c)::value;
      constexpr int s0 = (nn * KS) & 1;  // set holding ks = 0 of this block
      v16f32 z;
#pragma unroll
      for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
      for (int ks = K0; ks < K1; ++ks) {
        const int cur = (s0 + ks) & 1, nxt = cur ^ 1;
        bool more = true;
        if (ks + 1 < KS) {
          switch (ks + 1) {  // compile-time offsets
#define AS_CASE(K) case K: if (nxt) AS_READ(1, K, nn) else AS_READ(0, K, nn) break;
            AS_CASE(1) AS_CASE(2) AS_CASE(3) AS_CASE(4) AS_CASE(5) AS_CASE(6) AS_CASE(7)
#undef AS_CASE
          }
        } else if (next_nn >= 0) {
          if (nxt) AS_READ(1, 0, (next_nn < 0 ? 0 : next_nn)) else AS_READ(0, 0, (next_nn < 0 ? 0 : next_nn))
        } else {
          more = false;
        }
        if (more) {
          if (cur) lds_wait<1>(b_hi[1]); else lds_wait<1>(b_hi[0]);
        } else {
          if (cur) lds_wait<0>(b_hi[1]); else lds_wait<0>(b_hi[0]);
        }
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[ks], b_hi[cur], ks == 0 ? z : acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[ks], b_hi[cur], ks == 0 ? z : acc1, 0, 0, 0);
      }
    };

    v16f32 accA0, accA1, accB0, accB1;
#pragma unroll
    for (int r = 0; r < 16; ++r) accA0[r] = accA1[r] = accB0[r] = accB1[r] = -__builtin_inff();  // nothing pending

    const uint32_t g_lo = unit_groups ? unit_groups[2 * I] : 0u, g_hi = unit_groups ? unit_groups[2 * I + 1] : 0u;
    // groups that every row of the unit sweeps alike (no group_start: the band of a row's own neighbourhood) lie side by side in
    // the image: ONE sweep over all their tiles instead of a prologue (two barriers and an exposed stage load) per group -- a group
    // is an even number of tiles, so every tile keeps its parity bin
    const bool merged = unit_groups && !group_start && (s_tiles & 1u) == 0u;
    const uint32_t g_last = merged ? g_lo : g_hi;
    const uint32_t st_g = merged ? (g_hi - g_lo + 1u) * s_tiles : st;
    for (uint32_t g = g_lo; g <= g_last && st_g > 0; ++g) {
    PSc = unit_groups ? PS + (size_t)g * s_tiles * kTileVecs : PS + (size_t)t_lo * kTileVecs;
    if (group_start && g_lo != g_hi) {
      const uint32_t p_lo = group_start[g], p_hi = group_start[g + 1];
      in0 = in1 = 0u;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t pos = I * 256u + (uint32_t)wv * 64u + (uint32_t)((r & 3) + 8 * (r >> 2) + 4 * lh);
        in0 |= (pos >= p_lo && pos < p_hi) ? (1u << r) : 0u;
        in1 |= (pos + 32u >= p_lo && pos + 32u < p_hi) ? (1u << r) : 0u;
      }
    }
    __syncthreads();  // the selection of the unit before is done with s_max (it aliases the ring); the sweep before with its buffers
    stage(0, 0, -1);
    __syncthreads();
    const uint32_t b_addr0 = lds_addr(bufB) + (uint32_t)lane * 16u;
    for (uint32_t J = 0; J < st_g; J += 2) {
#define AS_IC(V) std::integral_constant<int, (V)>{}
// block NN of the tile in buffer PAR (= its parity bin); PREV_PAR = bin of the block whose maxima are taken now
#define AS_BLOCK(PAR, NN, NEXT, CUR0, CUR1, PREV0, PREV1, PREV_PAR, MORE, JN)  \
  {                                                                 \
    constexpr int KH = (KS + 1) / 2;                                \
    const uint32_t b_addr = b_addr0 + (PAR) * kTileVecs * 16;       \
    if ((MORE) && wv < 2) stage(JN, (PAR) ^ 1, NN);                 \
    contract(b_addr, AS_IC(NN), AS_IC(NEXT), AS_IC(0), AS_IC(KH), CUR0, CUR1); \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                \
      rmax0[PREV_PAR][r] = __builtin_fmaxf(rmax0[PREV_PAR][r], value0(PREV0[r], r)); \
      rmax1[PREV_PAR][r] = __builtin_fmaxf(rmax1[PREV_PAR][r], value1(PREV1[r], r)); \
    }                                                               \
    _Pragma("unroll") for (int i = 0; i < 2 * KH; ++i) {            \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            \
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);            \
    }                                                               \
    if ((MORE) && wv >= 2) stage(JN, (PAR) ^ 1, NN);                \
    contract(b_addr, AS_IC(NN), AS_IC(NEXT), AS_IC(KH), AS_IC(KS), CUR0, CUR1); \
    pin16(rmax0[PREV_PAR]);  /* the maxima are taken by here: PREV's registers are free for the next block */ \
    pin16(rmax1[PREV_PAR]);                                         \
  }
      {  // even tile J (buffer 0, bin 0); tile J + 1 always exists
        const uint32_t b_addr = b_addr0;
        AS_READ(0, 0, 0)
        AS_BLOCK(0, 0, 1, accA0, accA1, accB0, accB1, 1, true, J + 1)
        AS_BLOCK(0, 1, 2, accB0, accB1, accA0, accA1, 0, true, J + 1)
        AS_BLOCK(0, 2, 3, accA0, accA1, accB0, accB1, 0, true, J + 1)
        AS_BLOCK(0, 3, -1, accB0, accB1, accA0, accA1, 0, true, J + 1)
      }
      __syncthreads();  // tile J + 1 has landed; everyone is done with buffer 0
      {  // odd tile J + 1 (buffer 1, bin 1)
        const bool more = J + 2 < st_g;
        const uint32_t b_addr = b_addr0 + kTileVecs * 16;
        AS_READ(0, 0, 0)
        AS_BLOCK(1, 0, 1, accA0, accA1, accB0, accB1, 0, more, J + 2)
        AS_BLOCK(1, 1, 2, accB0, accB1, accA0, accA1, 1, more, J + 2)
        AS_BLOCK(1, 2, 3, accA0, accA1, accB0, accB1, 1, more, J + 2)
        AS_BLOCK(1, 3, -1, accB0, accB1, accA0, accA1, 1, more, J + 2)
      }
#undef AS_BLOCK
#undef AS_IC
      __syncthreads();
    }
#undef AS_READ
    // the last block of the sweep (bin 1)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      rmax0[1][r] = __builtin_fmaxf(rmax0[1][r], value0(accB0[r], r));
      rmax1[1][r] = __builtin_fmaxf(rmax1[1][r], value1(accB1[r], r));
      accA0[r] = accA1[r] = accB0[r] = accB1[r] = -__builtin_inff();  // nothing pending for the next sweep
    }
    }

    // 64 sub-sample maxima per row -> m-th largest.  s_max[wave][row of the wave's 64][parity * 32 + lane column]
    float *wm = s_max + (size_t)wv * 64 * 65;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ro = (r & 3) + 8 * (r >> 2) + 4 * lh;
      wm[ro * 65 + lc] = rmax0[0][r];
      wm[ro * 65 + 32 + lc] = rmax0[1][r];
      wm[(32 + ro) * 65 + lc] = rmax1[0][r];
      wm[(32 + ro) * 65 + 32 + lc] = rmax1[1][r];
    }
    wave_sync();
    {
      const uint32_t row = I * 256 + wv * 64 + lane;
      float vals[64];
#pragma unroll
      for (int c = 0; c < 64; ++c) vals[c] = wm[lane * 65 + c];
      if (bins_out) {  // (one part of the sample: the selection follows in theta_bins_select_kernel)
        if (row < row_count) {
          float *dst = bins_out + ((size_t)row * parts + part) * 64;
#pragma unroll
          for (int c = 0; c < 64; ++c) dst[c] = vals[c];
        }
      } else {
        // The m-th (and mm-th) largest of the 64: theta_f16_kernel removes the maximum mm times (64 compares and 64 selects a
        // round: 18 rounds for the band's K + 1 were a third of a unit's time); a sorting network does it in 543 compare-exchanges
        // whatever m.  The same value: the m-th largest of a multiset does not depend on how it is found (the maxima hold no NaN:
        // they start at -inf and only take v_max of products).
        float t = -__builtin_inff(), t1 = -__builtin_inff();
        const uint32_t mm = theta2_out ? max(m, m2) : m;  // theta2_out: a second, looser order statistic (m2 >= m)
        sort64_desc(vals, std::make_index_sequence<SortNet<64>::kCount>{});
#pragma unroll
        for (int c = 0; c < 64; ++c) {
          t1 = (uint32_t)c + 1u == m ? vals[c] : t1;
          t = (uint32_t)c + 1u == mm ? vals[c] : t;
        }
        // the image carries 2^7 per operand: products are 2^14 times the cosines
        if (row < row_count && theta2_out) theta2_out[row] = t * (1.f / 16384.f);
        t = t1;
        if (row < row_count) theta_out[row] = t * (1.f / 16384.f);
      }
    }
    wave_sync();
  }
}


// the thresholds from the bins the parts of the sample left: maxima over the parts, then the selection of theta_f16_kernel
// (a wave a row, a lane a bin -- round 6: a thread a row with its 64 bins in registers read 256 B per lane per part and ran m rounds of
// remove-the-maximum, 40 us for the ~2 000 rows of a check on 35 waves; here the parts arrive as coalesced 256-byte rows, and a bin's
// rank in the descending order -- bins with a larger value, or the same value at a lower index -- is counted against the 64 lanes:
// the m-th largest value is the bin of rank m - 1, whichever way equal values are numbered)
__global__ __launch_bounds__(256) void theta_bins_select_kernel(const float *__restrict__ bins, uint32_t row_count, uint32_t parts, uint32_t m,
                                                                float *__restrict__ theta_out, uint32_t m2, float *__restrict__ theta2_out) {
  const uint32_t lane = threadIdx.x & 63u, row = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (row >= row_count) return;  // (the wave's)
  float v = -__builtin_inff();
  for (uint32_t p = 0; p < parts; ++p) v = __builtin_fmaxf(v, bins[((size_t)row * parts + p) * 64 + lane]);
  uint32_t rank = 0;
  for (uint32_t c = 0; c < 64; ++c) {
    const float o = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), (int)c));
    rank += (o > v || (o == v && c < lane)) ? 1u : 0u;
  }
  const uint32_t mm = theta2_out ? max(m, m2) : m;
  if (rank + 1 == m) theta_out[row] = v * (1.f / 16384.f);
  if (theta2_out && rank + 1 == mm) theta2_out[row] = v * (1.f / 16384.f);
}

// ---- the sample by pivot group (own join with the row order found on the rotated coordinates, row_order.hip) -------------------
// The 4th best product of a row against a 1-in-32 sample of all rows is a product with a row of its own cluster: with the
// image ordered by 1024 pivots, restricting a row's sample to the sampled rows of the 16 pivot groups nearest to its own
// moves the threshold of 0.5 % of the rows by more than 1e-3 and of none by 1e-2 (scripts/attic/explore_theta_window.py) -- and
// any threshold is correct -- for 1.7 % of the products.  So every group gets its own small sample image (the sampled rows
// of its nearest groups, group_tiles() tiles, zero rows beyond), and a unit sweeps the images of the groups it covers.
constexpr uint32_t kGroupNear = 16;
// sample rows kept per group: 128-row tiles, 4 of them where a group is ~1 000 rows (16 near groups x one position in 32 = 500
// rows) and more for larger groups, up to 32 (4 M rows in 1 024 groups: 2 000 rows, half as many again for the groups larger
// than the average -- with 4 tiles only the four nearest groups fitted, the check against the plain sample failed and the
// plain pass cost 116 ms of a 574 ms table)
static uint32_t group_tiles(uint32_t n, uint32_t np, uint32_t rate = 32) {
  const uint64_t near_rows = ((uint64_t)n / std::max(1u, np)) * kGroupNear / rate;  // rows the near groups' samples hold
  if (near_rows <= 192) return 2;  // (groups of fewer than 384 rows: more than 1 024 pivots, row_order.hip)
  if (near_rows <= 512) return 4;
  uint32_t t = (uint32_t)((near_rows * 3 / 2 + 127u) / 128u);
  t = (t + 1u) & ~1u;  // whole 256-row super tiles
  return std::min(rate >= 32 ? 32u : 128u, t);
}

// near[g][0 .. 16): the pivots most similar to pivot g (itself first).  piv: np x pitch floats.
__global__ __launch_bounds__(256) void near_groups_kernel(const float *__restrict__ piv, uint32_t pitch, uint32_t dim, uint32_t np,
                                                          uint32_t *__restrict__ near) {
  __shared__ float s_sim[4096];
  __shared__ float s_best[256];
  __shared__ uint32_t s_at[256];
  const uint32_t g = blockIdx.x;
  for (uint32_t h = threadIdx.x; h < np; h += 256) {
    float a = 0.f;
    for (uint32_t k = 0; k < dim; ++k) a = __builtin_fmaf(piv[(size_t)g * pitch + k], piv[(size_t)h * pitch + k], a);
    s_sim[h] = h == g ? __builtin_inff() : a;
  }
  __syncthreads();
  for (uint32_t k = 0; k < kGroupNear; ++k) {
    float best = -__builtin_inff();
    uint32_t at = 0xFFFFFFFFu;
    for (uint32_t h = threadIdx.x; h < np; h += 256)
      if (s_sim[h] > best) {
        best = s_sim[h];
        at = h;
      }
    s_best[threadIdx.x] = best;
    s_at[threadIdx.x] = at;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o && (s_best[threadIdx.x + o] > s_best[threadIdx.x] ||
                                   (s_best[threadIdx.x + o] == s_best[threadIdx.x] && s_at[threadIdx.x + o] < s_at[threadIdx.x]))) {
        s_best[threadIdx.x] = s_best[threadIdx.x + o];
        s_at[threadIdx.x] = s_at[threadIdx.x + o];
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      near[(size_t)g * kGroupNear + k] = s_at[0];
      if (s_at[0] != 0xFFFFFFFFu) s_sim[s_at[0]] = -__builtin_inff();
    }
    __syncthreads();
  }
}

// ids[g][slot]: the sampled rows (one image position in 32, hashed) of the groups near g, ~0 beyond; one wave per group, the
// slots in the order of the positions (a thread per group walking them alone was 0.55 ms of the 2.1 ms of the thresholds)
__global__ __launch_bounds__(64) void group_sample_ids_kernel(const uint32_t *__restrict__ near, const uint32_t *__restrict__ group_start,
                                                              uint32_t np, const uint32_t *__restrict__ perm, uint32_t *__restrict__ ids,
                                                              uint32_t slots, uint32_t rate = 32) {
  const uint32_t g = blockIdx.x, lane = threadIdx.x;
  uint32_t *out = ids + (size_t)g * slots;
  uint32_t at = 0;  // (wave-uniform)
  for (uint32_t k = 0; k < kGroupNear && at < slots; ++k) {
    const uint32_t h = near[(size_t)g * kGroupNear + k];
    if (h >= np) continue;
    const uint32_t lo = group_start[h], hi = group_start[h + 1];
    for (uint32_t j0 = lo / rate; (uint64_t)j0 * rate < hi && at < slots; j0 += 64) {
      const uint32_t j = j0 + lane;
      bool ok = false;
      uint32_t p = 0;
      if ((uint64_t)j * rate < hi) {
        uint32_t x = j * 0x9E3779B9u;
        x ^= x >> 15;
        x *= 0x85EBCA6Bu;
        x ^= x >> 13;
        p = j * rate + (x & (rate - 1u));  // (rate: a power of two)
        ok = p >= lo && p < hi;
      }
      const unsigned long long m = __ballot(ok);
      const uint32_t rank = (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
      if (ok && at + rank < slots) out[at + rank] = perm ? perm[p] : p;  // (perm null: the image POSITIONS themselves)
      at += (uint32_t)__builtin_popcountll(m);
    }
  }
  for (uint32_t i = min(at, slots) + lane; i < slots; i += 64) out[i] = 0xFFFFFFFFu;
}

// first and last pivot group of the 256 image positions of every unit
__global__ void unit_groups_kernel(const uint32_t *__restrict__ group_start, uint32_t np, uint32_t n, uint32_t units,
                                   uint32_t *__restrict__ out) {
  const uint32_t I = blockIdx.x * blockDim.x + threadIdx.x;
  if (I >= units) return;
  auto group_of = [&](uint32_t p) -> uint32_t {  // the last g with group_start[g] <= p
    uint32_t lo = 0, hi = np;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) / 2;
      if (group_start[mid] <= p) lo = mid;
      else hi = mid;
    }
    return lo;
  };
  const uint32_t p0 = min(I * 256u, n - 1), p1 = min(I * 256u + 255u, n - 1);
  out[2 * I] = group_of(p0);
  out[2 * I + 1] = group_of(p1);
}

}  // namespace

uint64_t packed_vecs_f16(uint32_t tiles, uint32_t d);
int pack_f16(as_ctx *ctx, const double *R, const double *norms, const uint32_t *row_ids, uint64_t row0, uint64_t n, uint32_t d,
             void *P, uint32_t tiles);
int pack_f16_from_image(as_ctx *ctx, const void *row_image, uint64_t n_src, const uint32_t *row_ids, uint64_t n, uint32_t d, void *P, uint32_t tiles);

// theta_out[r], r < row_count, for rows [row_begin, row_begin + row_count) (row_begin a multiple of 256) against the
// sample `sample_ids` (ns global row ids, ns a multiple of 256).  P_all: the f16 image of ALL rows (whole 512-row super
// tiles: pack_f16_all).  dim <= 128; AS_ERR_UNSUPPORTED beyond.
int join_thresholds_f16(as_ctx *ctx, const double *rows, const double *norms, uint32_t d, const void *P_all, uint32_t row_begin,
                        uint32_t row_count, const uint32_t *sample_ids, uint32_t ns, uint32_t m, float *theta_out, uint32_t m2,
                        float *theta2_out, const void *row_image, uint64_t n_image) {
  const uint32_t KS = (d + 15) / 16;
  if (!(KS >= 1 && KS <= 8)) return fail(ctx, AS_ERR_UNSUPPORTED, "f16 threshold pass: dim not covered");
  if (m < 1 || m > 32 || m2 > 48 || ns < 256 || ns % 256 != 0 || row_begin % 256 != 0)
    return fail(ctx, AS_ERR_UNSUPPORTED, "f16 threshold pass: the sample must be a multiple of 256 rows");
  const uint32_t s_tiles = ns / kTileRows;
  const uint32_t a_supers = (row_count + 255) / 256;
  DevBuf<unsigned char> PS;
  DevBuf<uint32_t> unitc;
  AS_TRY(PS.alloc(ctx, packed_vecs_f16(s_tiles, d) * 16));
  // (row_image: the f16 image of all n_image rows in row order, a build sharded by rows -- the sample's rows are copied from it)
  if (row_image) AS_TRY(pack_f16_from_image(ctx, row_image, n_image, sample_ids, ns, d, PS.p, s_tiles));
  else AS_TRY(pack_f16(ctx, rows, norms, sample_ids, 0, ns, d, PS.p, s_tiles));
  AS_TRY(unitc.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(unitc.p, 0, sizeof(uint32_t), ctx->stream));
  const f16x8 *PA = (const f16x8 *)P_all + (size_t)(row_begin / kTileRows) * KS * 4 * 64;
  const size_t lds = std::max((size_t)2 * KS * 4 * 64 * 16, (size_t)8 * 32 * 65 * sizeof(float));
  // few rows (the check of the grouped thresholds, the open rows of the second pass): the sample split over several units per
  // 256 rows, so that there are a few hundred units
  const uint32_t parts = (a_supers >= 64 || !ctx->opt_join_theta_parts) ? 1u : std::max(1u, std::min(std::min(16u, s_tiles / 2), 256u / std::max(1u, a_supers)));
  DevBuf<float> bins;
  if (parts > 1) AS_TRY(bins.alloc(ctx, (size_t)a_supers * 256 * parts * 64));
  const unsigned grid = std::min<uint32_t>(a_supers * parts, 256u);
  ProfScope ps(ctx, "knn_join_thresholds");
#define AS_THETA(K)                                                                                                          \
  case K:                                                                                                                    \
    if (ctx->opt_join_theta_rows64) {                                                                                        \
      AS_HIP(ctx, hipFuncSetAttribute((const void *)theta_f16x2_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      hipLaunchKernelGGL((theta_f16x2_kernel<K>), dim3(std::min<uint32_t>(a_supers * parts, 512u)), dim3(256), lds, ctx->stream, PA, a_supers, row_count, \
                         (const f16x8 *)PS.p, s_tiles, m, theta_out, m2, theta2_out, unitc.p, (const uint32_t *)nullptr,     \
                         parts, parts > 1 ? bins.p : (float *)nullptr);                                                      \
      break;                                                                                                                 \
    }                                                                                                                        \
    AS_HIP(ctx, hipFuncSetAttribute((const void *)theta_f16_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((theta_f16_kernel<K>), dim3(grid), dim3(512), lds, ctx->stream, PA, a_supers, row_count,              \
                       (const f16x8 *)PS.p, s_tiles, m, theta_out, m2, theta2_out, unitc.p, (const uint32_t *)nullptr,       \
                       parts, parts > 1 ? bins.p : (float *)nullptr);                                                        \
    break;
  switch (KS) {
    AS_THETA(1) AS_THETA(2) AS_THETA(3) AS_THETA(4) AS_THETA(5) AS_THETA(6) AS_THETA(7) AS_THETA(8)
    default: return fail(ctx, AS_ERR_UNSUPPORTED, "f16 threshold pass: dim not covered");
  }
#undef AS_THETA
  if (parts > 1)
    hipLaunchKernelGGL(theta_bins_select_kernel, dim3(cdiv(row_count, 4)), dim3(256), 0, ctx->stream, (const float *)bins.p, row_count, parts, m,
                       theta_out, m2, theta2_out);
  AS_HIP(ctx, hipGetLastError());
  // (no synchronisation to hand the workspace back: a block returned to the cache is reused by later work on this same stream only)
  return AS_OK;
}

// The same for ALL rows of an image ordered by pivot groups (perm: image position -> row; group_start: np + 1 positions; piv:
// the np pivots, `piv_dim` coordinates of pitch `piv_pitch`): thresholds by image position against the sampled rows of the 16
// groups nearest to a row's own.
// pos_begin / pos_count (a multiple of 256 / any; 0 / 0 = all): only the thresholds of that range of image positions are computed
// (theta_out / theta2_out stay indexed by position of the WHOLE image) -- a rank of a build sharded by positions computes its own.
// The plan of the grouped sample by itself (rows wider than 128 sweep it with the panelled kernel, mfma_join_f16_wide.hip):
// ids[np x grouped_sample_rows_per_group()] = the sampled rows of the 16 groups nearest to each group (~0 beyond), unit_ranges[2 x units] = for every unit
// of 256 image positions the range [lo, hi) of 256-row sample super tiles of the groups it covers.
uint32_t grouped_sample_rows_per_group(uint32_t n, uint32_t np) { return group_tiles(n, np) * 128u; }
namespace {
__global__ void unit_ranges_kernel(const uint32_t *__restrict__ ug, uint32_t units, uint32_t *__restrict__ out, uint32_t supers_per_group) {
  const uint32_t I = blockIdx.x * blockDim.x + threadIdx.x;
  if (I >= units) return;
  out[2 * I] = ug[2 * I] * supers_per_group;
  out[2 * I + 1] = (ug[2 * I + 1] + 1) * supers_per_group;
}
}  // namespace
// near_pre (optional): the 16 nearest groups of every group, itself first -- what the chain of the row order has found already
// (row_order.hip: at 4 096 pivots the search by itself takes 1.3 ms)
int grouped_sample_plan(as_ctx *ctx, const uint32_t *perm, const uint32_t *group_start, uint32_t np, const float *piv, uint32_t piv_pitch,
                        uint32_t piv_dim, uint32_t n, uint32_t *ids, uint32_t *unit_ranges, const uint32_t *near_pre) {
  const uint32_t kGroupTiles = group_tiles(n, np), kGroupSlots = kGroupTiles * 128u;
  const uint32_t units = (n + 255) / 256;
  DevBuf<uint32_t> near, ug;
  AS_TRY(near.alloc(ctx, (size_t)np * kGroupNear));
  AS_TRY(ug.alloc(ctx, (size_t)2 * units));
  if (!near_pre) hipLaunchKernelGGL(near_groups_kernel, dim3(np), dim3(256), 0, ctx->stream, piv, piv_pitch, piv_dim, np, near.p);
  hipLaunchKernelGGL(group_sample_ids_kernel, dim3(np), dim3(64), 0, ctx->stream, near_pre ? near_pre : (const uint32_t *)near.p, group_start, np, perm, ids,
                     kGroupSlots);
  hipLaunchKernelGGL(unit_groups_kernel, dim3(cdiv(units, 256)), dim3(256), 0, ctx->stream, group_start, np, n, units, ug.p);
  hipLaunchKernelGGL(unit_ranges_kernel, dim3(cdiv(units, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)ug.p, units, unit_ranges,
                     kGroupTiles / 2);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

int join_thresholds_f16_grouped(as_ctx *ctx, const double *rows, const double *norms, uint32_t n, uint32_t d, const void *P_all,
                                const uint32_t *perm, const uint32_t *group_start, uint32_t np, const float *piv, uint32_t piv_pitch,
                                uint32_t piv_dim, uint32_t m, float *theta_out, uint32_t m2, float *theta2_out, uint32_t pos_begin,
                                uint32_t pos_count, const void *row_image, const uint32_t *near_pre) {
  const uint32_t KS = (d + 15) / 16;
  if (!(KS >= 1 && KS <= 8)) return fail(ctx, AS_ERR_UNSUPPORTED, "f16 threshold pass: dim not covered");
  if (m < 1 || m > 32 || m2 > 48 || np == 0 || np > 4096) return fail(ctx, AS_ERR_UNSUPPORTED, "f16 threshold pass: parameters not covered");
  const uint32_t a_supers = (n + 255) / 256;
  const uint32_t rate = (uint32_t)std::max<int64_t>(2, std::min<int64_t>(ctx->opt_join_theta_rate, 32));  // (a power of two: api.hip)
  const uint32_t kGroupTiles = group_tiles(n, np, rate), kGroupSlots = kGroupTiles * 128u;
  if (pos_count == 0 && pos_begin == 0) pos_count = n;
  if (pos_count == 0) return AS_OK;  // (a rank without positions)
  if (pos_begin % 256 != 0 || (uint64_t)pos_begin + pos_count > n) return fail(ctx, AS_ERR_ARG, "threshold pass: bad position range");
  const uint32_t u_begin = pos_begin / 256, u_count = (pos_count + 255) / 256;
  DevBuf<unsigned char> PS;
  DevBuf<uint32_t> unitc, near, ids, ug;
  AS_TRY(PS.alloc(ctx, packed_vecs_f16(np * kGroupTiles, d) * 16));
  AS_TRY(near.alloc(ctx, (size_t)np * kGroupNear));
  AS_TRY(ids.alloc(ctx, (size_t)np * kGroupSlots));
  AS_TRY(ug.alloc(ctx, (size_t)2 * a_supers));
  AS_TRY(unitc.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(unitc.p, 0, sizeof(uint32_t), ctx->stream));
  ProfScope ps(ctx, "knn_join_thresholds");
  if (!near_pre) hipLaunchKernelGGL(near_groups_kernel, dim3(np), dim3(256), 0, ctx->stream, piv, piv_pitch, piv_dim, np, near.p);
  // the sampled rows are gathered from an f16 image (256 B a row at 128 columns; from the f64 rows: 1 KB and the conversion again,
  // 0.22 ms of the 0.8 at 1 M rows): the image by row id where the caller has one (sharded by rows), else P_all itself -- it is in
  // the order of the positions, which is what the sample is drawn in (the ids are then positions: perm is not applied)
  const bool by_position = !row_image && perm && KS <= 8;
  hipLaunchKernelGGL(group_sample_ids_kernel, dim3(np), dim3(64), 0, ctx->stream, near_pre ? near_pre : (const uint32_t *)near.p, group_start, np,
                     by_position ? (const uint32_t *)nullptr : perm, ids.p, kGroupSlots, rate);
  hipLaunchKernelGGL(unit_groups_kernel, dim3(cdiv(a_supers, 256)), dim3(256), 0, ctx->stream, group_start, np, n, a_supers, ug.p);
  AS_HIP(ctx, hipGetLastError());
  if (row_image || by_position)
    AS_TRY(pack_f16_from_image(ctx, by_position ? P_all : row_image, n, ids.p, (uint64_t)np * kGroupSlots, d, PS.p, np * kGroupTiles));
  else AS_TRY(pack_f16(ctx, rows, norms, ids.p, 0, (uint64_t)np * kGroupSlots, d, PS.p, np * kGroupTiles));
  const size_t lds = std::max((size_t)2 * KS * 4 * 64 * 16, (size_t)8 * 32 * 65 * sizeof(float));
  const unsigned grid = std::min<uint32_t>(u_count, 512u);
  const f16x8 *PA_r = (const f16x8 *)P_all + (size_t)(pos_begin / kTileRows) * KS * 4 * 64;
  float *t1_r = theta_out + pos_begin, *t2_r = theta2_out ? theta2_out + pos_begin : nullptr;
  const uint32_t *ug_r = ug.p + 2 * (size_t)u_begin;
#define AS_THETA(K)                                                                                                          \
  case K:                                                                                                                    \
    if (ctx->opt_join_theta_rows64) {                                                                                        \
      AS_HIP(ctx, hipFuncSetAttribute((const void *)theta_f16x2_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      hipLaunchKernelGGL((theta_f16x2_kernel<K>), dim3(grid), dim3(256), lds, ctx->stream, PA_r, u_count, pos_count,         \
                         (const f16x8 *)PS.p, kGroupTiles, m, t1_r, m2, t2_r, unitc.p, ug_r, 1u, (float *)nullptr, group_start); \
      break;                                                                                                                 \
    }                                                                                                                        \
    AS_HIP(ctx, hipFuncSetAttribute((const void *)theta_f16_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((theta_f16_kernel<K>), dim3(grid), dim3(512), lds, ctx->stream, PA_r, u_count, pos_count,             \
                       (const f16x8 *)PS.p, kGroupTiles, m, t1_r, m2, t2_r, unitc.p, ug_r, 1u, (float *)nullptr, group_start); \
    break;
  switch (KS) {
    AS_THETA(1) AS_THETA(2) AS_THETA(3) AS_THETA(4) AS_THETA(5) AS_THETA(6) AS_THETA(7) AS_THETA(8)
    default: return fail(ctx, AS_ERR_UNSUPPORTED, "f16 threshold pass: dim not covered");
  }
#undef AS_THETA
  AS_HIP(ctx, hipGetLastError());
  // (no synchronisation to hand the workspace back: a block returned to the cache is reused by later work on this same stream only;
  // the one that stood here cost 28 us of idle GPU per build)
  return AS_OK;
}

// Thresholds from a row's OWN NEIGHBOURHOOD in the ordered image, not from a sample: theta_out[p] = a value that at least m
// products of row p reach among ALL rows of its own 512-position super tile and the `band` super tiles on either side (the m-th
// largest of 64 bin maxima, as everywhere in this file).  With m = K + 1 (the row meets itself) that is a certificate: K other rows
// have an f16 product >= theta_out[p], so a list cut at theta_out[p] - 2 delta holds K rows whose exact cosines are above the cut
// + delta -- the row is certified whatever the rest of the matrix holds, and the cut sits near the K-th best cosine itself instead
// of near the 128th (the 4th best of a 1-in-32 sample).  The kernel sweeps the image in place: the "sample of group g" is super
// tile g of P_all (4 tiles), a unit's groups the super tiles around its own.  1 M x 128, band 1: 1 536 columns per row, 0.4 TFLOP.
__global__ void band_groups_kernel(uint32_t units, uint32_t T5, uint32_t band, uint32_t *__restrict__ ug) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= units) return;
  const uint32_t I5 = u >> 1;  // the unit's 512-position super tile (a unit = 256 positions)
  ug[2 * u] = I5 > band ? I5 - band : 0u;
  ug[2 * u + 1] = min(T5 - 1u, I5 + band);
}
int join_thresholds_f16_band(as_ctx *ctx, uint32_t n, uint32_t d, const void *P_all, uint32_t band, uint32_t m, float *theta_out,
                             uint32_t pos_begin, uint32_t pos_count) {
  const uint32_t KS = (d + 15) / 16;
  if (!(KS >= 1 && KS <= 8)) return fail(ctx, AS_ERR_UNSUPPORTED, "f16 threshold pass: dim not covered");
  if (m < 1 || m > 32) return fail(ctx, AS_ERR_UNSUPPORTED, "f16 threshold pass: parameters not covered");
  if (pos_count == 0 && pos_begin == 0) pos_count = n;
  if (pos_count == 0) return AS_OK;
  if (pos_begin % 256 != 0 || (uint64_t)pos_begin + pos_count > n) return fail(ctx, AS_ERR_ARG, "threshold pass: bad position range");
  const uint32_t a_supers = (n + 255) / 256, T5 = (n + 511) / 512;
  const uint32_t u_begin = pos_begin / 256, u_count = (pos_count + 255) / 256;
  DevBuf<uint32_t> ug, unitc;
  AS_TRY(ug.alloc(ctx, (size_t)2 * a_supers));
  AS_TRY(unitc.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(unitc.p, 0, sizeof(uint32_t), ctx->stream));
  ProfScope ps(ctx, "knn_join_thresholds_band");
  hipLaunchKernelGGL(band_groups_kernel, dim3(cdiv(a_supers, 256)), dim3(256), 0, ctx->stream, a_supers, T5, band, ug.p);
  const size_t lds = std::max((size_t)2 * KS * 4 * 64 * 16, (size_t)8 * 32 * 65 * sizeof(float));
  const unsigned grid = std::min<uint32_t>(u_count, 512u);
  const f16x8 *PA_r = (const f16x8 *)P_all + (size_t)(pos_begin / kTileRows) * KS * 4 * 64;
  float *t1_r = theta_out + pos_begin;
  const uint32_t *ug_r = ug.p + 2 * (size_t)u_begin;
#define AS_THETA(K)                                                                                                          \
  case K:                                                                                                                    \
    if (ctx->opt_join_theta_rows64) {                                                                                        \
      AS_HIP(ctx, hipFuncSetAttribute((const void *)theta_f16x2_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      hipLaunchKernelGGL((theta_f16x2_kernel<K>), dim3(grid), dim3(256), lds, ctx->stream, PA_r, u_count, pos_count,         \
                         (const f16x8 *)P_all, 4u, m, t1_r, 0u, (float *)nullptr, unitc.p, ug_r);                            \
      break;                                                                                                                 \
    }                                                                                                                        \
    AS_HIP(ctx, hipFuncSetAttribute((const void *)theta_f16_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((theta_f16_kernel<K>), dim3(grid), dim3(512), lds, ctx->stream, PA_r, u_count, pos_count,             \
                       (const f16x8 *)P_all, 4u, m, t1_r, 0u, (float *)nullptr, unitc.p, ug_r);                              \
    break;
  switch (KS) {
    AS_THETA(1) AS_THETA(2) AS_THETA(3) AS_THETA(4) AS_THETA(5) AS_THETA(6) AS_THETA(7) AS_THETA(8)
    default: return fail(ctx, AS_ERR_UNSUPPORTED, "f16 threshold pass: dim not covered");
  }
#undef AS_THETA
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

}  // namespace asx
