// mfma_join_bf16.hip -- the symmetric threshold join (see mfma_join_f32.hip for the scheme and for what the
// lists mean) contracted on the bf16 matrix pipe with split operands: x = hi + lo, and the product of two rows
// taken as hi*hi' + lo*hi' + hi*lo' (v_mfma_f32_32x32x16_bf16: 16x the f32 MFMA rate).
//
// Exactness does not depend on the precision of this pass: a pair is appended to row r's list when its
// approximate cosine s reaches theta[r], so what is NOT in the list has s < theta[r] and therefore an exact cosine
// below theta[r] + delta, delta = cos_delta_bf16x3(d) (mfma_common.h: split residual + dropped lo*lo + f32
// accumulation); the exact f64 re-rank certifies a row only if its K-th neighbour is clear of that, everything
// else falls through to the second pass / f32 / f64 / scan stages (mfma_drivers.hip).
//
// Two phases per 32x32 block: the hi planes alone first (KS MFMAs); |s - s_hi| <= 3.915e-3 for unit rows, so a block
// whose hi-only products all stay eps = 3.94e-3 below their thresholds cannot hold a pair the full product would
// list, and its other 2 KS MFMAs are never issued (87 % of the blocks at 1 M x 128).  The rest run phase B (the two
// cross terms on top of the same accumulator) and the exact test.  The three modes of the kernel: 0 = symmetric
// join, 1 = one-sided (a gathered subset of the rows against everything: the second pass), 2 = one-sided with the
// search score as the tested quantity (the threshold scan of the search).
//
// Geometry: 8 waves x 32 rows per workgroup (one 256-row super tile per work unit, the same cyclic half pattern
// and column blocking as the f32 form), split A fragments of a wave in registers (8 VGPRs per k step), B tiles of
// 128 rows (hi plane + lo plane, 64 KB at dim 128) through a 2-deep LDS-DMA ring, one barrier per tile.  What is
// not an MFMA is arranged around the MFMA stream: hand-issued LDS fragment reads three k steps ahead, the coarse
// test of a block behind the phase-A MFMAs of the next one, DMA pieces spread over the blocks, listed pairs through
// a per-wave LDS queue (details at the block pipeline below; measurements in DESIGN.md section 4).
#include <hip/hip_bf16.h>

#include <type_traits>

#include "mfma_common.h"

namespace asx {

namespace {

typedef float v16f32 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

constexpr int kSuper = 256;  // rows per super tile (two packed 128-row tiles)
constexpr uint32_t kQueue = 256;  // queued appends per wave (one 32x32 row step adds at most 128)

__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(size_t)(lds_void *)p; }

// LDS fragment reads whose latency the compiler must not wait out on the spot: issued by hand, waited for by
// hand (the "+v" operands of lds_wait pin the consumers behind the wait; cdna_hip_programming.md, asm loads).
template <int OFF>
__device__ __forceinline__ void lds_read16(bf16x8 &dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int PENDING>
__device__ __forceinline__ void lds_wait(bf16x8 &x, bf16x8 &y) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x), "+v"(y) : "n"(PENDING));
}
template <int PENDING>
__device__ __forceinline__ void lds_wait1(bf16x8 &x) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(PENDING));
}
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}
constexpr float kHiEps = 3.94e-3f;  // >= 2 * 2^-9 (1 + 2^-9)(1 + 1e-6): what the lo planes can add to a hi-only product

// a list slot is two words: the other row and the approximate score that put it there (the re-rank drops the
// candidates whose score cannot reach the top K before it fetches any row)
__device__ __forceinline__ void append(uint32_t *__restrict__ cnt, uint32_t *__restrict__ cand, uint32_t cap,
                                       uint32_t target, uint32_t other, float score) {
  const uint32_t pos = atomicAdd(cnt + target, 1u);
  if (pos < cap) reinterpret_cast<uint2 *>(cand)[(size_t)target * cap + pos] = make_uint2(other, __float_as_uint(score));
}

// P = split bf16 image of ALL rows (mfma_theta_bf16.hip: tiles of 128 rows, hi plane then lo plane, even tile
// count, zero padded).  Work units as in pair_join_mfma32_kernel.
//
// RECT: the one-sided form.  The A rows are another matrix (the queries of a search) or a gathered subset of the
// rows (what a stage left uncertified): image PA of nA rows, super tile I of it against EVERY column block, a pair
// is appended to the A row's list only (row-side test), lists / thresholds are indexed by A row; a_ids (or nullptr)
// gives the global row of an A row, for the self exclusion.
//
// MODE 2 (search): one-sided, and what is tested is not the product but the approximate search score built on it,
// alpha * cos + beta * (1 - min(|lambda_q - lambda_x|, 1)) >= guess[q]  (0 < alpha <= 1, beta = 1 - alpha; src/core.rs:229-239),
// rearranged as  alpha * s - min(|beta lq - beta lx|, beta) >= guess - beta  (four vector instructions per product).
// theta = the score guesses of the queries (+inf on padding rows).
struct JoinSearchTerm {
  const float *a_lam;  // lambdas of the A rows (padded to whole super tiles)
  const float *x_lam;  // lambdas of all rows of P (padded to whole tiles)
  float alpha, beta;
};
template <int KS, int MODE>
__global__ __launch_bounds__(512, 2) void pair_join_bf16_kernel(const bf16x8 *__restrict__ P, const bf16x8 *__restrict__ PA,
                                                                uint32_t nA, const uint32_t *__restrict__ a_ids,
                                                                JoinSearchTerm st,
                                                                uint32_t T2, uint32_t I_begin,
                                                                uint32_t I_count, uint32_t W, uint32_t n,
                                                                const float *__restrict__ theta, uint32_t cap,
                                                                uint32_t *__restrict__ cnt, uint32_t *__restrict__ cand,
                                                                uint32_t *__restrict__ unit_counter, int dbg,
                                                                unsigned long long *__restrict__ stamps) {
  constexpr bool RECT = MODE >= 1, SEARCH = MODE == 2;
  constexpr int kPlaneVecs = KS * 4 * 64;    // 16-B vectors per plane of a 128-row tile
  constexpr int kTileVecs = 2 * kPlaneVecs;  // hi plane + lo plane
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16x8 *bufB = reinterpret_cast<bf16x8 *>(smem);  // 2 x tile
  __shared__ uint32_t s_unit;

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lh = lane >> 5, lc = lane & 31;
  // Passing pairs are queued per wave in LDS as (list owner, other row) and appended kQueue/2 or more at a time:
  // an append is an atomic that returns the slot, ~2 us before the store can go out, and taken one by one inside
  // the sweep those waits made the wave with the hit late at the tile's barrier, where the other seven wait for it
  // (stamps: 17 % of the kernel).  From the queue 64 lanes take one entry each: one such wait per 64 appends.
  uint2 *queue = reinterpret_cast<uint2 *>(smem + (size_t)2 * kTileVecs * 16) + (size_t)wv * kQueue;
  float *queue_s = reinterpret_cast<float *>(smem + (size_t)2 * kTileVecs * 16 + (size_t)8 * kQueue * sizeof(uint2)) + (size_t)wv * kQueue;
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
        append(cnt, cand, cap, v.x, v.y, queue_s[e]);
      }
    }
    wave_sync();
    qn = 0;
  };
  const uint32_t n_norm = (T2 - 1) / 2;
  const uint32_t n_blocks = (T2 + W - 1) / W;
  const uint32_t total_units = n_blocks * I_count;

  // 128-row tile -> LDS: 1 KB pieces, KS per wave.  part < 0: all of them; part = 0..3: the pieces of that
  // quarter (one quarter is issued inside each of the four blocks of the tile before, among its MFMAs: eight
  // back-to-back pieces right after the barrier, from every wave at once, leave the matrix pipe idle)
  auto stage = [&](uint32_t tile, int buf, int part) {
    const bf16x8 *src = P + (size_t)tile * kTileVecs;
    bf16x8 *dst = bufB + (size_t)buf * kTileVecs;
#pragma unroll
    for (int p = 0; p < KS; ++p)
      if (part < 0 || (p & 3) == part)
        __builtin_amdgcn_global_load_lds((gbl_void *)(src + (wv * KS + p) * 64 + lane), (lds_void *)(dst + (wv * KS + p) * 64),
                                         16, 0, 0);
  };

  // Diagnostics (as_ctx_set_option("debug", bits); results are undefined except with 8 and 32): 2 no emit, 4 no DMA of
  // the next tile, 8 cycle stamps, 64 no phase B, 128 no coarse test.
  // dbg 8: cycle stamps (diagnostic run of the SAME kernel; the sums go to a buffer nothing else reads)
  auto now = [&]() -> unsigned long long {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
  };
  unsigned long long t_unit = 0, t_blocks = 0, t_dma = 0, t_bar = 0, t_mark = 0;
  unsigned long long n_blk = 0, n_coarse = 0;  // blocks contracted, blocks that pass the coarse test (phase B)
  const bool stamp = (dbg & 8) != 0;
  if (stamp) t_mark = now();

  for (;;) {
    __syncthreads();
    if (tid == 0) s_unit = atomicAdd(unit_counter, 1u);
    __syncthreads();
    const uint32_t u = s_unit;
    if (u >= total_units) break;
    const uint32_t c = u / I_count, I = I_begin + (u - c * I_count);
    const uint32_t B0 = c * W, Wc = min(W, T2 - B0);
    uint32_t p_lo = 0, p_hi = Wc;  // RECT: the whole column block, offsets counted from B0
    if (!RECT) {
      const bool antipodal = (T2 % 2 == 0) && (I < T2 / 2) && T2 > 1;
      const uint32_t L = 1 + n_norm + (antipodal ? 1u : 0u);  // offsets of this super tile: 0 = diagonal
      const uint32_t delta = B0 >= I ? B0 - I : B0 + T2 - I;  // offset of the block's first super tile
      if (delta + Wc <= T2) {
        p_lo = delta;
        p_hi = min(delta + Wc, L);
      } else {  // the block straddles I: its tail is offsets 0 .. (W <= T2 / 4 keeps the head out of the pattern)
        p_lo = 0;
        p_hi = min(delta + Wc - T2, L);
      }
      if (p_hi <= p_lo) continue;
    }
    const uint32_t steps = (p_hi - p_lo) * 2;  // 128-row B tiles to visit

    bf16x8 a_hi[KS], a_lo[KS];
    {
      const int row0 = wv * 32;
      const size_t T = (size_t)2 * I + (row0 >> 7);
      const int s0 = (row0 & 127) >> 5;
      const bf16x8 *a_src = (RECT ? PA : P) + T * kTileVecs + (size_t)s0 * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        a_hi[ks] = a_src[(size_t)ks * 256];
        a_lo[ks] = a_src[kPlaneVecs + (size_t)ks * 256];
      }
    }
    const uint32_t row_base = I * kSuper + wv * 32 + 4 * lh;  // + (r & 3) + 8 * (r >> 2)
    // Thresholds are held lowered by eps, the slack of the hi-only products of phase A (see the block comment
    // below); the exact test of phase B adds eps back (two f32 roundings: 2.4e-7, carried by cos_delta_bf16x3).
    const float eps = SEARCH ? kHiEps * st.alpha : kHiEps;
    float tre[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) tre[r] = theta[row_base + (r & 3) + 8 * (r >> 2)];
    float lqb[SEARCH ? 16 : 1];  // beta * lambda of this lane's 16 rows
    if (SEARCH) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        lqb[r] = st.beta * st.a_lam[row_base + (r & 3) + 8 * (r >> 2)];
        tre[r] -= st.beta;  // guess - beta
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) tre[r] -= eps;
    // search: the tested quantity of a product (tcv = beta * lambda of the column)
    auto weigh = [&](float sdot, int r, float tcv) -> float {
      const float t = __builtin_fminf(__builtin_fabsf(lqb[SEARCH ? r : 0] - tcv), st.beta);
      return __builtin_fmaf(st.alpha, sdot, -t);
    };

    auto tile_of = [&](uint32_t step) -> uint32_t {
      uint32_t S = (RECT ? B0 : I) + p_lo + (step >> 1);
      S = S >= T2 ? S - T2 : S;
      return 2 * S + (step & 1u);
    };

    // ---- a 32x32 block in two phases -------------------------------------------------------------------------
    // Phase A contracts the hi planes only (KS MFMAs instead of 3 KS).  |s - s_hi| = |sum hi lo' + lo hi'| <=
    // 2 * 2^-9 (1 + 2^-9) sum |x||x'| <= 3.915e-3 for unit rows (Cauchy-Schwarz), so a block none of whose hi-only
    // products reaches its threshold minus eps = 3.94e-3 cannot hold a pair that the full product would list: its
    // other 2 KS MFMAs are skipped (89 % of the blocks at 1 M x 128).  The others run phase B (lo hi' + hi lo' on
    // top of the same accumulator: the full split product) and the exact test.
    // Pipeline: the coarse test of block b rides behind the phase-A MFMAs of block b + 1 (two accumulators
    // alternate); phase B of b, when needed, follows.  Phase-A fragments (hi plane) are requested three k steps
    // ahead through a ring of four registers sets, across block boundaries of the tile; phase B reads its own.
    bf16x8 fa[4], fb_hi[2], fb_lo[2];
    v16f32 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;

    // request the hi fragment of global k step G = nn * KS + ks of the staged tile
    auto request_a = [&](uint32_t b_addr, auto g_c) {
      constexpr int G = decltype(g_c)::value;
      if constexpr (G < 4 * KS) lds_read16<((G % KS) * 4 + (G / KS)) * 1024>(fa[G & 3], b_addr);
    };
    auto phase_a = [&](uint32_t b_addr, auto nn_c, v16f32 &acc) {
      constexpr int nn = decltype(nn_c)::value;
      static_for<0, KS>([&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value, G = nn * KS + ks;
        request_a(b_addr, std::integral_constant<int, G + 3>{});
        constexpr int pending = (4 * KS - 1 - G) < 3 ? (4 * KS - 1 - G) : 3;  // younger requests in flight
        lds_wait1<pending>(fa[G & 3]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[ks], fa[G & 3], ks == 0 ? zero16 : acc, 0, 0, 0);
      });
    };
    auto phase_b = [&](uint32_t b_addr, auto nn_c, v16f32 &acc) {
      constexpr int nn = decltype(nn_c)::value;
      lds_read16<nn * 1024>(fb_hi[0], b_addr);
      lds_read16<(kPlaneVecs + nn * 64) * 16>(fb_lo[0], b_addr);
      static_for<0, KS>([&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value, cur = ks & 1, nxt = cur ^ 1;
        if constexpr (ks + 1 < KS) {
          lds_read16<((ks + 1) * 4 + nn) * 1024>(fb_hi[nxt], b_addr);
          lds_read16<(kPlaneVecs + ((ks + 1) * 4 + nn) * 64) * 16>(fb_lo[nxt], b_addr);
          lds_wait<2>(fb_hi[cur], fb_lo[cur]);
        } else {
          lds_wait<0>(fb_hi[cur], fb_lo[cur]);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[ks], fb_hi[cur], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[ks], fb_lo[cur], acc, 0, 0, 0);
      });
    };
    // coarse test on the hi-only products: can any full product of the block reach a threshold?
    // tcv: the column's lowered threshold (join) or its lambda term (search)
    auto coarse = [&](const v16f32 &acc, float tcv) -> bool {
      bool h = false;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (SEARCH) h = h || (weigh(acc[r], r, tcv) >= tre[r]);
        else h = h || (acc[r] >= tre[r]) || (acc[r] >= tcv);
      }
      return h;
    };
    // exact test on the full products, row by row with wave-uniform branches; what passes goes to the wave's queue
    auto emit = [&](const v16f32 &acc, float tcv, uint32_t col) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = SEARCH ? weigh(acc[r], r, tcv) : acc[r];
        const bool pr = v >= tre[r] + eps, pc = SEARCH ? false : v >= tcv + eps;
        if (__any(pr || pc)) {
          const uint32_t row = row_base + (r & 3) + 8 * (r >> 2);
          bool inside;
          if (RECT) inside = row < nA && col < n && (a_ids == nullptr || a_ids[row] != col);
          else inside = row < n && col < n && row != col;
          const bool w1 = inside && pr, w2 = inside && pc;
          const unsigned long long m1 = __ballot(w1), m2 = __ballot(w2);
          const uint32_t n1 = (uint32_t)__builtin_popcountll(m1), n2 = (uint32_t)__builtin_popcountll(m2);
          if (qn + n1 + n2 > kQueue) flush_queue();
          if (w1) {
            queue[qn + lanes_below(m1)] = make_uint2(row, col);
            queue_s[qn + lanes_below(m1)] = v;
          }
          if (w2) {
            queue[qn + n1 + lanes_below(m2)] = make_uint2(col, row);
            queue_s[qn + n1 + lanes_below(m2)] = v;
          }
          qn += n1 + n2;
        }
      }
    };

    __syncthreads();
    stage(tile_of(0), 0, -1);
    __syncthreads();
    if (stamp) {
      const unsigned long long t = now();
      t_unit += t - t_mark;
      t_mark = t;
    }
    v16f32 accA, accB;
    float tc_next[4];
    {
      const uint32_t tile0 = tile_of(0);
      const bool diag0 = RECT || p_lo == 0;
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) {
        if (SEARCH) tc_next[nn] = st.beta * st.x_lam[tile0 * kTileRows + nn * 32 + lc];
        else tc_next[nn] = diag0 ? __builtin_inff() : theta[tile0 * kTileRows + nn * 32 + lc] - eps;
      }
    }
    for (uint32_t step = 0; step < steps; ++step) {
      const uint32_t tile = tile_of(step);
      const int pbuf = (int)(step & 1u);
      const bool more_tiles = step + 1 < steps && !(dbg & 4);  // dbg 4: timing ablation
      const uint32_t tile_next = more_tiles ? tile_of(step + 1) : 0;
      // this lane's four columns: lowered thresholds (the diagonal block is row-side only), or lambda terms (search).
      // Loaded one tile ahead: a tile is ~4 us now and the first use comes 0.3 us after its start.
      float tc[4];
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) tc[nn] = tc_next[nn];
      if (more_tiles) {
        const bool diag_next = RECT || (p_lo + ((step + 1) >> 1)) == 0;
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
          if (SEARCH) tc_next[nn] = st.beta * st.x_lam[tile_next * kTileRows + nn * 32 + lc];
          else tc_next[nn] = diag_next ? __builtin_inff() : theta[tile_next * kTileRows + nn * 32 + lc] - eps;
        }
      }
      const uint32_t b_addr = lds_addr(bufB + (size_t)pbuf * kTileVecs) + (uint32_t)lane * 16u;
      const uint32_t col0 = tile * kTileRows + lc;

#define AS_IC(V) std::integral_constant<int, (V)>{}
// decide block NN (products in ACC): full products and the exact test only if the hi-only ones come close
#define AS_DECIDE(NN, ACC, PASS)                                    \
  {                                                                 \
    n_blk += 1;                                                     \
    n_coarse += __any(PASS) ? 1 : 0;                                \
    if (__any(PASS) && !(dbg & 64)) {                               \
      phase_b(b_addr, AS_IC(NN), ACC);                              \
      if (!(dbg & 2)) emit(ACC, tc[NN], col0 + (NN) * 32);          \
    }                                                               \
  }
// phase A of block NN into CUR; the coarse test of the block before it (PREV) rides behind these MFMAs.  The
// quarter of the next tile's DMA pieces that this block issues goes in front in waves 0-3 and behind in waves 4-7:
// a piece holds its wave's issue for 60-190 cycles, and the two waves of a SIMD run in lockstep between barriers.
#define AS_BLOCK(NN, CUR, PREV)                                     \
  {                                                                 \
    if (more_tiles && wv < 4) stage(tile_next, pbuf ^ 1, NN);       \
    phase_a(b_addr, AS_IC(NN), CUR);                                \
    bool pass = false;                                              \
    if ((NN) > 0 && !(dbg & 128)) pass = coarse(PREV, tc[(NN) > 0 ? (NN)-1 : 0]); \
    _Pragma("unroll") for (int i = 0; i < KS; ++i) {                \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            \
      __builtin_amdgcn_sched_group_barrier(0x002, SEARCH ? 10 : 4, 0); \
    }                                                               \
    if (more_tiles && wv >= 4) stage(tile_next, pbuf ^ 1, NN);      \
    if ((NN) > 0) AS_DECIDE((NN) > 0 ? (NN)-1 : 0, PREV, pass)      \
  }
      request_a(b_addr, AS_IC(0));
      request_a(b_addr, AS_IC(1));
      request_a(b_addr, AS_IC(2));
      AS_BLOCK(0, accA, accB)
      AS_BLOCK(1, accB, accA)
      AS_BLOCK(2, accA, accB)
      AS_BLOCK(3, accB, accA)
      {  // the last block of the tile: its fragments go with the barrier
        const bool pass = coarse(accB, tc[3]);
        AS_DECIDE(3, accB, pass)
      }
#undef AS_BLOCK
#undef AS_DECIDE
#undef AS_IC
      if (stamp) {
        unsigned long long t = now();
        t_blocks += t - t_mark;
        t_mark = t;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t = now();
        t_dma += t - t_mark;
        t_mark = t;
      }
      __syncthreads();  // DMA of the next tile landed; everyone is done reading bufB[pbuf]
      if (stamp) {
        const unsigned long long t = now();
        t_bar += t - t_mark;
        t_mark = t;
      }
    }
  }
  flush_queue();
  if (lane == 0) {  // blocks contracted, and how many of them in full (the executed MFMAs of the launch)
    atomicAdd(stamps + 5, n_blk);
    atomicAdd(stamps + 7, n_coarse);
  }
  if (stamp && lane == 0) {
    atomicAdd(stamps + 0, t_unit);
    atomicAdd(stamps + 1, t_blocks);
    atomicAdd(stamps + 2, t_dma);
    atomicAdd(stamps + 3, t_bar);
    atomicAdd(stamps + 4, 1ull);
  }
}

}  // namespace

uint64_t packed_vecs_bf16(uint32_t tiles, uint32_t d);

static int join_bf16_launch(as_ctx *ctx, int mode, const JoinSearchTerm &st, const void *P, const void *PA, uint32_t nA, const uint32_t *a_ids,
                            uint32_t n, uint32_t d, uint32_t I_begin, uint32_t I_count, const float *theta, uint32_t cap,
                            uint32_t *cnt, uint32_t *cand, uint32_t *unit_counter, const char *prof_name) {
  const uint32_t KS = (d + 15) / 16;
  if (!(KS >= 1 && KS <= 8)) return fail(ctx, AS_ERR_UNSUPPORTED, "bf16 join: dim not covered");
  const uint32_t b_tiles = (n + kTileRows - 1) / kTileRows;
  const uint32_t T2 = (b_tiles + 1) / 2;
  const size_t lds = (size_t)2 * 2 * KS * 4 * 64 * 16 + (size_t)8 * kQueue * (sizeof(uint2) + sizeof(float));

  const bool rect = mode >= 1;
  const uint32_t w_opt = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(ctx->opt_join_block, 64));
  const uint32_t W = rect ? std::min(std::min(w_opt, 32u), T2) : std::max(1u, std::min(w_opt, T2 / 4));  // (one-sided: few row super tiles, see join_f16_search)
  const uint64_t units = (uint64_t)((T2 + W - 1) / W) * I_count;
  const unsigned grid = (unsigned)std::min<uint64_t>(units, 256u);
  DevBuf<unsigned long long> stamps;
  AS_TRY(stamps.alloc(ctx, 8));
  AS_HIP(ctx, hipMemsetAsync(stamps.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
  {
    ProfScope ps(ctx, prof_name);
#define AS_JOIN1(K, R)                                                                                                 \
  AS_HIP(ctx, hipFuncSetAttribute((const void *)pair_join_bf16_kernel<K, R>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)lds));                                                                          \
  hipLaunchKernelGGL((pair_join_bf16_kernel<K, R>), dim3(grid), dim3(512), lds, ctx->stream, (const bf16x8 *)P,        \
                     (const bf16x8 *)PA, nA, a_ids, st, T2, I_begin, I_count, W, n, theta, cap, cnt, cand, unit_counter, \
                     (int)ctx->opt_debug, stamps.p)
#define AS_JOIN(K)                                                                                                     \
  case K:                                                                                                              \
    if (mode == 2) { AS_JOIN1(K, 2); } else if (mode == 1) { AS_JOIN1(K, 1); } else { AS_JOIN1(K, 0); }                \
    break;
    switch (KS) {
      AS_JOIN(1) AS_JOIN(2) AS_JOIN(3) AS_JOIN(4) AS_JOIN(5) AS_JOIN(6) AS_JOIN(7) AS_JOIN(8)
      default: return fail(ctx, AS_ERR_UNSUPPORTED, "bf16 join: dim not covered");
    }
#undef AS_JOIN
#undef AS_JOIN1
    AS_HIP(ctx, hipGetLastError());
  }
  {
    unsigned long long h[8];
    AS_HIP(ctx, copy_d2h(ctx, h, stamps.p, sizeof(h)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // 32x32 blocks contracted on the hi planes (KS MFMAs each) and, of those, in full (2 KS more)
    ctx->stats[std::string(prof_name) + "_blocks"] = (int64_t)h[5];
    ctx->stats[std::string(prof_name) + "_blocks_full"] = (int64_t)h[7];
    if (ctx->opt_debug & 8) {
      const double w = (double)std::max<unsigned long long>(1, h[4]);
      std::fprintf(stderr,
                   "[join_bf16 stamps] per wave, Mcycles: unit prologue %.2f  blocks %.2f  dma wait %.2f  barrier %.2f; %llu blocks, "
                   "%.3f in full\n",
                   h[0] / w / 1e6, h[1] / w / 1e6, h[2] / w / 1e6, h[3] / w / 1e6, h[5], h[5] ? (double)h[7] / h[5] : 0.0);
    }
  }
  return AS_OK;
}

// Threshold join over super tiles [I_begin, I_begin + I_count) of the T2 super tiles of the split bf16 image P
// (all n rows).  cnt / cand (n x cap) are indexed by global row; cnt must be zeroed by the caller, unit_counter too.
// dim <= 128 (the split A fragments stay in registers); AS_ERR_UNSUPPORTED beyond.
int join_bf16(as_ctx *ctx, const void *P, uint32_t n, uint32_t d, uint32_t I_begin, uint32_t I_count, const float *theta,
              uint32_t cap, uint32_t *cnt, uint32_t *cand, uint32_t *unit_counter, const char *prof_name) {
  return join_bf16_launch(ctx, 0, JoinSearchTerm{nullptr, nullptr, 0.f, 0.f}, P, P, n, nullptr, n, d, I_begin, I_count, theta, cap,
                          cnt, cand, unit_counter, prof_name);
}

// One-sided form: the nA rows of image PA (even tile count; theta_a padded to whole super tiles with +inf) against
// all n rows of P; cnt / cand (nA x cap) are indexed by A row.  a_ids: global row of an A row (self exclusion) or nullptr.
int join_bf16_rect(as_ctx *ctx, const void *P, uint32_t n, uint32_t d, const void *PA, uint32_t nA, const uint32_t *a_ids,
                   const float *theta_a, uint32_t cap, uint32_t *cnt, uint32_t *cand, uint32_t *unit_counter,
                   const char *prof_name) {
  const uint32_t a_supers = (nA + 255) / 256;
  return join_bf16_launch(ctx, 1, JoinSearchTerm{nullptr, nullptr, 0.f, 0.f}, P, PA, nA, a_ids, n, d, 0, a_supers, theta_a, cap, cnt,
                          cand, unit_counter, prof_name);
}

// Search form of the one-sided join (MODE 2 of the kernel): the nq queries of image PQ against all n items of P;
// guess (padded to whole super tiles with +inf): score guess per query; q_lam / x_lam: f32 lambdas, padded.
int join_bf16_search(as_ctx *ctx, const void *P, uint32_t n, uint32_t d, const void *PQ, uint32_t nq, const float *guess,
                     const float *q_lam, const float *x_lam, float alpha, float beta, uint32_t cap, uint32_t *cnt,
                     uint32_t *cand, uint32_t *unit_counter, const char *prof_name) {
  const uint32_t a_supers = (nq + 255) / 256;
  return join_bf16_launch(ctx, 2, JoinSearchTerm{q_lam, x_lam, alpha, beta}, P, PQ, nq, nullptr, n, d, 0, a_supers, guess, cap, cnt,
                          cand, unit_counter, prof_name);
}

}  // namespace asx
