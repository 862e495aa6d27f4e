// mfma_join_bf16.hip -- the symmetric threshold join (see mfma_join_f32.hip for the scheme and for what the
// lists mean) contracted on the bf16 matrix pipe with split operands: x = hi + lo, three MFMAs per 16-wide k step
// (lo*hi + hi*lo + hi*hi, v_mfma_f32_32x32x16_bf16: 16x the f32 MFMA rate, so 16/3 of its throughput).
//
// Exactness does not depend on the precision of this pass: a pair is appended to row r's list when its
// approximate cosine s reaches theta[r], so what is NOT in the list has s < theta[r] and therefore an exact cosine
// below theta[r] + delta, delta = cos_delta_bf16x3(d) (mfma_common.cuh: split residual + dropped lo*lo + f32
// accumulation); the exact f64 re-rank certifies a row only if its K-th neighbour is clear of that, everything
// else falls through to the f32 / f64 / scan stages (mfma_drivers.hip).  The wider delta (1e-4 against 8e-6 at
// dim 128) costs a few more uncertified rows, the contraction is 4-5x shorter.
//
// Geometry: 8 waves x 32 rows per workgroup (one 256-row super tile per work unit, the same cyclic half pattern
// and column blocking as the f32 form), split A fragments of a wave in registers (8 VGPRs per k step), B tiles of
// 128 rows (hi plane + lo plane, 64 KB at dim 128) through a 2-deep LDS-DMA ring, one barrier per tile.
// The compare epilogue of a 32x32 block costs about as much vector issue as a quarter of its MFMAs, so it is
// software-pipelined one block behind: the products of block b sit in one accumulator while block b + 1 is
// contracted into the other, and the compares issue in the shadow of those MFMAs.
#include <hip/hip_bf16.h>

#include <type_traits>

#include "mfma_common.cuh"

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

  // dbg 8: cycle stamps (diagnostic build of the SAME kernel; the sums go to a buffer nothing else reads)
  auto now = [&]() -> unsigned long long {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
  };
  unsigned long long t_unit = 0, t_blocks = 0, t_dma = 0, t_bar = 0, t_mark = 0;
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
    float tr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) tr[r] = theta[row_base + (r & 3) + 8 * (r >> 2)];
    float lqb[SEARCH ? 16 : 1];  // beta * lambda of this lane's 16 rows
    if (SEARCH) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        lqb[r] = st.beta * st.a_lam[row_base + (r & 3) + 8 * (r >> 2)];
        tr[r] -= st.beta;  // guess - beta
      }
    }
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

    // products of one 32x32 block: rows of this wave x columns [32 nn, 32 nn + 32) of the staged tile
    // Fragment sets b_hi[s], b_lo[s], s = k step parity; the reads of step ks + 1 are issued before the wait for
    // step ks.  ks = 0 of a block is issued by the block before it (AS_PREFETCH) or after the tile's barrier.
    bf16x8 b_hi[2], b_lo[2];
#define AS_READ(SET, KSTEP, NN)                                                   \
  {                                                                               \
    lds_read16<((KSTEP)*4 + (NN)) * 1024>(b_hi[SET], b_addr);                     \
    lds_read16<(kPlaneVecs + ((KSTEP)*4 + (NN)) * 64) * 16>(b_lo[SET], b_addr);   \
  }
    // products of one 32x32 block: rows of this wave x columns [32 nn, 32 nn + 32) of the staged tile.
    // NEXT_NN >= 0: the first fragments of that block of the same tile are requested before the last MFMAs.
    // k steps [K0, K1) of the block
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
          if (cur) lds_wait<2>(b_hi[1], b_lo[1]); else lds_wait<2>(b_hi[0], b_lo[0]);
        } else {
          if (cur) lds_wait<0>(b_hi[1], b_lo[1]); else lds_wait<0>(b_hi[0], b_lo[0]);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[ks], b_hi[cur], ks == 0 ? z : acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[ks], b_lo[cur], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[ks], b_hi[cur], acc, 0, 0, 0);
      }
    };
    // does any product of the block reach its row's or its column's threshold?
    auto any_hit = [&](const v16f32 &acc, float tcv) -> bool {
      bool h = false;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (SEARCH) h = h || (weigh(acc[r], r, tcv) >= tr[r]);
        else h = h || (acc[r] >= tr[r]) || (acc[r] >= tcv);
      }
      return h;
    };
    // the rows of a block that carries a hit are tested one by one with wave-uniform branches; what passes goes to
    // the wave's queue (flush_queue below)
    auto emit = [&](const v16f32 &acc, float tcv, uint32_t col) {
      // the fragments requested for the next block are still in flight: nothing may touch their registers
      // before they land (the code below is long enough for the allocator to want to move things around)
      lds_wait<0>(b_hi[0], b_lo[0]);
      lds_wait<0>(b_hi[1], b_lo[1]);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = SEARCH ? weigh(acc[r], r, tcv) : acc[r];
        const bool pr = v >= tr[r], pc = SEARCH ? false : v >= tcv;
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

    // block pipeline: accA / accB alternate; `prev` = the block whose products wait for their compares
    v16f32 accA, accB;
#pragma unroll
    for (int r = 0; r < 16; ++r) accA[r] = accB[r] = -__builtin_inff();  // nothing pending: never a hit
    float tc_prev = __builtin_inff();
    uint32_t col_prev = 0;

    __syncthreads();
    stage(tile_of(0), 0, -1);
    __syncthreads();
    if (stamp) {
      const unsigned long long t = now();
      t_unit += t - t_mark;
      t_mark = t;
    }
    for (uint32_t step = 0; step < steps; ++step) {
      const uint32_t tile = tile_of(step);
      const int pbuf = (int)(step & 1u);
      const bool more_tiles = step + 1 < steps && !(dbg & 4);  // dbg 4: timing ablation
      const uint32_t tile_next = more_tiles ? tile_of(step + 1) : 0;
      const bool diag = RECT || (p_lo + (step >> 1)) == 0;
      // column thresholds of this lane's four columns (the diagonal block is row-side only)
      float tc[4];
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) {
        if (SEARCH) tc[nn] = st.beta * st.x_lam[tile * kTileRows + nn * 32 + lc];  // the column's term, not a threshold
        else tc[nn] = diag ? __builtin_inff() : theta[tile * kTileRows + nn * 32 + lc];
      }
      const uint32_t b_addr = lds_addr(bufB + (size_t)pbuf * kTileVecs) + (uint32_t)lane * 16u;
      const uint32_t col0 = tile * kTileRows + lc;

// A 32x32 block in two halves of its k steps; the compares of the block before ride behind the MFMAs of the first
// half.  The quarter of the next tile's DMA pieces that this block issues goes in front of the first half in waves
// 0-3 and between the halves in waves 4-7: the two waves of a SIMD run in lockstep between barriers, and a DMA piece
// holds its wave's issue for 60-190 cycles, so issued at the same moment they leave the matrix pipe idle.
#define AS_IC(V) std::integral_constant<int, (V)>{}
#define AS_BLOCK(NN, NEXT, CUR, PREV)                               \
  {                                                                 \
    constexpr int KH = (KS + 1) / 2;                                \
    if (more_tiles && wv < 4) stage(tile_next, pbuf ^ 1, NN);       \
    contract(b_addr, AS_IC(NN), AS_IC(NEXT), AS_IC(0), AS_IC(KH), CUR); \
    const bool h = any_hit(PREV, tc_prev);                          \
    _Pragma("unroll") for (int i = 0; i < 3 * KH; ++i) {            \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            \
      __builtin_amdgcn_sched_group_barrier(0x002, SEARCH ? 6 : 3, 0); \
    }                                                               \
    if (more_tiles && wv >= 4) stage(tile_next, pbuf ^ 1, NN);      \
    contract(b_addr, AS_IC(NN), AS_IC(NEXT), AS_IC(KH), AS_IC(KS), CUR); \
    if (!(dbg & 2) && __any(h)) emit(PREV, tc_prev, col_prev);      \
    tc_prev = tc[NN];                                               \
    col_prev = col0 + NN * 32;                                      \
  }
      AS_READ(0, 0, 0)
      AS_BLOCK(0, 1, accA, accB)
      AS_BLOCK(1, 2, accB, accA)
      AS_BLOCK(2, 3, accA, accB)
      AS_BLOCK(3, -1, accB, accA)
#undef AS_BLOCK
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
#undef AS_READ
    // the last block of the unit
    {
      const bool h = any_hit(accB, tc_prev);
      if (!(dbg & 2) && __any(h)) emit(accB, tc_prev, col_prev);
    }
  }
  flush_queue();
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
  const uint32_t W = rect ? std::min(w_opt, T2) : std::max(1u, std::min(w_opt, T2 / 4));
  const uint64_t units = (uint64_t)((T2 + W - 1) / W) * I_count;
  const unsigned grid = (unsigned)std::min<uint64_t>(units, 256u);
  DevBuf<unsigned long long> stamps;
  if (ctx->opt_debug & 8) {
    AS_TRY(stamps.alloc(ctx, 8));
    AS_HIP(ctx, hipMemsetAsync(stamps.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
  }
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
  if (ctx->opt_debug & 8) {
    unsigned long long h[8];
    AS_HIP(ctx, copy_d2h(ctx, h, stamps.p, sizeof(h)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const double w = (double)std::max<unsigned long long>(1, h[4]);
    std::fprintf(stderr, "[join_bf16 stamps] per wave, Mcycles: unit prologue %.2f  blocks %.2f  dma wait %.2f  barrier %.2f\n",
                 h[0] / w / 1e6, h[1] / w / 1e6, h[2] / w / 1e6, h[3] / w / 1e6);
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
