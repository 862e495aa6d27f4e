// mfma_theta_bf16.hip -- candidate thresholds of the symmetric join (mfma_join_f32.hip) on the bf16 matrix pipe.
//
// theta[r] only has to be a sensible cut: the join is exact for ANY threshold (the re-rank certifies against
// whatever theta was used, mfma_drivers.hip), theta decides how many candidates a row collects and how many rows
// fall through to the later stages.  So the products of every row with a hashed sample of the rows are taken on
// the bf16 pipe (v_mfma_f32_32x32x16_bf16, 16x the f32 MFMA rate) with split operands, x = hi + lo, and three MFMAs
// per k step (hi*hi + hi*lo + lo*hi: ~2^-16 relative; plain bf16 is too coarse: on clustered data thousands of
// neighbours sit within 0.02 of the cut and 1e-3 of noise moves it by hundreds of candidates), and reduced with
// one v_max per product: lane (h, c) of
// a wave sees, for each of its 16 rows, the sample columns congruent to c mod 32, and keeps their maximum.  A row
// (split again by the parity of the sample tile) therefore ends with 64 maxima over 64 disjoint sub-samples;
// theta[r] is the m-th largest of them -- the m-th
// best sampled product when the best m fall into different sub-samples, a little lower otherwise (which only
// makes the cut more generous).  No lists, no insertions: the f32 top-k form of this pass took 11 % of the build.
//
// Geometry: 8 waves x 32 rows per workgroup (one 256-row super tile per work unit, dealt from an atomic
// counter), the A fragments of a wave in registers (4 VGPRs per 16-wide k step), sample tiles of 128 rows
// through a 2-deep LDS-DMA ring (32 KB per tile at dim 128), all workgroups walking the sample in the same order.
#include <hip/hip_bf16.h>

#include <type_traits>

#include "mfma_common.h"

namespace asx {

namespace {

typedef float v16f32 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// Packed bf16 image: tiles of 128 rows; fragment (T, ks, s) = 64 lanes x 8 bf16 (16 B per lane):
//   P[((T * KS + ks) * 4 + s) * 64 + l][j] = bf16( unit(R)[128 T + 32 s + (l & 31)][16 ks + 8 (l >> 5) + j] )
// which is the A / B operand of v_mfma_f32_32x32x16_bf16 as it stands (cdna_hip_programming.md, operand lane maps).
__global__ void pack_bf16_kernel(const double *__restrict__ R, const double *__restrict__ norms,
                                 const uint32_t *__restrict__ row_ids, uint64_t row0, uint64_t n, uint32_t d, uint32_t KS,
                                 uint64_t total_frag_lanes, bf16x8 *__restrict__ P /* hi plane, then lo plane per tile */) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total_frag_lanes) return;
  const uint32_t l = g & 63;
  const uint32_t s = (g >> 6) & 3;
  const uint64_t tk = g >> 8;
  const uint32_t ks = (uint32_t)(tk % KS);
  const uint64_t T = tk / KS;
  const uint64_t row = T * kTileRows + s * 32 + (l & 31);
  const uint32_t k0 = ks * 16 + 8 * (l >> 5);
  bf16x8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) hi[j] = lo[j] = (__bf16)0.0f;
  if (row < n) {
    const uint64_t src = row_ids ? row_ids[row] : row0 + row;
    const double nr = norms[src];
    if (nr > 0.0) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k0 + j < d) {
          const float x = (float)(R[src * d + k0 + j] / nr);
          hi[j] = (__bf16)x;
          lo[j] = (__bf16)(x - (float)hi[j]);
        }
    }
  }
  // tile T: [hi plane: KS*4*64 vectors][lo plane: the same]
  const uint64_t per_tile = (uint64_t)KS * 4 * 64;
  const uint64_t in_tile = g - T * per_tile;
  P[T * 2 * per_tile + in_tile] = hi;
  P[T * 2 * per_tile + per_tile + in_tile] = lo;
}

__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(size_t)(lds_void *)p; }

// LDS fragment reads issued and waited for by hand (see mfma_join_bf16.hip)
template <int OFF>
__device__ __forceinline__ void lds_read16(bf16x8 &dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int PENDING>
__device__ __forceinline__ void lds_wait(bf16x8 &x, bf16x8 &y) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x), "+v"(y) : "n"(PENDING));
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
struct SearchTerm {
  const float *a_lam, *s_lam;
  float alpha, beta;
};
template <int KS, bool SEARCH>
__global__ __launch_bounds__(512, 2) void theta_bf16_kernel(const bf16x8 *__restrict__ PA, uint32_t a_supers,
                                                            uint32_t row_count, const bf16x8 *__restrict__ PS,
                                                            uint32_t s_tiles, uint32_t m, float *__restrict__ theta_out,
                                                            uint32_t m2, float *__restrict__ theta2_out,
                                                            uint32_t *__restrict__ unit_counter, SearchTerm st, uint32_t slices,
                                                            float *__restrict__ part) {
  // slices > 1: a unit is (super tile, slice of the sample) -- few rows against a long sample (a batch of queries: 40 super
  // tiles on 256 CUs) -- and leaves its 64 sub-sample maxima per row in part[row][slice * 64 ..] for theta_select_kernel:
  // more, smaller sub-samples, so the m-th largest of them is at least what one unit per super tile finds and never above
  // the m-th best sampled value.
  constexpr int kPlaneVecs = KS * 4 * 64;   // 16-B vectors per plane of a 128-row tile
  constexpr int kTileVecs = 2 * kPlaneVecs;  // hi plane + lo plane
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16x8 *bufB = reinterpret_cast<bf16x8 *>(smem);                               // 2 x tile
  float *s_max = reinterpret_cast<float *>(bufB);  // [8 waves][32 rows][65], aliases the ring once the sweep is over
  __shared__ uint32_t s_unit;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lh = lane >> 5, lc = lane & 31;

  // 128-row sample tile J -> LDS: 1 KB pieces, KS per wave; part < 0: all, else the pieces of that quarter
  auto stage = [&](uint32_t J, int buf, int part) {
    const bf16x8 *src = PS + (size_t)J * kTileVecs;
    bf16x8 *dst = bufB + (size_t)buf * kTileVecs;
#pragma unroll
    for (int p = 0; p < KS; ++p)
      if (part < 0 || (p & 3) == part)
        __builtin_amdgcn_global_load_lds((gbl_void *)(src + (wv * KS + p) * 64 + lane), (lds_void *)(dst + (wv * KS + p) * 64),
                                         16, 0, 0);
  };

  for (;;) {
    __syncthreads();
    if (tid == 0) s_unit = atomicAdd(unit_counter, 1u);
    __syncthreads();
    const uint32_t unit = s_unit;
    if (unit >= a_supers * slices) break;
    const uint32_t I = unit / slices, sl = unit - I * slices;
    const uint32_t J_begin = (uint32_t)((uint64_t)(s_tiles / 2) * sl / slices) * 2, J_end = (uint32_t)((uint64_t)(s_tiles / 2) * (sl + 1) / slices) * 2;

    bf16x8 a_hi[KS], a_lo[KS];
    {
      const int row0 = wv * 32;
      const size_t T = (size_t)2 * I + (row0 >> 7);
      const int s0 = (row0 & 127) >> 5;
      const bf16x8 *a_src = PA + T * kTileVecs + (size_t)s0 * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        a_hi[ks] = a_src[(size_t)ks * 256];
        a_lo[ks] = a_src[kPlaneVecs + (size_t)ks * 256];
      }
    }
    float rmax[2][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rmax[0][r] = rmax[1][r] = -__builtin_inff();
    float lq[SEARCH ? 16 : 1];  // lambdas of this lane's 16 rows
    float lx_prev = 0.f;        // lambda of the column of the block whose maxima are pending
    if (SEARCH) {
#pragma unroll
      for (int r = 0; r < 16; ++r) lq[r] = st.a_lam[I * 256 + wv * 32 + 4 * lh + (r & 3) + 8 * (r >> 2)];
    }
    // what a pending product contributes: the product itself, or the score built on it
    auto value = [&](float sdot, int r) -> float {
      if (!SEARCH) return sdot;
      const float t = __builtin_fminf(__builtin_fabsf(lq[SEARCH ? r : 0] - lx_prev), 1.f);
      return __builtin_fmaf(-st.beta, t, __builtin_fmaf(st.alpha, sdot, st.beta));
    };

    bf16x8 b_hi[2], b_lo[2];
#define AS_READ(SET, KSTEP, NN)                                                   \
  {                                                                               \
    lds_read16<((KSTEP)*4 + (NN)) * 1024>(b_hi[SET], b_addr);                     \
    lds_read16<(kPlaneVecs + ((KSTEP)*4 + (NN)) * 64) * 16>(b_lo[SET], b_addr);   \
  }
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

    v16f32 accA, accB;
#pragma unroll
    for (int r = 0; r < 16; ++r) accA[r] = accB[r] = -__builtin_inff();  // nothing pending

    __syncthreads();  // the selection of the unit before is done with s_max (it aliases the ring)
    if (J_begin < J_end) stage(J_begin, 0, -1);
    __syncthreads();
    const uint32_t b_addr0 = lds_addr(bufB) + (uint32_t)lane * 16u;
    for (uint32_t J = J_begin; J < J_end; J += 2) {
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
    _Pragma("unroll") for (int i = 0; i < 3 * KH; ++i) {            \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            \
      __builtin_amdgcn_sched_group_barrier(0x002, kValu, 0);        \
    }                                                               \
    if ((MORE) && wv >= 4) stage(JN, (PAR) ^ 1, NN);                \
    contract(b_addr, AS_IC(NN), AS_IC(NEXT), AS_IC(KH), AS_IC(KS), CUR); \
    if (SEARCH) {                                                   \
      _Pragma("unroll") for (int i = 0; i < 3 * (KS - KH); ++i) {   \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          \
        __builtin_amdgcn_sched_group_barrier(0x002, kValu, 0);      \
      }                                                             \
    }                                                               \
    pin16(rmax[PREV_PAR]);  /* the maxima are taken by here: PREV's registers are free for the next block */ \
    if (SEARCH) lx_prev = lx[NN];                                   \
  }
      {  // even tile J (buffer 0, bin 0); tile J + 1 always exists
        const uint32_t b_addr = b_addr0;
        float lx[4] = {0.f, 0.f, 0.f, 0.f};
        if (SEARCH) {
#pragma unroll
          for (int nn = 0; nn < 4; ++nn) lx[nn] = st.s_lam[(size_t)J * kTileRows + nn * 32 + lc];
        }
        AS_READ(0, 0, 0)
        AS_BLOCK(0, 0, 1, accA, accB, 1, true, J + 1)
        AS_BLOCK(0, 1, 2, accB, accA, 0, true, J + 1)
        AS_BLOCK(0, 2, 3, accA, accB, 0, true, J + 1)
        AS_BLOCK(0, 3, -1, accB, accA, 0, true, J + 1)
      }
      __syncthreads();  // tile J + 1 has landed; everyone is done with buffer 0
      {  // odd tile J + 1 (buffer 1, bin 1)
        const bool more = J + 2 < J_end;
        const uint32_t b_addr = b_addr0 + kTileVecs * 16;
        float lx[4] = {0.f, 0.f, 0.f, 0.f};
        if (SEARCH) {
#pragma unroll
          for (int nn = 0; nn < 4; ++nn) lx[nn] = st.s_lam[(size_t)(J + 1) * kTileRows + nn * 32 + lc];
        }
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
    for (int r = 0; r < 16; ++r) rmax[1][r] = __builtin_fmaxf(rmax[1][r], value(accB[r], r));

    // 64 sub-sample maxima per row -> m-th largest.  s_max[wave][row][parity * 32 + lane column]
    float *wm = s_max + (size_t)wv * 32 * 65;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      wm[((r & 3) + 8 * (r >> 2) + 4 * lh) * 65 + lc] = rmax[0][r];
      wm[((r & 3) + 8 * (r >> 2) + 4 * lh) * 65 + 32 + lc] = rmax[1][r];
    }
    wave_sync();
    if (slices > 1) {  // (wave-uniform) the maxima of this slice, 64 floats per row side by side
      for (int r = 0; r < 32; ++r) part[((size_t)I * 256 + wv * 32 + r) * (64 * slices) + sl * 64 + lane] = wm[r * 65 + lane];
    } else if (lane < 32) {
      const uint32_t row = I * 256 + wv * 32 + lane;
      float vals[64];
#pragma unroll
      for (int c = 0; c < 64; ++c) vals[c] = wm[lane * 65 + c];
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
      if (row < row_count && theta2_out) theta2_out[row] = t;
      t = t1;
      if (row < row_count) theta_out[row] = t;
    }
    wave_sync();
  }
}

// theta[row] = the m-th largest (theta2: the max(m, m2)-th) of the `count` (a multiple of 64) sub-sample maxima of a row; a wave per row
__global__ __launch_bounds__(256) void theta_select_kernel(const float *__restrict__ part, uint32_t count, uint32_t row_count, uint32_t m,
                                                           float *__restrict__ theta_out, uint32_t m2, float *__restrict__ theta2_out) {
  const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= row_count) return;
  constexpr int kMine = 8;  // count <= 512
  float v[kMine];
#pragma unroll
  for (int j = 0; j < kMine; ++j) v[j] = (uint32_t)j * 64 + lane < count ? part[(size_t)row * count + (uint32_t)j * 64 + lane] : -__builtin_inff();
  const uint32_t mm = theta2_out ? max(m, m2) : m;
  float t = -__builtin_inff(), t1 = -__builtin_inff();
  for (uint32_t k = 0; k < mm; ++k) {  // remove the current maximum (one instance of it) mm - 1 times
    float best = v[0];
    int at = 0;
#pragma unroll
    for (int j = 1; j < kMine; ++j) {
      const bool gt = v[j] > best;
      best = gt ? v[j] : best;
      at = gt ? j : at;
    }
    float wbest = best;
    for (int o = 32; o > 0; o >>= 1) wbest = __builtin_fmaxf(wbest, __shfl_xor(wbest, o, 64));
    const unsigned long long holders = __ballot(best == wbest);
    t = wbest;
    if (k + 1 == m) t1 = wbest;
    if (holders == 0) break;  // (NaN: cannot happen for finite images; leave the loop rather than spin)
    if (lane == (uint32_t)__builtin_ctzll(holders)) {
#pragma unroll
      for (int j = 0; j < kMine; ++j) v[j] = j == at ? -__builtin_inff() : v[j];
    }
  }
  if (lane == 0) {
    if (theta2_out) theta2_out[row] = t;
    theta_out[row] = t1;
  }
}

}  // namespace

// 16-B vectors of a split image (hi plane + lo plane per tile)
uint64_t packed_vecs_bf16(uint32_t tiles, uint32_t d) { return (uint64_t)tiles * ((d + 15) / 16) * 4 * 64 * 2; }

int pack_bf16(as_ctx *ctx, const double *R, const double *norms, const uint32_t *row_ids, uint64_t row0, uint64_t n,
                     uint32_t d, void *P, uint32_t tiles) {
  const uint32_t KS = (d + 15) / 16;
  const uint64_t total = packed_vecs_bf16(tiles, d) / 2;  // one thread per fragment lane (writes hi and lo)
  ProfScope ps(ctx, "pack_unit_rows");
  hipLaunchKernelGGL(pack_bf16_kernel, dim3(cdiv(total, 256)), dim3(256), 0, ctx->stream, R, norms, row_ids, row0, n, d, KS,
                     total, (bf16x8 *)P);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

// one sweep: thresholds of the rows of image PA (a_supers super tiles) against the sample image PS
static int launch_theta(as_ctx *ctx, uint32_t KS, const void *PA, uint32_t a_supers, uint32_t row_count, const void *PS,
                        uint32_t s_tiles, uint32_t m, float *theta_out, uint32_t m2, float *theta2_out, const SearchTerm *st,
                        const char *prof_name) {
  DevBuf<uint32_t> unitc;
  AS_TRY(unitc.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(unitc.p, 0, sizeof(uint32_t), ctx->stream));
  const size_t lds = std::max((size_t)2 * 2 * KS * 4 * 64 * 16, (size_t)8 * 32 * 65 * sizeof(float));
  // few super tiles against a long sample: the sample is cut into slices, a unit per (super tile, slice)
  uint32_t slices = 1;
  if (st && a_supers < 128) slices = std::max(1u, std::min({8u, 256u / std::max(1u, a_supers), s_tiles / 16u}));
  DevBuf<float> part;
  if (slices > 1) AS_TRY(part.alloc(ctx, (size_t)a_supers * 256 * 64 * slices));
  const unsigned grid = std::min<uint32_t>(a_supers * slices, 512u);
  const SearchTerm none{nullptr, nullptr, 0.f, 0.f};
  ProfScope ps(ctx, prof_name);
#define AS_THETA1(K, S)                                                                                               \
  AS_HIP(ctx, hipFuncSetAttribute((const void *)theta_bf16_kernel<K, S>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                  (int)lds));                                                                         \
  hipLaunchKernelGGL((theta_bf16_kernel<K, S>), dim3(grid), dim3(512), lds, ctx->stream, (const bf16x8 *)PA, a_supers, \
                     row_count, (const bf16x8 *)PS, s_tiles, m, theta_out, m2, theta2_out, unitc.p, st ? *st : none, slices, part.p)
#define AS_THETA(K)                                                                                                    \
  case K:                                                                                                              \
    if (st) { AS_THETA1(K, true); } else { AS_THETA1(K, false); }                                                      \
    break;
  switch (KS) {
    AS_THETA(1) AS_THETA(2) AS_THETA(3) AS_THETA(4) AS_THETA(5) AS_THETA(6) AS_THETA(7) AS_THETA(8)
    default: return fail(ctx, AS_ERR_UNSUPPORTED, "bf16 threshold pass: dim not covered");
  }
#undef AS_THETA
#undef AS_THETA1
  if (slices > 1)
    hipLaunchKernelGGL(theta_select_kernel, dim3(cdiv(row_count, 4)), dim3(256), 0, ctx->stream, (const float *)part.p, 64u * slices, row_count, m,
                       theta_out, m2, theta2_out);
  AS_HIP(ctx, hipGetLastError());
  // (no synchronisation to hand the workspace back: a block returned to the cache is reused by later work on this same stream only)
  return AS_OK;
}

// theta_out[r], r < row_count, for rows [row_begin, row_begin + row_count) against the sample `sample_ids`
// (ns global row ids, ns a multiple of 256).  dim <= 128 (the split A fragments stay in registers);
// AS_ERR_UNSUPPORTED beyond.
int join_thresholds_bf16(as_ctx *ctx, const double *rows, const double *norms, uint32_t d, uint32_t row_begin,
                         uint32_t row_count, const uint32_t *sample_ids, uint32_t ns, uint32_t m, float *theta_out,
                         uint32_t m2, float *theta2_out) {
  const uint32_t KS = (d + 15) / 16;
  if (!(KS >= 1 && KS <= 8)) return fail(ctx, AS_ERR_UNSUPPORTED, "bf16 threshold pass: dim not covered");
  if (m < 1 || m > 32 || ns < 256 || ns % 256 != 0)
    return fail(ctx, AS_ERR_UNSUPPORTED, "bf16 threshold pass: the sample must be a multiple of 256 rows");
  const uint32_t a_tiles = (row_count + kTileRows - 1) / kTileRows, a_alloc = (a_tiles + 1) & ~1u;
  const uint32_t s_tiles = ns / kTileRows;
  DevBuf<unsigned char> PA, PS;
  AS_TRY(PA.alloc(ctx, packed_vecs_bf16(a_alloc, d) * 16));
  AS_TRY(PS.alloc(ctx, packed_vecs_bf16(s_tiles, d) * 16));
  AS_TRY(pack_bf16(ctx, rows, norms, nullptr, row_begin, row_count, d, PA.p, a_alloc));
  AS_TRY(pack_bf16(ctx, rows, norms, sample_ids, 0, ns, d, PS.p, s_tiles));
  return launch_theta(ctx, KS, PA.p, a_alloc / 2, row_count, PS.p, s_tiles, m, theta_out, m2, theta2_out, nullptr,
                      "knn_join_thresholds");
}

// Search form: score_out[q] = the m-th largest approximate search score of query q over the sampled items.
// PQ: split image of the nq queries (even tile count); items / sample as above; q_lam (padded to whole super
// tiles) and s_lam (ns): f32 lambdas.
int search_thresholds_bf16(as_ctx *ctx, const void *PQ, uint32_t nq, uint32_t d, const double *items, const double *norms,
                           const uint32_t *sample_ids, uint32_t ns, uint32_t m, const float *q_lam, const float *s_lam,
                           float alpha, float beta, float *score_out, uint32_t m2, float *score2_out) {
  const uint32_t KS = (d + 15) / 16;
  if (!(KS >= 1 && KS <= 8)) return fail(ctx, AS_ERR_UNSUPPORTED, "bf16 threshold pass: dim not covered");
  if (m < 1 || m > 32 || m2 > 48 || ns < 256 || ns % 256 != 0)
    return fail(ctx, AS_ERR_UNSUPPORTED, "bf16 threshold pass: the sample must be a multiple of 256 rows");
  const uint32_t a_tiles = (nq + kTileRows - 1) / kTileRows, a_alloc = (a_tiles + 1) & ~1u;
  const uint32_t s_tiles = ns / kTileRows;
  DevBuf<unsigned char> PS;
  AS_TRY(PS.alloc(ctx, packed_vecs_bf16(s_tiles, d) * 16));
  AS_TRY(pack_bf16(ctx, items, norms, sample_ids, 0, ns, d, PS.p, s_tiles));
  const SearchTerm st{q_lam, s_lam, alpha, beta};
  return launch_theta(ctx, KS, PQ, a_alloc / 2, nq, PS.p, s_tiles, m, score_out, m2, score2_out, &st, "search_thresholds");
}

}  // namespace asx
