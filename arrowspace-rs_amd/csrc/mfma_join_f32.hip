// mfma_join_f32.hip -- symmetric threshold join: the f32 MFMA candidate stage of K2 when the A side
// and the B side are the same rows (the item graph: `CosinePair::with_top_k` over all rows,
// src/laplacian.rs:211-246).
//
// cos(i, j) = cos(j, i), and the f32 MFMA product of two unit rows is bitwise symmetric (the same
// products accumulated in the same k order), so every unordered pair of row tiles is contracted ONCE
// and its products serve both sides.  Running top-k lists cannot be kept for the column side (those
// rows belong to other workgroups), so candidates are selected against a per-row threshold
// theta[r] fixed beforehand (an order statistic of the row's products with a 1/16 sample of the
// rows, mfma_drivers.hip): every pair with s(i, j) >= theta[i] is appended to the candidate list of
// i, and with s(i, j) >= theta[j] to the list of j.  What is NOT in row r's list therefore has
// s < theta[r], which is exactly what the certification of the exact re-rank needs; rows whose
// list overflows or whose K-th exact neighbour is not clear of theta[r] + delta fall through to the
// next stage, as in the top-k form (mfma_topk_f32.hip).
//
// Geometry: as mfma_topk_f32.hip with 8 waves x 32 rows (256 A rows per workgroup, A fragments in
// registers, B tiles through a 2-deep LDS-DMA ring, 32x32x2 f32 MFMA).  The tile pairs follow a
// cyclic half pattern: row super-tile I (256 rows) meets column super-tiles I, I+1, ..., I+(T2-1)/2
// (mod T2; for even T2 the antipodal one only from the smaller index), so every I carries the same
// work, whichever rank owns it.  The epilogue is two compares per product against thresholds held
// in registers; only waves that see a passing product walk their accumulators again.
#include "mfma_common.h"

namespace asx {

namespace {

typedef float v16f32 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

constexpr int kKs2Unit = 16;                    // k-steps (2 k each) per packed 16 KB unit
constexpr int kUnitFloats = kKs2Unit * 4 * 64;  // 4096 floats = 16 KB
constexpr int kSuper = 256;                     // rows per super tile (two packed 128-row tiles)

template <int CPB>
__device__ __forceinline__ void stage_chunk(const float *src, float *dst_lds, int wv, int lane) {
  constexpr int PER = 16 * CPB / 8;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int piece = wv * PER + q;
    __builtin_amdgcn_global_load_lds((gbl_void *)(src + piece * 256 + lane * 4), (lds_void *)(dst_lds + piece * 256), 16, 0,
                                     0);
  }
}

// a list slot is two words: the other row and the approximate score that put it there
__device__ __forceinline__ void append(uint32_t *__restrict__ cnt, uint32_t *__restrict__ cand, uint32_t cap,
                                       uint32_t target, uint32_t other, float score) {
  const uint32_t pos = atomicAdd(cnt + target, 1u);
  if (pos < cap) reinterpret_cast<uint2 *>(cand)[(size_t)target * cap + pos] = make_uint2(other, __float_as_uint(score));
}

// P = packed unit-row image of ALL rows (tiles of 128 rows, even tile count, zero padded).
// Work unit u = c * I_count + (I - I_begin): super tile I against the W column super tiles of block c
// (those of them that fall into I's half pattern: one contiguous run of offsets).  Units are dealt
// column block by column block, so the workgroups running at any one time read the same ~2 MB of
// the B image (L2-resident per XCD) instead of 256 different places of it.
template <int NC, int CPB>
__global__ __launch_bounds__(512, 2) void pair_join_mfma32_kernel(
    const float *__restrict__ P, uint32_t T2, uint32_t I_begin, uint32_t I_count, uint32_t W, uint32_t panels, uint32_t n,
    const float *__restrict__ theta, uint32_t cap, uint32_t *__restrict__ cnt, uint32_t *__restrict__ cand,
    uint32_t *__restrict__ unit_counter, int dbg) {
  constexpr int KS2 = NC * kKs2Unit;
  constexpr int kKs2Chunk = kKs2Unit * CPB;
  constexpr int kChunkFloats = kUnitFloats * CPB;
  constexpr int NCB = NC / CPB;  // chunks (barriers) per 128-row B tile
  static_assert(NC % CPB == 0, "CPB must divide NC");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *bufB = reinterpret_cast<float *>(smem);  // 2 x chunk
  __shared__ uint32_t s_unit;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lh = lane >> 5, lc = lane & 31;
  const uint32_t n_norm = (T2 - 1) / 2;
  const uint32_t n_blocks = (T2 + W - 1) / W;
  const uint32_t total_units = n_blocks * I_count;

  for (;;) {
    __syncthreads();
    if (tid == 0) s_unit = atomicAdd(unit_counter, 1u);
    __syncthreads();
    const uint32_t u = s_unit;
    if (u >= total_units) break;
    const uint32_t c = u / I_count, I = I_begin + (u - c * I_count);
    const bool antipodal = (T2 % 2 == 0) && (I < T2 / 2) && T2 > 1;
    const uint32_t L = 1 + n_norm + (antipodal ? 1u : 0u);  // offsets of this super tile: 0 = diagonal
    // offsets p (column super tile = I + p mod T2) that fall into block c = [B0, B0 + Wc)
    const uint32_t B0 = c * W, Wc = min(W, T2 - B0);
    const uint32_t delta = B0 >= I ? B0 - I : B0 + T2 - I;  // offset of the block's first super tile
    uint32_t p_lo, p_hi;
    if (delta + Wc <= T2) {
      p_lo = delta;
      p_hi = min(delta + Wc, L);
    } else {  // the block straddles I: its tail is offsets 0 .. (W <= T2 / 4 keeps the head out of the pattern)
      p_lo = 0;
      p_hi = min(delta + Wc - T2, L);
    }
    if (p_hi <= p_lo) continue;
    const uint32_t steps = (p_hi - p_lo) * 2;  // 128-row B tiles to visit

    // A fragments of this wave's 32 rows (dim <= 128: one k panel, loaded once per unit; wider rows: re-read from
    // L2 once per (B tile, 128-wide k panel)), row thresholds of this lane's 16 rows
    float a[KS2];
    const float *a_src;
    {
      const int row0 = wv * 32;
      const size_t T = (size_t)2 * I + (row0 >> 7);
      const int s0 = (row0 & 127) >> 5;
      a_src = P + ((T * KS2 * panels) * 4 + s0) * 64 + lane;  // + (panel * KS2 + ks) * 256
    }
    auto load_a = [&](uint32_t panel) {
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) a[ks] = a_src[((size_t)panel * KS2 + ks) * 256];
    };
    if (panels == 1) load_a(0);
    const uint32_t row_base = I * kSuper + wv * 32 + 4 * lh;  // + (r & 3) + 8 * (r >> 2)
    float tr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) tr[r] = theta[row_base + (r & 3) + 8 * (r >> 2)];

    v16f32 acc[4];
#pragma unroll
    for (int nn = 0; nn < 4; ++nn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nn][r] = 0.f;

    auto tile_of = [&](uint32_t step) -> uint32_t {
      uint32_t S = I + p_lo + (step >> 1);
      S = S >= T2 ? S - T2 : S;
      return 2 * S + (step & 1u);
    };
    const uint32_t cpt = panels * NCB;  // chunks per B tile
    auto chunk_src = [&](uint32_t g) -> const float * {
      const uint32_t step = g / cpt, c = g - step * cpt;
      return P + ((size_t)tile_of(step) * cpt + c) * kChunkFloats;
    };
    const uint32_t n_chunks = steps * cpt;
    __syncthreads();
    stage_chunk<CPB>(chunk_src(0), bufB, wv, lane);
    __syncthreads();
    uint32_t g = 0;
    int pbuf = 0;
    for (uint32_t step = 0; step < steps; ++step) {
      const uint32_t tile = tile_of(step);
      const bool diag = (p_lo + (step >> 1)) == 0;
      // column thresholds of this lane's four columns (the diagonal block is row-side only)
      float tc[4];
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) tc[nn] = diag ? __builtin_inff() : theta[tile * kTileRows + nn * 32 + lc];
      for (uint32_t panel = 0; panel < panels; ++panel) {
      if (panels > 1) load_a(panel);
#pragma unroll
      for (int c = 0; c < NCB; ++c, ++g) {
        if (g + 1 < n_chunks) stage_chunk<CPB>(chunk_src(g + 1), bufB + (pbuf ^ 1) * kChunkFloats, wv, lane);
        const float *bb = bufB + pbuf * kChunkFloats;
#pragma unroll
        for (int ks = 0; ks < kKs2Chunk; ++ks) {
          float b[4];
#pragma unroll
          for (int nn = 0; nn < 4; ++nn) b[nn] = bb[(ks * 4 + nn) * 64 + lane];
#pragma unroll
          for (int nn = 0; nn < 4; ++nn)
            acc[nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c * kKs2Chunk + ks], b[nn], acc[nn], 0, 0, 0);
        }
        // LDS reads one MFMA group ahead of their consumers (as in mfma_topk_f32.hip)
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int i = 0; i < 2 * kKs2Chunk - 2; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __syncthreads();  // DMA of the next chunk landed; everyone is done reading bufB[pbuf]
        pbuf ^= 1;
      }
      }
      // ---- epilogue: compare against the row and the column thresholds ----
      if (dbg == 1) continue;  // diagnostic: contraction only (accumulators keep summing)
      // Fast pass: per 32x32 block, does any lane hold a product that reaches a threshold?  Slow pass (about
      // half of the blocks see one or two such products): the rows of the block are tested one by one with
      // wave-uniform branches, so only rows that really carry a hit run the append code.
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) {
        bool h = false;
#pragma unroll
        for (int r = 0; r < 16; ++r) h = h || (acc[nn][r] >= tr[r]) || (acc[nn][r] >= tc[nn]);
        if (dbg != 2 && __any(h)) {
          const uint32_t col = tile * kTileRows + nn * 32 + lc;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[nn][r];
            const bool pr = v >= tr[r], pc = v >= tc[nn];
            if (__any(pr || pc)) {
              const uint32_t row = row_base + (r & 3) + 8 * (r >> 2);
              const bool inside = row < n && col < n && row != col;
              if (inside && pr) append(cnt, cand, cap, row, col, v);
              if (inside && pc) append(cnt, cand, cap, col, row, v);
            }
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nn][r] = 0.f;
      }
    }
  }
}

}  // namespace

static size_t join_lds_bytes(int CPB) { return (size_t)2 * kUnitFloats * CPB * sizeof(float); }

// Threshold join over super tiles [I_begin, I_begin + I_count) of the T2 super tiles of the packed
// image P (all n rows).  cnt / cand (n x cap) are indexed by global row; cnt must be zeroed by the
// caller, unit_counter too.
int join_f32(as_ctx *ctx, const float *P, uint32_t n, uint32_t d, uint32_t I_begin, uint32_t I_count,
             const float *theta, uint32_t cap, uint32_t *cnt, uint32_t *cand, uint32_t *unit_counter,
             const char *prof_name) {
  const uint32_t units = packed_units(d);  // 32-wide units; whole 128-wide panels beyond 4
  if (units == 0) return fail(ctx, AS_ERR_UNSUPPORTED, "threshold join: dim == 0");
  const uint32_t b_tiles = (n + kTileRows - 1) / kTileRows;
  const uint32_t T2 = (b_tiles + 1) / 2;
  const int NC = units <= 4 ? (int)units : 4;
  const uint32_t panels = units <= 4 ? 1 : units / 4;
  // units staged per barrier: at 128-wide panels the whole tile (one barrier per tile, 2 x 64 KB of LDS: 1000 ms
  // against 1008 ms with a barrier per half tile; debug 8 selects the latter)
  const int CPB = NC == 4 ? (ctx->opt_debug == 8 ? 2 : 4) : (NC % 2 == 0 ? 2 : 1);
  const size_t lds = join_lds_bytes(CPB);
  // column super tiles per block: 16 x 2 x 64 KB = 2 MB of the B image in flight (the XCD's L2 holds 4 MB)
  // (wider rows: fewer super tiles per block, same bytes)
  const uint32_t W = std::max(1u, std::min(std::max(1u, 16u / panels), T2 / 4));
  const unsigned grid = std::min<uint32_t>(I_count, 256u);
  ProfScope ps(ctx, prof_name);
#define AS_JOIN(N, C)                                                                                             \
  AS_HIP(ctx, hipFuncSetAttribute((const void *)pair_join_mfma32_kernel<N, C>,                                    \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                         \
  hipLaunchKernelGGL((pair_join_mfma32_kernel<N, C>), dim3(grid), dim3(512), lds, ctx->stream, P, T2, I_begin,    \
                     I_count, W, panels, n, theta, cap, cnt, cand, unit_counter, (int)ctx->opt_debug)
  switch (NC * 10 + CPB) {
    case 11: AS_JOIN(1, 1); break;
    case 22: AS_JOIN(2, 2); break;
    case 31: AS_JOIN(3, 1); break;
    case 42: AS_JOIN(4, 2); break;
    case 44: AS_JOIN(4, 4); break;
    default: return fail(ctx, AS_ERR_UNSUPPORTED, "bad k-panel shape");
  }
#undef AS_JOIN
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

}  // namespace asx
