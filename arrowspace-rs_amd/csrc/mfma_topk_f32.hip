// mfma_topk_f32.hip -- f32-input MFMA candidate pass (v_mfma_f32_32x32x2_f32) of K2 / K9.
//
// First and fastest stage of the neighbour search / search scan: the f32 matrix pipe runs at
// 64 flop/clk/SIMD (155 TF measured, MI355X_MICROARCH.md), 4x the f64 MFMA rate.  Candidates only
// have to be a SUPERSET of the wanted top-K: the exact re-rank and the certification against
// delta = (d+4) 2^-24 (mfma_drivers.hip) make the final result bit-identical to the exact path;
// rows that cannot be certified at this precision go to the f64 pass.
//
// Geometry: a workgroup (4 waves, one per SIMD) owns 256 A rows (two packed 128-row tiles); wave w
// keeps the fragments of rows 64 w .. 64 w + 63 in registers for the whole sweep (2 x 64 k-steps at
// d = 128) and accumulates a 64 x 128 block per B tile in 8 independent 32x32 accumulators
// (128 VGPRs).  B tiles stream through a 2 x 16 KB LDS ring as linear copies of the packed image:
//   P[((T*KS2 + ks)*4 + s)*64 + l] = (float) unit(R)[128 T + 32 s + (l & 31)][2 ks + (l >> 5)]
// so each B fragment read is 256 contiguous bytes.  Top-KC lists (f32 value, index) per row live in
// LDS; the 128 accumulators per lane are filtered against per-row thresholds held in registers.
#include <type_traits>

#include "mfma_common.cuh"

namespace asx {

typedef float v16f32 __attribute__((ext_vector_type(16)));

constexpr int kKs2Chunk = 16;                     // k-steps (2 k each) per LDS chunk = 32 k values
constexpr int kChunkFloats = kKs2Chunk * 4 * 64;  // 4096 floats = 16 KB
constexpr int kRows32 = 256;                      // A rows per workgroup

__global__ void pack_f32_kernel(const double *__restrict__ R, const double *__restrict__ norms,
                                const uint32_t *__restrict__ row_ids, uint64_t row0, uint64_t n, uint32_t d,
                                uint32_t KS2, uint64_t total, float *__restrict__ P) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total) return;
  const uint32_t l = g & 63;
  const uint32_t s = (g >> 6) & 3;
  const uint64_t tk = g >> 8;
  const uint32_t ks = (uint32_t)(tk % KS2);
  const uint64_t T = tk / KS2;
  const uint64_t row = T * kTileRows + s * 32 + (l & 31);
  const uint32_t k = ks * 2 + (l >> 5);
  float v = 0.f;
  if (row < n && k < d) {
    const uint64_t src = row_ids ? row_ids[row] : row0 + row;
    const double nr = norms[src];
    v = nr > 0.0 ? (float)(R[src * d + k] / nr) : 0.f;
  }
  P[g] = v;
}

int pack_f32(as_ctx *ctx, const double *R, const double *norms, const uint32_t *row_ids, uint64_t row0, uint64_t n,
             uint32_t d, float *P, uint32_t tiles) {
  const uint32_t KS2 = (d + 31) / 32 * 16;
  const uint64_t total = packed_elems(tiles, d);
  ProfScope ps(ctx, "pack_unit_rows");
  hipLaunchKernelGGL(pack_f32_kernel, dim3(cdiv(total, 256)), dim3(256), 0, ctx->stream, R, norms, row_ids, row0, n,
                     d, KS2, total, P);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

// epilogues: approximate VALUE of one (row, col) product, higher = better
struct GraphEpi32 {
  __device__ __forceinline__ void load_row(uint32_t, float &) const {}
  __device__ __forceinline__ void load_col(uint32_t, uint32_t, float &) const {}
  __device__ __forceinline__ float value(float s, float, float) const { return s; }
};
struct SearchEpi32 {
  ARows ar;
  const double *q_lambdas;
  const double *x_lambdas;
  float alpha, beta;
  __device__ __forceinline__ void load_row(uint32_t arow, float &v) const { v = (float)q_lambdas[ar.gid(arow)]; }
  __device__ __forceinline__ void load_col(uint32_t col, uint32_t nB, float &v) const {
    v = col < nB ? (float)x_lambdas[col] : 0.f;
  }
  __device__ __forceinline__ float value(float s, float lq, float lx) const {
    const float lam = 1.f - fminf(fabsf(lq - lx), 1.f);
    return fmaf(alpha, s, beta * lam);
  }
};

// sorted (value desc, idx asc) list of KC entries in LDS
__device__ __forceinline__ void list_insert32(float *lv, uint32_t *li, int KC, int lane, float v, uint32_t idx) {
  wave_sync();
  const bool in = lane < KC;
  const float vp = in ? lv[lane] : -__builtin_inff();
  const uint32_t ip = in ? li[lane] : kNoIdx;
  const bool before = in && (vp > v || (vp == v && ip < idx));
  const int pos = __popcll(__ballot(before));
  if (pos >= KC) return;
  if (lane >= pos && lane + 1 < KC) {
    lv[lane + 1] = vp;
    li[lane + 1] = ip;
  }
  if (lane == pos) {
    lv[pos] = v;
    li[pos] = idx;
  }
  wave_sync();
}

// LDS-DMA copy of one 16 KB chunk: wave w moves 4 x 1 KB pieces (global_load_lds_dwordx4: per-lane
// global address, wave-uniform LDS base + 16 B per lane).
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
__device__ __forceinline__ void stage_chunk(const float *src, float *dst_lds, int wv, int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int piece = wv * 4 + q;  // 1 KB pieces 0..15
    __builtin_amdgcn_global_load_lds((gbl_void *)(src + piece * 256 + lane * 4), (lds_void *)(dst_lds + piece * 256), 16, 0,
                                     0);
  }
}

template <int NC, class Epi>
__global__ __launch_bounds__(256, 1) void pair_topk_mfma32_kernel(const float *__restrict__ PA, uint32_t a_items,
                                                                   const float *__restrict__ PB, uint32_t b_tiles,
                                                                   uint32_t nA, uint32_t nB, uint32_t KC,
                                                                   uint32_t splits, Epi epi,
                                                                   double *__restrict__ cand_key,
                                                                   uint32_t *__restrict__ cand_idx) {
  constexpr int KS2 = NC * kKs2Chunk;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *bufB = reinterpret_cast<float *>(smem);                  // 2 x 16 KB
  float *lvals = bufB + 2 * kChunkFloats;                         // [4][64][KC]
  uint32_t *lidx = reinterpret_cast<uint32_t *>(lvals + 4 * 64 * KC);
  float *qvals = reinterpret_cast<float *>(lidx + 4 * 64 * KC);   // [4][kQueueCap]
  uint32_t *qcols = reinterpret_cast<uint32_t *>(qvals + 4 * kQueueCap);
  uint32_t *qrows = qcols + 4 * kQueueCap;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int lh = lane >> 5;  // 0..1
  const int lc = lane & 31;
  float *wl = lvals + (size_t)wv * 64 * KC;
  uint32_t *wi = lidx + (size_t)wv * 64 * KC;
  float *qv = qvals + wv * kQueueCap;
  uint32_t *qc = qcols + wv * kQueueCap;
  uint32_t *qr = qrows + wv * kQueueCap;

  const uint32_t items = a_items * splits;
  for (uint32_t item = blockIdx.x; item < items; item += gridDim.x) {
    const uint32_t I = item / splits, sp = item - I * splits;
    const uint32_t j_lo = (uint32_t)(((uint64_t)b_tiles * sp) / splits);
    const uint32_t j_hi = (uint32_t)(((uint64_t)b_tiles * (sp + 1)) / splits);

    // A fragments of this wave's 64 rows (packed tile 2I + (wv >> 1), 32-row groups 2 (wv & 1) + m)
    float a[2][KS2];
    {
      const size_t T = (size_t)2 * I + (wv >> 1);
      const int s0 = 2 * (wv & 1);
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) {
#pragma unroll
        for (int m = 0; m < 2; ++m) a[m][ks] = PA[((T * KS2 + ks) * 4 + s0 + m) * 64 + lane];
      }
    }
    // row of (m, reg) inside the wave: 32 m + (reg & 3) + 8 (reg >> 2) + 4 lh
    float rowv[2][16], thr[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t arow = I * kRows32 + wv * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        rowv[m][r] = 0.f;
        if (arow < nA) epi.load_row(arow, rowv[m][r]);
        thr[m][r] = -__builtin_inff();
      }
    for (int e = lane; e < 64 * (int)KC; e += 64) {
      wl[e] = -__builtin_inff();
      wi[e] = kNoIdx;
    }
    wave_sync();

    v16f32 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const uint32_t n_chunks = (j_hi - j_lo) * NC;
    const float *src = PB + (size_t)j_lo * NC * kChunkFloats;
    // LDS-DMA staging: chunk g+1 is copied global -> LDS (linear image, 16 B per lane) while chunk g
    // is consumed; the one barrier per chunk both retires the DMA (vmcnt) and frees the buffer.
    __syncthreads();
    if (n_chunks > 0) stage_chunk(src, bufB, wv, lane);
    __syncthreads();
    uint32_t g = 0;
    int pbuf = 0;
    for (uint32_t J = j_lo; J < j_hi; ++J) {
      float colv[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) epi.load_col(J * kTileRows + n * 32 + lc, nB, colv[n]);
#pragma unroll
      for (int c = 0; c < NC; ++c, ++g) {
        if (g + 1 < n_chunks) stage_chunk(src + (size_t)(g + 1) * kChunkFloats, bufB + (pbuf ^ 1) * kChunkFloats, wv, lane);
        const float *bb = bufB + pbuf * kChunkFloats;
#pragma unroll
        for (int ks = 0; ks < kKs2Chunk; ++ks) {
          float b[4];
#pragma unroll
          for (int n = 0; n < 4; ++n) b[n] = bb[(ks * 4 + n) * 64 + lane];
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][c * kKs2Chunk + ks], b[n], acc[0][n], 0, 0, 0);
            acc[1][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][c * kKs2Chunk + ks], b[n], acc[1][n], 0, 0, 0);
          }
        }
        __syncthreads();  // DMA of the next chunk landed; everyone is done reading bufB[pbuf]
        pbuf ^= 1;
      }
      // ---- epilogue of column tile J: threshold filter; passing elements (a few per tile once the
      // lists are warm) are compacted into a per-wave queue and inserted by ONE drain loop.  All
      // control values are wave-uniform scalars; if a tile overflows the queue (warm-up only) the scan
      // resumes from the slot that did not fit, with the tightened thresholds.
      {
        const bool tail = (J + 1 == b_tiles) && (nB % kTileRows != 0);
        // one pass over the 128 accumulators of this lane, starting at slot `lim`; returns the slot to
        // resume from when the queue filled up (0xFFFFFFFF: done).  TAIL masks the padded columns.
        auto scan = [&](auto tail_c, uint32_t lim, uint32_t &qlen) -> uint32_t {
          constexpr bool TAIL = decltype(tail_c)::value;
          uint32_t resume = 0xFFFFFFFFu;
#pragma unroll
          for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
              const uint32_t base = (m * 4 + n) * 16;
              if (base + 15 >= lim) {
                const uint32_t col = J * kTileRows + n * 32 + lc;
                const bool colok = !TAIL || col < nB;
                float vv[16];
                uint64_t any = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                  vv[r] = epi.value(acc[m][n][r], rowv[m][r], colv[n]);
                  any |= __ballot(vv[r] > thr[m][r] && colok);
                }
                if (any) {
#pragma unroll
                  for (int r = 0; r < 16; ++r) {
                    if (base + r >= lim) {
                      const bool pass = vv[r] > thr[m][r] && colok;
                      const uint64_t mask = __ballot(pass);
                      if (mask) {
                        const uint32_t cnt = (uint32_t)__popcll(mask);
                        if (qlen + cnt > kQueueCap) {
                          resume = base + r;
                          lim = 0xFFFFFFFFu;  // skip the remaining slots of this pass
                        } else {
                          if (pass) {
                            const uint32_t pos = qlen + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                            qv[pos] = vv[r];
                            qc[pos] = col;
                            qr[pos] = (uint32_t)(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh);
                          }
                          qlen = __builtin_amdgcn_readfirstlane(qlen + cnt);
                        }
                      }
                    }
                  }
                }
              }
            }
          }
          return resume;
        };
        uint32_t cur = 0;
        for (;;) {
          uint32_t qlen = 0;
          const uint32_t resume = tail ? scan(std::true_type{}, cur, qlen) : scan(std::false_type{}, cur, qlen);
          if (qlen) {
            wave_sync();
            for (uint32_t e = 0; e < qlen; ++e) {
              const float v_s = qv[e];
              const uint32_t c_s = qc[e], r_s = qr[e];
              if (v_s > wl[r_s * KC + KC - 1]) list_insert32(wl + r_s * KC, wi + r_s * KC, (int)KC, lane, v_s, c_s);
            }
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
#pragma unroll
              for (int rr = 0; rr < 16; ++rr)
                thr[mm][rr] = wl[(mm * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * lh) * KC + KC - 1];
          }
          if (resume == 0xFFFFFFFFu) break;
          cur = resume;
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
      }
    }
    // ---- write this work item's candidate lists (key = -value, ascending = better) ----
    for (int e = lane; e < 64 * (int)KC; e += 64) {
      const int rowl = e / (int)KC, slot = e - rowl * (int)KC;
      const uint32_t arow = I * kRows32 + wv * 64 + rowl;
      if (arow < nA) {
        const size_t o = ((size_t)arow * splits + sp) * KC + slot;
        cand_key[o] = -(double)wl[e];
        cand_idx[o] = wi[e];
      }
    }
    wave_sync();
  }
}

template <class Epi>
static int launch_f32(as_ctx *ctx, int NC, const float *PA, uint32_t a_items, const float *PB, uint32_t b_tiles,
                      uint32_t nA, uint32_t nB, uint32_t KC, uint32_t splits, Epi epi, double *cand_key,
                      uint32_t *cand_idx, const char *prof_name) {
  const size_t lds = 2 * kChunkFloats * sizeof(float) + (size_t)4 * 64 * KC * (sizeof(float) + sizeof(uint32_t)) +
                     (size_t)4 * kQueueCap * (sizeof(float) + 2 * sizeof(uint32_t));
  const unsigned grid = std::min<uint32_t>(a_items * splits, 256u);
  ProfScope ps(ctx, prof_name);
#define AS_CAND(N)                                                                                                  \
  case N:                                                                                                           \
    AS_HIP(ctx, hipFuncSetAttribute((const void *)pair_topk_mfma32_kernel<N, Epi>,                                  \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                         \
    hipLaunchKernelGGL((pair_topk_mfma32_kernel<N, Epi>), dim3(grid), dim3(256), lds, ctx->stream, PA, a_items, PB, \
                       b_tiles, nA, nB, KC, splits, epi, cand_key, cand_idx);                                       \
    break;
  switch (NC) {
    AS_CAND(1)
    AS_CAND(2)
    AS_CAND(3)
    AS_CAND(4)
    default: return fail(ctx, AS_ERR_UNSUPPORTED, "dim > 128 on the MFMA path");
  }
#undef AS_CAND
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

// a_tiles counts 128-row packed tiles of the A image (allocated to an even count)
int candidates_f32(as_ctx *ctx, const CandProblem &pb, const float *PA, uint32_t a_tiles, const float *PB,
                   uint32_t b_tiles, const char *prof_name) {
  const int NC = (int)((pb.d + 31) / 32);
  const uint32_t a_items = (a_tiles + 1) / 2;
  if (pb.mode == 0)
    return launch_f32<GraphEpi32>(ctx, NC, PA, a_items, PB, b_tiles, pb.nA, pb.nB, pb.KC, pb.splits, GraphEpi32{},
                                  pb.cand_key, pb.cand_idx, prof_name);
  SearchEpi32 epi{pb.arows, pb.q_lambdas, pb.x_lambdas, (float)pb.alpha, (float)(1.0 - pb.alpha)};
  return launch_f32<SearchEpi32>(ctx, NC, PA, a_items, PB, b_tiles, pb.nA, pb.nB, pb.KC, pb.splits, epi, pb.cand_key,
                                 pb.cand_idx, prof_name);
}

}  // namespace asx
