// mfma_topk_f32.hip -- f32-input MFMA candidate pass (v_mfma_f32_32x32x2_f32) of K2 / K9.
//
// First and fastest stage of the neighbour search / search scan: the f32 matrix pipe runs at
// 64 flop/clk/SIMD (155 TF measured, MI355X_MICROARCH.md), 4x the f64 MFMA rate.  Candidates only
// have to be a SUPERSET of the wanted top-K: the exact re-rank and the certification against
// delta = (d+4) 2^-24 (mfma_drivers.hip) make the final result bit-identical to the exact path;
// rows that cannot be certified at this precision go to the f64 pass.
//
// Geometry (default, MW = 1): a workgroup of 8 waves (two per SIMD: one wave's epilogue overlaps the other's
// MFMAs) owns 256 A rows (two packed 128-row tiles); wave w keeps the fragments of rows 32 w .. 32 w + 31 in
// registers for the whole sweep (64 k-steps at d = 128) and accumulates a 32 x 128 block per B tile in four
// independent 32x32 accumulators.  (MW = 2: 4 waves x 64 rows, one wave per SIMD.)  B tiles stream through a 2-deep
// LDS ring as linear copies of the packed image:
//   P[((T*KS2 + ks)*4 + s)*64 + l] = (float) unit(R)[128 T + 32 s + (l & 31)][2 ks + (l >> 5)]
// so each B fragment read is 256 contiguous bytes.  Top-KC lists (f32 value, index) per row live in LDS; the
// accumulators of a lane are filtered against per-row thresholds read from LDS.
// This pass is the K9 search scan, the fall-back stage of the item graph and its form for sharded / small inputs;
// the item graph over a whole matrix takes the symmetric threshold join (mfma_join_f32.hip).
#include <type_traits>

#include "mfma_common.h"

namespace asx {

typedef float v16f32 __attribute__((ext_vector_type(16)));

constexpr int kKs2Unit = 16;                      // k-steps (2 k each) per packed 16 KB unit = 32 k values
constexpr int kUnitFloats = kKs2Unit * 4 * 64;    // 4096 floats = 16 KB
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
    if (norms) {
      const double nr = norms[src];
      v = nr > 0.0 ? (float)(R[src * d + k] / nr) : 0.f;
    } else {  // the rows as they are (Euclidean candidates of the clustering pre-stage)
      v = (float)R[src * d + k];
    }
  }
  P[g] = v;
}

int pack_f32(as_ctx *ctx, const double *R, const double *norms, const uint32_t *row_ids, uint64_t row0, uint64_t n,
             uint32_t d, float *P, uint32_t tiles) {
  const uint32_t KS2 = packed_units(d) * 16;
  const uint64_t total = packed_elems(tiles, d);
  ProfScope ps(ctx, "pack_unit_rows");
  hipLaunchKernelGGL(pack_f32_kernel, dim3(cdiv(total, 256)), dim3(256), 0, ctx->stream, R, norms, row_ids, row0, n,
                     d, KS2, total, P);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

// epilogues: approximate VALUE of one (row, col) product, higher = better
struct GraphEpi32 {
  using T = float;  // per-row / per-column epilogue data (unused here)
  __device__ __forceinline__ void load_row(uint32_t, T &) const {}
  __device__ __forceinline__ void load_col(uint32_t, uint32_t, T &) const {}
  __device__ __forceinline__ float value(float s, T, T) const { return s; }
};
// Search epilogue, f32 lambda term.  Used when max|lambda_q| + max|lambda_x| = Lsum is small
// (search_mfma checks): with lq, lx rounded to f32,
//   | lam32 - (1 - min(|lq - lx|, 1)) | <= Lsum 2^-23 + 2^-24,
// which the certification bound carries as |beta| (Lsum 2^-23 + 2^-22).
struct SearchEpi32 {
  using T = float;
  ARows ar;
  const double *q_lambdas;
  const double *x_lambdas;
  float alpha, beta;
  __device__ __forceinline__ void load_row(uint32_t arow, T &v) const { v = (float)q_lambdas[ar.gid(arow)]; }
  __device__ __forceinline__ void load_col(uint32_t col, uint32_t nB, T &v) const {
    v = col < nB ? (float)x_lambdas[col] : 0.f;
  }
  __device__ __forceinline__ float value(float s, T lq, T lx) const {
    const float lam = 1.f - fminf(fabsf(lq - lx), 1.f);
    return fmaf(alpha, s, beta * lam);
  }
};
// Search epilogue, f64 lambda term: lambdas of any magnitude (a large tau makes lambdas large), three
// times the epilogue cost.  Only the sum is rounded to f32:
// |value - exact| <= |alpha| delta_cos + (|alpha| + |beta|) 2^-23
struct SearchEpi32D {
  using T = double;
  ARows ar;
  const double *q_lambdas;
  const double *x_lambdas;
  double alpha, beta;
  __device__ __forceinline__ void load_row(uint32_t arow, T &v) const { v = q_lambdas[ar.gid(arow)]; }
  __device__ __forceinline__ void load_col(uint32_t col, uint32_t nB, T &v) const {
    v = col < nB ? x_lambdas[col] : 0.0;
  }
  __device__ __forceinline__ float value(float s, T lq, T lx) const {
    const double lam = 1.0 - fmin(fabs(lq - lx), 1.0);
    return (float)(alpha * (double)s + beta * lam);
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

// LDS-DMA copy of one chunk of CPB x 16 KB: the 1 KB pieces are dealt to the waves
// (global_load_lds_dwordx4: per-lane global address, wave-uniform LDS base + 16 B per lane).
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
template <int CPB, int WAVES>
__device__ __forceinline__ void stage_chunk(const float *src, float *dst_lds, int wv, int lane) {
  constexpr int PER = 16 * CPB / WAVES;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int piece = wv * PER + q;
    __builtin_amdgcn_global_load_lds((gbl_void *)(src + piece * 256 + lane * 4), (lds_void *)(dst_lds + piece * 256), 16, 0,
                                     0);
  }
}

// NC  = 16 KB units per B tile (d_pad / 32);  CPB = units staged and consumed per barrier;
// MW  = 32-row groups per wave: MW = 2 -> 4 waves x 64 rows (one wave per SIMD, 512 registers each),
//       MW = 1 -> 8 waves x 32 rows (two waves per SIMD: one wave's epilogue overlaps the other's MFMAs).
template <int NC, int CPB, int MW, class Epi>
__global__ __launch_bounds__(512 / MW, 2 / MW) void pair_topk_mfma32_kernel(
    const float *__restrict__ PA, uint32_t a_items, const float *__restrict__ PB, uint32_t b_tiles, uint32_t nA,
    uint32_t nB, uint32_t KC, uint32_t splits, uint32_t panels, Epi epi, double *__restrict__ cand_key,
    uint32_t *__restrict__ cand_idx, int dbg) {
  // panels = 128-wide k panels per row (1 for dim <= 128: the A fragments then stay in registers for
  // the whole sweep; more: they are re-read from L2 once per (B tile, panel))
  constexpr int WAVES = 8 / MW;
  constexpr int WROWS = 32 * MW;                  // rows per wave
  constexpr uint32_t QCAP = kQueueCap * MW / 2;   // per-wave queue entries
  constexpr int KS2 = NC * kKs2Unit;
  constexpr int kKs2Chunk = kKs2Unit * CPB;       // k-steps per staged chunk
  constexpr int kChunkFloats = kUnitFloats * CPB;
  constexpr int NCB = NC / CPB;                   // chunks (barriers) per B tile
  static_assert(NC % CPB == 0, "CPB must divide NC");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *bufB = reinterpret_cast<float *>(smem);                        // 2 x chunk
  float *lvals = bufB + 2 * kChunkFloats;                               // [256 rows][KC]
  uint32_t *lidx = reinterpret_cast<uint32_t *>(lvals + kRows32 * KC);
  float *qvals = reinterpret_cast<float *>(lidx + kRows32 * KC);        // [WAVES][QCAP]
  uint32_t *qcols = reinterpret_cast<uint32_t *>(qvals + WAVES * QCAP);
  uint32_t *qrows = qcols + WAVES * QCAP;
  using ET = typename Epi::T;
  ET *rowdata = reinterpret_cast<ET *>(qrows + WAVES * QCAP);   // [256] (8-byte aligned: all blocks above are multiples of 8 B)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int lh = lane >> 5;  // 0..1
  const int lc = lane & 31;
  float *wl = lvals + (size_t)wv * WROWS * KC;
  uint32_t *wi = lidx + (size_t)wv * WROWS * KC;
  float *qv = qvals + wv * QCAP;
  uint32_t *qc = qcols + wv * QCAP;
  uint32_t *qr = qrows + wv * QCAP;
  ET *wrow = rowdata + wv * WROWS;

  const uint32_t items = a_items * splits;
  for (uint32_t item = blockIdx.x; item < items; item += gridDim.x) {
    const uint32_t I = item / splits, sp = item - I * splits;
    const uint32_t j_lo = (uint32_t)(((uint64_t)b_tiles * sp) / splits);
    const uint32_t j_hi = (uint32_t)(((uint64_t)b_tiles * (sp + 1)) / splits);

    // A fragments of this wave's rows: rows [wv*WROWS, +WROWS) of the 256-row workgroup tile
    float a[MW][KS2];
    const float *a_src;
    {
      const int row0 = wv * WROWS;
      const size_t T = (size_t)2 * I + (row0 >> 7);
      const int s0 = (row0 & 127) >> 5;
      a_src = PA + ((T * KS2 * panels) * 4 + s0) * 64 + lane;  // + (panel*KS2 + ks)*256 + m*64
    }
    auto load_a = [&](uint32_t panel) {
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) {
#pragma unroll
        for (int m = 0; m < MW; ++m) a[m][ks] = a_src[((size_t)panel * KS2 + ks) * 256 + m * 64];
      }
    };
    if (panels == 1) load_a(0);
    if (lane < WROWS) {  // per-row epilogue data (search: the query lambdas), kept in LDS
      const uint32_t arow = I * kRows32 + wv * WROWS + lane;
      ET rv = 0;
      if (arow < nA) epi.load_row(arow, rv);
      wrow[lane] = rv;
    }
    for (int e = lane; e < WROWS * (int)KC; e += 64) {
      wl[e] = -__builtin_inff();
      wi[e] = kNoIdx;
    }
    wave_sync();

    v16f32 acc[MW][4];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const uint32_t n_chunks = (j_hi - j_lo) * panels * NCB;
    const float *src = PB + (size_t)j_lo * panels * NCB * kChunkFloats;
    // LDS-DMA staging: chunk g+1 is copied global -> LDS (linear image, 16 B per lane) while chunk g
    // is consumed; the one barrier per chunk both retires the DMA (vmcnt) and frees the buffer.
    __syncthreads();
    if (n_chunks > 0) stage_chunk<CPB, WAVES>(src, bufB, wv, lane);
    __syncthreads();
    uint32_t g = 0;
    int pbuf = 0;
    for (uint32_t J = j_lo; J < j_hi; ++J) {
      ET colv[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) epi.load_col(J * kTileRows + n * 32 + lc, nB, colv[n]);
      for (uint32_t panel = 0; panel < panels; ++panel) {
      if (panels > 1) load_a(panel);
#pragma unroll
      for (int c = 0; c < NCB; ++c, ++g) {
        if (g + 1 < n_chunks)
          stage_chunk<CPB, WAVES>(src + (size_t)(g + 1) * kChunkFloats, bufB + (pbuf ^ 1) * kChunkFloats, wv, lane);
        const float *bb = bufB + pbuf * kChunkFloats;
#pragma unroll
        for (int ks = 0; ks < kKs2Chunk; ++ks) {
          float b[4];
#pragma unroll
          for (int n = 0; n < 4; ++n) b[n] = bb[(ks * 4 + n) * 64 + lane];
#pragma unroll
          for (int n = 0; n < 4; ++n) {
#pragma unroll
            for (int m = 0; m < MW; ++m)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][c * kKs2Chunk + ks], b[n], acc[m][n], 0, 0, 0);
          }
        }
        // Schedule shape for the chunk: LDS reads run one MFMA group ahead of their consumers so the
        // LDS latency hides behind the matrix pipe (an in-order wave otherwise waits ~100 cycles per group).
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int i = 0; i < 2 * kKs2Chunk - 2; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2 * MW, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * MW, 0);
        __syncthreads();  // DMA of the next chunk landed; everyone is done reading bufB[pbuf]
        pbuf ^= 1;
      }
      }
      // ---- epilogue of column tile J: threshold filter; passing elements (a few per tile once the
      // lists are warm) are compacted into a per-wave queue and inserted by ONE drain loop.  All
      // control values are wave-uniform scalars; if a tile overflows the queue (warm-up only) the scan
      // resumes from the slot that did not fit, with the tightened thresholds.
      {
        const bool tail = (J + 1 == b_tiles) && (nB % kTileRows != 0);
        if (tail) {  // padded columns of the last tile can never win: overwrite their products once
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            if (J * kTileRows + n * 32 + lc >= nB) {
#pragma unroll
              for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = -__builtin_inff();
            }
          }
        }
        // one pass over this lane's accumulators starting at slot `lim`; returns the slot to resume
        // from when the queue filled up (0xFFFFFFFF: done)
        auto scan = [&](uint32_t lim, uint32_t &qlen) -> uint32_t {
          uint32_t resume = 0xFFFFFFFFu;
#pragma unroll
          for (int m = 0; m < MW; ++m) {
            // thresholds / row data of this lane's 16 rows of group m (LDS, 2 addresses per read)
            float thr[16];
            ET rowv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int rl = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
              thr[r] = wl[rl * KC + KC - 1];
              rowv[r] = wrow[rl];
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
              __builtin_amdgcn_sched_barrier(0);  // keep the accumulator reads of one group together
              const uint32_t base = (m * 4 + n) * 16;
              if (base + 15 >= lim) {
                const uint32_t col = J * kTileRows + n * 32 + lc;
                float vv[16];
                uint64_t any = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                  vv[r] = epi.value(acc[m][n][r], rowv[r], colv[n]);
                  any |= __ballot(vv[r] > thr[r]);
                }
                if (any) {
#pragma unroll
                  for (int r = 0; r < 16; ++r) {
                    if (base + r >= lim) {
                      const bool pass = vv[r] > thr[r];
                      const uint64_t mask = __ballot(pass);
                      if (mask) {
                        const uint32_t cnt = (uint32_t)__popcll(mask);
                        if (qlen + cnt > QCAP) {
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
          uint32_t resume = 0xFFFFFFFFu;
          if (dbg != 2) resume = scan(cur, qlen);
          if (dbg == 1) { qlen = 0; resume = 0xFFFFFFFFu; }  // diagnostic: filter only, no insertion
          if (qlen) {
            // drain: queue entries are pulled into registers 64 at a time (one per lane) and broadcast
            // with readlane, so the insertion loop has no dependent LDS read besides the list itself
            wave_sync();
            for (uint32_t e0 = 0; e0 < qlen; e0 += 64) {
              const uint32_t nb = min(64u, qlen - e0);
              float ev = 0.f;
              uint32_t ec = 0, er = 0;
              if ((uint32_t)lane < nb) {
                ev = qv[e0 + lane];
                ec = qc[e0 + lane];
                er = qr[e0 + lane];
              }
              for (uint32_t e = 0; e < nb; ++e) {
                const float v_s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ev), e));
                const uint32_t c_s = (uint32_t)__builtin_amdgcn_readlane((int)ec, e);
                const uint32_t r_s = (uint32_t)__builtin_amdgcn_readlane((int)er, e);
                list_insert32(wl + r_s * KC, wi + r_s * KC, (int)KC, lane, v_s, c_s);
              }
            }
          }
          if (resume == 0xFFFFFFFFu) break;
          cur = resume;
        }
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
      }
    }
    // ---- write this work item's candidate lists (key = -value, ascending = better) ----
    for (int e = lane; e < WROWS * (int)KC; e += 64) {
      const int rowl = e / (int)KC, slot = e - rowl * (int)KC;
      const uint32_t arow = I * kRows32 + wv * WROWS + rowl;
      if (arow < nA) {
        const size_t o = ((size_t)arow * splits + sp) * KC + slot;
        cand_key[o] = -(double)wl[e];
        cand_idx[o] = wi[e];
      }
    }
    wave_sync();
  }
}

static size_t lds_bytes_f32(uint32_t KC, int CPB) {
  return (size_t)2 * kUnitFloats * CPB * sizeof(float) + (size_t)4 * 64 * KC * (sizeof(float) + sizeof(uint32_t)) +
         (size_t)4 * kQueueCap * (sizeof(float) + 2 * sizeof(uint32_t)) + kRows32 * sizeof(double);
}

bool fits_f32(uint32_t KC) { return lds_bytes_f32(KC, 1) <= 160 * 1024; }

template <class Epi>
static int launch_f32(as_ctx *ctx, int NC, const float *PA, uint32_t a_items, const float *PB, uint32_t b_tiles,
                      uint32_t nA, uint32_t nB, uint32_t KC, uint32_t splits, uint32_t panels, Epi epi,
                      double *cand_key, uint32_t *cand_idx, const char *prof_name) {
  // two 16 KB units per barrier when the LDS budget allows (half the barriers per tile)
  const int CPB = (NC % 2 == 0 && lds_bytes_f32(KC, 2) <= 160 * 1024) ? 2 : 1;
  const size_t lds = lds_bytes_f32(KC, CPB);
  if (lds > 160 * 1024) return fail(ctx, AS_ERR_UNSUPPORTED, "candidate capacity too large for the f32 MFMA pass");
  const unsigned grid = std::min<uint32_t>(a_items * splits, 256u);
  const int MW = 1;  // (8 waves x 32 rows; the 4 x 64 form was never faster)
  ProfScope ps(ctx, prof_name);
#define AS_CAND(N, C)                                                                                                \
  do {                                                                                                               \
    if (MW == 2) {                                                                                                   \
      AS_HIP(ctx, hipFuncSetAttribute((const void *)pair_topk_mfma32_kernel<N, C, 2, Epi>,                           \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                        \
      hipLaunchKernelGGL((pair_topk_mfma32_kernel<N, C, 2, Epi>), dim3(grid), dim3(256), lds, ctx->stream, PA,       \
                         a_items, PB, b_tiles, nA, nB, KC, splits, panels, epi, cand_key, cand_idx,              \
                         (int)ctx->opt_debug);                                                                      \
    } else {                                                                                                         \
      AS_HIP(ctx, hipFuncSetAttribute((const void *)pair_topk_mfma32_kernel<N, C, 1, Epi>,                           \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                        \
      hipLaunchKernelGGL((pair_topk_mfma32_kernel<N, C, 1, Epi>), dim3(grid), dim3(512), lds, ctx->stream, PA,       \
                         a_items, PB, b_tiles, nA, nB, KC, splits, panels, epi, cand_key, cand_idx,              \
                         (int)ctx->opt_debug);                                                                      \
    }                                                                                                                \
  } while (0)
  switch (NC * 10 + CPB) {
    case 11: AS_CAND(1, 1); break;
    case 21: AS_CAND(2, 1); break;
    case 22: AS_CAND(2, 2); break;
    case 31: AS_CAND(3, 1); break;
    case 41: AS_CAND(4, 1); break;
    case 42: AS_CAND(4, 2); break;
    default: return fail(ctx, AS_ERR_UNSUPPORTED, "bad k-panel shape");
  }
#undef AS_CAND
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

// a_tiles counts 128-row packed tiles of the A image (allocated to an even count)
int candidates_f32(as_ctx *ctx, const CandProblem &pb, const float *PA, uint32_t a_tiles, const float *PB,
                   uint32_t b_tiles, const char *prof_name) {
  // dim <= 128: one panel of NC 32-wide units; larger: dim padded to a multiple of 128, 4 units per panel
  const uint32_t units = packed_units(pb.d);
  const int NC = units <= 4 ? (int)units : 4;
  const uint32_t panels = units <= 4 ? 1 : units / 4;
  const uint32_t a_items = (a_tiles + 1) / 2;
  if (pb.mode == 0)
    return launch_f32<GraphEpi32>(ctx, NC, PA, a_items, PB, b_tiles, pb.nA, pb.nB, pb.KC, pb.splits, panels,
                                  GraphEpi32{}, pb.cand_key, pb.cand_idx, prof_name);
  if (pb.lam_f32) {
    SearchEpi32 epi{pb.arows, pb.q_lambdas, pb.x_lambdas, (float)pb.alpha, (float)(1.0 - pb.alpha)};
    return launch_f32<SearchEpi32>(ctx, NC, PA, a_items, PB, b_tiles, pb.nA, pb.nB, pb.KC, pb.splits, panels, epi,
                                   pb.cand_key, pb.cand_idx, prof_name);
  }
  SearchEpi32D epi{pb.arows, pb.q_lambdas, pb.x_lambdas, pb.alpha, 1.0 - pb.alpha};
  return launch_f32<SearchEpi32D>(ctx, NC, PA, a_items, PB, b_tiles, pb.nA, pb.nB, pb.KC, pb.splits, panels, epi,
                                  pb.cand_key, pb.cand_idx, prof_name);
}

}  // namespace asx
