// mfma_topk_f64.hip -- f64 MFMA candidate pass (v_mfma_f64_16x16x4_f64) of K2 / K9.
//
// Replaces the O(n^2 d) work inside smartcore's CosinePair (src/laplacian.rs:211,221,246) and
// the sequential scan of ArrowSpace::search_lambda_aware (src/core.rs:778-786).
//
// Rows are L2-normalised once and stored "fragment packed": for every tile of 128 rows, k-step ks
// (4 consecutive k) and 16-row group s, the 64 doubles one MFMA operand needs (lane l -> row
// 16 s + (l & 15), k = 4 ks + (l >> 4)) are contiguous.  A workgroup (4 waves, one per SIMD) owns a
// 128-row tile of A whose fragments stay in registers for the whole sweep (wave w: rows
// 32 w .. 32 w + 31) and streams the B tiles through LDS in 32 KB k-chunks that are plain linear
// copies of the packed image (fragment reads are 512 contiguous bytes: conflict free).  Each wave
// keeps a sorted top-KC list per row in LDS and filters the 64 accumulators per lane against the
// rows' current thresholds; passing elements go through a small queue to one insertion loop.
//
// On MI355X this instruction issues every 128 cycles per SIMD (39.3 TF at 2.4 GHz): this pass is
// the high-precision second stage behind the f32 pass (mfma_topk_f32.hip).
#include "mfma_common.h"

namespace asx {

typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int kKsChunk = 8;                       // k-steps per LDS chunk (32 k values)
constexpr int kChunkDoubles = kKsChunk * 8 * 64;  // 4096 doubles = 32 KB

// ---- packing ------------------------------------------------------------------------------
// P[((T*KS + ks)*8 + s)*64 + l] = unit(R)[T*128 + s*16 + (l&15)][ks*4 + (l>>4)], zero padded.
__global__ void pack_f64_kernel(const double *__restrict__ R, const double *__restrict__ norms,
                                const uint32_t *__restrict__ row_ids, uint64_t row0, uint64_t n, uint32_t d,
                                uint32_t KS, uint64_t total, double *__restrict__ P) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total) return;
  const uint32_t l = g & 63;
  const uint32_t s = (g >> 6) & 7;
  const uint64_t tk = g >> 9;
  const uint32_t ks = (uint32_t)(tk % KS);
  const uint64_t T = tk / KS;
  const uint64_t row = T * kTileRows + s * 16 + (l & 15);
  const uint32_t k = ks * 4 + (l >> 4);
  double v = 0.0;
  if (row < n && k < d) {
    const uint64_t src = row_ids ? row_ids[row] : row0 + row;
    const double nr = norms[src];
    v = nr > 0.0 ? R[src * d + k] / nr : 0.0;
  }
  P[g] = v;
}

uint32_t packed_units(uint32_t d) {
  const uint32_t u = (d + 31) / 32;
  return u <= 4 ? u : (u + 3) / 4 * 4;  // beyond 128 columns: whole 128-wide panels
}
uint64_t packed_elems(uint32_t tiles, uint32_t d) { return (uint64_t)tiles * kTileRows * packed_units(d) * 32; }

int pack_f64(as_ctx *ctx, const double *R, const double *norms, const uint32_t *row_ids, uint64_t row0, uint64_t n,
             uint32_t d, double *P, uint32_t tiles) {
  const uint32_t KS = packed_units(d) * 8;
  const uint64_t total = packed_elems(tiles, d);
  ProfScope ps(ctx, "pack_unit_rows");
  hipLaunchKernelGGL(pack_f64_kernel, dim3(cdiv(total, 256)), dim3(256), 0, ctx->stream, R, norms, row_ids, row0, n,
                     d, KS, total, P);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

// ---- epilogues: approximate key of one (row, col) product (ascending = better) ---------------
struct GraphEpi {
  __device__ __forceinline__ void load_row(uint32_t, double &) const {}
  __device__ __forceinline__ void load_col(uint32_t, uint32_t, double &) const {}
  __device__ __forceinline__ double key(double s, double, double) const { return -s; }
};

struct SearchEpi {
  ARows ar;
  const double *q_lambdas;
  const double *x_lambdas;
  double alpha, one_minus_alpha;
  __device__ __forceinline__ void load_row(uint32_t arow, double &v) const { v = q_lambdas[ar.gid(arow)]; }
  __device__ __forceinline__ void load_col(uint32_t col, uint32_t nB, double &v) const {
    v = col < nB ? x_lambdas[col] : 0.0;
  }
  __device__ __forceinline__ double key(double s, double lq, double lx) const {
    const double lam = 1.0 - fmin(fabs(lq - lx), 1.0);
    return -(alpha * s + one_minus_alpha * lam);
  }
};

// wave-cooperative insertion into a sorted (key asc, idx asc) list of KC entries held in LDS
__device__ __forceinline__ void list_insert(double *lk, uint32_t *li, int KC, int lane, double key, uint32_t idx) {
  wave_sync();
  const bool in = lane < KC;
  const double kp = in ? lk[lane] : __builtin_inf();
  const uint32_t ip = in ? li[lane] : kNoIdx;
  const bool before = in && lex_less(kp, ip, key, idx);
  const int pos = __popcll(__ballot(before));
  if (pos >= KC) return;
  if (lane >= pos && lane + 1 < KC) {
    lk[lane + 1] = kp;
    li[lane + 1] = ip;
  }
  if (lane == pos) {
    lk[pos] = key;
    li[pos] = idx;
  }
  wave_sync();
}

// One work item = (A row tile I, column split sp).  Candidates go to
// cand[(I*128 + r) * splits * KC + sp * KC + e].
template <int NC, class Epi>
__global__ __launch_bounds__(256, 1) void pair_topk_mfma_kernel(const double *__restrict__ PA, uint32_t a_tiles,
                                                                 const double *__restrict__ PB, uint32_t b_tiles,
                                                                 uint32_t nA, uint32_t nB, uint32_t KC,
                                                                 uint32_t splits, Epi epi,
                                                                 double *__restrict__ cand_key,
                                                                 uint32_t *__restrict__ cand_idx) {
  constexpr int KS = NC * kKsChunk;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *bufB = reinterpret_cast<double *>(smem);                  // 2 x 32 KB
  double *lkeys = bufB + 2 * kChunkDoubles;                         // [4][32][KC]
  double *qkeys = lkeys + 4 * 32 * KC;                              // [4][kQueueCap]
  uint32_t *lidx = reinterpret_cast<uint32_t *>(qkeys + 4 * kQueueCap);
  uint32_t *qcols = lidx + 4 * 32 * KC;
  uint32_t *qrows = qcols + 4 * kQueueCap;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int lq = lane >> 4;   // 0..3
  const int lc = lane & 15;
  double *wk = lkeys + (size_t)wv * 32 * KC;
  uint32_t *wi = lidx + (size_t)wv * 32 * KC;
  double *qk = qkeys + wv * kQueueCap;
  uint32_t *qc = qcols + wv * kQueueCap;
  uint32_t *qr = qrows + wv * kQueueCap;

  const uint32_t items = a_tiles * splits;
  for (uint32_t item = blockIdx.x; item < items; item += gridDim.x) {
    const uint32_t I = item / splits, sp = item - I * splits;
    const uint32_t j_lo = (uint32_t)(((uint64_t)b_tiles * sp) / splits);
    const uint32_t j_hi = (uint32_t)(((uint64_t)b_tiles * (sp + 1)) / splits);

    // A fragments of this wave's 32 rows: resident for the whole sweep
    double a[2][KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
        a[m][ks] = PA[(((size_t)I * KS + ks) * 8 + (2 * wv + m)) * 64 + lane];
    }
    // per-row epilogue data and thresholds (row of (m, r): 16 m + lq + 4 r)
    double rowv[2][4], thr[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t arow = I * kTileRows + wv * 32 + m * 16 + lq + 4 * r;
        rowv[m][r] = 0.0;
        if (arow < nA) epi.load_row(arow, rowv[m][r]);
        thr[m][r] = __builtin_inf();
      }
    for (int e = lane; e < 32 * (int)KC; e += 64) {
      wk[e] = __builtin_inf();
      wi[e] = kNoIdx;
    }
    wave_sync();

    v4f64 acc[2][8];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 8; ++n) acc[m][n] = (v4f64){0.0, 0.0, 0.0, 0.0};

    const uint32_t n_chunks = (j_hi - j_lo) * NC;
    const double *src = PB + (size_t)j_lo * NC * kChunkDoubles;
    // prologue: chunk 0 -> LDS buffer 0
    __syncthreads();
    if (n_chunks > 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const double2 v = reinterpret_cast<const double2 *>(src)[tid + 256 * q];
        reinterpret_cast<double2 *>(bufB)[tid + 256 * q] = v;
      }
    }
    __syncthreads();
    uint32_t g = 0;  // flat chunk counter
    int pbuf = 0;
    for (uint32_t J = j_lo; J < j_hi; ++J) {
      double colv[8];
#pragma unroll
      for (int n = 0; n < 8; ++n) epi.load_col(J * kTileRows + n * 16 + lc, nB, colv[n]);
#pragma unroll
      for (int c = 0; c < NC; ++c, ++g) {
        // prefetch the next chunk into registers; its latency hides behind the MFMAs below
        double2 stage[8];
        const bool more = g + 1 < n_chunks;
        if (more) {
          const double2 *nsrc = reinterpret_cast<const double2 *>(src + (size_t)(g + 1) * kChunkDoubles);
#pragma unroll
          for (int q = 0; q < 8; ++q) stage[q] = nsrc[tid + 256 * q];
        }
        const double *bb = bufB + pbuf * kChunkDoubles;
#pragma unroll
        for (int ks = 0; ks < kKsChunk; ++ks) {
          double b[8];
#pragma unroll
          for (int n = 0; n < 8; ++n) b[n] = bb[(ks * 8 + n) * 64 + lane];
#pragma unroll
          for (int n = 0; n < 8; ++n) {
            acc[0][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0][c * kKsChunk + ks], b[n], acc[0][n], 0, 0, 0);
            acc[1][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1][c * kKsChunk + ks], b[n], acc[1][n], 0, 0, 0);
          }
        }
        __syncthreads();  // everyone is done reading bufB[pbuf]
        if (more) {
          double2 *dst = reinterpret_cast<double2 *>(bufB + (pbuf ^ 1) * kChunkDoubles);
#pragma unroll
          for (int q = 0; q < 8; ++q) dst[tid + 256 * q] = stage[q];
        }
        __syncthreads();
        pbuf ^= 1;
      }
      {
        // ---- epilogue of column tile J: threshold filter, rare insertions ----
        // Passing elements are compacted into a small per-wave queue and inserted by ONE
        // drain loop; if a tile overflows the queue (warm-up only) the scan resumes from the
        // slot that did not fit, with the tightened thresholds.
        const bool tail = (J + 1 == b_tiles) && (nB % kTileRows != 0);
        uint32_t start_slot = 0;
        bool overflow;
        do {
          overflow = false;
          uint32_t qlen = 0;
#pragma unroll
          for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int n = 0; n < 8; ++n) {
              const uint32_t col = J * kTileRows + n * 16 + lc;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const uint32_t slot = (m * 8 + n) * 4 + r;
                if (!overflow && slot >= start_slot) {
                  const double key = epi.key(acc[m][n][r], rowv[m][r], colv[n]);
                  bool pass = key < thr[m][r];
                  if (tail) pass = pass && col < nB;
                  const uint64_t mask = __ballot(pass);
                  if (mask) {
                    const uint32_t cnt = __popcll(mask);
                    if (qlen + cnt > kQueueCap) {
                      overflow = true;
                      start_slot = slot;
                    } else {
                      if (pass) {
                        const uint32_t pos = qlen + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                        qk[pos] = key;
                        qc[pos] = col;
                        qr[pos] = (uint32_t)(m * 16 + lq + 4 * r);
                      }
                      qlen += cnt;
                    }
                  }
                }
              }
            }
          }
          if (qlen) {
            wave_sync();
            for (uint32_t e = 0; e < qlen; ++e) {
              const double k_s = qk[e];
              const uint32_t c_s = qc[e], r_s = qr[e];
              if (k_s < wk[r_s * KC + KC - 1]) list_insert(wk + r_s * KC, wi + r_s * KC, (int)KC, lane, k_s, c_s);
            }
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
#pragma unroll
              for (int rr = 0; rr < 4; ++rr) thr[mm][rr] = wk[(mm * 16 + lq + 4 * rr) * KC + KC - 1];
          }
        } while (overflow);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 8; ++n) acc[m][n] = (v4f64){0.0, 0.0, 0.0, 0.0};
      
      }
    }
    // ---- write this work item's candidate lists ----
    for (int e = lane; e < 32 * (int)KC; e += 64) {
      const int rowl = e / (int)KC, slot = e - rowl * (int)KC;
      const uint32_t arow = I * kTileRows + wv * 32 + rowl;
      if (arow < nA) {
        const size_t o = ((size_t)arow * splits + sp) * KC + slot;
        cand_key[o] = wk[e];
        cand_idx[o] = wi[e];
      }
    }
  }
}

static size_t lds_bytes_f64(uint32_t KC) {
  return 2 * kChunkDoubles * sizeof(double) + (size_t)4 * 32 * KC * (sizeof(double) + sizeof(uint32_t)) +
         (size_t)4 * kQueueCap * (sizeof(double) + 2 * sizeof(uint32_t));
}
bool fits_f64(uint32_t KC) { return lds_bytes_f64(KC) <= 160 * 1024; }

template <class Epi>
static int launch_f64(as_ctx *ctx, int NC, const double *PA, uint32_t a_tiles, const double *PB, uint32_t b_tiles,
                      uint32_t nA, uint32_t nB, uint32_t KC, uint32_t splits, Epi epi, double *cand_key,
                      uint32_t *cand_idx, const char *prof_name) {
  const size_t lds = lds_bytes_f64(KC);
  if (lds > 160 * 1024) return fail(ctx, AS_ERR_UNSUPPORTED, "candidate capacity too large for the f64 MFMA pass");
  const unsigned grid = std::min<uint32_t>(a_tiles * splits, 256u);
  ProfScope ps(ctx, prof_name);
#define AS_CAND(N)                                                                                               \
  case N:                                                                                                        \
    AS_HIP(ctx, hipFuncSetAttribute((const void *)pair_topk_mfma_kernel<N, Epi>,                                 \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                      \
    hipLaunchKernelGGL((pair_topk_mfma_kernel<N, Epi>), dim3(grid), dim3(256), lds, ctx->stream, PA, a_tiles, PB, \
                       b_tiles, nA, nB, KC, splits, epi, cand_key, cand_idx);                                    \
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

int candidates_f64(as_ctx *ctx, const CandProblem &pb, const double *PA, uint32_t a_tiles, const double *PB,
                   uint32_t b_tiles, const char *prof_name) {
  const int NC = (int)((pb.d + 31) / 32);
  if (pb.mode == 0)
    return launch_f64<GraphEpi>(ctx, NC, PA, a_tiles, PB, b_tiles, pb.nA, pb.nB, pb.KC, pb.splits, GraphEpi{},
                                pb.cand_key, pb.cand_idx, prof_name);
  SearchEpi epi{pb.arows, pb.q_lambdas, pb.x_lambdas, pb.alpha, 1.0 - pb.alpha};
  return launch_f64<SearchEpi>(ctx, NC, PA, a_tiles, PB, b_tiles, pb.nA, pb.nB, pb.KC, pb.splits, epi, pb.cand_key,
                               pb.cand_idx, prof_name);
}

}  // namespace asx
