// mfma_topk.hip -- the MFMA leg of K2 (all-pairs cosine + per-row top-(topk+1)) and K9
// (query batch x items blended score + per-query top-k).
//
// Replaces the O(n^2 d) work inside smartcore's CosinePair (src/laplacian.rs:211,221,246) and
// the sequential scan of ArrowSpace::search_lambda_aware (src/core.rs:778-786).
//
// Scheme (exactness first):
//   1. CANDIDATES.  Rows are L2-normalised once and stored "fragment packed": for every tile of
//      128 rows, k-step ks (4 consecutive k) and 16-row group s, the 64 doubles one
//      v_mfma_f64_16x16x4_f64 operand needs (lane l -> row 16 s + (l & 15), k = 4 ks + (l >> 4))
//      are contiguous.  A workgroup (4 waves, one per SIMD) owns a 128-row tile of A whose
//      fragments stay in registers for the whole sweep (wave w: rows 32 w .. 32 w + 31), and
//      streams the B tiles through LDS in 32 KB k-chunks that are plain linear copies of the
//      packed image (fragment reads are 512 contiguous bytes: conflict free).  The n x n product
//      is never materialised: each wave keeps a sorted top-KC list per row in LDS and filters the
//      64 accumulators per lane against the rows' current thresholds.
//   2. EXACT RE-RANK.  The KC >= topk+1+slack candidates of a row are re-scored with the
//      reference's arithmetic (sequential f64 multiply/add, same formulas as the exact scan) and
//      re-ordered by (key, index).
//   3. CERTIFICATION.  A row is accepted only if its worst accepted exact key is strictly better
//      than a rigorous lower bound on every non-candidate's exact key (approximate threshold
//      minus the MFMA rounding bound).  Rows that fail (mass ties, tiny norms) are recomputed by
//      the exact scan kernel.  The result is therefore bit-identical to the exact path.
#include "scan_topk.cuh"

namespace asx {

typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int kTileRows = 128;   // rows per packed tile / per workgroup
constexpr int kKsChunk = 8;      // k-steps per LDS chunk (32 k values)
constexpr int kChunkDoubles = kKsChunk * 8 * 64;  // 4096 doubles = 32 KB
constexpr int kMaxNC = 4;        // A fragments stay in registers up to d_pad = 128
constexpr uint32_t kQueueCap = 512;  // per-wave queue of elements that passed the threshold filter

// ---- packing ------------------------------------------------------------------------------
// P[((T*KS + ks)*8 + s)*64 + l] = unit(R)[T*128 + s*16 + (l&15)][ks*4 + (l>>4)], zero padded.
__global__ void pack_unit_rows_kernel(const double *__restrict__ R, const double *__restrict__ norms, uint64_t n,
                                      uint32_t d, uint32_t KS, uint64_t row0, uint64_t n_pad,
                                      double *__restrict__ P) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t total = n_pad * KS * 4;
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
    const double nr = norms[row0 + row];
    v = nr > 0.0 ? R[(row0 + row) * d + k] / nr : 0.0;
  }
  P[g] = v;
}

static int pack_unit_rows(as_ctx *ctx, const double *R, const double *norms, uint64_t row0, uint64_t n, uint32_t d,
                          uint32_t KS, uint64_t n_pad, double *P) {
  const uint64_t total = n_pad * KS * 4;
  ProfScope ps(ctx, "pack_unit_rows");
  hipLaunchKernelGGL(pack_unit_rows_kernel, dim3(cdiv(total, 256)), dim3(256), 0, ctx->stream, R, norms, n, d, KS,
                     row0, n_pad, P);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

// ---- epilogues: approximate key of one (row, col) product -----------------------------------
struct GraphEpi {
  uint32_t a_row0;  // global index of A row 0 (self exclusion)
  __device__ __forceinline__ void load_row(uint32_t, double &) const {}
  __device__ __forceinline__ void load_col(uint32_t, uint32_t, double &) const {}
  __device__ __forceinline__ double key(double s, double, double) const { return -s; }
  __device__ __forceinline__ bool excluded(uint32_t arow, uint32_t col) const { return a_row0 + arow == col; }
};

struct SearchEpi {
  const double *q_lambdas;
  const double *x_lambdas;
  double alpha, one_minus_alpha;
  __device__ __forceinline__ void load_row(uint32_t arow, double &v) const { v = q_lambdas[arow]; }
  __device__ __forceinline__ void load_col(uint32_t col, uint32_t nB, double &v) const {
    v = col < nB ? x_lambdas[col] : 0.0;
  }
  __device__ __forceinline__ double key(double s, double lq, double lx) const {
    const double lam = 1.0 - fmin(fabs(lq - lx), 1.0);
    return -(alpha * s + one_minus_alpha * lam);
  }
  __device__ __forceinline__ bool excluded(uint32_t, uint32_t) const { return false; }
};

// LDS traffic between lanes of ONE wave: the hardware keeps a wave's LDS operations in order, so
// only the compiler has to be told not to move memory operations across this point.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

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
        if (c == NC - 1) {
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
                    const uint32_t arow = I * kTileRows + wv * 32 + m * 16 + lq + 4 * r;
                    pass = pass && !epi.excluded(arow, col);
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
        __syncthreads();  // everyone is done reading bufB[pbuf]
        if (more) {
          double2 *dst = reinterpret_cast<double2 *>(bufB + (pbuf ^ 1) * kChunkDoubles);
#pragma unroll
          for (int q = 0; q < 8; ++q) dst[tid + 256 * q] = stage[q];
        }
        __syncthreads();
        pbuf ^= 1;
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

// ---- exact re-rank + certification ------------------------------------------------------------
// One wave per A row.  C = splits*KC candidates.  Policy = the exact scan's policy (same formulas).
struct GraphBound {
  double delta;
  int rectify;
  // lower bound on the exact key (distance) of any column that is NOT a candidate, given the
  // largest approximate key (= -smallest kept cosine) among full candidate lists
  __device__ __forceinline__ double lower_bound(double worst_approx_key) const {
    const double cmax = -worst_approx_key + delta;  // its exact cosine is at most this
    return rectify ? 1.0 - fmax(cmax, 0.0) : 1.0 - cmax;
  }
};
struct SearchBound {
  double delta;
  __device__ __forceinline__ double lower_bound(double worst_approx_key) const { return worst_approx_key - delta; }
};

template <class Policy, class Bound>
__global__ __launch_bounds__(256) void rerank_kernel(const double *__restrict__ A, uint64_t lda,
                                                     const double *__restrict__ B, uint64_t ldb, uint32_t d,
                                                     uint32_t a_begin, uint32_t a_count, uint32_t K, uint32_t KC,
                                                     uint32_t splits, const double *__restrict__ cand_key,
                                                     const uint32_t *__restrict__ cand_idx,
                                                     const uint8_t *__restrict__ force_exact, Policy pol, Bound bnd,
                                                     double *__restrict__ out_key, uint32_t *__restrict__ out_idx,
                                                     uint32_t *__restrict__ fail_rows,
                                                     uint32_t *__restrict__ fail_count, int *__restrict__ err_flag) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t C = splits * KC;
  // per wave: row copy (d) + exact keys (C) + idx (C)
  double *xrow = reinterpret_cast<double *>(smem) + (size_t)wv * (d + C);
  double *ek = xrow + d;
  uint32_t *ei = reinterpret_cast<uint32_t *>(reinterpret_cast<double *>(smem) + (size_t)4 * (d + C)) + (size_t)wv * C;

  for (uint32_t r = blockIdx.x * 4 + wv; r < a_count; r += gridDim.x * 4) {
    const uint32_t a = a_begin + r;
    for (uint32_t k = lane; k < d; k += 64) xrow[k] = A[(uint64_t)a * lda + k];
    wave_sync();
    // worst approximate key over FULL lists (a list that is not full saw every column of its split)
    double worst = __builtin_inf();
    bool bounded = false;
    for (uint32_t s = 0; s < splits; ++s) {
      const size_t o = ((size_t)r * splits + s) * KC + KC - 1;
      if (cand_idx[o] != kNoIdx) {  // full list: columns outside it have approx key >= its last key
        bounded = true;
        worst = fmin(worst, cand_key[o]);
      }
    }
    // exact keys of the candidates
    for (uint32_t c = lane; c < C; c += 64) {
      const uint32_t j = cand_idx[(size_t)r * C + c];
      double key = __builtin_inf();
      uint32_t id = kNoIdx;
      if (j != kNoIdx) {
        const double *bj = B + (uint64_t)j * ldb;
        double acc = 0.0;
        for (uint32_t k = 0; k < d; ++k) acc = __dadd_rn(acc, __dmul_rn(xrow[k], bj[k]));
        double kx;
        if (pol.key(acc, a, j, kx, err_flag)) {
          key = kx;
          id = j;
        }
      }
      ek[c] = key;
      ei[c] = id;
    }
    wave_sync();
    // rank by (key, idx); ranks < K go to the output
    double kth = -__builtin_inf();
    uint32_t n_valid = 0;
    for (uint32_t c = lane; c < C; c += 64) {
      const double kc = ek[c];
      const uint32_t ic = ei[c];
      if (ic == kNoIdx) continue;
      ++n_valid;
      uint32_t rank = 0;
      for (uint32_t c2 = 0; c2 < C; ++c2) rank += lex_less(ek[c2], ei[c2], kc, ic) ? 1u : 0u;
      if (rank < K) {
        out_key[(uint64_t)r * K + rank] = kc;
        out_idx[(uint64_t)r * K + rank] = ic;
        kth = fmax(kth, kc);
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      kth = fmax(kth, __shfl_xor(kth, o, 64));
      n_valid += __shfl_xor(n_valid, o, 64);
    }
    for (uint32_t s = n_valid + lane; s < K; s += 64) {  // fewer than K candidates exist at all
      out_key[(uint64_t)r * K + s] = __builtin_inf();
      out_idx[(uint64_t)r * K + s] = kNoIdx;
    }
    bool ok = true;
    if (bounded) ok = n_valid >= K && kth < bnd.lower_bound(worst);
    if (force_exact && force_exact[a]) ok = false;
    if (!ok && lane == 0) fail_rows[atomicAdd(fail_count, 1u)] = r;
  }
}

// Pairs with 0 < |a||b| <= 1e-12 get cos := 0 in the exact rule but a genuine cosine from the unit
// rows; a row that can form such a pair is always re-done exactly.  (Exactly-zero rows are
// consistent: their unit row is 0, so the product is 0 as well.)
__global__ void flag_tiny_norm_rows_kernel(const double *__restrict__ norms, uint32_t n, double min_pos_norm,
                                           uint8_t *__restrict__ flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const double x = norms[i];
    flag[i] = (x != x) || (x > 0.0 && !(x * min_pos_norm > 1e-12)) ? 1 : 0;
  }
}

__global__ void min_norm_kernel(const double *__restrict__ norms, uint32_t n, unsigned long long *out) {
  // norms are >= 0 (or NaN): the bit pattern of a non-negative double orders like the value
  double v = __builtin_inf();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double x = norms[i];
    if (x > 0.0 && x < v) v = x;  // smallest POSITIVE norm
  }
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMin(out, (unsigned long long)__double_as_longlong(v));
}

__global__ void scatter_rows_kernel(const uint32_t *__restrict__ rows, uint32_t count, uint32_t K,
                                    const double *__restrict__ src_key, const uint32_t *__restrict__ src_idx,
                                    double *__restrict__ dst_key, uint32_t *__restrict__ dst_idx) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count * K) return;
  const uint32_t r = t / K, s = t - r * K;
  dst_key[(uint64_t)rows[r] * K + s] = src_key[t];
  dst_idx[(uint64_t)rows[r] * K + s] = src_idx[t];
}

__global__ void offset_rows_kernel(const uint32_t *__restrict__ in, uint32_t count, uint32_t off,
                                   uint32_t *__restrict__ out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < count) out[t] = in[t] + off;
}

template <class Epi>
static int launch_candidates(as_ctx *ctx, int NC, const double *PA, uint32_t a_tiles, const double *PB,
                             uint32_t b_tiles, uint32_t nA, uint32_t nB, uint32_t KC, uint32_t splits, Epi epi,
                             double *cand_key, uint32_t *cand_idx, const char *prof_name) {
  const size_t lds = 2 * kChunkDoubles * sizeof(double) + (size_t)4 * 32 * KC * (sizeof(double) + sizeof(uint32_t)) +
                     (size_t)4 * kQueueCap * (sizeof(double) + 2 * sizeof(uint32_t));
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

static uint32_t pick_splits(uint32_t a_tiles, uint32_t b_tiles) {
  // keep >= ~4 work items per CU when the A side is short (query batches, small graphs)
  uint32_t s = 1;
  while (a_tiles * s < 1024 && s * 2 <= b_tiles && s < 16) s *= 2;
  return s;
}

static uint32_t cand_capacity(as_ctx *ctx, uint32_t K) {
  uint32_t kc = K + (uint32_t)ctx->opt_cand_slack;
  kc = (kc + 7) / 8 * 8;
  return std::min<uint32_t>(kc, 64);
}

// ---- K2 driver ----------------------------------------------------------------------------------
int knn_mfma(as_ctx *ctx, const double *rows, const double *norms, uint64_t n, uint64_t d, uint64_t row_begin,
             uint64_t row_count, uint32_t K, int rectify, uint32_t *out_idx, double *out_dist) {
  const uint32_t KS = (uint32_t)((d + 31) / 32) * 8;
  const int NC = (int)(KS / kKsChunk);
  const uint32_t KC = cand_capacity(ctx, K);
  if (NC > kMaxNC || K > 56 || n < 2 * kTileRows || row_count == 0)  // small or wide problems: exact scan
    return knn_exact(ctx, rows, norms, n, d, nullptr, row_begin, row_count, K, rectify, out_idx, out_dist, 0);

  const uint32_t b_tiles = (uint32_t)((n + kTileRows - 1) / kTileRows);
  const uint64_t n_pad = (uint64_t)b_tiles * kTileRows;
  DevBuf<double> PB, PAbuf;
  AS_TRY(PB.alloc(ctx, n_pad * KS * 4));
  AS_TRY(pack_unit_rows(ctx, rows, norms, 0, n, (uint32_t)d, KS, n_pad, PB.p));
  const double *PA = PB.p;
  uint32_t a_tiles = (uint32_t)((row_count + kTileRows - 1) / kTileRows);
  if (row_begin % kTileRows == 0) {
    PA = PB.p + (row_begin / kTileRows) * (uint64_t)KS * 512;
  } else {
    AS_TRY(PAbuf.alloc(ctx, (uint64_t)a_tiles * kTileRows * KS * 4));
    AS_TRY(pack_unit_rows(ctx, rows, norms, row_begin, row_count, (uint32_t)d, KS, (uint64_t)a_tiles * kTileRows,
                          PAbuf.p));
    PA = PAbuf.p;
  }
  const uint32_t splits = pick_splits(a_tiles, b_tiles);
  DevBuf<double> ckey;
  DevBuf<uint32_t> cidx, fail_rows, fail_count;
  AS_TRY(ckey.alloc(ctx, row_count * splits * KC));
  AS_TRY(cidx.alloc(ctx, row_count * splits * KC));
  GraphEpi epi{(uint32_t)row_begin};
  AS_TRY(launch_candidates<GraphEpi>(ctx, NC, PA, a_tiles, PB.p, b_tiles, (uint32_t)row_count, (uint32_t)n, KC,
                                     splits, epi, ckey.p, cidx.p, "knn_mfma_candidates"));

  // rows whose pairs may hit the |a||b| <= 1e-12 rule are always re-done exactly
  DevBuf<unsigned long long> minb;
  DevBuf<uint8_t> force;
  AS_TRY(minb.alloc(ctx, 1));
  AS_TRY(force.alloc(ctx, n));
  const unsigned long long inf_bits = 0x7FF0000000000000ull;
  AS_HIP(ctx, hipMemcpyAsync(minb.p, &inf_bits, sizeof(inf_bits), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(min_norm_kernel, dim3(256), dim3(256), 0, ctx->stream, norms, (uint32_t)n, minb.p);
  unsigned long long mb = 0;
  AS_HIP(ctx, hipMemcpyAsync(&mb, minb.p, sizeof(mb), hipMemcpyDeviceToHost, ctx->stream));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double min_norm;
  std::memcpy(&min_norm, &mb, sizeof(double));
  hipLaunchKernelGGL(flag_tiny_norm_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, norms, (uint32_t)n,
                     min_norm, force.p);

  AS_TRY(fail_rows.alloc(ctx, row_count));
  AS_TRY(fail_count.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(fail_count.p, 0, sizeof(uint32_t), ctx->stream));
  DevBuf<int> err;
  AS_TRY(err.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(err.p, 0, sizeof(int), ctx->stream));
  GraphDistPolicy pol{norms, rectify};
  GraphBound bnd{(double)(d + 16) * 2.3e-16, rectify};
  const uint32_t C = splits * KC;
  const size_t lds = (size_t)4 * (d + C) * sizeof(double) + (size_t)4 * C * sizeof(uint32_t);
  {
    ProfScope ps(ctx, "knn_rerank");
    AS_HIP(ctx, hipFuncSetAttribute((const void *)rerank_kernel<GraphDistPolicy, GraphBound>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((rerank_kernel<GraphDistPolicy, GraphBound>), dim3(std::min<uint64_t>(cdiv(row_count, 4), 4096)),
                       dim3(256), lds, ctx->stream, rows, d, rows, d, (uint32_t)d, (uint32_t)row_begin,
                       (uint32_t)row_count, K, KC, splits, ckey.p, cidx.p, force.p, pol, bnd, out_dist, out_idx,
                       fail_rows.p, fail_count.p, err.p);
  }
  AS_HIP(ctx, hipGetLastError());
  uint32_t n_fail = 0;
  AS_HIP(ctx, hipMemcpyAsync(&n_fail, fail_count.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stats["knn_uncertified_rows"] = n_fail;
  ctx->stats["knn_cand_capacity"] = KC;
  ctx->stats["knn_col_splits"] = splits;
  if (n_fail) {
    DevBuf<double> fk;
    DevBuf<uint32_t> fi, gids;
    AS_TRY(fk.alloc(ctx, (size_t)n_fail * K));
    AS_TRY(fi.alloc(ctx, (size_t)n_fail * K));
    AS_TRY(gids.alloc(ctx, n_fail));
    hipLaunchKernelGGL(offset_rows_kernel, dim3(cdiv(n_fail, 256)), dim3(256), 0, ctx->stream, fail_rows.p, n_fail,
                       (uint32_t)row_begin, gids.p);
    AS_TRY(knn_exact(ctx, rows, norms, n, d, gids.p, 0, n_fail, K, rectify, fi.p, fk.p, 0));
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv((uint64_t)n_fail * K, 256)), dim3(256), 0, ctx->stream,
                       fail_rows.p, n_fail, K, fk.p, fi.p, out_dist, out_idx);
    AS_HIP(ctx, hipGetLastError());
  }
  return AS_OK;
}

// ---- K9 driver ----------------------------------------------------------------------------------
int search_mfma(as_ctx *ctx, as_index *ix, const double *queries, const double *q_norms, const double *q_lambdas,
                uint64_t nq, uint32_t k, double alpha, uint32_t *out_idx, double *out_key) {
  const uint64_t n = ix->n, d = ix->d;
  const uint32_t KS = (uint32_t)((d + 31) / 32) * 8;
  const int NC = (int)(KS / kKsChunk);
  const uint32_t K = (uint32_t)std::min<uint64_t>(k, n);
  if (NC > kMaxNC || K > 56 || n < 2 * kTileRows || nq == 0)
    return search_exact(ctx, ix, queries, q_norms, q_lambdas, nullptr, nq, nq, k, alpha, out_idx, out_key);
  const uint32_t KC = cand_capacity(ctx, K);
  const uint32_t b_tiles = (uint32_t)((n + kTileRows - 1) / kTileRows);
  if (!ix->packed) {
    ix->n_pad = (uint64_t)b_tiles * kTileRows;
    ix->d_pad = (uint64_t)KS * 4;
    AS_HIP(ctx, hipMalloc((void **)&ix->packed, sizeof(double) * ix->n_pad * ix->d_pad));
    AS_TRY(pack_unit_rows(ctx, ix->items, ix->norms, 0, n, (uint32_t)d, KS, ix->n_pad, ix->packed));
  }
  const uint32_t a_tiles = (uint32_t)((nq + kTileRows - 1) / kTileRows);
  DevBuf<double> PA, ckey;
  DevBuf<uint32_t> cidx, fail_rows, fail_count;
  AS_TRY(PA.alloc(ctx, (uint64_t)a_tiles * kTileRows * KS * 4));
  AS_TRY(pack_unit_rows(ctx, queries, q_norms, 0, nq, (uint32_t)d, KS, (uint64_t)a_tiles * kTileRows, PA.p));
  const uint32_t splits = pick_splits(a_tiles, b_tiles);
  AS_TRY(ckey.alloc(ctx, nq * splits * KC));
  AS_TRY(cidx.alloc(ctx, nq * splits * KC));
  SearchEpi epi{q_lambdas, ix->lambdas, alpha, 1.0 - alpha};
  AS_TRY(launch_candidates<SearchEpi>(ctx, NC, PA.p, a_tiles, ix->packed, b_tiles, (uint32_t)nq, (uint32_t)n, KC,
                                      splits, epi, ckey.p, cidx.p, "search_mfma_candidates"));
  AS_TRY(fail_rows.alloc(ctx, nq));
  AS_TRY(fail_count.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(fail_count.p, 0, sizeof(uint32_t), ctx->stream));
  DevBuf<int> err;
  AS_TRY(err.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(err.p, 0, sizeof(int), ctx->stream));
  SearchScorePolicy pol{q_norms, q_lambdas, ix->norms, ix->lambdas, alpha};
  SearchBound bnd{(double)(d + 16) * 2.3e-16 * std::fabs(alpha) + 1e-15};
  const uint32_t C = splits * KC;
  const size_t lds = (size_t)4 * (d + C) * sizeof(double) + (size_t)4 * C * sizeof(uint32_t);
  // outputs are nq x k; the re-rank writes K = min(k, n) per row with leading dimension K == k here
  {
    ProfScope ps(ctx, "search_rerank");
    AS_HIP(ctx, hipFuncSetAttribute((const void *)rerank_kernel<SearchScorePolicy, SearchBound>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((rerank_kernel<SearchScorePolicy, SearchBound>), dim3(std::min<uint64_t>(cdiv(nq, 4), 4096)),
                       dim3(256), lds, ctx->stream, queries, d, ix->items, d, (uint32_t)d, 0u, (uint32_t)nq, K, KC,
                       splits, ckey.p, cidx.p, (const uint8_t *)nullptr, pol, bnd, out_key, out_idx, fail_rows.p,
                       fail_count.p, err.p);
  }
  AS_HIP(ctx, hipGetLastError());
  uint32_t n_fail = 0;
  int h_err = 0;
  AS_HIP(ctx, hipMemcpyAsync(&n_fail, fail_count.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  AS_HIP(ctx, hipMemcpyAsync(&h_err, err.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stats["search_uncertified_rows"] = n_fail;
  ctx->stats["search_cand_capacity"] = KC;
  ctx->stats["search_col_splits"] = splits;
  if (h_err) return fail(ctx, AS_ERR_NAN_SCORE, as_status_message(AS_ERR_NAN_SCORE));
  if (n_fail) {
    DevBuf<double> fk;
    DevBuf<uint32_t> fi;
    AS_TRY(fk.alloc(ctx, (size_t)n_fail * K));
    AS_TRY(fi.alloc(ctx, (size_t)n_fail * K));
    AS_TRY(search_exact(ctx, ix, queries, q_norms, q_lambdas, fail_rows.p, n_fail, nq, K, alpha, fi.p, fk.p));
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(cdiv((uint64_t)n_fail * K, 256)), dim3(256), 0, ctx->stream,
                       fail_rows.p, n_fail, K, fk.p, fi.p, out_key, out_idx);
    AS_HIP(ctx, hipGetLastError());
  }
  return AS_OK;
}

}  // namespace asx
