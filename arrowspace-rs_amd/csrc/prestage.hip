// prestage.hip -- the clustering pre-stage (src/clustering.rs) and the JL projection (src/reduction.rs) behind the
// C ABI: what runs immediately before the hot path inside ArrowSpaceBuilder::build.
//
// Division of labour.  Everything that touches all N rows runs on the device, results in the reference's arithmetic:
//   - Two-NN intrinsic dimension (src/clustering.rs:101-164): 500 sampled rows against all N rows, the two smallest
//     Euclidean distances of each                                    two_nn_mfma (centroids.hip) / two_nn_kernel
//   - the nearest-centroid searches of the incremental clustering (:571-594) and of Lloyd's assignment step
//                                                                     nearest_centroids (centroids.hip)
//   - the decisions and running means of the incremental clustering (:596-905)
//                                             cluster_seq_kernel, cluster_decide_kernel, cluster_apply_kernel
//   - the JL projection of rows / queries (src/reduction.rs:180-203)                            jl_project_kernel
// What works on the <= 1000-row sample (Lloyd's update, the Calinski-Harabasz score, the pilot threshold) is sequential
// bookkeeping on the host, in the same order as the reference so that the oracle's restatement matches bit for bit; so is
// the incremental clustering of rows wider than 1024 or with cluster_on_device = 0 (the older form, kept for comparison).
//
// The reference's random draws (StdRng shuffles, smartcore's k-means, the ChaCha8 Gaussian stream) cannot be reproduced
// without its Rust dependencies: the build's own seeded definitions replace them (same in oracle/prestage.c), so
// agreement with a reference run is statistical, agreement with the oracle exact.
#include <cmath>

#include "common.h"

namespace asx {

namespace {

inline uint64_t sm64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// m distinct indices of [0, n): the first m steps of a Fisher-Yates shuffle driven by SplitMix64
std::vector<uint64_t> sample_indices(uint64_t n, uint64_t m, uint64_t seed) {
  std::vector<uint64_t> idx(n), out;
  for (uint64_t i = 0; i < n; ++i) idx[i] = i;
  m = std::min(m, n);
  out.reserve(m);
  for (uint64_t t = 0; t < m; ++t) {
    const uint64_t j = t + sm64(sm64(seed) ^ sm64(t + 0x51ED270B27B4F3CFull)) % (n - t);
    std::swap(idx[t], idx[j]);
    out.push_back(idx[t]);
  }
  return out;
}

inline double dist2(const double *a, const double *b, uint64_t f) {
  double d2 = 0.0;
  for (uint64_t j = 0; j < f; ++j) {
    const double diff = a[j] - b[j];
    d2 += diff * diff;  // -ffp-contract=off: two roundings, as the reference
  }
  return d2;
}

// ---- Two-NN: per sampled row the two smallest squared distances to the other rows ---------------------------------
// One workgroup = 64 sampled rows x one chunk of the rows; the geometry of nearest_centroids_kernel (features in
// chunks of 32 through LDS, wave g owns the columns g, g + 4, ..., lane r owns sample row r, 16 partial sums per
// lane in registers, every sum over the features in order).  Partial (b1, b2) per (chunk, sample row).
constexpr int kTnRows = 64, kTnCols = 64, kTnFeat = 32, kTnPitch = kTnFeat + 1;

__global__ __launch_bounds__(256) void two_nn_kernel(const double *__restrict__ rows, uint64_t n, uint64_t d,
                                                     const uint64_t *__restrict__ sample_ids, uint64_t ns, uint64_t chunk,
                                                     double *__restrict__ part /* [chunks][ns][2] */) {
  __shared__ double s_a[kTnRows * kTnPitch];
  __shared__ double s_b[kTnCols * kTnPitch];
  __shared__ double s_m[4][kTnRows][2];
  const int tid = threadIdx.x, r = tid & 63, g = tid >> 6;
  const uint64_t s0 = (uint64_t)blockIdx.x * kTnRows;  // first sample row of this workgroup
  const uint64_t c_begin = (uint64_t)blockIdx.y * chunk, c_end = c_begin + chunk < n ? c_begin + chunk : n;
  const uint64_t my = s0 + r < ns ? sample_ids[s0 + r] : ~0ull;
  double b1 = __builtin_inf(), b2 = __builtin_inf();
  for (uint64_t c0 = c_begin; c0 < c_end; c0 += kTnCols) {
    double acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.0;
    for (uint64_t k0 = 0; k0 < d; k0 += kTnFeat) {
      const int feats = (int)((d - k0) < (uint64_t)kTnFeat ? (d - k0) : (uint64_t)kTnFeat);
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int e = q * 256 + tid, rr = e >> 5, kk = e & 31;
        const uint64_t sr = s0 + rr < ns ? sample_ids[s0 + rr] : ~0ull;
        s_a[rr * kTnPitch + kk] = (sr != ~0ull && kk < feats) ? rows[sr * d + k0 + kk] : 0.0;
        s_b[rr * kTnPitch + kk] = (c0 + rr < c_end && kk < feats) ? rows[(c0 + rr) * d + k0 + kk] : 0.0;
      }
      __syncthreads();
      for (int kk = 0; kk < feats; ++kk) {
        const double a = s_a[r * kTnPitch + kk];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const double diff = __dsub_rn(a, s_b[(g + 4 * j) * kTnPitch + kk]);
          acc[j] = __dadd_rn(acc[j], __dmul_rn(diff, diff));
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint64_t c = c0 + g + 4 * j;
      if (c < c_end && c != my) {
        const double v = acc[j];
        if (v < b1) {
          b2 = b1;
          b1 = v;
        } else if (v < b2) {
          b2 = v;
        }
      }
    }
  }
  s_m[g][r][0] = b1;
  s_m[g][r][1] = b2;
  __syncthreads();
  if (g == 0 && s0 + r < ns) {
    double m1 = __builtin_inf(), m2 = __builtin_inf();
    for (int w = 0; w < 4; ++w)
      for (int e = 0; e < 2; ++e) {
        const double v = s_m[w][r][e];
        if (v < m1) {
          m2 = m1;
          m1 = v;
        } else if (v < m2) {
          m2 = v;
        }
      }
    part[((uint64_t)blockIdx.y * ns + s0 + r) * 2] = m1;
    part[((uint64_t)blockIdx.y * ns + s0 + r) * 2 + 1] = m2;
  }
}

// ---- JL projection ---------------------------------------------------------------------------------------------------
// G[i][r] ~ N(0, 1), i < original_dim, r < reduced_dim: counter-based SplitMix64 + Box-Muller (the build's own seed
// format; the reference draws the same shape from ChaCha8, src/reduction.rs:186-199)
__global__ void jl_gaussian_kernel(uint64_t seed, uint64_t total, double *__restrict__ G) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  auto mix = [](uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
  };
  const uint64_t key = mix(seed * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull);
  const double u1 = ((double)(mix((2 * e) ^ key) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
  const double u2 = ((double)(mix((2 * e + 1) ^ key) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
  G[e] = sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}

// out[row][t] = sum_i (x[row][i] * G[i][t]) * scale in the reference's order and rounding: result += original * sample * scale,
// original-major (src/reduction.rs:191-199).  One thread per (row, t); the row's values are LDS broadcasts.
__global__ __launch_bounds__(256) void jl_project_kernel(const double *__restrict__ X, uint64_t n, uint64_t d,
                                                         const double *__restrict__ G, uint64_t r, double scale,
                                                         double *__restrict__ out) {
  extern __shared__ double s_x[];
  const uint64_t row = blockIdx.x;
  for (uint64_t i = threadIdx.x; i < d; i += blockDim.x) s_x[i] = X[row * d + i];
  __syncthreads();
  for (uint64_t t = threadIdx.x; t < r; t += blockDim.x) {
    double acc = 0.0;
    for (uint64_t i = 0; i < d; ++i) acc = __dadd_rn(acc, __dmul_rn(__dmul_rn(s_x[i], G[i * r + t]), scale));
    out[row * r + t] = acc;
  }
}

// ---- incremental clustering once every centroid exists (src/clustering.rs:721-905 with the cluster cap reached) --------------
// From then on a row's fate depends on its nearest centroid of the batch's snapshot alone: within `radius` it joins and moves
// the running mean, within 1.5 radius it is counted, else it is dropped -- no new centroid can appear, so nothing of the batch
// has to visit the host.  decide: one thread per row.  apply: one workgroup per centroid walks the batch IN ROW ORDER (the
// reference's order of the running-mean updates of one centroid; different centroids do not interact) with a thread per
// feature: c += (x - c) / k_new, the same two roundings as the host loop.
__global__ void cluster_decide_kernel(const uint64_t *__restrict__ idx, const double *__restrict__ d2, uint64_t bn, double radius,
                                      uint32_t *__restrict__ act /* 0 dropped, 1 counted, 2 joins */, long long *__restrict__ assign) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= bn) return;
  const double bd = d2[r];
  const uint32_t a = bd <= radius ? 2u : (bd <= radius * 1.5 ? 1u : 0u);
  act[r] = a;
  assign[r] = a ? (long long)idx[r] : -1ll;
}
__global__ __launch_bounds__(256) void cluster_apply_kernel(const double *__restrict__ rows, uint64_t bn, uint64_t f,
                                                            const uint64_t *__restrict__ idx, const uint32_t *__restrict__ act,
                                                            double *__restrict__ cents, unsigned long long *__restrict__ counts) {
  const uint64_t c = blockIdx.x;
  __shared__ uint32_t s_rows[256];
  __shared__ uint32_t s_n;
  unsigned long long cnt = counts[c];
  for (uint64_t base = 0; base < bn; base += 256) {
    // the rows of this chunk that belong to centroid c, in row order (a stable compaction by a block-wide prefix count)
    __syncthreads();
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint64_t r = base + threadIdx.x;
    const uint32_t a = (r < bn && idx[r] == c) ? act[r] : 0u;
    const unsigned long long m = __ballot(a != 0);
    __shared__ uint32_t s_wave[4];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) s_wave[wv] = (uint32_t)__builtin_popcountll(m);
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wv; ++w) before += s_wave[w];
    if (a) s_rows[before + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = (uint32_t)threadIdx.x | (a << 16);
    if (threadIdx.x == 0) s_n = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
    const uint32_t nn = s_n;
    for (uint32_t e = 0; e < nn; ++e) {
      const uint32_t w = s_rows[e], rr = w & 0xFFFFu, aa = w >> 16;
      if (aa == 2u) {
        const double k_new = (double)cnt + 1.0;
        const double *row = rows + (base + rr) * f;
        for (uint64_t j = threadIdx.x; j < f; j += 256) {
          const double cj = cents[c * f + j];
          cents[c * f + j] = __dadd_rn(cj, __ddiv_rn(__dsub_rn(row[j], cj), k_new));
        }
      }
      cnt += 1;
    }
  }
  if (threadIdx.x == 0) counts[c] = cnt;
}

// ---- a batch in which centroids can still appear: sequential in its rows, parallel in the centroids ------------------------
// (src/clustering.rs:596-905 for one batch.)  A row has to be measured against the centroids created earlier in its own batch
// at their CURRENT values, and its own decision (new centroid / running-mean update) changes what the next row sees, so the rows
// of such a batch form a chain.  One workgroup walks it: thread t owns the live centroids live_from + t, + 512, ... and runs
// the reference's sum over the features for each (d2 += (a - b) * (a - b), two roundings, features ascending) from a
// TRANSPOSED copy of the live centroids (lanes read neighbouring words); the minimum with the smallest index among equal
// minima meets the snapshot's (idx, d2) of the row (strict '<': the snapshot wins ties, as in the host loop that starts from
// it), every thread takes the same decision, and the threads of the first f lanes apply it to both copies.  live_from = the
// snapshot's centroid count (batched mode) or 0 with no snapshot at all (batch = 1: the reference's sequential mode, where a
// row sees every centroid as it stands).
// What a row costs is latency, so: the centroid words come from L2 through a ring of four chunks of 16 (64 loads in flight
// per lane before the first is needed); the NEXT row and its snapshot entry are fetched by the workgroup's upper half only
// (memory operations retire in order per wave: a row fetch from HBM issued by a wave that also measures centroids would sit in
// front of every wait of its chain), at the start of the row, and reach LDS before the row's last barrier.
constexpr int kSeqThreads = 512, kSeqWaves = kSeqThreads / 64, kSeqChunk = 16, kSeqRing = 4, kSeqLoaders = 256;
constexpr uint64_t kSeqMaxF = 1024;
__global__ __launch_bounds__(kSeqThreads) void cluster_seq_kernel(const double *__restrict__ rows, uint64_t bn, uint32_t f,
                                                                  const uint64_t *__restrict__ snap_idx,
                                                                  const double *__restrict__ snap_d2, uint64_t live_from, double radius,
                                                                  uint64_t max_clusters, double *cents, double *centsT,
                                                                  unsigned long long *counts, long long *__restrict__ assign,
                                                                  unsigned long long *nc_io, int stop_at_cap, unsigned long long *stamps) {
  extern __shared__ double s_row[];  // two rows of f
  __shared__ double s_wd[kSeqWaves];
  __shared__ uint64_t s_wc[kSeqWaves];
  __shared__ double s_sd[2];    // the snapshot's entry of the row, by row parity
  __shared__ uint64_t s_si[2];
  const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
  const bool loader = t >= (uint32_t)(kSeqThreads - kSeqLoaders);
  const uint32_t lt = t - (uint32_t)(kSeqThreads - kSeqLoaders);
  const uint64_t M = max_clusters;
  // nc_io[0]: cluster count (in / out); nc_io[1]: rows of the batch consumed (out).  stop_at_cap: the batch ends with the row
  // that creates the last centroid the cap allows (the rule of oracle/prestage.c: from there on no centroid can appear and the
  // rows go through the parallel kernels with a snapshot of all of them)
  uint64_t nc = nc_io[0];
  const bool may_stop = stop_at_cap && nc < M;
  uint64_t rows_done = bn;
  for (uint32_t j = t; j < f; j += kSeqThreads) s_row[j] = rows[j];
  if (t == 0) {
    s_sd[0] = snap_d2 ? snap_d2[0] : __builtin_inf();
    s_si[0] = snap_idx ? snap_idx[0] : 0;
  }
  __syncthreads();
  const uint32_t nch = f / kSeqChunk;
  for (uint64_t r = 0; r < bn; ++r) {
    const double *cur = s_row + (r & 1) * f;
    double *nxt = s_row + ((r + 1) & 1) * f;
    double pre[kSeqMaxF / kSeqLoaders];
    double p_bd = __builtin_inf();
    uint64_t p_bi = 0;
    const unsigned long long t0 = stamps ? __builtin_amdgcn_s_memtime() : 0;
    if (loader && r + 1 < bn) {
#pragma unroll
      for (uint32_t q = 0; q < kSeqMaxF / kSeqLoaders; ++q) {
        const uint32_t j = lt + q * kSeqLoaders;
        pre[q] = j < f ? rows[(r + 1) * f + j] : 0.0;
      }
      if (lt == 0) {
        if (snap_d2) p_bd = snap_d2[r + 1];
        if (snap_idx) p_bi = snap_idx[r + 1];
      }
    }
    double best = __builtin_inf();
    uint64_t bc = ~0ull;
    for (uint64_t c = live_from + t; c < nc; c += kSeqThreads) {
      // (row j of the transposed copy is a uniform base, the centroid a 32-bit lane offset)
      const uint32_t co = (uint32_t)c;
#define AS_CT(J) (centsT + (uint64_t)(J) * M)[co]
      double acc = 0.0;
      double v[kSeqRing][kSeqChunk];
      // whole turns of the ring, branch-free (a conditional refill would make the compiler drain the ring at every join)
      const uint32_t turns = nch / kSeqRing;
      uint32_t j = 0;
      if (turns) {
#pragma unroll
        for (int b = 0; b < kSeqRing; ++b) {
#pragma unroll
          for (int u = 0; u < kSeqChunk; ++u) v[b][u] = AS_CT(b * kSeqChunk + u);
        }
        // the row's words of a chunk are read from LDS one chunk ahead of their use (an LDS read takes longer than the three
        // operations it feeds)
        typedef double d2 __attribute__((ext_vector_type(2)));
        const d2 *cur2 = reinterpret_cast<const d2 *>(cur);
        d2 xa[kSeqChunk / 2], xb[kSeqChunk / 2];
#pragma unroll
        for (int u = 0; u < kSeqChunk / 2; ++u) xa[u] = cur2[u];
#define AS_CHUNK(XA, XB, B, NEXT)                                                      \
  {                                                                                    \
    _Pragma("unroll") for (int u = 0; u < kSeqChunk / 2; ++u) XB[u] = cur2[(NEXT) / 2 + u]; \
    _Pragma("unroll") for (int u = 0; u < kSeqChunk / 2; ++u) {                        \
      const double d0 = __dsub_rn(XA[u].x, v[B][2 * u]);                               \
      acc = __dadd_rn(acc, __dmul_rn(d0, d0));                                         \
      const double d1 = __dsub_rn(XA[u].y, v[B][2 * u + 1]);                           \
      acc = __dadd_rn(acc, __dmul_rn(d1, d1));                                         \
    }                                                                                  \
  }
        static_assert(kSeqRing == 4, "the chunks of a turn below");
        const uint32_t last = turns * kSeqRing * kSeqChunk - kSeqChunk;  // the read-ahead of the last chunk stays inside the row
        for (uint32_t tn = 0; tn + 1 < turns; ++tn) {
          AS_CHUNK(xa, xb, 0, j + 1 * kSeqChunk)
#pragma unroll
          for (int u = 0; u < kSeqChunk; ++u) v[0][u] = AS_CT(j + (kSeqRing + 0) * kSeqChunk + u);
          AS_CHUNK(xb, xa, 1, j + 2 * kSeqChunk)
#pragma unroll
          for (int u = 0; u < kSeqChunk; ++u) v[1][u] = AS_CT(j + (kSeqRing + 1) * kSeqChunk + u);
          AS_CHUNK(xa, xb, 2, j + 3 * kSeqChunk)
#pragma unroll
          for (int u = 0; u < kSeqChunk; ++u) v[2][u] = AS_CT(j + (kSeqRing + 2) * kSeqChunk + u);
          AS_CHUNK(xb, xa, 3, j + 4 * kSeqChunk)
#pragma unroll
          for (int u = 0; u < kSeqChunk; ++u) v[3][u] = AS_CT(j + (kSeqRing + 3) * kSeqChunk + u);
          j += kSeqRing * kSeqChunk;
        }
        AS_CHUNK(xa, xb, 0, j + 1 * kSeqChunk)
        AS_CHUNK(xb, xa, 1, j + 2 * kSeqChunk)
        AS_CHUNK(xa, xb, 2, j + 3 * kSeqChunk)
        AS_CHUNK(xb, xa, 3, (j + 4 * kSeqChunk <= last ? j + 4 * kSeqChunk : last))
#undef AS_CHUNK
        j += kSeqRing * kSeqChunk;
      }
      for (; j < f; ++j) {  // widths that are not a multiple of 64: the rest one word at a time
        const double diff = __dsub_rn(cur[j], AS_CT(j));
        acc = __dadd_rn(acc, __dmul_rn(diff, diff));
      }
#undef AS_CT
      if (acc < best) {  // this thread's centroids come in index order: strict '<' keeps the first of equal minima
        best = acc;
        bc = c;
      }
    }
    const unsigned long long t1 = stamps ? __builtin_amdgcn_s_memtime() : 0;
    for (int o = 32; o > 0; o >>= 1) {
      const double od = __shfl_xor(best, o, 64);
      const uint64_t oc = __shfl_xor((unsigned long long)bc, o, 64);
      if (od < best || (od == best && oc < bc)) {
        best = od;
        bc = oc;
      }
    }
    if (lane == 0) {
      s_wd[wv] = best;
      s_wc[wv] = bc;
    }
    __syncthreads();
    double bd = s_sd[r & 1];
    uint64_t bi = s_si[r & 1];
    {
      double ld = __builtin_inf();
      uint64_t lc = ~0ull;
#pragma unroll
      for (int w = 0; w < kSeqWaves; ++w) {
        const double od = s_wd[w];
        const uint64_t oc = s_wc[w];
        if (od < ld || (od == ld && oc < lc)) {
          ld = od;
          lc = oc;
        }
      }
      if (lc != ~0ull && ld < bd) {
        bd = ld;
        bi = lc;
      }
    }
    const unsigned long long t2 = stamps ? __builtin_amdgcn_s_memtime() : 0;
    // the reference's decisions (src/clustering.rs:637-656, 670-713, 721-905), the same for every thread
    if (nc == 0 || (nc < M && bd > radius * 0.5)) {
      for (uint32_t j = t; j < f; j += kSeqThreads) {
        const double vj = cur[j];
        cents[nc * f + j] = vj;
        centsT[(uint64_t)j * M + nc] = vj;
      }
      if (t == 0) {
        counts[nc] = 1;
        assign[r] = (long long)nc;
      }
      ++nc;
    } else if (bd <= radius) {
      const double k_new = (double)counts[bi] + 1.0;
      for (uint32_t j = t; j < f; j += kSeqThreads) {
        const double cj = cents[bi * f + j];
        const double vj = __dadd_rn(cj, __ddiv_rn(__dsub_rn(cur[j], cj), k_new));
        cents[bi * f + j] = vj;
        centsT[(uint64_t)j * M + bi] = vj;
      }
      __syncthreads();  // every thread has read the count
      if (t == 0) {
        counts[bi] += 1;
        assign[r] = (long long)bi;
      }
    } else if (t == 0) {
      if (bd <= radius * 1.5) {
        counts[bi] += 1;
        assign[r] = (long long)bi;
      } else {
        assign[r] = -1;
      }
    }
    const unsigned long long t3 = stamps ? __builtin_amdgcn_s_memtime() : 0;
    if (loader && r + 1 < bn) {
#pragma unroll
      for (uint32_t q = 0; q < kSeqMaxF / kSeqLoaders; ++q) {
        const uint32_t j = lt + q * kSeqLoaders;
        if (j < f) nxt[j] = pre[q];
      }
      if (lt == 0) {
        s_sd[(r + 1) & 1] = p_bd;
        s_si[(r + 1) & 1] = p_bi;
      }
    }
    __syncthreads();  // the centroids as the next row has to see them; the row buffers; s_wd / s_wc / the snapshot slots
    if (may_stop && nc == M) {
      rows_done = r + 1;
      break;
    }
    if (stamps && lane == 0) {  // diagnostic build of a launch: cycles per phase, per wave
      const unsigned long long t4 = __builtin_amdgcn_s_memtime();
      stamps[wv * 4 + 0] += t1 - t0;
      stamps[wv * 4 + 1] += t2 - t1;
      stamps[wv * 4 + 2] += t3 - t2;
      stamps[wv * 4 + 3] += t4 - t3;
    }
  }
  if (t == 0) {
    nc_io[0] = nc;
    nc_io[1] = rows_done;
  }
}

// rows [begin, begin + count) of a host- or device-resident matrix as a host pointer
struct HostRows {
  std::vector<double> buf;
  const double *p = nullptr;
  int get(as_ctx *ctx, const double *rows, bool on_device, uint64_t begin, uint64_t count, uint64_t f) {
    if (!on_device) {
      p = rows + begin * f;
      return AS_OK;
    }
    buf.resize((size_t)count * f);
    AS_HIP(ctx, copy_d2h(ctx, buf.data(), rows + begin * f, sizeof(double) * count * f));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    p = buf.data();
    return AS_OK;
  }
};

__global__ void gather_rows_kernel(const double *__restrict__ rows, const uint64_t *__restrict__ ids, uint64_t count, uint64_t f,
                                   double *__restrict__ out) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < count * f) out[t] = rows[ids[t / f] * f + t % f];
}
// the listed rows of a host- or device-resident matrix as one host block (device rows: one gather kernel and ONE copy back;
// it used to be a copy per row: 1 000 round trips for the sample of compute_optimal_k)
int gather_rows_host(as_ctx *ctx, const double *rows, bool on_device, const std::vector<uint64_t> &ids, uint64_t f,
                     std::vector<double> &out) {
  out.resize(ids.size() * f);
  if (ids.empty()) return AS_OK;
  if (!on_device) {
    for (size_t s = 0; s < ids.size(); ++s) std::memcpy(out.data() + s * f, rows + ids[s] * f, sizeof(double) * f);
    return AS_OK;
  }
  DevBuf<uint64_t> idd;
  DevBuf<double> g;
  AS_TRY(idd.alloc(ctx, ids.size()));
  AS_TRY(g.alloc(ctx, ids.size() * f));
  AS_HIP(ctx, copy_h2d(ctx, idd.p, ids.data(), sizeof(uint64_t) * ids.size()));
  hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(ids.size() * f, 256)), dim3(256), 0, ctx->stream, rows, (const uint64_t *)idd.p,
                     (uint64_t)ids.size(), f, g.p);
  AS_HIP(ctx, hipGetLastError());
  AS_HIP(ctx, copy_d2h(ctx, out.data(), g.p, sizeof(double) * ids.size() * f));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return AS_OK;
}

}  // namespace

// ---- stage functions (also the C ABI's workers) -------------------------------------------------------------------------

int two_nn_ratios(as_ctx *ctx, const double *rows_dev, uint64_t n, uint64_t f, const uint64_t *sample_ids_host, uint64_t ns,
                  double *ratios_host) {
  if (ns == 0) return AS_OK;
  const uint64_t chunks = std::max<uint64_t>(1, std::min<uint64_t>(256, cdiv(n, 4096)));
  const uint64_t chunk = (cdiv(n, chunks) + kTnCols - 1) / kTnCols * kTnCols;
  const uint64_t n_chunks = cdiv(n, chunk);
  DevBuf<uint64_t> ids;
  DevBuf<double> part;
  AS_TRY(ids.alloc(ctx, ns));
  AS_TRY(part.alloc(ctx, n_chunks * ns * 2));
  AS_HIP(ctx, copy_h2d(ctx, ids.p, sample_ids_host, sizeof(uint64_t) * ns));
  ctx->stats["two_nn_mfma"] = 0;
  // keys on the f32 matrix pipe, exact distances of the points inside the error window (centroids.hip); the samples whose
  // window holds too many points -- and every sample when the shape or the data rule the pass out -- take the exact kernel
  std::vector<double> hb;
  std::vector<uint32_t> redo;
  bool have_mfma = false;
  {
    DevBuf<double> b12;
    AS_TRY(b12.alloc(ctx, ns * 2));
    const int rc = two_nn_mfma(ctx, rows_dev, n, f, ids.p, ns, b12.p, &redo);
    if (rc == AS_OK) {
      hb.resize(ns * 2);
      AS_HIP(ctx, copy_d2h(ctx, hb.data(), b12.p, sizeof(double) * hb.size()));
      AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
      have_mfma = true;
    } else if (rc != AS_ERR_UNSUPPORTED) {
      return rc;
    }
  }
  auto ratio = [&](double b1, double b2) {
    const double d1 = std::sqrt(b1), d2 = std::sqrt(b2);  // src/clustering.rs:135,141-147
    return (n >= 3 && d1 > 1e-12) ? d2 / d1 : (double)NAN;
  };
  std::vector<uint64_t> ex_ids;       // samples of the exact pass and where their ratios go
  std::vector<uint64_t> ex_slot;
  if (have_mfma) {
    for (uint64_t s = 0; s < ns; ++s) ratios_host[s] = ratio(hb[2 * s], hb[2 * s + 1]);
    for (uint32_t s : redo) {
      ex_ids.push_back(sample_ids_host[s]);
      ex_slot.push_back(s);
    }
    if (ex_ids.empty()) return AS_OK;
    AS_HIP(ctx, copy_h2d(ctx, ids.p, ex_ids.data(), sizeof(uint64_t) * ex_ids.size()));
  } else {
    for (uint64_t s = 0; s < ns; ++s) ex_slot.push_back(s);
  }
  const uint64_t ns_all = ns;
  ns = ex_slot.size();
  (void)ns_all;
  {
    ProfScope ps(ctx, "two_nn");
    hipLaunchKernelGGL(two_nn_kernel, dim3(cdiv(ns, kTnRows), (unsigned)n_chunks), dim3(256), 0, ctx->stream, rows_dev, n, f, ids.p, ns,
                       chunk, part.p);
    AS_HIP(ctx, hipGetLastError());
  }
  std::vector<double> h(n_chunks * ns * 2);
  AS_HIP(ctx, copy_d2h(ctx, h.data(), part.p, sizeof(double) * h.size()));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (uint64_t s = 0; s < ns; ++s) {
    double b1 = INFINITY, b2 = INFINITY;
    for (uint64_t c = 0; c < n_chunks; ++c)
      for (int e = 0; e < 2; ++e) {
        const double v = h[(c * ns + s) * 2 + e];
        if (v < b1) {
          b2 = b1;
          b1 = v;
        } else if (v < b2) {
          b2 = v;
        }
      }
    ratios_host[ex_slot[s]] = ratio(b1, b2);
  }
  return AS_OK;
}

static uint64_t intrinsic_dimension(uint64_t n, uint64_t f, const double *ratios, uint64_t ns) {  // src/clustering.rs:101-164
  if (n < 10) return std::min<uint64_t>(f, 2);
  double sum = 0.0;
  uint64_t cnt = 0;
  for (uint64_t s = 0; s < ns; ++s)
    if (ratios[s] == ratios[s]) {
      sum += ratios[s];
      ++cnt;
    }
  if (cnt == 0) return std::min<uint64_t>(f, 3);
  const double mean = sum / (double)cnt;
  const double id = mean > 1.001 ? 1.0 / std::log(mean) : (double)f;
  double r = std::round(id);
  r = std::max(1.0, std::min((double)f, r));
  return (uint64_t)r;
}

// Lloyd's k-means on rows that live on BOTH sides (the <= 1000-row sample): assignment on the device (nearest_centroids: the
// reference's nearest_centroid arithmetic and tie rule), update on the host in row order.
int kmeans_lloyd(as_ctx *ctx, const double *rows_dev, const double *rows_host, uint64_t n, uint64_t f, uint64_t k, uint64_t max_iter,
                 uint64_t seed, std::vector<uint64_t> &assign, std::vector<double> &cents) {
  assign.assign(n, 0);
  if (n == 0) return AS_OK;
  k = std::min(k, n);
  cents.resize(k * f);
  const std::vector<uint64_t> init = sample_indices(n, k, seed);
  for (uint64_t c = 0; c < k; ++c) std::memcpy(cents.data() + c * f, rows_host + init[c] * f, sizeof(double) * f);
  DevBuf<double> cd, d2d;
  DevBuf<uint64_t> idxd;
  AS_TRY(cd.alloc(ctx, k * f));
  AS_TRY(d2d.alloc(ctx, n));
  AS_TRY(idxd.alloc(ctx, n));
  std::vector<uint64_t> prev(n), count(k);
  std::vector<double> sum(k * f);
  for (uint64_t it = 0;; ++it) {
    AS_HIP(ctx, copy_h2d(ctx, cd.p, cents.data(), sizeof(double) * k * f));
    AS_TRY(nearest_centroids(ctx, rows_dev, n, f, cd.p, k, idxd.p, d2d.p));
    AS_HIP(ctx, copy_d2h(ctx, assign.data(), idxd.p, sizeof(uint64_t) * n));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    bool same = it > 0;
    if (it > 0)
      for (uint64_t i = 0; i < n && same; ++i) same = prev[i] == assign[i];
    if (same || it == max_iter) break;
    prev = assign;
    std::fill(sum.begin(), sum.end(), 0.0);
    std::fill(count.begin(), count.end(), 0);
    for (uint64_t i = 0; i < n; ++i) {
      const uint64_t a = assign[i];
      for (uint64_t j = 0; j < f; ++j) sum[a * f + j] += rows_host[i * f + j];
      ++count[a];
    }
    for (uint64_t c = 0; c < k; ++c)
      if (count[c] > 0)
        for (uint64_t j = 0; j < f; ++j) cents[c * f + j] = sum[c * f + j] / (double)count[c];
  }
  return AS_OK;
}

static void centroids_of(const double *rows, uint64_t n, uint64_t f, const std::vector<uint64_t> &assign, uint64_t k,
                         std::vector<double> &cents, std::vector<uint64_t> &count) {
  cents.assign(k * f, 0.0);
  count.assign(k, 0);
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t a = assign[i];
    if (a >= k) continue;
    for (uint64_t j = 0; j < f; ++j) cents[a * f + j] += rows[i * f + j];
    ++count[a];
  }
  for (uint64_t c = 0; c < k; ++c)
    if (count[c] > 0)
      for (uint64_t j = 0; j < f; ++j) cents[c * f + j] /= (double)count[c];
}

double calinski_harabasz(const double *rows, uint64_t n, uint64_t f, const std::vector<uint64_t> &assign, uint64_t k) {  // :314-381
  if (k <= 1 || k >= n) return 0.0;
  std::vector<double> g(f), cents;
  std::vector<uint64_t> count;
  for (uint64_t j = 0; j < f; ++j) {
    double s = 0.0;
    for (uint64_t i = 0; i < n; ++i) s += rows[i * f + j];
    g[j] = s / (double)n;
  }
  centroids_of(rows, n, f, assign, k, cents, count);
  double bgss = 0.0;
  for (uint64_t c = 0; c < k; ++c)
    if (count[c] > 0) bgss += (double)count[c] * dist2(cents.data() + c * f, g.data(), f);
  double wgss = 0.0;
  for (uint64_t i = 0; i < n; ++i)
    if (assign[i] < k) wgss += dist2(rows + i * f, cents.data() + assign[i] * f, f);
  if (wgss < 1e-10) return 0.0;
  return (bgss / (double)(k - 1)) / (wgss / (double)(n - k));
}

static int best_ch(as_ctx *ctx, const double *rd, const double *rh, uint64_t n, uint64_t f, uint64_t k, uint64_t seed0, double *out) {
  double best = 0.0;
  bool have = false;
  std::vector<uint64_t> assign;
  std::vector<double> cents;
  for (uint64_t trial = 0; trial < 3; ++trial) {
    AS_TRY(kmeans_lloyd(ctx, rd, rh, n, f, k, 20, seed0 + trial, assign, cents));
    const double s = calinski_harabasz(rh, n, f, assign, k);
    if (!have || s >= best) {
      best = s;
      have = true;
    }
  }
  *out = best;
  return AS_OK;
}

static int step2_calinski_harabasz(as_ctx *ctx, const double *rd, const double *rh, uint64_t n, uint64_t f, uint64_t k_min,
                                   uint64_t k_max, uint64_t seed, uint64_t *k_out) {  // :167-310
  if (n < 10) {
    *k_out = k_min;
    return AS_OK;
  }
  const uint64_t range = k_max - k_min, step = range <= 5 ? 1 : (range <= 15 ? 2 : 3);
  uint64_t best_k = k_min;
  double best_score = -INFINITY;
  bool have = false;
  for (uint64_t k = k_min; k <= k_max; k += step) {
    if (!(k < n && k >= 2)) continue;
    double ch;
    AS_TRY(best_ch(ctx, rd, rh, n, f, k, seed + k * 1000, &ch));
    const double sc = ch - 0.8 * (double)k * std::log((double)n);
    if (!have || !(sc < best_score)) {  // max_by: the greater score; equal or incomparable: the larger k
      best_k = k;
      best_score = sc;
      have = true;
    }
  }
  if (step > 1) {
    const uint64_t fine[5] = {best_k >= step - 1 ? best_k - (step - 1) : 0, best_k >= 1 ? best_k - 1 : 0, best_k,
                              std::min(best_k + 1, k_max), std::min(best_k + step - 1, k_max)};
    uint64_t fk = 0;
    double fs = 0.0;
    bool fhave = false;
    for (int t = 0; t < 5; ++t) {
      const uint64_t k = fine[t];
      if (!(k >= k_min && k <= k_max && k < n)) continue;
      if ((k - k_min) % step == 0) continue;  // already a candidate
      double ch;
      AS_TRY(best_ch(ctx, rd, rh, n, f, k, seed + k * 10000, &ch));
      const double sc = ch - 0.8 * (double)k * std::log((double)n);
      if (!fhave || !(sc < fs)) {
        fk = k;
        fs = sc;
        fhave = true;
      }
    }
    if (fhave && fs > best_score) {
      best_k = fk;
      best_score = fs;
    }
  }
  *k_out = best_k < k_max ? best_k : k_max;
  return AS_OK;
}

static int threshold_from_pilot(as_ctx *ctx, const double *rd, const double *rh, uint64_t n, uint64_t f, uint64_t k, uint64_t seed,
                                double *radius_out) {  // :384-492
  k = std::min(k, n);
  std::vector<uint64_t> assign, count;
  std::vector<double> cents, dists;
  AS_TRY(kmeans_lloyd(ctx, rd, rh, n, f, k, 20, seed + 100000, assign, cents));
  centroids_of(rh, n, f, assign, k, cents, count);
  for (uint64_t i = 0; i < n; ++i)
    if (assign[i] < k) dists.push_back(dist2(rh + i * f, cents.data() + assign[i] * f, f));
  if (dists.empty()) {
    *radius_out = 1.0;
    return AS_OK;
  }
  std::sort(dists.begin(), dists.end());
  const uint64_t idx = std::min<uint64_t>((uint64_t)std::ceil((double)dists.size() * 0.9), dists.size() - 1);
  const double p90 = dists[idx];
  double min_inter = INFINITY;
  bool any_inter = false;
  for (uint64_t a = 0; a < k; ++a)
    for (uint64_t b = a + 1; b < k; ++b)
      if (count[a] > 0 && count[b] > 0) {
        const double d = dist2(cents.data() + a * f, cents.data() + b * f, f);
        any_inter = true;
        if (d < min_inter) min_inter = d;
      }
  const double ratio = (min_inter < INFINITY && min_inter > 0.0) ? p90 / min_inter : 1.0;
  if (p90 < 1e-8 || ratio < 0.01) *radius_out = any_inter ? std::max(min_inter * 0.15, 1e-6) : 1e-6;
  else *radius_out = std::max(p90 * 1.5, 1e-6);
  return AS_OK;
}

}  // namespace asx

using namespace asx;

struct as_jl {
  int device = 0;
  uint64_t original_dim = 0, reduced_dim = 0, seed = 0;
  double *G = nullptr;  // original_dim x reduced_dim (device)
};

static int pre_check(as_ctx *ctx) {
  if (!ctx) return AS_ERR_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, AS_ERR_HIP, "hipSetDevice failed");
  ctx->err.clear();
  return AS_OK;
}

extern "C" {

int as_two_nn_ratios(as_ctx *ctx, const double *rows, uint64_t n_rows, uint64_t n_features, const uint64_t *sample_ids,
                     uint64_t n_samples, double *ratios_out) {
  AS_TRY(pre_check(ctx));
  if (!rows || (n_samples && (!sample_ids || !ratios_out))) return fail(ctx, AS_ERR_ARG, "null argument");
  for (uint64_t s = 0; s < n_samples; ++s)
    if (sample_ids[s] >= n_rows) return fail(ctx, AS_ERR_ARG, "sample id out of range");
  DevIn<double> R;
  AS_TRY(R.bind(ctx, rows, n_rows * n_features));
  return two_nn_ratios(ctx, R.p, n_rows, n_features, sample_ids, n_samples, ratios_out);
}

int as_kmeans_lloyd(as_ctx *ctx, const double *rows, uint64_t n_rows, uint64_t n_features, uint64_t k, uint64_t max_iter,
                    uint64_t seed, uint64_t *assignments_out, double *centroids_out) {
  AS_TRY(pre_check(ctx));
  if (!rows || !assignments_out || k == 0) return fail(ctx, AS_ERR_ARG, "null argument");
  DevIn<double> R;
  AS_TRY(R.bind(ctx, rows, n_rows * n_features));
  HostRows H;
  AS_TRY(H.get(ctx, rows, is_device_ptr(rows), 0, n_rows, n_features));
  std::vector<uint64_t> assign;
  std::vector<double> cents;
  AS_TRY(kmeans_lloyd(ctx, R.p, H.p, n_rows, n_features, k, max_iter, seed, assign, cents));
  std::memcpy(assignments_out, assign.data(), sizeof(uint64_t) * n_rows);
  if (centroids_out) std::memcpy(centroids_out, cents.data(), sizeof(double) * cents.size());
  return AS_OK;
}

int as_calinski_harabasz(as_ctx *ctx, const double *rows, uint64_t n_rows, uint64_t n_features, const uint64_t *assignments,
                         uint64_t k, double *score_out) {
  AS_TRY(pre_check(ctx));
  if (!rows || !assignments || !score_out) return fail(ctx, AS_ERR_ARG, "null argument");
  HostRows H;
  AS_TRY(H.get(ctx, rows, is_device_ptr(rows), 0, n_rows, n_features));
  const std::vector<uint64_t> a(assignments, assignments + n_rows);
  *score_out = calinski_harabasz(H.p, n_rows, n_features, a, k);
  return AS_OK;
}

// ClusteringHeuristic::compute_optimal_k, src/clustering.rs:36-72
int as_compute_optimal_k(as_ctx *ctx, const double *rows, uint64_t n, uint64_t f, uint64_t seed, uint64_t *k_out, double *radius_out,
                         uint64_t *id_out) {
  AS_TRY(pre_check(ctx));
  if (!rows || !k_out || !radius_out || !id_out) return fail(ctx, AS_ERR_ARG, "null argument");
  if (n < 2 || f == 0) return fail(ctx, AS_ERR_SHAPE, "compute_optimal_k needs at least two rows");
  const bool on_dev = is_device_ptr(rows);
  DevIn<double> R;
  AS_TRY(R.bind(ctx, rows, n * f));
  // step 1: bounds from N / F and the Two-NN intrinsic dimension (:75-164)
  uint64_t id_est;
  if (n < 10) {
    id_est = std::min<uint64_t>(f, 2);
  } else {
    const std::vector<uint64_t> ids = sample_indices(n, std::min<uint64_t>(n, 500), seed + 1);
    std::vector<double> ratios(ids.size());
    AS_TRY(two_nn_ratios(ctx, R.p, n, f, ids.data(), ids.size(), ratios.data()));
    id_est = intrinsic_dimension(n, f, ratios.data(), ratios.size());
  }
  uint64_t k_min = std::max<uint64_t>((uint64_t)std::ceil(std::sqrt((double)n / 10.0)), 2);
  uint64_t k_max = std::min({f, n / 10, 5 * id_est, (uint64_t)std::pow((double)n, 0.5)});
  k_max = std::min(std::max(k_max, k_min + 1), n / 2);
  // the <= 1000-row sample of steps 2 and 3 (:52-64), on both sides
  const uint64_t ss = std::min<uint64_t>(n, 1000);
  std::vector<double> sample;
  if (n > ss) {
    AS_TRY(gather_rows_host(ctx, R.p, true, sample_indices(n, ss, seed), f, sample));
  } else {
    HostRows H;
    AS_TRY(H.get(ctx, rows, on_dev, 0, n, f));
    sample.assign(H.p, H.p + n * f);
  }
  DevBuf<double> sd;
  AS_TRY(sd.alloc(ctx, ss * f));
  AS_HIP(ctx, copy_h2d(ctx, sd.p, sample.data(), sizeof(double) * ss * f));
  uint64_t k_opt = k_min;
  AS_TRY(step2_calinski_harabasz(ctx, sd.p, sample.data(), ss, f, k_min, k_max, seed, &k_opt));
  AS_TRY(threshold_from_pilot(ctx, sd.p, sample.data(), ss, f, k_opt, seed, radius_out));
  *k_out = k_opt;
  *id_out = id_est;
  return AS_OK;
}

// run_incremental_clustering_with_sampling without a sampler, src/clustering.rs:547-910, for a batch size (see the top of
// the file and oracle/prestage.c): nearest centroids of a batch against the snapshot on the device, decisions and running
// means in row order on the host.
int as_incremental_clustering(as_ctx *ctx, const double *rows, uint64_t n, uint64_t f, uint64_t max_clusters, double radius,
                              uint64_t batch, double *centroids_out, uint64_t *n_clusters_out, int64_t *assignments_out,
                              uint64_t *counts_out) {
  AS_TRY(pre_check(ctx));
  if (!rows || !centroids_out || !n_clusters_out || !assignments_out || !counts_out || max_clusters == 0)
    return fail(ctx, AS_ERR_ARG, "null argument");
  if (batch == 0) batch = 1;
  const bool on_dev = is_device_ptr(rows);
  DevIn<double> R;
  AS_TRY(R.bind(ctx, rows, n * f));
  DevBuf<double> cd, d2d;
  DevBuf<uint64_t> idxd;
  AS_TRY(cd.alloc(ctx, max_clusters * f));
  AS_TRY(d2d.alloc(ctx, batch));
  AS_TRY(idxd.alloc(ctx, batch));
  std::vector<double> snap_d2(batch);
  std::vector<uint64_t> snap_i(batch);
  double *cents = centroids_out;
  uint64_t nc = 0;
  for (uint64_t i = 0; i < n; ++i) assignments_out[i] = -1;
  HostRows H;
  // device state of the phase in which every centroid exists (no row of it visits the host)
  DevBuf<unsigned long long> counts_d;
  DevBuf<long long> assign_d;
  DevBuf<uint32_t> act_d;
  bool on_device_phase = false;
  uint64_t first_device_row = 0;
  ctx->stats["clustering_device_batches"] = 0;
  ctx->stats["clustering_host_batches"] = 0;
  ctx->stats["clustering_sequential_batches"] = 0;
  ctx->stats["nearest_mfma_calls"] = 0;
  if (ctx->opt_cluster_on_device && f <= kSeqMaxF) {
    // No row visits the host.  Batches in which centroids can still appear: cluster_seq_kernel (a chain over the rows of the
    // batch); once the cap is reached: decisions and running means per centroid in parallel.  batch = 1 (every row sees every
    // centroid as it stands) is ONE chain over all rows with no snapshot, cut into launches of 65 536 rows.
    DevBuf<double> cdT;
    DevBuf<unsigned long long> nc_d;
    AS_TRY(cdT.alloc(ctx, max_clusters * f));
    AS_TRY(nc_d.alloc(ctx, 2));
    AS_TRY(counts_d.alloc(ctx, max_clusters));
    AS_TRY(assign_d.alloc(ctx, n));
    AS_TRY(act_d.alloc(ctx, batch));
    AS_HIP(ctx, hipMemsetAsync(counts_d.p, 0, sizeof(unsigned long long) * max_clusters, ctx->stream));
    AS_HIP(ctx, hipMemsetAsync(nc_d.p, 0, sizeof(unsigned long long) * 2, ctx->stream));
    AS_HIP(ctx, hipMemsetAsync(cd.p, 0, sizeof(double) * max_clusters * f, ctx->stream));
    const size_t lds = sizeof(double) * 2 * f;
    DevBuf<unsigned long long> stamps_d;  // debug bit 0x100000: cycles per phase of the chain kernel (diagnostic only)
    if (ctx->opt_debug & 0x100000) {
      AS_TRY(stamps_d.alloc(ctx, kSeqWaves * 4));
      AS_HIP(ctx, hipMemsetAsync(stamps_d.p, 0, sizeof(unsigned long long) * kSeqWaves * 4, ctx->stream));
    }
    AS_HIP(ctx, hipFuncSetAttribute((const void *)cluster_seq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // the nearest-centroid searches of the batches take their candidates from the f32 matrix pipe (centroids.hip): one bound
    // of |row|^2 for the whole run (a centroid is a running mean of rows: inside their hull up to rounding)
    double qmax2 = 0.0;
    if (ctx->opt_nearest_mfma && batch >= 1024 && n >= 1024) {
      AS_TRY(max_row_sqnorm(ctx, R.p, n, f, &qmax2));
      qmax2 *= 1.0 + 1e-6;
    }
    const uint64_t step = batch == 1 ? 65536 : batch;
    uint64_t done = 0;
    for (uint64_t b0 = 0; b0 < n; b0 += done) {
      const uint64_t bn = std::min(step, n - b0);
      done = bn;
      if (nc == max_clusters && batch != 1) {
        AS_TRY(nearest_centroids(ctx, R.p + b0 * f, bn, f, cd.p, nc, idxd.p, d2d.p, qmax2));
        hipLaunchKernelGGL(cluster_decide_kernel, dim3(cdiv(bn, 256)), dim3(256), 0, ctx->stream, (const uint64_t *)idxd.p,
                           (const double *)d2d.p, bn, radius, act_d.p, assign_d.p + b0);
        hipLaunchKernelGGL(cluster_apply_kernel, dim3((unsigned)nc), dim3(256), 0, ctx->stream, R.p + b0 * f, bn, f,
                           (const uint64_t *)idxd.p, (const uint32_t *)act_d.p, cd.p, counts_d.p);
        AS_HIP(ctx, hipGetLastError());
        ctx->stats["clustering_device_batches"] += 1;
        continue;
      }
      const bool snap = batch != 1 && nc > 0;
      if (snap) AS_TRY(nearest_centroids(ctx, R.p + b0 * f, bn, f, cd.p, nc, idxd.p, d2d.p, qmax2));
      {
        ProfScope ps(ctx, "cluster_sequential");
        hipLaunchKernelGGL(cluster_seq_kernel, dim3(1), dim3(kSeqThreads), lds, ctx->stream, R.p + b0 * f, bn, (uint32_t)f,
                           snap ? (const uint64_t *)idxd.p : nullptr, snap ? (const double *)d2d.p : nullptr, batch == 1 ? 0 : nc, radius,
                           max_clusters, cd.p, cdT.p, counts_d.p, assign_d.p + b0, nc_d.p, batch != 1 ? 1 : 0, stamps_d.p);
      }
      AS_HIP(ctx, hipGetLastError());
      ctx->stats["clustering_sequential_batches"] += 1;
      if (batch != 1 || b0 + step >= n) {  // the next batch's path and start depend on the cluster count and on where this one ended
        unsigned long long h[2] = {0, 0};
        AS_HIP(ctx, copy_d2h(ctx, h, nc_d.p, sizeof(h)));
        AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        nc = h[0];
        if (batch != 1) done = h[1];
        if (done == 0 || done > bn) return fail(ctx, AS_ERR_HIP, "clustering chain: row count");
      }
    }
    if (stamps_d.p) {
      unsigned long long hs[kSeqWaves * 4];
      AS_HIP(ctx, copy_d2h(ctx, hs, stamps_d.p, sizeof(hs)));
      AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
      for (int w = 0; w < kSeqWaves; ++w)
        for (int q = 0; q < 4; ++q) ctx->stats["clustering_chain_cycles_w" + std::to_string(w) + "_p" + std::to_string(q)] = (double)hs[w * 4 + q];
    }
    std::vector<unsigned long long> hc(nc);
    if (nc) {
      AS_HIP(ctx, copy_d2h(ctx, cents, cd.p, sizeof(double) * nc * f));
      AS_HIP(ctx, copy_d2h(ctx, hc.data(), counts_d.p, sizeof(unsigned long long) * nc));
    }
    static_assert(sizeof(long long) == sizeof(int64_t), "assignments");
    if (n) AS_HIP(ctx, copy_d2h(ctx, assignments_out, assign_d.p, sizeof(long long) * n));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (uint64_t c = 0; c < nc; ++c) counts_out[c] = hc[c];
    *n_clusters_out = nc;
    return AS_OK;
  }
  uint64_t done_h = 0;
  for (uint64_t b0 = 0; b0 < n; b0 += done_h) {
    const uint64_t bn = std::min(batch, n - b0), nc_snap = nc;
    done_h = bn;
    if (nc == max_clusters && ctx->opt_cluster_on_device) {
      if (!on_device_phase) {  // the state so far moves to the device once
        AS_TRY(counts_d.alloc(ctx, max_clusters));
        AS_TRY(assign_d.alloc(ctx, n));
        AS_TRY(act_d.alloc(ctx, batch));
        std::vector<unsigned long long> hc(max_clusters);
        for (uint64_t c = 0; c < max_clusters; ++c) hc[c] = counts_out[c];
        AS_HIP(ctx, copy_h2d(ctx, counts_d.p, hc.data(), sizeof(unsigned long long) * max_clusters));
        AS_HIP(ctx, copy_h2d(ctx, cd.p, cents, sizeof(double) * max_clusters * f));
        on_device_phase = true;
        first_device_row = b0;
      }
      AS_TRY(nearest_centroids(ctx, R.p + b0 * f, bn, f, cd.p, nc, idxd.p, d2d.p));
      hipLaunchKernelGGL(cluster_decide_kernel, dim3(cdiv(bn, 256)), dim3(256), 0, ctx->stream, (const uint64_t *)idxd.p, (const double *)d2d.p, bn,
                         radius, act_d.p, assign_d.p + b0);
      hipLaunchKernelGGL(cluster_apply_kernel, dim3((unsigned)nc), dim3(256), 0, ctx->stream, R.p + b0 * f, bn, f, (const uint64_t *)idxd.p,
                         (const uint32_t *)act_d.p, cd.p, counts_d.p);
      AS_HIP(ctx, hipGetLastError());
      ctx->stats["clustering_device_batches"] += 1;
      continue;
    }
    ctx->stats["clustering_host_batches"] += 1;
    if (nc_snap == 0) {
      std::fill(snap_d2.begin(), snap_d2.begin() + bn, INFINITY);
      std::fill(snap_i.begin(), snap_i.begin() + bn, 0);
    } else {
      AS_HIP(ctx, copy_h2d(ctx, cd.p, cents, sizeof(double) * nc_snap * f));
      AS_TRY(nearest_centroids(ctx, R.p + b0 * f, bn, f, cd.p, nc_snap, idxd.p, d2d.p));
      AS_HIP(ctx, copy_d2h(ctx, snap_i.data(), idxd.p, sizeof(uint64_t) * bn));
      AS_HIP(ctx, copy_d2h(ctx, snap_d2.data(), d2d.p, sizeof(double) * bn));
      AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    AS_TRY(H.get(ctx, rows, on_dev, b0, bn, f));
    for (uint64_t r = 0; r < bn; ++r) {
      const double *row = H.p + r * f;
      uint64_t bi = snap_i[r];
      double bd = snap_d2[r];
      for (uint64_t c = nc_snap; c < nc; ++c) {  // centroids created earlier in this batch, current values
        const double d = dist2(row, cents + c * f, f);
        if (d < bd) {
          bd = d;
          bi = c;
        }
      }
      if (nc == 0) {
        std::memcpy(cents, row, sizeof(double) * f);
        counts_out[0] = 1;
        assignments_out[b0 + r] = 0;
        nc = 1;
      } else if (nc < max_clusters && bd > radius * 0.5) {
        std::memcpy(cents + nc * f, row, sizeof(double) * f);
        counts_out[nc] = 1;
        assignments_out[b0 + r] = (int64_t)nc;
        ++nc;
      } else if (bd <= radius) {
        const double k_new = (double)counts_out[bi] + 1.0;
        for (uint64_t j = 0; j < f; ++j) cents[bi * f + j] += (row[j] - cents[bi * f + j]) / k_new;
        counts_out[bi] += 1;
        assignments_out[b0 + r] = (int64_t)bi;
      } else if (bd <= radius * 1.5) {
        counts_out[bi] += 1;
        assignments_out[b0 + r] = (int64_t)bi;
      }
      if (nc_snap < max_clusters && nc == max_clusters) {  // the cap was reached inside this batch: it ends here
        done_h = r + 1;
        break;
      }
    }
  }
  if (on_device_phase) {  // the device phase's centroids, counts and assignments come back once
    std::vector<unsigned long long> hc(max_clusters);
    AS_HIP(ctx, copy_d2h(ctx, cents, cd.p, sizeof(double) * max_clusters * f));
    AS_HIP(ctx, copy_d2h(ctx, hc.data(), counts_d.p, sizeof(unsigned long long) * max_clusters));
    for (uint64_t c = 0; c < max_clusters; ++c) counts_out[c] = hc[c];
    // (assignments of the host phase are in assignments_out already: the device phase covers the batches from its first on)
    const uint64_t first = first_device_row;
    if (first < n) AS_HIP(ctx, copy_d2h(ctx, assignments_out + first, assign_d.p + first, sizeof(long long) * (n - first)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  *n_clusters_out = nc;
  return AS_OK;
}

// compute_jl_dimension, src/reduction.rs:126-140
uint64_t as_jl_dimension(uint64_t n_points, double epsilon) {
  const double log_n = std::log((double)n_points);
  const double eps_sq = std::pow(epsilon, 2.0);
  const uint64_t jl = (uint64_t)std::ceil(8.0 * log_n / eps_sq);
  return std::max<uint64_t>(jl, 32);
}

int as_jl_create(as_ctx *ctx, uint64_t original_dim, uint64_t reduced_dim, uint64_t seed, as_jl **out) {
  AS_TRY(pre_check(ctx));
  if (!out || original_dim == 0 || reduced_dim == 0) return fail(ctx, AS_ERR_ARG, "bad projection shape");
  as_jl *j = new as_jl();
  j->device = ctx->device;
  j->original_dim = original_dim;
  j->reduced_dim = reduced_dim;
  j->seed = seed;
  AS_HIP(ctx, handle_malloc((void **)&j->G, sizeof(double) * original_dim * reduced_dim));
  const uint64_t total = original_dim * reduced_dim;
  hipLaunchKernelGGL(jl_gaussian_kernel, dim3(cdiv(total, 256)), dim3(256), 0, ctx->stream, seed, total, j->G);
  AS_HIP(ctx, hipGetLastError());
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *out = j;
  return AS_OK;
}

void as_jl_destroy(as_jl *jl) {
  if (!jl) return;
  (void)hipSetDevice(jl->device);
  (void)handle_free(jl->G);
  delete jl;
}

uint64_t as_jl_original_dim(const as_jl *jl) { return jl ? jl->original_dim : 0; }
uint64_t as_jl_reduced_dim(const as_jl *jl) { return jl ? jl->reduced_dim : 0; }

int as_jl_matrix(as_ctx *ctx, const as_jl *jl, double *g_out) {
  AS_TRY(pre_check(ctx));
  if (!jl || !g_out) return fail(ctx, AS_ERR_ARG, "null argument");
  const size_t bytes = sizeof(double) * jl->original_dim * jl->reduced_dim;
  if (is_device_ptr(g_out)) AS_HIP(ctx, hipMemcpyAsync(g_out, jl->G, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  else AS_HIP(ctx, copy_d2h(ctx, g_out, jl->G, bytes));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return AS_OK;
}

// project_matrix / ImplicitProjection::project, src/reduction.rs:142-203 (batched)
int as_jl_project(as_ctx *ctx, const as_jl *jl, const double *rows, uint64_t n_rows, double *out) {
  AS_TRY(pre_check(ctx));
  if (!jl || (n_rows && (!rows || !out))) return fail(ctx, AS_ERR_ARG, "null argument");
  if (n_rows == 0) return AS_OK;
  const uint64_t d = jl->original_dim, r = jl->reduced_dim;
  if (d * sizeof(double) > 64 * 1024) return fail(ctx, AS_ERR_UNSUPPORTED, "original dimension too large for the projection kernel");
  DevIn<double> X;
  AS_TRY(X.bind(ctx, rows, n_rows * d));
  DevOut<double> O;
  AS_TRY(O.bind(ctx, out, n_rows * r));
  {
    ProfScope ps(ctx, "jl_project");
    hipLaunchKernelGGL(jl_project_kernel, dim3((unsigned)n_rows), dim3(256), d * sizeof(double), ctx->stream, X.p, n_rows, d, jl->G, r,
                       1.0 / std::sqrt((double)r), O.p);
    AS_HIP(ctx, hipGetLastError());
  }
  AS_TRY(O.finish(ctx));
  return AS_OK;
}

}  // extern "C"
