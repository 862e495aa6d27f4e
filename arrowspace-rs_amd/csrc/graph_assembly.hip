// graph_assembly.hip -- K3..K5: neighbour table -> eps filter + kernel weights (+ optional 50 %
// pruning) -> union symmetrisation -> CSR Laplacian with explicit diagonal.
// Replaces src/laplacian.rs:217-417 (_build_adjacency's post-processing,
// _symmetrise_adjancency, _build_sparse_laplacian) and TriMat::to_csr (:160).
//
// The reference symmetrises through a concurrent hash map and gathers each row by scanning the
// whole map (O(n*E)).  Here: count in-edges, prefix-sum, scatter reverse edges, per-row union,
// per-row sort.  All of it is HBM-bound integer/byte work (n*(topk+1)*16 B in, nnz*16 B out).
#include "common.h"

namespace asx {

constexpr int kMaxK = 256;  // neighbours per row of the table (topk + 1)

// ---- device exclusive scan (u32 in, u64 out, out[n] = total) -------------------------------
constexpr int kScanBlock = 1024;

__global__ __launch_bounds__(kScanBlock) void scan_block_sums(const uint32_t *__restrict__ in, uint64_t n,
                                                              uint64_t *__restrict__ bsum) {
  __shared__ uint64_t s[kScanBlock / 64];
  const uint64_t i = (uint64_t)blockIdx.x * kScanBlock + threadIdx.x;
  uint64_t v = i < n ? in[i] : 0;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t t = 0;
    for (int w = 0; w < kScanBlock / 64; ++w) t += s[w];
    bsum[blockIdx.x] = t;
  }
}

// exclusive scan of the block sums in place, one workgroup of 1024: chunks of 1024 sums, a wave-level inclusive scan and the
// waves' totals through LDS, the running total carried from chunk to chunk.  (It was one thread walking the sums: a dependent
// global load per block, 0.1 ms per thousand blocks, six scans per build.)
__global__ __launch_bounds__(1024) void scan_of_sums(uint64_t *bsum, uint64_t nb, uint64_t *total_out) {
  __shared__ uint64_t wsum[16];
  __shared__ uint64_t s_run;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_run = 0;
  __syncthreads();
  for (uint64_t base = 0; base < nb; base += 1024) {
    const uint64_t i = base + threadIdx.x;
    const uint64_t v = i < nb ? bsum[i] : 0;
    uint64_t inc = v;
    for (int o = 1; o < 64; o <<= 1) {
      const uint64_t t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    uint64_t before = s_run;
    for (int w = 0; w < wv; ++w) before += wsum[w];
    if (i < nb) bsum[i] = before + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) s_run = before + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_out = s_run;
}

__global__ __launch_bounds__(kScanBlock) void scan_finish(const uint32_t *__restrict__ in, uint64_t n,
                                                          const uint64_t *__restrict__ bsum,
                                                          uint64_t *__restrict__ out) {
  __shared__ uint64_t wsum[kScanBlock / 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t i = (uint64_t)blockIdx.x * kScanBlock + threadIdx.x;
  const uint64_t v = i < n ? in[i] : 0;
  uint64_t inc = v;
  for (int o = 1; o < 64; o <<= 1) {
    uint64_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  uint64_t base = bsum[blockIdx.x];
  for (int w = 0; w < wv; ++w) base += wsum[w];
  if (i < n) out[i] = base + inc - v;
}

int exclusive_scan_u32_u64(as_ctx *ctx, const uint32_t *in, uint64_t n, uint64_t *out) {
  const uint64_t nb = (n + kScanBlock - 1) / kScanBlock;
  DevBuf<uint64_t> bsum;
  AS_TRY(bsum.alloc(ctx, nb + 1));
  if (n == 0) {
    AS_HIP(ctx, hipMemsetAsync(out, 0, sizeof(uint64_t), ctx->stream));
    return AS_OK;
  }
  hipLaunchKernelGGL(scan_block_sums, dim3((unsigned)nb), dim3(kScanBlock), 0, ctx->stream, in, n, bsum.p);
  hipLaunchKernelGGL(scan_of_sums, dim3(1), dim3(1024), 0, ctx->stream, bsum.p, nb, out + n);
  hipLaunchKernelGGL(scan_finish, dim3((unsigned)nb), dim3(kScanBlock), 0, ctx->stream, in, n, bsum.p, out);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

// ---- K3a: degrees (src/laplacian.rs:217-227) -------------------------------------------------
__global__ void degree_kernel(const uint32_t *__restrict__ nbr_idx, const double *__restrict__ nbr_dist,
                              uint32_t n, uint32_t K, double eps, uint32_t *__restrict__ deg,
                              unsigned long long *__restrict__ deg_sum) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t c = 0;
  if (i < n) {
    for (uint32_t s = 0; s < K; ++s) {
      const uint32_t j = nbr_idx[(uint64_t)i * K + s];
      if (j != kNoIdx && j != i && nbr_dist[(uint64_t)i * K + s] <= eps) ++c;
    }
    deg[i] = c;
  }
  unsigned long long v = c;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  block_add_u64(deg_sum, 1u, v);  // (one add per block: common.h)
}

// ---- K3b: weights + optional pruning (src/laplacian.rs:243-288) ------------------------------
__device__ __forceinline__ double kernel_weight(double dist, double sigma, double p) {
  // 1 / (1 + (dist/sigma)^p).  p = 2 and p = 1 are evaluated as the correctly rounded
  // product / identity (what a correctly rounded powf returns); other exponents use pow().
  const double x = __ddiv_rn(dist, sigma);
  double xp;
  if (p == 2.0) xp = __dmul_rn(x, x);
  else if (p == 1.0) xp = x;
  else xp = pow(x, p);
  return __ddiv_rn(1.0, __dadd_rn(1.0, xp));
}

template <int KMAX>
__global__ void adjacency_kernel(const uint32_t *__restrict__ nbr_idx, const double *__restrict__ nbr_dist,
                                 uint32_t n, uint32_t K, double eps, double sigma, double p,
                                 const uint32_t *__restrict__ deg,
                                 const unsigned long long *__restrict__ deg_sum,
                                 uint32_t *__restrict__ adj_j, double *__restrict__ adj_w,
                                 uint32_t *__restrict__ adj_len, uint32_t *__restrict__ cnt_in, uint32_t r0, uint32_t nr) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // avg_degree > 10.0 (src/laplacian.rs:229-230)
  const bool sparsify = ((double)(*deg_sum) / (double)n) > 10.0;
  uint32_t vj[KMAX];  // K <= KMAX (64 for the usual depths, 256 beyond: per-thread scratch)
  double vw[KMAX], vs[KMAX];
  uint32_t len = 0;
  for (uint32_t s = 0; s < K; ++s) {
    const uint32_t j = nbr_idx[(uint64_t)i * K + s];
    const double dist = nbr_dist[(uint64_t)i * K + s];
    if (j == kNoIdx || j == i || !(dist <= eps)) continue;
    const double w = kernel_weight(dist, sigma, p);
    if (w > 1e-12) {
      vj[len] = j;
      vw[len] = w;
      vs[len] = sparsify ? __dmul_rn(w, __dsqrt_rn((double)((uint64_t)deg[i] * (uint64_t)deg[j]))) : w;
      ++len;
    }
  }
  if (sparsify && len > 2) {
    // stable insertion sort by score descending (reference: sort_unstable_by, ties unspecified)
    for (uint32_t a = 1; a < len; ++a) {
      const uint32_t tj = vj[a];
      const double tw = vw[a], ts = vs[a];
      uint32_t b = a;
      while (b > 0 && vs[b - 1] < ts) {
        vj[b] = vj[b - 1];
        vw[b] = vw[b - 1];
        vs[b] = vs[b - 1];
        --b;
      }
      vj[b] = tj;
      vw[b] = tw;
      vs[b] = ts;
    }
    uint32_t keep = len / 2;
    if (keep < 1) keep = 1;
    len = keep;
  }
  adj_len[i] = len;
  for (uint32_t s = 0; s < len; ++s) {
    adj_j[(uint64_t)i * K + s] = vj[s];
    adj_w[(uint64_t)i * K + s] = vw[s];
    if (vj[s] - r0 < nr) atomicAdd(&cnt_in[vj[s] - r0], 1u);  // in-edges of the rows [r0, r0 + nr) only
  }
}

// ---- K3a / K3b with 32 lanes per row (K <= 32: every depth the reference's defaults reach) -----------------------------------
// One thread per row reads its K entries 68 bytes apart from its neighbours' and keeps three K-element arrays in scratch for the
// pruning's insertion sort; here lane s of a half wave holds entry s of the row: the table is read side by side, the stable
// order by score (descending, equal scores in their original order -- what the insertion sort with a strict '<' leaves) is a
// rank count over the half wave, and nothing leaves the registers.
__device__ __forceinline__ uint32_t half_ballot(bool p, int half) { return (uint32_t)(__ballot(p) >> (32 * half)); }

// R rows per half wave with their loads issued together (measured at 1 M rows: R = 1 0.41 ms, R = 2 0.47 ms -- the kernel is bound
// by its f64 divisions, the square root and the shuffles of the ranking, not by the round trips: R = 1 is what runs).
// BY_ID (the row-sharded build): the table holds the n rows of ONE rank, row t of it is node row_ids[t] of the n_nodes of the graph;
// deg is by node, the outputs stay by row of the table (no in-edge counting here: the ranks exchange the adjacency first).
template <int R, bool BY_ID = false>
__global__ __launch_bounds__(256) void adjacency32_kernel(const uint32_t *__restrict__ nbr_idx, const double *__restrict__ nbr_dist, uint32_t n, uint32_t K,
                                                          double eps, double sigma, double p, const uint32_t *__restrict__ deg,
                                                          const unsigned long long *__restrict__ deg_sum, uint32_t *__restrict__ adj_j,
                                                          double *__restrict__ adj_w, uint32_t *__restrict__ adj_len, uint32_t *__restrict__ cnt_in,
                                                          uint32_t r0, uint32_t nr, uint32_t *__restrict__ adj_slot,
                                                          const uint32_t *__restrict__ row_ids = nullptr, uint32_t n_nodes = 0) {
  // adj_slot (or nullptr): the place of each kept edge among the in-edges of its target, as the counting atomic returns it -- the
  // scatter of the reverse edges then needs no second round of atomics (fill_reverse32_kernel)
  const int lane = threadIdx.x & 63, half = lane >> 5, s = lane & 31;
  const uint32_t i0 = ((blockIdx.x * blockDim.x + threadIdx.x) >> 5) * R;
  const bool sparsify = ((double)(*deg_sum) / (double)(BY_ID ? n_nodes : n)) > 10.0;  // avg_degree > 10.0 (src/laplacian.rs:229-230)
  uint32_t j[R], di[R], dj[R], pos[R], len[R], node[R];
  double dist[R], w[R];
  bool live[R], keep[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t i = i0 + r;
    live[r] = i < n && (uint32_t)s < K;
    const uint64_t at = live[r] ? (uint64_t)i * K + s : 0;
    j[r] = nbr_idx[at];
    dist[r] = nbr_dist[at];
    node[r] = BY_ID ? row_ids[i < n ? i : 0] : i;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t i = i0 + r;
    keep[r] = live[r] && j[r] != kNoIdx && j[r] != node[r] && dist[r] <= eps;
    di[r] = dj[r] = 0;
    if (sparsify) {
      di[r] = deg[i < n ? node[r] : 0];
      dj[r] = deg[keep[r] ? j[r] : 0];
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    double score = 0.0;
    w[r] = 0.0;
    if (keep[r]) {
      w[r] = kernel_weight(dist[r], sigma, p);
      keep[r] = w[r] > 1e-12;
      if (keep[r]) score = sparsify ? __dmul_rn(w[r], __dsqrt_rn((double)((uint64_t)di[r] * (uint64_t)dj[r]))) : w[r];
    }
    const uint32_t km = half_ballot(keep[r], half);
    len[r] = (uint32_t)__popc(km);
    pos[r] = (uint32_t)__popc(km & ((1u << s) - 1u));  // place among the kept entries, in table order
    if (sparsify && len[r] > 2) {
      // place in the stable descending order by score: kept entries with a larger score, or an equal one earlier in the table
      uint32_t rank = 0;
      for (int t = 0; t < (int)K; ++t) {  // (entries past K are never kept)
        const double so = __shfl(score, t, 32);
        const bool ko = (km >> t) & 1u;
        rank += (ko && (so > score || (so == score && t < s))) ? 1u : 0u;
      }
      uint32_t keepn = len[r] / 2;
      if (keepn < 1) keepn = 1;
      keep[r] = keep[r] && rank < keepn;
      pos[r] = rank;
      len[r] = keepn;
    }
  }
  uint32_t slot[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    slot[r] = 0;
    if (!BY_ID && keep[r] && j[r] - r0 < nr) slot[r] = atomicAdd(&cnt_in[j[r] - r0], 1u);  // in-edges of the rows [r0, r0 + nr) only
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t i = i0 + r;
    if (i < n) {
      if (s == 0) adj_len[i] = len[r];
      if (keep[r]) {
        adj_j[(uint64_t)i * K + pos[r]] = j[r];
        adj_w[(uint64_t)i * K + pos[r]] = w[r];
        if (!BY_ID && adj_slot && j[r] - r0 < nr) adj_slot[(uint64_t)i * K + pos[r]] = slot[r];
      }
    }
  }
}

// ---- the round-4 form of the stages around it (K <= 32, "assembly_lanes" 2) -----------------------------------------------------
// Degrees: a wave takes 64 rows = 64 K consecutive entries of the table and reads them side by side (the 32-lanes-per-row form
// leaves 15 of 32 lanes idle at K = 17 and moved 0.8 TB/s); the hit bits of the K loads go through LDS as K ballots, lane r counts
// the K bits of row r.
// (LDS traffic between the lanes of one wave: the hardware keeps a wave's LDS operations in order, the compiler is told not to move
// memory operations across this point)
__device__ __forceinline__ void lanes_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__global__ __launch_bounds__(256) void degree_rows64_kernel(const uint32_t *__restrict__ nbr_idx, const double *__restrict__ nbr_dist, uint32_t n, uint32_t K,
                                                            double eps, uint32_t *__restrict__ deg, unsigned long long *__restrict__ deg_sum) {
  __shared__ unsigned long long s_bits[4][34];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned long long total = 0;
  for (uint64_t row0 = ((uint64_t)blockIdx.x * 4 + wv) * 64; row0 < n; row0 += (uint64_t)gridDim.x * 256) {
    const uint64_t e0 = row0 * K, e_end = min((uint64_t)n, row0 + 64) * K;
    for (uint32_t t = 0; t < K; ++t) {
      const uint64_t e = e0 + (uint64_t)t * 64 + lane;
      bool hit = false;
      if (e < e_end) {
        const uint32_t j = nbr_idx[e];
        const uint32_t i = (uint32_t)row0 + (t * 64u + (uint32_t)lane) / K;
        hit = j != kNoIdx && j != i && nbr_dist[e] <= eps;
      }
      const unsigned long long b = __ballot(hit);
      if (lane == 0) s_bits[wv][t] = b;
    }
    if (lane == 0) s_bits[wv][K] = 0;
    lanes_sync();
    const uint32_t p = (uint32_t)lane * K, w0 = p >> 6, sh = p & 63u;
    unsigned long long v = s_bits[wv][w0] >> sh;
    if (sh) v |= s_bits[wv][w0 + 1] << (64u - sh);
    const uint32_t c = (uint32_t)__builtin_popcountll(v & ((1ull << K) - 1ull));
    if (row0 + lane < n) deg[row0 + lane] = c;
    total += c;
    lanes_sync();
  }
  for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, 64);
  block_add_u64(deg_sum, 1u, total);  // (one add per block: common.h)
}

// a reverse edge as the scatter stores it: one 16-byte write instead of a 4-byte and an 8-byte one in two arrays
struct __align__(16) RevEdge {
  uint32_t src, pad;
  double w;
};

// Reverse edges, 32 lanes per row: the kept edges of a row are read side by side, an edge goes to the place the counting atomic of
// adjacency32_kernel gave it (no cursor atomics; one thread per row with a returning atomic per edge took 0.61 ms at 1 M rows)
template <int R>
__global__ __launch_bounds__(256) void fill_reverse32_kernel(const uint32_t *__restrict__ adj_j, const double *__restrict__ adj_w,
                                                             const uint32_t *__restrict__ adj_slot, const uint32_t *__restrict__ adj_len, uint32_t n,
                                                             uint32_t K, const uint64_t *__restrict__ rev_ptr, RevEdge *__restrict__ rev, uint32_t r0,
                                                             uint32_t nr) {
  // (R rows per half wave, each round of loads issued for all of them: three dependent round trips per edge)
  const uint32_t s = threadIdx.x & 31u;
  const uint32_t i0 = ((blockIdx.x * blockDim.x + threadIdx.x) >> 5) * R;
  uint32_t len[R], j[R], slot[R];
  double w[R];
  uint64_t base[R];
  bool on[R];
#pragma unroll
  for (int r = 0; r < R; ++r) len[r] = i0 + r < n ? adj_len[i0 + r] : 0u;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    on[r] = s < len[r];
    const uint64_t at = on[r] ? (uint64_t)(i0 + r) * K + s : 0;
    j[r] = adj_j[at] - r0;
    slot[r] = adj_slot[at];
    w[r] = adj_w[at];
    on[r] = on[r] && j[r] < nr;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) base[r] = rev_ptr[on[r] ? j[r] : 0];
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (on[r]) {
      RevEdge e;
      e.src = i0 + r;
      e.pad = 0;
      e.w = w[r];
      rev[base[r] + slot[r]] = e;
    }
}

// union size of a row over the packed reverse edges (union_count32_kernel's walk)
__global__ __launch_bounds__(256) void union_count32r_kernel(const uint32_t *__restrict__ adj_j, const uint32_t *__restrict__ adj_len, uint32_t nr, uint32_t K,
                                                             const uint64_t *__restrict__ rev_ptr, const RevEdge *__restrict__ rev,
                                                             uint32_t *__restrict__ ucnt, uint32_t r0) {
  const int lane = threadIdx.x & 63, half = lane >> 5, s = lane & 31;
  const uint32_t li = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
  const bool live = li < nr;
  const uint32_t i = r0 + (live ? li : 0u);
  const uint32_t len = live ? adj_len[i] : 0u;
  const uint32_t mine = (uint32_t)s < len ? adj_j[(uint64_t)i * K + s] : kNoIdx;
  const uint64_t e0 = live ? rev_ptr[li] : 0, e1 = live ? rev_ptr[li + 1] : 0;
  uint32_t c = len;
  const uint64_t span = e1 - e0;
  const uint64_t span_o = __shfl_xor((unsigned long long)span, 32, 64);
  const uint64_t trips = ((span > span_o ? span : span_o) + 31) / 32;
  const uint32_t len_o = __shfl_xor(len, 32, 64), len_m = len > len_o ? len : len_o;
  for (uint64_t t = 0; t < trips; ++t) {
    const uint64_t e = e0 + t * 32 + s;
    const bool have = e < e1;
    const uint32_t src = have ? rev[e].src : kNoIdx;
    bool found = false;
    for (uint32_t q = 0; q < len_m; ++q) {
      const uint32_t a = __shfl(mine, (int)q, 32);
      found = found || (q < len && a == src);
    }
    c += (uint32_t)__popc(half_ballot(have && !found, half));
  }
  if (live && s == 0) ucnt[li] = c;
}

// ---- the row-sharded form of the first half (sharded.hip): a rank ranks and prunes the edges of ITS rows only ------------------
// degrees of the rows of one rank's table (row t is node row_ids[t])
__global__ void degree_by_ids_kernel(const uint32_t *__restrict__ nbr_idx, const double *__restrict__ nbr_dist, const uint32_t *__restrict__ row_ids,
                                     uint32_t n, uint32_t K, double eps, uint32_t *__restrict__ deg) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const uint32_t i = row_ids[t];
  uint32_t c = 0;
  for (uint32_t s = 0; s < K; ++s) {
    const uint32_t j = nbr_idx[(uint64_t)t * K + s];
    if (j != kNoIdx && j != i && nbr_dist[(uint64_t)t * K + s] <= eps) ++c;
  }
  deg[t] = c;
}
// degrees by position of the image order -> by node, and their sum
__global__ void degree_to_nodes_kernel(const uint32_t *__restrict__ deg_pos, const uint32_t *__restrict__ perm, uint32_t n, uint32_t *__restrict__ deg,
                                       unsigned long long *__restrict__ deg_sum) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long v = 0;
  if (p < n) {
    v = deg_pos[p];
    deg[perm[p]] = (uint32_t)v;
  }
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  block_add_u64(deg_sum, 1u, v);
}
// adjacency by position -> by node (32 lanes per row)
__global__ __launch_bounds__(256) void adjacency_to_nodes_kernel(const uint32_t *__restrict__ pj, const double *__restrict__ pw, const uint32_t *__restrict__ plen,
                                                                 const uint32_t *__restrict__ perm, uint32_t n, uint32_t K, uint32_t *__restrict__ adj_j,
                                                                 double *__restrict__ adj_w, uint32_t *__restrict__ adj_len) {
  const uint32_t s = threadIdx.x & 31u, p = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
  if (p >= n) return;
  const uint32_t i = perm[p], len = plen[p];
  if (s == 0) adj_len[i] = len;
  if (s < len) {
    adj_j[(uint64_t)i * K + s] = pj[(uint64_t)p * K + s];
    adj_w[(uint64_t)i * K + s] = pw[(uint64_t)p * K + s];
  }
}
// in-edges of the rows [r0, r0 + nr) counted over the adjacency of all nodes; an edge keeps the place the atomic returned
__global__ __launch_bounds__(256) void count_in_edges32_kernel(const uint32_t *__restrict__ adj_j, const uint32_t *__restrict__ adj_len, uint32_t n, uint32_t K,
                                                               uint32_t r0, uint32_t nr, uint32_t *__restrict__ cnt_in, uint32_t *__restrict__ adj_slot) {
  const uint32_t s = threadIdx.x & 31u, i = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
  if (i >= n || s >= adj_len[i]) return;
  const uint32_t j = adj_j[(uint64_t)i * K + s] - r0;
  if (j < nr) adj_slot[(uint64_t)i * K + s] = atomicAdd(&cnt_in[j], 1u);
}

// ---- K4: reverse edges ------------------------------------------------------------------------
__global__ void fill_reverse_kernel(const uint32_t *__restrict__ adj_j, const double *__restrict__ adj_w,
                                    const uint32_t *__restrict__ adj_len, uint32_t n, uint32_t K,
                                    const uint64_t *__restrict__ rev_ptr, uint32_t *__restrict__ cursor,
                                    uint32_t *__restrict__ rev_src, double *__restrict__ rev_w, uint32_t r0, uint32_t nr) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t len = adj_len[i];
  for (uint32_t s = 0; s < len; ++s) {
    const uint32_t j = adj_j[(uint64_t)i * K + s] - r0;  // rev_ptr / cursor are indexed by position in the row range
    if (j >= nr) continue;
    const uint64_t pos = rev_ptr[j] + atomicAdd(&cursor[j], 1u);
    rev_src[pos] = i;
    rev_w[pos] = adj_w[(uint64_t)i * K + s];
  }
}

__device__ __forceinline__ bool in_adj(const uint32_t *aj, uint32_t len, uint32_t v) {
  for (uint32_t s = 0; s < len; ++s)
    if (aj[s] == v) return true;
  return false;
}

// union size of row i: out-edges + in-edges whose source is not already an out-neighbour
__global__ void union_count_kernel(const uint32_t *__restrict__ adj_j, const uint32_t *__restrict__ adj_len,
                                   uint32_t nr, uint32_t K, const uint64_t *__restrict__ rev_ptr,
                                   const uint32_t *__restrict__ rev_src, uint32_t *__restrict__ ucnt, uint32_t r0) {
  const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;  // position in the row range
  if (li >= nr) return;
  const uint32_t i = r0 + li;
  const uint32_t len = adj_len[i];
  const uint32_t *aj = adj_j + (uint64_t)i * K;
  uint32_t c = len;
  for (uint64_t e = rev_ptr[li]; e < rev_ptr[li + 1]; ++e)
    if (!in_adj(aj, len, rev_src[e])) ++c;
  ucnt[li] = c;
}

__device__ void sort_row(uint32_t *col, double *w, uint32_t L) {
  if (L <= 32) {
    for (uint32_t a = 1; a < L; ++a) {
      const uint32_t tc = col[a];
      const double tw = w[a];
      uint32_t b = a;
      while (b > 0 && col[b - 1] > tc) {
        col[b] = col[b - 1];
        w[b] = w[b - 1];
        --b;
      }
      col[b] = tc;
      w[b] = tw;
    }
    return;
  }
  // heapsort for long (hub) rows
  auto sift = [&](uint32_t start, uint32_t end) {
    uint32_t root = start;
    while (2 * root + 1 < end) {
      uint32_t child = 2 * root + 1;
      if (child + 1 < end && col[child] < col[child + 1]) ++child;
      if (col[root] >= col[child]) return;
      const uint32_t tc = col[root];
      col[root] = col[child];
      col[child] = tc;
      const double tw = w[root];
      w[root] = w[child];
      w[child] = tw;
      root = child;
    }
  };
  for (int32_t s = (int32_t)(L / 2) - 1; s >= 0; --s) sift((uint32_t)s, L);
  for (uint32_t end = L - 1; end > 0; --end) {
    const uint32_t tc = col[0];
    col[0] = col[end];
    col[end] = tc;
    const double tw = w[0];
    w[0] = w[end];
    w[end] = tw;
    sift(0, end);
  }
}

// ---- K5: per-row union, sort, degree, final CSR (src/laplacian.rs:339-341,369-411) -----------
// serial form of one row (any length): gather into the global scratch, sort there, write
__device__ void write_row_serial(uint32_t i, uint32_t li, const uint32_t *__restrict__ adj_j, const double *__restrict__ adj_w,
                                 const uint32_t *__restrict__ adj_len, uint32_t K, const uint64_t *__restrict__ rev_ptr,
                                 const uint32_t *__restrict__ rev_src, const double *__restrict__ rev_w,
                                 const uint64_t *__restrict__ eptr, uint32_t *__restrict__ ecol, double *__restrict__ ew,
                                 uint64_t *__restrict__ indptr, uint64_t *__restrict__ indices, double *__restrict__ data) {
  const uint32_t len = adj_len[i];
  const uint32_t *aj = adj_j + (uint64_t)i * K;
  const double *aw = adj_w + (uint64_t)i * K;
  uint32_t *col = ecol + eptr[li];
  double *w = ew + eptr[li];
  uint32_t L = 0;
  for (uint32_t s = 0; s < len; ++s) {
    col[L] = aj[s];
    w[L++] = aw[s];
  }
  for (uint64_t e = rev_ptr[li]; e < rev_ptr[li + 1]; ++e) {
    const uint32_t src = rev_src[e];
    if (!in_adj(aj, len, src)) {
      col[L] = src;
      w[L++] = rev_w[e];
    }
  }
  sort_row(col, w, L);
  double degree = 0.0;  // sequential sum in column order (src/laplacian.rs:369)
  for (uint32_t s = 0; s < L; ++s) degree = __dadd_rn(degree, w[s]);
  uint64_t o = eptr[li] + li;
  indptr[li] = o;
  bool diag_done = false;
  for (uint32_t s = 0; s < L; ++s) {
    if (!diag_done && col[s] > i) {
      indices[o] = i;
      data[o++] = degree;
      diag_done = true;
    }
    indices[o] = col[s];
    data[o++] = -w[s];
  }
  if (!diag_done) {
    indices[o] = i;
    data[o] = degree;
  }
}

// W lanes per row (64, or 32 with two rows side by side in a wave): lane e of the row's lanes holds one entry of the union
// (out-edges first, then the in-edges whose source is not an out-neighbour), a bitonic network over the W lanes orders them by
// column, the degree is the sequential sum in column order (src/laplacian.rs:369), and the lanes write the finished row side
// by side.  len / rs / nrev / li / i: per row, the same in all of its lanes.  (One thread per row with its insertion sort on
// the global scratch arrays was 7 of the 11 ms of the assembly at 1 M rows.)
template <int W>
__device__ __forceinline__ void write_row_union(const int lane, const uint32_t li, const uint32_t i, const uint32_t len, const uint64_t rs,
                                                const uint32_t nrev, const uint32_t *__restrict__ adj_j, const double *__restrict__ adj_w,
                                                const uint32_t K, const uint32_t *__restrict__ rev_src, const double *__restrict__ rev_w,
                                                const uint64_t *__restrict__ eptr, uint64_t *__restrict__ indptr,
                                                uint64_t *__restrict__ indices, double *__restrict__ data) {
  const int l = lane & (W - 1), first = lane & ~(W - 1);  // lane within the row, the row's first lane
  uint32_t col = 0xFFFFFFFFu;  // invalid entries sort to the end
  double w = 0.0;
  if ((uint32_t)l < len) {
    col = adj_j[(uint64_t)i * K + l];
    w = adj_w[(uint64_t)i * K + l];
  } else if ((uint32_t)l < len + nrev) {
    col = rev_src[rs + l - len];
    w = rev_w[rs + l - len];
  }
  // in-edges whose source is already an out-neighbour are dropped (the out-edge carries the pair)
  const uint32_t len_max = W == 64 ? len : max((uint32_t)__builtin_amdgcn_readlane((int)len, 0), (uint32_t)__builtin_amdgcn_readlane((int)len, 32));
  bool dup = false;
  for (uint32_t s = 0; s < len_max; ++s) {
    const uint32_t a = __shfl(col, first + (int)s, 64);
    dup = dup || (s < len && (uint32_t)l >= len && a == col);
  }
  if (dup) col = 0xFFFFFFFFu;
  // bitonic sort of the W (col, w) pairs, ascending by column (columns of valid entries are distinct)
#pragma unroll
  for (int k = 2; k <= W; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const uint32_t oc = __shfl_xor(col, j, 64);
      const double ow = __shfl_xor(w, j, 64);
      const bool up = (l & k) == 0;          // this lane's pair sorts ascending
      const bool lower = (l & j) == 0;       // this lane keeps the smaller one when ascending
      const bool take_min = up == lower;
      const bool swap = take_min ? (oc < col) : (oc > col);
      col = swap ? oc : col;
      w = swap ? ow : w;
    }
  }
  const unsigned long long row_lanes = W == 64 ? ~0ull : 0xFFFFFFFFull;
  const unsigned long long valid = (__ballot(col != 0xFFFFFFFFu) >> first) & row_lanes;
  const uint32_t L = (uint32_t)__builtin_popcountll(valid);  // = eptr[i + 1] - eptr[i]
  const uint32_t L_max = W == 64 ? L : max((uint32_t)__builtin_amdgcn_readlane((int)L, 0), (uint32_t)__builtin_amdgcn_readlane((int)L, 32));
  double degree = 0.0;
  for (uint32_t s = 0; s < L_max; ++s) {
    const double t = __shfl(w, first + (int)s, 64);
    if (s < L) degree = __dadd_rn(degree, t);
  }
  const uint64_t o = eptr[li] + li;
  const uint32_t below = (uint32_t)__builtin_popcountll((__ballot(col < i) >> first) & row_lanes);  // entries left of the diagonal
  if (l == 0) {
    indptr[li] = o;
    indices[o + below] = i;
    data[o + below] = degree;
  }
  if ((uint32_t)l < L) {
    const uint64_t at = o + l + ((uint32_t)l >= below ? 1 : 0);
    indices[at] = col;
    data[at] = -w;
  }
}

// A wave takes rows 2 w and 2 w + 1 of the range: side by side on its two halves when both unions have at most 32 entries
// before the duplicate check (nearly all rows of a k-NN graph with k = 16: write_rows was 0.98 ms of the 2.9 ms assembly with
// a wave per row), one after the other on all 64 lanes otherwise.  Rows with more than 64 entries (hubs) go to the list of
// write_rows_big_kernel.
__global__ __launch_bounds__(256) void write_rows_kernel(
    const uint32_t *__restrict__ adj_j, const double *__restrict__ adj_w, const uint32_t *__restrict__ adj_len, uint32_t n,
    uint32_t K, const uint64_t *__restrict__ rev_ptr, const uint32_t *__restrict__ rev_src, const double *__restrict__ rev_w,
    const uint64_t *__restrict__ eptr, uint32_t *__restrict__ ecol, double *__restrict__ ew, uint64_t *__restrict__ indptr,
    uint64_t *__restrict__ indices, double *__restrict__ data, uint32_t *__restrict__ big_rows,
    uint32_t *__restrict__ big_count, uint32_t r0) {
  // n = rows of the range [r0, r0 + n): li = position in it (eptr, rev_ptr, indptr), i = global row (adj_*, the diagonal)
  const int lane = threadIdx.x & 63;
  const uint32_t liA = 2u * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)), liB = liA + 1;
  if (liA > n) return;
  if ((liA == n || liB == n) && lane == 0) indptr[n] = eptr[n] + n;
  if (liA == n) return;
  const bool haveB = liB < n;
  const uint32_t lenA = adj_len[r0 + liA], lenB = haveB ? adj_len[r0 + liB] : 0u;
  const uint64_t rsA = rev_ptr[liA], reA = rev_ptr[liA + 1], rsB = reA, reB = haveB ? rev_ptr[liB + 1] : reA;
  const uint64_t cntA = lenA + (reA - rsA), cntB = lenB + (reB - rsB);
  // hub rows (in-degree in the hundreds or thousands): one workgroup each, below
  if (cntA > 64 && lane == 0) big_rows[atomicAdd(big_count, 1u)] = liA;
  if (haveB && cntB > 64 && lane == 0) big_rows[atomicAdd(big_count, 1u)] = liB;
  if (haveB && cntA <= 32 && cntB <= 32) {
    const bool hb = lane >= 32;
    write_row_union<32>(lane, hb ? liB : liA, r0 + (hb ? liB : liA), hb ? lenB : lenA, hb ? rsB : rsA, (uint32_t)(hb ? reB - rsB : reA - rsA), adj_j,
                        adj_w, K, rev_src, rev_w, eptr, indptr, indices, data);
  } else {
    if (cntA <= 64) write_row_union<64>(lane, liA, r0 + liA, lenA, rsA, (uint32_t)(reA - rsA), adj_j, adj_w, K, rev_src, rev_w, eptr, indptr, indices, data);
    if (haveB && cntB <= 64)
      write_row_union<64>(lane, liB, r0 + liB, lenB, rsB, (uint32_t)(reB - rsB), adj_j, adj_w, K, rev_src, rev_w, eptr, indptr, indices, data);
  }
}

// Hub rows: one workgroup per row, the union gathered into LDS (up to kBigMax entries before the duplicate check),
// ordered by a bitonic network over LDS, degree summed sequentially by one thread, written side by side.  Longer rows
// still take the serial form.  (k-NN graphs have hubs: at 1 M x 128, 9 % of the rows have more than 47 in-edges, the
// largest 1 691; left to one thread each with a heap sort in global memory they were most of this stage's time.)
constexpr uint32_t kBigMax = 4096;
__global__ __launch_bounds__(256) void write_rows_big_kernel(
    const uint32_t *__restrict__ big_rows, const uint32_t *__restrict__ big_count, const uint32_t *__restrict__ adj_j,
    const double *__restrict__ adj_w, const uint32_t *__restrict__ adj_len, uint32_t K, const uint64_t *__restrict__ rev_ptr,
    const uint32_t *__restrict__ rev_src, const double *__restrict__ rev_w, const uint64_t *__restrict__ eptr,
    uint32_t *__restrict__ ecol, double *__restrict__ ew, uint64_t *__restrict__ indptr, uint64_t *__restrict__ indices,
    double *__restrict__ data, uint32_t r0) {
  __shared__ uint32_t s_col[kBigMax];
  __shared__ double s_w[kBigMax];
  __shared__ double s_degree;
  __shared__ uint32_t s_below;
  const uint32_t tid = threadIdx.x;
  for (uint32_t b = blockIdx.x; b < *big_count; b += gridDim.x) {
    const uint32_t li = big_rows[b], i = r0 + li;
    const uint32_t len = adj_len[i];
    const uint64_t rs = rev_ptr[li];
    const uint32_t nrev = (uint32_t)(rev_ptr[li + 1] - rs), total = len + nrev;
    if (total > kBigMax) {
      if (tid == 0) write_row_serial(i, li, adj_j, adj_w, adj_len, K, rev_ptr, rev_src, rev_w, eptr, ecol, ew, indptr, indices, data);
      continue;
    }
    uint32_t P = 64;
    while (P < total) P <<= 1;
    __syncthreads();  // the row before is done with the arrays
    if (tid == 0) s_below = 0;
    for (uint32_t e = tid; e < P; e += blockDim.x) {
      uint32_t c = 0xFFFFFFFFu;
      double v = 0.0;
      if (e < len) {
        c = adj_j[(uint64_t)i * K + e];
        v = adj_w[(uint64_t)i * K + e];
      } else if (e < total) {
        c = rev_src[rs + e - len];
        v = rev_w[rs + e - len];
        if (in_adj(adj_j + (uint64_t)i * K, len, c)) c = 0xFFFFFFFFu;  // the out-edge carries the pair
      }
      s_col[e] = c;
      s_w[e] = v;
    }
    __syncthreads();
    for (uint32_t k = 2; k <= P; k <<= 1)
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
        for (uint32_t e = tid; e < P; e += blockDim.x) {
          const uint32_t partner = e ^ j;
          if (partner > e) {
            const bool up = (e & k) == 0;
            const uint32_t a = s_col[e], c2 = s_col[partner];
            if (up ? (a > c2) : (a < c2)) {
              s_col[e] = c2;
              s_col[partner] = a;
              const double t = s_w[e];
              s_w[e] = s_w[partner];
              s_w[partner] = t;
            }
          }
        }
        __syncthreads();
      }
    const uint32_t L = (uint32_t)(eptr[li + 1] - eptr[li]);  // valid entries (the invalid ones sorted behind them)
    uint32_t below = 0;
    for (uint32_t e = tid; e < L; e += blockDim.x) below += s_col[e] < i ? 1u : 0u;
    if (below) atomicAdd(&s_below, below);
    if (tid == 0) {
      double degree = 0.0;  // sequential sum in column order (src/laplacian.rs:369)
      for (uint32_t e = 0; e < L; ++e) degree = __dadd_rn(degree, s_w[e]);
      s_degree = degree;
    }
    __syncthreads();
    const uint64_t o = eptr[li] + li;
    const uint32_t nb = s_below;
    if (tid == 0) {
      indptr[li] = o;
      indices[o + nb] = i;
      data[o + nb] = s_degree;
    }
    for (uint32_t e = tid; e < L; e += blockDim.x) {
      const uint64_t at = o + e + (e >= nb ? 1 : 0);
      indices[at] = s_col[e];
      data[at] = -s_w[e];
    }
  }
}

// ---- K5, round-4 form: the union ordered by RANK COUNTING ----------------------------------------------------------------------
// W lanes per row as above, but instead of a bitonic network over the lanes (15 stages of three shuffles at W = 32) every lane
// counts the entries with a smaller column -- the rows of a k-NN graph hold ~14 entries, so that is ~14 shuffles --, the entries
// move to their rank with three ds_permute, and everything behind (degree in column order, diagonal, the writes) is as before.
template <int W>
__device__ __forceinline__ void row_entry_load(const int lane, const uint32_t i, const uint32_t len, const uint64_t rs, const uint32_t nrev,
                                               const uint32_t *__restrict__ adj_j, const double *__restrict__ adj_w, const uint32_t K,
                                               const RevEdge *__restrict__ rev, uint32_t &col, double &w) {
  const int l = lane & (W - 1);
  col = 0xFFFFFFFFu;  // invalid entries sort to the end
  w = 0.0;
  if ((uint32_t)l < len) {
    col = adj_j[(uint64_t)i * K + l];
    w = adj_w[(uint64_t)i * K + l];
  } else if ((uint32_t)l < len + nrev) {
    const RevEdge e = rev[rs + l - len];
    col = e.src;
    w = e.w;
  }
}
// (o: where the row starts in the CSR = eptr[li] + li)
template <int W>
__device__ __forceinline__ void row_finish_ranked(const int lane, const uint32_t li, const uint32_t i, const uint32_t len, const uint32_t nrev,
                                                  const uint64_t o, uint32_t col, double w, uint64_t *__restrict__ indptr,
                                                  uint64_t *__restrict__ indices, double *__restrict__ data) {
  const int l = lane & (W - 1), first = lane & ~(W - 1);
  const uint32_t len_max = W == 64 ? len : max((uint32_t)__builtin_amdgcn_readlane((int)len, 0), (uint32_t)__builtin_amdgcn_readlane((int)len, 32));
  bool dup = false;  // in-edges whose source is already an out-neighbour are dropped (the out-edge carries the pair)
  for (uint32_t s = 0; s < len_max; ++s) {
    const uint32_t a = __shfl(col, first + (int)s, 64);
    dup = dup || (s < len && (uint32_t)l >= len && a == col);
  }
  if (dup) col = 0xFFFFFFFFu;
  const uint32_t tot = len + nrev;
  const uint32_t tot_max = W == 64 ? tot : max((uint32_t)__builtin_amdgcn_readlane((int)tot, 0), (uint32_t)__builtin_amdgcn_readlane((int)tot, 32));
  uint32_t rank = 0;  // valid entries with a smaller column (the columns of valid entries are distinct)
  for (uint32_t s = 0; s < tot_max; ++s) {
    const uint32_t a = __shfl(col, first + (int)s, 64);
    rank += (s < tot && a < col) ? 1u : 0u;
  }
  const unsigned long long row_lanes = W == 64 ? ~0ull : 0xFFFFFFFFull;
  const bool valid = col != 0xFFFFFFFFu;
  const unsigned long long vm = (__ballot(valid) >> first) & row_lanes;
  const uint32_t L = (uint32_t)__builtin_popcountll(vm);  // = eptr[li + 1] - eptr[li]
  // the invalid lanes take the places behind the valid ones, in lane order: the ranks are a permutation of the row's lanes
  if (!valid) rank = L + (uint32_t)__builtin_popcountll(~vm & row_lanes & ((1ull << l) - 1ull));
  const int dest = (first + (int)rank) * 4;
  col = (uint32_t)__builtin_amdgcn_ds_permute(dest, (int)col);
  {
    const int lo = __builtin_amdgcn_ds_permute(dest, __double2loint(w)), hi = __builtin_amdgcn_ds_permute(dest, __double2hiint(w));
    w = __hiloint2double(hi, lo);
  }
  const uint32_t L_max = W == 64 ? L : max((uint32_t)__builtin_amdgcn_readlane((int)L, 0), (uint32_t)__builtin_amdgcn_readlane((int)L, 32));
  double degree = 0.0;  // sequential sum in column order (src/laplacian.rs:369)
  for (uint32_t s = 0; s < L_max; ++s) {
    const double t = __shfl(w, first + (int)s, 64);
    if (s < L) degree = __dadd_rn(degree, t);
  }
  const uint32_t below = (uint32_t)__builtin_popcountll((__ballot(col < i) >> first) & row_lanes);  // entries left of the diagonal
  if (l == 0) {
    indptr[li] = o;
    indices[o + below] = i;
    data[o + below] = degree;
  }
  if ((uint32_t)l < L) {
    const uint64_t at = o + l + ((uint32_t)l >= below ? 1 : 0);
    indices[at] = col;
    data[at] = -w;
  }
}

constexpr uint32_t kMidMax = 512, kMidPer = kMidMax / 64;  // rows a wave orders by itself (write_rows_mid_kernel)
// A wave takes R pairs of rows (2 w R .. 2 w R + 2 R - 1), their loads issued round by round for all of them (a pair alone: two
// dependent round trips and ~14 shuffles of work -- at 1 M rows the 32 waves of a CU mostly waited: 0.46 ms).  A pair sits side
// by side on the wave's halves when both unions have at most 32 entries before the duplicate check (nearly all rows of a k-NN
// graph with k = 16); otherwise its rows take all 64 lanes one after the other; rows of more than 64 entries go to the lists of
// the two kernels below.
template <int R>
__global__ __launch_bounds__(256) void write_rows_ranked_kernel(
    const uint32_t *__restrict__ adj_j, const double *__restrict__ adj_w, const uint32_t *__restrict__ adj_len, uint32_t n,
    uint32_t K, const uint64_t *__restrict__ rev_ptr, const RevEdge *__restrict__ rev, const uint64_t *__restrict__ eptr,
    uint64_t *__restrict__ indptr, uint64_t *__restrict__ indices, double *__restrict__ data, uint32_t *__restrict__ mid_rows,
    uint32_t *__restrict__ big_rows, uint32_t *__restrict__ counts, uint32_t r0) {
  // n = rows of the range [r0, r0 + n): li = position in it (eptr, rev_ptr, indptr), i = global row (adj_*, the diagonal)
  const int lane = threadIdx.x & 63;
  const bool hb = lane >= 32;
  if (blockIdx.x == 0 && threadIdx.x == 0) indptr[n] = eptr[n] + n;
  if (n == 0) return;
  const uint32_t p0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * R;
  uint32_t liA[R], lenA[R], lenB[R];
  uint64_t rp0[R], rp1[R], rp2[R], eA[R], eB[R];
  bool haveA[R], haveB[R], fast[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    liA[r] = 2u * (p0 + r);
    haveA[r] = liA[r] < n;
    haveB[r] = liA[r] + 1 < n;
    const uint32_t a = haveA[r] ? liA[r] : 0u, b = haveB[r] ? liA[r] + 1 : a;
    lenA[r] = adj_len[r0 + a];
    lenB[r] = adj_len[r0 + b];
    rp0[r] = rev_ptr[a];
    rp1[r] = rev_ptr[a + 1];
    rp2[r] = rev_ptr[b + 1];
    eA[r] = eptr[a];
    eB[r] = eptr[b];
  }
  uint32_t col[R];
  double w[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint64_t cntA = lenA[r] + (rp1[r] - rp0[r]), cntB = lenB[r] + (rp2[r] - rp1[r]);
    if (lane == 0) {
      if (haveA[r] && cntA > 64) {
        if (cntA <= kMidMax) mid_rows[atomicAdd(counts, 1u)] = liA[r];
        else big_rows[atomicAdd(counts + 1, 1u)] = liA[r];
      }
      if (haveB[r] && cntB > 64) {
        if (cntB <= kMidMax) mid_rows[atomicAdd(counts, 1u)] = liA[r] + 1;
        else big_rows[atomicAdd(counts + 1, 1u)] = liA[r] + 1;
      }
    }
    fast[r] = haveB[r] && cntA <= 32 && cntB <= 32;
    col[r] = 0xFFFFFFFFu;
    w[r] = 0.0;
    if (fast[r])
      row_entry_load<32>(lane, r0 + liA[r] + (hb ? 1u : 0u), hb ? lenB[r] : lenA[r], hb ? rp1[r] : rp0[r],
                         (uint32_t)(hb ? rp2[r] - rp1[r] : rp1[r] - rp0[r]), adj_j, adj_w, K, rev, col[r], w[r]);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (!haveA[r]) continue;
    if (fast[r]) {
      const uint32_t li = liA[r] + (hb ? 1u : 0u);
      row_finish_ranked<32>(lane, li, r0 + li, hb ? lenB[r] : lenA[r], (uint32_t)(hb ? rp2[r] - rp1[r] : rp1[r] - rp0[r]), (hb ? eB[r] : eA[r]) + li,
                            col[r], w[r], indptr, indices, data);
    } else {
      const uint64_t cntA = lenA[r] + (rp1[r] - rp0[r]), cntB = lenB[r] + (rp2[r] - rp1[r]);
      if (cntA <= 64) {
        uint32_t c;
        double v;
        row_entry_load<64>(lane, r0 + liA[r], lenA[r], rp0[r], (uint32_t)(rp1[r] - rp0[r]), adj_j, adj_w, K, rev, c, v);
        row_finish_ranked<64>(lane, liA[r], r0 + liA[r], lenA[r], (uint32_t)(rp1[r] - rp0[r]), eA[r] + liA[r], c, v, indptr, indices, data);
      }
      if (haveB[r] && cntB <= 64) {
        uint32_t c;
        double v;
        row_entry_load<64>(lane, r0 + liA[r] + 1, lenB[r], rp1[r], (uint32_t)(rp2[r] - rp1[r]), adj_j, adj_w, K, rev, c, v);
        row_finish_ranked<64>(lane, liA[r] + 1, r0 + liA[r] + 1, lenB[r], (uint32_t)(rp2[r] - rp1[r]), eB[r] + liA[r] + 1, c, v, indptr, indices, data);
      }
    }
  }
}

// Hub rows of 65 .. kMidMax entries: one WAVE per row (four per workgroup), the union in the wave's LDS, ordered by rank counting
// like the short rows: a lane holds up to kMidMax / 64 entries in registers, counts for each the entries with a smaller column
// (broadcast reads of the LDS copy), and the entries go back to LDS at their ranks; the degree is the sequential sum in column
// order (src/laplacian.rs:369) over values passed from lane to lane (a workgroup per row with a bitonic network and a barrier per
// stage was 0.42 ms at 1 M rows; a wave per row with the same network 0.36 ms: ~30 dependent passes over LDS per row).
__global__ __launch_bounds__(256) void write_rows_mid_kernel(
    const uint32_t *__restrict__ rows_list, const uint32_t *__restrict__ rows_count, const uint32_t *__restrict__ adj_j,
    const double *__restrict__ adj_w, const uint32_t *__restrict__ adj_len, uint32_t K, const uint64_t *__restrict__ rev_ptr,
    const RevEdge *__restrict__ rev, const uint64_t *__restrict__ eptr, uint64_t *__restrict__ indptr, uint64_t *__restrict__ indices,
    double *__restrict__ data, uint32_t r0) {
  __shared__ uint32_t s_col_all[4 * kMidMax];
  __shared__ double s_w_all[4 * kMidMax];
  const uint32_t unit = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  uint32_t *s_col = s_col_all + unit * kMidMax;
  double *s_w = s_w_all + unit * kMidMax;
  const uint32_t count = *rows_count;
  for (uint32_t b = blockIdx.x * 4 + unit; b < count; b += gridDim.x * 4) {
    const uint32_t li = rows_list[b], i = r0 + li;
    const uint32_t len = adj_len[i];
    const uint64_t rs = rev_ptr[li], o = eptr[li] + li;
    const uint32_t nrev = (uint32_t)(rev_ptr[li + 1] - rs), total = len + nrev;  // 64 < total <= kMidMax
    const uint32_t mine = lane < len ? adj_j[(uint64_t)i * K + lane] : 0xFFFFFFFFu;  // (K <= 32: the out-neighbours sit one per lane)
    uint32_t col[kMidPer], rank[kMidPer];
    double w[kMidPer];
    lanes_sync();  // the row before is done with the arrays
#pragma unroll
    for (uint32_t m = 0; m < kMidPer; ++m) {
      const uint32_t e = m * 64 + lane;
      col[m] = 0xFFFFFFFFu;
      w[m] = 0.0;
      if (e < len) {
        col[m] = adj_j[(uint64_t)i * K + e];
        w[m] = adj_w[(uint64_t)i * K + e];
      } else if (e < total) {
        const RevEdge r = rev[rs + e - len];
        col[m] = r.src;
        w[m] = r.w;
      }
    }
    // in-edges whose source is an out-neighbour are dropped (the out-edge carries the pair)
    for (uint32_t q = 0; q < len; ++q) {
      const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)q);
#pragma unroll
      for (uint32_t m = 0; m < kMidPer; ++m)
        if (m * 64 + lane >= len && col[m] == a) col[m] = 0xFFFFFFFFu;
    }
#pragma unroll
    for (uint32_t m = 0; m < kMidPer; ++m) {
      if (m * 64 < total) s_col[m * 64 + lane] = col[m];
      rank[m] = 0;
    }
    lanes_sync();
    for (uint32_t s_ = 0; s_ < total; ++s_) {
      const uint32_t a = s_col[s_];  // (one address for the wave: a broadcast)
#pragma unroll
      for (uint32_t m = 0; m < kMidPer; ++m) rank[m] += a < col[m] ? 1u : 0u;
    }
    lanes_sync();  // every lane has read the unordered copy
    uint32_t below = 0;
#pragma unroll
    for (uint32_t m = 0; m < kMidPer; ++m)
      if (col[m] != 0xFFFFFFFFu) {  // (valid columns are distinct: the ranks of the valid entries are 0 .. L - 1)
        s_col[rank[m]] = col[m];
        s_w[rank[m]] = w[m];
        below += col[m] < i ? 1u : 0u;
      }
    for (int o_ = 32; o_ > 0; o_ >>= 1) below += __shfl_xor(below, o_, 64);
    lanes_sync();
    const uint32_t L = (uint32_t)(eptr[li + 1] - eptr[li]);  // valid entries
    double ws[kMidPer];
#pragma unroll
    for (uint32_t m = 0; m < kMidPer; ++m) ws[m] = m * 64 + lane < L ? s_w[m * 64 + lane] : 0.0;
    double degree = 0.0;
#pragma unroll
    for (uint32_t m = 0; m < kMidPer; ++m) {
      if (m * 64 >= L) break;
      const uint32_t upto = min(64u, L - m * 64);
      for (uint32_t t = 0; t < upto; ++t) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(ws[m]), (int)t), hi = __builtin_amdgcn_readlane(__double2hiint(ws[m]), (int)t);
        degree = __dadd_rn(degree, __hiloint2double(hi, lo));
      }
    }
    if (lane == 0) {
      indptr[li] = o;
      indices[o + below] = i;
      data[o + below] = degree;
    }
#pragma unroll
    for (uint32_t m = 0; m < kMidPer; ++m) {
      const uint32_t e = m * 64 + lane;
      if (e < L) {
        const uint64_t at = o + e + (e >= below ? 1 : 0);
        indices[at] = s_col[e];
        data[at] = -ws[m];
      }
    }
  }
}

// Rows of kMidMax + 1 .. kBigMax entries: one workgroup per row, a bitonic network over LDS with a barrier per stage.
__global__ __launch_bounds__(256) void write_rows_hub_kernel(
    const uint32_t *__restrict__ rows_list, const uint32_t *__restrict__ rows_count, const uint32_t *__restrict__ adj_j,
    const double *__restrict__ adj_w, const uint32_t *__restrict__ adj_len, uint32_t K, const uint64_t *__restrict__ rev_ptr,
    const RevEdge *__restrict__ rev, const uint64_t *__restrict__ eptr, uint64_t *__restrict__ indptr, uint64_t *__restrict__ indices,
    double *__restrict__ data, uint32_t r0) {
  __shared__ uint32_t s_col[kBigMax];
  __shared__ double s_w[kBigMax];
  __shared__ double s_degree;
  __shared__ uint32_t s_below;
  const uint32_t tid = threadIdx.x;
  const uint32_t count = *rows_count;
  for (uint32_t b = blockIdx.x; b < count; b += gridDim.x) {
    const uint32_t li = rows_list[b], i = r0 + li;
    const uint32_t len = adj_len[i];
    const uint64_t rs = rev_ptr[li];
    const uint32_t nrev = (uint32_t)(rev_ptr[li + 1] - rs), total = len + nrev;
    if (total > kBigMax) continue;  // (write_rows_serial_kernel)
    uint32_t P = 64;
    while (P < total) P <<= 1;
    __syncthreads();  // the row before is done with the arrays
    if (tid == 0) s_below = 0;
    for (uint32_t e = tid; e < P; e += blockDim.x) {
      uint32_t c = 0xFFFFFFFFu;
      double v = 0.0;
      if (e < len) {
        c = adj_j[(uint64_t)i * K + e];
        v = adj_w[(uint64_t)i * K + e];
      } else if (e < total) {
        const RevEdge r = rev[rs + e - len];
        c = r.src;
        v = r.w;
        if (in_adj(adj_j + (uint64_t)i * K, len, c)) c = 0xFFFFFFFFu;  // the out-edge carries the pair
      }
      s_col[e] = c;
      s_w[e] = v;
    }
    __syncthreads();
    for (uint32_t k = 2; k <= P; k <<= 1)
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
        for (uint32_t e = tid; e < P; e += blockDim.x) {
          const uint32_t partner = e ^ j;
          if (partner > e) {
            const bool up = (e & k) == 0;
            const uint32_t a = s_col[e], c2 = s_col[partner];
            if (up ? (a > c2) : (a < c2)) {
              s_col[e] = c2;
              s_col[partner] = a;
              const double t = s_w[e];
              s_w[e] = s_w[partner];
              s_w[partner] = t;
            }
          }
        }
        __syncthreads();
      }
    const uint32_t L = (uint32_t)(eptr[li + 1] - eptr[li]);  // valid entries (the invalid ones sorted behind them)
    uint32_t below = 0;
    for (uint32_t e = tid; e < L; e += blockDim.x) below += s_col[e] < i ? 1u : 0u;
    if (below) atomicAdd(&s_below, below);
    if (tid == 0) {
      double degree = 0.0;  // sequential sum in column order (src/laplacian.rs:369)
      for (uint32_t e = 0; e < L; ++e) degree = __dadd_rn(degree, s_w[e]);
      s_degree = degree;
    }
    __syncthreads();
    const uint64_t o = eptr[li] + li;
    const uint32_t nb = s_below;
    if (tid == 0) {
      indptr[li] = o;
      indices[o + nb] = i;
      data[o + nb] = s_degree;
    }
    for (uint32_t e = tid; e < L; e += blockDim.x) {
      const uint64_t at = o + e + (e >= nb ? 1 : 0);
      indices[at] = s_col[e];
      data[at] = -s_w[e];
    }
  }
}

// rows past kBigMax entries: the serial form on the packed reverse edges (one thread per row; ecol / ew: scratch of the union's size)
__global__ void write_rows_serial_kernel(const uint32_t *__restrict__ rows_list, const uint32_t *__restrict__ rows_count, const uint32_t *__restrict__ adj_j,
                                         const double *__restrict__ adj_w, const uint32_t *__restrict__ adj_len, uint32_t K,
                                         const uint64_t *__restrict__ rev_ptr, const RevEdge *__restrict__ rev, const uint64_t *__restrict__ eptr,
                                         uint32_t *__restrict__ ecol, double *__restrict__ ew, uint64_t *__restrict__ indptr,
                                         uint64_t *__restrict__ indices, double *__restrict__ data, uint32_t r0) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= *rows_count) return;
  const uint32_t li = rows_list[b], i = r0 + li;
  const uint32_t len = adj_len[i];
  if (len + (rev_ptr[li + 1] - rev_ptr[li]) <= kBigMax) return;
  const uint32_t *aj = adj_j + (uint64_t)i * K;
  const double *aw = adj_w + (uint64_t)i * K;
  uint32_t *col = ecol + eptr[li];
  double *w = ew + eptr[li];
  uint32_t L = 0;
  for (uint32_t s = 0; s < len; ++s) {
    col[L] = aj[s];
    w[L++] = aw[s];
  }
  for (uint64_t e = rev_ptr[li]; e < rev_ptr[li + 1]; ++e) {
    const uint32_t src = rev[e].src;
    if (!in_adj(aj, len, src)) {
      col[L] = src;
      w[L++] = rev[e].w;
    }
  }
  sort_row(col, w, L);
  double degree = 0.0;
  for (uint32_t s = 0; s < L; ++s) degree = __dadd_rn(degree, w[s]);
  uint64_t o = eptr[li] + li;
  indptr[li] = o;
  bool diag_done = false;
  for (uint32_t s = 0; s < L; ++s) {
    if (!diag_done && col[s] > i) {
      indices[o] = i;
      data[o++] = degree;
      diag_done = true;
    }
    indices[o] = col[s];
    data[o++] = -w[s];
  }
  if (!diag_done) {
    indices[o] = i;
    data[o] = degree;
  }
}

// Rows [row_begin, row_begin + row_count) of the Laplacian of the n-node graph whose complete neighbour table is given
// (row-sharded builds: every rank holds the all-gathered table and assembles its own row block; the whole matrix is the
// range [0, n)).  Weights and the pruning need the degrees of ALL rows (two passes over the table, replicated: they are
// the cheap part); the reverse-edge scatter, the unions, the sorts and the CSR rows are done for the range only.
// The CSR that comes out has row_count rows, GLOBAL column ids, indptr starting at 0.
// given != nullptr: the ranked and pruned adjacency of ALL nodes is there already (the row-sharded build ranks each rank's rows on
// their owner and exchanges the result: graph_adjacency_of_rows below); the table arguments are not read.
static int assemble_rows(as_ctx *ctx, const uint32_t *nbr_idx, const double *nbr_dist, const GraphAdjacency *given, uint64_t n64,
                         uint32_t K, const as_graph_params *gp, as_csr **out, uint64_t row_begin, uint64_t row_count) {
  if (K > kMaxK) return fail(ctx, AS_ERR_UNSUPPORTED, "topk + 1 must be <= 256");
  const uint32_t n = (uint32_t)n64;
  if (row_begin + row_count > n64) return fail(ctx, AS_ERR_ARG, "row range out of bounds");
  const uint32_t r0 = (uint32_t)row_begin, nr = (uint32_t)row_count;
  const double sigma = gp->has_sigma ? gp->sigma : 1.0;  // src/laplacian.rs:254
  const unsigned gb = cdiv(n, 256), gr = std::max(1u, cdiv(nr, 256));
  hipStream_t st = ctx->stream;

  DevBuf<uint32_t> deg, adj_j, adj_len, cnt_in, cursor, rev_src, ucnt, ecol, adj_slot;
  DevBuf<double> adj_w, rev_w, ew;
  DevBuf<RevEdge> rev;
  DevBuf<unsigned long long> deg_sum;
  DevBuf<uint64_t> rev_ptr, eptr;
  // K <= 32: degrees 64 rows per wave, the reverse edges placed by the counting atomic's return value into one array of 16-byte
  // records sized by its bound n K (no read-back of their number), unions ordered by rank counting, hub rows a wave each.
  // K > 32 (or "assembly_lanes" 0, the independent form the tests compare with): one thread per row.
  const bool ranked = K <= 32 && ctx->opt_assembly_lanes != 0;
  if (given && !ranked) return fail(ctx, AS_ERR_ARG, "graph assembly: a given adjacency needs the ranked form");
  if (!given) {
    AS_TRY(deg.alloc(ctx, n));
    AS_TRY(adj_j.alloc(ctx, (size_t)n * K));
    AS_TRY(adj_w.alloc(ctx, (size_t)n * K));
    AS_TRY(adj_len.alloc(ctx, n));
  }
  AS_TRY(deg_sum.alloc(ctx, 1));
  const uint32_t *aj = given ? given->adj_j : adj_j.p, *alen = given ? given->adj_len : adj_len.p;
  const double *aw = given ? given->adj_w : adj_w.p;
  AS_TRY(cnt_in.alloc(ctx, (size_t)nr + 1));
  if (!ranked) AS_TRY(cursor.alloc(ctx, (size_t)nr + 1));
  AS_TRY(rev_ptr.alloc(ctx, (size_t)nr + 1));
  AS_TRY(ucnt.alloc(ctx, (size_t)nr + 1));
  AS_TRY(eptr.alloc(ctx, (size_t)nr + 1));
  if (ranked) {
    AS_TRY(adj_slot.alloc(ctx, (size_t)n * K));
    AS_TRY(rev.alloc(ctx, (size_t)n * K));  // (every kept edge could point into the range)
  }
  AS_HIP(ctx, hipMemsetAsync(deg_sum.p, 0, sizeof(unsigned long long), st));
  AS_HIP(ctx, hipMemsetAsync(cnt_in.p, 0, sizeof(uint32_t) * ((size_t)nr + 1), st));
  if (!ranked) AS_HIP(ctx, hipMemsetAsync(cursor.p, 0, sizeof(uint32_t) * ((size_t)nr + 1), st));

  uint64_t n_edges_dir = 0, n_union = 0;
  {
    ProfScope ps(ctx, "graph_assembly");
    const unsigned gb32 = (unsigned)cdiv((uint64_t)n * 32, (uint64_t)256), gr32 = (unsigned)std::max<uint64_t>(1, cdiv((uint64_t)nr * 32, (uint64_t)256));
    if (given) {
      hipLaunchKernelGGL(count_in_edges32_kernel, dim3(gb32), dim3(256), 0, st, aj, alen, n, K, r0, nr, cnt_in.p, adj_slot.p);
    } else if (ranked) {
      hipLaunchKernelGGL(degree_rows64_kernel, dim3((unsigned)std::min<uint64_t>(cdiv((uint64_t)n, (uint64_t)256), 256 * 8)), dim3(256), 0, st, nbr_idx,
                         nbr_dist, n, K, gp->eps, deg.p, deg_sum.p);
      hipLaunchKernelGGL(adjacency32_kernel<1>, dim3(gb32), dim3(256), 0, st, nbr_idx, nbr_dist, n, K, gp->eps, sigma, gp->p, deg.p, deg_sum.p,
                         adj_j.p, adj_w.p, adj_len.p, cnt_in.p, r0, nr, adj_slot.p);
    } else {
      hipLaunchKernelGGL(degree_kernel, dim3(gb), dim3(256), 0, st, nbr_idx, nbr_dist, n, K, gp->eps, deg.p, deg_sum.p);
    }
    if (ranked) {
    } else if (K <= 64)
      hipLaunchKernelGGL(adjacency_kernel<64>, dim3(gb), dim3(256), 0, st, nbr_idx, nbr_dist, n, K, gp->eps, sigma,
                         gp->p, deg.p, deg_sum.p, adj_j.p, adj_w.p, adj_len.p, cnt_in.p, r0, nr);
    else
      hipLaunchKernelGGL(adjacency_kernel<kMaxK>, dim3(gb), dim3(256), 0, st, nbr_idx, nbr_dist, n, K, gp->eps, sigma,
                         gp->p, deg.p, deg_sum.p, adj_j.p, adj_w.p, adj_len.p, cnt_in.p, r0, nr);
    AS_TRY(exclusive_scan_u32_u64(ctx, cnt_in.p, nr, rev_ptr.p));
    if (ranked) {
      hipLaunchKernelGGL(fill_reverse32_kernel<2>, dim3((unsigned)cdiv((uint64_t)cdiv((uint64_t)n, (uint64_t)2) * 32, (uint64_t)256)), dim3(256), 0, st, aj, aw, adj_slot.p, alen, n, K, rev_ptr.p, rev.p, r0, nr);
      hipLaunchKernelGGL(union_count32r_kernel, dim3(gr32), dim3(256), 0, st, aj, alen, nr, K, rev_ptr.p, rev.p, ucnt.p, r0);
    } else {
      AS_HIP(ctx, copy_d2h(ctx, &n_edges_dir, rev_ptr.p + nr, sizeof(uint64_t)));
      AS_HIP(ctx, hipStreamSynchronize(st));
      AS_TRY(rev_src.alloc(ctx, n_edges_dir));
      AS_TRY(rev_w.alloc(ctx, n_edges_dir));
      hipLaunchKernelGGL(fill_reverse_kernel, dim3(gb), dim3(256), 0, st, adj_j.p, adj_w.p, adj_len.p, n, K,
                         rev_ptr.p, cursor.p, rev_src.p, rev_w.p, r0, nr);
      hipLaunchKernelGGL(union_count_kernel, dim3(gr), dim3(256), 0, st, adj_j.p, adj_len.p, nr, K, rev_ptr.p,
                           rev_src.p, ucnt.p, r0);
    }
    AS_TRY(exclusive_scan_u32_u64(ctx, ucnt.p, nr, eptr.p));
    if (ranked) {  // (the directed count is a statistic here: both words behind ONE wait for the stream)
      AS_HIP(ctx, ensure_pinned(ctx));
      uint64_t *two = reinterpret_cast<uint64_t *>(ctx->pinned);
      AS_HIP(ctx, hipMemcpyAsync(two, rev_ptr.p + nr, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
      AS_HIP(ctx, hipMemcpyAsync(two + 1, eptr.p + nr, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
      AS_HIP(ctx, hipStreamSynchronize(st));
      n_edges_dir = two[0];
      n_union = two[1];
    } else {
      AS_HIP(ctx, copy_d2h(ctx, &n_union, eptr.p + nr, sizeof(uint64_t)));
    }
  }
  const uint64_t nnz = n_union + nr;
  // the handle leaves through *out on success only: every early return below frees it and its slab
  struct CsrHolder {
    as_csr *p;
    ~CsrHolder() {
      if (p) as_csr_free(p);
    }
  } holder{new as_csr()};
  as_csr *m = holder.p;
  m->device = ctx->device;
  m->n = nr;
  m->n_cols = n;
  m->row_begin = r0;
  m->nnz = nnz;
  m->known_symmetric = 1;  // a union of pairs, one weight per pair (src/laplacian.rs:303-320): L_ij == L_ji bit for bit
  {  // one allocation for the three arrays (the handle outlives the call: not from the workspace cache)
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t b_ptr = up(sizeof(uint64_t) * ((size_t)nr + 1)), b_idx = up(sizeof(uint64_t) * (nnz ? nnz : 1)), b_val = up(sizeof(double) * (nnz ? nnz : 1));
    if (handle_malloc(&m->slab, b_ptr + b_idx + b_val) != hipSuccess) {
      (void)hipGetLastError();
      m->slab = nullptr;
      return fail(ctx, AS_ERR_HIP, "graph assembly: out of device memory for the CSR");
    }
    m->indptr = reinterpret_cast<uint64_t *>(m->slab);
    m->indices = reinterpret_cast<uint64_t *>(static_cast<char *>(m->slab) + b_ptr);
    m->data = reinterpret_cast<double *>(static_cast<char *>(m->slab) + b_ptr + b_idx);
  }
  AS_TRY(ecol.alloc(ctx, n_union));
  AS_TRY(ew.alloc(ctx, n_union));
  DevBuf<uint32_t> big_rows, big_count, mid_rows;
  AS_TRY(big_rows.alloc(ctx, (size_t)nr + 1));
  AS_TRY(big_count.alloc(ctx, 2));
  AS_HIP(ctx, hipMemsetAsync(big_count.p, 0, 2 * sizeof(uint32_t), st));
  if (ranked) {
    AS_TRY(mid_rows.alloc(ctx, (size_t)nr + 1));
    ProfScope ps(ctx, "graph_assembly");
    hipLaunchKernelGGL(write_rows_ranked_kernel<2>, dim3(cdiv((uint64_t)nr + 1, 16)), dim3(256), 0, st, aj, aw, alen, nr, K, rev_ptr.p, rev.p,
                       eptr.p, m->indptr, m->indices, m->data, mid_rows.p, big_rows.p, big_count.p, r0);
    hipLaunchKernelGGL(write_rows_mid_kernel, dim3(2048), dim3(256), 0, st, mid_rows.p, big_count.p, aj, aw, alen, K, rev_ptr.p, rev.p,
                       eptr.p, m->indptr, m->indices, m->data, r0);
    hipLaunchKernelGGL(write_rows_hub_kernel, dim3(512), dim3(256), 0, st, big_rows.p, big_count.p + 1, aj, aw, alen, K, rev_ptr.p,
                       rev.p, eptr.p, m->indptr, m->indices, m->data, r0);
    hipLaunchKernelGGL(write_rows_serial_kernel, dim3(cdiv((uint64_t)nr + 1, 256)), dim3(256), 0, st, big_rows.p, big_count.p + 1, aj, aw, alen,
                       K, rev_ptr.p, rev.p, eptr.p, ecol.p, ew.p, m->indptr, m->indices, m->data, r0);
  } else {
    ProfScope ps(ctx, "graph_assembly");
    hipLaunchKernelGGL(write_rows_kernel, dim3(cdiv((uint64_t)nr + 1, 8)), dim3(256), 0, st, adj_j.p, adj_w.p,
                       adj_len.p, nr, K, rev_ptr.p, rev_src.p, rev_w.p, eptr.p, ecol.p, ew.p, m->indptr,
                       m->indices, m->data, big_rows.p, big_count.p, r0);
    hipLaunchKernelGGL(write_rows_big_kernel, dim3(4096), dim3(256), 0, st, big_rows.p, big_count.p, adj_j.p, adj_w.p, adj_len.p,
                       K, rev_ptr.p, rev_src.p, rev_w.p, eptr.p, ecol.p, ew.p, m->indptr, m->indices, m->data, r0);
  }
  AS_HIP(ctx, hipGetLastError());
  ctx->stats["graph_directed_edges"] = (int64_t)n_edges_dir;
  ctx->stats["graph_nnz"] = (int64_t)nnz;
  *out = m;
  holder.p = nullptr;
  return AS_OK;
}

int laplacian_from_knn(as_ctx *ctx, const uint32_t *nbr_idx, const double *nbr_dist, uint64_t n, uint32_t K, const as_graph_params *gp, as_csr **out,
                       uint64_t row_begin, uint64_t row_count) {
  return assemble_rows(ctx, nbr_idx, nbr_dist, nullptr, n, K, gp, out, row_begin, row_count);
}
int laplacian_from_adjacency(as_ctx *ctx, const GraphAdjacency *adj, uint64_t n, uint32_t K, const as_graph_params *gp, as_csr **out, uint64_t row_begin,
                             uint64_t row_count) {
  return assemble_rows(ctx, nullptr, nullptr, adj, n, K, gp, out, row_begin, row_count);
}

// ---- the first half for the rows of one rank (row t of the table is node row_ids[t]) ---------------------------------------------
int graph_degrees_of_rows(as_ctx *ctx, const uint32_t *nbr_idx, const double *nbr_dist, const uint32_t *row_ids, uint64_t n_rows, uint32_t K,
                          const as_graph_params *gp, uint32_t *deg_out) {
  if (!n_rows) return AS_OK;
  hipLaunchKernelGGL(degree_by_ids_kernel, dim3(cdiv(n_rows, 256)), dim3(256), 0, ctx->stream, nbr_idx, nbr_dist, row_ids, (uint32_t)n_rows, K, gp->eps, deg_out);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}
int graph_degrees_to_nodes(as_ctx *ctx, const uint32_t *deg_pos, const uint32_t *perm, uint64_t n, uint32_t *deg, unsigned long long *deg_sum) {
  AS_HIP(ctx, hipMemsetAsync(deg_sum, 0, sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(degree_to_nodes_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, deg_pos, perm, (uint32_t)n, deg, deg_sum);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}
int graph_adjacency_of_rows(as_ctx *ctx, const uint32_t *nbr_idx, const double *nbr_dist, const uint32_t *row_ids, uint64_t n_rows, uint64_t n_nodes,
                            uint32_t K, const as_graph_params *gp, const uint32_t *deg, const unsigned long long *deg_sum, uint32_t *adj_j, double *adj_w,
                            uint32_t *adj_len) {
  if (K > 32) return fail(ctx, AS_ERR_ARG, "graph_adjacency_of_rows: topk + 1 must be <= 32");
  if (!n_rows) return AS_OK;
  const double sigma = gp->has_sigma ? gp->sigma : 1.0;
  hipLaunchKernelGGL((adjacency32_kernel<1, true>), dim3((unsigned)cdiv(n_rows * 32, (uint64_t)256)), dim3(256), 0, ctx->stream, nbr_idx, nbr_dist,
                     (uint32_t)n_rows, K, gp->eps, sigma, gp->p, deg, deg_sum, adj_j, adj_w, adj_len, (uint32_t *)nullptr, 0u, 0u, (uint32_t *)nullptr, row_ids,
                     (uint32_t)n_nodes);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}
int graph_adjacency_to_nodes(as_ctx *ctx, const uint32_t *pj, const double *pw, const uint32_t *plen, const uint32_t *perm, uint64_t n, uint32_t K,
                             uint32_t *adj_j, double *adj_w, uint32_t *adj_len) {
  hipLaunchKernelGGL(adjacency_to_nodes_kernel, dim3((unsigned)cdiv(n * 32, (uint64_t)256)), dim3(256), 0, ctx->stream, pj, pw, plen, perm, (uint32_t)n, K,
                     adj_j, adj_w, adj_len);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

}  // namespace asx
