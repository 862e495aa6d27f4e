// stream_scan.hip -- exact scan of a FEW rows against all rows in one pass over the matrix at HBM speed.
//
// The bandwidth-bound regime of K9 (ArrowSpace::search_lambda_aware for one query or a small batch,
// src/core.rs:760-798: every call reads all N items, N (8 D + 8) bytes) and the tail of K2 (the handful of graph rows
// the MFMA stages cannot certify).  The MFMA pipeline (guesses, threshold scan, re-rank, certification) is built for
// thousands of rows per call: at one query its launches and host round trips cost 1.4 ms for 1 GB of items.  Here the
// scores are the EXACT ones from the start -- the reference's sequential f64 multiply / add chain over k (scan_topk.h
// policies) -- so there is nothing to certify:
//   1. sample pass   exact keys of every A row against 1 in 16 of the 8-row tiles of B (a hashed pick per 16 tiles);
//   2. guess         g_a = the K-th smallest sampled key of row a.  The K-th smallest key over ALL of B is <= g_a, so
//                    every member of the true top K has key <= g_a: a guaranteed superset, not a heuristic;
//   3. list pass     all of B once: (column, key) with key <= g_a appended to row a's list (~16 K entries);
//   4. final         per row, the listed entries ordered by (key, column), the first K written.
// A list that overflows (a sample that misses the good region entirely: ordered data) is reported and the caller
// takes the old path for that row.
//
// Geometry (dim 128, rows of 1 KB): 6 waves per workgroup, a wave owns tiles of 8 B rows which arrive by LDS-DMA (one
// global_load_lds per row, whole cache lines, 1040-B pitch) into a 2-deep ring: ~100 KB in flight per CU.  Eight lanes
// share a B row and split the A rows (lane s takes rows s, s + 8, ...: up to 4 accumulators); B values are LDS
// broadcasts, A values come from an LDS copy of the A rows at the same pitch (8 rows per read, 32 distinct banks).
#include <type_traits>

#include "scan_topk.h"

namespace asx {

namespace {

typedef __attribute__((address_space(3))) void ss_lds_void;
typedef const __attribute__((address_space(1))) void ss_gbl_void;

constexpr int kSsWaves = 6;
constexpr int kSsTile = 8;            // B rows per wave tile
constexpr int kSsPitchB = 1040;       // bytes between staged B rows
constexpr int kSsPitchA = 1040;       // bytes between A rows (16-B aligned rows, 4 banks apart)
constexpr int kSsMaxA = 32;           // A rows per launch
constexpr uint32_t kSsSampleStride = 16;  // 1 in 16 tiles is sampled
constexpr uint32_t kSsD = 128;
constexpr uint32_t kSsShares = 4096;  // running minima per A row in the sample pass (disjoint shares of the sample)

__global__ void fill_u64_kernel(unsigned long long *__restrict__ p, uint64_t n, unsigned long long v) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

__device__ __forceinline__ uint32_t ss_lds_addr(const void *p) { return (uint32_t)(size_t)(ss_lds_void *)p; }
template <int B, int E, class F>
__device__ __forceinline__ void ss_static_for(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    ss_static_for<B + 1, E>(f);
  }
}

__device__ __forceinline__ void ss_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// order-preserving map of a double onto uint64 (ascending) and back
__device__ __forceinline__ unsigned long long ss_orderable(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double ss_from_orderable(unsigned long long u) {
  const unsigned long long b = (u >> 63) ? (u & 0x7FFFFFFFFFFFFFFFull) : ~u;
  return __longlong_as_double((long long)b);
}

// LIST = false: sample pass, 4096 running minima per row over disjoint shares of the sampled columns.
// LIST = true: list pass over all tiles, entries with key <= guess[a] to list[a].
// LDS reads of the chain are issued by hand: behind a pending LDS-DMA the compiler puts s_waitcnt vmcnt(0) in front of
// every LDS read it generates itself (the DMA is a pending LDS write it cannot tell apart), which drained the whole ring
// before each tile: one tile in flight instead of NBUF - 1 (2.6 TB/s).
template <int OFF>
__device__ __forceinline__ void ss_lds_read16(double2 &dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void ss_lds_wait_all() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);  // consumers of the registers just waited for stay behind the wait
}

// The per-row operands of the policies (norms, lambdas) without a plain global load inside the sweep: a plain load
// issued behind the DMA of the next tiles cannot complete before them (vmcnt retires in order), so waiting for it
// would drain the ring.  The B-side values ride with the tile (one more DMA piece: 8 norms + 8 lambdas), the A-side
// values are read once before the sweep.
template <class Policy>
struct StreamOperands;
template <>
struct StreamOperands<SearchScorePolicy> {
  static constexpr bool kLambda = true;
  double qn, ql;  // of one A row
  __device__ __forceinline__ void load_a(const SearchScorePolicy &p, uint32_t ga) {
    qn = p.q_norms[ga];
    ql = p.q_lambdas[ga];
  }
  __device__ static __forceinline__ const double *b_norms(const SearchScorePolicy &p) { return p.x_norms; }
  __device__ static __forceinline__ const double *b_lambdas(const SearchScorePolicy &p) { return p.x_lambdas; }
  // the arithmetic of SearchScorePolicy::key (scan_topk.h), operands passed by value
  __device__ __forceinline__ bool key(const SearchScorePolicy &p, double dot, uint32_t, uint32_t, double xn, double xl, double &out,
                                      int *err) const {
    const double denom = __dmul_rn(qn, xn);
    const double c = denom > 0.0 ? __ddiv_rn(dot, denom) : 0.0;
    const double diff = fabs(__dsub_rn(ql, xl));
    const double lam = __dsub_rn(1.0, fmin(diff, 1.0));
    const double sc = __dadd_rn(__dmul_rn(p.alpha, c), __dmul_rn(__dsub_rn(1.0, p.alpha), lam));
    if (sc != sc) {
      atomicOr(err, 1);
      return false;
    }
    out = -sc;
    return true;
  }
  // the same arithmetic without the error flag (a NaN key comes back as NaN): the filter below only decides who is looked at
  __device__ __forceinline__ double key_quiet(const SearchScorePolicy &p, double dot, uint32_t, uint32_t, double xn, double xl) const {
    const double denom = __dmul_rn(qn, xn);
    const double c = denom > 0.0 ? __ddiv_rn(dot, denom) : 0.0;
    const double diff = fabs(__dsub_rn(ql, xl));
    const double lam = __dsub_rn(1.0, fmin(diff, 1.0));
    return -__dadd_rn(__dmul_rn(p.alpha, c), __dmul_rn(__dsub_rn(1.0, p.alpha), lam));
  }
};
template <>
struct StreamOperands<GraphDistPolicy> {
  static constexpr bool kLambda = false;
  double na;
  __device__ __forceinline__ void load_a(const GraphDistPolicy &p, uint32_t ga) { na = p.norms[ga]; }
  __device__ static __forceinline__ const double *b_norms(const GraphDistPolicy &p) { return p.norms; }
  __device__ static __forceinline__ const double *b_lambdas(const GraphDistPolicy &) { return nullptr; }
  // the arithmetic of GraphDistPolicy::key
  __device__ __forceinline__ bool key(const GraphDistPolicy &p, double dot, uint32_t a, uint32_t j, double nj, double, double &out,
                                      int *) const {
    if (a == j) return false;
    const double denom = __dmul_rn(na, nj);
    double c = denom > 1e-12 ? __ddiv_rn(dot, denom) : 0.0;
    if (p.rectify) {
      c = fmin(fmax(c, -1.0), 1.0);
      c = fmax(c, 0.0);
    }
    out = __dsub_rn(1.0, c);
    return !(out != out);
  }
  __device__ __forceinline__ double key_quiet(const GraphDistPolicy &p, double dot, uint32_t a, uint32_t j, double nj, double) const {
    if (a == j) return __builtin_inf();  // (a row is not its own neighbour: never looked at)
    const double denom = __dmul_rn(na, nj);
    double c = denom > 1e-12 ? __ddiv_rn(dot, denom) : 0.0;
    if (p.rectify) {
      c = fmin(fmax(c, -1.0), 1.0);
      c = fmax(c, 0.0);
    }
    return __dsub_rn(1.0, c);
  }
};

// ---- the list pass at HBM speed for up to 8 A rows (the first search of an index, one query or a few) ------------------------
// The exact chain of the reference (128 dependent multiply / add pairs per item, one lane per (item, query)) keeps 8 of a
// wave's 64 lanes busy at one query and pulls every B row through LDS eight times wider than it is used: 0.28 ms per GB, LDS-
// and latency-bound.  But the chain is only needed for the items that can reach the cut.  Here every lane takes 16 B of every
// row straight from HBM (a wave reads 8 rows = 8 KB per trip with eight fully coalesced loads, two trips in flight: no LDS),
// multiplies them with its 16 B of each query, and the partial products are added across the wave by a halving butterfly
// (shuffles; a fixed order: the same sums in every run).  That dot product differs from the reference's sequential one by at
// most 2 x 128 x 2^-53 x |x||q| (both are within 128 ulp-sized steps of the true sum), i.e. the cosine by < 3e-14 and the key
// by less still, so every pair whose QUICK key is within `margin` = 1e-12 of the cut is queued for stream_pairs_kernel, which
// runs the reference's chain for exactly those (~16 K per query) and lists the keys at or below the cut: the lists, and with
// them the results, are those of the full exact pass.  Pairs with a non-finite quick key are queued too.
template <int CUR, int LB>
__device__ __forceinline__ void ssq_halve(double *v, int lane) {
  constexpr int H = CUR / 2;
  const bool up = ((lane >> LB) & 1) != 0;
#pragma unroll
  for (int i = 0; i < H; ++i) {
    const double keep = up ? v[i + H] : v[i];
    const double send = up ? v[i] : v[i + H];
    v[i] = keep + __shfl_xor(send, 1 << LB, 64);
  }
}
template <class Policy, int NQ>
__global__ __launch_bounds__(256) void stream_filter_f64_kernel(const double *__restrict__ A, const uint32_t *__restrict__ a_ids, uint32_t a_begin,
                                                               uint32_t nA, const double *__restrict__ B, uint32_t nB, Policy pol,
                                                               const double *__restrict__ guess, double margin, uint32_t cap,
                                                               uint32_t *__restrict__ count, uint32_t *__restrict__ pairs,
                                                               int *__restrict__ overflow) {
  static_assert(NQ == 1 || NQ == 2 || NQ == 4 || NQ == 8, "queries per launch");
  constexpr int QB = NQ == 1 ? 0 : (NQ == 2 ? 1 : (NQ == 4 ? 2 : 3));
  const int lane = threadIdx.x & 63;
  // the wave's finds go through an LDS queue and out with one reservation on the global counter per flush (stream_filter_i8_kernel)
  constexpr uint32_t kQ = 1024;
  __shared__ uint32_t s_q[4][kQ];
  __shared__ uint32_t s_qn[4];
  uint32_t *q = s_q[threadIdx.x >> 6];
  uint32_t *qn_p = &s_qn[threadIdx.x >> 6];
  if (lane == 0) *qn_p = 0;
  ss_wave_sync();
  auto flush = [&]() {
    ss_wave_sync();
    const uint32_t qn = *(volatile uint32_t *)qn_p;
    if (qn != 0) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(count, qn);
      base = __builtin_amdgcn_readfirstlane(base);
      if (base + qn > cap) {
        if (lane == 0) *overflow = 1;
      } else {
        for (uint32_t e = lane; e < qn; e += 64) pairs[base + e] = q[e];
      }
    }
    ss_wave_sync();
    if (lane == 0) *qn_p = 0;
    ss_wave_sync();
  };
  // this lane's 16 B of every query (elements 2 lane, 2 lane + 1); queries beyond the batch are zero
  double2 qv[NQ];
#pragma unroll
  for (int a = 0; a < NQ; ++a) {
    qv[a] = double2{0.0, 0.0};
    if ((uint32_t)a < nA) {
      const uint32_t ga = a_ids ? a_ids[a] : a_begin + (uint32_t)a;
      qv[a] = reinterpret_cast<const double2 *>(A + (uint64_t)ga * kSsD)[lane];
    }
  }
  // the (row of the trip, query) this lane holds once the butterfly is through: row = lane >> 3, query = the top QB of its low
  // three bits; the lanes whose other low bits are zero speak for the pair
  const uint32_t row_sel = (uint32_t)lane >> 3;
  const uint32_t a_fin = NQ == 1 ? 0u : ((uint32_t)lane & 7u) >> (3 - QB);
  const bool speaker = (((uint32_t)lane & 7u) & ((8u >> QB) - 1u)) == 0u && a_fin < nA;
  const uint32_t ga_fin = a_fin < nA ? (a_ids ? a_ids[a_fin] : a_begin + a_fin) : 0u;
  StreamOperands<Policy> op;
  op.load_a(pol, ga_fin);
  const double cut = a_fin < nA ? guess[a_fin] + margin : -__builtin_inf();
  const double *b_norms = StreamOperands<Policy>::b_norms(pol);
  const double *b_lams = StreamOperands<Policy>::b_lambdas(pol);
  const uint32_t n_tiles = (nB + kSsTile - 1) / kSsTile;
  const uint32_t wave_id = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
  auto load_tile = [&](uint32_t t, double2 (&x)[kSsTile], double &xn, double &xl) {
    const uint32_t row0 = t * kSsTile;
#pragma unroll
    for (int r = 0; r < kSsTile; ++r) {
      const uint32_t row = min(row0 + (uint32_t)r, nB - 1);
      typedef double ssq_d2 __attribute__((ext_vector_type(2)));
      const ssq_d2 w = __builtin_nontemporal_load(reinterpret_cast<const ssq_d2 *>(B + (uint64_t)row * kSsD) + lane);
      x[r] = double2{w.x, w.y};
    }
    const uint32_t jr = min(row0 + row_sel, nB - 1);
    xn = b_norms[jr];
    xl = StreamOperands<Policy>::kLambda ? b_lams[jr] : 0.0;
  };
  double2 xa[kSsTile], xb[kSsTile];
  double na = 0.0, la = 0.0, nb = 0.0, lb = 0.0;
  uint32_t t = wave_id;
  if (t < n_tiles) load_tile(t, xa, na, la);
  auto consume = [&](uint32_t tt, const double2 (&x)[kSsTile], double xn, double xl) {
    double v[kSsTile * NQ];  // index = row * NQ + query: the row bits on top
#pragma unroll
    for (int r = 0; r < kSsTile; ++r)
#pragma unroll
      for (int a = 0; a < NQ; ++a) v[r * NQ + a] = __builtin_fma(x[r].y, qv[a].y, x[r].x * qv[a].x);
    ssq_halve<8 * NQ, 5>(v, lane);
    ssq_halve<4 * NQ, 4>(v, lane);
    ssq_halve<2 * NQ, 3>(v, lane);
    if constexpr (NQ >= 2) ssq_halve<NQ, 2>(v, lane);
    if constexpr (NQ >= 4) ssq_halve<NQ / 2, 1>(v, lane);
    if constexpr (NQ >= 8) ssq_halve<NQ / 4, 0>(v, lane);
    double dot = v[0];
    if constexpr (NQ <= 4) dot += __shfl_xor(dot, 1, 64);
    if constexpr (NQ <= 2) dot += __shfl_xor(dot, 2, 64);
    if constexpr (NQ <= 1) dot += __shfl_xor(dot, 4, 64);
    const uint32_t j = tt * kSsTile + row_sel;
    bool take = false;
    if (speaker && j < nB) {
      const double key = op.key_quiet(pol, dot, ga_fin, j, xn, xl);
      take = !(key > cut);  // (NaN included: the exact arithmetic has to see it)
    }
    if (__any(take)) {
      ss_wave_sync();
      if (*(volatile uint32_t *)qn_p + 64 > kQ) flush();
      if (take) q[atomicAdd(qn_p, 1u)] = (a_fin << 27) | j;
    }
  };
  for (; t < n_tiles; t += 2 * n_waves) {
    const uint32_t t1 = t + n_waves, t2 = t + 2 * n_waves;
    if (t1 < n_tiles) load_tile(t1, xb, nb, lb);
    consume(t, xa, na, la);
    if (t1 >= n_tiles) break;
    if (t2 < n_tiles) load_tile(t2, xa, na, la);
    consume(t1, xb, nb, lb);
  }
  flush();
}

constexpr int kSsTileBytes = kSsTile * kSsPitchB + 128;  // 8 rows + 8 norms + 8 lambdas
constexpr int kSsPieces = kSsTile + 1;                   // DMA instructions per tile

// G = accumulators per lane = ceil(nA / 8): lane s holds the A rows s, s + 8, ...
typedef int ssf_i32x4 __attribute__((ext_vector_type(4)));
typedef int ssf_v16i32 __attribute__((ext_vector_type(16)));

// The filter of the small-batch search on an index that keeps a rotated int8 image of its items (mfma_prefilter_i8.hip: 31
// dominant directions + tail norm, 32 bytes per item: 32 MB for a million items instead of the 1 GB of rows).  score =
// alpha cos + (1 - alpha) lam with lam in [0, 1], and cos <= s8 / 127^2 + delta8, so an item can only reach query a's cut
// S_a = -guess[a] (guaranteed to keep its top K: stream_guess_kernel) if  s8 >= T_a = ceil(((S_a - (1 - alpha)) / alpha -
// delta8) 127^2): one v_mfma_i32_32x32x32_i8 per 32 items against the (up to 32) queries with -T_a as its C operand.  Items
// that pass, and still pass with the lambda term of the pair in place of its largest value, are queued as (query, item) pairs
// for stream_pairs_kernel: ~0.4 % of all pairs.
__global__ __launch_bounds__(256) void stream_filter_i8_kernel(const ssf_i32x4 *__restrict__ P8, uint32_t n, const ssf_i32x4 *__restrict__ PQ8,
                                                              uint32_t nq, const double *__restrict__ guess, double key_offset,
                                                              const double *__restrict__ q_lams, double alpha,
                                                              float err_items, const unsigned int *__restrict__ err_q_bits,
                                                              const double *__restrict__ x_lams, uint32_t cap,
                                                              uint32_t *__restrict__ count, uint32_t *__restrict__ pairs,
                                                              int *__restrict__ overflow) {
  const int lane = threadIdx.x & 63, lh = lane >> 5;
  const double E = (double)__builtin_fmaxf(err_items, __uint_as_float(*err_q_bits));
  const double delta8 = (2.0 * E + E * E) * (1.0 + 1e-6) + 1e-4;
  ssf_v16i32 nth;
  double cutv[16], qlv[16];  // the cut and the lambda of the lane's 16 query rows
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const uint32_t a = 4u * lh + (r & 3) + 8u * (r >> 2);
    int T = 1 << 29;  // rows beyond the batch: never
    cutv[r] = __builtin_inf();
    qlv[r] = 0.0;
    if (a < nq) {
      const double cut = key_offset - guess[a];  // the score (search: key = -score) / cosine (graph: key = 1 - cos) to reach
      cutv[r] = cut == cut ? cut : -__builtin_inf();
      qlv[r] = q_lams ? q_lams[a] : 0.0;
      // (a NaN or -inf cut -- fewer than K sampled keys -- lets everything through)
      const double x = __builtin_ceil((((cut - (1.0 - alpha)) / alpha) - delta8 - 1e-9) * (127.0 * 127.0));
      T = x == x ? (int)__builtin_fmax(__builtin_fmin(x, 536870911.0), -536870911.0) : -(1 << 29);
    }
    nth[r] = -T;
  }
  const ssf_i32x4 aq = PQ8[lane];  // the first 32 positions of the queries' image
  // the wave's finds -- (query << 27 | item) -- are queued in LDS and go out with ONE reservation on the global counter per
  // flush (a returning atomic per find on one address serialises in L2: 0.3 ms for 50 k finds)
  constexpr uint32_t kQ = 2304;  // a trip of four sub tiles adds at most 4096 -> flushed before every sub tile that might not fit
  __shared__ uint32_t s_q[4][kQ];
  __shared__ uint32_t s_qn[4];
  uint32_t *q = s_q[threadIdx.x >> 6];
  uint32_t *qn_p = &s_qn[threadIdx.x >> 6];
  if (lane == 0) *qn_p = 0;
  ss_wave_sync();
  auto flush = [&]() {
    ss_wave_sync();
    const uint32_t qn = *(volatile uint32_t *)qn_p;
    if (qn != 0) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(count, qn);
      base = __builtin_amdgcn_readfirstlane(base);
      if (base + qn > cap) {
        if (lane == 0) *overflow = 1;
      } else {
        for (uint32_t e = lane; e < qn; e += 64) pairs[base + e] = q[e];
      }
    }
    ss_wave_sync();
    if (lane == 0) *qn_p = 0;
    ss_wave_sync();
  };
  const uint32_t n_sub = (n + 31) / 32, wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
  for (uint32_t st0 = wave * 4; st0 < n_sub; st0 += n_waves * 4) {  // four sub tiles per trip: their loads fly together
    ssf_i32x4 bx[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) bx[u] = P8[(size_t)min(st0 + (uint32_t)u, n_sub - 1) * 64 + lane];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t st = st0 + (uint32_t)u;
      const ssf_v16i32 acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(aq, bx[u], nth, 0, 0, 0);
      int m = max(max(acc[0], acc[1]), acc[2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) m = max(max(m, acc[r]), acc[r + 1]);
      m = max(m, acc[15]);
      const uint32_t j = st * 32 + (uint32_t)(lane & 31);
      const bool ok = m >= 0 && st < n_sub && j < n;
      if (!__any(ok)) continue;
      // (wave-uniform: every lane reads the same word -- through a volatile access: the count is advanced by LDS atomics of this
      // wave's lanes, and a value the compiler kept from an earlier trip would let the queue run into the next wave's)
      ss_wave_sync();
      if (*(volatile uint32_t *)qn_p + 1024 > kQ) flush();
      if (ok) {  // the lambda term as the score computes it, in place of its largest value 1 (rare path: a few lanes)
        const double lx = x_lams ? x_lams[j] : 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (acc[r] < 0) continue;
          const double cos_ub = (double)(acc[r] - nth[r]) * (1.0 / (127.0 * 127.0)) + delta8;
          const double diff = fabs(__dsub_rn(qlv[r], lx));
          const double lam = __dsub_rn(1.0, fmin(diff, 1.0));
          // (a NaN lambda keeps the pair: the exact arithmetic has to see it)
          if (!(alpha * cos_ub + (1.0 - alpha) * lam + 1e-9 < cutv[r])) {
            const uint32_t a = 4u * lh + (r & 3) + 8u * (r >> 2);
            q[atomicAdd(qn_p, 1u)] = (a << 27) | j;
          }
        }
      }
    }
  }
  flush();
}

// ---- the same quick pass for rows of 256 .. 1024 values (128 P), and as the SAMPLE pass of such rows -------------------------
// A row arrives as P pieces of 1 KB (eight rows x one piece per trip: the same eight coalesced loads); the partial products of a
// (row, query) pair add up over the pieces before the butterfly.  The queries live in LDS (P KB each), a lane re-reads its 16 B
// of every query per piece.  SAMPLE: one tile in 16 (the hashed pick of the exact sample pass), the quick keys go to the share
// minima stream_guess_kernel reads: guess = the K-th smallest SAMPLED quick key, so the K-th smallest exact key over all rows
// is at most guess + margin, the list must hold every exact key <= guess + margin, and the filter lets every quick key
// <= guess + 2 margin through.
template <class Policy, int NQ, bool SAMPLE>
__global__ __launch_bounds__(256) void stream_quick_wide_kernel(const double *__restrict__ A, const uint32_t *__restrict__ a_ids, uint32_t a_begin,
                                                               uint32_t nA, const double *__restrict__ B, uint32_t nB, uint32_t P, Policy pol,
                                                               const double *__restrict__ guess, double cut_margin,
                                                               unsigned long long *__restrict__ sample_min, uint32_t cap,
                                                               uint32_t *__restrict__ count, uint32_t *__restrict__ pairs,
                                                               int *__restrict__ overflow) {
  static_assert(NQ == 1 || NQ == 2 || NQ == 4, "queries per launch");
  constexpr int QB = NQ == 1 ? 0 : (NQ == 2 ? 1 : 2);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double2 *s_q = reinterpret_cast<double2 *>(smem);  // [NQ][P][64]
  const int lane = threadIdx.x & 63;
  const uint32_t D = 128u * P;
  for (uint32_t e = threadIdx.x; e < (uint32_t)NQ * P * 64u; e += blockDim.x) {
    const uint32_t a = e / (P * 64u), rest = e - a * (P * 64u);
    double2 v{0.0, 0.0};
    if (a < nA) {
      const uint32_t ga = a_ids ? a_ids[a] : a_begin + a;
      v = reinterpret_cast<const double2 *>(A + (uint64_t)ga * D)[rest];
    }
    s_q[e] = v;
  }
  constexpr uint32_t kQ = 512;
  __shared__ uint32_t s_qq[4][kQ];
  __shared__ uint32_t s_qn[4];
  uint32_t *q = s_qq[threadIdx.x >> 6];
  uint32_t *qn_p = &s_qn[threadIdx.x >> 6];
  if (lane == 0) *qn_p = 0;
  __syncthreads();
  auto flush = [&]() {
    ss_wave_sync();
    const uint32_t qn = *(volatile uint32_t *)qn_p;
    if (qn != 0) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(count, qn);
      base = __builtin_amdgcn_readfirstlane(base);
      if (base + qn > cap) {
        if (lane == 0) *overflow = 1;
      } else {
        for (uint32_t e = lane; e < qn; e += 64) pairs[base + e] = q[e];
      }
    }
    ss_wave_sync();
    if (lane == 0) *qn_p = 0;
    ss_wave_sync();
  };
  const uint32_t row_sel = (uint32_t)lane >> 3;
  const uint32_t a_fin = NQ == 1 ? 0u : ((uint32_t)lane & 7u) >> (3 - QB);
  const bool speaker = (((uint32_t)lane & 7u) & ((8u >> QB) - 1u)) == 0u && a_fin < nA;
  const uint32_t ga_fin = a_fin < nA ? (a_ids ? a_ids[a_fin] : a_begin + a_fin) : 0u;
  StreamOperands<Policy> op;
  op.load_a(pol, ga_fin);
  const double cut = (!SAMPLE && a_fin < nA) ? guess[a_fin] + cut_margin : -__builtin_inf();
  const double *b_norms = StreamOperands<Policy>::b_norms(pol);
  const double *b_lams = StreamOperands<Policy>::b_lambdas(pol);
  const uint32_t n_tiles = (nB + kSsTile - 1) / kSsTile;
  const uint32_t n_my_all = SAMPLE ? (n_tiles + kSsSampleStride - 1) / kSsSampleStride : n_tiles;
  const uint32_t wave_id = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
  auto tile_of = [&](uint32_t t) -> uint32_t {
    if (!SAMPLE) return t;
    uint32_t h = t * 0x9E3779B9u;
    h ^= h >> 15;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    const uint32_t pick = t * kSsSampleStride + (h & (kSsSampleStride - 1));
    return pick < n_tiles ? pick : t * kSsSampleStride;
  };
  // this wave's tiles: wave_id, wave_id + n_waves, ...; a unit = (tile ordinal i, piece p)
  const uint32_t n_mine = wave_id < n_my_all ? (n_my_all - wave_id + n_waves - 1) / n_waves : 0u;
  const uint32_t n_units = n_mine * P;
  typedef double ssq_d2 __attribute__((ext_vector_type(2)));
  auto load_unit = [&](uint32_t u, double2 (&x)[kSsTile]) {
    const uint32_t i = u / P, p = u - i * P;
    const uint32_t row0 = tile_of(wave_id + i * n_waves) * kSsTile;
#pragma unroll
    for (int r = 0; r < kSsTile; ++r) {
      const uint32_t row = min(row0 + (uint32_t)r, nB - 1);
      const ssq_d2 w = __builtin_nontemporal_load(reinterpret_cast<const ssq_d2 *>(B + (uint64_t)row * D + (uint64_t)p * 128u) + lane);
      x[r] = double2{w.x, w.y};
    }
  };
  double v[kSsTile * NQ];
  auto consume = [&](uint32_t u, const double2 (&x)[kSsTile]) {
    const uint32_t i = u / P, p = u - i * P;
    double2 qv[NQ];
#pragma unroll
    for (int a = 0; a < NQ; ++a) qv[a] = s_q[((uint32_t)a * P + p) * 64u + (uint32_t)lane];
    if (p == 0) {
#pragma unroll
      for (int e = 0; e < kSsTile * NQ; ++e) v[e] = 0.0;
    }
#pragma unroll
    for (int r = 0; r < kSsTile; ++r)
#pragma unroll
      for (int a = 0; a < NQ; ++a) v[r * NQ + a] = __builtin_fma(x[r].y, qv[a].y, __builtin_fma(x[r].x, qv[a].x, v[r * NQ + a]));
    if (p + 1 != P) return;
    const uint32_t t_ord = wave_id + i * n_waves, tt = tile_of(t_ord);
    const uint32_t jr = min(tt * kSsTile + row_sel, nB - 1);
    const double xn = b_norms[jr];
    const double xl = StreamOperands<Policy>::kLambda ? b_lams[jr] : 0.0;
    ssq_halve<8 * NQ, 5>(v, lane);
    ssq_halve<4 * NQ, 4>(v, lane);
    ssq_halve<2 * NQ, 3>(v, lane);
    if constexpr (NQ >= 2) ssq_halve<NQ, 2>(v, lane);
    if constexpr (NQ >= 4) ssq_halve<NQ / 2, 1>(v, lane);
    double dot = v[0];
    if constexpr (NQ <= 4) dot += __shfl_xor(dot, 1, 64);
    if constexpr (NQ <= 2) dot += __shfl_xor(dot, 2, 64);
    if constexpr (NQ <= 1) dot += __shfl_xor(dot, 4, 64);
    const uint32_t j = tt * kSsTile + row_sel;
    if (SAMPLE) {
      if (speaker && j < nB) {
        const double key = op.key_quiet(pol, dot, ga_fin, j, xn, xl);
        if (key == key && key < __builtin_inf())
          atomicMin(sample_min + (size_t)a_fin * kSsShares + ((t_ord * kSsTile + row_sel) & (kSsShares - 1)), ss_orderable(key));
      }
    } else {
      bool take = false;
      if (speaker && j < nB) {
        const double key = op.key_quiet(pol, dot, ga_fin, j, xn, xl);
        take = !(key > cut);
      }
      if (__any(take)) {
        ss_wave_sync();
        if (*(volatile uint32_t *)qn_p + 64 > kQ) flush();
        if (take) q[atomicAdd(qn_p, 1u)] = (a_fin << 27) | j;
      }
    }
  };
  double2 xa[kSsTile], xb[kSsTile];
  if (n_units) load_unit(0, xa);
  for (uint32_t u = 0; u < n_units; u += 2) {
    if (u + 1 < n_units) load_unit(u + 1, xb);
    consume(u, xa);
    if (u + 1 >= n_units) break;
    if (u + 2 < n_units) load_unit(u + 2, xa);
    consume(u + 1, xb);
  }
  if (!SAMPLE) flush();
}

// The exact keys of the pairs the filter left: one lane per (query, item) pair, the reference's chain over k (multiply and add
// rounded separately, k ascending: src/core.rs:185-197) with the query rows in LDS; keys at or below the query's cut are listed
// for stream_final_kernel exactly as the list pass of the streaming scan lists them.
template <class Policy>
__global__ __launch_bounds__(256) void stream_pairs_kernel(const double *__restrict__ A, const uint32_t *__restrict__ a_ids, uint32_t nA,
                                                          const double *__restrict__ B, const uint32_t *__restrict__ perm, Policy pol,
                                                          const uint32_t *__restrict__ pairs,
                                                          const uint32_t *__restrict__ count, const int *__restrict__ overflow,
                                                          const double *__restrict__ guess, uint32_t cap, uint32_t *__restrict__ cnt,
                                                          uint32_t *__restrict__ list_idx, double *__restrict__ list_key,
                                                          int *__restrict__ err_flag, uint32_t d = kSsD, uint32_t a_begin = 0,
                                                          double list_margin = 0.0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *sA = reinterpret_cast<double *>(smem);  // nA x (d + 1) (the pad keeps the rows of different queries in different banks)
  if (*overflow) return;
  const uint32_t pitch = d + 1;
  for (uint32_t e = threadIdx.x; e < nA * d; e += blockDim.x) {
    const uint32_t a = e / d, k = e % d;
    sA[a * pitch + k] = A[(uint64_t)(a_ids ? a_ids[a] : a_begin + a) * d + k];
  }
  __syncthreads();
  const uint32_t n_pairs = *count;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n_pairs; p += gridDim.x * blockDim.x) {
    const uint32_t w = pairs[p], a = w >> 27, jpos = w & 0x7FFFFFFu;
    const uint32_t j = perm ? perm[jpos] : jpos, ga = a_ids ? a_ids[a] : a_begin + a;  // (the filter's image may be in a permuted order)
    const double *x = B + (uint64_t)j * d, *qa = sA + a * pitch;
    double acc = 0.0;
#pragma unroll 8
    for (uint32_t k = 0; k < d; ++k) acc = __dadd_rn(acc, __dmul_rn(qa[k], x[k]));
    double key;
    if (pol.key(acc, ga, j, key, err_flag) && key <= guess[a] + list_margin) {
      const uint32_t pos = atomicAdd(cnt + a, 1u);
      if (pos < cap) {
        list_idx[(size_t)a * cap + pos] = j;
        list_key[(size_t)a * cap + pos] = key;
      }
    }
  }
}

template <class Policy, bool LIST, int NBUF, int G>
__global__ __launch_bounds__(kSsWaves * 64) void stream_scan_d128_kernel(
    const double *__restrict__ A, const uint32_t *__restrict__ a_ids, uint32_t a_begin, uint32_t nA,
    const double *__restrict__ B, uint32_t nB, uint32_t n_tiles, Policy pol, unsigned long long *__restrict__ sample_min,
    const double *__restrict__ guess, uint32_t cap, uint32_t *__restrict__ cnt,
    uint32_t *__restrict__ list_idx, double *__restrict__ list_key, int *__restrict__ err_flag, const int *__restrict__ run_if) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // (LIST behind the int8 filter: only when its pair list ran out of room -- stream_pairs_kernel did the work otherwise)
  if (run_if && *run_if == 0) return;
  // [A rows: 8 ceil(nA / 8) x 1040][per wave: NBUF tiles]   (NBUF = 3 while the A rows leave room: nA <= 8)
  unsigned char *sA = smem;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char *ring = smem + (size_t)((nA + 7) / 8 * 8) * kSsPitchA + (size_t)wv * NBUF * kSsTileBytes;
  for (uint32_t e = tid; e < nA * kSsD; e += blockDim.x) {
    const uint32_t a = e / kSsD, k = e - a * kSsD;
    const uint32_t ga = a_ids ? a_ids[a] : a_begin + a;
    *reinterpret_cast<double *>(sA + (size_t)a * kSsPitchA + (size_t)k * 8) = A[(uint64_t)ga * kSsD + k];
  }
  const int item = lane >> 3, s = lane & 7;
  uint32_t ga[G];
  StreamOperands<Policy> ops[G];
  double my_guess[G];
#pragma unroll
  for (int q = 0; q < G; ++q) {
    const uint32_t a = (uint32_t)s + 8u * q;
    ga[q] = a < nA ? (a_ids ? a_ids[a] : a_begin + a) : 0u;
    ops[q].load_a(pol, ga[q]);
    my_guess[q] = (LIST && a < nA) ? guess[a] : 0.0;
  }
  __syncthreads();  // (also: every plain global load above has been waited for before the first DMA is issued)
  const double *b_norms = StreamOperands<Policy>::b_norms(pol);
  const double *b_lams = StreamOperands<Policy>::b_lambdas(pol);
  // this wave's tiles: all of them (LIST) or one in kSsSampleStride (sample), dealt round-robin over the waves of the grid
  const uint32_t n_my = LIST ? n_tiles : (n_tiles + kSsSampleStride - 1) / kSsSampleStride;
  const uint32_t wave_id = blockIdx.x * kSsWaves + wv, n_waves = gridDim.x * kSsWaves;
  // sampled tile of group t: a hashed member of tiles [16 t, 16 t + 16) -- a plain stride aliases with periodic row
  // orders (rows dealt round-robin over clusters: every sampled row would come from the same few clusters)
  auto tile_of = [&](uint32_t t) -> uint32_t {
    if (LIST) return t;
    uint32_t h = t * 0x9E3779B9u;
    h ^= h >> 15;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    const uint32_t pick = t * kSsSampleStride + (h & (kSsSampleStride - 1));
    return pick < n_tiles ? pick : t * kSsSampleStride;
  };
  // kSsPieces DMA instructions per tile, always (rows past the end of B are clamped to its last row: their keys are
  // never used), so that the vmcnt arithmetic below is exact
  auto issue = [&](uint32_t t, int buf) {
    const uint32_t row0 = tile_of(t) * kSsTile;
    unsigned char *dst = ring + (size_t)buf * kSsTileBytes;
#pragma unroll
    for (int r = 0; r < kSsTile; ++r) {
      const uint32_t row = min(row0 + (uint32_t)r, nB - 1);
      __builtin_amdgcn_global_load_lds((ss_gbl_void *)((const char *)(B + (uint64_t)row * kSsD) + lane * 16),
                                       (ss_lds_void *)(dst + r * kSsPitchB), 16, 0, 0);
    }
    if (lane < 8) {  // lanes 0-3: the 8 norms, lanes 4-7: the 8 lambdas (two doubles per lane; the other lanes are masked off)
      // a partial last tile reads a safe address instead (its operands are fetched by plain loads, below)
      const uint32_t r2 = row0 + kSsTile <= nB ? row0 + 2u * (uint32_t)(lane & 3) : 0u;
      const double *src = (lane < 4 || !StreamOperands<Policy>::kLambda) ? b_norms + r2 : b_lams + r2;
      __builtin_amdgcn_global_load_lds((ss_gbl_void *)src, (ss_lds_void *)(dst + kSsTile * kSsPitchB), 16, 0, 0);
    }
  };
  // NBUF - 1 tiles are in flight while one is being consumed
  uint32_t t = wave_id;
  int buf = 0;
#pragma unroll
  for (int p = 0; p < NBUF - 1; ++p)
    if (t + (uint32_t)p * n_waves < n_my) issue(t + (uint32_t)p * n_waves, p);
  for (; t < n_my; t += n_waves) {
    const uint32_t t_pre = t + (uint32_t)(NBUF - 1) * n_waves;
    if (t_pre < n_my) issue(t_pre, (buf + NBUF - 1) % NBUF);
    // tile t has landed once only the pieces of the younger tiles are in flight: `ahead` of them were issued
    uint32_t ahead = 0;
#pragma unroll
    for (int p = 1; p < NBUF; ++p) ahead += (t + (uint32_t)p * n_waves < n_my) ? 1u : 0u;
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_assert(kSsPieces == 9 && NBUF <= 3, "the vmcnt constants above are kSsPieces x tiles ahead");
    const unsigned char *tile = ring + (size_t)buf * kSsTileBytes;
    const unsigned char *xrow = tile + (size_t)item * kSsPitchB;
    double acc[G];
#pragma unroll
    for (int q = 0; q < G; ++q) acc[q] = 0.0;
    // the reference's chain: k ascending, multiply and add rounded separately (src/core.rs:185-197).  Chunks of 2 H k:
    // the B values and the A values of every accumulator go to registers first (LDS latency once per chunk, not once
    // per step of the dependent add chain); two register sets alternate, the next chunk's reads fly during this one's
    // arithmetic.
    constexpr int H = G == 1 ? 8 : (G == 2 ? 4 : 2);  // double2 per chunk and operand (register budget: 2 sets x (1 + G) x H x 4)
    struct Set {
      double2 x[H];
      double2 a[G][H];
    };
    const uint32_t x_addr = ss_lds_addr(xrow), a_addr = ss_lds_addr(sA) + (uint32_t)s * kSsPitchA;
    auto load = [&](Set &R, uint32_t k0) {
      ss_static_for<0, H>([&](auto h_c) {
        constexpr int h = decltype(h_c)::value;
        ss_lds_read16<h * 16>(R.x[h], x_addr + k0 * 8u);
      });
      ss_static_for<0, G * H>([&](auto e_c) {
        constexpr int q = decltype(e_c)::value / H, h = decltype(e_c)::value % H;
        ss_lds_read16<q * 8 * kSsPitchA + h * 16>(R.a[q][h], a_addr + k0 * 8u);
      });
    };
    auto compute = [&](const Set &R) {
#pragma unroll
      for (int h = 0; h < H; ++h)
#pragma unroll
        for (int q = 0; q < G; ++q) {
          acc[q] = __dadd_rn(acc[q], __dmul_rn(R.a[q][h].x, R.x[h].x));
          acc[q] = __dadd_rn(acc[q], __dmul_rn(R.a[q][h].y, R.x[h].y));
        }
    };
    Set ra, rb;
    // the policy operands of this lane's B row: norm and lambda, staged behind the rows
    double2 meta;  // (norm, lambda)
    {
      double2 nn, ll;
      ss_lds_read16<0>(nn, ss_lds_addr(tile + kSsTile * kSsPitchB) + (uint32_t)(item >> 1) * 16u);
      ss_lds_read16<64>(ll, ss_lds_addr(tile + kSsTile * kSsPitchB) + (uint32_t)(item >> 1) * 16u);
      load(ra, 0);
      ss_lds_wait_all();
      meta.x = (item & 1) ? nn.y : nn.x;
      meta.y = (item & 1) ? ll.y : ll.x;
    }
    if ((tile_of(t) + 1) * kSsTile > nB) {  // the partial last tile of B (wave-uniform, once per sweep): plain loads
      const uint32_t jj = min(tile_of(t) * kSsTile + (uint32_t)item, nB - 1);
      meta.x = b_norms[jj];
      if (StreamOperands<Policy>::kLambda) meta.y = b_lams[jj];
    }
#pragma unroll 1
    for (uint32_t k0 = 0; k0 < kSsD; k0 += 4 * H) {
      ss_lds_wait_all();
      load(rb, k0 + 2 * H);
      compute(ra);
      ss_lds_wait_all();
      if (k0 + 4 * H < kSsD) load(ra, k0 + 4 * H);
      compute(rb);
    }
    const uint32_t j = tile_of(t) * kSsTile + (uint32_t)item;
#pragma unroll
    for (int q = 0; q < G; ++q) {
      const uint32_t a = (uint32_t)s + 8u * q;
      if (a < nA && j < nB) {
        double key;
        const bool valid = ops[q].key(pol, acc[q], ga[q], j, meta.x, meta.y, key, err_flag);
        if (LIST) {
          if (valid && key <= my_guess[q]) {
            const uint32_t pos = atomicAdd(cnt + a, 1u);
            if (pos < cap) {
              list_idx[(size_t)a * cap + pos] = j;
              list_key[(size_t)a * cap + pos] = key;
            }
          }
        } else if (valid) {
          // running minima per row over disjoint shares of the sample (share = sampled column mod 4096), kept as
          // order-preserving integers so that atomicMin orders them like the keys
          atomicMin(sample_min + (size_t)a * kSsShares + ((t * kSsTile + (uint32_t)item) & (kSsShares - 1)), ss_orderable(key));
        }
      }
    }
    ss_wave_sync();  // everyone is done reading buffer `buf` before the next DMA into it is issued
    buf = (buf + 1) % NBUF;
  }
}

// guess[a] >= the K-th smallest sampled key of row a: K of the 256 share minima are K distinct sampled keys, so the
// K-th smallest minimum bounds the K-th smallest sampled key (and with it the K-th smallest key over all of B) from
// above.  +inf when fewer than K shares hold a key: everything is listed then (and the list overflows).
__global__ __launch_bounds__(256) void stream_guess_kernel(const unsigned long long *__restrict__ sample_min, uint32_t K,
                                                           double *__restrict__ guess) {
  __shared__ unsigned long long s_min[256];
  {  // 4096 share minima -> 256 (the minimum of 16 shares is the minimum of their union: still disjoint shares)
    unsigned long long v = ~0ull;
    for (uint32_t e = 0; e < kSsShares / 256; ++e) {
      const unsigned long long w = sample_min[(size_t)blockIdx.x * kSsShares + e * 256 + threadIdx.x];
      v = w < v ? w : v;
    }
    s_min[threadIdx.x] = v;
  }
  __syncthreads();
  for (uint32_t k = 2; k <= 256; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      const uint32_t e = threadIdx.x, partner = e ^ j;
      if (partner > e) {
        const bool up = (e & k) == 0;
        const unsigned long long a = s_min[e], b = s_min[partner];
        if (up ? (a > b) : (a < b)) {
          s_min[e] = b;
          s_min[partner] = a;
        }
      }
      __syncthreads();
    }
  if (threadIdx.x == 0) guess[blockIdx.x] = ss_from_orderable(s_min[K - 1]);
}

// per row: the listed entries ordered by (key, column), the first K written; overflow[a] = 1 when the list ran over
constexpr uint32_t kSsSortMax = 4096;
__global__ __launch_bounds__(256) void stream_final_kernel(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ list_idx,
                                                           const double *__restrict__ list_key, uint32_t cap, uint32_t K,
                                                           uint32_t out_ld, uint32_t out_off, double *__restrict__ out_key,
                                                           uint32_t *__restrict__ out_idx, int *__restrict__ overflow,
                                                           double *__restrict__ guess, int *__restrict__ done) {
  __shared__ double s_key[kSsSortMax];
  __shared__ uint32_t s_idx[kSsSortMax];
  const uint32_t a = blockIdx.x;
  if (done[a]) return;  // finished in an earlier round (its guess is -inf: nothing was listed for it since)
  const uint32_t c_all = cnt[a], c = min(c_all, cap);
  uint32_t P = 64;
  while (P < c) P <<= 1;
  for (uint32_t e = threadIdx.x; e < P; e += blockDim.x) {
    s_key[e] = e < c ? list_key[(size_t)a * cap + e] : __builtin_inf();
    s_idx[e] = e < c ? list_idx[(size_t)a * cap + e] : kNoIdx;
  }
  __syncthreads();
  for (uint32_t k = 2; k <= P; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t e = threadIdx.x; e < P; e += blockDim.x) {
        const uint32_t partner = e ^ j;
        if (partner > e) {
          const bool up = (e & k) == 0;
          const bool swap = up ? lex_less(s_key[partner], s_idx[partner], s_key[e], s_idx[e])
                               : lex_less(s_key[e], s_idx[e], s_key[partner], s_idx[partner]);
          if (swap) {
            const double tk = s_key[e];
            s_key[e] = s_key[partner];
            s_key[partner] = tk;
            const uint32_t ti = s_idx[e];
            s_idx[e] = s_idx[partner];
            s_idx[partner] = ti;
          }
        }
      }
      __syncthreads();
    }
  if (c_all > cap) {
    // The list ran over (a sample that misses the row's neighbourhood gives a cut that is far too loose).  The K-th best of
    // what WAS listed bounds the K-th best over all of B from above: it is the cut of the next round.
    if (threadIdx.x == 0) {
      overflow[a] = 1;
      guess[a] = s_key[K - 1];
    }
    return;
  }
  if (threadIdx.x == 0) {
    overflow[a] = 0;
    done[a] = 1;
    guess[a] = -__builtin_inf();
  }
  for (uint32_t e = threadIdx.x; e < K; e += blockDim.x) {
    const bool have = e < c && s_idx[e] != kNoIdx;
    out_key[(size_t)a * out_ld + out_off + e] = have ? s_key[e] : __builtin_inf();
    out_idx[(size_t)a * out_ld + out_off + e] = have ? s_idx[e] : kNoIdx;
  }
}

}  // namespace

bool stream_scan_applies(uint64_t nA, uint64_t nB, uint64_t d, uint32_t K) {
  return nA >= 1 && nA <= kSsMaxA && d == kSsD && K >= 1 && K <= 64 && nB >= 65536 && nB < 0xFFFFFFF0ull;
}

// The K best (key asc, column asc) of every listed A row against all nB rows of B (both 128 wide).  Rows of out are in
// list order.  *n_redo / redo_pos: the rows (positions in the list) whose candidate list overflowed; their output is
// undefined and the caller re-does them on the older path.
template <class Policy>
static int stream_scan(as_ctx *ctx, const char *prof_name, const double *A, const uint32_t *a_ids, uint32_t a_begin, uint32_t nA,
                       const double *B, uint32_t nB, uint32_t K, Policy pol, double *out_key, uint32_t *out_idx, uint32_t out_ld,
                       int *err_flag, std::vector<uint32_t> *redo_pos, const StreamFilter *filter_in = nullptr) {
  const StreamFilter *filter = (filter_in && (filter_in->PQ8 || (!a_ids && a_begin == 0)) && nB < (1u << 27)) ? filter_in : nullptr;  // (pairs: query << 27 | item)
  const uint32_t n_tiles = (nB + kSsTile - 1) / kSsTile;
  const uint32_t n_stiles = (n_tiles + kSsSampleStride - 1) / kSsSampleStride;
  const uint32_t cap = kSsSortMax;
  DevBuf<double> guess, lkey;
  DevBuf<unsigned long long> smin;
  DevBuf<uint32_t> lidx;
  // one block for the small zeroed arrays (one memset): cnt[nA] | over[nA + 2] | done[nA] | the queries' image error
  DevBuf<uint32_t> small;
  struct { uint32_t *p; } cnt{nullptr};
  struct { int *p; } over{nullptr}, done{nullptr};
  struct { unsigned int *p; } q_err{nullptr};
  std::vector<int> h(nA + 2);  // [nA], [nA + 1]: the filter's count and overflow flag (one copy back for all of it)
  int64_t retries = 0;
  AS_TRY(smin.alloc(ctx, (size_t)nA * kSsShares));
  AS_TRY(guess.alloc(ctx, nA));
  AS_TRY(lidx.alloc(ctx, (size_t)nA * cap));
  AS_TRY(lkey.alloc(ctx, (size_t)nA * cap));
  AS_TRY(small.alloc(ctx, (size_t)3 * nA + 3));
  AS_HIP(ctx, hipMemsetAsync(small.p, 0, sizeof(uint32_t) * ((size_t)3 * nA + 3), ctx->stream));
  cnt.p = small.p;
  over.p = reinterpret_cast<int *>(small.p + nA);
  done.p = reinterpret_cast<int *>(small.p + 2 * nA + 2);
  q_err.p = small.p + 3 * nA + 2;
  // the int8 filter in front of the list pass (search on an index that keeps a rotated image)
  DevBuf<uint32_t> g_pairs;
  DevBuf<unsigned char> pq8;
  uint32_t *const g_count = reinterpret_cast<uint32_t *>(over.p + nA);
  int *const g_over = over.p + nA + 1;
  const uint32_t g_cap = 1u << 20;
  // without that image (the first search of an index, leftover rows of a join without coordinates): up to 8 rows take the
  // quick f64 pass over B (stream_filter_f64_kernel) in front of the exact chains of the pairs it leaves
  const bool quick = !filter && nA <= 8 && nB < (1u << 27) && ctx->opt_search_stream_quick;
  if (filter) {
    AS_TRY(g_pairs.alloc(ctx, g_cap));
    AS_TRY(pq8.alloc(ctx, rotated_image_bytes(1)));
  } else if (quick) {
    AS_TRY(g_pairs.alloc(ctx, g_cap));
  }
  const int nbuf = nA <= 8 ? 3 : 2;  // a third tile in flight per wave while the A rows leave the LDS for it
  const size_t lds = (size_t)((nA + 7) / 8 * 8) * kSsPitchA + (size_t)kSsWaves * nbuf * kSsTileBytes;
  bool first_list = true;  // (the retries sweep all of B: their cut is the tighter one of a truncated list)
  {
    ProfScope ps(ctx, prof_name);
    hipLaunchKernelGGL(fill_u64_kernel, dim3(cdiv((uint64_t)nA * kSsShares, 256)), dim3(256), 0, ctx->stream, smin.p,
                       (uint64_t)nA * kSsShares, 0xFFF0000000000000ull);  // +inf as an order-preserving integer
#define AS_SS1(LISTV, NB, GV, GRID)                                                                                          \
  AS_HIP(ctx, hipFuncSetAttribute((const void *)stream_scan_d128_kernel<Policy, LISTV, NB, GV>,                              \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                     \
  hipLaunchKernelGGL((stream_scan_d128_kernel<Policy, LISTV, NB, GV>), dim3(GRID), dim3(kSsWaves * 64), lds, ctx->stream, A,   \
                     a_ids, a_begin, nA, B, nB, n_tiles, pol, smin.p, guess.p, cap, cnt.p, lidx.p, lkey.p, err_flag,       \
                     (LISTV && (filter || quick) && first_list) ? (const int *)g_over : (const int *)nullptr)
#define AS_SS(LISTV, GRID)                                                             \
  switch ((nA + 7) / 8) {                                                              \
    case 1: AS_SS1(LISTV, 3, 1, GRID); break;                                          \
    case 2: AS_SS1(LISTV, 2, 2, GRID); break;                                          \
    case 3: AS_SS1(LISTV, 2, 3, GRID); break;                                          \
    default: AS_SS1(LISTV, 2, 4, GRID); break;                                         \
  }
    const unsigned g_s = std::min<uint32_t>(256, cdiv(n_stiles, kSsWaves));
    AS_SS(false, g_s)
    hipLaunchKernelGGL(stream_guess_kernel, dim3(nA), dim3(256), 0, ctx->stream, smin.p, K, guess.p);
    if (filter) {
      ProfScope pf(ctx, "search_stream_filter");
      constexpr bool kSearch = std::is_same<Policy, SearchScorePolicy>::value;
      const void *q_img = filter->PQ8;
      if (!q_img) {
        AS_TRY(rotated_image_pack_async(ctx, A, filter->a_norms, a_ids, nA, kSsD, filter->Qd, pq8.p, 1, q_err.p));
        q_img = pq8.p;
      }
      hipLaunchKernelGGL(stream_filter_i8_kernel, dim3(256), dim3(256), 0, ctx->stream, (const ssf_i32x4 *)filter->P8, nB,
                         (const ssf_i32x4 *)q_img, nA, (const double *)guess.p, kSearch ? 0.0 : 1.0, kSearch ? filter->a_lams : nullptr,
                         kSearch ? filter->alpha : 1.0, std::max(filter->err_items, filter->err_a), (const unsigned int *)q_err.p,
                         kSearch ? filter->x_lams : nullptr, g_cap, g_count, g_pairs.p, g_over);
      const size_t lds_p = (size_t)nA * 129 * sizeof(double);
      hipLaunchKernelGGL(stream_pairs_kernel<Policy>, dim3(256), dim3(256), lds_p, ctx->stream, A, a_ids, nA, B, filter->perm, pol,
                         (const uint32_t *)g_pairs.p, (const uint32_t *)g_count, (const int *)g_over, (const double *)guess.p, cap, cnt.p,
                         lidx.p, lkey.p, err_flag);
    }
    if (quick) {
      // (its own group: the one launch that streams all of B -- the HBM-bound kernel bench.py prices against N (8 D + 8) bytes)
      const std::string list_name = std::string(prof_name) + "_list";
      ProfScope pl(ctx, list_name.c_str());
      const double margin = 1e-12;
#define AS_QF(NQV)                                                                                                                  \
  hipLaunchKernelGGL((stream_filter_f64_kernel<Policy, NQV>), dim3(NQV == 1 ? 1024 : (NQV == 2 ? 768 : (NQV == 4 ? 512 : 256))), dim3(256), 0,    \
                     ctx->stream, A, a_ids, a_begin, nA, B, nB, pol,                                                                  \
                     (const double *)guess.p, margin, g_cap, g_count, g_pairs.p, g_over)
      if (nA == 1) AS_QF(1);
      else if (nA == 2) AS_QF(2);
      else if (nA <= 4) AS_QF(4);
      else AS_QF(8);
#undef AS_QF
    }
    if (quick) {
      const size_t lds_p = (size_t)nA * 129 * sizeof(double);
      hipLaunchKernelGGL(stream_pairs_kernel<Policy>, dim3(256), dim3(256), lds_p, ctx->stream, A, a_ids, nA, B, (const uint32_t *)nullptr, pol,
                         (const uint32_t *)g_pairs.p, (const uint32_t *)g_count, (const int *)g_over, (const double *)guess.p, cap, cnt.p,
                         lidx.p, lkey.p, err_flag);
    }
    {
      // (the exact list pass: everything when there is no filter in front, else only when a filter's pair list ran out of room)
      const std::string list_name = std::string(prof_name) + (quick ? "_list_exact" : "_list");
      ProfScope pl(ctx, list_name.c_str());
      AS_SS(true, 256u)
    }
    hipLaunchKernelGGL(stream_final_kernel, dim3(nA), dim3(256), 0, ctx->stream, cnt.p, lidx.p, lkey.p, cap, K, out_ld, 0u, out_key,
                       out_idx, over.p, guess.p, done.p);
    AS_HIP(ctx, hipGetLastError());
    // rows whose list ran over: two more rounds of the list pass with the cut their truncated list gives (one more sweep of
    // B each: 0.3 ms; the exact scan they would otherwise fall to walks B with ONE wave per row: 300 ms at 1 M x 128)
    for (int round = 0; round < 2; ++round) {
      AS_HIP(ctx, copy_d2h(ctx, h.data(), over.p, sizeof(int) * nA));
      AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
      bool any = false;
      for (uint32_t a = 0; a < nA; ++a) any = any || h[a] != 0;
      if (!any) break;
      AS_HIP(ctx, hipMemsetAsync(cnt.p, 0, sizeof(uint32_t) * nA, ctx->stream));
      first_list = false;
      AS_SS(true, 256u)
      hipLaunchKernelGGL(stream_final_kernel, dim3(nA), dim3(256), 0, ctx->stream, cnt.p, lidx.p, lkey.p, cap, K, out_ld, 0u, out_key,
                         out_idx, over.p, guess.p, done.p);
      AS_HIP(ctx, hipGetLastError());
      ++retries;
    }
#undef AS_SS1
#undef AS_SS
  }
  AS_HIP(ctx, copy_d2h(ctx, h.data(), over.p, sizeof(int) * (nA + 2)));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const uint32_t h_count = (uint32_t)h[nA];
  const int h_gover = h[nA + 1];
  redo_pos->clear();
  for (uint32_t a = 0; a < nA; ++a)
    if (h[a]) redo_pos->push_back(a);
  ctx->stats["stream_scan_filter"] = filter ? 1 : 0;
  ctx->stats["stream_scan_quick"] = quick ? 1 : 0;
  ctx->stats["stream_scan_filter_items"] = (int64_t)h_count;
  ctx->stats["stream_scan_filter_overflow"] = h_gover;
  ctx->stats["stream_scan_retries"] = retries;
  ctx->stats["stream_scan_rows"] = nA;
  ctx->stats["stream_scan_overflow_rows"] = (int64_t)redo_pos->size();
  return AS_OK;
}

// ---- rows of 256 .. 1024 values: sample, guess, list and final for up to 32 A rows, four at a time -------------------------------
bool stream_quick_wide_applies(uint64_t nA, uint64_t nB, uint64_t d, uint32_t K) {
  return nA >= 1 && nA <= kSsMaxA && d > kSsD && d <= 1024 && d % 128 == 0 && K >= 1 && K <= 64 && nB >= 65536 && nB < (1ull << 27);
}
template <class Policy>
static int stream_quick_wide(as_ctx *ctx, const char *prof_name, const double *A, const uint32_t *a_ids, uint32_t nA, const double *B, uint32_t nB,
                             uint32_t d, uint32_t K, Policy pol, double *out_key, uint32_t *out_idx, int *err_flag,
                             std::vector<uint32_t> *redo_pos) {
  const uint32_t P = d / 128, cap = kSsSortMax, g_cap = 1u << 20;
  const double margin = 1e-12;
  DevBuf<double> guess, lkey;
  DevBuf<unsigned long long> smin;
  DevBuf<uint32_t> lidx, small, g_pairs;
  AS_TRY(smin.alloc(ctx, (size_t)4 * kSsShares));
  AS_TRY(guess.alloc(ctx, 4));
  AS_TRY(lidx.alloc(ctx, (size_t)4 * cap));
  AS_TRY(lkey.alloc(ctx, (size_t)4 * cap));
  AS_TRY(small.alloc(ctx, 16));  // cnt[4] | over[4] | done[4] | g_count | g_over
  AS_TRY(g_pairs.alloc(ctx, g_cap));
  redo_pos->clear();
  ProfScope ps(ctx, prof_name);
  for (uint32_t a0 = 0; a0 < nA; a0 += 4) {
    const uint32_t na = std::min<uint32_t>(4, nA - a0);
    const uint32_t *ids = a_ids ? a_ids + a0 : nullptr;
    uint32_t *cnt = small.p;
    int *over = reinterpret_cast<int *>(small.p + 4), *done = reinterpret_cast<int *>(small.p + 8);
    uint32_t *g_count = small.p + 12;
    int *g_over = reinterpret_cast<int *>(small.p + 13);
    AS_HIP(ctx, hipMemsetAsync(small.p, 0, sizeof(uint32_t) * 16, ctx->stream));
    hipLaunchKernelGGL(fill_u64_kernel, dim3(cdiv((uint64_t)4 * kSsShares, 256)), dim3(256), 0, ctx->stream, smin.p, (uint64_t)4 * kSsShares,
                       0xFFF0000000000000ull);
#define AS_QW(NQV, SAMPLEV, GRID)                                                                                                         \
  {                                                                                                                                        \
    const size_t lds_q = (size_t)NQV * P * 1024;                                                                                           \
    AS_HIP(ctx, hipFuncSetAttribute((const void *)stream_quick_wide_kernel<Policy, NQV, SAMPLEV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)lds_q));                                                                                          \
    hipLaunchKernelGGL((stream_quick_wide_kernel<Policy, NQV, SAMPLEV>), dim3(GRID), dim3(256), lds_q, ctx->stream, A, ids, a0, na, B, nB, P, \
                       pol, (const double *)guess.p, 2.0 * margin, smin.p, g_cap, g_count, g_pairs.p, g_over);                              \
  }
#define AS_QWN(SAMPLEV, GRID)              \
  if (na == 1) AS_QW(1, SAMPLEV, GRID)     \
  else if (na == 2) AS_QW(2, SAMPLEV, GRID) \
  else AS_QW(4, SAMPLEV, GRID)
    AS_QWN(true, 256u)
    hipLaunchKernelGGL(stream_guess_kernel, dim3(na), dim3(256), 0, ctx->stream, smin.p, K, guess.p);
    {
      const std::string list_name = std::string(prof_name) + "_list";
      ProfScope pl(ctx, list_name.c_str());
      AS_QWN(false, 512u)
    }
#undef AS_QWN
#undef AS_QW
    const size_t lds_p = (size_t)na * (d + 1) * sizeof(double);
    AS_HIP(ctx, hipFuncSetAttribute((const void *)stream_pairs_kernel<Policy>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
    hipLaunchKernelGGL(stream_pairs_kernel<Policy>, dim3(256), dim3(256), lds_p, ctx->stream, A, ids, na, B, (const uint32_t *)nullptr, pol,
                       (const uint32_t *)g_pairs.p, (const uint32_t *)g_count, (const int *)g_over, (const double *)guess.p, cap, cnt, lidx.p,
                       lkey.p, err_flag, d, a0, margin);
    hipLaunchKernelGGL(stream_final_kernel, dim3(na), dim3(256), 0, ctx->stream, cnt, lidx.p, lkey.p, cap, K, K, 0u, out_key + (size_t)a0 * K,
                       out_idx + (size_t)a0 * K, over, guess.p, done);
    AS_HIP(ctx, hipGetLastError());
    int h[16];
    AS_HIP(ctx, copy_d2h(ctx, h, small.p, sizeof(h)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (uint32_t a = 0; a < na; ++a)
      if (h[4 + a] || h[13]) redo_pos->push_back(a0 + a);  // list ran over / the pair queue did: the caller's older path
  }
  ctx->stats["stream_scan_quick"] = 1;
  ctx->stats["stream_scan_rows"] = nA;
  ctx->stats["stream_scan_overflow_rows"] = (int64_t)redo_pos->size();
  return AS_OK;
}
int stream_quick_wide_search(as_ctx *ctx, const double *queries, uint32_t nq, const double *items, uint32_t n, uint32_t d, uint32_t K,
                             const SearchScorePolicy &pol, double *out_key, uint32_t *out_idx, int *err_flag, std::vector<uint32_t> *redo,
                             const uint32_t *q_ids) {
  // (q_ids: the listed queries of a batch -- what a large batch's threshold scan left open -- results in list order)
  return stream_quick_wide<SearchScorePolicy>(ctx, "search_stream_scan", queries, q_ids, nq, items, n, d, K, pol, out_key, out_idx, err_flag, redo);
}
int stream_quick_wide_graph(as_ctx *ctx, const double *rows, const uint32_t *row_ids, uint32_t n_rows, uint32_t n, uint32_t d, uint32_t K,
                            const GraphDistPolicy &pol, double *out_key, uint32_t *out_idx, int *err_flag, std::vector<uint32_t> *redo) {
  return stream_quick_wide<GraphDistPolicy>(ctx, "knn_stream_scan", rows, row_ids, n_rows, rows, n, d, K, pol, out_key, out_idx, err_flag, redo);
}

int stream_scan_search(as_ctx *ctx, const double *queries, const uint32_t *q_ids, uint32_t nq, const double *items, uint32_t n, uint32_t K,
                       const SearchScorePolicy &pol, double *out_key, uint32_t *out_idx, int *err_flag, std::vector<uint32_t> *redo,
                       const StreamFilter *filter) {
  return stream_scan<SearchScorePolicy>(ctx, "search_stream_scan", queries, q_ids, 0, nq, items, n, K, pol, out_key, out_idx, K, err_flag, redo,
                                        filter);
}
int stream_scan_graph(as_ctx *ctx, const double *rows, const uint32_t *row_ids, uint32_t n_rows, uint32_t n, uint32_t K,
                      const GraphDistPolicy &pol, double *out_key, uint32_t *out_idx, int *err_flag, std::vector<uint32_t> *redo,
                      const StreamFilter *filter) {
  return stream_scan<GraphDistPolicy>(ctx, "knn_stream_scan", rows, row_ids, 0, n_rows, rows, n, K, pol, out_key, out_idx, K, err_flag, redo,
                                      filter);
}

}  // namespace asx
