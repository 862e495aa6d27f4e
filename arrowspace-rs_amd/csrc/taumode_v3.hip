// taumode_v3.hip -- K6/K7/K8 streaming kernel for n_features == 128 (the reference mapping of the
// BASELINE configs: F = D = 128, L is the F x F feature graph).
// Replaces TauMode::select_tau (src/taumode.rs:87-127), TauMode::compute_synthetic_lambda_csr
// (src/taumode.rs:552-660) and their driver loop (src/taumode.rs:230-259) / the batched form of
// ArrowSpace::prepare_query_item (src/core.rs:533-549).
//
// Shape of the work: 1 KB per item arrives from HBM once; per item the kernel needs the median (or
// percentile / mean) of its 128 values and ~|E| edge terms w (x_i - x_j)^2 over the feature graph.
// Per item that is ~2.6 k f64 min/max for the order statistic and ~7 f64 ops per edge, so the
// kernel sits where the HBM stream, the f64 vector pipe and the LDS all run at comparable rates.
//
// Mapping (wave-autonomous, no workgroup barrier in the loop; 8 waves per CU = 2 per SIMD, which is
// what the f64 vector pipe needs to issue at full rate: one wave alone gets half of it):
//   * one wave owns a tile of 16 items; the tile is copied HBM -> LDS by LDS-DMA
//     (global_load_lds_dwordx4: one instruction = one 1 KB row; LDS row pitch 1056 B);
//   * four adjacent lanes share an item (lane = 4 * item + q).  Edge pass: the off-diagonal entries
//     of L are dealt into four interleaved streams, lane q walks stream q; the entries (byte
//     offsets of x_i, x_j, L_ij, w) sit in LDS and are read as broadcasts, x_i and x_j come from
//     the item's LDS row.  Diagonal, denominator and tau use registers: lane q holds features
//     [32 q, 32 q + 32) of its item;
//   * tau: each lane sorts its 32 values with a Batcher network (static register indices); lanes
//     (0,1) and (2,3) merge into two sorted runs of 64 (one half-cleaner exchange over DPP
//     quad_perm + a local bitonic merge); the median of the 128 values then needs only the
//     half-cleaner between the two runs: max_i min(a_i, b_63-i) and min_i max(a_i, b_63-i).
//     Percentile, and median with non-finite values present, finish the merge and pick by rank;
//   * the DMA of the wave's next tile is issued as soon as the tile has been pulled into
//     registers, so it lands while the sorting network runs.
// LDS banks: a ds_read_b64 is served 32 lanes at a time (8 items x 4 streams) and the bank pair of
// x_i in item row r is (4 r + i) mod 32 at this pitch, so the two x reads of a step are conflict-free
// when the four entries of the step have pairwise different i mod 4 and j mod 4.  The host deals the
// entries into steps with that property (an entry's two ends may be swapped: its terms are
// symmetric in x_i, x_j) and pads with entries that point at the always-zero slots of the row.
// Sums are accumulated in a fixed order (stream order per lane, then the quad), so results are
// deterministic; they differ from the oracle's sequential sums by rounding only (1e-12 bar).
#include <cstring>
#include <utility>

#include "common.h"
#include "sortnet.h"

namespace asx {

namespace {

constexpr int kRows = 16;                  // items per wave tile
constexpr int kRowB = 1056;                // LDS row pitch (bytes): 132 doubles, see the bank note below
constexpr int kTileB = kRows * kRowB;      // 16,896 B per wave
constexpr int kZeroSlot = 128;             // doubles 128..131 of every LDS row stay 0.0 (never written by the DMA)
constexpr int kWaves = 8;                  // waves per workgroup (one workgroup per CU)
constexpr int kU = 4;                      // edge steps per software-pipeline block
constexpr size_t kLdsBudget = 160 * 1024;

struct __align__(16) V3Entry {
  uint32_t oi, oj;  // byte offsets of x_i, x_j inside an item row
  double lij;
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ bool finite_bits(double x) {
  return (((uint32_t)__double2hiint(x) >> 20) & 0x7FFu) != 0x7FFu;
}

static_assert(kU == 4, "the asm helpers below are written for four steps per block");
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(size_t)(lds_void *)p; }

// Asynchronous LDS reads: the destination registers are NOT valid until lds_landed() on them.
__device__ __forceinline__ void read_entries(u32x4 (&e)[kU], uint32_t addr) {
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:128\n\t"
               "ds_read_b128 %3, %4 offset:192"
               : "=&v"(e[0]), "=&v"(e[1]), "=&v"(e[2]), "=&v"(e[3])
               : "v"(addr));
}
// Same reads, issued inside the pipeline: the x values of the block about to be computed pass
// through the statement, so none of that block's arithmetic can be scheduled above the issue.
__device__ __forceinline__ void read_entries_before(u32x4 (&e)[kU], uint32_t addr, double (&xi)[kU], double (&xj)[kU]) {
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:128\n\t"
               "ds_read_b128 %3, %4 offset:192"
               : "=&v"(e[0]), "=&v"(e[1]), "=&v"(e[2]), "=&v"(e[3])
               : "v"(addr), "v"(xi[0])
               :);
  asm volatile("" : "+v"(xi[0]), "+v"(xi[1]), "+v"(xi[2]), "+v"(xi[3]), "+v"(xj[0]), "+v"(xj[1]), "+v"(xj[2]), "+v"(xj[3]));
}
__device__ __forceinline__ void read_x(double (&xi)[kU], double (&xj)[kU], const u32x4 (&e)[kU], uint32_t row_a) {
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    asm volatile("ds_read_b64 %0, %1" : "=v"(xi[u]) : "v"(row_a + e[u].x));
    asm volatile("ds_read_b64 %0, %1" : "=v"(xj[u]) : "v"(row_a + e[u].y));
  }
}
// s_waitcnt lgkmcnt(0) that the register allocator sees as the definition of the given registers:
// nothing that reads them can be scheduled above it.
__device__ __forceinline__ void lds_landed(u32x4 (&a)[kU], u32x4 (&b)[kU]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
}
// The accumulators ride along so that the arithmetic of the block before stays above the wait (it
// is what covers the latency of the reads being waited for).
__device__ __forceinline__ void lds_landed(double (&xi)[kU], double (&xj)[kU], u32x4 (&e)[kU], double &a0, double &a1,
                                           double &a2) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(xi[0]), "+v"(xi[1]), "+v"(xi[2]), "+v"(xi[3]), "+v"(xj[0]), "+v"(xj[1]), "+v"(xj[2]), "+v"(xj[3]),
                 "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(a0), "+v"(a1), "+v"(a2));
}

// value held by another lane of the quad: DPP quad_perm, CTRL = p0 | p1 << 2 | p2 << 4 | p3 << 6
template <int CTRL>
__device__ __forceinline__ double quad_get(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ uint32_t quad_get_u32(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, true);
}
constexpr int kXor1 = 0xB1;  // [1,0,3,2]
constexpr int kXor2 = 0x4E;  // [2,3,0,1]
constexpr int kXor3 = 0x1B;  // [3,2,1,0]

template <typename T>
__device__ __forceinline__ T quad_sum(T v) {
  if constexpr (sizeof(T) == 8) {
    v = v + quad_get<kXor1>(v);
    v = v + quad_get<kXor2>(v);
  } else {
    v = v + quad_get_u32<kXor1>(v);
    v = v + quad_get_u32<kXor2>(v);
  }
  return v;
}

template <int NP, size_t... Is>
__device__ __forceinline__ void sortnet_run(double (&v)[NP], std::index_sequence<Is...>) {
  auto ce = [&](int a, int b) {
    const double lo = __builtin_fmin(v[a], v[b]);
    const double hi = __builtin_fmax(v[a], v[b]);
    v[a] = lo;
    v[b] = hi;
  };
  (ce(SortNet<NP>::kA[Is], SortNet<NP>::kB[Is]), ...);
}

// sorts a bitonic sequence of 32 ascending (5 half-cleaner stages, 80 compare-exchanges); all
// register indices are compile-time constants
template <int J, size_t... Is>
__device__ __forceinline__ void bitonic_stage(double (&w)[32], std::index_sequence<Is...>) {
  auto ce = [&](int a) {
    const double lo = __builtin_fmin(w[a], w[a + J]);
    const double hi = __builtin_fmax(w[a], w[a + J]);
    w[a] = lo;
    w[a + J] = hi;
  };
  (ce((int)((Is / J) * 2 * J + (Is % J))), ...);
}
__device__ __forceinline__ void bitonic_merge32(double (&w)[32]) {
  using S = std::make_index_sequence<16>;
  bitonic_stage<16>(w, S{});
  bitonic_stage<8>(w, S{});
  bitonic_stage<4>(w, S{});
  bitonic_stage<2>(w, S{});
  bitonic_stage<1>(w, S{});
}

// Half-cleaner between this lane's run and the partner lane's run (partner = lane ^ X within the
// quad, X given by CTRL), in place: position k meets the partner's position 31-k.  keep_hi lanes
// keep the larger element of every pair.  Positions k and 31-k are exchanged together so that
// both lanes still see the other's old values.
template <int CTRL>
__device__ __forceinline__ void cross_clean(double (&v)[32], bool keep_hi) {
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const double r1 = quad_get<CTRL>(v[31 - k]), r2 = quad_get<CTRL>(v[k]);
    const double lo1 = __builtin_fmin(v[k], r1), hi1 = __builtin_fmax(v[k], r1);
    const double lo2 = __builtin_fmin(v[31 - k], r2), hi2 = __builtin_fmax(v[31 - k], r2);
    v[k] = keep_hi ? hi1 : lo1;
    v[31 - k] = keep_hi ? hi2 : lo2;
  }
}
// Same-position compare-exchange with the partner lane (second stage of a 64-wide bitonic merge
// that spans two lanes).
template <int CTRL>
__device__ __forceinline__ void cross_same(double (&v)[32], bool keep_hi) {
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const double r = quad_get<CTRL>(v[k]);
    const double lo = __builtin_fmin(v[k], r), hi = __builtin_fmax(v[k], r);
    v[k] = keep_hi ? hi : lo;
  }
}

// Elements of rank i1 and i2 (0..127, per lane) of the quad's 128 values; w = this lane's quarter
// of the sorted sequence (lane q: ranks 32 q .. 32 q + 31).  The values pass through an empty asm
// so that the chain of selects stays a chain of v_cndmask on registers: left alone, the optimiser
// rewrites it into one load from a selected ADDRESS, which forces the whole array out of registers
// into scratch.
__device__ __forceinline__ void pick_ranks(const double (&w)[32], uint32_t i1, uint32_t i2, double &r1,
                                           double &r2) {
  const uint32_t p1 = i1 & 31u, p2 = i2 & 31u;
  double a = w[0], b = w[0];
#pragma unroll
  for (int f = 1; f < 32; ++f) {
    double x = w[f];
    asm volatile("" : "+v"(x));
    a = p1 == (uint32_t)f ? x : a;
    b = p2 == (uint32_t)f ? x : b;
  }
  // a / b are right in the lane that owns the rank: fetch from lane (i >> 5) of the quad
  const double a0 = quad_get<0x00>(a), a1 = quad_get<0x55>(a), a2 = quad_get<0xAA>(a), a3 = quad_get<0xFF>(a);
  const double b0 = quad_get<0x00>(b), b1 = quad_get<0x55>(b), b2 = quad_get<0xAA>(b), b3 = quad_get<0xFF>(b);
  const uint32_t o1 = i1 >> 5, o2 = i2 >> 5;
  r1 = o1 == 0 ? a0 : o1 == 1 ? a1 : o1 == 2 ? a2 : a3;
  r2 = o2 == 0 ? b0 : o2 == 1 ? b1 : o2 == 2 ? b2 : b3;
}

// LDS-DMA of one tile: row r (1 KB, contiguous in HBM) -> LDS at tile + r * kRowB.  Written as
// asm so that the global base and the LDS address stay scalar (SGPR base + 13-bit immediate, one
// s_mov to M0 per row); the builtin form costs a 64-bit vector add and a readfirstlane per row.
// The instruction offset is added to the LDS address as well as to the global one (measured:
// scripts/micro/dma_offset_probe.hip), so M0 carries the row's LDS address minus that offset.
// M0 is used by nothing else in this kernel (gfx950 DS instructions do not read it).
template <int R>
__device__ __forceinline__ void dma_row(const char *gbase, uint32_t lds_tile, uint32_t voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3"
               :
               : "s"(lds_tile + (R * kRowB - (R % 4) * 1024)), "v"(voff), "s"(gbase + (R / 4) * 4096), "n"((R % 4) * 1024)
               : "memory", "m0");
}
template <size_t... Rs>
__device__ __forceinline__ void issue_rows(const char *gbase, uint32_t lds_tile, uint32_t voff, uint32_t rows_here,
                                           std::index_sequence<Rs...>) {
  if (rows_here == kRows) {
    (dma_row<(int)Rs>(gbase, lds_tile, voff), ...);
  } else {
    ((Rs < rows_here ? dma_row<(int)Rs>(gbase, lds_tile, voff) : (void)0), ...);
  }
}
__device__ __forceinline__ void issue_tile(const double *items, uint64_t item0, uint32_t rows_here,
                                           uint32_t lds_tile, int lane) {
  const char *gbase = (const char *)(items + item0 * 128);
  issue_rows(gbase, lds_tile, (uint32_t)lane * 16u, rows_here, std::make_index_sequence<kRows>{});
}

// Requires every off-diagonal entry of L to be <= 0 (a Laplacian): w = max(-L_ij, 0) = -L_ij.
__global__ __launch_bounds__(kWaves * 64) void lambda_kernel_v3(
    const double *__restrict__ items, uint64_t n_items, const V3Entry *__restrict__ g_ent,
    const double *__restrict__ g_diag, uint32_t n_steps, double scale, int mode,
    double arg, int check_finite, double *__restrict__ out, int *__restrict__ err_flag,
    uint32_t *__restrict__ redo_cnt, uint32_t *__restrict__ redo_list, unsigned long long *__restrict__ dbg, int stagger) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, nw = blockDim.x >> 6;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: tile addresses stay in SGPRs
  const int q = lane & 3;
  unsigned char *tile = smem + (size_t)wv * kTileB;
  V3Entry *s_ent = (V3Entry *)(smem + (size_t)nw * kTileB);
  const uint32_t n_slots = 4 * (n_steps + 2 * kU);  // two padding blocks past the end for the prefetch
  double *s_diag = (double *)(s_ent + n_slots);

  const uint64_t n_tiles = (n_items + kRows - 1) / kRows;
  // Tiles of this workgroup: blockIdx.x + gridDim.x * k, k = 0, 1, ...; the waves draw k from a
  // counter in LDS.  (With a fixed share per wave the older wave of each SIMD pair wins the issue
  // arbitration, finishes early and leaves its partner to run alone at half the issue rate.)
  uint32_t *s_ctr = (uint32_t *)(s_diag + 128);
  if (tid == 0) *s_ctr = (uint32_t)nw;
  uint64_t t = (uint64_t)blockIdx.x + (uint64_t)gridDim.x * wv;
  const uint32_t lds_tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)(lds_void *)tile);
  if (t < n_tiles) issue_tile(items, t * kRows, (uint32_t)min((uint64_t)kRows, n_items - t * kRows), lds_tile, lane);

  for (uint32_t i = tid; i < n_slots; i += blockDim.x) s_ent[i] = g_ent[i];
  for (uint32_t i = tid; i < 128; i += blockDim.x) s_diag[i] = g_diag[i];
  __syncthreads();

  const unsigned char *row = tile + (lane >> 2) * kRowB;
  ((double *)(tile + (lane >> 2) * kRowB))[kZeroSlot + q] = 0.0;  // the row's four zero slots (padding entries read them)
  const double *dg = s_diag + q * 32;

  unsigned long long c_wait = 0, c_edge = 0, c_regs = 0, c_tau = 0, c_fin = 0, stamp = 0;
#define V3_STAMP(acc)                                           \
  if (dbg) {                                                    \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    acc += now_ - stamp;                                        \
    stamp = now_;                                               \
  }
  unsigned long long t_begin = 0, r_begin = 0;
  if (dbg) {
    stamp = t_begin = __builtin_amdgcn_s_memtime();
    r_begin = __builtin_amdgcn_s_memrealtime();
  }
  // Stagger: the second half of the waves (the SIMD partners of the first half) starts about half
  // a tile late, so that one partner is in its LDS-heavy edge pass while the other sorts in
  // registers.  Measured at 1 M items (Median): 0.41 ms in lockstep, 0.35 ms staggered.
  if (stagger && wv >= nw / 2) __builtin_amdgcn_s_sleep(127);
  while (t < n_tiles) {
    const uint64_t item0 = t * kRows;
    const uint32_t rows_here = (uint32_t)min((uint64_t)kRows, n_items - item0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tile has landed in LDS
    V3_STAMP(c_wait)

    // ---- edge pass (src/taumode.rs:565-588, 611-639): lane q walks stream q ----
    // Three-stage software pipeline, kU steps per stage: the entries of block b+2 and the x values
    // of block b+1 are in flight while block b is computed.  The LDS reads are issued as asm
    // (left to the optimiser, each prefetch sinks into the iteration that consumes it and the loop
    // pays two LDS round trips per block); lds_landed() is the s_waitcnt that hands the registers
    // back to compiled code.  The three register sets rotate by unrolling, not by copies.
    double numo = 0.0, eh = 0.0, s2 = 0.0;
    {
      const uint32_t row_a = lds_addr(row);
      uint32_t ent_a = lds_addr(s_ent) + (uint32_t)q * 16u;
      u32x4 E[3][kU];
      double XI[3][kU], XJ[3][kU];
      read_entries(E[0], ent_a);
      read_entries(E[1], ent_a + kU * 64);
      ent_a += 2 * kU * 64;
      lds_landed(E[0], E[1]);
      read_x(XI[0], XJ[0], E[0], row_a);
      for (uint32_t s = 0; s < n_steps; s += 3 * kU) {
#pragma unroll
        for (int ph = 0; ph < 3; ++ph) {
          constexpr int kNext[3] = {1, 2, 0};
          const int b0 = ph, b1 = kNext[ph], b2 = kNext[kNext[ph]];
          lds_landed(XI[b0], XJ[b0], E[b1], numo, eh, s2);  // x of this block, entries of the next
          read_x(XI[b1], XJ[b1], E[b1], row_a);
          read_entries_before(E[b2], ent_a, XI[b0], XJ[b0]);
          ent_a += kU * 64;
#pragma unroll
          for (int u = 0; u < kU; ++u) {
            const double lij = __hiloint2double((int)E[b0][u].w, (int)E[b0][u].z);
            const double p = XI[b0][u] * XJ[b0][u];
            numo = __builtin_fma(lij, p, numo);  // L_ij * (x_i x_j)
            const double dd = XI[b0][u] - XJ[b0][u];
            // w = -L_ij: the term is accumulated negated (no negation per entry) and the sign is
            // restored once at the end; tt^2 does not see the sign
            const double tt = (lij * dd) * dd;  // -(w * d * d)
            eh = eh + tt;
            s2 = __builtin_fma(tt, tt, s2);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the two prefetched padding blocks
    }
    if (dbg) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    V3_STAMP(c_edge)

    // ---- pull this lane's 32 features into registers; diagonal term and denominator ----
    double v[32];
    {
      const double *part = (const double *)(row + q * 256);
#pragma unroll
      for (int k = 0; k < 32; ++k) v[k] = part[k];
    }
    double den0 = 0.0, den1 = 0.0, nd0 = 0.0, nd1 = 0.0;
#pragma unroll
    for (int k = 0; k < 32; k += 2) {
      const double a = v[k], b = v[k + 1];
      const double a2 = a * a, b2 = b * b;
      den0 = den0 + a2;
      den1 = den1 + b2;
      nd0 = __builtin_fma(dg[k], a2, nd0);  // L_ii x_i^2
      nd1 = __builtin_fma(dg[k + 1], b2, nd1);
    }
    // A non-finite value makes the sum of squares non-finite; so does a finite value beyond 1e154,
    // hence the exact per-value test behind this cheap one.
    bool fin = finite_bits(den0 + den1);
    if (__any(!fin)) {
      fin = true;
#pragma unroll
      for (int k = 0; k < 32; ++k) fin = fin && finite_bits(v[k]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every LDS read of this tile has returned
    uint64_t tn;
    {
      uint32_t kn = 0;
      if (lane == 0) kn = atomicAdd(s_ctr, 1u);
      tn = (uint64_t)blockIdx.x + (uint64_t)gridDim.x * (uint32_t)__builtin_amdgcn_readfirstlane((int)kn);
      if (tn < n_tiles) issue_tile(items, tn * kRows, (uint32_t)min((uint64_t)kRows, n_items - tn * kRows), lds_tile, lane);
    }
    V3_STAMP(c_regs)

    const bool valid = (uint32_t)(lane >> 2) < rows_here;
    if (check_finite && __any(valid && !fin)) {
      if (lane == 0) atomicOr(err_flag, 1);
    }

    // ---- tau from the item's own values (src/taumode.rs:87-127, 233-234) ----
    double tau;
    if (mode == AS_TAU_FIXED) {
      tau = (finite_bits(arg) && arg > 0.0) ? arg : kTauFloor;
    } else if (mode == AS_TAU_MEAN) {
      double sum = 0.0;
      uint32_t m = 0;
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const bool f = finite_bits(v[k]);
        sum = sum + (f ? v[k] : 0.0);
        m += f ? 1u : 0u;
      }
      sum = quad_sum(sum);
      m = quad_sum(m);
      tau = fmax(m > 0 ? sum / (double)m : 0.0, kTauFloor);
    } else {
      const bool fast = mode == AS_TAU_MEDIAN && __all(fin);
      uint32_t m = 0;
      if (!fast) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          const bool f = finite_bits(v[k]);
          m += f ? 1u : 0u;
          v[k] = f ? v[k] : __builtin_inf();  // non-finite values are not ranked (sorted past every finite one)
        }
      }
      sortnet_run<32>(v, std::make_index_sequence<SortNet<32>::kCount>{});
      // lanes (0,1) and (2,3): two sorted runs of 64, lower half in the even lane
      cross_clean<kXor1>(v, (q & 1) != 0);
      bitonic_merge32(v);
      if (fast) {
        // a = run of lanes (0,1), b = run of lanes (2,3): the lower 64 of the 128 are
        // { min(a_i, b_63-i) }.  a_i sits in lane 0/1 at position i mod 32, b_63-i in lane 3/2 at
        // position 31 - (i mod 32): lane q meets lane 3-q; each lane takes 16 of the 32 pairs.
        double mlo = -__builtin_inf(), mhi = __builtin_inf();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const double r = quad_get<kXor3>(v[31 - k]);
          const double a = v[k];
          mlo = __builtin_fmax(mlo, __builtin_fmin(a, r));
          mhi = __builtin_fmin(mhi, __builtin_fmax(a, r));
        }
        mlo = __builtin_fmax(mlo, quad_get<kXor1>(mlo));
        mhi = __builtin_fmin(mhi, quad_get<kXor1>(mhi));
        mlo = __builtin_fmax(mlo, quad_get<kXor2>(mlo));
        mhi = __builtin_fmin(mhi, quad_get<kXor2>(mhi));
        tau = fmax(0.5 * (mlo + mhi), kTauFloor);
      } else {
        m = quad_sum(m);
        // finish the merge: half-cleaner between the two runs (lanes 2,3 keep the larger
        // elements; their run comes out in reversed position order, which is still bitonic),
        // then a 64-wide bitonic merge inside each lane pair
        cross_clean<kXor3>(v, q >= 2);
        cross_same<kXor1>(v, (q & 1) != 0);
        bitonic_merge32(v);  // lane q now holds ranks 32 q .. 32 q + 31
        // ranks wanted (src/taumode.rs:109-124); m == 0 is resolved after the picks
        uint32_t i1, i2;
        if (mode == AS_TAU_PERCENTILE) {
          const double pp = arg != arg ? 0.0 : fmin(fmax(arg, 0.0), 1.0);
          i1 = i2 = (uint32_t)round((double)((m > 0 ? m : 1u) - 1) * pp);  // half away from zero
        } else {
          i2 = m / 2;
          i1 = (m & 1u) ? i2 : (i2 > 0 ? i2 - 1 : 0);
        }
        double r1, r2;
        pick_ranks(v, i1, i2, r1, r2);
        const double mid = i1 == i2 ? r1 : 0.5 * (r1 + r2);
        tau = m == 0 ? kTauFloor : fmax(mid, kTauFloor);
      }
    }
    if (dbg) asm volatile("" ::"v"(tau));
    V3_STAMP(c_tau)

    // ---- combine the quad, lambda (src/taumode.rs:596-647) ----
    numo = quad_sum(numo);
    eh = quad_sum(-eh);
    s2 = quad_sum(s2);
    const double den = quad_sum(den0 + den1);
    const double nd = quad_sum(nd0 + nd1);
    const double num = nd + scale * numo;
    const double edge = scale * eh;
    const double s2t = scale * s2;
    const double e_raw = den > 1e-12 ? num / den : 0.0;
    double g = 0.0;
    bool redo = false;
    if (edge > 0.0) {
      const double e2 = edge * edge;
      if (finite_bits(s2t) && finite_bits(e2) && e2 > 0.0 && s2t > 1e-280) g = s2t / e2;  // sum t^2 / edge^2
      else redo = true;  // extreme magnitudes: the literal two-pass form runs in lambda_redo_kernel
    }
    const double g_raw = fmin(fmax(g, 0.0), 1.0);
    const double e_b = e_raw / (e_raw + tau);
    const double lam = tau * e_b + (1.0 - tau) * g_raw;
    if (valid && q == 0) {
      const uint64_t item = item0 + (lane >> 2);
      out[item] = redo ? tau : lam;
      if (redo) redo_list[atomicAdd(redo_cnt, 1u)] = (uint32_t)item;
    }
    V3_STAMP(c_fin)
    t = tn;
  }
  if (dbg && lane == 0) {
    unsigned long long *d = dbg + ((size_t)blockIdx.x * nw + wv) * 8;
    d[0] = c_wait;
    d[1] = c_edge;
    d[2] = c_regs;
    d[3] = c_tau;
    d[4] = c_fin;
    d[5] = __builtin_amdgcn_s_memtime() - t_begin;
    d[6] = __builtin_amdgcn_s_memrealtime() - r_begin;
  }
#undef V3_STAMP
}

struct __align__(16) PackedEntry {
  uint32_t i, j;
  double v;
};

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Items the streaming kernel could not finish with sum t^2 / edge^2 (extreme magnitudes): literal
// two-pass form of src/taumode.rs:565-647, one wave per item; out[item] holds tau on entry.
__global__ __launch_bounds__(256) void lambda_redo_kernel(const double *__restrict__ items, uint32_t F,
                                                          const PackedEntry *__restrict__ ent, uint64_t nnz,
                                                          const uint32_t *__restrict__ redo_cnt,
                                                          const uint32_t *__restrict__ redo_list,
                                                          double *__restrict__ out) {
  const uint32_t cnt = *redo_cnt;
  const int lane = threadIdx.x & 63;
  for (uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6); r < cnt; r += gridDim.x * 4) {
    const uint32_t item = redo_list[r];
    const double *x = items + (uint64_t)item * F;
    const double tau = out[item];
    double num = 0.0, edge = 0.0, den = 0.0;
    for (uint64_t e = lane; e < nnz; e += 64) {
      const PackedEntry p = ent[e];
      const double xi = x[p.i], xj = x[p.j];
      num = num + (xi * p.v) * xj;
      const double w = fmax(-p.v, 0.0);
      if (p.i != p.j && w > 0.0) {
        const double dd = xi - xj;
        edge = edge + (w * dd) * dd;
      }
    }
    for (uint32_t f = lane; f < F; f += 64) den = den + x[f] * x[f];
    num = wave_sum(num);
    edge = wave_sum(edge);
    den = wave_sum(den);
    double g = 0.0;
    if (edge > 0.0) {
      for (uint64_t e = lane; e < nnz; e += 64) {
        const PackedEntry p = ent[e];
        const double w = fmax(-p.v, 0.0);
        if (p.i != p.j && w > 0.0) {
          const double dd = x[p.i] - x[p.j];
          const double share = ((w * dd) * dd) / edge;
          g = g + share * share;
        }
      }
      g = wave_sum(g);
    }
    const double e_raw = den > 1e-12 ? num / den : 0.0;
    const double g_raw = fmin(fmax(g, 0.0), 1.0);
    const double e_b = e_raw / (e_raw + tau);
    if (lane == 0) out[item] = tau * e_b + (1.0 - tau) * g_raw;
  }
}

}  // namespace

// Host-side preparation of the entry streams (L is at most 128 x 128 here: a few KB).
static int prepare_v3(as_ctx *ctx, as_csr *L) {
  if (L->v3_state != 0) return AS_OK;
  L->v3_state = -1;  // unsupported until proven otherwise
  if (L->n > 128 || L->nnz > 128 * 128) return AS_OK;
  std::vector<uint64_t> indptr(L->n + 1), indices(L->nnz);
  std::vector<double> data(L->nnz);
  if (L->slab && (const void *)L->indptr == L->slab && (const char *)L->data > (const char *)L->slab) {
    // the three arrays of an assembled graph share one allocation: one copy and one wait instead of three
    const size_t off_idx = (size_t)((const char *)L->indices - (const char *)L->slab), off_val = (size_t)((const char *)L->data - (const char *)L->slab);
    std::vector<unsigned char> whole(off_val + sizeof(double) * (L->nnz ? L->nnz : 1));
    AS_HIP(ctx, copy_d2h(ctx, whole.data(), L->slab, whole.size()));
    std::memcpy(indptr.data(), whole.data(), sizeof(uint64_t) * (L->n + 1));
    if (L->nnz) {
      std::memcpy(indices.data(), whole.data() + off_idx, sizeof(uint64_t) * L->nnz);
      std::memcpy(data.data(), whole.data() + off_val, sizeof(double) * L->nnz);
    }
  } else {
    AS_HIP(ctx, copy_d2h(ctx, indptr.data(), L->indptr, sizeof(uint64_t) * (L->n + 1)));
    if (L->nnz) {
      AS_HIP(ctx, copy_d2h(ctx, indices.data(), L->indices, sizeof(uint64_t) * L->nnz));
      AS_HIP(ctx, copy_d2h(ctx, data.data(), L->data, sizeof(double) * L->nnz));
    }
  }
  const uint64_t n = L->n;
  std::vector<uint64_t> img(n * n, 0);
  std::vector<uint8_t> has(n * n, 0);
  bool dup = false;
  for (uint64_t i = 0; i < n; ++i)
    for (uint64_t e = indptr[i]; e < indptr[i + 1]; ++e) {
      const uint64_t j = indices[e];
      if (j >= n) return AS_OK;  // malformed for this path: leave it to the generic kernel
      if (has[i * n + j]) dup = true;
      has[i * n + j] = 1;
      std::memcpy(&img[i * n + j], &data[e], 8);
    }
  if (dup) return AS_OK;
  bool sym = true;
  for (uint64_t i = 0; i < n && sym; ++i)
    for (uint64_t j = i + 1; j < n; ++j)
      if (has[i * n + j] != has[j * n + i] || img[i * n + j] != img[j * n + i]) {
        sym = false;
        break;
      }
  std::vector<double> diag(128, 0.0);
  std::vector<V3Entry> ent;
  std::vector<double> w;
  bool negw = true;
  for (uint64_t i = 0; i < n; ++i)
    for (uint64_t e = indptr[i]; e < indptr[i + 1]; ++e) {
      const uint64_t j = indices[e];
      if (j == i) {
        diag[i] = data[e];
        continue;
      }
      if (sym && j < i) continue;
      V3Entry q;
      q.oi = (uint32_t)(i * 8);
      q.oj = (uint32_t)(j * 8);
      q.lij = data[e];
      ent.push_back(q);
      const double ww = -data[e] > 0.0 ? -data[e] : 0.0;  // w = max(-L_ij, 0); NaN -> 0 as f64::max does
      w.push_back(ww);
      if (!(data[e] <= 0.0)) negw = false;
    }
  if (!negw) return AS_OK;  // a positive (or NaN) off-diagonal entry: w = max(-L_ij, 0) is not -L_ij; generic kernels
  // Deal the entries into steps of four with pairwise different (i mod 4) and (j mod 4): see the
  // bank note at the top.  Greedy: for each wanted i-residue take an unused entry (either
  // orientation) whose other end has a j-residue still free in this step, preferring the fullest
  // class; what cannot be filled is padded with an entry between two zero slots.
  struct Cand {
    V3Entry e;
    double w;
  };
  std::vector<Cand> pool[4][4];
  for (size_t k = 0; k < ent.size(); ++k) pool[(ent[k].oi / 8) % 4][(ent[k].oj / 8) % 4].push_back({ent[k], w[k]});
  size_t left = ent.size();
  std::vector<V3Entry> sched;
  std::vector<double> sched_w;
  while (left > 0) {
    bool j_used[4] = {false, false, false, false};
    for (int ri = 0; ri < 4; ++ri) {
      int best_a = -1, best_b = -1;
      size_t best_n = 0;
      for (int rj = 0; rj < 4; ++rj) {
        if (j_used[rj]) continue;
        if (pool[ri][rj].size() > best_n) best_n = pool[ri][rj].size(), best_a = ri, best_b = rj;
        if (pool[rj][ri].size() > best_n) best_n = pool[rj][ri].size(), best_a = rj, best_b = ri;
      }
      Cand c;
      if (best_n > 0) {
        c = pool[best_a][best_b].back();
        pool[best_a][best_b].pop_back();
        --left;
        if (best_a != ri) std::swap(c.e.oi, c.e.oj);  // stored the other way round
        j_used[(c.e.oj / 8) % 4] = true;
      } else {
        int rj = 0;
        while (j_used[rj]) ++rj;
        j_used[rj] = true;
        c.e.oi = (uint32_t)((kZeroSlot + ri) * 8);
        c.e.oj = (uint32_t)((kZeroSlot + rj) * 8);
        c.e.lij = 0.0;
        c.w = 0.0;
      }
      sched.push_back(c.e);
      sched_w.push_back(c.w);
    }
  }
  const uint32_t n_steps = (uint32_t)((sched.size() / 4 + 3 * kU - 1) / (3 * kU) * (3 * kU));  // whole pipeline rounds
  const size_t n_slots = 4 * ((size_t)n_steps + 2 * kU);
  for (size_t k = sched.size(); k < n_slots; ++k) {  // whole padding steps keep the residue pattern
    V3Entry z;
    z.oi = z.oj = (uint32_t)((kZeroSlot + k % 4) * 8);
    z.lij = 0.0;
    sched.push_back(z);
    sched_w.push_back(0.0);
  }
  ent.swap(sched);
  w.swap(sched_w);
  const size_t fixed = n_slots * sizeof(V3Entry) + 128 * sizeof(double) + 16;
  if (fixed + kTileB > kLdsBudget) return AS_OK;  // entry table too large for LDS
  AS_HIP(ctx, handle_malloc(&L->v3_ent, n_slots * sizeof(V3Entry)));
  AS_HIP(ctx, handle_malloc(&L->v3_diag, 128 * sizeof(double)));
  AS_HIP(ctx, copy_h2d(ctx, L->v3_ent, ent.data(), n_slots * sizeof(V3Entry)));
  AS_HIP(ctx, copy_h2d(ctx, L->v3_diag, diag.data(), 128 * sizeof(double)));
  L->v3_steps = n_steps;
  L->v3_scale = sym ? 2.0 : 1.0;
  L->v3_state = 1;
  return AS_OK;
}

// Streaming lambda-tau for n_features == 128.  *handled = false when the shape is outside this
// kernel's domain (the caller then uses the generic kernels).
int lambda_v3(as_ctx *ctx, const double *items, uint64_t n_items, uint64_t F, as_csr *L, int mode, double arg,
              bool check_finite, double *out, bool *handled) {
  *handled = false;
  if (F != 128 || n_items == 0 || n_items >= 0xFFFFFFFFull || ((uintptr_t)items & 15) != 0) return AS_OK;
  if (mode < AS_TAU_FIXED || mode > AS_TAU_PERCENTILE) return AS_OK;
  AS_TRY(prepare_v3(ctx, L));
  if (L->v3_state != 1) return AS_OK;
  const size_t n_slots = 4 * ((size_t)L->v3_steps + 2 * kU);
  const size_t fixed = n_slots * sizeof(V3Entry) + 128 * sizeof(double) + 16;
  int nw = (int)std::min<size_t>(kWaves, (kLdsBudget - fixed) / kTileB);
  if (nw < 1) return AS_OK;
  const size_t lds = (size_t)nw * kTileB + fixed;
  ctx->stats["taumode_v3_steps"] = (int64_t)L->v3_steps;  // edge steps per lane and item (four entries each): bench.py's f64 operation count
  const uint64_t n_tiles = (n_items + kRows - 1) / kRows;
  const unsigned blocks = (unsigned)std::min<uint64_t>((n_tiles + nw - 1) / nw, 256);

  static bool attr_set_dev[64] = {};  // allow the full 160 KB of dynamic LDS: once per device, outside the timed region
  bool &attr_set = attr_set_dev[ctx->device & 63];
  if (!attr_set) {
    AS_HIP(ctx, hipFuncSetAttribute((const void *)lambda_kernel_v3, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)kLdsBudget));
    attr_set = true;
  }
  DevBuf<int> err;
  DevBuf<uint32_t> redo;  // [0] = count, [1..] = item ids
  AS_TRY(err.alloc(ctx, 1));
  AS_TRY(redo.alloc(ctx, n_items + 1));
  AS_HIP(ctx, hipMemsetAsync(err.p, 0, sizeof(int), ctx->stream));
  AS_HIP(ctx, hipMemsetAsync(redo.p, 0, sizeof(uint32_t), ctx->stream));
  DevBuf<unsigned long long> dbg;
  const size_t n_dbg = (size_t)blocks * nw * 8;
  if (ctx->opt_debug == 3) {
    AS_TRY(dbg.alloc(ctx, n_dbg));
    AS_HIP(ctx, hipMemsetAsync(dbg.p, 0, n_dbg * 8, ctx->stream));
  }
  {
    ProfScope ps(ctx, "taumode_lambda");
    hipLaunchKernelGGL(lambda_kernel_v3, dim3(blocks), dim3(nw * 64), lds, ctx->stream, items, n_items,
                       (const V3Entry *)L->v3_ent, (const double *)L->v3_diag, L->v3_steps, L->v3_scale, mode, arg,
                       check_finite ? 1 : 0, out, err.p, redo.p, redo.p + 1, dbg.p, (int)(ctx->opt_debug != 4));
    AS_HIP(ctx, hipGetLastError());
    if (L->nnz) {
      hipLaunchKernelGGL(lambda_redo_kernel, dim3(64), dim3(256), 0, ctx->stream, items, (uint32_t)F,
                         (const PackedEntry *)L->packed, L->nnz, redo.p, redo.p + 1, out);
      AS_HIP(ctx, hipGetLastError());
    }
  }
  if (dbg.p) {  // diagnostics: mean cycles per wave and phase
    std::vector<unsigned long long> hd(n_dbg);
    AS_HIP(ctx, copy_d2h(ctx, hd.data(), dbg.p, n_dbg * 8));
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    for (size_t w = 0; w < n_dbg / 8; ++w)
      for (int k = 0; k < 7; ++k) acc[k] += (double)hd[w * 8 + k];
    double life_max = 0, life_min = 1e30;
    for (size_t w = 0; w < n_dbg / 8; ++w) {
      life_max = std::max(life_max, (double)hd[w * 8 + 6]);
      life_min = std::min(life_min, (double)hd[w * 8 + 6]);
    }
    fprintf(stderr, "[v3 wave life] min %.1f us max %.1f us\n", life_min / 100.0, life_max / 100.0);
    const double per = (double)n_tiles;
    fprintf(stderr, "[v3 cycles per tile] wait %.0f edge %.0f regs %.0f tau %.0f final %.0f (memtime ticks; %llu tiles, %u waves; clock %.0f MHz, wave life %.1f us)\n",
            acc[0] / per, acc[1] / per, acc[2] / per, acc[3] / per, acc[4] / per, (unsigned long long)n_tiles,
            (unsigned)(blocks * nw), acc[5] / acc[6] * 100.0, acc[6] / (blocks * nw) / 100.0);
  }
  if (check_finite) {
    int hflag = 0;
    AS_HIP(ctx, copy_d2h(ctx, &hflag, err.p, sizeof(int)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (hflag) return fail(ctx, AS_ERR_NONFINITE, as_status_message(AS_ERR_NONFINITE));
  }
  *handled = true;
  return AS_OK;
}

}  // namespace asx
