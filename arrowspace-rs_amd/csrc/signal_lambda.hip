// signal_lambda.hip -- K7 on the ITEM GRAPH: lambda-tau of length-N signals against an n x n CSR Laplacian with n large
// (SURVEY.md section 8(a'), item-graph mapping; section 8(e) row 6 for the sharded form).
//
// Replaces, for this shape, TauMode::compute_synthetic_lambda_csr (src/taumode.rs:552-660) and the tau it is called with
// (TauMode::select_tau on the signal's own values, src/taumode.rs:87-127, 233-234): the S signals are the COLUMNS of a row-major
// N x S matrix X (the feature columns of the item matrix: "feature row over item nodes"), the graph is the N x N Laplacian
// build_laplacian_matrix returns for the untransposed items (src/energymaps.rs:270).  The reference walks the CSR once per signal
// and reads x[i], x[j] by index (taumode.rs:570-575); here the CSR is streamed ONCE for all S signals and an entry (i, j) fetches the
// two ROWS X[i], X[j] (S values each, 1 KB for S = 128): a sparse matrix times a tall dense matrix, bound by the HBM / gather rate
// (algorithmic bytes nnz * 16 + N * S * 8) and by the f64 vector rate (7 operations per entry and signal).
//
// Layout and mapping
//   - A wave owns one row of X at a time: lane l holds columns 2 l, 2 l + 1 of a 128-column slab (one 16-B load per lane, 1 KB per
//     wave-instruction, whole 128-B lines).  Slabs of 128 columns are independent launches' y index.
//   - The graph is kept as a stream of 16-byte records in the order the rows are visited: per row a header (row, entry count,
//     diagonal value) and one record (column, value) per stored off-diagonal entry.  Records are wave-uniform: they come through the
//     scalar cache (s_load), only the rows of X go through the vector memory pipe.
//   - A bitwise symmetric matrix (every Laplacian this library builds) is streamed as its upper triangle: a pair {i, j} is fetched
//     once and stands for (i, j) and (j, i) -- half the gathers.
//   - Rows are visited in an order in which neighbours sit together (as_csr_set_order, or the order the join found), so that the
//     rows a workgroup gathers are the ones its neighbours on the same XCD have just fetched: the gathers are served by L2 /
//     Infinity Cache and HBM sees each row about once.  Workgroups are dealt to XCDs so that one XCD walks one contiguous eighth
//     of the order.
//   - Sums: a row's terms are added in entry order (as the oracle's row-local sums), rows are added per wave in visiting order,
//     waves per workgroup and workgroups per column in index order, every level a fixed order: bit-identical from run to run.
//     Against the oracle's row-order sum the difference is rounding (about 1e-14 relative at N = 1 M).
//   - G = sum (t / edge)^2 is evaluated as (sum t^2) / edge^2 from the same pass; where that quotient is not safe (overflow /
//     underflow of the squares) a second pass with the literal shares runs (the reference's own two passes).
//   - tau of a signal is an order statistic of N values: a radix descent on order-preserving 64-bit keys (11 bits per pass over X,
//     histograms in LDS) until the bucket of the wanted rank holds at most 4096 values, which are then collected and sorted.
#include <cmath>

#include "common.h"

namespace asx {

namespace {

struct __align__(16) SigRec {
  uint32_t a;  // row of X: header kHdr | (row_begin + local row), entry: its column (global)
  uint32_t b;  // 0
  double v;    // header: the stored diagonal (half of it when the stream is the upper triangle); entry: L_ij; padding: 0
};

constexpr int kSlab = 128;          // columns per slab (2 per lane)
constexpr int kWavesPerBlock = 4;

__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint64_t uni64(uint64_t x) {
  return ((uint64_t)uni((uint32_t)(x >> 32)) << 32) | uni((uint32_t)x);
}
__device__ __forceinline__ uint64_t sort_key(double x) {
  const uint64_t u = (uint64_t)__double_as_longlong(x);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double unsort_key(uint64_t k) {
  const uint64_t u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)u);
}
__device__ __forceinline__ bool finite_bits(double x) { return (((uint64_t)__double_as_longlong(x) >> 52) & 0x7FF) != 0x7FF; }

// ---- the record stream ------------------------------------------------------------------------------------------------------
//
// Geometry.  The rows are visited in `order` (position -> row).  Positions are cut into chunks of CH rows; XCD group x (the
// workgroups with blockIdx % 8 == x) owns the x-th eighth of the chunks, and inside a group chunk cg goes to wave cg % Wx as its
// round cg / Wx.  So at any time the Wx waves of a group work on Wx neighbouring chunks -- a window of Wx * CH positions whose rows
// and gathered neighbours fit the group's L2 -- while each wave owns K = ceil(chunks per group / Wx) chunks in all and writes ONE
// set of partial sums.  The stream stores every wave's records contiguously (slot s = ((x * Wx + wave) * K + round) * CH + r), so
// the kernel walks a plain range; the interleave lives in the slot -> position map below.
struct SigGeom {
  uint32_t n;      // rows of the handle (positions)
  uint32_t ch;     // rows per chunk
  uint32_t wx;     // waves per XCD group (a multiple of 4)
  uint32_t k;      // chunks per wave
  uint32_t c8;     // chunks per XCD group
};
__host__ __device__ inline uint64_t sig_slots(const SigGeom &g) { return (uint64_t)8 * g.wx * g.k * g.ch; }
__device__ __forceinline__ bool sig_slot_pos(const SigGeom &g, uint64_t s, uint32_t *pos) {
  const uint32_t per_wave = g.k * g.ch;
  const uint32_t widx = (uint32_t)(s / per_wave), in = (uint32_t)(s % per_wave);
  const uint32_t round = in / g.ch, r = in % g.ch, x = widx / g.wx, wv = widx % g.wx;
  const uint64_t cg = (uint64_t)round * g.wx + wv;
  if (cg >= g.c8) return false;
  const uint64_t p = ((uint64_t)x * g.c8 + cg) * g.ch + r;
  if (p >= g.n) return false;
  *pos = (uint32_t)p;
  return true;
}

constexpr uint32_t kHdr = 0x80000000u;

// is the matrix bitwise symmetric?  (binary search of the transposed entry; whole matrices only)
__global__ void sig_symmetry_kernel(const uint64_t *__restrict__ indptr, const uint64_t *__restrict__ indices, const double *__restrict__ data,
                                    uint64_t n, int *__restrict__ not_symmetric) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool bad = false;
  for (uint64_t e = indptr[i]; e < indptr[i + 1]; ++e) {
    const uint64_t j = indices[e];
    if (j == i) continue;
    if (j >= n) {
      bad = true;
      continue;
    }
    uint64_t lo = indptr[j], hi = indptr[j + 1];
    bool found = false;
    while (lo < hi) {
      const uint64_t mid = (lo + hi) >> 1;
      const uint64_t c = indices[mid];
      if (c == i) {
        found = __double_as_longlong(data[mid]) == __double_as_longlong(data[e]);
        break;
      }
      if (c < i) lo = mid + 1;
      else hi = mid;
    }
    bad |= !found;
  }
  if (bad) atomicOr(not_symmetric, 1);
}

// records of slot s: 1 + the kept off-diagonal entries of its row (0 for an empty slot)
__global__ void sig_count_kernel(const uint64_t *__restrict__ indptr, const uint64_t *__restrict__ indices, SigGeom g, uint64_t row_begin,
                                 const uint32_t *__restrict__ order, int sym, uint32_t *__restrict__ cnt) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= sig_slots(g)) return;
  uint32_t p;
  if (!sig_slot_pos(g, s, &p)) {
    cnt[s] = 0;
    return;
  }
  const uint64_t r = order ? order[p] : p, gi = row_begin + r;
  uint32_t c = 1;
  for (uint64_t e = indptr[r]; e < indptr[r + 1]; ++e) {
    const uint64_t j = indices[e];
    if (j == gi) continue;
    if (!sym || j > gi) ++c;
  }
  cnt[s] = c;
}

__global__ void sig_fill_kernel(const uint64_t *__restrict__ indptr, const uint64_t *__restrict__ indices, const double *__restrict__ data,
                                SigGeom g, uint64_t row_begin, const uint32_t *__restrict__ order, int sym, const uint64_t *__restrict__ off,
                                const uint32_t *__restrict__ pad_row, SigRec *__restrict__ rec) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= sig_slots(g)) return;
  uint32_t p;
  if (!sig_slot_pos(g, s, &p)) return;
  const uint64_t r = order ? order[p] : p, gi = row_begin + r;
  uint64_t o = off[s];
  SigRec h;
  h.a = kHdr | (uint32_t)gi;  // the row's index in X
  h.b = 0;
  double diag = 0.0;
  uint64_t w = o + 1;
  for (uint64_t e = indptr[r]; e < indptr[r + 1]; ++e) {
    const uint64_t j = indices[e];
    if (j == gi) {
      diag = __dadd_rn(diag, data[e]);
      continue;
    }
    if (!sym || j > gi) {
      SigRec q;
      q.a = (uint32_t)j;
      q.b = 0;
      q.v = data[e];
      rec[w++] = q;
    }
  }
  // upper triangle: every sum of the pass is doubled at the end (a pair stands for two entries), so the header carries half the
  // diagonal (exact: a power of two)
  h.v = sym ? __dmul_rn(0.5, diag) : diag;
  rec[o] = h;
  for (const uint64_t end = off[s + 1]; w < end; ++w) {  // the wave's padding to whole batches: zero entries against its last row
    SigRec q;
    q.a = pad_row[s / ((uint64_t)g.k * g.ch)];
    q.b = 0;
    q.v = 0.0;
    rec[w] = q;
  }
}

// pads every wave's record count to a multiple of 8 (two batches of the pass): the extra records go behind the wave's last row
__global__ void sig_pad_kernel(SigGeom g, uint64_t row_begin, const uint32_t *__restrict__ order, uint32_t *__restrict__ cnt,
                               uint32_t *__restrict__ pad_row) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= 8 * g.wx) return;
  const uint64_t per = (uint64_t)g.k * g.ch, s0 = (uint64_t)w * per;
  uint64_t total = 0, last = ~0ull;
  for (uint64_t s = s0; s < s0 + per; ++s)
    if (cnt[s]) {
      total += cnt[s];
      last = s;
    }
  pad_row[w] = 0;
  if (last == ~0ull) return;
  uint32_t p = 0;
  sig_slot_pos(g, last, &p);
  pad_row[w] = (uint32_t)(row_begin + (order ? order[p] : p));
  cnt[last] += (uint32_t)((8 - total % 8) % 8);
}

__global__ void order_check_kernel(const uint32_t *__restrict__ order, uint64_t n, uint32_t *__restrict__ seen, int *__restrict__ bad) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint32_t r = order[p];
  if (r >= n) {
    atomicOr(bad, 1);
    return;
  }
  if (atomicAdd(&seen[r], 1u) != 0u) atomicOr(bad, 1);
}

// ---- the pass over the graph -------------------------------------------------------------------------------------------------

struct Acc2 {
  double x, y;
};

constexpr int kNQ = 4;  // partial sums per column: 0 num, 1 edge, 2 sum t^2 (each to be doubled when the stream is the upper triangle), 3 den

// One record against the current row: the same arithmetic for an entry (x = x_j) and for a header, which is the diagonal entry
// with x_j = x_i (so d = 0 and no edge term) after the gathered row has become x_i.  PASS2: squared shares only.
template <bool PASS2, bool TWO>
__device__ __forceinline__ void sig_term(const Acc2 &xi, const Acc2 &x, double v, double w, const Acc2 &tot_e, Acc2 &num, Acc2 &edge, Acc2 &s2) {
  if (!PASS2) {
    num.x = fma(__dmul_rn(xi.x, v), x.x, num.x);  // xi * lij * xj
    if (TWO) num.y = fma(__dmul_rn(xi.y, v), x.y, num.y);
  }
  const double dx = __dsub_rn(xi.x, x.x), dy = TWO ? __dsub_rn(xi.y, x.y) : 0.0;
  const double tx = __dmul_rn(__dmul_rn(w, dx), dx), ty = TWO ? __dmul_rn(__dmul_rn(w, dy), dy) : 0.0;  // w * d * d
  if (PASS2) {
    const double sx = __ddiv_rn(tx, tot_e.x);  // share = contrib / edge_energy_sum
    s2.x = fma(sx, sx, s2.x);
    if (TWO) {
      const double sy = __ddiv_rn(ty, tot_e.y);
      s2.y = fma(sy, sy, s2.y);
    }
  } else {
    edge.x = __dadd_rn(edge.x, tx);
    s2.x = fma(tx, tx, s2.x);
    if (TWO) {
      edge.y = __dadd_rn(edge.y, ty);
      s2.y = fma(ty, ty, s2.y);
    }
  }
}

// part: [slab][blocks + 1][kNQ][128]; this kernel writes blocks 0 .. gridDim.x - 1 (PASS2: [slab][blocks][128], the sum of squared
// shares).  A wave walks its records in batches of four with the gathers of the next batch in flight.
//
// Records reach the wave through the VECTOR memory pipe, 64 at a time (lane l holds record base + l: one coalesced 1-KB load),
// each lane turns its record's row into that row's address, and a record is handed to all lanes by four v_readlane (address,
// value).  Scalar loads would be the natural fit for wave-uniform data, but they return out of order, so every use waits for ALL
// of them (lgkmcnt(0)), and the address arithmetic of a record would be five scalar instructions instead of one vector
// instruction per 64 records.  The loop has no bounds checks: the stream pads every wave's records to whole batches with
// entries of value 0 against the wave's last row (they add exactly 0).
// W: how a lane reads its columns.  2: two neighbouring columns in one 16-byte load (every row slice 16-byte aligned: even S,
// aligned base); 0: the same two columns in two 8-byte loads; 1: ONE column per lane, blockIdx.y = 2 * slab + half -- the rows a
// workgroup gathers are 512 bytes, so the neighbourhood an XCD works in is half as large against its L2 (the partials keep
// their layout: a block writes its half of the 128 columns).
template <bool PASS2, int W>
__global__ __launch_bounds__(256) void signal_sums_kernel(const double *__restrict__ X, uint64_t ldx, uint32_t S,
                                                             const SigRec *__restrict__ rec, const uint64_t *__restrict__ off,
                                                             uint32_t slots_per_wave, uint32_t waves_per_group,
                                                             const double *__restrict__ edge_tot, const int *__restrict__ need_pass2,
                                                             double *__restrict__ part) {
  if (PASS2 && *need_pass2 == 0) return;
  constexpr uint32_t NQ = PASS2 ? 1 : kNQ;
  constexpr bool TWO = W != 1;
  __shared__ double s_red[kWavesPerBlock][kNQ][kSlab];
  const int lane = threadIdx.x & 63;
  const uint32_t wv = uni(threadIdx.x >> 6);
  const uint32_t slab = TWO ? blockIdx.y : blockIdx.y / 2, half = TWO ? 0 : blockIdx.y % 2;
  const uint32_t widx = (blockIdx.x % 8) * waves_per_group + (blockIdx.x / 8) * kWavesPerBlock + wv;
  // Lanes past the last column read a valid one (their sums land in the padding of the partials, which nobody reads): no
  // predicate in the loop.
  const uint32_t cc0 = TWO ? 2 * lane : half * 64 + lane;  // this lane's (first) column inside the slab
  const uint32_t col = slab * kSlab + cc0;
  const bool on0 = col < S, on1 = TWO && col + 1 < S;
  const uint32_t c_lo = W == 2 ? min(col, S - 2) : min(col, S - 1), c_hi = min(col + 1, S - 1);
  const uint32_t off_lo = c_lo * 8u, off_hi = c_hi * 8u;  // byte offsets of this lane's columns inside a row
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  auto load_row = [&](uint32_t olo, uint32_t ohi) -> Acc2 {  // the row's byte offset in X (wave-uniform; bit 0 is the header flag)
    Acc2 r;
    const char *src = reinterpret_cast<const char *>(X) + (((uint64_t)ohi << 32) | (olo & ~3u));
    if (W == 2) {
      const dbl2 t = *reinterpret_cast<const dbl2 *>(src + off_lo);
      r.x = t.x;
      r.y = t.y;
    } else {
      r.x = *reinterpret_cast<const double *>(src + off_lo);
      r.y = TWO ? *reinterpret_cast<const double *>(src + off_hi) : 0.0;
    }
    return r;
  };

  Acc2 num = {0.0, 0.0}, edge = {0.0, 0.0}, s2 = {0.0, 0.0}, den = {0.0, 0.0}, xi = {0.0, 0.0};
  Acc2 tot_e = {1.0, 1.0};
  if (PASS2) {
    tot_e.x = on0 ? edge_tot[col] : 1.0;
    tot_e.y = on1 ? edge_tot[col + 1] : 1.0;
  }
  const uint64_t q0 = uni64(off[(uint64_t)widx * slots_per_wave]), q1 = uni64(off[(uint64_t)(widx + 1) * slots_per_wave]);
  const uint32_t nrec = uni((uint32_t)(q1 - q0));  // a multiple of 8
  if (nrec) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const SigRec *rp = rec + q0;
    const uint64_t row_bytes = ldx * 8;
    struct Block {  // 64 records, one per lane: the row's byte offset in X (bit 0: header) and the value
      uint32_t alo, ahi, vlo, vhi;
    };
    auto load_block = [&](uint32_t blk) -> Block {
      const uint32_t idx = min(blk * 64u + (uint32_t)lane, nrec - 1);  // past the end: the last record again (never processed)
      const u32x4 r = *reinterpret_cast<const u32x4 *>(rp + idx);
      // rows are 8-byte multiples: bit 0 of the offset says "header", bit 1 "the value is negative" (w = -v, else w = 0:
      // (-lij).max(0.0), taumode.rs:578; a NaN gives 0 as f64::max does)
      const double rv = __longlong_as_double((long long)(((uint64_t)r.w << 32) | r.z));
      const uint64_t ad = (uint64_t)(r.x & ~kHdr) * row_bytes + (r.x >> 31) + (rv < 0.0 ? 2u : 0u);
      Block b;
      b.alo = (uint32_t)ad, b.ahi = (uint32_t)(ad >> 32), b.vlo = r.z, b.vhi = r.w;
      return b;
    };
    struct Rec {  // one record in scalar registers
      uint32_t alo, ahi;
      double v, w;
    };
    auto read_rec = [&](const Block &b, uint32_t l) -> Rec {
      Rec r;
      const uint32_t alo = (uint32_t)__builtin_amdgcn_readlane((int)b.alo, (int)l), ahi = (uint32_t)__builtin_amdgcn_readlane((int)b.ahi, (int)l);
      const uint32_t vlo = (uint32_t)__builtin_amdgcn_readlane((int)b.vlo, (int)l), vhi = (uint32_t)__builtin_amdgcn_readlane((int)b.vhi, (int)l);
      r.alo = alo, r.ahi = ahi;
      r.v = __longlong_as_double((long long)(((uint64_t)vhi << 32) | vlo));
      const bool neg = (alo & 2u) != 0;  // scalar selects: the vector unit sees w as an operand only
      r.w = __longlong_as_double((long long)(((uint64_t)(neg ? vhi ^ 0x80000000u : 0u) << 32) | (neg ? vlo : 0u)));
      return r;
    };
    auto one = [&](const Rec &r, const Acc2 &x) {
      if (r.alo & 1u) {  // header, wave-uniform: the gathered row becomes x_i
        xi = x;
        if (!PASS2) {
          den.x = fma(x.x, x.x, den.x);
          if (TWO) den.y = fma(x.y, x.y, den.y);
        }
      }
      sig_term<PASS2, TWO>(xi, x, r.v, r.w, tot_e, num, edge, s2);
    };
    Block cur = load_block(0), nxt = load_block(1);
    Rec ra0 = read_rec(cur, 0), ra1 = read_rec(cur, 1), ra2 = read_rec(cur, 2), ra3 = read_rec(cur, 3), rb0, rb1, rb2, rb3;
    Acc2 xa0 = load_row(ra0.alo, ra0.ahi), xa1 = load_row(ra1.alo, ra1.ahi), xa2 = load_row(ra2.alo, ra2.ahi), xa3 = load_row(ra3.alo, ra3.ahi);
    Acc2 xb0, xb1, xb2, xb3;
    const uint32_t T = nrec / 4;  // batches: even
    for (uint32_t t = 0; t < T; t += 2) {
      uint32_t l = (4 * t + 4) & 63;  // batch t + 1: t + 1 is odd, so it never opens a block
      rb0 = read_rec(cur, l), rb1 = read_rec(cur, l + 1), rb2 = read_rec(cur, l + 2), rb3 = read_rec(cur, l + 3);
      xb0 = load_row(rb0.alo, rb0.ahi), xb1 = load_row(rb1.alo, rb1.ahi), xb2 = load_row(rb2.alo, rb2.ahi), xb3 = load_row(rb3.alo, rb3.ahi);
      one(ra0, xa0), one(ra1, xa1), one(ra2, xa2), one(ra3, xa3);
      if (((t + 2) & 15) == 0) {  // batch t + 2 opens the next block of 64 records
        cur = nxt;
        nxt = load_block((t + 2) / 16 + 1);
      }
      l = (4 * t + 8) & 63;  // past the end: some loaded record again, gathered and never used
      ra0 = read_rec(cur, l), ra1 = read_rec(cur, l + 1), ra2 = read_rec(cur, l + 2), ra3 = read_rec(cur, l + 3);
      xa0 = load_row(ra0.alo, ra0.ahi), xa1 = load_row(ra1.alo, ra1.ahi), xa2 = load_row(ra2.alo, ra2.ahi), xa3 = load_row(ra3.alo, ra3.ahi);
      one(rb0, xb0), one(rb1, xb1), one(rb2, xb2), one(rb3, xb3);
    }
  }
  // waves of the block in index order
  double *mine = &s_red[wv][0][0];
  if (TWO) {
    mine[0 * kSlab + cc0] = num.x, mine[0 * kSlab + cc0 + 1] = num.y;
    mine[1 * kSlab + cc0] = edge.x, mine[1 * kSlab + cc0 + 1] = edge.y;
    mine[2 * kSlab + cc0] = s2.x, mine[2 * kSlab + cc0 + 1] = s2.y;
    mine[3 * kSlab + cc0] = den.x, mine[3 * kSlab + cc0 + 1] = den.y;
  } else {
    mine[0 * kSlab + cc0] = num.x, mine[1 * kSlab + cc0] = edge.x, mine[2 * kSlab + cc0] = s2.x, mine[3 * kSlab + cc0] = den.x;
  }
  __syncthreads();
  const uint32_t nb = gridDim.x + (PASS2 ? 0 : 1);
  constexpr uint32_t kMine = TWO ? kSlab : kSlab / 2;  // columns this block holds
  for (uint32_t t = threadIdx.x; t < kNQ * kMine; t += 256) {
    const uint32_t qn = t / kMine, cc = half * 64 + t % kMine;
    if (PASS2 && qn != 2) continue;
    double s = s_red[0][qn][cc];
#pragma unroll
    for (int w = 1; w < kWavesPerBlock; ++w) s = __dadd_rn(s, s_red[w][qn][cc]);
    part[(((uint64_t)slab * nb + blockIdx.x) * NQ + (PASS2 ? 0 : qn)) * kSlab + cc] = s;
  }
}

// sum of squares of the rows [r0, r1) of X that no graph row covers (the denominator runs over ALL entries of the signal,
// src/taumode.rs:596): the extra block `nb - 1` of the partials (its other quantities are zero)
__global__ __launch_bounds__(256) void signal_tail_den_kernel(const double *__restrict__ X, uint64_t ldx, uint32_t S, uint64_t r0, uint64_t r1,
                                                              uint32_t nb, double *__restrict__ part) {
  // one block per slab, thread = (row phase, column): columns coalesced, rows strided by 2, the two phases added at the end
  __shared__ double s_half[kSlab];
  const uint32_t slab = blockIdx.x, cc = threadIdx.x % kSlab, ph = threadIdx.x / kSlab;
  const uint32_t col = slab * kSlab + cc;
  double s = 0.0;
  if (col < S)
    for (uint64_t r = r0 + ph; r < r1; r += 2) {
      const double x = X[r * ldx + col];
      s = fma(x, x, s);
    }
  if (ph == 1) s_half[cc] = s;
  __syncthreads();
  if (ph == 0) {
    double *dst = part + ((uint64_t)slab * nb + (nb - 1)) * kNQ * kSlab;
    for (int qn = 0; qn < 3; ++qn) dst[qn * kSlab + cc] = 0.0;
    dst[3 * kSlab + cc] = __dadd_rn(s, s_half[cc]);
  }
}

// level 1 of the column sums: group g of `per` consecutive blocks, thread = (quantity, column), the group's blocks in order
__global__ __launch_bounds__(kNQ * kSlab) void signal_reduce1_kernel(const double *__restrict__ part, uint32_t nb, uint32_t per, uint32_t nq,
                                                                     const int *__restrict__ only_if, double *__restrict__ out) {
  if (only_if && *only_if == 0) return;
  const uint32_t slab = blockIdx.y, g = blockIdx.x, qn = threadIdx.x / kSlab, cc = threadIdx.x % kSlab;
  if (qn >= nq) return;
  const uint32_t b0 = g * per, b1 = min(b0 + per, nb);
  const double *src = part + (((uint64_t)slab * nb + b0) * nq + qn) * kSlab + cc;
  double s = 0.0;
  uint32_t b = b0;
  for (; b + 8 <= b1; b += 8) {  // eight loads in flight, added in block order
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(uint64_t)(b - b0 + u) * nq * kSlab];
#pragma unroll
    for (int u = 0; u < 8; ++u) s = __dadd_rn(s, v[u]);
  }
  for (; b < b1; ++b) s = __dadd_rn(s, src[(uint64_t)(b - b0) * nq * kSlab]);
  out[(((uint64_t)slab * gridDim.x + g) * nq + qn) * kSlab + cc] = s;
}

// level 2: the groups in order, then per column (num, edge, sum t^2, den) of the whole pass -- what the ranks of a sharded call
// exchange: sums[4][s_pad].  factor = 2 when the stream is the upper triangle (a pair stands for both of its entries).
__global__ __launch_bounds__(kNQ * kSlab) void signal_reduce2_kernel(const double *__restrict__ in, uint32_t ng, uint32_t s_pad, double factor,
                                                                     double *__restrict__ sums) {
  __shared__ double s_tot[kNQ][kSlab];
  const uint32_t slab = blockIdx.x, qn = threadIdx.x / kSlab, cc = threadIdx.x % kSlab;
  double s = 0.0;
  for (uint32_t g = 0; g < ng; ++g) s = __dadd_rn(s, in[(((uint64_t)slab * ng + g) * kNQ + qn) * kSlab + cc]);
  s_tot[qn][cc] = s;
  __syncthreads();
  if (qn == 0) {
    const uint32_t c = slab * kSlab + cc;
    sums[c] = __dmul_rn(factor, s_tot[0][cc]);  // the upper triangle's headers carry half the diagonal
    sums[(uint64_t)s_pad + c] = __dmul_rn(factor, s_tot[1][cc]);
    sums[2 * (uint64_t)s_pad + c] = __dmul_rn(factor, s_tot[2][cc]);
    sums[3 * (uint64_t)s_pad + c] = s_tot[3][cc];
  }
}
// the same for the second pass's single quantity: g2[s_pad]
__global__ __launch_bounds__(kSlab) void signal_reduce2_shares_kernel(const double *__restrict__ in, uint32_t ng, const int *__restrict__ only_if,
                                                                      double *__restrict__ g2) {
  if (*only_if == 0) return;
  const uint32_t slab = blockIdx.x, cc = threadIdx.x;
  double s = 0.0;
  for (uint32_t g = 0; g < ng; ++g) s = __dadd_rn(s, in[((uint64_t)slab * ng + g) * kSlab + cc]);
  g2[slab * kSlab + cc] = s;
}

// does any signal need the literal second pass?  (the same test as the per-item kernels, taumode.hip)
__global__ void signal_need_pass2_kernel(const double *__restrict__ sums, uint32_t s_pad, uint32_t S, int *__restrict__ need) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= S) return;
  const double edge = sums[(uint64_t)s_pad + c], s2 = sums[2 * (uint64_t)s_pad + c];
  if (edge > 0.0) {
    const double e2 = __dmul_rn(edge, edge);
    if (!(finite_bits(s2) && finite_bits(e2) && e2 > 0.0 && s2 > 1e-280)) atomicOr(need, 1);
  }
}

// lambda = tau * E / (E + tau) + (1 - tau) * G   (src/taumode.rs:597-647)
__global__ void signal_lambda_kernel(const double *__restrict__ sums, const double *__restrict__ g2, const int *__restrict__ need, double g2_factor,
                                     uint32_t s_pad, uint32_t S, const double *__restrict__ tau, double *__restrict__ out) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= S) return;
  const double num = sums[c], edge = sums[(uint64_t)s_pad + c], s2 = sums[2 * (uint64_t)s_pad + c], den = sums[3 * (uint64_t)s_pad + c];
  const double e_raw = den > 1e-12 ? __ddiv_rn(num, den) : 0.0;
  double g = 0.0;
  if (edge > 0.0) {
    const double e2 = __dmul_rn(edge, edge);
    if (finite_bits(s2) && finite_bits(e2) && e2 > 0.0 && s2 > 1e-280) g = __ddiv_rn(s2, e2);
    else g = *need ? __dmul_rn(g2_factor, g2[c]) : 0.0;  // the second pass saw a pair once when the stream is the upper triangle
  }
  const double g_raw = fmin(fmax(g, 0.0), 1.0);
  const double t = tau[c];
  const double e_b = __ddiv_rn(e_raw, __dadd_rn(e_raw, t));
  out[c] = __dadd_rn(__dmul_rn(t, e_b), __dmul_rn(__dsub_rn(1.0, t), g_raw));
}

// ---- tau of a column: order statistics of N values ----------------------------------------------------------------------------

constexpr int kSelCols = 16;       // columns per workgroup (one 128-byte line of every row)
constexpr int kSelBits = 11;
constexpr int kSelBins = 1 << kSelBits;
constexpr int kSelRowGroups = 32;  // workgroups along the rows
constexpr int kSelCap = 4096;      // values collected and sorted per column
constexpr int kSelUnroll = 8;      // rows a lane has in flight

struct SelState {
  unsigned long long prefix;  // the leading `bits` bits of the wanted key (right-aligned)
  unsigned long long k;       // wanted rank among the keys that share the prefix (0-based)
  unsigned long long m;       // finite values in the column
  unsigned long long bucket;  // keys that share the prefix
  unsigned long long min_above;  // smallest key above the prefix's range
  uint32_t bits;              // bits fixed so far
  uint32_t phase;             // 0 descending, 1 collect, 2 collected, 3 no finite value / no order statistic wanted
  uint32_t n_list;            // values collected
  uint32_t pad;
};

__global__ void sel_init_kernel(SelState *__restrict__ st, uint32_t S, int want) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= S) return;
  SelState s;
  s.prefix = 0, s.k = 0, s.m = 0, s.bucket = 0, s.min_above = ~0ull, s.bits = 0, s.phase = want ? 0u : 3u, s.n_list = 0, s.pad = 0;
  st[c] = s;
}

// one pass: histogram of the next digit of the keys that share the column's prefix
__global__ __launch_bounds__(1024) void sel_hist_kernel(const double *__restrict__ X, uint64_t ldx, uint64_t n_len, uint32_t S,
                                                        const SelState *__restrict__ st, const int *__restrict__ n_descending,
                                                        uint32_t *__restrict__ hist) {
  if (*n_descending == 0) return;
  extern __shared__ uint32_t s_hist[];  // [kSelCols][kSelBins]
  const uint32_t c0 = blockIdx.y * kSelCols, cl = threadIdx.x % kSelCols, col = c0 + cl;
  for (uint32_t t = threadIdx.x; t < kSelCols * kSelBins; t += 1024) s_hist[t] = 0;
  __syncthreads();
  const bool on = col < S;
  SelState s;
  s.phase = 3;
  if (on) s = st[col];
  const bool act = on && s.phase == 0;
  const uint32_t width = s.bits + kSelBits <= 64 ? kSelBits : 64 - s.bits;  // the last digit is narrower
  const uint32_t shift = 64 - s.bits - width;
  const uint64_t rows_per = (n_len + kSelRowGroups - 1) / kSelRowGroups;
  const uint64_t r0 = (uint64_t)blockIdx.x * rows_per, r1 = min(r0 + rows_per, n_len);
  constexpr uint64_t rstep = 1024 / kSelCols;
  if (act) for (uint64_t r = r0 + threadIdx.x / kSelCols; r < r1; r += rstep * kSelUnroll) {
    double x[kSelUnroll];  // loads first: kSelUnroll lines in flight per lane
#pragma unroll
    for (int u = 0; u < kSelUnroll; ++u) x[u] = r + u * rstep < r1 ? X[(r + u * rstep) * ldx + col] : __builtin_nan("");
#pragma unroll
    for (int u = 0; u < kSelUnroll; ++u) {
      if (!finite_bits(x[u])) continue;
      const uint64_t key = sort_key(x[u]);
      if (s.bits && (key >> (64 - s.bits)) != s.prefix) continue;
      atomicAdd(&s_hist[cl * kSelBins + (uint32_t)((key >> shift) & ((1u << width) - 1))], 1u);
    }
  }
  __syncthreads();
  for (uint32_t t = threadIdx.x; t < kSelCols * kSelBins; t += 1024) {
    const uint32_t v = s_hist[t];
    if (v && c0 + t / kSelBins < S) atomicAdd(&hist[(uint64_t)(c0 + t / kSelBins) * kSelBins + (t % kSelBins)], v);
  }
}

// after a pass: the digit that holds the wanted rank; the first pass also fixes the rank itself (it needs the finite count)
__global__ __launch_bounds__(256) void sel_choose_kernel(SelState *__restrict__ st, uint32_t *__restrict__ hist, uint32_t S, int mode, double arg,
                                                         int *__restrict__ n_descending, int *__restrict__ n_collect) {
  __shared__ unsigned long long s_sum[256];
  __shared__ unsigned long long s_total;
  const uint32_t c = blockIdx.x;
  SelState s = st[c];
  if (s.phase != 0) return;
  uint32_t *h = hist + (uint64_t)c * kSelBins;
  constexpr int per = kSelBins / 256;
  uint32_t v[per];
  unsigned long long mine = 0;
  for (int u = 0; u < per; ++u) {
    v[u] = h[threadIdx.x * per + u];
    mine += v[u];
    h[threadIdx.x * per + u] = 0;  // ready for the next pass
  }
  s_sum[threadIdx.x] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long run = 0;
    for (int t = 0; t < 256; ++t) {
      const unsigned long long x = s_sum[t];
      s_sum[t] = run;
      run += x;
    }
    s_total = run;
  }
  __syncthreads();
  const unsigned long long total = s_total;
  if (s.bits == 0) {  // ranks are taken over the finite values (src/taumode.rs:99-107)
    s.m = total;
    if (total == 0) {
      if (threadIdx.x == 0) {
        s.phase = 3;
        st[c] = s;
        atomicSub(n_descending, 1);
      }
      return;
    }
    if (mode == AS_TAU_PERCENTILE) {
      const double pp = arg != arg ? 0.0 : fmin(fmax(arg, 0.0), 1.0);
      s.k = (unsigned long long)round((double)(total - 1) * pp);  // round half away from zero, :115
    } else {
      s.k = (total & 1ull) ? total / 2 : total / 2 - 1;  // the lower of the two middle values when the count is even
    }
  }
  // the digit whose cumulative count first exceeds k
  const unsigned long long below_me = s_sum[threadIdx.x];
  unsigned long long run = below_me;
  for (int u = 0; u < per; ++u) {
    if (s.k >= run && s.k < run + v[u]) {  // exactly one thread and digit
      const uint32_t width = s.bits + kSelBits <= 64 ? kSelBits : 64 - s.bits;
      s.prefix = (s.prefix << width) | (unsigned long long)(threadIdx.x * per + u);
      s.k -= run;
      s.bucket = v[u];
      s.bits += width;
      if (s.bucket <= (unsigned long long)kSelCap || s.bits == 64) {
        s.phase = 1;
        atomicSub(n_descending, 1);
        atomicAdd(n_collect, 1);
      }
      st[c] = s;
    }
    run += v[u];
  }
}

// the keys of a column's bucket, and the smallest key above it (the upper middle value of an even count may be there)
__global__ __launch_bounds__(1024) void sel_collect_kernel(const double *__restrict__ X, uint64_t ldx, uint64_t n_len, uint32_t S,
                                                           SelState *__restrict__ st, const int *__restrict__ n_collect,
                                                           unsigned long long *__restrict__ lists) {
  if (*n_collect == 0) return;
  __shared__ unsigned long long s_min[kSelCols];
  const uint32_t c0 = blockIdx.y * kSelCols, cl = threadIdx.x % kSelCols, col = c0 + cl;
  if (threadIdx.x < kSelCols) s_min[threadIdx.x] = ~0ull;
  __syncthreads();
  const bool on = col < S;
  SelState s;
  s.phase = 3;
  if (on) s = st[col];
  const bool act = on && s.phase == 1;
  const uint64_t rows_per = (n_len + kSelRowGroups - 1) / kSelRowGroups;
  const uint64_t r0 = (uint64_t)blockIdx.x * rows_per, r1 = min(r0 + rows_per, n_len);
  unsigned long long mn = ~0ull;
  constexpr uint64_t rstep = 1024 / kSelCols;
  if (act) for (uint64_t r = r0 + threadIdx.x / kSelCols; r < r1; r += rstep * kSelUnroll) {
    double x[kSelUnroll];
#pragma unroll
    for (int u = 0; u < kSelUnroll; ++u) x[u] = r + u * rstep < r1 ? X[(r + u * rstep) * ldx + col] : __builtin_nan("");
#pragma unroll
    for (int u = 0; u < kSelUnroll; ++u) {
      if (!finite_bits(x[u])) continue;
      const uint64_t key = sort_key(x[u]);
      const uint64_t top = s.bits == 64 ? key : (key >> (64 - s.bits));
      if (top == s.prefix) {
        if (s.bits < 64) {  // with all 64 bits fixed the bucket is copies of one value: nothing to list
          const uint32_t slot = atomicAdd(&st[col].n_list, 1u);
          if (slot < (uint32_t)kSelCap) lists[(uint64_t)col * kSelCap + slot] = key;
        }
      } else if (top > s.prefix && key < mn) {
        mn = key;
      }
    }
  }
  if (act && mn != ~0ull) atomicMin(&s_min[cl], mn);
  __syncthreads();
  if (threadIdx.x < kSelCols && c0 + threadIdx.x < S && s_min[threadIdx.x] != ~0ull) atomicMin(&st[c0 + threadIdx.x].min_above, s_min[threadIdx.x]);
}

// sort a column's collected keys (bitonic, LDS) and read tau off them
__global__ __launch_bounds__(1024) void sel_pick_kernel(SelState *__restrict__ st, const unsigned long long *__restrict__ lists, uint32_t S, int mode,
                                                        const int *__restrict__ n_collect, double *__restrict__ tau) {
  if (*n_collect == 0) return;
  __shared__ unsigned long long s_key[kSelCap];
  const uint32_t c = blockIdx.x;
  SelState s = st[c];
  if (s.phase != 1) return;
  const uint32_t n = min(s.n_list, (uint32_t)kSelCap);
  for (uint32_t t = threadIdx.x; t < (uint32_t)kSelCap; t += 1024) s_key[t] = t < n ? lists[(uint64_t)c * kSelCap + t] : ~0ull;
  __syncthreads();
  for (uint32_t k = 2; k <= (uint32_t)kSelCap; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t t = threadIdx.x; t < (uint32_t)kSelCap; t += 1024) {
        const uint32_t p = t ^ j;
        if (p > t) {
          const unsigned long long a = s_key[t], b = s_key[p];
          const bool up = (t & k) == 0;
          if ((a > b) == up) {
            s_key[t] = b;
            s_key[p] = a;
          }
        }
      }
      __syncthreads();
    }
  if (threadIdx.x == 0) {
    // with all 64 bits fixed every key of the bucket is the same value (and more than kSelCap of them may exist)
    const unsigned long long k1 = s.bits == 64 ? s.prefix : s_key[s.k];
    double v = unsort_key(k1);
    if (mode == AS_TAU_MEDIAN && (s.m & 1ull) == 0) {  // even count: mean of the two middle values (src/taumode.rs:109-113)
      unsigned long long k2;
      if (s.k + 1 < s.bucket) k2 = s.bits == 64 ? s.prefix : s_key[s.k + 1];
      else k2 = s.min_above;
      v = __dmul_rn(0.5, __dadd_rn(unsort_key(k1), unsort_key(k2)));
    }
    tau[c] = fmax(v, kTauFloor);
    s.phase = 2;
    st[c] = s;
  }
}

// ---- ... in ONE pass over the column for long columns (round 5) ---------------------------------------------------------------
// The digit passes above read the N x S matrix three times (two histograms and the collection: 0.78 ms at 1 M x 128, more than
// the sums pass of the lambdas themselves).  An order statistic of a long column is known in advance to within a few per cent of
// the ranks: 8 192 sampled values of the column, sorted, bracket it -- the sample's values 4.5 standard deviations of the rank to
// either side -- and ONE pass over the column then counts the finite values, counts those below the bracket and lists the ~5 %
// inside it; the wanted rank is found among the listed ones (same keys, same tie rules, the same value bit for bit).  A column
// whose bracket misses (4.5 sigma: 1e-5), whose list runs over (heavy ties) or with too few finite sampled values is handed to
// the digit passes, as are all columns of short signals.
constexpr uint32_t kBrSample = 8192;      // sampled values per column (64 KB of keys in LDS)
constexpr uint32_t kBrMinLen = 131072;    // columns shorter than this take the digit passes
constexpr unsigned long long kBrTop = 0xFFFFFFFFFFFFFFFEull;
struct SelBracket {
  unsigned long long lo, hi;  // keys; lo > hi: no bracket (digit passes)
};
__device__ __forceinline__ uint32_t br_hash(uint32_t j) {
  uint32_t x = j * 0x9E3779B9u;
  x ^= x >> 15;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x;
}
// The sort of the 8 192 sampled keys: thread `tid` of 1 024 holds the elements e * 1024 + tid, e < 8, in registers.  Of the 91
// compare-exchange stages of the bitonic network those with a partner 1 024 or more away stay inside the thread, those with a
// partner less than 64 away inside the wave (two 32-bit shuffles); only the 22 stages in between go through LDS and a barrier --
// with all of them in LDS the sort alone took 0.12 ms a column.
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask) {
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, mask, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), mask, 64);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ void sort_sample_8192(unsigned long long (&v)[8], unsigned long long *s_buf /* 2 x 8192 */, uint32_t tid) {
  int flip = 0;
  for (uint32_t k = 2; k <= kBrSample; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      if (j >= 1024) {
        const uint32_t je = j >> 10;
#pragma unroll
        for (uint32_t e = 0; e < 8; ++e) {
          const uint32_t q = e ^ je;
          if (q > e) {
            const bool up = (((e << 10) | tid) & k) == 0;
            const unsigned long long a = v[e], b = v[q];
            if ((a > b) == up) v[e] = b, v[q] = a;
          }
        }
      } else {
        unsigned long long o[8];
        if (j >= 64) {
          unsigned long long *buf = s_buf + (size_t)flip * kBrSample;
#pragma unroll
          for (uint32_t e = 0; e < 8; ++e) buf[(e << 10) | tid] = v[e];
          __syncthreads();
#pragma unroll
          for (uint32_t e = 0; e < 8; ++e) o[e] = buf[(e << 10) | (tid ^ j)];
          flip ^= 1;  // (the next exchange writes the other half: no second barrier)
        } else {
#pragma unroll
          for (uint32_t e = 0; e < 8; ++e) o[e] = shfl_xor_u64(v[e], (int)j);
        }
        const bool lower = (tid & j) == 0;
#pragma unroll
        for (uint32_t e = 0; e < 8; ++e) {
          const bool up = (((e << 10) | tid) & k) == 0;
          // the lower index of the pair keeps the smaller key where the run ascends
          const bool take_min = lower == up;
          const unsigned long long a = v[e], b = o[e];
          v[e] = take_min ? (a < b ? a : b) : (a > b ? a : b);
        }
      }
    }
}
__global__ __launch_bounds__(1024) void sel_sample_kernel(const double *__restrict__ X, uint64_t ldx, uint64_t n_len, int mode, double arg,
                                                          SelBracket *__restrict__ br) {
  extern __shared__ unsigned long long s_samp[];  // 2 x kBrSample keys
  __shared__ uint32_t s_fin;
  const uint32_t c = blockIdx.x, tid = threadIdx.x;
  const uint64_t stride = n_len / kBrSample;  // >= 16
  if (tid == 0) s_fin = 0;
  __syncthreads();
  uint32_t fin = 0;
  unsigned long long v[8];
#pragma unroll
  for (uint32_t e = 0; e < 8; ++e) {
    const uint32_t j = (e << 10) | tid;
    const uint64_t r = (uint64_t)j * stride + br_hash(j) % stride;  // (jittered: rows dealt round-robin over clusters have periods)
    const double x = X[r * ldx + c];
    const bool f = finite_bits(x);
    v[e] = f ? sort_key(x) : ~0ull;
    fin += f ? 1u : 0u;
  }
  atomicAdd(&s_fin, fin);
  sort_sample_8192(v, s_samp, tid);
  __syncthreads();
#pragma unroll
  for (uint32_t e = 0; e < 8; ++e) s_samp[(e << 10) | tid] = v[e];
  __syncthreads();
  if (tid == 0) {
    const uint32_t ms = s_fin;
    SelBracket b;
    b.lo = 1, b.hi = 0;
    if (ms >= 256) {
      const double pp = mode == AS_TAU_PERCENTILE ? (arg != arg ? 0.0 : fmin(fmax(arg, 0.0), 1.0)) : 0.5;
      const double at = (double)(ms - 1) * pp;
      const long long w = (long long)ceil(2.25 * sqrt((double)ms)) + 2;
      const long long lo_r = (long long)floor(at) - w, hi_r = (long long)ceil(at) + w;
      b.lo = lo_r < 0 ? 0ull : s_samp[lo_r];
      b.hi = hi_r >= (long long)ms ? kBrTop : s_samp[hi_r];
    }
    br[c] = b;
  }
}
// the pass over the columns: part[(col * kSelRowGroups + row group) * 4 + {0, 1, 2}] = values listed, finite values below the
// bracket, finite values; lists[(col * kSelRowGroups + row group) * cap ..]: the keys inside the bracket
__global__ __launch_bounds__(1024) void sel_bracket_kernel(const double *__restrict__ X, uint64_t ldx, uint64_t n_len, uint32_t S,
                                                           const SelBracket *__restrict__ br, uint32_t cap, uint32_t *__restrict__ part,
                                                           unsigned long long *__restrict__ lists) {
  __shared__ uint32_t s_cnt[kSelCols], s_below[kSelCols], s_fin[kSelCols];
  const uint32_t c0 = blockIdx.y * kSelCols, cl = threadIdx.x % kSelCols, col = c0 + cl;
  if (threadIdx.x < kSelCols) s_cnt[threadIdx.x] = s_below[threadIdx.x] = s_fin[threadIdx.x] = 0;
  __syncthreads();
  const bool on = col < S;
  SelBracket b;
  b.lo = 1, b.hi = 0;
  if (on) b = br[col];
  const bool act = on && b.lo <= b.hi;
  unsigned long long *mine = lists + ((uint64_t)col * kSelRowGroups + blockIdx.x) * cap;
  uint32_t fin = 0, below = 0;
  // (the row groups take chunks of 64 rows in turn, not one range each: a sorted column -- a time series, a trend -- would put its
  // whole bracket into the list of one group)
  constexpr uint64_t rstep = 1024 / kSelCols;
  const uint64_t chunks = (n_len + rstep - 1) / rstep;
  if (act) for (uint64_t ch = blockIdx.x; ch < chunks; ch += (uint64_t)kSelRowGroups * kSelUnroll) {
    double x[kSelUnroll];
#pragma unroll
    for (int u = 0; u < kSelUnroll; ++u) {
      const uint64_t r = (ch + (uint64_t)u * kSelRowGroups) * rstep + threadIdx.x / kSelCols;
      x[u] = r < n_len ? X[r * ldx + col] : __builtin_nan("");
    }
#pragma unroll
    for (int u = 0; u < kSelUnroll; ++u) {
      if (!finite_bits(x[u])) continue;
      const uint64_t key = sort_key(x[u]);
      ++fin;
      if (key < b.lo) {
        ++below;
      } else if (key <= b.hi) {
        const uint32_t slot = atomicAdd(&s_cnt[cl], 1u);
        if (slot < cap) mine[slot] = key;
      }
    }
  }
  if (act) {
    atomicAdd(&s_fin[cl], fin);
    atomicAdd(&s_below[cl], below);
  }
  __syncthreads();
  if (threadIdx.x < kSelCols && c0 + threadIdx.x < S) {
    uint32_t *o = part + ((uint64_t)(c0 + threadIdx.x) * kSelRowGroups + blockIdx.x) * 4;
    o[0] = s_cnt[threadIdx.x], o[1] = s_below[threadIdx.x], o[2] = s_fin[threadIdx.x];
  }
}
// the wanted rank among the listed keys of a column: digit by digit from the first bit the bracket's ends differ in, then a sort of
// the bucket (same rules as sel_choose / sel_pick above).  A column this cannot serve raises *n_fallback.
__global__ __launch_bounds__(1024) void sel_bracket_pick_kernel(const SelBracket *__restrict__ br, const uint32_t *__restrict__ part,
                                                                const unsigned long long *__restrict__ lists, uint32_t cap, int mode, double arg,
                                                                double *__restrict__ tau, int *__restrict__ n_fallback) {
  __shared__ unsigned long long s_key[kSelCap];
  __shared__ uint32_t s_hist[kSelBins];
  __shared__ uint32_t s_scan[1024];
  __shared__ unsigned long long s_min;
  __shared__ uint32_t s_n, s_bin, s_before;
  const uint32_t c = blockIdx.x, tid = threadIdx.x;
  const SelBracket b = br[c];
  const uint32_t *pc = part + (uint64_t)c * kSelRowGroups * 4;
  unsigned long long m = 0, below = 0, len = 0;
  bool over = false;
  for (int g = 0; g < kSelRowGroups; ++g) {
    len += pc[g * 4], below += pc[g * 4 + 1], m += pc[g * 4 + 2];
    over |= pc[g * 4] > cap;
  }
  auto give_up = [&]() {
    if (tid == 0) atomicAdd(n_fallback, 1);
  };
  if (b.lo > b.hi || over) return give_up();
  if (m == 0) {
    if (tid == 0) tau[c] = kTauFloor;
    return;
  }
  unsigned long long k;
  if (mode == AS_TAU_PERCENTILE) {
    const double pp = arg != arg ? 0.0 : fmin(fmax(arg, 0.0), 1.0);
    k = (unsigned long long)round((double)(m - 1) * pp);
  } else {
    k = (m & 1ull) ? m / 2 : m / 2 - 1;
  }
  const bool two = mode == AS_TAU_MEDIAN && (m & 1ull) == 0;
  if (k < below || k - below + (two ? 1ull : 0ull) >= len) return give_up();
  unsigned long long kk = k - below, prefix = 0, bucket = len;
  uint32_t bits = b.lo == b.hi ? 64u : (uint32_t)__builtin_clzll(b.lo ^ b.hi);
  if (bits) prefix = bits == 64 ? b.lo : b.lo >> (64 - bits);
  auto for_keys = [&](auto &&f) {
    for (int g = 0; g < kSelRowGroups; ++g) {
      const uint32_t n = pc[g * 4];
      const unsigned long long *src = lists + ((uint64_t)c * kSelRowGroups + g) * cap;
      for (uint32_t i = tid; i < n; i += 1024) f(src[i]);
    }
  };
  while (bits < 64 && bucket > (unsigned long long)kSelCap) {
    const uint32_t width = bits + kSelBits <= 64 ? kSelBits : 64 - bits, shift = 64 - bits - width;
    for (uint32_t t = tid; t < (uint32_t)kSelBins; t += 1024) s_hist[t] = 0;
    __syncthreads();
    for_keys([&](unsigned long long key) {
      if (bits == 0 || (key >> (64 - bits)) == prefix) atomicAdd(&s_hist[(uint32_t)((key >> shift) & ((1ull << width) - 1))], 1u);
    });
    __syncthreads();
    const uint32_t h0 = s_hist[2 * tid], h1 = s_hist[2 * tid + 1];
    s_scan[tid] = h0 + h1;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const uint32_t v = tid >= (uint32_t)o ? s_scan[tid - o] : 0u;
      __syncthreads();
      s_scan[tid] += v;
      __syncthreads();
    }
    const unsigned long long before = s_scan[tid] - (h0 + h1);
    if (kk >= before && kk < before + h0) s_bin = 2 * tid, s_before = (uint32_t)before, s_n = h0;
    else if (kk >= before + h0 && kk < before + h0 + h1) s_bin = 2 * tid + 1, s_before = (uint32_t)(before + h0), s_n = h1;
    __syncthreads();
    prefix = (prefix << width) | s_bin;
    kk -= s_before;
    bucket = s_n;
    bits += width;
    __syncthreads();
  }
  unsigned long long k1, k2 = 0;
  if (bits == 64 && bucket > (unsigned long long)kSelCap) {  // copies of one value
    k1 = k2 = prefix;
  } else {
    if (tid == 0) s_n = 0, s_min = ~0ull;
    for (uint32_t t = tid; t < (uint32_t)kSelCap; t += 1024) s_key[t] = ~0ull;
    __syncthreads();
    unsigned long long mn = ~0ull;
    for_keys([&](unsigned long long key) {
      const unsigned long long top = bits == 0 ? 0ull : (bits == 64 ? key : key >> (64 - bits));
      if (bits == 0 || top == prefix) {
        const uint32_t slot = atomicAdd(&s_n, 1u);
        if (slot < (uint32_t)kSelCap) s_key[slot] = key;
      } else if (top > prefix && key < mn) {
        mn = key;
      }
    });
    if (mn != ~0ull) atomicMin(&s_min, mn);
    __syncthreads();
    uint32_t n_sort = 2;  // (the bucket of a digit holds a few dozen keys: the network over the next power of two, not over 4 096)
    while (n_sort < min(s_n, (uint32_t)kSelCap)) n_sort <<= 1;
    for (uint32_t kq = 2; kq <= n_sort; kq <<= 1)
      for (uint32_t j = kq >> 1; j > 0; j >>= 1) {
        for (uint32_t t = tid; t < n_sort; t += 1024) {
          const uint32_t q = t ^ j;
          if (q > t) {
            const unsigned long long a = s_key[t], bb = s_key[q];
            if ((a > bb) == ((t & kq) == 0)) {
              s_key[t] = bb;
              s_key[q] = a;
            }
          }
        }
        __syncthreads();
      }
    k1 = s_key[kk];
    if (two) {
      if (kk + 1 < bucket) k2 = s_key[kk + 1];
      else k2 = s_min;
      if (k2 == ~0ull) return give_up();  // (cannot happen: the rank after is inside the list)
    }
  }
  if (tid == 0) {
    double v = unsort_key(k1);
    if (two) v = __dmul_rn(0.5, __dadd_rn(unsort_key(k1), unsort_key(k2)));
    tau[c] = fmax(v, kTauFloor);
  }
}

// columns without a finite value: the floor (src/taumode.rs:101-107)
__global__ void sel_floor_kernel(const SelState *__restrict__ st, uint32_t S, double *__restrict__ tau) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < S && st[c].phase == 3) tau[c] = kTauFloor;
}

__global__ void set_int_kernel(int *p, int v) { *p = v; }

// Mean: sum and count of the finite values of every column, two fixed-order levels
__global__ __launch_bounds__(1024) void col_mean_part_kernel(const double *__restrict__ X, uint64_t ldx, uint64_t n_len, uint32_t S,
                                                             double *__restrict__ psum, unsigned long long *__restrict__ pcnt) {
  __shared__ double s_sum[1024];
  __shared__ unsigned long long s_cnt[1024];
  const uint32_t c0 = blockIdx.y * kSelCols, cl = threadIdx.x % kSelCols, col = c0 + cl, rg = threadIdx.x / kSelCols;
  const uint64_t rows_per = (n_len + kSelRowGroups - 1) / kSelRowGroups;
  const uint64_t r0 = (uint64_t)blockIdx.x * rows_per, r1 = min(r0 + rows_per, n_len);
  double s = 0.0;
  unsigned long long m = 0;
  constexpr uint64_t rstep = 1024 / kSelCols;
  if (col < S)
    for (uint64_t r = r0 + rg; r < r1; r += rstep * kSelUnroll) {
      double x[kSelUnroll];
#pragma unroll
      for (int u = 0; u < kSelUnroll; ++u) x[u] = r + u * rstep < r1 ? X[(r + u * rstep) * ldx + col] : __builtin_nan("");
#pragma unroll
      for (int u = 0; u < kSelUnroll; ++u)
        if (finite_bits(x[u])) {
          s = __dadd_rn(s, x[u]);
          ++m;
        }
    }
  s_sum[threadIdx.x] = s;
  s_cnt[threadIdx.x] = m;
  __syncthreads();
  if (threadIdx.x < kSelCols && c0 + threadIdx.x < S) {
    double a = 0.0;
    unsigned long long b = 0;
    for (uint32_t g = 0; g < 1024 / kSelCols; ++g) {
      a = __dadd_rn(a, s_sum[g * kSelCols + threadIdx.x]);
      b += s_cnt[g * kSelCols + threadIdx.x];
    }
    psum[(uint64_t)blockIdx.x * S + c0 + threadIdx.x] = a;
    pcnt[(uint64_t)blockIdx.x * S + c0 + threadIdx.x] = b;
  }
}
__global__ void col_mean_final_kernel(const double *__restrict__ psum, const unsigned long long *__restrict__ pcnt, uint32_t S, double *__restrict__ tau) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= S) return;
  double a = 0.0;
  unsigned long long b = 0;
  for (int g = 0; g < kSelRowGroups; ++g) {
    a = __dadd_rn(a, psum[(uint64_t)g * S + c]);
    b += pcnt[(uint64_t)g * S + c];
  }
  tau[c] = fmax(b ? __ddiv_rn(a, (double)b) : 0.0, kTauFloor);
}
__global__ void fill_f64_kernel(double *__restrict__ p, uint32_t n, double v) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n) p[c] = v;
}

// out[c][r] -> out[r][c]: in is n_rows x n_cols, out n_cols x n_rows (both row-major)
__global__ __launch_bounds__(256) void transpose_f64_kernel(const double *__restrict__ in, uint64_t n_rows, uint64_t n_cols, double *__restrict__ out) {
  __shared__ double tile[32][33];
  const uint64_t c0 = (uint64_t)blockIdx.x * 32, r0 = (uint64_t)blockIdx.y * 32;
  const uint32_t tx = threadIdx.x % 32, ty = threadIdx.x / 32;
  for (uint32_t k = ty; k < 32; k += 8)
    if (r0 + k < n_rows && c0 + tx < n_cols) tile[k][tx] = in[(r0 + k) * n_cols + c0 + tx];
  __syncthreads();
  for (uint32_t k = ty; k < 32; k += 8)
    if (c0 + k < n_cols && r0 + tx < n_rows) out[(c0 + k) * n_rows + r0 + tx] = tile[tx][k];
}

__global__ void nonfinite_kernel(const double *__restrict__ p, uint64_t n, int *__restrict__ flag) {
  bool bad = false;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) bad |= !finite_bits(p[i]);
  if (bad) atomicOr(flag, 1);
}

}  // namespace

int any_nonfinite(as_ctx *ctx, const double *p, uint64_t count, bool *out) {
  DevBuf<int> flag;
  AS_TRY(flag.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(flag.p, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL(nonfinite_kernel, dim3((unsigned)std::min<uint64_t>(2048, (count + 255) / 256 + 1)), dim3(256), 0, ctx->stream, p, count, flag.p);
  AS_HIP(ctx, hipGetLastError());
  int h = 0;
  AS_HIP(ctx, copy_d2h(ctx, &h, flag.p, sizeof(int)));
  *out = h != 0;
  return AS_OK;
}

// ---- host side --------------------------------------------------------------------------------------------------------------

void signal_stream_release(as_csr *L) {
  if (L->sig_rec) (void)handle_free(L->sig_rec);
  if (L->sig_off) (void)handle_free(L->sig_off);
  L->sig_rec = nullptr;
  L->sig_off = nullptr;
  L->sig_state = 0;
}

// position -> row order of the rows of L, copied into the handle (any permutation gives the same lambdas up to rounding)
int csr_set_order(as_ctx *ctx, as_csr *L, const uint32_t *order_dev) {
  signal_stream_release(L);
  if (L->order) (void)handle_free(L->order);
  L->order = nullptr;
  L->order_state = 0;
  if (!order_dev) return AS_OK;
  DevBuf<uint32_t> seen;
  DevBuf<int> bad;
  AS_TRY(seen.alloc(ctx, L->n));
  AS_TRY(bad.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(seen.p, 0, sizeof(uint32_t) * L->n, ctx->stream));
  AS_HIP(ctx, hipMemsetAsync(bad.p, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL(order_check_kernel, dim3(cdiv(L->n, 256)), dim3(256), 0, ctx->stream, order_dev, L->n, seen.p, bad.p);
  AS_HIP(ctx, hipGetLastError());
  int h = 0;
  AS_HIP(ctx, copy_d2h(ctx, &h, bad.p, sizeof(int)));
  if (h) return fail(ctx, AS_ERR_ARG, "the order is not a permutation of the rows");
  AS_HIP(ctx, handle_malloc((void **)&L->order, sizeof(uint32_t) * (L->n ? L->n : 1)));
  AS_HIP(ctx, hipMemcpyAsync(L->order, order_dev, sizeof(uint32_t) * L->n, hipMemcpyDeviceToDevice, ctx->stream));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  L->order_state = 1;
  return AS_OK;
}
static SigGeom geom_of(const as_csr *L) {
  SigGeom g;
  g.n = (uint32_t)L->n, g.ch = L->sig_ch, g.wx = L->sig_wx, g.k = L->sig_k, g.c8 = L->sig_c8;
  return g;
}

static int signal_stream_prepare(as_ctx *ctx, as_csr *L) {
  if (L->sig_state == 1) return AS_OK;
  if (L->n >= 0x7FFFFFFFull || (L->n_cols ? L->n_cols : L->n) >= 0xFFFFFFFFull) return fail(ctx, AS_ERR_UNSUPPORTED, "graph too large");
  ProfScope ps(ctx, "signal_stream_prepare");
  const bool block = (L->n_cols && L->n_cols != L->n) || L->row_begin;
  int sym = L->known_symmetric;
  if (!sym && !block && L->nnz) {
    DevBuf<int> flag;
    AS_TRY(flag.alloc(ctx, 1));
    AS_HIP(ctx, hipMemsetAsync(flag.p, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(sig_symmetry_kernel, dim3(cdiv(L->n, 256)), dim3(256), 0, ctx->stream, L->indptr, L->indices, L->data, L->n, flag.p);
    AS_HIP(ctx, hipGetLastError());
    int h = 1;
    AS_HIP(ctx, copy_d2h(ctx, &h, flag.p, sizeof(int)));
    sym = h == 0;
  }
  if (ctx->opt_signal_sym == 0) sym = 0;
  // geometry: as many waves per XCD group as are resident together (the window of the interleave is what they cover at one time)
  int blocks_per_cu = 0;
  AS_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, (const void *)signal_sums_kernel<false, 2>, 256, 0));
  hipDeviceProp_t prop;
  AS_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
  const uint32_t cus = (uint32_t)std::max(8, prop.multiProcessorCount);
  // Default: ONE chunk per wave, so the dispatcher hands the chunks out in order as waves retire and the window of rows in
  // flight on an XCD stays compact whatever the pace of its CUs (measured at 1 M x 128, rows by cluster: 0.67 ms against 0.97 ms
  // for waves that are all resident and interleave their chunks statically -- their window drifts apart; the option that fixed the
  // wave count instead was removed in round 5).
  uint32_t wx_max = 0x7FFFFFF0u / 8;
  (void)blocks_per_cu, (void)cus;
  const uint32_t ch = 16;  // consecutive rows a wave takes at a time (8 / 32 measured: no better)
  const uint64_t n_chunks = std::max<uint64_t>(1, (L->n + ch - 1) / ch);
  const uint32_t c8 = (uint32_t)((n_chunks + 7) / 8);
  uint32_t wx = std::min<uint32_t>(wx_max, (c8 + kWavesPerBlock - 1) / kWavesPerBlock * kWavesPerBlock);
  L->sig_ch = ch, L->sig_wx = wx, L->sig_c8 = c8, L->sig_k = (c8 + wx - 1) / wx;
  const SigGeom g = geom_of(L);
  const uint64_t slots = sig_slots(g);
  const uint32_t *order = ctx->opt_signal_order ? L->order : nullptr;
  DevBuf<uint32_t> cnt;
  AS_TRY(cnt.alloc(ctx, slots + 1));
  hipLaunchKernelGGL(sig_count_kernel, dim3(cdiv(slots, 256)), dim3(256), 0, ctx->stream, L->indptr, L->indices, g, L->row_begin, order, sym, cnt.p);
  DevBuf<uint32_t> pad_row;
  AS_TRY(pad_row.alloc(ctx, 8 * wx));
  hipLaunchKernelGGL(sig_pad_kernel, dim3(cdiv(8 * wx, 64)), dim3(64), 0, ctx->stream, g, L->row_begin, order, cnt.p, pad_row.p);
  AS_HIP(ctx, hipGetLastError());
  AS_HIP(ctx, handle_malloc((void **)&L->sig_off, sizeof(uint64_t) * (slots + 1)));
  AS_TRY(exclusive_scan_u32_u64(ctx, cnt.p, slots, L->sig_off));
  uint64_t n_rec = 0;
  AS_HIP(ctx, copy_d2h(ctx, &n_rec, L->sig_off + slots, sizeof(uint64_t)));
  AS_HIP(ctx, handle_malloc(&L->sig_rec, sizeof(SigRec) * (n_rec ? n_rec : 1)));
  hipLaunchKernelGGL(sig_fill_kernel, dim3(cdiv(slots, 256)), dim3(256), 0, ctx->stream, L->indptr, L->indices, L->data, g, L->row_begin, order, sym,
                     L->sig_off, (const uint32_t *)pad_row.p, (SigRec *)L->sig_rec);
  AS_HIP(ctx, hipGetLastError());
  L->sig_nrec = n_rec;
  L->sig_sym = sym;
  L->sig_state = 1;
  ctx->stats["signal_stream_sym"] = sym;
  ctx->stats["signal_stream_records"] = (int64_t)n_rec;
  ctx->stats["signal_stream_ordered"] = order ? 1 : 0;
  ctx->stats["signal_stream_waves"] = (int64_t)8 * wx;
  return AS_OK;
}

template <bool PASS2>
static void launch_sums(as_ctx *ctx, bool vec, uint32_t nbk, uint32_t n_slabs, const double *X, uint32_t S, const as_csr *L, uint32_t spw,
                        const double *edge_tot, const int *need, double *part) {
#define AS_SIG_LAUNCH(W, GY)                                                                                                          \
  hipLaunchKernelGGL((signal_sums_kernel<PASS2, W>), dim3(nbk, GY), dim3(256), 0, ctx->stream, X, (uint64_t)S, S, (const SigRec *)L->sig_rec, \
                     (const uint64_t *)L->sig_off, spw, L->sig_wx, edge_tot, need, part)
  if (vec)
    AS_SIG_LAUNCH(2, n_slabs);
  else
    AS_SIG_LAUNCH(0, n_slabs);
#undef AS_SIG_LAUNCH
}

// (num, edge, sum t^2, den) of every signal over the rows of L: sums[4][s_pad], s_pad = 128 * ceil(S / 128).  X: n_len x S row-major
// (ldx = S); the graph's rows are rows [L->row_begin, L->row_begin + L->n) of X; den also covers rows [tail_begin, tail_end) of X.
int signal_sums(as_ctx *ctx, const double *X, uint64_t n_len, uint32_t S, as_csr *L, uint64_t tail_begin, uint64_t tail_end, double *sums) {
  AS_TRY(signal_stream_prepare(ctx, L));
  const uint32_t n_slabs = (S + kSlab - 1) / kSlab, s_pad = n_slabs * kSlab;
  const uint32_t nbk = 8 * L->sig_wx / kWavesPerBlock;  // workgroups of the graph pass
  const uint32_t nb = nbk + 1;                          // + the tail rows' block
  const uint32_t spw = L->sig_k * L->sig_ch;
  const bool vec = S >= 2 && (S & 1) == 0 && (((uintptr_t)X) & 15) == 0;
  const uint32_t ng = 32, ng_mid = nb > 4096 ? 1024 : 0;  // levels of the column sums: blocks -> [1024 groups ->] 32 groups -> totals, each in index order
  DevBuf<double> part, r0, r1;
  AS_TRY(part.alloc(ctx, (size_t)n_slabs * nb * kNQ * kSlab));
  if (ng_mid) AS_TRY(r0.alloc(ctx, (size_t)n_slabs * ng_mid * kNQ * kSlab));
  AS_TRY(r1.alloc(ctx, (size_t)n_slabs * ng * kNQ * kSlab));
  {
    ProfScope ps(ctx, "signal_sums");
    hipLaunchKernelGGL(signal_tail_den_kernel, dim3(n_slabs), dim3(256), 0, ctx->stream, X, (uint64_t)S, S, tail_begin, std::max(tail_begin, tail_end), nb, part.p);
    launch_sums<false>(ctx, vec, nbk, n_slabs, X, S, L, spw, nullptr, nullptr, part.p);
    AS_HIP(ctx, hipGetLastError());
  }
  {
    ProfScope ps(ctx, "signal_reduce");
    const double *src = part.p;
    uint32_t count = nb;
    if (ng_mid) {  // many workgroups: an extra level keeps every level's sequential run short
      hipLaunchKernelGGL(signal_reduce1_kernel, dim3(ng_mid, n_slabs), dim3(kNQ * kSlab), 0, ctx->stream, src, count, (count + ng_mid - 1) / ng_mid, (uint32_t)kNQ,
                         (const int *)nullptr, r0.p);
      src = r0.p;
      count = ng_mid;
    }
    hipLaunchKernelGGL(signal_reduce1_kernel, dim3(ng, n_slabs), dim3(kNQ * kSlab), 0, ctx->stream, src, count, (count + ng - 1) / ng, (uint32_t)kNQ,
                       (const int *)nullptr, r1.p);
    hipLaunchKernelGGL(signal_reduce2_kernel, dim3(n_slabs), dim3(kNQ * kSlab), 0, ctx->stream, (const double *)r1.p, ng, s_pad, L->sig_sym ? 2.0 : 1.0, sums);
    AS_HIP(ctx, hipGetLastError());
  }
  return AS_OK;
}

// second round: g2[s_pad] = sum of squared shares given the total edge energies (row 1 of sums); runs only where *need != 0
int signal_shares(as_ctx *ctx, const double *X, uint32_t S, as_csr *L, const double *sums, const int *need, double *g2) {
  const uint32_t n_slabs = (S + kSlab - 1) / kSlab, s_pad = n_slabs * kSlab;
  const uint32_t nbk = 8 * L->sig_wx / kWavesPerBlock;
  const uint32_t spw = L->sig_k * L->sig_ch;
  const uint32_t ng = 32, ng_mid = nbk > 4096 ? 1024 : 0;
  DevBuf<double> part, r0, r1;
  AS_TRY(part.alloc(ctx, (size_t)n_slabs * nbk * kSlab));
  if (ng_mid) AS_TRY(r0.alloc(ctx, (size_t)n_slabs * ng_mid * kSlab));
  AS_TRY(r1.alloc(ctx, (size_t)n_slabs * ng * kSlab));
  const bool vec = S >= 2 && (S & 1) == 0 && (((uintptr_t)X) & 15) == 0;
  ProfScope ps(ctx, "signal_shares");
  // a skipped pass leaves g2 unwritten: the lambda kernel reads it only where *need is set
  launch_sums<true>(ctx, vec, nbk, n_slabs, X, S, L, spw, sums + s_pad, need, part.p);
  const double *src = part.p;
  uint32_t count = nbk;
  if (ng_mid) {
    hipLaunchKernelGGL(signal_reduce1_kernel, dim3(ng_mid, n_slabs), dim3(kNQ * kSlab), 0, ctx->stream, src, count, (count + ng_mid - 1) / ng_mid, 1u, need, r0.p);
    src = r0.p;
    count = ng_mid;
  }
  hipLaunchKernelGGL(signal_reduce1_kernel, dim3(ng, n_slabs), dim3(kNQ * kSlab), 0, ctx->stream, src, count, (count + ng - 1) / ng, 1u, need, r1.p);
  hipLaunchKernelGGL(signal_reduce2_shares_kernel, dim3(n_slabs), dim3(kSlab), 0, ctx->stream, (const double *)r1.p, ng, need, g2);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

// the bracket form of the order statistics (kernels above): its buffers, and its three launches on a stream
struct TauBracket {
  DevBuf<SelBracket> br;
  DevBuf<uint32_t> part;
  DevBuf<unsigned long long> lists;
  DevBuf<int> n_fb;  // columns it could not serve (read back by the caller)
};
// Mean: its two launches on a stream (the partial sums live until the stream has run them)
struct TauMean {
  DevBuf<double> psum;
  DevBuf<unsigned long long> pcnt;
};
static int tau_mean_launch(as_ctx *ctx, hipStream_t st, const double *X, uint64_t n_len, uint32_t S, double *tau, TauMean &tm) {
  const uint32_t col_groups = (S + kSelCols - 1) / kSelCols;
  AS_TRY(tm.psum.alloc(ctx, (size_t)kSelRowGroups * S));
  AS_TRY(tm.pcnt.alloc(ctx, (size_t)kSelRowGroups * S));
  hipLaunchKernelGGL(col_mean_part_kernel, dim3(kSelRowGroups, col_groups), dim3(1024), 0, st, X, (uint64_t)S, n_len, S, tm.psum.p, tm.pcnt.p);
  hipLaunchKernelGGL(col_mean_final_kernel, dim3(cdiv(S, 256)), dim3(256), 0, st, (const double *)tm.psum.p, (const unsigned long long *)tm.pcnt.p, S, tau);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}
static uint32_t tau_bracket_cap(uint64_t n_len) {
  const uint64_t rows_per = (n_len + kSelRowGroups - 1) / kSelRowGroups;
  // (the bracket holds (2 w + 2) / 8192 of the ranks, w = 2.25 sqrt(8192) + 2: 5.1 %; room for 2.5 times that per row group)
  return (uint32_t)std::min<uint64_t>(0xFFFFFFFFull, std::max<uint64_t>(1024, (uint64_t)((double)rows_per * 0.051 * 2.5)));
}
// (S columns: the lists take S x 32 x cap keys -- 13 % of the matrix; beyond 4 GB of them the digit passes, which need none)
static bool tau_bracket_applies(const as_ctx *ctx, uint64_t n_len, int mode, uint32_t S) {
  return (mode == AS_TAU_MEDIAN || mode == AS_TAU_PERCENTILE) && n_len >= (uint64_t)kBrMinLen && ctx->opt_signal_tau_bracket &&
         (uint64_t)S * kSelRowGroups * tau_bracket_cap(n_len) * sizeof(unsigned long long) <= (4ull << 30);
}
// the buffers first, apart from the launches: a caller that cannot have the lists (hipMalloc fails: up to 4 GB of them) takes the digit
// passes, which need none, instead of failing -- and nothing has been queued on any stream by then
static int tau_bracket_alloc(as_ctx *ctx, uint64_t n_len, uint32_t S, TauBracket &tb) {
  const uint32_t cap = tau_bracket_cap(n_len);
  AS_TRY(tb.br.alloc(ctx, S));
  AS_TRY(tb.part.alloc(ctx, (size_t)S * kSelRowGroups * 4));
  AS_TRY(tb.lists.alloc(ctx, (size_t)S * kSelRowGroups * cap));
  AS_TRY(tb.n_fb.alloc(ctx, 1));
  return AS_OK;
}
static int tau_bracket_launch(as_ctx *ctx, hipStream_t st, const double *X, uint64_t n_len, uint32_t S, int mode, double arg, double *tau, TauBracket &tb) {
  const uint32_t col_groups = (S + kSelCols - 1) / kSelCols;
  const uint32_t cap = tau_bracket_cap(n_len);
  AS_HIP(ctx, hipMemsetAsync(tb.n_fb.p, 0, sizeof(int), st));
  AS_HIP(ctx, hipFuncSetAttribute((const void *)sel_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * kBrSample * sizeof(unsigned long long))));
  hipLaunchKernelGGL(sel_sample_kernel, dim3(S), dim3(1024), 2 * kBrSample * sizeof(unsigned long long), st, X, (uint64_t)S, n_len, mode, arg, tb.br.p);
  hipLaunchKernelGGL(sel_bracket_kernel, dim3(kSelRowGroups, col_groups), dim3(1024), 0, st, X, (uint64_t)S, n_len, S, (const SelBracket *)tb.br.p, cap, tb.part.p,
                     tb.lists.p);
  hipLaunchKernelGGL(sel_bracket_pick_kernel, dim3(S), dim3(1024), 0, st, (const SelBracket *)tb.br.p, (const uint32_t *)tb.part.p,
                     (const unsigned long long *)tb.lists.p, cap, mode, arg, tau, tb.n_fb.p);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

// tau of every column of X (n_len x S): TauMode::select_tau on the column's own values
static int signal_taus_impl(as_ctx *ctx, const double *X, uint64_t n_len, uint32_t S, int mode, double arg, double *tau, bool digits_only) {
  if (mode < AS_TAU_FIXED || mode > AS_TAU_PERCENTILE) return fail(ctx, AS_ERR_ARG, "bad tau_mode");
  ProfScope ps(ctx, "signal_taus");
  if (mode == AS_TAU_FIXED) {
    const double t = (std::isfinite(arg) && arg > 0.0) ? arg : kTauFloor;
    hipLaunchKernelGGL(fill_f64_kernel, dim3(cdiv(S, 256)), dim3(256), 0, ctx->stream, tau, S, t);
    AS_HIP(ctx, hipGetLastError());
    return AS_OK;
  }
  const uint32_t col_groups = (S + kSelCols - 1) / kSelCols;
  if (mode == AS_TAU_MEAN) {
    TauMean tm;
    return tau_mean_launch(ctx, ctx->stream, X, n_len, S, tau, tm);
  }
  if (!digits_only && tau_bracket_applies(ctx, n_len, mode, S)) {  // long columns: a sampled bracket and one pass
    TauBracket tb;
    // (the only thing that can fail in there is hipMalloc: no room for the lists -> the digit passes, which need none)
    const int rc_alloc = tau_bracket_alloc(ctx, n_len, S, tb);
    if (rc_alloc == AS_OK) {
      AS_TRY(tau_bracket_launch(ctx, ctx->stream, X, n_len, S, mode, arg, tau, tb));
      int h_fb = 0;
      AS_HIP(ctx, copy_d2h(ctx, &h_fb, tb.n_fb.p, sizeof(int)));
      ctx->stats["signal_tau_bracket_fallback_columns"] = h_fb;
      if (h_fb == 0) return AS_OK;
    }
    // (some column's bracket missed or its list ran over, or there was no room for the lists: the digit passes for all of them)
  }
  DevBuf<SelState> st;
  DevBuf<uint32_t> hist;
  DevBuf<int> counters;  // [0] columns still descending, [1] columns ready to collect
  DevBuf<unsigned long long> lists;
  AS_TRY(st.alloc(ctx, S));
  AS_TRY(hist.alloc(ctx, (size_t)S * kSelBins));
  AS_TRY(counters.alloc(ctx, 2));
  AS_TRY(lists.alloc(ctx, (size_t)S * kSelCap));
  AS_HIP(ctx, hipMemsetAsync(hist.p, 0, sizeof(uint32_t) * (size_t)S * kSelBins, ctx->stream));
  hipLaunchKernelGGL(sel_init_kernel, dim3(cdiv(S, 256)), dim3(256), 0, ctx->stream, st.p, S, 1);
  hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(1), 0, ctx->stream, counters.p, (int)S);
  hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(1), 0, ctx->stream, counters.p + 1, 0);
  const size_t lds = sizeof(uint32_t) * kSelCols * kSelBins;
  AS_HIP(ctx, hipFuncSetAttribute((const void *)sel_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  auto collect = [&]() -> int {
    hipLaunchKernelGGL(sel_collect_kernel, dim3(kSelRowGroups, col_groups), dim3(1024), 0, ctx->stream, X, (uint64_t)S, n_len, S, st.p, (const int *)(counters.p + 1),
                       lists.p);
    hipLaunchKernelGGL(sel_pick_kernel, dim3(S), dim3(1024), 0, ctx->stream, st.p, (const unsigned long long *)lists.p, S, mode, (const int *)(counters.p + 1), tau);
    hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(1), 0, ctx->stream, counters.p + 1, 0);
    AS_HIP(ctx, hipGetLastError());
    return AS_OK;
  };
  // 64 bits = five digits of 11 and one of 9: after two passes the bucket of a smooth distribution holds N / 2^21 .. N / 2^11 values
  for (int pass = 0; pass < 6; ++pass) {
    hipLaunchKernelGGL(sel_hist_kernel, dim3(kSelRowGroups, col_groups), dim3(1024), lds, ctx->stream, X, (uint64_t)S, n_len, S, (const SelState *)st.p,
                       (const int *)counters.p, hist.p);
    hipLaunchKernelGGL(sel_choose_kernel, dim3(S), dim3(256), 0, ctx->stream, st.p, hist.p, S, mode, arg, counters.p, counters.p + 1);
    AS_HIP(ctx, hipGetLastError());
    if (pass == 1 || pass == 3 || pass == 5) AS_TRY(collect());
  }
  hipLaunchKernelGGL(sel_floor_kernel, dim3(cdiv(S, 256)), dim3(256), 0, ctx->stream, (const SelState *)st.p, S, tau);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

int signal_taus(as_ctx *ctx, const double *X, uint64_t n_len, uint32_t S, int mode, double arg, double *tau) {
  return signal_taus_impl(ctx, X, n_len, S, mode, arg, tau, false);
}

// ---- pieces of the sharded form (sharded.hip): every rank sums over its row block, the sums meet in rank order ----------------
__global__ void signal_rank_sum_kernel(const double *__restrict__ all, uint32_t world, uint32_t count, double *__restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double s = 0.0;
  for (uint32_t r = 0; r < world; ++r) s = __dadd_rn(s, all[(uint64_t)r * count + i]);
  out[i] = s;
}
int signal_rank_sum(as_ctx *ctx, const double *all, uint32_t world, uint32_t count, double *out) {
  hipLaunchKernelGGL(signal_rank_sum_kernel, dim3(cdiv(count, 256)), dim3(256), 0, ctx->stream, all, world, count, out);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}
uint32_t signal_pad(uint32_t S) { return (S + kSlab - 1) / kSlab * kSlab; }
// *need_dev = 1 when some signal's G cannot be taken from the sums of squares (then the literal second pass runs)
int signal_need_pass2(as_ctx *ctx, const double *sums, uint32_t S, int *need_dev) {
  AS_HIP(ctx, hipMemsetAsync(need_dev, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL(signal_need_pass2_kernel, dim3(cdiv(S, 256)), dim3(256), 0, ctx->stream, sums, signal_pad(S), S, need_dev);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}
int signal_lambda_from_sums(as_ctx *ctx, const double *sums, const double *g2, const int *need_dev, bool upper_only, uint32_t S, const double *tau,
                            double *out) {
  hipLaunchKernelGGL(signal_lambda_kernel, dim3(cdiv(S, 256)), dim3(256), 0, ctx->stream, sums, g2, need_dev, upper_only ? 2.0 : 1.0, signal_pad(S), S, tau, out);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

int signal_lambda_finish(as_ctx *ctx, const double *X, uint32_t S, as_csr *L, const double *sums, const double *tau, double *out) {
  const uint32_t s_pad = (S + kSlab - 1) / kSlab * kSlab;
  DevBuf<int> need;
  DevBuf<double> g2;
  AS_TRY(need.alloc(ctx, 1));
  AS_TRY(g2.alloc(ctx, s_pad));
  AS_HIP(ctx, hipMemsetAsync(need.p, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL(signal_need_pass2_kernel, dim3(cdiv(S, 256)), dim3(256), 0, ctx->stream, sums, s_pad, S, need.p);
  AS_TRY(signal_shares(ctx, X, S, L, sums, need.p, g2.p));
  hipLaunchKernelGGL(signal_lambda_kernel, dim3(cdiv(S, 256)), dim3(256), 0, ctx->stream, sums, (const double *)g2.p, (const int *)need.p, L->sig_sym ? 2.0 : 1.0, s_pad, S, tau, out);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

// Once anything has been queued on the context's side stream no exit may hand the buffers it uses back to the workspace cache (which
// orders blocks against the MAIN stream only) before the side stream has run: declared behind those buffers, drained in front of them.
struct SideDrain {
  as_ctx *ctx;
  bool armed = false;
  ~SideDrain() {
    if (armed) (void)hipStreamSynchronize(ctx->side);
  }
};

// lambdas of the S columns of X (n_len x S) against L (n <= n_len), one GPU
int signal_lambdas(as_ctx *ctx, const double *X, uint64_t n_len, uint32_t S, as_csr *L, int mode, double arg, double *out) {
  if (S == 0) return AS_OK;
  if (L->n > n_len) return fail(ctx, AS_ERR_DIM, "graph has more nodes than the signals have entries");
  const uint32_t s_pad = (S + kSlab - 1) / kSlab * kSlab;
  DevBuf<double> sums, tau;
  AS_TRY(sums.alloc(ctx, (size_t)4 * s_pad));
  AS_TRY(tau.alloc(ctx, S));
  // The order statistics need nothing of the graph and the sums nothing of tau: the bracket form runs on the context's side stream
  // UNDER the sums pass (0.45 ms of sampling, one pass over the columns and the pick, hidden behind 0.66 ms of sums at 1 M x 128).
  TauBracket tb;  // (declared in front of the guard: the guard drains the side stream BEFORE these blocks go back to the cache)
  bool bracket = tau_bracket_applies(ctx, n_len, mode, S) && ctx->opt_signal_tau_overlap;
  if (bracket) {
    // (the only thing that can fail in there is hipMalloc) no room for the lists: the digit passes on the main stream (below)
    if (tau_bracket_alloc(ctx, n_len, S, tb) != AS_OK) {
      bracket = false;
      tb.br.release(), tb.part.release(), tb.lists.release(), tb.n_fb.release();
    }
  }
  const bool digits_only = tau_bracket_applies(ctx, n_len, mode, S) && ctx->opt_signal_tau_overlap && !bracket;
  if (bracket) {
    AS_TRY(ctx_side_stream(ctx));
    SideDrain drain{ctx};
    AS_HIP(ctx, hipEventRecord(ctx->side_start, ctx->stream));  // (X, tau may be blocks the main stream has just written or recycled)
    AS_HIP(ctx, hipStreamWaitEvent(ctx->side, ctx->side_start, 0));
    drain.armed = true;  // from here on every exit waits for the side stream before tb's blocks are released
    AS_TRY(tau_bracket_launch(ctx, ctx->side, X, n_len, S, mode, arg, tau.p, tb));
    AS_HIP(ctx, hipMemcpyAsync(ctx->side_word, tb.n_fb.p, sizeof(int), hipMemcpyDeviceToHost, ctx->side));
    AS_HIP(ctx, hipEventRecord(ctx->side_done, ctx->side));
    const int rc_sums = signal_sums(ctx, X, n_len, S, L, L->n, n_len, sums.p);
    // (the side stream is drained before anything of it goes out of scope, whatever the sums pass returned)
    AS_HIP(ctx, hipStreamSynchronize(ctx->side));
    drain.armed = false;
    const int h_fb = *ctx->side_word;
    AS_TRY(rc_sums);
    AS_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_done, 0));
    ctx->stats["signal_tau_bracket_fallback_columns"] = h_fb;
    ctx->stats["signal_tau_overlapped"] = 1;
    if (h_fb) AS_TRY(signal_taus_impl(ctx, X, n_len, S, mode, arg, tau.p, true));
    AS_TRY(signal_lambda_finish(ctx, X, S, L, sums.p, tau.p, out));
    return AS_OK;
  }
  if (mode == AS_TAU_MEAN && n_len >= (uint64_t)kBrMinLen && ctx->opt_signal_tau_overlap) {  // the column means likewise (no read-back)
    AS_TRY(ctx_side_stream(ctx));
    TauMean tm;
    SideDrain drain{ctx};
    AS_HIP(ctx, hipEventRecord(ctx->side_start, ctx->stream));
    AS_HIP(ctx, hipStreamWaitEvent(ctx->side, ctx->side_start, 0));
    drain.armed = true;
    AS_TRY(tau_mean_launch(ctx, ctx->side, X, n_len, S, tau.p, tm));
    AS_HIP(ctx, hipEventRecord(ctx->side_done, ctx->side));
    const int rc_sums = signal_sums(ctx, X, n_len, S, L, L->n, n_len, sums.p);
    AS_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_done, 0));
    // (tm's blocks go back to the cache of the main stream, which has just been told to wait for their last use)
    drain.armed = false;
    AS_TRY(rc_sums);
    ctx->stats["signal_tau_overlapped"] = 1;
    AS_TRY(signal_lambda_finish(ctx, X, S, L, sums.p, tau.p, out));
    return AS_OK;
  }
  ctx->stats["signal_tau_overlapped"] = 0;
  AS_TRY(signal_taus_impl(ctx, X, n_len, S, mode, arg, tau.p, digits_only));
  AS_TRY(signal_sums(ctx, X, n_len, S, L, L->n, n_len, sums.p));
  AS_TRY(signal_lambda_finish(ctx, X, S, L, sums.p, tau.p, out));
  return AS_OK;
}

// signals stored as ROWS (n_sig x n_len, the layout of as_compute_taumode): batches of up to 128 signals are turned into columns
int signal_lambdas_rows(as_ctx *ctx, const double *items, uint64_t n_sig, uint64_t n_len, as_csr *L, int mode, double arg, double *out) {
  const uint64_t batch = 128;
  DevBuf<double> xt;
  AS_TRY(xt.alloc(ctx, (size_t)std::min(batch, n_sig) * n_len));
  for (uint64_t s0 = 0; s0 < n_sig; s0 += batch) {
    const uint64_t sb = std::min(batch, n_sig - s0);
    hipLaunchKernelGGL(transpose_f64_kernel, dim3(cdiv(n_len, 32), cdiv(sb, 32)), dim3(256), 0, ctx->stream, items + s0 * n_len, sb, n_len, xt.p);
    AS_HIP(ctx, hipGetLastError());
    AS_TRY(signal_lambdas(ctx, xt.p, n_len, (uint32_t)sb, L, mode, arg, out + s0));
  }
  return AS_OK;
}

// TauMode::select_tau for rows too long for the per-item kernels: batches of rows turned into columns
int signal_taus_rows(as_ctx *ctx, const double *rows, uint64_t n_rows, uint64_t len, int mode, double arg, double *taus) {
  const uint64_t batch = 128;
  DevBuf<double> xt;
  AS_TRY(xt.alloc(ctx, (size_t)std::min(batch, n_rows) * len));
  for (uint64_t s0 = 0; s0 < n_rows; s0 += batch) {
    const uint64_t sb = std::min(batch, n_rows - s0);
    hipLaunchKernelGGL(transpose_f64_kernel, dim3(cdiv(len, 32), cdiv(sb, 32)), dim3(256), 0, ctx->stream, rows + s0 * len, sb, len, xt.p);
    AS_HIP(ctx, hipGetLastError());
    AS_TRY(signal_taus(ctx, xt.p, len, (uint32_t)sb, mode, arg, taus + s0));
  }
  return AS_OK;
}

}  // namespace asx
