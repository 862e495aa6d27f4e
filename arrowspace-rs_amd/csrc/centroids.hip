// centroids.hip -- batch form of `nearest_centroid` (src/clustering.rs:913-928; its callers :590, :721, :764 walk the
// rows of the matrix one by one): for every row the index of the nearest centroid and the squared distance,
// in the reference's arithmetic -- d2 = sequential sum over the features of (a - b) * (a - b), two roundings per
// term, and a strict '<' over the centroids in index order (the first of equal minima wins; a row whose distances
// are all NaN keeps (0, +inf)).  This is the data-parallel step of the clustering pre-stage.
//
// One workgroup = 64 rows x 64 centroids at a time, features in chunks of 32 through LDS.  Wave g of the workgroup
// owns the centroids g, g + 4, ... of the chunk (16 of them: a wave's lanes all read the same centroid value, an LDS
// broadcast), lane r owns row r; the 16 partial sums of a lane live in registers across the feature chunks, so every
// (row, centroid) sum runs over the features in order.  The four waves' minima meet in LDS.
//
// A batch of the incremental clustering is 4 096 rows = 64 row tiles: too few workgroups for 256 CUs.  Then the centroid
// chunks are dealt to gridDim.y workgroups per row tile, each leaves its minimum per row, and nearest_merge_kernel takes the
// minimum of those with the smallest index among equal minima -- the same (index, distance) as the single sweep.
#include <algorithm>
#include <cstring>
#include <vector>

#include "common.h"
#include "mfma_common.h"

namespace asx {

namespace {

constexpr int kCtRows = 64, kCtCents = 64, kCtFeat = 32, kCtPitch = kCtFeat + 1;

__global__ __launch_bounds__(256) void nearest_centroids_kernel(const double *__restrict__ rows, uint64_t n, uint64_t d,
                                                                const double *__restrict__ cents, uint64_t n_cents,
                                                                uint64_t *__restrict__ out_idx, double *__restrict__ out_d2,
                                                                const uint32_t *__restrict__ row_list,
                                                                const uint32_t *__restrict__ row_count) {
  // row_list: the rows to do are row_list[0 .. *row_count) (the leftovers of the MFMA candidate path; results by row id)
  if (row_list) n = *row_count;
  // gridDim.y > 1: this workgroup sweeps the centroid chunks blockIdx.y, + gridDim.y, ... and writes slice blockIdx.y of
  // out_idx / out_d2 (n entries each; index ~0 = no centroid of the slice was comparable)
  __shared__ double s_row[kCtRows * kCtPitch];
  __shared__ double s_cen[kCtCents * kCtPitch];
  __shared__ double s_best[4][kCtRows];
  __shared__ uint64_t s_besti[4][kCtRows];
  const int tid = threadIdx.x, r = tid & 63, g = tid >> 6;
  for (uint64_t row0 = (uint64_t)blockIdx.x * kCtRows; row0 < n; row0 += (uint64_t)gridDim.x * kCtRows) {
    double best = __builtin_inf();
    uint64_t besti = 0;
    bool found = false;
    for (uint64_t c0 = (uint64_t)blockIdx.y * kCtCents; c0 < n_cents; c0 += (uint64_t)gridDim.y * kCtCents) {
      double acc[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = 0.0;
      for (uint64_t k0 = 0; k0 < d; k0 += kCtFeat) {
        const int feats = (int)((d - k0) < (uint64_t)kCtFeat ? (d - k0) : (uint64_t)kCtFeat);
        __syncthreads();
        // 64 x 32 elements each: element e = q * 256 + tid -> (e / 32, e % 32); lanes read 256-B row segments
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int e = q * 256 + tid, rr = e >> 5, kk = e & 31;
          s_row[rr * kCtPitch + kk] =
              (row0 + rr < n && kk < feats) ? rows[(row_list ? (uint64_t)row_list[row0 + rr] : row0 + rr) * d + k0 + kk] : 0.0;
          s_cen[rr * kCtPitch + kk] = (c0 + rr < n_cents && kk < feats) ? cents[(c0 + rr) * d + k0 + kk] : 0.0;
        }
        __syncthreads();
        for (int kk = 0; kk < feats; ++kk) {
          const double a = s_row[r * kCtPitch + kk];
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const double diff = __dsub_rn(a, s_cen[(g + 4 * j) * kCtPitch + kk]);
            acc[j] = __dadd_rn(acc[j], __dmul_rn(diff, diff));
          }
        }
      }
      // this lane's 16 centroids in index order; strict '<' keeps the first of equal minima
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const uint64_t c = c0 + g + 4 * j;
        if (c < n_cents && acc[j] < best) {
          best = acc[j];
          besti = c;
          found = true;
        }
      }
    }
    // the four waves hold disjoint centroid sets: the minimum, and among equal minima the smallest index
    s_best[g][r] = found ? best : __builtin_inf();
    s_besti[g][r] = found ? besti : ~0ull;
    __syncthreads();
    if (g == 0 && row0 + r < n) {
      double b = __builtin_inf();
      uint64_t bi = ~0ull;
      for (int w = 0; w < 4; ++w) {
        const double v = s_best[w][r];
        const uint64_t vi = s_besti[w][r];
        if (vi != ~0ull && (v < b || (v == b && vi < bi))) {
          b = v;
          bi = vi;
        }
      }
      const uint64_t at = row_list ? (uint64_t)row_list[row0 + r] : (uint64_t)blockIdx.y * n + row0 + r;
      out_idx[at] = (bi == ~0ull && gridDim.y == 1) ? 0 : bi;
      out_d2[at] = bi == ~0ull ? __builtin_inf() : b;
    }
    __syncthreads();
  }
}

__global__ void nearest_merge_kernel(const uint64_t *__restrict__ part_idx, const double *__restrict__ part_d2, uint64_t n, uint32_t slices,
                                     uint64_t *__restrict__ out_idx, double *__restrict__ out_d2) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  double b = __builtin_inf();
  uint64_t bi = ~0ull;
  for (uint32_t s = 0; s < slices; ++s) {
    const double v = part_d2[(uint64_t)s * n + r];
    const uint64_t vi = part_idx[(uint64_t)s * n + r];
    if (vi != ~0ull && (v < b || (v == b && vi < bi))) {
      b = v;
      bi = vi;
    }
  }
  out_idx[r] = bi == ~0ull ? 0 : bi;
  out_d2[r] = bi == ~0ull ? __builtin_inf() : b;
}


// ---- keys on the matrix pipe, exact re-rank of the few centroids that can be the nearest ---------------
// |q - p|^2 = |q|^2 + (|p|^2 - 2 q.p): euclid_keys_kernel writes the approximate key  |p|^2 - 2 q.p  of every (row, centroid)
// of the batch from an f32 MFMA contraction of the raw rows rounded to f32 (the packed images of mfma_topk_f32.hip, norms ==
// nullptr; one wave = 32 rows x 128 centroids, fragments straight from L2, no LDS).  With E bounding |approximate key - true
// key| every centroid whose REFERENCE distance can be the smallest has its key within 2 E of the row's best key, so: a row with
// at most kEuCap keys in that window is decided by summing those centroids' distances in the reference's arithmetic (d2 += (a -
// b) * (a - b), features ascending, two roundings) -- the minimum, smallest index among equal minima, is exactly what the sweep
// over all centroids returns.  Every other row (many centroids tied, non-finite values, keys that overflow f32) goes to the
// exact kernel by a row list whose length stays on the device.
//   E = ((2 d + 19) 2^-23 Qmax^2) 1.05:  operands rounded to f32 (2^-24 each), d products summed in f32 by the matrix pipe
//   (2^-23 per addition covers a truncating adder), |p|^2 rounded to f32, the final fma; |q|, |p| <= Qmax (a centroid is a
//   running mean of rows); the reference's own rounding ((d + 2) 2^-53 relative) is inside the factor 1.05.
constexpr int kEuCap = 4;
typedef float v16f32e __attribute__((ext_vector_type(16)));

// The centroids as the B image of the contraction, P[((T KS2 + ks) 4 + s) 64 + l] = (float) c[128 T + 32 s + (l & 31)][(l >> 5) KS2 + ks]
// (the tile shape of mfma_topk_f32.hip with the features dealt in two contiguous halves, so that the A side reads each row in
// 16-byte pieces; zero beyond the last centroid and the last feature), their |c|^2 rounded to f32, and the reset of the
// leftover-row counter: one wave per row of the padded image.
__global__ __launch_bounds__(256) void euclid_pack_cents_kernel(const double *__restrict__ C, uint32_t n, uint32_t d, uint32_t KS2, uint32_t rows_padded,
                                                                float *__restrict__ P, float *__restrict__ bsq, uint32_t *__restrict__ redo_count) {
  const int lane = threadIdx.x & 63;
  const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row == 0 && lane == 0) *redo_count = 0;
  if (row >= rows_padded) return;
  const uint32_t T = row >> 7, sg = (row & 127) >> 5, lc = row & 31;
  float *dst = P + ((size_t)T * KS2 * 4 + sg) * 64 + lc;
  double sq = 0.0;
  for (uint32_t k = lane; k < 2 * KS2; k += 64) {
    const double v = (row < n && k < d) ? C[(size_t)row * d + k] : 0.0;
    sq = fma(v, v, sq);
    dst[(size_t)(k % KS2) * 256 + 32 * (k / KS2)] = (float)v;
  }
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  if (lane == 0 && row < n) bsq[row] = (float)sq;
}

// The same image for MANY rows (Two-NN: every row of the matrix is a point): a workgroup per 128-row tile, panels of 64 k steps
// through a padded LDS transpose, so that rows are read side by side (a row per wave instruction) and the image is written in
// whole 1 KB lines (the wave-per-row form above scatters 4-byte words 1 KB apart: 1.4 ms per million rows against 0.4).
__global__ __launch_bounds__(256) void euclid_pack_tile_kernel(const double *__restrict__ C, uint32_t n, uint32_t d, uint32_t KS2, float *__restrict__ P,
                                                               float *__restrict__ bsq) {
  __shared__ float s_t[64 * 257];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t T = blockIdx.x;
  double sq[32];
#pragma unroll
  for (int r = 0; r < 32; ++r) sq[r] = 0.0;
  for (uint32_t p0 = 0; p0 < KS2; p0 += 64) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      const uint32_t row = T * 128 + (uint32_t)w * 32 + r;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t ks = p0 + lane, k = h * KS2 + ks;
        const double v = (row < n && ks < KS2 && k < d) ? C[(size_t)row * d + k] : 0.0;
        sq[r] = fma(v, v, sq[r]);
        s_t[lane * 257 + w * 64 + r + 32 * h] = (float)v;
      }
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < 64u * 256u; e += 256) {
      const uint32_t ks = e >> 8, at = e & 255u;
      if (p0 + ks < KS2) P[((size_t)T * KS2 + p0 + ks) * 256 + at] = s_t[ks * 257 + at];
    }
  }
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    double v = sq[r];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const uint32_t row = T * 128 + (uint32_t)w * 32 + r;
    if (lane == 0 && row < n) bsq[row] = (float)v;
  }
}

// largest row norm (upper bound: fma sum, rounded up by a relative 1e-12 on the host); positive doubles order as integers
__global__ __launch_bounds__(256) void max_sqnorm_kernel(const double *__restrict__ R, uint64_t n, uint32_t d, unsigned long long *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  double m = 0.0;
  bool bad = false;
  for (uint64_t r = wave; r < n; r += waves) {
    double s = 0.0;
    for (uint32_t k = lane; k < d; k += 64) s = fma(R[r * d + k], R[r * d + k], s);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (!(s <= 1.7e308)) bad = true;  // NaN or inf
    m = fmax(m, s);
  }
  if (lane == 0) atomicMax(out, bad ? 0x7FF0000000000000ull : (unsigned long long)__double_as_longlong(m));
}

// keys[row * ldk + col]: one wave = the 32 rows of group g against the 32 centroids of column group cg; the rows are read as
// they are (f64 -> f32 on the way: lane l holds row g 32 + (l & 31), features (l >> 5) KS2 + ks), the centroids from the packed
// image; all fragments of the wave are requested before the first MFMA (KS2 <= 64: d <= 128) or in panels of 64 k steps.
__global__ __launch_bounds__(64) void euclid_keys_kernel(const double *__restrict__ rows, const float *__restrict__ PB, const float *__restrict__ bsq,
                                                         uint32_t col_groups, uint32_t g_splits, uint32_t a_groups, uint32_t KS2, uint32_t nA, uint32_t nB,
                                                         uint32_t d, uint64_t ldk, float *__restrict__ keys, const float *__restrict__ PA) {
  // blockIdx.x = split * col_groups + cg: the wave keeps the fragments of column group cg and meets the row groups
  // split, split + g_splits, ... (a batch of the clustering: one row group per wave; Two-NN's 16 row groups against a million
  // points: all of them per wave, so that the points are read once)
  const int lane = threadIdx.x, lh = lane >> 5, lc = lane & 31;
  const uint32_t split = blockIdx.x / col_groups, cg = blockIdx.x - split * col_groups;
  const uint32_t J = cg >> 2, ng = cg & 3;
  const float *b_src = PB + ((size_t)J * KS2 * 4 + ng) * 64 + lane;
  const bool pairs = (d & 1u) == 0;  // KS2 is a multiple of 16: every pair of features of a lane is 16-byte aligned then
  typedef double d2e __attribute__((ext_vector_type(2)));
  const uint32_t col = cg * 32 + lc;
  const float nb = col < nB ? bsq[col] : 0.f;
  if (KS2 <= 64) {
    float b[64];
#pragma unroll
    for (int u = 0; u < 64; ++u) b[u] = (uint32_t)u < KS2 ? b_src[(size_t)u * 256] : 0.f;
    for (uint32_t g = split; g < a_groups; g += g_splits) {
      const uint32_t arow = g * 32 + lc;
      const double *a_src = rows + (size_t)arow * d + (size_t)lh * KS2;
      float a[64];
      if (PA) {  // the rows as a packed f32 image as well (Two-NN's samples: read once per column group, 256 bytes per instruction)
        const float *pa = PA + ((size_t)(g >> 2) * KS2 * 4 + (g & 3)) * 64 + lane;
#pragma unroll
        for (int u = 0; u < 64; ++u) a[u] = (uint32_t)u < KS2 ? pa[(size_t)u * 256] : 0.f;
      } else
#pragma unroll
      for (int u = 0; u < 64; u += 2) {
        const uint32_t k = lh * KS2 + u;
        if (pairs) {
          d2e v = {0.0, 0.0};
          if (arow < nA && k < d) v = *reinterpret_cast<const d2e *>(a_src + u);
          a[u] = (float)v.x;
          a[u + 1] = (float)v.y;
        } else {
          a[u] = (arow < nA && k < d) ? (float)a_src[u] : 0.f;
          a[u + 1] = (arow < nA && k + 1 < d) ? (float)a_src[u + 1] : 0.f;
        }
      }
      v16f32e acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int u = 0; u < 64; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
      if (col < nB) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t row = g * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row < nA) keys[(size_t)row * ldk + col] = fmaf(-2.f, acc[r], nb);
        }
      }
    }
    return;
  }
  for (uint32_t g = split; g < a_groups; g += g_splits) {  // rows wider than 128: panels of 64 k steps, both operands per panel
    const uint32_t arow = g * 32 + lc;
    const double *a_src = rows + (size_t)arow * d + (size_t)lh * KS2;
    v16f32e acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (uint32_t k0 = 0; k0 < KS2; k0 += 64) {
      float a[64], b[64];
#pragma unroll
      for (int u = 0; u < 64; u += 2) {
        const uint32_t ks = k0 + u, k = lh * KS2 + ks;
        if (pairs) {
          d2e v = {0.0, 0.0};
          if (arow < nA && k < d) v = *reinterpret_cast<const d2e *>(a_src + ks);
          a[u] = (float)v.x;
          a[u + 1] = (float)v.y;
        } else {
          a[u] = (arow < nA && k < d) ? (float)a_src[ks] : 0.f;
          a[u + 1] = (arow < nA && k + 1 < d) ? (float)a_src[ks + 1] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 64; ++u) b[u] = k0 + u < KS2 ? b_src[(size_t)(k0 + u) * 256] : 0.f;
#pragma unroll
      for (int u = 0; u < 64; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
    if (col < nB) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t row = g * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < nA) keys[(size_t)row * ldk + col] = fmaf(-2.f, acc[r], nb);
      }
    }
  }
}

// Two-NN (src/clustering.rs:117-147) from the same keys: per sampled row (one workgroup) the three smallest keys over all
// points give the window k3 + 2 E that holds every point whose reference distance can be among the two smallest to OTHER
// rows (the 2nd smallest key among the others is at most the 3rd overall); the points inside are measured in the reference's
// arithmetic and the two smallest values kept.  A window with more than kTnList points (hundreds of duplicates of the
// sampled row) raises `overflow`: the caller then runs the exact kernel for all samples.
constexpr uint32_t kTnList = 512, kTnThreads = 1024;
// per-point error of a key: c_dot |q| |p| (operands rounded to f32, d products summed by the matrix pipe) + c_sq |p|^2 (|p|^2
// rounded to f32, the final fma); both factors carry the 1.05 of E above
struct TnBound {
  float c_dot, c_sq;
};
__global__ __launch_bounds__(kTnThreads) void two_nn_select_kernel(const double *__restrict__ rows, uint64_t n, uint32_t d, const double *__restrict__ samples,
                                                                   const uint64_t *__restrict__ sample_ids, uint32_t ns, const float *__restrict__ keys,
                                                                   const float *__restrict__ bsq, uint64_t ldk, TnBound bnd, double *__restrict__ out,
                                                                   uint32_t *__restrict__ redo /* [0]: count, [1 ..]: samples */) {
  __shared__ float s_k[kTnThreads][3];
  __shared__ uint32_t s_list[kTnList];
  __shared__ uint32_t s_cnt;
  __shared__ double s_b[kTnThreads][2];
  __shared__ float s_w, s_qn;
  const uint32_t s = blockIdx.x, tid = threadIdx.x;
  if (s >= ns) return;
  const float *kr = keys + (size_t)s * ldk;
  const double *a = samples + (size_t)s * d;
  if (tid < 64) {  // |q| rounded up
    double q2 = 0.0;
    for (uint32_t k = tid; k < d; k += 64) q2 = fma(a[k], a[k], q2);
    for (int o = 32; o > 0; o >>= 1) q2 += __shfl_xor(q2, o, 64);
    if (tid == 0) s_qn = (float)(sqrt(q2) * (1.0 + 1e-6));
  }
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  const float qn = s_qn;
  auto err = [&](float b2) -> float { return fmaf(bnd.c_dot * qn, sqrtf(b2) * 1.000001f, bnd.c_sq * b2); };
  // the three smallest UPPER ends key + e over all points
  float k1 = __builtin_inff(), k2 = k1, k3 = k1;
  auto push = [&](float k) {
    if (k < k3) {
      if (k < k2) {
        k3 = k2;
        if (k < k1) {
          k2 = k1;
          k1 = k;
        } else {
          k2 = k;
        }
      } else {
        k3 = k;
      }
    }
  };
  const uint64_t n4 = n & ~3ull;
  for (uint64_t c = (uint64_t)tid * 4; c < n4; c += (uint64_t)kTnThreads * 4) {
    const float4 kv = *reinterpret_cast<const float4 *>(kr + c), bv = *reinterpret_cast<const float4 *>(bsq + c);
    push(kv.x + err(bv.x));
    push(kv.y + err(bv.y));
    push(kv.z + err(bv.z));
    push(kv.w + err(bv.w));
  }
  for (uint64_t c = n4 + tid; c < n; c += kTnThreads) push(kr[c] + err(bsq[c]));
  s_k[tid][0] = k1;
  s_k[tid][1] = k2;
  s_k[tid][2] = k3;
  __syncthreads();
  if (tid < 64) {  // 1024 triples -> 64 -> 1
    k1 = k2 = k3 = __builtin_inff();
    for (uint32_t t = tid; t < kTnThreads; t += 64)
      for (int e = 0; e < 3; ++e) push(s_k[t][e]);
    // (lane t reads and then overwrites only entries t, t + 64, ...: no other lane's input)
    s_k[tid][0] = k1;
    s_k[tid][1] = k2;
    s_k[tid][2] = k3;
  }
  __syncthreads();
  if (tid == 0) {
    k1 = k2 = k3 = __builtin_inff();
    for (int t = 0; t < 64; ++t)
      for (int e = 0; e < 3; ++e) push(s_k[t][e]);
    s_w = k3;  // (+inf when fewer than three comparable keys: every point is inside)
  }
  __syncthreads();
  const float w = s_w;
  // every point whose LOWER end key - e reaches the window
  for (uint64_t c0 = 0; c0 < n; c0 += kTnThreads) {
    const uint64_t c = c0 + tid;
    if (c < n) {
      const float b2 = bsq[c];
      if (kr[c] - err(b2) <= w) {
        const uint32_t at = atomicAdd(&s_cnt, 1u);
        if (at < kTnList) s_list[at] = (uint32_t)c;
      }
    }
  }
  __syncthreads();
  const uint32_t cnt = s_cnt;
  if (cnt > kTnList) {  // hundreds of points inside the window (duplicates of the sample, an outlier's loose bound): the exact kernel
    if (tid == 0) redo[1 + atomicAdd(redo, 1u)] = s;
    return;
  }
  const uint64_t my = sample_ids[s];
  double b1 = __builtin_inf(), b2v = __builtin_inf();
  for (uint32_t e = tid; e < cnt; e += kTnThreads) {
    const uint64_t c = s_list[e];
    if (c == my) continue;
    const double *b = rows + c * d;
    double v = 0.0;
    for (uint32_t k = 0; k < d; ++k) {
      const double diff = __dsub_rn(a[k], b[k]);
      v = __dadd_rn(v, __dmul_rn(diff, diff));
    }
    if (v < b1) {
      b2v = b1;
      b1 = v;
    } else if (v < b2v) {
      b2v = v;
    }
  }
  s_b[tid][0] = b1;
  s_b[tid][1] = b2v;
  __syncthreads();
  if (tid == 0) {
    double m1 = __builtin_inf(), m2 = __builtin_inf();
    const uint32_t lim = cnt < kTnThreads ? cnt : kTnThreads;
    for (uint32_t t = 0; t < lim; ++t)
      for (int e = 0; e < 2; ++e) {
        const double v = s_b[t][e];
        if (v < m1) {
          m2 = m1;
          m1 = v;
        } else if (v < m2) {
          m2 = v;
        }
      }
    out[2 * (size_t)s] = m1;
    out[2 * (size_t)s + 1] = m2;
  }
}

__global__ void gather_sample_rows_kernel(const double *__restrict__ rows, const uint64_t *__restrict__ ids, uint32_t ns, uint32_t d,
                                          double *__restrict__ out) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (uint64_t)ns * d) return;
  out[e] = rows[ids[e / d] * d + e % d];
}

// one wave per row
__global__ __launch_bounds__(256) void euclid_rerank_kernel(const double *__restrict__ rows, uint64_t n, uint32_t d, const double *__restrict__ cents,
                                                            uint32_t n_cents, const float *__restrict__ keys, uint32_t ldk, float two_e,
                                                            uint64_t *__restrict__ out_idx, double *__restrict__ out_d2,
                                                            uint32_t *__restrict__ redo_rows, uint32_t *__restrict__ redo_count) {
  __shared__ double s_t[4][kEuCap][128];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t row = (uint64_t)blockIdx.x * 4 + wv;
  if (row >= n) return;
  const float *kr = keys + row * ldk;
  float k1 = __builtin_inff();
  bool odd = false;  // a key that is NaN or -inf: the row is not decided here
  for (uint32_t c = lane; c < n_cents; c += 64) {
    const float k = kr[c];
    odd = odd || !(k == k) || k < -3.0e38f;
    k1 = fminf(k1, k);
  }
  for (int o = 32; o > 0; o >>= 1) k1 = fminf(k1, __shfl_xor(k1, o, 64));
  // the window in f32: k <= k1 + 2E evaluated as k - k1 <= 2E (exact or rounded to nearest: two_e carries a relative 2^-20 for it)
  uint32_t cnt = 0;
  uint32_t ci[kEuCap];
#pragma unroll
  for (int c = 0; c < kEuCap; ++c) ci[c] = kNoIdx;
  if (!__any(odd) && k1 < 3.0e38f) {
    for (uint32_t c0 = 0; c0 < n_cents && cnt <= (uint32_t)kEuCap; c0 += 64) {
      const uint32_t c = c0 + lane;
      const bool near = c < n_cents && (kr[c] - k1) <= two_e;
      unsigned long long mask = __ballot(near);
      while (mask && cnt <= (uint32_t)kEuCap) {  // (wave-uniform) centroids in index order
        const int l = __builtin_ctzll(mask);
        mask &= mask - 1;
        if (cnt < (uint32_t)kEuCap) {
#pragma unroll
          for (int q = 0; q < kEuCap; ++q)
            if ((uint32_t)q == cnt) ci[q] = c0 + (uint32_t)l;
        }
        ++cnt;
      }
    }
  }
  if (cnt == 0 || cnt > (uint32_t)kEuCap) {
    if (lane == 0) redo_rows[atomicAdd(redo_count, 1u)] = (uint32_t)row;
    return;
  }
  double acc = 0.0;
  for (uint32_t k0 = 0; k0 < d; k0 += 128) {
    const uint32_t ka = k0 + 2 * lane, kb = ka + 1;
    const double q0 = ka < d ? rows[row * d + ka] : 0.0, q1 = kb < d ? rows[row * d + kb] : 0.0;
#pragma unroll
    for (int c = 0; c < kEuCap; ++c) {
      if ((uint32_t)c < cnt) {
        const double p0 = ka < d ? cents[(uint64_t)ci[c] * d + ka] : 0.0, p1 = kb < d ? cents[(uint64_t)ci[c] * d + kb] : 0.0;
        const double d0 = __dsub_rn(q0, p0), d1 = __dsub_rn(q1, p1);
        s_t[wv][c][2 * lane] = __dmul_rn(d0, d0);
        s_t[wv][c][2 * lane + 1] = __dmul_rn(d1, d1);
      }
    }
    wave_sync();
    if ((uint32_t)lane < cnt) {
      const uint32_t m = d - k0 < 128u ? d - k0 : 128u;
      for (uint32_t j = 0; j < m; ++j) acc = __dadd_rn(acc, s_t[wv][lane][j]);  // the reference's sequential sum
    }
    wave_sync();
  }
  // the candidates in index order with a strict '<' = smallest distance, smallest index among equal ones; NaN never wins
  double best = __builtin_inf();
  uint32_t bi = kNoIdx;
  bool found = false;
#pragma unroll
  for (int c = 0; c < kEuCap; ++c) {
    const double dc = __shfl(acc, c, 64);
    if ((uint32_t)c < cnt && dc < best) {
      best = dc;
      bi = ci[c];
      found = true;
    }
  }
  if (lane == 0) {
    out_idx[row] = found ? bi : 0;
    out_d2[row] = found ? best : __builtin_inf();
  }
}

}  // namespace

// Two-NN on the matrix pipe: (b1, b2)[ns] = the two smallest squared distances of every sampled row to the other rows, the values
// of the exact kernel bit for bit.  AS_ERR_UNSUPPORTED when the shape or the data do not allow it (the caller keeps the exact kernel).
int two_nn_mfma(as_ctx *ctx, const double *rows, uint64_t n, uint64_t d, const uint64_t *sample_ids_dev, uint64_t ns, double *out_dev,
                std::vector<uint32_t> *redo_samples) {
  redo_samples->clear();
  if (!ctx->opt_nearest_mfma || n < 65536 || n >= (1ull << 31) || ns == 0 || ns > 4096 || d < 1 || d > 1024)
    return AS_ERR_UNSUPPORTED;
  if ((double)ns * (double)n * 4.0 > 16e9) return AS_ERR_UNSUPPORTED;  // the key matrix (ns x n floats) is kept whole: 16 GB at most
  double qmax2 = 0.0;
  AS_TRY(max_row_sqnorm(ctx, rows, n, d, &qmax2));
  if (!(qmax2 > 0.0 && qmax2 < 1e30)) return AS_ERR_UNSUPPORTED;  // (non-finite values, or keys that would leave f32)
  ProfScope ps(ctx, "two_nn_mfma");
  const uint32_t b_tiles = (uint32_t)cdiv(n, (uint64_t)kTileRows), KS2 = packed_units((uint32_t)d) * 16;
  const uint64_t ldk = (uint64_t)b_tiles * kTileRows;
  const uint32_t a_groups = (uint32_t)cdiv(ns, (uint64_t)32), col_groups = (uint32_t)cdiv(n, (uint64_t)32);
  const uint32_t a_tiles = (a_groups + 3) / 4;
  DevBuf<float> PB, PA, bsq, asq, keys;
  DevBuf<double> S;
  DevBuf<uint32_t> redo;
  AS_TRY(PB.alloc(ctx, packed_elems(b_tiles, (uint32_t)d)));
  AS_TRY(PA.alloc(ctx, packed_elems(a_tiles, (uint32_t)d)));
  AS_TRY(bsq.alloc(ctx, ldk));
  AS_TRY(asq.alloc(ctx, (size_t)a_tiles * kTileRows));
  AS_TRY(keys.alloc(ctx, (size_t)ns * ldk));
  AS_TRY(S.alloc(ctx, (size_t)a_tiles * kTileRows * d));
  AS_TRY(redo.alloc(ctx, ns + 1));
  AS_HIP(ctx, hipMemsetAsync(S.p, 0, sizeof(double) * (size_t)a_tiles * kTileRows * d, ctx->stream));
  AS_HIP(ctx, hipMemsetAsync(redo.p, 0, sizeof(uint32_t), ctx->stream));
  AS_HIP(ctx, hipMemsetAsync(bsq.p, 0, sizeof(float) * ldk, ctx->stream));
  hipLaunchKernelGGL(gather_sample_rows_kernel, dim3((unsigned)cdiv(ns * d, (uint64_t)256)), dim3(256), 0, ctx->stream, rows, sample_ids_dev, (uint32_t)ns,
                     (uint32_t)d, S.p);
  hipLaunchKernelGGL(euclid_pack_tile_kernel, dim3(b_tiles), dim3(256), 0, ctx->stream, rows, (uint32_t)n, (uint32_t)d, KS2, PB.p, bsq.p);
  hipLaunchKernelGGL(euclid_pack_tile_kernel, dim3(a_tiles), dim3(256), 0, ctx->stream, (const double *)S.p, (uint32_t)ns, (uint32_t)d, KS2, PA.p, asq.p);
  hipLaunchKernelGGL(euclid_keys_kernel, dim3(col_groups), dim3(64), 0, ctx->stream, (const double *)S.p, (const float *)PB.p, (const float *)bsq.p,
                     col_groups, 1u, a_groups, KS2, (uint32_t)ns, (uint32_t)n, (uint32_t)d, ldk, keys.p, KS2 <= 64 ? (const float *)PA.p : nullptr);
  // |key - true key| <= c_dot |q| |p| + c_sq |p|^2 (see E above: (2 d + 2) 2^-23 for the products and their f32 sum, 2^-23 for |p|^2
  // and the fma; the same 1.05, and the comparison itself in f32)
  TnBound bnd{(float)(1.06 * (double)(2 * d + 4) * 0x1p-23), (float)(1.06 * 4.0 * 0x1p-23)};
  hipLaunchKernelGGL(two_nn_select_kernel, dim3((unsigned)ns), dim3(kTnThreads), 0, ctx->stream, rows, n, (uint32_t)d, (const double *)S.p, sample_ids_dev,
                     (uint32_t)ns, (const float *)keys.p, (const float *)bsq.p, ldk, bnd, out_dev, redo.p);
  AS_HIP(ctx, hipGetLastError());
  std::vector<uint32_t> h(ns + 1);
  AS_HIP(ctx, copy_d2h(ctx, h.data(), redo.p, sizeof(uint32_t) * (ns + 1)));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  redo_samples->assign(h.begin() + 1, h.begin() + 1 + std::min<uint64_t>(h[0], ns));
  std::sort(redo_samples->begin(), redo_samples->end());
  ctx->stats["two_nn_mfma_overflow"] = (int64_t)redo_samples->size();
  ctx->stats["two_nn_mfma"] = 1;
  return AS_OK;
}

int max_row_sqnorm(as_ctx *ctx, const double *rows, uint64_t n, uint64_t d, double *out) {
  DevBuf<unsigned long long> m;
  AS_TRY(m.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(m.p, 0, sizeof(unsigned long long), ctx->stream));
  if (n) hipLaunchKernelGGL(max_sqnorm_kernel, dim3(1024), dim3(256), 0, ctx->stream, rows, n, (uint32_t)d, m.p);
  AS_HIP(ctx, hipGetLastError());
  unsigned long long h = 0;
  AS_HIP(ctx, copy_d2h(ctx, &h, m.p, sizeof(h)));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double v;
  std::memcpy(&v, &h, sizeof(v));
  *out = v * (1.0 + 1e-12);
  return AS_OK;
}

// qmax2 > 0: an upper bound of |row|^2 and |centroid|^2 -> candidates from the f32 matrix pipe where the shape pays
int nearest_centroids(as_ctx *ctx, const double *rows, uint64_t n, uint64_t d, const double *cents, uint64_t n_cents,
                      uint64_t *out_idx, double *out_d2, double qmax2) {
  if (n == 0) return AS_OK;
  const bool mfma = ctx->opt_nearest_mfma && qmax2 > 0.0 && qmax2 < 1e30 && n >= 1024 && n_cents >= 64 && n < (1ull << 31) &&
                    n_cents <= 16384 && d >= 1 && d <= 1024;
  if (mfma) {
    ProfScope ps(ctx, "nearest_centroids_mfma");
    const uint32_t a_tiles = (uint32_t)cdiv(n, (uint64_t)kTileRows), b_tiles = (uint32_t)cdiv(n_cents, (uint64_t)kTileRows);
    const uint32_t KS2 = packed_units((uint32_t)d) * 16, ldk = b_tiles * kTileRows;
    DevBuf<float> PB, bsq, keys;
    DevBuf<uint32_t> redo;
    AS_TRY(PB.alloc(ctx, packed_elems(b_tiles, (uint32_t)d)));
    AS_TRY(bsq.alloc(ctx, n_cents));
    AS_TRY(keys.alloc(ctx, (size_t)n * ldk));
    AS_TRY(redo.alloc(ctx, n + 1));
    hipLaunchKernelGGL(euclid_pack_cents_kernel, dim3(b_tiles * kTileRows / 4), dim3(256), 0, ctx->stream, cents, (uint32_t)n_cents, (uint32_t)d, KS2,
                       b_tiles * kTileRows, PB.p, bsq.p, redo.p + n);
    const uint32_t a_groups = a_tiles * 4, col_groups = (uint32_t)cdiv(n_cents, (uint64_t)32);
    hipLaunchKernelGGL(euclid_keys_kernel, dim3(a_groups * col_groups), dim3(64), 0, ctx->stream, rows, (const float *)PB.p, (const float *)bsq.p,
                       col_groups, a_groups, a_groups, KS2, (uint32_t)n, (uint32_t)n_cents, (uint32_t)d, (uint64_t)ldk, keys.p, (const float *)nullptr);
    const double two_e = (2.0 * 1.05 * (double)(2 * d + 19) * 0x1p-23 * qmax2 + 1e-30) * (1.0 + 0x1p-18);
    const float two_e32 = std::nextafterf((float)two_e, INFINITY);
    hipLaunchKernelGGL(euclid_rerank_kernel, dim3((unsigned)cdiv(n, (uint64_t)4)), dim3(256), 0, ctx->stream, rows, n, (uint32_t)d, cents,
                       (uint32_t)n_cents, (const float *)keys.p, ldk, two_e32, out_idx, out_d2, redo.p, redo.p + n);
    // the rows that were not decided there: the exact sweep over their list (its length never visits the host)
    hipLaunchKernelGGL(nearest_centroids_kernel, dim3(64), dim3(256), 0, ctx->stream, rows, n, d, cents, n_cents, out_idx, out_d2,
                       (const uint32_t *)redo.p, (const uint32_t *)(redo.p + n));
    AS_HIP(ctx, hipGetLastError());
    if (ctx->opt_debug & 0x200000) {  // diagnostic: how many rows took the exact sweep
      uint32_t h = 0;
      AS_HIP(ctx, copy_d2h(ctx, &h, redo.p + n, sizeof(h)));
      AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
      ctx->stats["nearest_mfma_leftover_rows"] += h;
    }
    ctx->stats["nearest_mfma_calls"] += 1;
    return AS_OK;
  }
  ProfScope ps(ctx, "nearest_centroids");
  const uint64_t tiles = cdiv(n, (uint64_t)kCtRows), chunks = cdiv(n_cents, (uint64_t)kCtCents);
  const unsigned grid = (unsigned)std::min<uint64_t>(tiles, 256 * 8);
  const uint32_t slices = tiles >= 1024 ? 1u : (uint32_t)std::min<uint64_t>(chunks, cdiv((uint64_t)1024, tiles));
  if (slices <= 1) {
    hipLaunchKernelGGL(nearest_centroids_kernel, dim3(grid), dim3(256), 0, ctx->stream, rows, n, d, cents, n_cents, out_idx,
                       out_d2, (const uint32_t *)nullptr, (const uint32_t *)nullptr);
    AS_HIP(ctx, hipGetLastError());
    return AS_OK;
  }
  DevBuf<uint64_t> pi;
  DevBuf<double> pd;
  AS_TRY(pi.alloc(ctx, (size_t)slices * n));
  AS_TRY(pd.alloc(ctx, (size_t)slices * n));
  hipLaunchKernelGGL(nearest_centroids_kernel, dim3(grid, slices), dim3(256), 0, ctx->stream, rows, n, d, cents, n_cents, pi.p, pd.p,
                     (const uint32_t *)nullptr, (const uint32_t *)nullptr);
  hipLaunchKernelGGL(nearest_merge_kernel, dim3((unsigned)cdiv(n, (uint64_t)256)), dim3(256), 0, ctx->stream, (const uint64_t *)pi.p,
                     (const double *)pd.p, n, slices, out_idx, out_d2);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

}  // namespace asx
