// cone_tiles.hip -- in front of the int8 first pass of the item graph's join (mfma_prefilter_i8.hip): which PAIRS OF 128-POSITION
// TILES of the ordered image can hold a pair of rows whose cosine reaches a threshold at all (replaces nothing of the reference
// by itself: part of what stands in for the all-pairs loop of src/laplacian.rs:203-252).
//
// The first pass sweeps every pair of rows with a 32-wide product that bounds their cosine from above, cos(i, j) <= z_i . z_j
// (z: the 31 rotated coordinates and the tail norm, rotate_wide.hip).  On an image ordered by direction 97 % of the pairs of 512-row
// super tiles hold no pair at its threshold (scripts/attic/explore_super_level.py) -- but a bound on a pair of GROUPS from centre and
// radius is useless (it flags 79 %): the slack of both radii floods the gap between thresholds and inter-cluster products.  The
// bound here is one-sided: the rows of tile I sit in a cone of half angle a_I around their mean direction u_I, and the EXACT
// largest product of the rows of tile J with u_I is computed (an n x n/128 product on the matrix pipe: 1/128 of the pass).  On the
// unit sphere angle(i, j) >= angle(u_I, j) - angle(u_I, i), so for every i in I, j in J
//     z_i . z_j <= |z_i| |z_j| cos(max(0, acos(M_JI) - a_I)),   M_JI = max_j  z_j / |z_j| . u_I,
// and the same with the roles swapped; a tile pair is kept iff BOTH bounds reach min(theta) of its rows.  Bench mixture (1 M x
// 128): 6.3 % of the tile pairs are kept (1.3 % hold a pair); rank 64 / 4 096 clusters: 85 % -- how sharp it is depends on the data,
// never whether it holds.  Rounding: unit directions and rows enter the MFMA as f16 (error norm <= 2^-11 each: 1.5e-3 on M covers
// both, their product and the f32 accumulation), cos a_I is lowered by 1e-5, the thresholds by 2e-4 (what the f32 projection
// and the basis contribute, as in the first pass's own delta), |z| enters as the largest norm of any row.
#include <hip/hip_fp16.h>

#include "common.h"
#include "mfma_common.h"

namespace asx {

namespace {

typedef float v16f32 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kCK = 32;   // coordinates (two k steps of the f16 MFMA)
constexpr float kEpsM = 1.5e-3f, kEpsCone = 1e-5f, kEpsTheta = 2e-4f;

// One workgroup of 128 threads per tile: thread t is position T * 128 + t (row perm[position], or the position itself).
// rows_img: the unit directions of the rows in the operand layout of v_mfma_f32_32x32x16_f16 with KS = 2
//   rows_img[((T * 2 + ks) * 4 + s) * 64 + l][j] = f16(dir(128 T + 32 s + (l & 31))[16 ks + 8 (l >> 5) + j]);
// dirs_img: the same layout with "row" = tile index (the mean directions, 128 of them per image tile).
__global__ __launch_bounds__(128) void cone_tile_kernel(const double *__restrict__ Z, const uint32_t *__restrict__ perm, uint32_t n,
                                                        const float *__restrict__ theta, f16x8 *__restrict__ rows_img,
                                                        f16x8 *__restrict__ dirs_img, float *__restrict__ cos_a, float *__restrict__ sin_a,
                                                        float *__restrict__ theta_min, unsigned int *__restrict__ nmax_bits,
                                                        float *__restrict__ mean_out = nullptr) {
  __shared__ float s_u[128][kCK + 1];
  __shared__ float s_m[kCK];
  __shared__ float s_red[128];
  const uint32_t T = blockIdx.x, t = threadIdx.x, pos = T * 128u + t;
  float u[kCK];
  float nrm = 0.f;
  bool live = false;
  if (pos < n) {
    const double *z = Z + (size_t)(perm ? perm[pos] : pos) * kCK;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < kCK; ++k) {
      const double v = z[k];
      u[k] = (float)v;
      s += v * v;
    }
    const double r = __builtin_sqrt(s);
    nrm = (float)(r * (1.0 + 1e-6));  // rounded up
    live = r > 0.0 && r < __builtin_inf();
    const float inv = live ? (float)(1.0 / r) : 0.f;
#pragma unroll
    for (int k = 0; k < kCK; ++k) u[k] = live ? u[k] * inv : 0.f;
  } else {
#pragma unroll
    for (int k = 0; k < kCK; ++k) u[k] = 0.f;
  }
#pragma unroll
  for (int k = 0; k < kCK; ++k) s_u[t][k] = u[k];
  // largest |z| of any row
  {
    float m = live ? nrm : 0.f;
    for (int o = 32; o > 0; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o, 64));
    if ((t & 63) == 0 && m > 0.f) raise_max_u32(nmax_bits, __float_as_uint(m));  // non-negative floats order like their bits
  }
  __syncthreads();
  // mean direction: coordinate k summed over the 128 rows in index order (the same sum in every run)
  if (t < kCK) {
    float s = 0.f;
    for (int r = 0; r < 128; ++r) s += s_u[r][t];
    s_m[t] = s;
  }
  __syncthreads();
  float mm = 0.f;
#pragma unroll
  for (int k = 0; k < kCK; ++k) mm = __builtin_fmaf(s_m[k], s_m[k], mm);
  const float minv = mm > 0.f ? 1.f / __builtin_sqrtf(mm) : 0.f;  // (no direction: a cone of half angle pi below)
  float dotm = 0.f;
#pragma unroll
  for (int k = 0; k < kCK; ++k) dotm = __builtin_fmaf(u[k], s_m[k] * minv, dotm);
  // cos of the cone's half angle = the smallest product of a live row with the mean direction; smallest finite threshold
  s_red[t] = live ? dotm : 2.f;
  __syncthreads();
  for (int o = 64; o > 0; o >>= 1) {
    if (t < (uint32_t)o) s_red[t] = __builtin_fminf(s_red[t], s_red[t + o]);
    __syncthreads();
  }
  const float ca_raw = s_red[0];
  __syncthreads();
  const float th = theta[pos];  // (padded to whole super tiles with +inf)
  s_red[t] = (pos < n && th < __builtin_inff()) ? th : __builtin_inff();
  __syncthreads();
  for (int o = 64; o > 0; o >>= 1) {
    if (t < (uint32_t)o) s_red[t] = __builtin_fminf(s_red[t], s_red[t + o]);
    __syncthreads();
  }
  if (mean_out && t < kCK) mean_out[(size_t)T * kCK + t] = s_m[t] * minv;  // (the unit mean direction, or zeros: cone_group_kernel)
  if (t == 0) {
    float ca = (minv > 0.f && ca_raw <= 1.5f) ? __builtin_fminf(__builtin_fmaxf(ca_raw - kEpsCone, -1.f), 1.f) : -1.f;
    if (ca_raw > 1.5f) ca = 1.f;  // no live row: nothing to bound (its thresholds are +inf as well)
    cos_a[T] = ca;
    sin_a[T] = __builtin_sqrtf(__builtin_fmaxf(0.f, 1.f - ca * ca)) + 1e-6f;
    theta_min[T] = s_red[0];
  }
  // the rows' directions in operand order: thread t = row 32 s + r of the tile
  {
    const uint32_t s = t >> 5, r = t & 31u;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (_Float16)u[16 * ks + 8 * h + j];
        rows_img[(((size_t)T * 2 + ks) * 4 + s) * 64 + r + 32 * h] = v;
      }
  }
  // the tile's mean direction as "row" T of the direction image
  if (t < 4) {
    const uint32_t ks = t >> 1, h = t & 1u, Td = T >> 7, s = (T & 127u) >> 5, r = T & 31u;
    f16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (_Float16)(s_m[16 * ks + 8 * h + j] * minv);
    dirs_img[(((size_t)Td * 2 + ks) * 4 + s) * 64 + r + 32 * h] = v;
  }
}

__device__ __forceinline__ float max16f(const v16f32 &a) {
  float m = __builtin_fmaxf(__builtin_fmaxf(a[0], a[1]), a[2]);
  m = __builtin_fmaxf(__builtin_fmaxf(m, a[3]), a[4]);
  m = __builtin_fmaxf(__builtin_fmaxf(m, a[5]), a[6]);
  m = __builtin_fmaxf(__builtin_fmaxf(m, a[7]), a[8]);
  m = __builtin_fmaxf(__builtin_fmaxf(m, a[9]), a[10]);
  m = __builtin_fmaxf(__builtin_fmaxf(m, a[11]), a[12]);
  m = __builtin_fmaxf(__builtin_fmaxf(m, a[13]), a[14]);
  return __builtin_fmaxf(m, a[15]);
}

// ---- a cone around 32 tiles (round 5): the coarse level of the n^2 pass ----------------------------------------------------------
// The pass below meets every tile's rows with every tile's direction: 8 MFMAs per pair of tiles, 9.4 ms at 4 M rows where it
// keeps 2.7 % of the pairs.  The 32 tiles of a direction group (4 096 positions of the ordered image: a cluster or two) lie in a
// cone of their own: direction M = the normalised sum of their mean directions, half angle = the largest (angle(M, m_T) + a_T)
// over its tiles (the triangle inequality on the sphere: every row of tile T lies within a_T of m_T).  cone_max_kernel runs once
// against these directions (1/32 of the products) and then skips, tile by tile, the groups whose cone cannot reach the tile's rows --
// a bound on a superset of the rows bounds each of them, so the bits it leaves out are bits the fine test could only have set for
// pairs that cannot reach a threshold either.  One workgroup of 64 threads per group; lane l < 32 = tile 32 G + l.
__global__ __launch_bounds__(64) void cone_group_kernel(const float *__restrict__ mean, const float *__restrict__ cos_a, const float *__restrict__ sin_a,
                                                        const float *__restrict__ theta_min, uint32_t ntiles, f16x8 *__restrict__ dirs_img,
                                                        float *__restrict__ cos_g, float *__restrict__ sin_g, float *__restrict__ theta_g) {
  __shared__ float s_mean[32][kCK + 1];
  __shared__ float s_M[kCK];
  const uint32_t G = blockIdx.x, l = threadIdx.x, T = G * 32u + (l & 31u);
  const bool tile = l < 32u && T < ntiles;
  float m[kCK];
#pragma unroll
  for (int k = 0; k < kCK; ++k) m[k] = tile ? mean[(size_t)T * kCK + k] : 0.f;
  if (l < 32u) {
#pragma unroll
    for (int k = 0; k < kCK; ++k) s_mean[l][k] = m[k];
  }
  __syncthreads();
  if (l < (uint32_t)kCK) {  // coordinate l summed over the tiles in index order (the same sum in every run)
    float sum = 0.f;
    for (int t = 0; t < 32; ++t) sum += s_mean[t][l];
    s_M[l] = sum;
  }
  __syncthreads();
  float mm = 0.f;
#pragma unroll
  for (int k = 0; k < kCK; ++k) mm = __builtin_fmaf(s_M[k], s_M[k], mm);
  const float minv = mm > 0.f ? 1.f / __builtin_sqrtf(mm) : 0.f;
  float c = 2.f, th = __builtin_inff();  // (2: a tile that does not exist or has no live row says nothing)
  if (tile) {
    th = theta_min[T];
    const float ca = cos_a[T], sa = sin_a[T];
    // every tile with a live row widens the group's cone, whatever its thresholds: cone_max_kernel's fine test bounds a pair with
    // min(theta_I, theta_J), so a tile whose own thresholds are all +inf can still be matched through its partner's finite one,
    // and a group cone that left its rows out would let the coarse level clear that pair's bit.  (ca = 1 exactly is cone_tile_kernel's
    // mark for "no live row": a live row gives ca <= 1 - kEpsCone.)  th only feeds theta_g.
    if (ca < 1.f) {
      float cb = 0.f;
#pragma unroll
      for (int k = 0; k < kCK; ++k) cb = __builtin_fmaf(m[k], s_M[k] * minv, cb);
      cb = __builtin_fminf(__builtin_fmaxf(cb - kEpsCone, -1.f), 1.f);  // angle(M, m_T), rounded up
      const float sb = __builtin_sqrtf(__builtin_fmaxf(0.f, 1.f - cb * cb)) + 1e-6f;
      // cos(angle(M, m_T) + a_T), rounded down; -1 where the two angles add up to pi or more, where the tile has no direction
      // (ca = -1) or the group has none
      c = (minv > 0.f && ca > -1.f && cb > -ca) ? __builtin_fmaxf(cb * ca - sb * sa - kEpsCone, -1.f) : -1.f;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    c = __builtin_fminf(c, __shfl_xor(c, o, 64));
    th = __builtin_fminf(th, __shfl_xor(th, o, 64));
  }
  if (l == 0) {
    const float cg = c > 1.5f ? 1.f : c;  // (no tile with a live row: nothing to bound, theta_g = +inf says so as well)
    cos_g[G] = cg;
    sin_g[G] = __builtin_sqrtf(__builtin_fmaxf(0.f, 1.f - cg * cg)) + 1e-6f;
    theta_g[G] = th;
  }
  if (l < 4u) {  // the group's direction as "row" G of a direction image (the layout cone_tile_kernel writes)
    const uint32_t ks = l >> 1, h = l & 1u, Gd = G >> 7, sdx = (G & 127u) >> 5, r = G & 31u;
    f16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (_Float16)(s_M[16 * ks + 8 * h + j] * minv);
    dirs_img[(((size_t)Gd * 2 + ks) * 4 + sdx) * 64 + r + 32 * h] = v;
  }
}

// bit1[J * words + g], bit b: can a row of tile J and a row of tile I = 32 g + b reach a threshold, by the cone of I and the
// exact products of J's rows with I's direction?  (theta_dir: the smallest thresholds of the directions' tiles -- theta_min itself,
// or those of the groups when the directions are cone_group_kernel's; coarse: that pass's bits, groups it excluded are skipped.)  One wave per PAIR of row tiles of [jt_begin, jt_begin + jt_count) (the direction
// fragments serve both), sweeping the direction groups [dg_begin, dg_begin + dg_count); the fragments of the next group are
// requested before the products of this one.
__global__ __launch_bounds__(256, 4) void cone_max_kernel(const f16x8 *__restrict__ rows_img, const f16x8 *__restrict__ dirs_img,
                                                       uint32_t jt_begin, uint32_t jt_count, uint32_t dg_begin, uint32_t dg_count,
                                                       uint32_t ntiles, const float *__restrict__ cos_a, const float *__restrict__ sin_a,
                                                       const float *__restrict__ theta_min, const unsigned int *__restrict__ nmax_bits,
                                                       uint32_t words, uint32_t *__restrict__ bit1, const float *__restrict__ theta_dir,
                                                       const uint32_t *__restrict__ coarse, uint32_t words_c, uint32_t groups_c) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t wid = blockIdx.x * 4u + (uint32_t)wv;
  if (2u * wid >= jt_count) return;
  const uint32_t J0 = jt_begin + 2u * wid;
  const bool two = 2u * wid + 1u < jt_count;
  const uint32_t J1 = two ? J0 + 1u : J0;
  const int lc = lane & 31;
  f16x8 a[2][4][2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) a[q][s][ks] = rows_img[(((size_t)(q ? J1 : J0) * 2 + ks) * 4 + s) * 64 + lane];
  const float thJ[2] = {theta_min[J0], theta_min[J1]};
  const float nmax = __uint_as_float(*nmax_bits);
  const float inv_n2 = nmax > 0.f ? 1.f / (nmax * nmax) : 1.f;
  v16f32 zero;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;
  auto dir_frag = [&](uint32_t g, int ks) -> f16x8 { return dirs_img[(((size_t)(g >> 2) * 2 + ks) * 4 + (g & 3u)) * 64 + lane]; };
  const uint32_t g_end = dg_begin + dg_count;
  if (dg_count == 0) return;
  // (the cone of direction 32 g + lane is requested a group ahead, with the fragments: a load in the loop that is used in the
  // same trip makes the compiler wait for everything requested before it -- the prefetched fragments too)
  auto dir_cone = [&](uint32_t g, float &th, float &c, float &sn) {
    const uint32_t I = min(g * 32u + (uint32_t)lc, ntiles - 1u);
    th = theta_dir[I];
    c = cos_a[I];
    sn = sin_a[I];
  };
  f16x8 b0 = dir_frag(dg_begin, 0), b1 = dir_frag(dg_begin, 1);
  float thI, ca, sa;
  dir_cone(dg_begin, thI, ca, sa);
  bool have = true;           // b0 / b1 / the cones hold group g's
  uint32_t cw = 0xFFFFFFFFu;  // the coarse pass's bits of these two tiles for the groups 32 (g >> 5) ..
  for (uint32_t g = dg_begin; g < g_end; ++g) {
    if (coarse && ((g & 31u) == 0u || g == dg_begin)) {
      const uint32_t w = g >> 5;
      cw = w < words_c ? (coarse[(size_t)J0 * words_c + w] | coarse[(size_t)J1 * words_c + w]) : 0xFFFFFFFFu;
    }
    if (coarse && g < groups_c && !((cw >> (g & 31u)) & 1u)) {  // no row of either tile reaches the cone around the group's 32 tiles
      if (lane == 0) {
        bit1[(size_t)J0 * words + g] = 0u;
        if (two) bit1[(size_t)J1 * words + g] = 0u;
      }
      have = false;  // (what was requested ahead was this group's: the next group that is not skipped asks for its own)
      continue;
    }
    if (!have) {
      b0 = dir_frag(g, 0);
      b1 = dir_frag(g, 1);
      dir_cone(g, thI, ca, sa);
      have = true;
    }
    const uint32_t gn = g + 1 < g_end ? g + 1 : g;
    const f16x8 n0 = dir_frag(gn, 0), n1 = dir_frag(gn, 1);
    float thN, caN, saN;
    dir_cone(gn, thN, caN, saN);
    const uint32_t I = g * 32u + (uint32_t)lc;
    const bool in = I < ntiles;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float m = -__builtin_inff();
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        v16f32 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q][s][0], b0, zero, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q][s][1], b1, acc, 0, 0, 0);
        m = __builtin_fmaxf(m, max16f(acc));
      }
      m = __builtin_fmaxf(m, __shfl_xor(m, 32, 64));
      bool flag = false;
      const float t = __builtin_fminf(thI, thJ[q]);
      if (in && t < __builtin_inff()) {
        const float tpp = (t - kEpsTheta) * inv_n2;
        if (t <= 1e-3f) {
          flag = true;  // (thresholds at or below zero: rows of norm zero count, no bound on directions says anything)
        } else if (tpp <= 1.f) {
          if (tpp <= -ca) flag = true;  // the cone and the threshold's angle together span the half circle
          else flag = m + kEpsM >= ca * tpp - sa * __builtin_sqrtf(__builtin_fmaxf(0.f, 1.f - tpp * tpp)) - 1e-5f;
        }
      }
      const unsigned long long bal = __ballot(flag && lane < 32);
      if (lane == 0 && (q == 0 || two)) bit1[(size_t)(q ? J1 : J0) * words + g] = (uint32_t)bal;
    }
    b0 = n0;
    b1 = n1;
    thI = thN;
    ca = caN;
    sa = saN;
  }
}

// ---- the same with TWO cones per tile, split where a pivot group ends (round 5) -----------------------------------------------------
// Where a cluster is a few hundred rows (4 096 clusters in 1 M rows: 244) half of the 128-position tiles straddle two clusters.  Under
// this bound two unrelated rows still have z_i . z_j = t_i t_j (0.26 - 0.42 at rank 64 - 128), so the two clusters of such a tile stand
// at 65 - 75 degrees, its cone opens to 40 degrees, and against the rows of another such tile that leaves cos(10 .. 25 degrees) >= the
// thresholds: kept in both orientations -- 0.52^2 of all pairs, the 14 - 34 % measured.  Split at the boundary of the pivot groups inside
// it, a straddling tile is two narrow cones; a pair of tiles is kept iff one of its four pairs of SEGMENTS is.  (Fixed halves do not
// do it: a tile has a straddling half exactly when it straddles.)  Twice the directions against the same rows' image: it runs where
// the bound above keeps more than "join_cone_split_above_e3" (80) per mille.
// split[T] in [1, 127]: the first position of tile T's second segment -- the group boundary inside the tile nearest its middle, 64
// where there is none.
__global__ void cone_split_kernel(const uint32_t *__restrict__ group_start, uint32_t groups, uint32_t n, uint32_t ntiles, uint32_t *__restrict__ split) {
  const uint32_t T = blockIdx.x * blockDim.x + threadIdx.x;
  if (T >= ntiles) return;
  const uint32_t p0 = T * 128u;
  uint32_t lo = 0, hi = groups;  // the first group that starts behind p0
  while (lo < hi) {
    const uint32_t mid = (lo + hi) / 2;
    if (group_start[mid] <= p0) lo = mid + 1;
    else hi = mid;
  }
  uint32_t best = 64, dist = 0xFFFFFFFFu;
  for (uint32_t g = lo; g < groups && group_start[g] < p0 + 128u; ++g) {
    const uint32_t b = group_start[g] - p0;  // 1 .. 127
    const uint32_t dd = b > 64u ? b - 64u : 64u - b;
    if (b >= 1u && b <= 127u && dd < dist) best = b, dist = dd;
  }
  split[T] = best;
}
// One workgroup of 128 threads per segment H = 2 T + a (a = 0: positions [128 T, 128 T + split), 1: the rest): its mean direction as
// "row" H of the direction image, its cone and its smallest threshold.
__global__ __launch_bounds__(128) void cone_segment_kernel(const double *__restrict__ Z, const uint32_t *__restrict__ perm, uint32_t n,
                                                          const float *__restrict__ theta, const uint32_t *__restrict__ split,
                                                          f16x8 *__restrict__ dirs_img, float *__restrict__ cos_a, float *__restrict__ sin_a,
                                                          float *__restrict__ theta_min) {
  __shared__ float s_u[128][kCK + 1];
  __shared__ float s_m[kCK];
  __shared__ float s_red[128];
  const uint32_t H = blockIdx.x, T = H >> 1, a = H & 1u, t = threadIdx.x, pos = T * 128u + t;
  const uint32_t b = split[T];
  const bool mine = a ? t >= b : t < b;
  float u[kCK];
  bool live = false;
  if (pos < n && mine) {
    const double *z = Z + (size_t)(perm ? perm[pos] : pos) * kCK;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < kCK; ++k) {
      const double v = z[k];
      u[k] = (float)v;
      s += v * v;
    }
    const double r = __builtin_sqrt(s);
    live = r > 0.0 && r < __builtin_inf();
    const float inv = live ? (float)(1.0 / r) : 0.f;
#pragma unroll
    for (int k = 0; k < kCK; ++k) u[k] = live ? u[k] * inv : 0.f;
  } else {
#pragma unroll
    for (int k = 0; k < kCK; ++k) u[k] = 0.f;
  }
#pragma unroll
  for (int k = 0; k < kCK; ++k) s_u[t][k] = u[k];
  __syncthreads();
  if (t < kCK) {
    float s = 0.f;
    for (int r = 0; r < 128; ++r) s += s_u[r][t];  // (index order: the same sum in every run; rows outside the segment are zero)
    s_m[t] = s;
  }
  __syncthreads();
  float mm = 0.f;
#pragma unroll
  for (int k = 0; k < kCK; ++k) mm = __builtin_fmaf(s_m[k], s_m[k], mm);
  const float minv = mm > 0.f ? 1.f / __builtin_sqrtf(mm) : 0.f;
  float dotm = 0.f;
#pragma unroll
  for (int k = 0; k < kCK; ++k) dotm = __builtin_fmaf(u[k], s_m[k] * minv, dotm);
  s_red[t] = live ? dotm : 2.f;
  __syncthreads();
  for (int o = 64; o > 0; o >>= 1) {
    if (t < (uint32_t)o) s_red[t] = __builtin_fminf(s_red[t], s_red[t + o]);
    __syncthreads();
  }
  const float ca_raw = s_red[0];
  __syncthreads();
  const float th = theta[pos];  // (padded to whole super tiles with +inf)
  s_red[t] = (pos < n && mine && th < __builtin_inff()) ? th : __builtin_inff();
  __syncthreads();
  for (int o = 64; o > 0; o >>= 1) {
    if (t < (uint32_t)o) s_red[t] = __builtin_fminf(s_red[t], s_red[t + o]);
    __syncthreads();
  }
  if (t == 0) {
    float ca = (minv > 0.f && ca_raw <= 1.5f) ? __builtin_fminf(__builtin_fmaxf(ca_raw - kEpsCone, -1.f), 1.f) : -1.f;
    if (ca_raw > 1.5f) ca = 1.f;  // no live row: nothing to bound (its thresholds are +inf as well)
    cos_a[H] = ca;
    sin_a[H] = __builtin_sqrtf(__builtin_fmaxf(0.f, 1.f - ca * ca)) + 1e-6f;
    theta_min[H] = s_red[0];
  }
  if (t < 4) {
    const uint32_t ks = t >> 1, h = t & 1u, Td = H >> 7, s = (H & 127u) >> 5, r = H & 31u;
    f16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (_Float16)(s_m[16 * ks + 8 * h + j] * minv);
    dirs_img[(((size_t)Td * 2 + ks) * 4 + s) * 64 + r + 32 * h] = v;
  }
}
// bit1[H * words + g], bit b: can a row of SEGMENT H and a row of segment 32 g + b reach a threshold, by the cone of the latter and
// the exact products of H's rows with its direction?  One wave per 128-row tile J: the largest product of each of its four 32-row
// blocks with a direction; segment 2 J takes the blocks that hold one of its rows, segment 2 J + 1 likewise (the block the split
// falls into counts for both: a larger maximum only keeps more).
__global__ __launch_bounds__(256, 4) void cone_max_segment_kernel(const f16x8 *__restrict__ rows_img, const f16x8 *__restrict__ dirs_img, uint32_t ntiles,
                                                                const uint32_t *__restrict__ split, const float *__restrict__ cos_a,
                                                                const float *__restrict__ sin_a, const float *__restrict__ theta_min,
                                                                const unsigned int *__restrict__ nmax_bits, uint32_t words,
                                                                uint32_t *__restrict__ bit1) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t J = blockIdx.x * 4u + (uint32_t)wv;
  if (J >= ntiles) return;
  const uint32_t nseg = 2u * ntiles;
  const int lc = lane & 31;
  f16x8 a[4][2];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) a[s][ks] = rows_img[(((size_t)J * 2 + ks) * 4 + s) * 64 + lane];
  const uint32_t b = split[J];
  const int s_last0 = (int)((b - 1u) >> 5), s_first1 = (int)(b >> 5);  // blocks of the first / the second segment
  const float thJ[2] = {theta_min[2u * J], theta_min[2u * J + 1u]};
  const float nmax = __uint_as_float(*nmax_bits);
  const float inv_n2 = nmax > 0.f ? 1.f / (nmax * nmax) : 1.f;
  v16f32 zero;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;
  auto dir_frag = [&](uint32_t g, int ks) -> f16x8 { return dirs_img[(((size_t)(g >> 2) * 2 + ks) * 4 + (g & 3u)) * 64 + lane]; };
  auto dir_cone = [&](uint32_t g, float &th, float &c, float &sn) {
    const uint32_t I = min(g * 32u + (uint32_t)lc, nseg - 1u);
    th = theta_min[I];
    c = cos_a[I];
    sn = sin_a[I];
  };
  f16x8 b0 = dir_frag(0, 0), b1 = dir_frag(0, 1);
  float thI, ca, sa;
  dir_cone(0, thI, ca, sa);
  for (uint32_t g = 0; g < words; ++g) {
    const uint32_t gn = g + 1 < words ? g + 1 : g;
    const f16x8 n0 = dir_frag(gn, 0), n1 = dir_frag(gn, 1);
    float thN, caN, saN;
    dir_cone(gn, thN, caN, saN);
    const uint32_t I = g * 32u + (uint32_t)lc;
    const bool in = I < nseg;
    float ms[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      v16f32 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][0], b0, zero, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][1], b1, acc, 0, 0, 0);
      const float m = max16f(acc);
      ms[s] = __builtin_fmaxf(m, __shfl_xor(m, 32, 64));
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float m = -__builtin_inff();
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (q == 0 ? s <= s_last0 : s >= s_first1) m = __builtin_fmaxf(m, ms[s]);
      bool flag = false;
      const float t = __builtin_fminf(thI, thJ[q]);
      if (in && t < __builtin_inff()) {
        const float tpp = (t - kEpsTheta) * inv_n2;
        if (t <= 1e-3f) {
          flag = true;
        } else if (tpp <= 1.f) {
          if (tpp <= -ca) flag = true;
          else flag = m + kEpsM >= ca * tpp - sa * __builtin_sqrtf(__builtin_fmaxf(0.f, 1.f - tpp * tpp)) - 1e-5f;
        }
      }
      const unsigned long long bal = __ballot(flag && lane < 32);
      if (lane == 0) bit1[(size_t)(2u * J + (uint32_t)q) * words + g] = (uint32_t)bal;
    }
    b0 = n0;
    b1 = n1;
    thI = thN;
    ca = caN;
    sa = saN;
  }
}
// tile_mask[I * words + w], bit b (pair of 128-position tiles I, J = 32 w + b) from the mask of the segments (rows 2 I, 2 I + 1;
// bits 2 J, 2 J + 1): kept iff one of the four pairs of segments is
__global__ void cone_segments_to_tiles_kernel(const uint32_t *__restrict__ seg_mask, uint32_t ntiles, uint32_t words_s, uint32_t words,
                                              uint32_t *__restrict__ tile_mask, unsigned long long *__restrict__ kept) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t out = 0;
  if (t < (uint64_t)ntiles * words) {
    const uint32_t I = (uint32_t)(t / words), w = (uint32_t)(t % words);
    unsigned long long x = 0;
    for (uint32_t a = 0; a < 2; ++a) {
      const uint32_t H = 2u * I + a;
      const uint32_t lo = 2u * w < words_s ? seg_mask[(size_t)H * words_s + 2u * w] : 0u;
      const uint32_t hi = 2u * w + 1u < words_s ? seg_mask[(size_t)H * words_s + 2u * w + 1u] : 0u;
      x |= ((unsigned long long)hi << 32) | lo;
    }
    const unsigned long long y = (x | (x >> 1)) & 0x5555555555555555ull;  // bit 2 b of y: segment pair (., 2 J) or (., 2 J + 1)
    for (uint32_t bb = 0; bb < 32; ++bb) out |= (uint32_t)((y >> (2u * bb)) & 1ull) << bb;
    if (32u * w >= ntiles) out = 0;
    else if (ntiles - 32u * w < 32u) out &= (1u << (ntiles - 32u * w)) - 1u;
    tile_mask[t] = out;
  }
  uint32_t c = (uint32_t)__builtin_popcount(out);
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(kept + (blockIdx.x & 255u), (unsigned long long)c);
}

// how many tiles have a cone wider than `cos_limit` allows (the bound says nothing about them: a quick way to see that it will not pay)
__global__ void cone_wide_count_kernel(const float *__restrict__ cos_a, uint32_t ntiles, float cos_limit, unsigned int *__restrict__ count) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const bool wide = t < ntiles && cos_a[t] < cos_limit;
  const unsigned long long m = __ballot(wide);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, (unsigned int)__builtin_popcountll(m));
}

// tile_mask[(I - it_begin) * words + w], bit b: the pair (I, J = 32 w + b) is kept iff both orientations keep it:
// bit1[I][w] bit b AND bit1[J][I / 32] bit I % 32 -- the second is a 32 x 32 bit transpose: a wave takes the block (rows 32 w ..
// 32 w + 31 of bit1, word ig), lane l its row's word, and ballot c of "bit c of my word" is the transposed word of I = 32 ig + c.
__global__ __launch_bounds__(256) void cone_and_kernel(const uint32_t *__restrict__ bit1, uint32_t it_begin, uint32_t it_count, uint32_t ntiles,
                                                       uint32_t words, uint32_t ig_begin, uint32_t ig_count, uint32_t *__restrict__ tile_mask,
                                                       unsigned long long *__restrict__ kept) {
  const int lane = threadIdx.x & 63;
  const uint64_t wid = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (uint64_t)ig_count * words) return;
  const uint32_t ig = ig_begin + (uint32_t)(wid / words), w = (uint32_t)(wid % words);
  const uint32_t J = 32u * w + (uint32_t)lane;
  const uint32_t x = (lane < 32 && J < ntiles) ? bit1[(size_t)J * words + ig] : 0u;
  uint32_t mine = 0;
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    const uint32_t t = (uint32_t)__ballot((x >> c) & 1u);
    mine = lane == c ? t : mine;
  }
  const uint32_t I = 32u * ig + (uint32_t)lane;
  uint32_t out = 0;
  if (lane < 32 && I >= it_begin && I < it_begin + it_count) {
    out = bit1[(size_t)I * words + w] & mine;
    tile_mask[(size_t)(I - it_begin) * words + w] = out;
  }
  uint32_t c = (uint32_t)__builtin_popcount(out);
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if (lane == 0 && c) atomicAdd(kept + (blockIdx.x & 255u), (unsigned long long)c);  // (256 counters: one word takes ~90 adds per microsecond)
}

}  // namespace

uint32_t cone_tile_words(uint32_t n) { return (((n + 127u) / 128u) + 31u) / 32u; }

// tile_mask: (tiles of [pos_begin, pos_begin + pos_count)) x cone_tile_words(n) words; bit J of row I - first tile: the pair of
// 128-position tiles (I, J) may hold a pair of rows at a threshold.  Z: n x 32 rotated coordinates by ROW, perm: position -> row
// (or null), theta: thresholds by position padded to whole 512-position super tiles (+inf beyond n).  pos_begin a multiple of 512.
int cone_tile_mask(as_ctx *ctx, const double *Z, const uint32_t *perm, uint32_t n, const float *theta, uint32_t pos_begin,
                   uint32_t pos_count, uint32_t *tile_mask, double *kept_share, ConeDirs *keep) {
  const uint32_t ntiles = (n + 127u) / 128u, words = cone_tile_words(n);
  const uint32_t dir_tiles = (ntiles + 127u) / 128u;  // image tiles of the direction image
  const uint32_t it_begin = pos_begin / 128u, it_count = std::min(ntiles - std::min(ntiles, it_begin), (pos_count + 127u) / 128u);
  ProfScope ps(ctx, "knn_join_cone_tiles");
  ctx->stats["knn_join_cone_split"] = 0;
  ctx->stats["knn_join_cone_groups"] = 0;
  // (keep: the directions, the cones and the largest norm stay with the caller, for the one-sided bound of the second pass)
  ConeDirs local;
  ConeDirs &cd = keep ? *keep : local;
  cd.valid = false;
  DevBuf<unsigned char> rows_img;
  DevBuf<unsigned char> &dirs_img = cd.dirs_img;
  DevBuf<float> &ca = cd.cos_a, &sa = cd.sin_a;
  DevBuf<float> tmin;
  DevBuf<unsigned int> &nmax = cd.nmax;
  DevBuf<uint32_t> bit1;
  DevBuf<unsigned long long> kept;
  AS_TRY(rows_img.alloc(ctx, (size_t)ntiles * 2 * 4 * 64 * 16));
  AS_TRY(dirs_img.alloc(ctx, (size_t)dir_tiles * 2 * 4 * 64 * 16));
  AS_TRY(ca.alloc(ctx, ntiles));
  AS_TRY(sa.alloc(ctx, ntiles));
  AS_TRY(tmin.alloc(ctx, ntiles));
  AS_TRY(nmax.alloc(ctx, 1));
  AS_TRY(bit1.alloc(ctx, (size_t)ntiles * words));
  AS_TRY(kept.alloc(ctx, 256));
  AS_HIP(ctx, hipMemsetAsync(dirs_img.p, 0, (size_t)dir_tiles * 2 * 4 * 64 * 16, ctx->stream));
  AS_HIP(ctx, hipMemsetAsync(nmax.p, 0, sizeof(unsigned int), ctx->stream));
  AS_HIP(ctx, hipMemsetAsync(kept.p, 0, 256 * sizeof(unsigned long long), ctx->stream));
  DevBuf<float> mean;  // the tiles' unit mean directions (the coarse level of the pass: cone_group_kernel)
  if (ctx->opt_join_cone_groups && ntiles >= 2048u) AS_TRY(mean.alloc(ctx, (size_t)ntiles * kCK));
  hipLaunchKernelGGL(cone_tile_kernel, dim3(ntiles), dim3(128), 0, ctx->stream, Z, perm, n, theta, (f16x8 *)rows_img.p, (f16x8 *)dirs_img.p, ca.p,
                     sa.p, tmin.p, nmax.p, mean.p);
  // Pivot groups of fewer than 400 rows (the row order drew further pivots: clusters of a few hundred rows): most tiles straddle two
  // of them, the bound on whole tiles is known to keep a tenth of the pairs or more -- the split form below is taken at once.
  // (its two bit matrices are four times the whole-tile one each: up to 2 GB apiece, i.e. up to ~8 M rows)
  const bool can_split = cd.group_start && cd.groups && it_begin == 0 && it_count == ntiles && ctx->opt_join_cone_split_above_e3 >= 0 && ntiles >= 64 &&
                         (uint64_t)(2u * ntiles) * ((2u * ntiles + 31u) / 32u) * sizeof(uint32_t) <= (2ull << 30);
  const bool split_first = can_split && (uint64_t)n < 400ull * cd.groups;
  if (!split_first) {
    // Data whose tiles are not cones (rows of many clusters in one tile, no directions to tell apart): more than half of the tiles
    // wider than 0.8 rad -- the bound would keep nearly every pair (rank 64 / 4 096 clusters: 93 %), so its n^2 pass is not run and
    // the caller sweeps everything.  (The widths are functions of the gathered coordinates: every rank of a sharded build agrees.)
    DevBuf<unsigned int> wide;
    AS_TRY(wide.alloc(ctx, 1));
    AS_HIP(ctx, hipMemsetAsync(wide.p, 0, sizeof(unsigned int), ctx->stream));
    hipLaunchKernelGGL(cone_wide_count_kernel, dim3(cdiv(ntiles, 256)), dim3(256), 0, ctx->stream, (const float *)ca.p, ntiles, 0.6967f, wide.p);
    unsigned int hw = 0;
    AS_HIP(ctx, copy_d2h(ctx, &hw, wide.p, sizeof(hw)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stats["knn_join_cone_tiles_wide_e6"] = (int64_t)((double)hw / std::max(1u, ntiles) * 1e6);
    if ((uint64_t)hw * 2 > ntiles) {
      ctx->stats["knn_join_cone_tiles_kept_e6"] = 1000000;
      if (kept_share) *kept_share = 1.0;
      // (every pair kept: whoever reads the mask all the same sweeps everything)
      AS_HIP(ctx, hipMemsetAsync(tile_mask, 0xFF, sizeof(uint32_t) * (size_t)it_count * words, ctx->stream));
      return AS_OK;  // (cd.valid stays false: no one-sided bound for the second pass either)
    }
  }
  const uint32_t all_groups = words;
  // two levels from 2 048 tiles on: every tile's rows against the cones around 32 tiles first (cone_group_kernel) -- for the whole
  // range and, since round 6, for a range of row tiles as well (a rank of a sharded build swept its 2 / world of the fine pass
  // without the coarse level: 3.9 ms of a rank's 40 at 4 M rows on 8 ranks, more than the whole pass costs one GPU)
  const uint32_t n_groups = (ntiles + 31u) / 32u, words_c = (n_groups + 31u) / 32u;
  const bool two_level = !split_first && it_count && ctx->opt_join_cone_groups && ntiles >= 2048u && mean.p;
  DevBuf<unsigned char> dirs_g;
  DevBuf<float> ca_g, sa_g, tmin_g;
  DevBuf<uint32_t> bit_g;
  if (split_first) {
  } else if (it_begin == 0 && it_count == ntiles) {
    if (two_level) {
      const size_t dir_bytes = (size_t)((n_groups + 127u) / 128u) * 2 * 4 * 64 * 16;
      AS_TRY(dirs_g.alloc(ctx, dir_bytes));
      AS_TRY(ca_g.alloc(ctx, n_groups));
      AS_TRY(sa_g.alloc(ctx, n_groups));
      AS_TRY(tmin_g.alloc(ctx, n_groups));
      AS_TRY(bit_g.alloc(ctx, (size_t)ntiles * words_c));
      AS_HIP(ctx, hipMemsetAsync(dirs_g.p, 0, dir_bytes, ctx->stream));
      hipLaunchKernelGGL(cone_group_kernel, dim3(n_groups), dim3(64), 0, ctx->stream, (const float *)mean.p, (const float *)ca.p, (const float *)sa.p,
                         (const float *)tmin.p, ntiles, (f16x8 *)dirs_g.p, ca_g.p, sa_g.p, tmin_g.p);
      hipLaunchKernelGGL(cone_max_kernel, dim3(cdiv(ntiles, 8)), dim3(256), 0, ctx->stream, (const f16x8 *)rows_img.p, (const f16x8 *)dirs_g.p, 0u,
                         ntiles, 0u, words_c, n_groups, (const float *)ca_g.p, (const float *)sa_g.p, (const float *)tmin.p,
                         (const unsigned int *)nmax.p, words_c, bit_g.p, (const float *)tmin_g.p, (const uint32_t *)nullptr, 0u, 0u);
    }
    ctx->stats["knn_join_cone_groups"] = two_level ? 1 : 0;
    hipLaunchKernelGGL(cone_max_kernel, dim3(cdiv(ntiles, 8)), dim3(256), 0, ctx->stream, (const f16x8 *)rows_img.p, (const f16x8 *)dirs_img.p, 0u,
                       ntiles, 0u, all_groups, ntiles, (const float *)ca.p, (const float *)sa.p, (const float *)tmin.p,
                       (const unsigned int *)nmax.p, words, bit1.p, (const float *)tmin.p, (const uint32_t *)(two_level ? bit_g.p : nullptr),
                       words_c, n_groups);
  } else if (it_count) {
    // a range of row tiles (a rank of a sharded build): its rows against every direction, and every tile's rows against its directions
    AS_HIP(ctx, hipMemsetAsync(bit1.p, 0, sizeof(uint32_t) * (size_t)ntiles * words, ctx->stream));
    const uint32_t g0 = it_begin / 32u, g1 = (it_begin + it_count + 31u) / 32u;
    if (two_level) {  // (the coarse level for every tile's rows: the first fine pass below sweeps them all; the same bits on every rank)
      const size_t dir_bytes = (size_t)((n_groups + 127u) / 128u) * 2 * 4 * 64 * 16;
      AS_TRY(dirs_g.alloc(ctx, dir_bytes));
      AS_TRY(ca_g.alloc(ctx, n_groups));
      AS_TRY(sa_g.alloc(ctx, n_groups));
      AS_TRY(tmin_g.alloc(ctx, n_groups));
      AS_TRY(bit_g.alloc(ctx, (size_t)ntiles * words_c));
      AS_HIP(ctx, hipMemsetAsync(dirs_g.p, 0, dir_bytes, ctx->stream));
      hipLaunchKernelGGL(cone_group_kernel, dim3(n_groups), dim3(64), 0, ctx->stream, (const float *)mean.p, (const float *)ca.p, (const float *)sa.p,
                         (const float *)tmin.p, ntiles, (f16x8 *)dirs_g.p, ca_g.p, sa_g.p, tmin_g.p);
      hipLaunchKernelGGL(cone_max_kernel, dim3(cdiv(ntiles, 8)), dim3(256), 0, ctx->stream, (const f16x8 *)rows_img.p, (const f16x8 *)dirs_g.p, 0u,
                         ntiles, 0u, words_c, n_groups, (const float *)ca_g.p, (const float *)sa_g.p, (const float *)tmin.p,
                         (const unsigned int *)nmax.p, words_c, bit_g.p, (const float *)tmin_g.p, (const uint32_t *)nullptr, 0u, 0u);
    }
    ctx->stats["knn_join_cone_groups"] = two_level ? 1 : 0;
    const uint32_t *coarse = two_level ? bit_g.p : nullptr;
    hipLaunchKernelGGL(cone_max_kernel, dim3(cdiv(ntiles, 8)), dim3(256), 0, ctx->stream, (const f16x8 *)rows_img.p, (const f16x8 *)dirs_img.p, 0u,
                       ntiles, g0, g1 - g0, ntiles, (const float *)ca.p, (const float *)sa.p, (const float *)tmin.p,
                       (const unsigned int *)nmax.p, words, bit1.p, (const float *)tmin.p, coarse, words_c, n_groups);
    // (the rows of the range against the directions outside it; the groups [g0, g1) were written above, by the same rule)
    if (g0 > 0)
      hipLaunchKernelGGL(cone_max_kernel, dim3(cdiv(it_count, 8)), dim3(256), 0, ctx->stream, (const f16x8 *)rows_img.p, (const f16x8 *)dirs_img.p,
                         it_begin, it_count, 0u, g0, ntiles, (const float *)ca.p, (const float *)sa.p, (const float *)tmin.p,
                         (const unsigned int *)nmax.p, words, bit1.p, (const float *)tmin.p, coarse, words_c, n_groups);
    if (g1 < all_groups)
      hipLaunchKernelGGL(cone_max_kernel, dim3(cdiv(it_count, 8)), dim3(256), 0, ctx->stream, (const f16x8 *)rows_img.p, (const f16x8 *)dirs_img.p,
                         it_begin, it_count, g1, all_groups - g1, ntiles, (const float *)ca.p, (const float *)sa.p, (const float *)tmin.p,
                         (const unsigned int *)nmax.p, words, bit1.p, (const float *)tmin.p, coarse, words_c, n_groups);
  }
  if (it_count && !split_first) {
    const uint32_t ig0 = it_begin / 32u, ig1 = (it_begin + it_count + 31u) / 32u;
    hipLaunchKernelGGL(cone_and_kernel, dim3(cdiv((uint64_t)(ig1 - ig0) * words, 4)), dim3(256), 0, ctx->stream, (const uint32_t *)bit1.p, it_begin,
                       it_count, ntiles, words, ig0, ig1 - ig0, tile_mask, kept.p);
  }
  AS_HIP(ctx, hipGetLastError());
  unsigned long long hk[256], h = 0;
  if (!split_first) {
    AS_HIP(ctx, copy_d2h(ctx, hk, kept.p, sizeof(hk)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (unsigned long long v : hk) h += v;
  }
  double share = it_count ? (double)h / ((double)it_count * (double)ntiles) : 0.0;
  ctx->stats["knn_join_cone_split"] = 0;
  if (split_first || (can_split && share * 1000.0 > (double)ctx->opt_join_cone_split_above_e3)) {
    // two cones per tile, split where a pivot group ends (cone_segment_kernel)
    const uint32_t nseg = 2u * ntiles, words_s = (nseg + 31u) / 32u, dir_tiles_s = (nseg + 127u) / 128u;
    DevBuf<unsigned char> dirs_s;
    DevBuf<float> ca_s, sa_s, tmin_s;
    DevBuf<uint32_t> bit_s, mask_s, split;
    DevBuf<unsigned long long> kept_s;  // (the pairs of segments kept: not reported)
    AS_TRY(dirs_s.alloc(ctx, (size_t)dir_tiles_s * 2 * 4 * 64 * 16));
    AS_TRY(ca_s.alloc(ctx, nseg));
    AS_TRY(sa_s.alloc(ctx, nseg));
    AS_TRY(tmin_s.alloc(ctx, nseg));
    AS_TRY(bit_s.alloc(ctx, (size_t)nseg * words_s));
    AS_TRY(mask_s.alloc(ctx, (size_t)nseg * words_s));
    AS_TRY(split.alloc(ctx, ntiles));
    AS_TRY(kept_s.alloc(ctx, 256));
    AS_HIP(ctx, hipMemsetAsync(dirs_s.p, 0, (size_t)dir_tiles_s * 2 * 4 * 64 * 16, ctx->stream));
    AS_HIP(ctx, hipMemsetAsync(kept.p, 0, 256 * sizeof(unsigned long long), ctx->stream));
    AS_HIP(ctx, hipMemsetAsync(kept_s.p, 0, 256 * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(cone_split_kernel, dim3(cdiv(ntiles, 256)), dim3(256), 0, ctx->stream, cd.group_start, cd.groups, n, ntiles, split.p);
    hipLaunchKernelGGL(cone_segment_kernel, dim3(nseg), dim3(128), 0, ctx->stream, Z, perm, n, theta, (const uint32_t *)split.p, (f16x8 *)dirs_s.p, ca_s.p,
                       sa_s.p, tmin_s.p);
    hipLaunchKernelGGL(cone_max_segment_kernel, dim3(cdiv(ntiles, 4)), dim3(256), 0, ctx->stream, (const f16x8 *)rows_img.p, (const f16x8 *)dirs_s.p, ntiles,
                       (const uint32_t *)split.p, (const float *)ca_s.p, (const float *)sa_s.p, (const float *)tmin_s.p, (const unsigned int *)nmax.p,
                       words_s, bit_s.p);
    hipLaunchKernelGGL(cone_and_kernel, dim3(cdiv((uint64_t)words_s * words_s, 4)), dim3(256), 0, ctx->stream, (const uint32_t *)bit_s.p, 0u, nseg, nseg,
                       words_s, 0u, words_s, mask_s.p, kept_s.p);
    hipLaunchKernelGGL(cone_segments_to_tiles_kernel, dim3(cdiv((uint64_t)ntiles * words, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)mask_s.p, ntiles,
                       words_s, words, tile_mask, kept.p);
    AS_HIP(ctx, hipGetLastError());
    AS_HIP(ctx, copy_d2h(ctx, hk, kept.p, sizeof(hk)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    h = 0;
    for (unsigned long long v : hk) h += v;
    ctx->stats["knn_join_cone_tiles_kept_whole_e6"] = split_first ? -1 : (int64_t)(share * 1e6);
    share = (double)h / ((double)ntiles * (double)ntiles);
    ctx->stats["knn_join_cone_split"] = 1;
  }
  ctx->stats["knn_join_cone_tiles_kept_e6"] = (int64_t)(share * 1e6);
  if (kept_share) *kept_share = share;
  cd.ntiles = ntiles;
  cd.valid = true;
  return AS_OK;
}

// ---- one-sided: gathered rows against the tiles of the image (the second pass of the join) --------------------------------------
// The rows `ids` (nf of them, any rows; theta_a: their thresholds by place in the list, +inf beyond nf, `pad` entries: whole 512-row
// super tiles) are swept against ALL rows by the one-sided form of the f16 kernel, 64 of them per wave.  For a wave's 64 rows and a
// column tile J with direction u_J and cone a_J:  cos(i, j) <= |z_i||z_j| cos(max(0, acos(max_i z_i / |z_i| . u_J) - a_J)): when
// that stays below the smallest threshold of the 64 rows, the wave has nothing to list in J's 128 columns.  mask: what the kernel
// reads (mfma_join_f16.hip, MODE 1, MASKED): a byte per (super tile of 512 listed rows, stage of 256 columns, wave), a bit per
// 32-column block -- here a nibble per tile.  It pays when the listed rows are in the order of the image (neighbours in a wave).
namespace {
__global__ __launch_bounds__(256, 4) void cone_rect_kernel(const f16x8 *__restrict__ rows_img, const f16x8 *__restrict__ dirs_img, uint32_t gt_count,
                                                           uint32_t ntiles, uint32_t words, const float *__restrict__ cos_a,
                                                           const float *__restrict__ sin_a, const float *__restrict__ theta_a,
                                                           const unsigned int *__restrict__ nmax_a, const unsigned int *__restrict__ nmax_b,
                                                           uint32_t *__restrict__ bits) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t g = blockIdx.x * 4u + (uint32_t)wv;
  if (g >= gt_count) return;
  const int lc = lane & 31;
  f16x8 a[4][2];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) a[s][ks] = rows_img[(((size_t)g * 2 + ks) * 4 + s) * 64 + lane];
  float th[2];  // the smallest threshold of each half of the tile (64 rows = a wave of the f16 kernel)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float t = theta_a[(size_t)g * 128 + 64 * h + lane];
    t = t < __builtin_inff() ? t : __builtin_inff();
    for (int o = 32; o > 0; o >>= 1) t = __builtin_fminf(t, __shfl_xor(t, o, 64));
    th[h] = t;
  }
  const float nmax = __builtin_fmaxf(__uint_as_float(*nmax_a), __uint_as_float(*nmax_b));
  const float inv_n2 = nmax > 0.f ? 1.f / (nmax * nmax) : 1.f;
  v16f32 zero;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;
  auto dir_frag = [&](uint32_t gq, int ks) -> f16x8 { return dirs_img[(((size_t)(gq >> 2) * 2 + ks) * 4 + (gq & 3u)) * 64 + lane]; };
  auto dir_cone = [&](uint32_t gq, float &c, float &sn) {
    const uint32_t J = min(gq * 32u + (uint32_t)lc, ntiles - 1u);
    c = cos_a[J];
    sn = sin_a[J];
  };
  // (blockIdx.y: a share of the direction groups -- a few thousand open rows are a dozen workgroups, each walking all of them alone
  // took 0.14 ms)
  const uint32_t per = (words + gridDim.y - 1) / gridDim.y, w_lo = min(words, blockIdx.y * per), w_hi = min(words, w_lo + per);
  if (w_lo >= w_hi) return;
  f16x8 b0 = dir_frag(w_lo, 0), b1 = dir_frag(w_lo, 1);
  float ca, sa;
  dir_cone(w_lo, ca, sa);
  for (uint32_t gq = w_lo; gq < w_hi; ++gq) {
    const uint32_t gn = gq + 1 < w_hi ? gq + 1 : gq;
    const f16x8 n0 = dir_frag(gn, 0), n1 = dir_frag(gn, 1);
    float caN, saN;
    dir_cone(gn, caN, saN);
    const bool in = gq * 32u + (uint32_t)lc < ntiles;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float m = -__builtin_inff();
#pragma unroll
      for (int s = 2 * h; s < 2 * h + 2; ++s) {
        v16f32 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][0], b0, zero, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][1], b1, acc, 0, 0, 0);
        m = __builtin_fmaxf(m, max16f(acc));
      }
      m = __builtin_fmaxf(m, __shfl_xor(m, 32, 64));
      bool flag = false;
      const float t = th[h];
      if (in && t < __builtin_inff()) {
        const float tpp = (t - kEpsTheta) * inv_n2;
        if (t <= 1e-3f) flag = true;
        else if (tpp <= 1.f) {
          if (tpp <= -ca) flag = true;
          else flag = m + kEpsM >= ca * tpp - sa * __builtin_sqrtf(__builtin_fmaxf(0.f, 1.f - tpp * tpp)) - 1e-5f;
        }
      }
      const unsigned long long bal = __ballot(flag && lane < 32);
      if (lane == 0) bits[((size_t)2 * g + h) * words + gq] = (uint32_t)bal;
    }
    b0 = n0;
    b1 = n1;
    ca = caN;
    sa = saN;
  }
}

// the byte mask of the one-sided f16 kernel from the bits per (64 listed rows, column tile)
__global__ void cone_rect_expand_kernel(const uint32_t *__restrict__ bits, uint32_t halves, uint32_t ntiles, uint32_t words, uint32_t a_supers,
                                        uint32_t stages, unsigned char *__restrict__ mask, unsigned long long *__restrict__ kept) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t c = 0;
  if (e < (uint64_t)a_supers * stages * 8) {
    const uint32_t w = (uint32_t)(e & 7u), st = (uint32_t)((e >> 3) % stages), Ia = (uint32_t)((e >> 3) / stages);
    const uint32_t hh = 8u * Ia + w;
    unsigned char b = 0;
    if (hh < halves) {
      const uint32_t J0 = 2u * st, J1 = J0 + 1u;
      if (J0 < ntiles && ((bits[(size_t)hh * words + (J0 >> 5)] >> (J0 & 31u)) & 1u)) b |= 0x0F, ++c;
      if (J1 < ntiles && ((bits[(size_t)hh * words + (J1 >> 5)] >> (J1 & 31u)) & 1u)) b |= 0xF0, ++c;
    }
    mask[e] = b;
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(kept + (blockIdx.x & 255u), (unsigned long long)c);
}
}  // namespace

int cone_rect_mask(as_ctx *ctx, const double *Z, const uint32_t *ids, uint32_t nf, const float *theta_a, uint32_t pad, const ConeDirs &cone,
                   uint32_t n, unsigned char *mask, double *kept_share) {
  if (!cone.valid || cone.ntiles != (n + 127u) / 128u) return fail(ctx, AS_ERR_ARG, "cone bound of the second pass: no cones of this image");
  const uint32_t ntiles = cone.ntiles, words = cone_tile_words(n);
  const uint32_t gt = (nf + 127u) / 128u, a_supers = pad / 512u, stages = 2u * ((n + 511u) / 512u);
  ProfScope ps(ctx, "knn_join_cone_second");
  DevBuf<unsigned char> rows_img, dirs_g;
  DevBuf<float> ca_g, sa_g, tmin_g;
  DevBuf<unsigned int> nmax_g;
  DevBuf<uint32_t> bits;
  DevBuf<unsigned long long> kept;
  AS_TRY(rows_img.alloc(ctx, (size_t)gt * 2 * 4 * 64 * 16));
  AS_TRY(dirs_g.alloc(ctx, (size_t)((gt + 127u) / 128u) * 2 * 4 * 64 * 16));
  AS_TRY(ca_g.alloc(ctx, gt));
  AS_TRY(sa_g.alloc(ctx, gt));
  AS_TRY(tmin_g.alloc(ctx, gt));
  AS_TRY(nmax_g.alloc(ctx, 1));
  AS_TRY(bits.alloc(ctx, (size_t)2 * gt * words));
  AS_TRY(kept.alloc(ctx, 256));
  AS_HIP(ctx, hipMemsetAsync(nmax_g.p, 0, sizeof(unsigned int), ctx->stream));
  AS_HIP(ctx, hipMemsetAsync(kept.p, 0, 256 * sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(cone_tile_kernel, dim3(gt), dim3(128), 0, ctx->stream, Z, ids, nf, theta_a, (f16x8 *)rows_img.p, (f16x8 *)dirs_g.p, ca_g.p,
                     sa_g.p, tmin_g.p, nmax_g.p);
  hipLaunchKernelGGL(cone_rect_kernel, dim3(cdiv(gt, 4), std::max(1u, std::min(16u, 512u / std::max(1u, cdiv(gt, 4u))))), dim3(256), 0, ctx->stream, (const f16x8 *)rows_img.p, (const f16x8 *)cone.dirs_img.p, gt,
                     ntiles, words, (const float *)cone.cos_a.p, (const float *)cone.sin_a.p, theta_a, (const unsigned int *)cone.nmax.p,
                     (const unsigned int *)nmax_g.p, bits.p);
  hipLaunchKernelGGL(cone_rect_expand_kernel, dim3(cdiv((uint64_t)a_supers * stages * 8, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)bits.p,
                     2u * gt, ntiles, words, a_supers, stages, mask, kept.p);
  AS_HIP(ctx, hipGetLastError());
  unsigned long long hk[256], h = 0;
  AS_HIP(ctx, copy_d2h(ctx, hk, kept.p, sizeof(hk)));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (unsigned long long v : hk) h += v;
  const double share = gt ? (double)h / ((double)2 * gt * ntiles) : 0.0;
  ctx->stats["knn_join_cone_second_kept_e6"] = (int64_t)(share * 1e6);
  if (kept_share) *kept_share = share;
  return AS_OK;
}

}  // namespace asx
