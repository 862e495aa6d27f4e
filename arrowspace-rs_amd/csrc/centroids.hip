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
#include "common.h"

namespace asx {

namespace {

constexpr int kCtRows = 64, kCtCents = 64, kCtFeat = 32, kCtPitch = kCtFeat + 1;

__global__ __launch_bounds__(256) void nearest_centroids_kernel(const double *__restrict__ rows, uint64_t n, uint64_t d,
                                                                const double *__restrict__ cents, uint64_t n_cents,
                                                                uint64_t *__restrict__ out_idx, double *__restrict__ out_d2) {
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
          s_row[rr * kCtPitch + kk] = (row0 + rr < n && kk < feats) ? rows[(row0 + rr) * d + k0 + kk] : 0.0;
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
      const uint64_t at = (uint64_t)blockIdx.y * n + row0 + r;
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

}  // namespace

int nearest_centroids(as_ctx *ctx, const double *rows, uint64_t n, uint64_t d, const double *cents, uint64_t n_cents,
                      uint64_t *out_idx, double *out_d2) {
  if (n == 0) return AS_OK;
  ProfScope ps(ctx, "nearest_centroids");
  const uint64_t tiles = cdiv(n, (uint64_t)kCtRows), chunks = cdiv(n_cents, (uint64_t)kCtCents);
  const unsigned grid = (unsigned)std::min<uint64_t>(tiles, 256 * 8);
  const uint32_t slices = tiles >= 1024 ? 1u : (uint32_t)std::min<uint64_t>(chunks, cdiv((uint64_t)1024, tiles));
  if (slices <= 1) {
    hipLaunchKernelGGL(nearest_centroids_kernel, dim3(grid), dim3(256), 0, ctx->stream, rows, n, d, cents, n_cents, out_idx,
                       out_d2);
    AS_HIP(ctx, hipGetLastError());
    return AS_OK;
  }
  DevBuf<uint64_t> pi;
  DevBuf<double> pd;
  AS_TRY(pi.alloc(ctx, (size_t)slices * n));
  AS_TRY(pd.alloc(ctx, (size_t)slices * n));
  hipLaunchKernelGGL(nearest_centroids_kernel, dim3(grid, slices), dim3(256), 0, ctx->stream, rows, n, d, cents, n_cents, pi.p, pd.p);
  hipLaunchKernelGGL(nearest_merge_kernel, dim3((unsigned)cdiv(n, (uint64_t)256)), dim3(256), 0, ctx->stream, (const uint64_t *)pi.p,
                     (const double *)pd.p, n, slices, out_idx, out_d2);
  AS_HIP(ctx, hipGetLastError());
  return AS_OK;
}

}  // namespace asx
