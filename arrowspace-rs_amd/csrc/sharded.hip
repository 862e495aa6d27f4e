// sharded.hip -- the row-sharded build behind the C ABI: a communicator (RCCL, loaded at run time, or caller-supplied
// callbacks) and as_build_laplacian_sharded, the sequence of arrowspace_amd/distributed.py::build_item_graph in C++ so
// that a Rust or C++ host gets it without torch (reference loop replaced:
// src/laplacian.rs:203-417 over a row range per GPU).
//
// One process per GPU.  Exchange steps of one build: all-gather of the row shards, all-gather of
// the join thresholds, all-to-all of the candidate lists in compact form (sizes first), all-gather of the finished
// neighbour tables, all-reduce of nnz.  Every rank takes part in every step, with empty buffers when it owns no rows;
// which path runs is decided from rank-invariant inputs only.
//
// The transport is three primitives over device buffers (as_comm_ops).  The RCCL transport issues them as grouped
// ncclSend / ncclRecv on the context's stream (xGMI is point to point: a grouped exchange keeps all seven links busy)
// and ncclAllReduce; librccl is dlopen'ed on first use, so the library itself does not depend on it.  The callback
// transport is what the tests use to run several ranks on one GPU (RCCL refuses two ranks on a device) and what a host
// with its own fabric layer plugs in.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.h"

struct as_comm {
  int rank = 0, world = 1;
  as_comm_ops ops{};
  ncclComm_t nccl = nullptr;  // RCCL transport only
};

namespace asx {

namespace {

// ---- librccl, resolved at run time ----
struct Rccl {
  void *dl = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char *name : {"librccl.so.1", "librccl.so"}) {
      r.dl = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.dl) break;
    }
    if (!r.dl) return;
#define AS_SYM(F) r.F = reinterpret_cast<decltype(r.F)>(dlsym(r.dl, "nccl" #F))
    AS_SYM(GetUniqueId);
    AS_SYM(CommInitRank);
    AS_SYM(CommDestroy);
    AS_SYM(GroupStart);
    AS_SYM(GroupEnd);
    AS_SYM(Send);
    AS_SYM(Recv);
    AS_SYM(AllReduce);
    AS_SYM(GetErrorString);
#undef AS_SYM
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.AllReduce;
  });
  return r;
}

// the three primitives on RCCL (user = the as_comm)
int rccl_all_to_all_v(void *user, const void *send, const uint64_t *send_bytes, const uint64_t *send_offsets, void *recv,
                      const uint64_t *recv_bytes, const uint64_t *recv_offsets, void *stream) {
  as_comm *c = static_cast<as_comm *>(user);
  Rccl &r = rccl();
  if (r.GroupStart() != ncclSuccess) return AS_ERR_HIP;
  bool ok = true;
  for (int p = 0; p < c->world && ok; ++p) {
    // (an in-place exchange: this rank's own part already sits where it belongs)
    if (p == c->rank && send_bytes[p] == recv_bytes[p] &&
        static_cast<const char *>(send) + send_offsets[p] == static_cast<const char *>(recv) + recv_offsets[p])
      continue;
    if (send_bytes[p])
      ok = r.Send(static_cast<const char *>(send) + send_offsets[p], send_bytes[p], ncclInt8, p, c->nccl, (hipStream_t)stream) == ncclSuccess;
    if (ok && recv_bytes[p])
      ok = r.Recv(static_cast<char *>(recv) + recv_offsets[p], recv_bytes[p], ncclInt8, p, c->nccl, (hipStream_t)stream) == ncclSuccess;
  }
  // (the group is closed on the error path too: an open group would swallow every later call on this communicator)
  return (r.GroupEnd() == ncclSuccess && ok) ? AS_OK : AS_ERR_HIP;
}
int rccl_all_gather_v(void *user, const void *send, uint64_t send_bytes, void *recv, const uint64_t *recv_bytes,
                      const uint64_t *recv_offsets, void *stream) {
  as_comm *c = static_cast<as_comm *>(user);
  std::vector<uint64_t> sb((size_t)c->world, send_bytes), so((size_t)c->world, 0);
  return rccl_all_to_all_v(user, send, sb.data(), so.data(), recv, recv_bytes, recv_offsets, stream);
}
int rccl_all_reduce_sum_u64(void *user, uint64_t *buf, uint64_t count, void *stream) {
  as_comm *c = static_cast<as_comm *>(user);
  return rccl().AllReduce(buf, buf, count, ncclUint64, ncclSum, c->nccl, (hipStream_t)stream) == ncclSuccess ? AS_OK : AS_ERR_HIP;
}

void shard_range(uint64_t n, int rank, int world, uint64_t *lo, uint64_t *hi) {
  uint64_t per = (n + (uint64_t)world - 1) / (uint64_t)world;
  per = (per + 511) / 512 * 512;  // shard starts are multiples of 512: the row super tile of the join
  *lo = std::min<uint64_t>(n, (uint64_t)rank * per);
  *hi = std::min<uint64_t>(n, *lo + per);
}

// all-gather of row blocks: rank r contributes rows [lo_r, hi_r) x row_bytes, every rank ends with all n rows
int gather_rows(as_ctx *ctx, as_comm *c, const void *mine, void *all, uint64_t n, uint64_t row_bytes, hipStream_t stream = nullptr) {
  std::vector<uint64_t> rb((size_t)c->world), ro((size_t)c->world);
  uint64_t my = 0;
  for (int r = 0; r < c->world; ++r) {
    uint64_t lo, hi;
    shard_range(n, r, c->world, &lo, &hi);
    rb[(size_t)r] = (hi - lo) * row_bytes;
    ro[(size_t)r] = lo * row_bytes;
    if (r == c->rank) my = rb[(size_t)r];
  }
  const int rc = c->ops.all_gather_v(c->ops.user, mine, my, all, rb.data(), ro.data(), stream ? stream : ctx->stream);
  return rc == AS_OK ? AS_OK : fail(ctx, rc, "sharded build: all-gather failed");
}

// ---- every rank leaves a sharded build together ---------------------------------------------------------------------------------
// A rank that fails locally (an allocation, a kernel, an unsupported shape found late) returns early -- and its peers would wait
// for it in the next collective.  So every data collective of the build is preceded by a one-word all-reduce of "has anyone
// failed": the ranks that are well add 0 and go on, a failing rank adds 1 ONCE on its way out, which is the agreement its peers
// are about to enter, and they return AS_ERR_PEER.  (Option "sharded_agreement", default 1: ~10 all-reduces of 8 bytes per build.
// A transport that fails inside a collective is not covered: that is the transport's own error path.)
struct Agreement {
  as_ctx *ctx;
  as_comm *comm;
  bool enabled;
  bool settled = false;  // an agreement came back non-zero (or could not be run): every rank is leaving, nothing more to announce
  DevBuf<uint64_t> word;
  int run(uint64_t mine, uint64_t *sum) {
    if (!word.p && word.alloc(ctx, 1) != AS_OK) return AS_ERR_HIP;
    if (copy_h2d(ctx, word.p, &mine, sizeof(mine)) != hipSuccess) return AS_ERR_HIP;
    if (comm->ops.all_reduce_sum_u64(comm->ops.user, word.p, 1, ctx->stream) != AS_OK) return AS_ERR_HIP;
    if (copy_d2h(ctx, sum, word.p, sizeof(*sum)) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return AS_ERR_HIP;
    return AS_OK;
  }
  int before_collective() {  // on a rank that is well
    if (!enabled) return AS_OK;
    uint64_t sum = 0;
    if (run(0, &sum) != AS_OK) {
      settled = true;
      return fail(ctx, AS_ERR_HIP, "sharded build: the status agreement failed");
    }
    if (sum == 0) return AS_OK;
    settled = true;
    return fail(ctx, AS_ERR_PEER, "sharded build: " + std::to_string(sum) + " other rank(s) failed; this rank stops with them");
  }
  void announce_failure() {  // on a rank that failed locally, once
    if (!enabled || settled) return;
    settled = true;
    uint64_t sum = 0;
    (void)run(1, &sum);
  }
};

// ---- lists of a build sharded by IMAGE POSITIONS ---------------------------------------------------------------------------
// Every rank has prepared the same join (own_join_prepare: order, image, thresholds) and contracted the tile pairs of its own
// range of positions; what it listed for positions of other ranks has crossed the links in compact form.  Here, per local
// position r (one wave): source `me` (this rank's own pairs) becomes list 0, all other sources are appended into list 1, and
// every entry is translated from image position to row id (perm; nullptr: the image is in row order).  A foreign source whose
// list ran over, or a merged list longer than `slots`, is reported as slots + 1 entries: the re-rank takes that as an overflow
// and the row goes on to the second pass.
// cnt_src: W x rows (raw counts); recv: the used slots, source-major; off: exclusive scan of min(cnt, slots) in that order.
__global__ void scale_f64_kernel(double *__restrict__ p, uint64_t n, double f) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = __dmul_rn(p[i], f);
}

__global__ __launch_bounds__(256) void merge_lists_kernel(const uint32_t *__restrict__ cnt_src, const uint2 *__restrict__ recv,
                                                          const uint64_t *__restrict__ off, uint32_t W, uint32_t me, uint32_t rows,
                                                          uint32_t slots, const uint32_t *__restrict__ perm, uint32_t *__restrict__ cnt2,
                                                          uint2 *__restrict__ cand2) {
  const int lane = threadIdx.x & 63;
  for (uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += gridDim.x * 4) {
    uint32_t foreign = 0;
    bool over = false;
    for (uint32_t sidx = 0; sidx < W; ++sidx) {
      const uint32_t raw = cnt_src[(size_t)sidx * rows + r], used = min(raw, slots);
      const uint2 *src = recv + off[(size_t)sidx * rows + r];
      if (sidx == me) {
        for (uint32_t e = lane; e < used; e += 64) {
          uint2 v = src[e];
          v.x = perm ? perm[v.x] : v.x;
          cand2[(size_t)r * slots + e] = v;
        }
        if (lane == 0) cnt2[r] = raw;
      } else {
        over = over || raw > slots;
        for (uint32_t e = lane; e < used; e += 64) {
          if (foreign + e < slots) {
            uint2 v = src[e];
            v.x = perm ? perm[v.x] : v.x;
            cand2[((size_t)rows + r) * slots + foreign + e] = v;
          }
        }
        foreign += used;
      }
    }
    if (lane == 0) cnt2[(size_t)rows + r] = (over || foreign > slots) ? slots + 1 : foreign;
  }
}

// table rows by image position -> by row: out[perm[p]] = in[p] (K entries per row)
__global__ void unpermute_table_kernel(const uint64_t *__restrict__ idx_in, const double *__restrict__ dst_in,
                                       const uint32_t *__restrict__ perm, uint64_t n, uint32_t K, uint64_t *__restrict__ idx_out,
                                       double *__restrict__ dst_out) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * K) return;
  const uint64_t p = t / K, s_ = t - p * K;
  const uint64_t r = perm[p];
  idx_out[r * K + s_] = idx_in[t];
  dst_out[r * K + s_] = dst_in[t];
}
// ... from 32-bit indices (what crosses the links: row ids fit 32 bits, as_index_create / the builders refuse more rows), ~0 stays "none"
__global__ void unpermute_table32_kernel(const uint32_t *__restrict__ idx_in, const double *__restrict__ dst_in,
                                         const uint32_t *__restrict__ perm, uint64_t n, uint32_t K, uint64_t *__restrict__ idx_out,
                                         double *__restrict__ dst_out) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * K) return;
  const uint64_t p = t / K, s_ = t - p * K;
  const uint64_t r = perm[p];
  const uint32_t v = idx_in[t];
  idx_out[r * K + s_] = v == 0xFFFFFFFFu ? ~0ull : (uint64_t)v;
  dst_out[r * K + s_] = dst_in[t];
}
__global__ void invert_perm_u32_kernel(const uint32_t *__restrict__ perm, uint32_t n, uint32_t *__restrict__ inv) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) inv[perm[i]] = i;
}
__global__ void used_slots_kernel(const uint32_t *__restrict__ cnt, uint64_t n, uint32_t slots, uint32_t *__restrict__ used) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) used[i] = min(cnt[i], slots);
}
__global__ __launch_bounds__(256) void pack_lists_kernel(const uint32_t *__restrict__ cnt, uint64_t n, uint32_t slots,
                                                         const uint64_t *__restrict__ offsets, const uint2 *__restrict__ src,
                                                         uint2 *__restrict__ dst) {
  const int lane = threadIdx.x & 63;
  for (uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += (uint64_t)gridDim.x * 4) {
    const uint32_t used = min(cnt[r], slots);
    const uint64_t o = offsets[r];
    for (uint32_t e = lane; e < used; e += 64) dst[o + e] = src[r * slots + e];
  }
}
__global__ void narrow_u64_kernel(const uint64_t *__restrict__ in, uint64_t n, uint32_t *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] == 0xFFFFFFFFFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)in[i];
}
__global__ void widen_u32_kernel(const uint32_t *__restrict__ in, uint64_t n, uint64_t *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] == 0xFFFFFFFFu ? 0xFFFFFFFFFFFFFFFFull : (uint64_t)in[i];
}

// ---- lambda statistics and search of a sharded index -----------------------------------------------------------------------
// min / max / sum of a shard's lambdas: 256 blocks, each a fixed stride through the shard and a fixed tree inside, so the
// partial of a block is the same in every run; the host adds the blocks, and then the ranks, in index order.
__global__ __launch_bounds__(256) void lambda_stats_kernel(const double *__restrict__ lam, uint64_t n, double *__restrict__ part /* gridDim.x x 3 */) {
  __shared__ double s_mn[256], s_mx[256], s_sm[256];
  double mn = __builtin_inf(), mx = -__builtin_inf(), sm = 0.0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    const double v = lam[i];
    mn = v < mn ? v : mn;  // (f64::min / max folds of src/builder.rs:434-445: a NaN never replaces the running value)
    mx = v > mx ? v : mx;
    sm += v;
  }
  s_mn[threadIdx.x] = mn;
  s_mx[threadIdx.x] = mx;
  s_sm[threadIdx.x] = sm;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      s_mn[threadIdx.x] = s_mn[threadIdx.x + o] < s_mn[threadIdx.x] ? s_mn[threadIdx.x + o] : s_mn[threadIdx.x];
      s_mx[threadIdx.x] = s_mx[threadIdx.x + o] > s_mx[threadIdx.x] ? s_mx[threadIdx.x + o] : s_mx[threadIdx.x];
      s_sm[threadIdx.x] += s_sm[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[blockIdx.x * 3 + 0] = s_mn[0];
    part[blockIdx.x * 3 + 1] = s_mx[0];
    part[blockIdx.x * 3 + 2] = s_sm[0];
  }
}

// Global top-k of every query from the per-shard top-k lists (src/core.rs:778-798 sorts ALL items by score with a stable
// sort and truncates: ties keep the lower index first).  gi / gs: W x nq x k, shard s's list of query q sorted by (score desc,
// id asc), ids global and shards in ascending id order, exhausted slots id = UINT64_MAX.  One thread per query walks the W
// heads: the best head by (score desc, id asc) is the next result -- the order the stable sort of the whole table gives.
__global__ void merge_topk_kernel(const uint64_t *__restrict__ gi, const double *__restrict__ gs, uint32_t W, uint64_t nq, uint32_t k,
                                  uint64_t *__restrict__ out_i, double *__restrict__ out_s) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  uint32_t head[64];
  for (uint32_t w = 0; w < W; ++w) head[w] = 0;
  for (uint32_t t = 0; t < k; ++t) {
    int best = -1;
    double bs = 0.0;
    uint64_t bi = 0;
    for (uint32_t w = 0; w < W; ++w) {
      if (head[w] >= k) continue;
      const size_t o = ((size_t)w * nq + q) * k + head[w];
      const uint64_t id = gi[o];
      if (id == 0xFFFFFFFFFFFFFFFFull) continue;
      const double sc = gs[o];
      if (best < 0 || sc > bs || (sc == bs && id < bi)) best = (int)w, bs = sc, bi = id;
    }
    if (best < 0) {
      out_i[q * k + t] = 0xFFFFFFFFFFFFFFFFull;
      out_s[q * k + t] = -__builtin_inf();
    } else {
      out_i[q * k + t] = bi;
      out_s[q * k + t] = bs;
      ++head[best];
    }
  }
}
// nq x ks lists -> nq x k (ks < k: a shard with fewer than k items), the tail exhausted
__global__ void widen_topk_kernel(const uint64_t *__restrict__ si, const double *__restrict__ ss, uint64_t nq, uint32_t ks, uint32_t k,
                                  uint64_t *__restrict__ di, double *__restrict__ ds) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nq * k) return;
  const uint64_t q = t / k;
  const uint32_t c = (uint32_t)(t - q * k);
  di[t] = c < ks ? si[q * ks + c] : 0xFFFFFFFFFFFFFFFFull;
  ds[t] = c < ks ? ss[q * ks + c] : -__builtin_inf();
}
__global__ void fill_empty_topk_kernel(uint64_t *__restrict__ idx, double *__restrict__ sc, uint64_t count) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) {
    idx[i] = 0xFFFFFFFFFFFFFFFFull;
    sc[i] = -__builtin_inf();
  }
}

// self_in_results on a whole table (u64 ids): through the u32 form the one-GPU path uses
int table_self_in_results(as_ctx *ctx, const double *X, const double *norms, uint64_t n, uint64_t dim, uint32_t K, int rectify,
                          uint64_t *idx, double *dst, uint32_t *t32) {
  hipLaunchKernelGGL(narrow_u64_kernel, dim3(cdiv(n * K, 256)), dim3(256), 0, ctx->stream, (const uint64_t *)idx, n * K, t32);
  AS_TRY(apply_self_in_results(ctx, X, norms, dim, 0, n, K, rectify, t32, dst));
  hipLaunchKernelGGL(widen_u32_kernel, dim3(cdiv(n * K, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)t32, n * K, idx);
  AS_HIP(ctx, hipGetLastError());
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return AS_OK;
}

}  // namespace

}  // namespace asx

using namespace asx;

extern "C" {

void as_shard_range(uint64_t n, int rank, int world, uint64_t *lo, uint64_t *hi) {
  uint64_t a = 0, b = 0;
  if (world >= 1 && rank >= 0 && rank < world) shard_range(n, rank, world, &a, &b);
  if (lo) *lo = a;
  if (hi) *hi = b;
}

int as_comm_unique_id(void *id128) {
  if (!id128) return AS_ERR_ARG;
  Rccl &r = rccl();
  if (!r.ok) return AS_ERR_UNSUPPORTED;
  static_assert(sizeof(ncclUniqueId) == 128, "the id crosses the ABI as 128 bytes");
  ncclUniqueId id;
  if (r.GetUniqueId(&id) != ncclSuccess) return AS_ERR_HIP;
  std::memcpy(id128, &id, sizeof(id));
  return AS_OK;
}

int as_comm_create_rccl(as_ctx *ctx, const void *id128, int rank, int world, as_comm **out) {
  if (!ctx || !id128 || !out || world < 1 || rank < 0 || rank >= world) return ctx ? fail(ctx, AS_ERR_ARG, "invalid argument") : AS_ERR_ARG;
  *out = nullptr;
  Rccl &r = rccl();
  if (!r.ok) return fail(ctx, AS_ERR_UNSUPPORTED, "librccl could not be loaded");
  AS_HIP(ctx, hipSetDevice(ctx->device));
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  as_comm *c = new as_comm;
  c->rank = rank;
  c->world = world;
  const ncclResult_t rc = r.CommInitRank(&c->nccl, world, id, rank);
  if (rc != ncclSuccess) {
    delete c;
    return fail(ctx, AS_ERR_HIP, std::string("ncclCommInitRank: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error"));
  }
  c->ops = as_comm_ops{c, rccl_all_gather_v, rccl_all_to_all_v, rccl_all_reduce_sum_u64};
  *out = c;
  return AS_OK;
}

int as_comm_create_custom(as_ctx *ctx, const as_comm_ops *ops, int rank, int world, as_comm **out) {
  if (!ctx || !ops || !out || world < 1 || rank < 0 || rank >= world || !ops->all_gather_v || !ops->all_to_all_v ||
      !ops->all_reduce_sum_u64)
    return ctx ? fail(ctx, AS_ERR_ARG, "invalid argument") : AS_ERR_ARG;
  as_comm *c = new as_comm;
  c->rank = rank;
  c->world = world;
  c->ops = *ops;
  *out = c;
  return AS_OK;
}

void as_comm_destroy(as_comm *c) {
  if (!c) return;
  if (c->nccl) rccl().CommDestroy(c->nccl);
  delete c;
}

int as_comm_rank(const as_comm *c) { return c ? c->rank : -1; }
int as_comm_world(const as_comm *c) { return c ? c->world : 0; }

static int build_laplacian_sharded_body(as_ctx *ctx, as_comm *comm, const double *rows_shard, uint64_t n_total, uint64_t dim,
                                        const as_graph_params *params, as_csr **out_block, uint64_t *nnz_total, asx::Agreement &ag);

int as_build_laplacian_sharded(as_ctx *ctx, as_comm *comm, const double *rows_shard, uint64_t n_total, uint64_t dim,
                               const as_graph_params *params, as_csr **out_block, uint64_t *nnz_total) {
  if (!ctx) return AS_ERR_ARG;
  if (!comm || !params || !out_block) return fail(ctx, AS_ERR_ARG, "null argument");
  *out_block = nullptr;
  // (argument errors are the same on every rank: no agreement needed for them)
  if (n_total < 2 || dim < 2)
    return fail(ctx, AS_ERR_SHAPE, "items should be at least of shape (2,2): (" + std::to_string(n_total) + "," + std::to_string(dim) + ")");
  if (n_total >= 0xFFFFFFFFull || dim >= 0xFFFFFFFFull) return fail(ctx, AS_ERR_UNSUPPORTED, "n_nodes/dim must be < 2^32-1");
  AS_HIP(ctx, hipSetDevice(ctx->device));
  asx::Agreement ag{ctx, comm, comm->world > 1 && ctx->opt_sharded_agreement != 0};
  const int rc = build_laplacian_sharded_body(ctx, comm, rows_shard, n_total, dim, params, out_block, nnz_total, ag);
  if (rc != AS_OK) {
    const std::string msg = ctx->err;  // (the announcement must not replace what went wrong here)
    ag.announce_failure();
    ctx->err = msg;
    if (*out_block) as_csr_free(*out_block);  // a caller that treats rc != 0 as "no outputs" must not leak the block
    *out_block = nullptr;
  }
  return rc;
}

static int build_laplacian_sharded_body(as_ctx *ctx, as_comm *comm, const double *rows_shard, uint64_t n_total, uint64_t dim,
                                        const as_graph_params *params, as_csr **out_block, uint64_t *nnz_total, asx::Agreement &ag) {
  const int W = comm->world, R = comm->rank;
  auto inject = [&](int64_t at) -> int {  // test hook: this rank fails at checkpoint `at`
    return ctx->opt_sharded_inject_failure == at ? fail(ctx, AS_ERR_ARG, "sharded build: injected failure (test hook)") : AS_OK;
  };
  uint64_t lo, hi;
  shard_range(n_total, R, W, &lo, &hi);
  const uint64_t rows_here = hi - lo, K = params->topk + 1, n = n_total;
  if (rows_here && !rows_shard) return fail(ctx, AS_ERR_ARG, "null argument");
  DevBuf<double> staged;  // a host shard is staged once; the exchange works on device buffers
  if (rows_here && !is_device_ptr(rows_shard)) {
    AS_TRY(staged.alloc(ctx, (size_t)rows_here * dim));
    AS_HIP(ctx, copy_h2d(ctx, staged.p, rows_shard, sizeof(double) * rows_here * dim));
    rows_shard = staged.p;
  }

  // ---- all rows on every rank (candidates' rows for the exact re-rank, the exact scans) -------------------------------------
  // The join itself works on images the ranks make of their OWN rows (own_join_prepare with ShardedRows: 392 B per row cross the
  // links instead of the 1 KB of a 128-wide f64 row), so the all-gather of the f64 rows is issued on a second stream behind the
  // preparation and awaited in front of the re-rank: it runs under the first pass and the f16 pass.  Everything that is not the
  // join (small tables, sharded_own_rows = 0) gathers them first, as every build did until round 4.
  DevBuf<double> X;
  AS_TRY(X.alloc(ctx, (size_t)n * dim));
  const bool join = W > 1 && as_knn_join_applies(ctx, n, dim, params);
  const bool rows_late = join && ctx->opt_sharded_own_rows != 0 && dim <= 128;
  struct SideStream {  // the stream of the late all-gather and its two events; drained before it goes
    hipStream_t s = nullptr;
    hipEvent_t start = nullptr, done = nullptr;
    ~SideStream() {
      if (s) (void)hipStreamSynchronize(s);
      if (start) (void)hipEventDestroy(start);
      if (done) (void)hipEventDestroy(done);
      if (s) (void)hipStreamDestroy(s);
    }
  } side;
  bool rows_started = false, rows_awaited = false;
  auto rows_begin = [&]() -> int {
    if (rows_started) return AS_OK;
    AS_TRY(inject(1));
    AS_TRY(ag.before_collective());
    if (rows_late) {  // (the side stream starts where this one stands: X may be a recycled block, the shard may have just been staged)
      AS_HIP(ctx, hipEventRecord(side.start, ctx->stream));
      AS_HIP(ctx, hipStreamWaitEvent(side.s, side.start, 0));
    }
    AS_TRY(gather_rows(ctx, comm, rows_shard, X.p, n, dim * sizeof(double), rows_late ? side.s : nullptr));
    if (rows_late) AS_HIP(ctx, hipEventRecord(side.done, side.s));
    rows_started = true;
    return AS_OK;
  };
  auto rows_all = [&](const double **out) -> int {
    AS_TRY(rows_begin());
    if (rows_late && !rows_awaited) AS_HIP(ctx, hipStreamWaitEvent(ctx->stream, side.done, 0));
    rows_awaited = true;
    if (out) *out = X.p;
    return AS_OK;
  };
  if (rows_late) {
    AS_HIP(ctx, hipStreamCreateWithFlags(&side.s, hipStreamNonBlocking));
    AS_HIP(ctx, hipEventCreateWithFlags(&side.start, hipEventDisableTiming));
    AS_HIP(ctx, hipEventCreateWithFlags(&side.done, hipEventDisableTiming));
  } else {
    AS_TRY(rows_all(nullptr));
  }
  ctx->stats["sharded_rows_late"] = rows_late ? 1 : 0;

  DevBuf<uint64_t> idx, idx_all;
  DevBuf<double> dst, dst_all;
  // (the assembly split between the owner of a row and the owner of a row block: see "tables of all positions" below)
  DevBuf<uint32_t> deg_pos, deg_node, pj, plen, adj_j, adj_len;
  DevBuf<double> pw, adj_w;
  DevBuf<unsigned long long> deg_sum;
  AS_TRY(idx.alloc(ctx, (size_t)std::max<uint64_t>(rows_here, 1) * K));
  AS_TRY(dst.alloc(ctx, (size_t)std::max<uint64_t>(rows_here, 1) * K));
  AS_TRY(idx_all.alloc(ctx, (size_t)n * K));
  AS_TRY(dst_all.alloc(ctx, (size_t)n * K));
  ctx->stats["sharded_join"] = join ? 1 : 0;
  ctx->stats["sharded_by_positions"] = 0;
  if (join) {
    // ---- the one-GPU algorithm, sharded by IMAGE POSITIONS --------------------------------------------------------------
    // Every rank holds all rows, so every rank prepares the SAME join (own_join_prepare: rotated coordinates, row order,
    // f16 / int8 images, thresholds -- functions of the rows alone, every sum in a fixed order).  Rank g then owns the 512-
    // position super tiles [lo_g, hi_g) of the IMAGE: it runs the int8 first pass and the masked f16 pass over their tile
    // pairs only, sends what it listed for other ranks' positions to their owners, and finishes (exact re-rank, second pass,
    // exact scans) the rows perm[lo_g .. hi_g).  The tables come back by position and are put in row order at the end.
    // With the rows grouped by direction nearly all of a position's partners sit in its own or a neighbouring super tile, so
    // the foreign lists are short: the exchange is a few per cent of what the row-order join shipped.
    const uint32_t n32 = (uint32_t)n, d32 = (uint32_t)dim, K32 = (uint32_t)K, lo32 = (uint32_t)lo, here32 = (uint32_t)rows_here;
    DevBuf<double> norms;
    AS_TRY(norms.alloc(ctx, (size_t)n));
    OwnJoin oj;
    // (the thresholds, the one MFMA sweep of the preparation, are computed per rank for its own positions and all-gathered in
    // place: 8 bytes per position; with the rows late so are the norms, the coordinates and the f16 image, by row)
    ShardedRows sr;
    sr.own = rows_shard;
    sr.lo = lo32;
    sr.count = here32;
    sr.rank = R;
    sr.world = W;
    sr.row_lo.resize((size_t)W);
    sr.row_hi.resize((size_t)W);
    for (int g = 0; g < W; ++g) shard_range(n, g, W, &sr.row_lo[(size_t)g], &sr.row_hi[(size_t)g]);
    sr.gather = [&](void *all, size_t elem_bytes) -> int {
      AS_TRY(ag.before_collective());
      return gather_rows(ctx, comm, (const char *)all + lo * elem_bytes, all, n, elem_bytes);
    };
    sr.gather_v = [&](void *all, const uint64_t *bytes, const uint64_t *offsets) -> int {
      AS_TRY(ag.before_collective());
      const int rc = comm->ops.all_gather_v(comm->ops.user, (const char *)all + offsets[(size_t)R], bytes[(size_t)R], all, bytes, offsets, ctx->stream);
      return rc == AS_OK ? AS_OK : fail(ctx, rc, "sharded build: all-gather failed");
    };
    sr.rows_begin = rows_begin;
    sr.rows_all = rows_all;
    if (rows_late) {
      if (rows_here) AS_TRY(row_norms(ctx, rows_shard, rows_here, dim, norms.p + lo));
      AS_TRY(sr.gather(norms.p, sizeof(double)));
    } else {
      AS_TRY(row_norms(ctx, X.p, n, dim, norms.p));
    }
    AS_TRY(own_join_prepare(ctx, rows_late ? nullptr : X.p, norms.p, n32, d32, K32, oj, lo32, here32, &sr));
    // the f64 rows from here on: started now (nothing in front of the re-rank reads them on the rotated first pass -- they cross
    // the links under it), awaited at once where the join still wants them
    AS_TRY(rows_begin());
    const bool pairs_need_rows = !(oj.sharded_rows && oj.wide_pre);
    if (pairs_need_rows) AS_TRY(rows_all(nullptr));
    if (oj.reorder) {
      AS_TRY(oj.inv_own.alloc(ctx, (size_t)n));
      hipLaunchKernelGGL(invert_perm_u32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)oj.perm_own.p, n32, oj.inv_own.p);
    }
    ctx->stats["sharded_by_positions"] = oj.reorder ? 1 : 0;
    const uint32_t slots = oj.cap;
    DevBuf<uint32_t> cnt, cand, used, packed, cnt_src, recv, cnt2, cand2;
    DevBuf<uint64_t> offsets, sizes_d, off_src;
    AS_TRY(cnt.alloc(ctx, (size_t)n));
    AS_TRY(offsets.alloc(ctx, (size_t)n + 1));
    if (rows_here) {
      AS_TRY(cand.alloc(ctx, (size_t)n * slots * 2));
      AS_TRY(own_join_pairs(ctx, pairs_need_rows ? X.p : nullptr, norms.p, n32, d32, lo32, here32, oj, cnt.p, cand.p, /*lists by position*/ true));
    } else {  // nothing contracted here: empty lists for every position
      AS_HIP(ctx, hipMemsetAsync(cnt.p, 0, sizeof(uint32_t) * n, ctx->stream));
    }
    // compact form: the used slots side by side, sized by what was listed (not n x cap)
    AS_TRY(used.alloc(ctx, (size_t)n));
    hipLaunchKernelGGL(used_slots_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)cnt.p, n, slots, used.p);
    AS_TRY(exclusive_scan_u32_u64(ctx, used.p, n, offsets.p));
    std::vector<uint64_t> sb((size_t)W), so((size_t)W), rb((size_t)W), ro((size_t)W), bounds((size_t)W + 1);
    for (int g = 0; g < W; ++g) {
      uint64_t a, b;
      shard_range(n, g, W, &a, &b);
      sb[(size_t)g] = (b - a) * sizeof(uint32_t);
      so[(size_t)g] = a * sizeof(uint32_t);
      rb[(size_t)g] = rows_here * sizeof(uint32_t);
      ro[(size_t)g] = (uint64_t)g * rows_here * sizeof(uint32_t);
      AS_HIP(ctx, copy_d2h(ctx, &bounds[(size_t)g], offsets.p + a, sizeof(uint64_t)));
      if (g == W - 1) AS_HIP(ctx, copy_d2h(ctx, &bounds[(size_t)W], offsets.p + b, sizeof(uint64_t)));
    }
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t listed = bounds[(size_t)W];
    ctx->stats["sharded_listed_slots"] = (int64_t)listed;
    ctx->stats["sharded_foreign_slots"] = (int64_t)(listed - (bounds[(size_t)R + 1] - bounds[(size_t)R]));
    AS_TRY(packed.alloc(ctx, (size_t)std::max<uint64_t>(2 * listed, 2)));
    if (rows_here)
      hipLaunchKernelGGL(pack_lists_kernel, dim3((unsigned)std::min<uint64_t>(cdiv(n, 4), 1u << 20)), dim3(256), 0, ctx->stream,
                         (const uint32_t *)cnt.p, n, slots, (const uint64_t *)offsets.p, (const uint2 *)cand.p, (uint2 *)packed.p);
    AS_HIP(ctx, hipGetLastError());
    AS_TRY(cnt_src.alloc(ctx, (size_t)std::max<uint64_t>((uint64_t)W * rows_here, 1)));
    AS_TRY(inject(2));
    AS_TRY(ag.before_collective());  // (one agreement for the counts and the sizes; the slots have their own below, behind the allocation they need)
    {
      const int rc = comm->ops.all_to_all_v(comm->ops.user, cnt.p, sb.data(), so.data(), cnt_src.p, rb.data(), ro.data(), ctx->stream);
      if (rc != AS_OK) return fail(ctx, rc, "sharded build: all-to-all of the list counts failed");
    }
    // the used slots (2 words each) of every destination's positions: sizes first, then the slots
    std::vector<uint64_t> send_bytes((size_t)W), send_off((size_t)W), recv_bytes((size_t)W), recv_off((size_t)W), eight((size_t)W, 8), at8((size_t)W);
    for (int g = 0; g < W; ++g) {
      send_bytes[(size_t)g] = 2 * (bounds[(size_t)g + 1] - bounds[(size_t)g]) * sizeof(uint32_t);
      send_off[(size_t)g] = 2 * bounds[(size_t)g] * sizeof(uint32_t);
      at8[(size_t)g] = (uint64_t)g * 8;
    }
    AS_TRY(sizes_d.alloc(ctx, (size_t)2 * W));
    AS_HIP(ctx, copy_h2d(ctx, sizes_d.p, send_bytes.data(), sizeof(uint64_t) * W));
    {
      const int rc = comm->ops.all_to_all_v(comm->ops.user, sizes_d.p, eight.data(), at8.data(), sizes_d.p + W, eight.data(), at8.data(), ctx->stream);
      if (rc != AS_OK) return fail(ctx, rc, "sharded build: all-to-all of the list sizes failed");
    }
    AS_HIP(ctx, copy_d2h(ctx, recv_bytes.data(), sizes_d.p + W, sizeof(uint64_t) * W));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t total = 0;
    for (int g = 0; g < W; ++g) {
      recv_off[(size_t)g] = total;
      total += recv_bytes[(size_t)g];
    }
    // (sized by what the peers send: an allocation that can fail between two collectives gets its own agreement, so that a
    // rank that runs out of memory here meets its peers in a one-word all-reduce and not in their all-to-all)
    AS_TRY(recv.alloc(ctx, (size_t)std::max<uint64_t>(total / sizeof(uint32_t), 2)));
    AS_TRY(ag.before_collective());
    {
      const int rc = comm->ops.all_to_all_v(comm->ops.user, packed.p, send_bytes.data(), send_off.data(), recv.p, recv_bytes.data(), recv_off.data(),
                                            ctx->stream);
      if (rc != AS_OK) return fail(ctx, rc, "sharded build: all-to-all of the candidate lists failed");
    }
    cand.release();  // (the dense lists are done with: 4 KB per position of the whole image; stream-ordered reuse)
    packed.release();
    AS_TRY(rows_all(nullptr));  // the exact stages read the f64 rows of every rank
    DevBuf<uint32_t> idx32;  // this rank's rows of the table, 32-bit ids
    if (rows_here) {
      // own pairs -> list 0, all foreign sources -> list 1, positions -> rows
      AS_TRY(used.alloc(ctx, (size_t)W * rows_here));
      AS_TRY(off_src.alloc(ctx, (size_t)W * rows_here + 1));
      hipLaunchKernelGGL(used_slots_kernel, dim3(cdiv((uint64_t)W * rows_here, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)cnt_src.p,
                         (uint64_t)W * rows_here, slots, used.p);
      AS_TRY(exclusive_scan_u32_u64(ctx, used.p, (uint64_t)W * rows_here, off_src.p));
      AS_TRY(cnt2.alloc(ctx, (size_t)2 * rows_here));
      AS_TRY(cand2.alloc(ctx, (size_t)2 * rows_here * slots * 2));
      hipLaunchKernelGGL(merge_lists_kernel, dim3((unsigned)std::min<uint64_t>(cdiv(rows_here, 4), 1u << 16)), dim3(256), 0, ctx->stream,
                         (const uint32_t *)cnt_src.p, (const uint2 *)recv.p, (const uint64_t *)off_src.p, (uint32_t)W, (uint32_t)R, here32, slots,
                         oj.perm, cnt2.p, (uint2 *)cand2.p);
      AS_HIP(ctx, hipGetLastError());
      AS_TRY(idx32.alloc(ctx, (size_t)rows_here * K));
      JoinSources js{nullptr, oj.theta_own.p, 2u, slots, cnt2.p, cand2.p, nullptr, oj.P16own.p, oj.Phown.p};
      if (oj.have_theta2) js.theta2 = oj.reorder ? oj.theta2_own.p : oj.theta2_own.p + lo;  // (by global row with a_ids, else by row of the range)
      if (oj.reorder) {
        js.a_ids = oj.perm_own.p + lo;
        js.theta_local = oj.theta_pos.p + lo;
        js.list_base = lo32;
      }
      js.oj = &oj;
      AS_TRY(knn_mfma(ctx, X.p, norms.p, n, dim, lo, rows_here, K32, params->rectify, idx32.p, dst.p, &js));
      if (!oj.reorder)  // (with an image order the 32-bit table itself is exchanged below)
        hipLaunchKernelGGL(widen_u32_kernel, dim3(cdiv(rows_here * K, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)idx32.p, rows_here * K, idx.p);
      AS_HIP(ctx, hipGetLastError());
    }
    // ---- tables of all positions on every rank, then back to row order ----
    const bool split_assembly = ctx->opt_sharded_split_assembly && K <= 32 && ctx->opt_assembly_lanes != 0 && !params->self_in_results;
    ctx->stats["sharded_split_assembly"] = oj.reorder && split_assembly ? 1 : 0;
    if (oj.reorder && split_assembly) {
      // The first half of the assembly on the owner of each row: a rank counts the degrees of its rows, the degrees of all nodes are
      // exchanged (4 bytes a node), it ranks and prunes the edges of its rows, and the ADJACENCY crosses the links in place of the
      // table (the same 12 bytes an entry + 4 a row) -- the weights and the ranking of 1/W of the rows instead of all of them.
      AS_TRY(deg_pos.alloc(ctx, n));
      AS_TRY(deg_node.alloc(ctx, n));
      AS_TRY(deg_sum.alloc(ctx, 1));
      AS_TRY(pj.alloc(ctx, (size_t)n * K));
      AS_TRY(pw.alloc(ctx, (size_t)n * K));
      AS_TRY(plen.alloc(ctx, n));
      AS_TRY(adj_j.alloc(ctx, (size_t)n * K));
      AS_TRY(adj_w.alloc(ctx, (size_t)n * K));
      AS_TRY(adj_len.alloc(ctx, n));
      {
        ProfScope ps(ctx, "graph_assembly");
        if (rows_here) AS_TRY(graph_degrees_of_rows(ctx, idx32.p, dst.p, oj.perm_own.p + lo, rows_here, K32, params, deg_pos.p + lo));
      }
      AS_TRY(ag.before_collective());
      AS_TRY(gather_rows(ctx, comm, deg_pos.p + lo, deg_pos.p, n, sizeof(uint32_t)));
      {
        ProfScope ps(ctx, "graph_assembly");
        AS_TRY(graph_degrees_to_nodes(ctx, deg_pos.p, oj.perm_own.p, n, deg_node.p, deg_sum.p));
        if (rows_here)
          AS_TRY(graph_adjacency_of_rows(ctx, idx32.p, dst.p, oj.perm_own.p + lo, rows_here, n, K32, params, deg_node.p, deg_sum.p, pj.p + lo * K,
                                         pw.p + lo * K, plen.p + lo));
      }
      AS_TRY(ag.before_collective());
      AS_TRY(gather_rows(ctx, comm, pj.p + lo * K, pj.p, n, K * sizeof(uint32_t)));
      AS_TRY(gather_rows(ctx, comm, pw.p + lo * K, pw.p, n, K * sizeof(double)));
      AS_TRY(gather_rows(ctx, comm, plen.p + lo, plen.p, n, sizeof(uint32_t)));
      {
        ProfScope ps(ctx, "graph_assembly");
        AS_TRY(graph_adjacency_to_nodes(ctx, pj.p, pw.p, plen.p, oj.perm_own.p, n, K32, adj_j.p, adj_w.p, adj_len.p));
      }
    } else if (oj.reorder) {
      // (the indices cross the links as 32-bit values: 12 instead of 16 bytes per table entry, 0.8 GB instead of 1.1 at 4 M x 17)
      DevBuf<uint32_t> idx_pos;
      DevBuf<double> dst_pos;
      AS_TRY(idx_pos.alloc(ctx, (size_t)n * K));
      AS_TRY(dst_pos.alloc(ctx, (size_t)n * K));
      if (!idx32.p) AS_TRY(idx32.alloc(ctx, 1));  // (a rank without rows sends nothing)
      AS_TRY(ag.before_collective());
      AS_TRY(gather_rows(ctx, comm, idx32.p, idx_pos.p, n, K * sizeof(uint32_t)));
      AS_TRY(gather_rows(ctx, comm, dst.p, dst_pos.p, n, K * sizeof(double)));
      hipLaunchKernelGGL(unpermute_table32_kernel, dim3(cdiv(n * K, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)idx_pos.p,
                         (const double *)dst_pos.p, (const uint32_t *)oj.perm_own.p, n, K32, idx_all.p, dst_all.p);
      AS_HIP(ctx, hipGetLastError());
      AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    } else {
      AS_TRY(ag.before_collective());
      AS_TRY(gather_rows(ctx, comm, idx.p, idx_all.p, n, K * sizeof(uint64_t)));
      AS_TRY(gather_rows(ctx, comm, dst.p, dst_all.p, n, K * sizeof(double)));
    }
    if (params->self_in_results) {
      // CosinePair returning the query row itself (as_graph_params): applied to the whole table, the same on every rank
      DevBuf<uint32_t> t32;
      AS_TRY(t32.alloc(ctx, (size_t)n * K));
      AS_TRY(table_self_in_results(ctx, X.p, norms.p, n, dim, K32, params->rectify, idx_all.p, dst_all.p, t32.p));
    }
  } else {
    if (rows_here) AS_TRY(as_knn_cosine(ctx, X.p, n, dim, lo, rows_here, params, idx.p, dst.p));
    // ---- neighbour tables of all rows on every rank (the union symmetrisation needs the in-edges of a rank's rows) ----
    AS_TRY(ag.before_collective());
    AS_TRY(gather_rows(ctx, comm, idx.p, idx_all.p, n, K * sizeof(uint64_t)));
    AS_TRY(gather_rows(ctx, comm, dst.p, dst_all.p, n, K * sizeof(double)));
  }

  // ---- this rank's row block of the Laplacian; nnz of the whole matrix ----
  DevBuf<uint64_t> nnz_d;
  AS_TRY(nnz_d.alloc(ctx, 1));
  uint64_t nnz = 0;
  if (rows_here && adj_j.p) {
    const GraphAdjacency given{adj_j.p, adj_w.p, adj_len.p};
    AS_TRY(laplacian_from_adjacency(ctx, &given, n, (uint32_t)K, params, out_block, lo, rows_here));
    nnz = as_csr_nnz(*out_block);
  } else if (rows_here) {
    AS_TRY(as_laplacian_from_knn_rows(ctx, idx_all.p, dst_all.p, n, params, lo, rows_here, out_block));
    (*out_block)->known_symmetric = 1;  // the table is the library's own: one distance per pair
    nnz = as_csr_nnz(*out_block);
  }
  AS_TRY(inject(3));
  AS_TRY(ag.before_collective());
  // (a failure from here on must not leave the block behind: the caller of this body frees it)
  if (copy_h2d(ctx, nnz_d.p, &nnz, sizeof(nnz)) != hipSuccess) return fail(ctx, AS_ERR_HIP, "sharded build: copy failed");
  {
    const int rc = comm->ops.all_reduce_sum_u64(comm->ops.user, nnz_d.p, 1, ctx->stream);
    if (rc != AS_OK) return fail(ctx, rc, "sharded build: all-reduce failed");
  }
  if (copy_d2h(ctx, &nnz, nnz_d.p, sizeof(nnz)) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
    return fail(ctx, AS_ERR_HIP, "sharded build: copy failed");
  if (nnz_total) *nnz_total = nnz;
  return AS_OK;
}

// The collectives of the three calls below sit behind the same status agreement as the build's: every fallible local step
// (workspace, staging of host inputs, kernels, the local search) runs FIRST, then one word says whether anyone failed; a rank
// that failed returns its own status, its peers AS_ERR_PEER, and nobody is left waiting in an all-gather.
int as_lambda_stats_sharded(as_ctx *ctx, as_comm *comm, const double *lambdas_shard, uint64_t n_here, double *min_out, double *max_out,
                            double *mean_out) {
  if (!ctx) return AS_ERR_ARG;
  if (!comm || (n_here && !lambdas_shard)) return fail(ctx, AS_ERR_ARG, "null argument");
  AS_HIP(ctx, hipSetDevice(ctx->device));
  const int W = comm->world;
  Agreement ag{ctx, comm, ctx->opt_sharded_agreement != 0};
  const unsigned blocks = 256;
  DevBuf<double> part, mine, all;
  double st[4] = {__builtin_inf(), -__builtin_inf(), 0.0, (double)n_here};
  auto local = [&]() -> int {
    if (ctx->opt_sharded_inject_failure == 4) return fail(ctx, AS_ERR_HIP, "sharded lambda statistics: injected failure");
    DevIn<double> L;
    AS_TRY(L.bind(ctx, lambdas_shard, n_here));
    AS_TRY(part.alloc(ctx, (size_t)blocks * 3));
    AS_TRY(mine.alloc(ctx, 4));
    AS_TRY(all.alloc(ctx, (size_t)W * 4));
    hipLaunchKernelGGL(lambda_stats_kernel, dim3(blocks), dim3(256), 0, ctx->stream, L.p, n_here, part.p);
    AS_HIP(ctx, hipGetLastError());
    std::vector<double> hp((size_t)blocks * 3);
    AS_HIP(ctx, copy_d2h(ctx, hp.data(), part.p, hp.size() * sizeof(double)));
    AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (unsigned b = 0; b < blocks; ++b) {
      st[0] = hp[b * 3] < st[0] ? hp[b * 3] : st[0];
      st[1] = hp[b * 3 + 1] > st[1] ? hp[b * 3 + 1] : st[1];
      st[2] += hp[b * 3 + 2];
    }
    AS_HIP(ctx, copy_h2d(ctx, mine.p, st, sizeof(st)));
    return AS_OK;
  };
  {
    const int rc = local();
    if (rc != AS_OK) {
      const std::string err = ctx->err;
      ag.announce_failure();
      return fail(ctx, rc, err);
    }
  }
  AS_TRY(ag.before_collective());
  std::vector<uint64_t> rb((size_t)W, sizeof(st)), ro((size_t)W);
  for (int r = 0; r < W; ++r) ro[(size_t)r] = (uint64_t)r * sizeof(st);
  {
    const int rc = comm->ops.all_gather_v(comm->ops.user, mine.p, sizeof(st), all.p, rb.data(), ro.data(), ctx->stream);
    if (rc != AS_OK) return fail(ctx, rc, "sharded lambda statistics: all-gather failed");
  }
  std::vector<double> ha((size_t)W * 4);
  AS_HIP(ctx, copy_d2h(ctx, ha.data(), all.p, ha.size() * sizeof(double)));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double mn = __builtin_inf(), mx = -__builtin_inf(), sm = 0.0, cnt = 0.0;
  for (int r = 0; r < W; ++r) {  // rank order: the same sum on every rank
    mn = ha[(size_t)r * 4] < mn ? ha[(size_t)r * 4] : mn;
    mx = ha[(size_t)r * 4 + 1] > mx ? ha[(size_t)r * 4 + 1] : mx;
    sm += ha[(size_t)r * 4 + 2];
    cnt += ha[(size_t)r * 4 + 3];
  }
  if (min_out) *min_out = mn;
  if (max_out) *max_out = mx;
  if (mean_out) *mean_out = cnt > 0.0 ? sm / cnt : 0.0;
  return AS_OK;
}

int as_search_lambda_aware_sharded(as_ctx *ctx, as_comm *comm, const as_index *index_shard, const double *queries,
                                   const double *q_lambdas, uint64_t nq, uint64_t k, double alpha, uint64_t id_offset,
                                   uint64_t *idx_out, double *score_out) {
  if (!ctx) return AS_ERR_ARG;
  if (!comm || (nq && (!queries || !q_lambdas))) return fail(ctx, AS_ERR_ARG, "null argument");
  if (nq == 0 || k == 0) return AS_OK;
  if (!idx_out || !score_out) return fail(ctx, AS_ERR_ARG, "null output");
  if (k > 4096) return fail(ctx, AS_ERR_UNSUPPORTED, "k must be <= 4096");
  AS_HIP(ctx, hipSetDevice(ctx->device));
  const int W = comm->world;
  if (W > 64) return fail(ctx, AS_ERR_UNSUPPORTED, "sharded search: more than 64 ranks");
  const uint64_t per = nq * k;
  Agreement ag{ctx, comm, ctx->opt_sharded_agreement != 0};
  DevBuf<uint64_t> li, gi;
  DevBuf<double> ls, gs;
  DevOut<uint64_t> I;
  DevOut<double> S;
  // The local stage: workspace, output staging and this shard's search.  A local failure -- an allocation, or the reference's
  // panics (a zero query lambda: every rank sees it; a NaN score from a non-finite item: this shard alone) -- is announced in
  // the agreement: this rank returns its status, the others AS_ERR_PEER, as the reference fails the whole query.
  auto local = [&]() -> int {
    if (ctx->opt_sharded_inject_failure == 5) return fail(ctx, AS_ERR_HIP, "sharded search: injected failure");
    AS_TRY(li.alloc(ctx, (size_t)per));
    AS_TRY(ls.alloc(ctx, (size_t)per));
    AS_TRY(gi.alloc(ctx, (size_t)W * per));
    AS_TRY(gs.alloc(ctx, (size_t)W * per));
    AS_TRY(I.bind(ctx, idx_out, per));
    AS_TRY(S.bind(ctx, score_out, per));
    if (index_shard) {
      const uint64_t ks = std::min<uint64_t>(k, as_index_nitems(index_shard));  // (a shard smaller than k returns what it has)
      if (ks == k) {
        AS_TRY(as_search_lambda_aware(ctx, index_shard, queries, q_lambdas, nq, k, alpha, id_offset, li.p, ls.p));
      } else {
        DevBuf<uint64_t> ti;
        DevBuf<double> ts;
        AS_TRY(ti.alloc(ctx, (size_t)std::max<uint64_t>(nq * ks, 1)));
        AS_TRY(ts.alloc(ctx, (size_t)std::max<uint64_t>(nq * ks, 1)));
        if (ks) AS_TRY(as_search_lambda_aware(ctx, index_shard, queries, q_lambdas, nq, ks, alpha, id_offset, ti.p, ts.p));
        hipLaunchKernelGGL(widen_topk_kernel, dim3(cdiv(per, 256)), dim3(256), 0, ctx->stream, (const uint64_t *)ti.p, (const double *)ts.p, nq,
                           (uint32_t)ks, (uint32_t)k, li.p, ls.p);
        AS_HIP(ctx, hipGetLastError());
        AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
      }
    } else {
      hipLaunchKernelGGL(fill_empty_topk_kernel, dim3(cdiv(per, 256)), dim3(256), 0, ctx->stream, li.p, ls.p, per);
      AS_HIP(ctx, hipGetLastError());
    }
    return AS_OK;
  };
  int rc_local = local();
  std::string err_local = rc_local != AS_OK ? ctx->err : std::string();
  if (rc_local != AS_OK && ag.enabled) {
    ag.announce_failure();
    return fail(ctx, rc_local, err_local);
  }
  AS_TRY(ag.before_collective());
  if (rc_local != AS_OK) {
    // no agreement ("sharded_agreement" 0): the shard's lists are exchanged empty and the status is returned afterwards, so that
    // the peers at least are not left waiting (they merge what the other shards found)
    if (!li.p || !ls.p || !gi.p || !gs.p) return fail(ctx, rc_local, err_local);  // (without workspace there is nothing to exchange)
    hipLaunchKernelGGL(fill_empty_topk_kernel, dim3(cdiv(per, 256)), dim3(256), 0, ctx->stream, li.p, ls.p, per);
  }
  std::vector<uint64_t> rb((size_t)W, per * 8), ro((size_t)W);
  for (int r = 0; r < W; ++r) ro[(size_t)r] = (uint64_t)r * per * 8;
  {
    int rc = comm->ops.all_gather_v(comm->ops.user, li.p, per * 8, gi.p, rb.data(), ro.data(), ctx->stream);
    if (rc == AS_OK) rc = comm->ops.all_gather_v(comm->ops.user, ls.p, per * 8, gs.p, rb.data(), ro.data(), ctx->stream);
    if (rc != AS_OK) return fail(ctx, rc, "sharded search: all-gather failed");
  }
  if (rc_local != AS_OK) return fail(ctx, rc_local, err_local);
  hipLaunchKernelGGL(merge_topk_kernel, dim3(cdiv(nq, 128)), dim3(128), 0, ctx->stream, (const uint64_t *)gi.p, (const double *)gs.p, (uint32_t)W, nq,
                     (uint32_t)k, I.p, S.p);
  AS_HIP(ctx, hipGetLastError());
  AS_TRY(I.finish(ctx));
  AS_TRY(S.finish(ctx));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));  // li / ls / gi / gs go back to the workspace cache
  return AS_OK;
}

// Item-graph lambda over a row-sharded graph (TauMode::compute_synthetic_lambda_csr,
// src/taumode.rs:552-660, with the sums of :565-596 and :611-639 taken per row block): all-gather of the item shards (a stored
// pair (i, j) reads row j wherever it lives), every rank sums (num, edge, sum t^2, den) over the rows of ITS block -- a pair
// {i, j} is taken once, by the owner of min(i, j) -- all-gather of the 4 sums per signal, added in rank order; a second round of
// one sum per signal only where the literal shares are needed (the same decision on every rank: it is a function of the totals).
int as_compute_taumode_signals_sharded(as_ctx *ctx, as_comm *comm, const double *items_shard, uint64_t n_total, uint64_t n_features,
                                       const as_csr *L_block, int tau_mode, double tau_arg, double *lambdas_out) {
  if (!ctx) return AS_ERR_ARG;
  if (!comm || !lambdas_out) return fail(ctx, AS_ERR_ARG, "null argument");
  if (n_features == 0 || n_features >= 0xFFFFFFFFull || n_total == 0 || n_total >= 0x7FFFFFFFull) return fail(ctx, AS_ERR_ARG, "bad shape");
  if (tau_mode < AS_TAU_FIXED || tau_mode > AS_TAU_PERCENTILE) return fail(ctx, AS_ERR_ARG, "bad tau_mode");
  AS_HIP(ctx, hipSetDevice(ctx->device));
  const int W = comm->world;
  const uint32_t S = (uint32_t)n_features, s_pad = signal_pad(S);
  uint64_t lo, hi;
  shard_range(n_total, comm->rank, W, &lo, &hi);
  as_csr *L = const_cast<as_csr *>(L_block);
  Agreement ag{ctx, comm, ctx->opt_sharded_agreement != 0};
  auto leave = [&](int rc) -> int {  // a local failure: said once in the next agreement
    const std::string err = ctx->err;
    ag.announce_failure();
    return fail(ctx, rc, err);
  };
  // rank-local argument checks: a rank that fails them says so in the first agreement instead of leaving its peers in it
  if (ctx->opt_sharded_inject_failure == 8) return leave(fail(ctx, AS_ERR_HIP, "sharded item-graph lambda: injected failure"));
  if ((hi > lo) && !items_shard) return leave(fail(ctx, AS_ERR_ARG, "null shard"));
  if (L && (L->row_begin != lo || L->n != hi - lo || (L->n_cols ? L->n_cols : L->n) > n_total))
    return leave(fail(ctx, AS_ERR_ARG, "the row block does not belong to this rank's range of rows"));
#define AS_STAGE(expr)                  \
  do {                                  \
    const int _rc = (expr);             \
    if (_rc != AS_OK) return leave(_rc); \
  } while (0)
  DevIn<double> mine;
  DevBuf<double> all_rows, sums_mine, sums_all, sums, tau, g2_mine, g2_all, g2;
  DevBuf<int> need;
  DevOut<double> O;
  auto stage1 = [&]() -> int {
    if (ctx->opt_sharded_inject_failure == 6) return fail(ctx, AS_ERR_HIP, "sharded item-graph lambda: injected failure");
    AS_TRY(mine.bind(ctx, items_shard, (hi - lo) * n_features));
    AS_TRY(all_rows.alloc(ctx, (size_t)n_total * n_features));
    AS_TRY(sums_mine.alloc(ctx, (size_t)4 * s_pad));
    AS_TRY(sums_all.alloc(ctx, (size_t)W * 4 * s_pad));
    AS_TRY(sums.alloc(ctx, (size_t)4 * s_pad));
    AS_TRY(tau.alloc(ctx, S));
    AS_TRY(g2_mine.alloc(ctx, s_pad));
    AS_TRY(g2_all.alloc(ctx, (size_t)W * s_pad));
    AS_TRY(g2.alloc(ctx, s_pad));
    AS_TRY(need.alloc(ctx, 1));
    AS_TRY(O.bind(ctx, lambdas_out, S));
    return AS_OK;
  };
  AS_STAGE(stage1());
  AS_TRY(ag.before_collective());
  AS_TRY(gather_rows(ctx, comm, mine.p, all_rows.p, n_total, n_features * sizeof(double)));
  const uint64_t graph_n = L ? (L->n_cols ? L->n_cols : L->n) : n_total;
  auto stage2 = [&]() -> int {
    AS_TRY(signal_taus(ctx, all_rows.p, n_total, S, tau_mode, tau_arg, tau.p));  // every rank has all rows: tau is not exchanged
    if (L) {
      // the rows past the graph (a signal longer than the graph has nodes) count for the denominator only: rank 0 takes them
      const bool first = comm->rank == 0;
      AS_TRY(signal_sums(ctx, all_rows.p, n_total, S, L, first ? graph_n : 0, first ? n_total : 0, sums_mine.p));
    } else {
      AS_HIP(ctx, hipMemsetAsync(sums_mine.p, 0, sizeof(double) * 4 * s_pad, ctx->stream));
    }
    return AS_OK;
  };
  AS_STAGE(stage2());
  AS_TRY(ag.before_collective());
  {
    std::vector<uint64_t> rb((size_t)W, (uint64_t)4 * s_pad * 8), ro((size_t)W);
    for (int r = 0; r < W; ++r) ro[(size_t)r] = (uint64_t)r * 4 * s_pad * 8;
    const int rc = comm->ops.all_gather_v(comm->ops.user, sums_mine.p, (uint64_t)4 * s_pad * 8, sums_all.p, rb.data(), ro.data(), ctx->stream);
    if (rc != AS_OK) return fail(ctx, rc, "sharded item-graph lambda: all-gather failed");
  }
  // Stage 3 ends in an agreement of its own, whether a second round follows or not: a rank that fails here while its peers
  // need no second round would otherwise announce into a collective nobody else enters (and poison the NEXT sharded call).
  int need_h = 0;
  auto stage3 = [&]() -> int {
    if (ctx->opt_sharded_inject_failure == 7) return fail(ctx, AS_ERR_HIP, "sharded item-graph lambda: injected failure");
    AS_TRY(signal_rank_sum(ctx, sums_all.p, (uint32_t)W, 4 * s_pad, sums.p));
    AS_TRY(signal_need_pass2(ctx, sums.p, S, need.p));
    AS_HIP(ctx, copy_d2h(ctx, &need_h, need.p, sizeof(int)));  // the same value on every rank: the totals are
    return AS_OK;
  };
  AS_STAGE(stage3());
  AS_TRY(ag.before_collective());
  if (need_h) {
    auto stage4 = [&]() -> int {
      if (L) {
        AS_TRY(signal_shares(ctx, all_rows.p, S, L, sums.p, need.p, g2_mine.p));
        // A block streamed as its upper triangle saw every pair once: its shares count twice.  The factor belongs to the
        // block's cached record stream (L->sig_sym), not to the option's value at this call, and blocks may differ -- so every
        // rank applies its own before the exchange (a doubling is exact: the rank-ordered sum is the same bit for bit).
        if (L->sig_sym) {
          hipLaunchKernelGGL(scale_f64_kernel, dim3(cdiv(s_pad, 256)), dim3(256), 0, ctx->stream, g2_mine.p, (uint64_t)s_pad, 2.0);
          AS_HIP(ctx, hipGetLastError());
        }
      } else {
        AS_HIP(ctx, hipMemsetAsync(g2_mine.p, 0, sizeof(double) * s_pad, ctx->stream));
      }
      return AS_OK;
    };
    AS_STAGE(stage4());
    AS_TRY(ag.before_collective());
    std::vector<uint64_t> rb((size_t)W, (uint64_t)s_pad * 8), ro((size_t)W);
    for (int r = 0; r < W; ++r) ro[(size_t)r] = (uint64_t)r * s_pad * 8;
    const int rc = comm->ops.all_gather_v(comm->ops.user, g2_mine.p, (uint64_t)s_pad * 8, g2_all.p, rb.data(), ro.data(), ctx->stream);
    if (rc != AS_OK) return fail(ctx, rc, "sharded item-graph lambda: all-gather failed");
    AS_TRY(signal_rank_sum(ctx, g2_all.p, (uint32_t)W, s_pad, g2.p));
  }
#undef AS_STAGE
  AS_TRY(signal_lambda_from_sums(ctx, sums.p, g2.p, need.p, /*upper_only=*/false, S, tau.p, O.p));
  AS_TRY(O.finish(ctx));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return AS_OK;
}

}  // extern "C"
