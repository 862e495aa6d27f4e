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
  for (int p = 0; p < c->world; ++p) {
    if (send_bytes[p])
      if (r.Send(static_cast<const char *>(send) + send_offsets[p], send_bytes[p], ncclInt8, p, c->nccl, (hipStream_t)stream) != ncclSuccess)
        return AS_ERR_HIP;
    if (recv_bytes[p])
      if (r.Recv(static_cast<char *>(recv) + recv_offsets[p], recv_bytes[p], ncclInt8, p, c->nccl, (hipStream_t)stream) != ncclSuccess)
        return AS_ERR_HIP;
  }
  return r.GroupEnd() == ncclSuccess ? AS_OK : AS_ERR_HIP;
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
int gather_rows(as_ctx *ctx, as_comm *c, const void *mine, void *all, uint64_t n, uint64_t row_bytes) {
  std::vector<uint64_t> rb((size_t)c->world), ro((size_t)c->world);
  uint64_t my = 0;
  for (int r = 0; r < c->world; ++r) {
    uint64_t lo, hi;
    shard_range(n, r, c->world, &lo, &hi);
    rb[(size_t)r] = (hi - lo) * row_bytes;
    ro[(size_t)r] = lo * row_bytes;
    if (r == c->rank) my = rb[(size_t)r];
  }
  const int rc = c->ops.all_gather_v(c->ops.user, mine, my, all, rb.data(), ro.data(), ctx->stream);
  return rc == AS_OK ? AS_OK : fail(ctx, rc, "sharded build: all-gather failed");
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

int as_build_laplacian_sharded(as_ctx *ctx, as_comm *comm, const double *rows_shard, uint64_t n_total, uint64_t dim,
                               const as_graph_params *params, as_csr **out_block, uint64_t *nnz_total) {
  if (!ctx) return AS_ERR_ARG;
  if (!comm || !params || !out_block) return fail(ctx, AS_ERR_ARG, "null argument");
  *out_block = nullptr;
  if (n_total < 2 || dim < 2)
    return fail(ctx, AS_ERR_SHAPE, "items should be at least of shape (2,2): (" + std::to_string(n_total) + "," + std::to_string(dim) + ")");
  if (n_total >= 0xFFFFFFFFull || dim >= 0xFFFFFFFFull) return fail(ctx, AS_ERR_UNSUPPORTED, "n_nodes/dim must be < 2^32-1");
  AS_HIP(ctx, hipSetDevice(ctx->device));
  const int W = comm->world, R = comm->rank;
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

  // ---- all rows on every rank (B side of the neighbour search, candidates' rows for the exact re-rank) ----
  DevBuf<double> X;
  AS_TRY(X.alloc(ctx, (size_t)n * dim));
  AS_TRY(gather_rows(ctx, comm, rows_shard, X.p, n, dim * sizeof(double)));

  DevBuf<uint64_t> idx, idx_all;
  DevBuf<double> dst, dst_all;
  AS_TRY(idx.alloc(ctx, (size_t)std::max<uint64_t>(rows_here, 1) * K));
  AS_TRY(dst.alloc(ctx, (size_t)std::max<uint64_t>(rows_here, 1) * K));
  if (W > 1 && as_knn_join_applies(ctx, n, dim, params)) {
    // symmetric threshold join: thresholds all-gathered, lists to the owners of their rows in compact form
    DevBuf<float> theta, theta_all;
    AS_TRY(theta.alloc(ctx, (size_t)std::max<uint64_t>(rows_here, 1)));
    AS_TRY(theta_all.alloc(ctx, (size_t)n));
    if (rows_here) AS_TRY(as_knn_join_thresholds(ctx, X.p, n, dim, lo, rows_here, params, theta.p));
    AS_TRY(gather_rows(ctx, comm, theta.p, theta_all.p, n, sizeof(float)));
    const uint32_t cap = as_knn_join_capacity(ctx, params);
    DevBuf<uint32_t> cnt, cand, packed, cnt_src, recv, cand_src;
    DevBuf<uint64_t> offsets, sizes_d;
    AS_TRY(cnt.alloc(ctx, (size_t)n));
    AS_TRY(offsets.alloc(ctx, (size_t)n + 1));
    if (rows_here) {
      AS_TRY(cand.alloc(ctx, (size_t)n * cap));
      AS_TRY(packed.alloc(ctx, (size_t)n * cap));
      AS_TRY(as_knn_join_pairs(ctx, X.p, n, dim, lo, rows_here, params, theta_all.p, cap, cnt.p, cand.p));
      AS_TRY(as_knn_join_pack(ctx, cnt.p, cand.p, n, cap, offsets.p, packed.p));
    } else {  // nothing contracted here: empty lists for every row
      AS_HIP(ctx, hipMemsetAsync(cnt.p, 0, sizeof(uint32_t) * n, ctx->stream));
      AS_HIP(ctx, hipMemsetAsync(offsets.p, 0, sizeof(uint64_t) * (n + 1), ctx->stream));
      AS_TRY(packed.alloc(ctx, 1));
    }
    // counts of every rank's rows, source by source
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
    AS_TRY(cnt_src.alloc(ctx, (size_t)std::max<uint64_t>((uint64_t)W * rows_here, 1)));
    {
      const int rc = comm->ops.all_to_all_v(comm->ops.user, cnt.p, sb.data(), so.data(), cnt_src.p, rb.data(), ro.data(), ctx->stream);
      if (rc != AS_OK) return fail(ctx, rc, "sharded build: all-to-all of the list counts failed");
    }
    // the used slots (2 words each) of every destination's rows: sizes first, then the slots
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
    AS_TRY(recv.alloc(ctx, (size_t)std::max<uint64_t>(total / sizeof(uint32_t), 1)));
    {
      const int rc = comm->ops.all_to_all_v(comm->ops.user, packed.p, send_bytes.data(), send_off.data(), recv.p, recv_bytes.data(), recv_off.data(),
                                            ctx->stream);
      if (rc != AS_OK) return fail(ctx, rc, "sharded build: all-to-all of the candidate lists failed");
    }
    if (rows_here) {
      AS_TRY(cand_src.alloc(ctx, (size_t)W * rows_here * cap));
      AS_TRY(as_knn_join_unpack(ctx, cnt_src.p, recv.p, (uint64_t)W * rows_here, cap, cand_src.p));
      AS_TRY(as_knn_join_finish(ctx, X.p, n, dim, lo, rows_here, params, theta_all.p, (uint32_t)W, cap, cnt_src.p, cand_src.p, idx.p, dst.p));
    }
    ctx->stats["sharded_join"] = 1;
  } else {
    if (rows_here) AS_TRY(as_knn_cosine(ctx, X.p, n, dim, lo, rows_here, params, idx.p, dst.p));
    ctx->stats["sharded_join"] = 0;
  }

  // ---- neighbour tables of all rows on every rank (the union symmetrisation needs the in-edges of a rank's rows) ----
  AS_TRY(idx_all.alloc(ctx, (size_t)n * K));
  AS_TRY(dst_all.alloc(ctx, (size_t)n * K));
  AS_TRY(gather_rows(ctx, comm, idx.p, idx_all.p, n, K * sizeof(uint64_t)));
  AS_TRY(gather_rows(ctx, comm, dst.p, dst_all.p, n, K * sizeof(double)));

  // ---- this rank's row block of the Laplacian; nnz of the whole matrix ----
  DevBuf<uint64_t> nnz_d;
  AS_TRY(nnz_d.alloc(ctx, 1));
  uint64_t nnz = 0;
  if (rows_here) {
    AS_TRY(as_laplacian_from_knn_rows(ctx, idx_all.p, dst_all.p, n, params, lo, rows_here, out_block));
    nnz = as_csr_nnz(*out_block);
  }
  AS_HIP(ctx, copy_h2d(ctx, nnz_d.p, &nnz, sizeof(nnz)));
  {
    const int rc = comm->ops.all_reduce_sum_u64(comm->ops.user, nnz_d.p, 1, ctx->stream);
    if (rc != AS_OK) return fail(ctx, rc, "sharded build: all-reduce failed");
  }
  AS_HIP(ctx, copy_d2h(ctx, &nnz, nnz_d.p, sizeof(nnz)));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (nnz_total) *nnz_total = nnz;
  return AS_OK;
}

}  // extern "C"
