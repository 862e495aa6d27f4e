// api.hip -- the extern "C" surface declared in include/arrowspace_hip.h.
#include <algorithm>
#include <cmath>

#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <thread>

#include <unordered_map>
#include <map>
#include <set>
#include <deque>
#include <vector>
#include "common.h"

using namespace asx;

// ---- host copy threads (staging of host inputs and outputs, common.h: copy_h2d / copy_d2h) --------------------------------
// A handful of threads that split one memcpy between them.  Created at the first large transfer, kept for the process.
namespace {
struct CopyPool {
  std::mutex m;
  std::condition_variable cv_work, cv_done;
  std::vector<std::thread> threads;
  char *dst = nullptr;
  const char *src = nullptr;
  size_t bytes = 0;
  uint64_t generation = 0;
  int pending = 0, n_threads = 0;
  bool stop = false;
  void worker(int id) {
    uint64_t seen = 0;
    for (;;) {
      char *d;
      const char *s_;
      size_t b;
      {
        std::unique_lock<std::mutex> lk(m);
        cv_work.wait(lk, [&] { return stop || generation != seen; });
        if (stop) return;
        seen = generation;
        d = dst;
        s_ = src;
        b = bytes;
      }
      const size_t parts = (size_t)n_threads + 1, per = (b / parts + 4095) & ~(size_t)4095;
      const size_t lo = std::min(b, per * (size_t)(id + 1)), hi = std::min(b, lo + per);  // part 0 is the caller's
      if (hi > lo) std::memcpy(d + lo, s_ + lo, hi - lo);
      {
        std::lock_guard<std::mutex> lk(m);
        if (--pending == 0) cv_done.notify_one();
      }
    }
  }
  void start() {
    const unsigned hw = std::thread::hardware_concurrency();
    n_threads = (int)std::max(1u, std::min(7u, hw > 2 ? hw / 2 - 1 : 1u));
    for (int i = 0; i < n_threads; ++i) threads.emplace_back([this, i] { worker(i); });
  }
  ~CopyPool() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv_work.notify_all();
    for (auto &t : threads) t.join();
  }
};
CopyPool *copy_pool() {
  static CopyPool *pool = [] {
    CopyPool *p = new CopyPool();  // (leaked on purpose at exit: joining threads from a static destructor can deadlock the unload)
    p->start();
    return p;
  }();
  return pool;
}
std::mutex g_copy_call;  // one transfer at a time through the pool (contexts of different host threads share it)
}  // namespace

namespace asx {

// ---- handle memory cache (common.h) ---------------------------------------------------------------------------------------------
namespace {
struct HandleCache {
  std::mutex mu;
  // Streams the library's contexts run on (their own, or the caller's: as_ctx_set_stream), per device.  A block given back is
  // stamped with one event per such stream, recorded at that moment; its next owner waits for THOSE events on the host, i.e. for
  // the work queued before the free on the streams that can have touched the block -- not for the whole device, which stalled
  // every other context, torch's streams and concurrent searches of a shared index (hipDeviceSynchronize, until round 5).
  std::multiset<std::pair<int, hipStream_t>> streams;
  std::unordered_map<void *, std::vector<hipEvent_t>> stamps;  // idle block -> events to wait for
  std::vector<hipEvent_t> event_pool;
  std::unordered_map<void *, std::pair<int, size_t>> live;           // blocks handed out: device, bytes
  std::multimap<std::pair<int, size_t>, void *> idle;                // blocks given back, by (device, bytes)
  std::deque<std::pair<void *, std::pair<int, size_t>>> idle_order;  // ... oldest first, for eviction
  size_t idle_bytes = 0;
  // time this process spent in the allocator on behalf of handles (microseconds, counts): "handle_cache_*" statistics
  std::atomic<uint64_t> us_malloc{0}, us_free{0}, us_wait{0}, n_malloc{0}, n_free{0}, n_hit{0};
  size_t cap = [] {
    const char *e = std::getenv("ARROWSPACE_HANDLE_CACHE_MB");
    const long long mb = e ? std::atoll(e) : 2048;
    return (size_t)(mb > 0 ? mb : 0) << 20;
  }();
  // evicts until at most `limit` idle bytes remain; returns the blocks to hipFree (outside the lock)
  std::vector<void *> shrink(size_t limit) {
    std::vector<void *> out;
    while (idle_bytes > limit && !idle_order.empty()) {
      auto v = idle_order.front();
      idle_order.pop_front();
      auto range = idle.equal_range(v.second);
      for (auto it = range.first; it != range.second; ++it)
        if (it->second == v.first) {
          idle.erase(it);
          idle_bytes -= v.second.second;
          out.push_back(v.first);
          auto st = stamps.find(v.first);
          if (st != stamps.end()) {  // (the hipFree that follows waits for the device by itself)
            event_pool.insert(event_pool.end(), st->second.begin(), st->second.end());
            stamps.erase(st);
          }
          break;
        }
    }
    return out;
  }
};
struct UsTimer {
  std::atomic<uint64_t> &acc;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  explicit UsTimer(std::atomic<uint64_t> &a) : acc(a) {}
  ~UsTimer() { acc += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); }
};
HandleCache &handle_cache() {
  static HandleCache *c = new HandleCache();  // (never destroyed: handles may be freed during process exit)
  return *c;
}
}  // namespace

hipError_t handle_malloc(void **p, size_t bytes) {
  HandleCache &c = handle_cache();
  int dev = 0;
  (void)hipGetDevice(&dev);
  bytes = bytes ? bytes : 1;
  void *hit = nullptr;
  std::vector<hipEvent_t> wait;
  bool stamped = false;
  {
    std::lock_guard<std::mutex> g(c.mu);
    // the smallest idle block of the device that holds the request and is at most a quarter larger (until round 5: the same size to
    // the byte -- graphs of other data never matched, the cache filled with sizes nobody asked for again and every eviction was a
    // hipFree that stalls the device: 30 ms in the middle of a build, `item_graph_ms_each` of the bench legs)
    auto it = c.idle.lower_bound({dev, bytes});
    if (it != c.idle.end() && it->first.first == dev && it->first.second <= bytes + bytes / 4 + 4096) {
      hit = it->second;
      bytes = it->first.second;  // (what the block really holds: it goes back to its own bucket)
      c.idle.erase(it);
      c.idle_bytes -= bytes;
      for (auto o = c.idle_order.begin(); o != c.idle_order.end(); ++o)
        if (o->first == hit) {
          c.idle_order.erase(o);
          break;
        }
      auto st = c.stamps.find(hit);
      if (st != c.stamps.end()) {
        wait = std::move(st->second);
        c.stamps.erase(st);
        stamped = true;
      }
    }
  }
  if (hit) {
    // whatever still reads or writes the block finishes first (what hipFree did for its last owner): the events recorded when it
    // was given back, one per stream the library runs on; a block without a stamp (no event to be had) waits for the device
    hipError_t e = hipSuccess;
    UsTimer tw(c.us_wait);
    ++c.n_hit;
    if (!stamped) e = hipDeviceSynchronize();
    for (hipEvent_t ev : wait) {
      const hipError_t e1 = hipEventSynchronize(ev);
      if (e1 != hipSuccess) e = e1;
    }
    {
      std::lock_guard<std::mutex> g(c.mu);
      c.event_pool.insert(c.event_pool.end(), wait.begin(), wait.end());
    }
    if (e != hipSuccess) return e;
    *p = hit;
  } else {
    UsTimer tm(c.us_malloc);
    ++c.n_malloc;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {  // out of memory: give the idle blocks back and try once more
      (void)hipGetLastError();
      std::vector<void *> drop;
      {
        std::lock_guard<std::mutex> g(c.mu);
        drop = c.shrink(0);
      }
      for (void *q : drop) (void)hipFree(q);
      e = hipMalloc(p, bytes);
      if (e != hipSuccess) return e;
    }
  }
  std::lock_guard<std::mutex> g(c.mu);
  c.live[*p] = {dev, bytes};
  return hipSuccess;
}

hipError_t handle_free(void *p) {
  if (!p) return hipSuccess;
  HandleCache &c = handle_cache();
  std::vector<void *> drop;
  bool keep = false;
  {
    std::lock_guard<std::mutex> g(c.mu);
    auto it = c.live.find(p);
    if (it != c.live.end()) {
      const auto key = it->second;
      c.live.erase(it);
      if (c.cap > 0 && key.second <= c.cap) {
        // stamp: an event on every registered stream of the block's device (recorded under the lock: a handful of microseconds)
        std::vector<hipEvent_t> evs;
        bool ok = true;
        hipStream_t last = (hipStream_t)-1;
        for (auto sit = c.streams.lower_bound({key.first, (hipStream_t)0}); ok && sit != c.streams.end() && sit->first == key.first; ++sit) {
          if (sit->second == last) continue;  // (a stream several contexts share)
          last = sit->second;
          hipEvent_t ev = nullptr;
          if (!c.event_pool.empty()) {
            ev = c.event_pool.back();
            c.event_pool.pop_back();
          } else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            ok = false;
            break;
          }
          evs.push_back(ev);
          if (hipEventRecord(ev, sit->second) != hipSuccess) {
            (void)hipGetLastError();
            ok = false;
          }
        }
        if (ok) {
          c.stamps[p] = std::move(evs);
        } else {  // no stamp: the next owner synchronises the device
          c.event_pool.insert(c.event_pool.end(), evs.begin(), evs.end());
        }
        c.idle.insert({key, p});
        c.idle_order.push_back({p, key});
        c.idle_bytes += key.second;
        keep = true;
        drop = c.shrink(c.cap);
      }
    }
  }
  hipError_t e = hipSuccess;
  if (!drop.empty() || !keep) {
    UsTimer tf(c.us_free);
    c.n_free += drop.size() + (keep ? 0 : 1);
    for (void *q : drop) (void)hipFree(q);
    if (!keep) e = hipFree(p);
  }
  return e;
}

void handle_cache_stream(int device, hipStream_t s, bool add) {
  HandleCache &c = handle_cache();
  std::lock_guard<std::mutex> g(c.mu);
  if (add) {
    c.streams.insert({device, s});
  } else {
    auto it = c.streams.find({device, s});
    if (it != c.streams.end()) c.streams.erase(it);
  }
}

void handle_cache_release() {
  HandleCache &c = handle_cache();
  std::vector<void *> drop;
  {
    std::lock_guard<std::mutex> g(c.mu);
    drop = c.shrink(0);
  }
  for (void *q : drop) (void)hipFree(q);
}

void handle_cache_limit(int64_t mb) {
  HandleCache &c = handle_cache();
  std::vector<void *> drop;
  {
    std::lock_guard<std::mutex> g(c.mu);
    c.cap = (size_t)(mb > 0 ? mb : 0) << 20;
    drop = c.shrink(c.cap);
  }
  for (void *q : drop) (void)hipFree(q);
}
void host_copy_parallel(void *dst, const void *src, size_t bytes) {
  if (bytes < (1u << 20)) {
    std::memcpy(dst, src, bytes);
    return;
  }
  std::lock_guard<std::mutex> call(g_copy_call);
  CopyPool *p = copy_pool();
  const size_t parts = (size_t)p->n_threads + 1, per = (bytes / parts + 4095) & ~(size_t)4095;
  {
    std::lock_guard<std::mutex> lk(p->m);
    p->dst = (char *)dst;
    p->src = (const char *)src;
    p->bytes = bytes;
    p->pending = p->n_threads;
    ++p->generation;
  }
  p->cv_work.notify_all();
  std::memcpy(dst, src, std::min(bytes, per));  // the caller's own part
  std::unique_lock<std::mutex> lk(p->m);
  p->cv_done.wait(lk, [&] { return p->pending == 0; });
}
int ctx_side_stream(as_ctx *ctx) {
  if (ctx->side && ctx->side_word) return AS_OK;
  if (!ctx->side) AS_HIP(ctx, hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
  if (!ctx->side_start) AS_HIP(ctx, hipEventCreateWithFlags(&ctx->side_start, hipEventDisableTiming));
  if (!ctx->side_done) AS_HIP(ctx, hipEventCreateWithFlags(&ctx->side_done, hipEventDisableTiming));
  if (!ctx->side_word) AS_HIP(ctx, hipHostMalloc((void **)&ctx->side_word, 64, hipHostMallocDefault));  // (the side stream's own read-back word)
  return AS_OK;
}

}  // namespace asx

namespace {

__global__ void widen_idx_kernel(const uint32_t *__restrict__ in, uint64_t n, uint64_t offset,
                                 uint64_t *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] == kNoIdx ? ~0ull : (uint64_t)in[i] + offset;
}
__global__ void narrow_idx_kernel(const uint64_t *__restrict__ in, uint64_t n, uint32_t *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] == ~0ull ? kNoIdx : (uint32_t)in[i];
}
__global__ void negate_kernel(const double *__restrict__ in, uint64_t n, double *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = -in[i];
}
__global__ void any_zero_kernel(const double *__restrict__ v, uint64_t n, int *flag) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && v[i] == 0.0) atomicOr(flag, 1);
}

// search_lambda_aware_hybrid merge (src/core.rs:888-923), one thread per query.
// A = cosine top-k (key = -cos), B = lambda-aware top-k (key = -score); both sorted (key asc, idx asc).
// ui / us: scratch of 2 kk + 1 entries per query (global memory: k is not bounded by a register array).
__global__ void hybrid_merge_kernel(const uint32_t *__restrict__ a_idx, const double *__restrict__ a_key,
                                    const uint32_t *__restrict__ b_idx, const double *__restrict__ b_key, uint64_t nq,
                                    uint32_t kk, uint32_t k_out, uint64_t id_offset, uint32_t *__restrict__ scratch_i,
                                    double *__restrict__ scratch_s, uint64_t *__restrict__ out_idx,
                                    double *__restrict__ out_score) {
  const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  uint32_t *ui = scratch_i + q * (2 * (uint64_t)kk + 1);
  double *us = scratch_s + q * (2 * (uint64_t)kk + 1);
  uint32_t m = 0;
  const uint32_t *ai = a_idx + q * kk, *bi = b_idx + q * kk;
  const double *ak = a_key + q * kk, *bk = b_key + q * kk;
  // (a) high semantic matches: cosine > 0.9999 (only the best k of them can reach the output)
  for (uint32_t t = 0; t < kk; ++t)
    if (ai[t] != kNoIdx && -ak[t] > 0.9999) {
      ui[m] = ai[t];
      us[m++] = -ak[t];
    }
  // (b) lambda-aware top-k, score kept if the item is already present
  for (uint32_t t = 0; t < kk; ++t) {
    if (bi[t] == kNoIdx) continue;
    bool present = false;
    for (uint32_t u = 0; u < m; ++u) present |= ui[u] == bi[t];
    if (!present) {
      ui[m] = bi[t];
      us[m++] = -bk[t];
    }
  }
  // (c) the best-cosine item
  if (kk > 0 && ai[0] != kNoIdx) {
    bool present = false;
    for (uint32_t u = 0; u < m; ++u) present |= ui[u] == ai[0];
    if (!present) {
      ui[m] = ai[0];
      us[m++] = -ak[0];
    }
  }
  // sort by (score desc, idx asc), truncate
  for (uint32_t a = 1; a < m; ++a) {
    const uint32_t ti = ui[a];
    const double ts = us[a];
    uint32_t b = a;
    while (b > 0 && (us[b - 1] < ts || (us[b - 1] == ts && ui[b - 1] > ti))) {
      ui[b] = ui[b - 1];
      us[b] = us[b - 1];
      --b;
    }
    ui[b] = ti;
    us[b] = ts;
  }
  for (uint32_t t = 0; t < k_out; ++t) {
    out_idx[q * k_out + t] = t < m ? (uint64_t)ui[t] + id_offset : ~0ull;
    out_score[q * k_out + t] = t < m ? us[t] : -__builtin_inf();
  }
}

__global__ void range_flag_kernel(const double *__restrict__ lambdas, uint64_t n, double lq, double eps,
                                  uint32_t *__restrict__ flags) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = (lq - lambdas[i]) <= eps ? 1u : 0u;
}
__global__ void range_scatter_kernel(const double *__restrict__ lambdas, uint64_t n, double lq,
                                     const uint32_t *__restrict__ flags, const uint64_t *__restrict__ pos, uint64_t cap,
                                     uint64_t id_offset, uint64_t *__restrict__ out_idx, double *__restrict__ out_dist) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flags[i] && pos[i] < cap) {
    out_idx[pos[i]] = i + id_offset;
    out_dist[pos[i]] = lq - lambdas[i];
  }
}

int check_ctx(as_ctx *ctx) {
  if (!ctx) return AS_ERR_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, AS_ERR_HIP, "hipSetDevice failed");
  ctx->err.clear();
  return AS_OK;
}

// K > 64 (topk + 1 beyond the list depth of the MFMA stages) goes straight to the exact scan, which continues in passes
int knn_dispatch(as_ctx *ctx, const double *rows, const double *norms, uint64_t n, uint64_t d, uint64_t row_begin,
                 uint64_t row_count, uint32_t K, const as_graph_params *gp, uint32_t *idx, double *dist) {
  if (ctx->opt_knn_mfma && K <= 64) AS_TRY(knn_mfma(ctx, rows, norms, n, d, row_begin, row_count, K, gp->rectify, idx, dist));
  else AS_TRY(knn_exact(ctx, rows, norms, n, d, nullptr, row_begin, row_count, K, gp->rectify, idx, dist, 0));
  if (gp->self_in_results) AS_TRY(apply_self_in_results(ctx, rows, norms, d, row_begin, row_count, K, gp->rectify, idx, dist));
  return AS_OK;
}

int validate_graph_args(as_ctx *ctx, const double *rows, uint64_t n, uint64_t d, const as_graph_params *gp) {
  if (!rows || !gp) return fail(ctx, AS_ERR_ARG, "null argument");
  if (n < 2 || d < 2) {
    // the reference prints (rows, cols) of the matrix as passed: src/laplacian.rs:128-134
    return fail(ctx, AS_ERR_SHAPE,
                "items should be at least of shape (2,2): (" + std::to_string(n) + "," + std::to_string(d) + ")");
  }
  if (n >= 0xFFFFFFFFull || d >= 0xFFFFFFFFull) return fail(ctx, AS_ERR_UNSUPPORTED, "n_nodes/dim must be < 2^32-1");
  if (gp->topk + 1 > 256) return fail(ctx, AS_ERR_UNSUPPORTED, "topk + 1 must be <= 256");
  return AS_OK;
}

}  // namespace

extern "C" {

int as_abi_version(void) { return AS_ABI_VERSION; }

const char *as_status_message(int status) {
  switch (status) {
    case AS_OK: return "ok";
    case AS_ERR_SHAPE: return "items should be at least of shape (2,2)";
    case AS_ERR_NONFINITE: return "Query item contains invalid values (NaN or infinity). All values must be finite.";
    case AS_ERR_ZERO_LAMBDA: return "Lambda of the item is 0.0, prepare the item before searching";
    case AS_ERR_NAN_SCORE: return "NaN similarity score (partial_cmp on NaN)";
    case AS_ERR_DIM: return "dimension mismatch between graph and signal";
    case AS_ERR_TOO_SPARSE: return "Resulting laplacian matrix is too sparse";
    case AS_ERR_ARG: return "invalid argument";
    case AS_ERR_UNSUPPORTED: return "unsupported on the accelerated path";
    case AS_ERR_HIP: return "HIP runtime error";
    case AS_ERR_NO_DEVICE: return "no gfx950 device available (libarrowspace_hip has no CPU fallback)";
    case AS_ERR_PEER: return "another rank of a sharded call failed";
    default: return "unknown status";
  }
}

int as_ctx_create(int device, as_ctx **out) {
  if (!out) return AS_ERR_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    return AS_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= count) return AS_ERR_ARG;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return AS_ERR_HIP;
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) return AS_ERR_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return AS_ERR_HIP;
  as_ctx *c = new as_ctx();
  c->device = device;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return AS_ERR_HIP;
  }
  c->own_stream = true;
  handle_cache_stream(device, c->stream, true);
  *out = c;
  return AS_OK;
}

void as_ctx_destroy(as_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->side) {
    (void)hipStreamSynchronize(ctx->side);
    (void)hipStreamDestroy(ctx->side);
  }
  if (ctx->side_start) (void)hipEventDestroy(ctx->side_start);
  if (ctx->side_done) (void)hipEventDestroy(ctx->side_done);
  if (ctx->side_word) (void)hipHostFree(ctx->side_word);
  prof_resolve(ctx);
  handle_cache_stream(ctx->device, ctx->stream, false);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  join_session_close(ctx);
  if (ctx->join_order) (void)hipFree(ctx->join_order);
  join_keep_clear(ctx);
  ctx->deferred.clear();  // (their blocks go to the cache that is freed next)
  for (auto &kv : ctx->free_blocks) (void)hipFree(kv.second);
  delete ctx;
}

const char *as_last_error(const as_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int as_ctx_set_stream(as_ctx *ctx, void *hip_stream) {
  AS_TRY(check_ctx(ctx));
  (void)hipStreamSynchronize(ctx->stream);
  prof_resolve(ctx);
  handle_cache_stream(ctx->device, ctx->stream, false);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  ctx->own_stream = false;
  ctx->stream = (hipStream_t)hip_stream;  // NULL = HIP's default (null) stream
  handle_cache_stream(ctx->device, ctx->stream, true);
  return AS_OK;
}

int as_ctx_synchronize(as_ctx *ctx) {
  AS_TRY(check_ctx(ctx));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return AS_OK;
}

int as_ctx_set_option(as_ctx *ctx, const char *name, int64_t value) {
  if (!ctx || !name) return AS_ERR_ARG;
  const std::string n(name);
  if (n == "knn_mfma") ctx->opt_knn_mfma = value;
  else if (n == "search_mfma") ctx->opt_search_mfma = value;
  else if (n == "cand_slack") ctx->opt_cand_slack = std::max<int64_t>(0, value);
  else if (n == "debug") {
    // Diagnostics.  Bits 1 .. 128 (except 8 = cycle stamps, 16 / 32 / 64 = variants that keep the results) and 0xC000 are ABLATIONS
    // for timing experiments: they switch parts of kernels off and leave the outputs UNDEFINED.  They are honoured only in a
    // process that asked for them (ARROWSPACE_ALLOW_ABLATION=1: scripts/explore_*.py); anywhere else they are masked off, so no
    // product call can come back with an undefined table because of a stray option.
    static const bool allow = [] {
      const char *e = std::getenv("ARROWSPACE_ALLOW_ABLATION");
      return e && e[0] == '1';
    }();
    const int64_t ablation = (0xC0FF & ~(int64_t)(8 | 16 | 32 | 64)) | 0xE000000;  // (bit 4 also: the lister form's "tiles not listed")  // (0xE000000: stages of the masked join's listing, mfma_join_f16.hip)
    ctx->opt_debug = allow ? value : (value & ~ablation);
  }
  else if (n == "lambda_v2") ctx->opt_lambda_v2 = value;
  else if (n == "lambda_v3") ctx->opt_lambda_v3 = value;
  else if (n == "knn_join") ctx->opt_knn_join = value;
  else if (n == "join_sample") ctx->opt_join_sample = value;
  else if (n == "join_rank") ctx->opt_join_rank = value;
  else if (n == "join_cap") ctx->opt_join_cap = value;
  else if (n == "join_min_rows") ctx->opt_join_min_rows = value;
  else if (n == "join_bf16") ctx->opt_join_bf16 = value;
  else if (n == "join_pipe") ctx->opt_join_pipe = (value == 32 || value == 48) ? value : 16;
  else if (n == "join_block") ctx->opt_join_block = value;
  else if (n == "search_scan") ctx->opt_search_scan = value;
  else if (n == "stream_scan") ctx->opt_stream_scan = value;
  else if (n == "lambda_wide") ctx->opt_lambda_wide = value;
  else if (n == "join_cone_tiles") ctx->opt_join_cone_tiles = value != 0;
  else if (n == "handle_cache_mb") handle_cache_limit(value);
  else if (n == "rerank_pair_bound") ctx->opt_rerank_pair_bound = value != 0;
  else if (n == "join_theta_parts") ctx->opt_join_theta_parts = value != 0;
  else if (n == "join_theta_band_second") ctx->opt_join_theta_band_second = value != 0;
  else if (n == "join_theta_band") ctx->opt_join_theta_band = std::min<int64_t>(std::max<int64_t>(0, value), 8);
  else if (n == "join_theta_rate") {
    int64_t r = 32;
    while (r > 2 && r > value) r >>= 1;
    ctx->opt_join_theta_rate = r;
  }
  else if (n == "join_unit_list") ctx->opt_join_unit_list = value != 0;
  else if (n == "join_cone_second") ctx->opt_join_cone_second = value != 0;
  else if (n == "join_cone_tiles_max_e3") ctx->opt_join_cone_tiles_max_e3 = std::max<int64_t>(0, value);
  else if (n == "sharded_own_rows") ctx->opt_sharded_own_rows = value != 0;
  else if (n == "signal_sym") ctx->opt_signal_sym = value;
  else if (n == "signal_order") ctx->opt_signal_order = value;
  else if (n == "signal_min_len") ctx->opt_signal_min_len = std::max<int64_t>(1, value);
  else if (n == "second_pass_f16") ctx->opt_second_pass_f16 = value;
  else if (n == "join_row_order") ctx->opt_join_row_order = value;
  else if (n == "join_prefilter") ctx->opt_join_prefilter = value;
  else if (n == "join_order_pivots") ctx->opt_join_order_pivots = value;
  else if (n == "join_order_pivots_auto") ctx->opt_join_order_pivots_auto = value;
  else if (n == "join_theta_grouped_bad_e3") ctx->opt_join_theta_grouped_bad_e3 = value;
  else if (n == "join_theta_grouped") ctx->opt_join_theta_grouped = value;
  else if (n == "search_scan_order") ctx->opt_search_scan_order = value;
  else if (n == "search_prefilter") ctx->opt_search_prefilter = value;
  else if (n == "search_stream_filter") ctx->opt_search_stream_filter = value;
  else if (n == "search_staged_min_batch") ctx->opt_search_staged_min_batch = value;
  else if (n == "cluster_on_device") ctx->opt_cluster_on_device = value;
  else if (n == "nearest_mfma") ctx->opt_nearest_mfma = value;
  else if (n == "rerank_share") ctx->opt_rerank_share = value != 0;
  else if (n == "join_order_chain") ctx->opt_join_order_chain = value;
  else if (n == "join_theta_band_auto") ctx->opt_join_theta_band_auto = value;
  else if (n == "join_second_prefilter") ctx->opt_join_second_prefilter = value;
  else if (n == "join_zproj_mfma") ctx->opt_join_zproj_mfma = value;
  else if (n == "join_pack_wide_tiles") ctx->opt_join_pack_wide_tiles = value;
  else if (n == "join_fused_rows_pass") ctx->opt_join_fused_rows_pass = value;
  else if (n == "join_theta_rows64") ctx->opt_join_theta_rows64 = value;
  else if (n == "join_one_sided_block") ctx->opt_join_one_sided_block = value;
  else if (n == "sharded_order_assign") ctx->opt_sharded_order_assign = value;
  else if (n == "join_basis_fused") ctx->opt_join_basis_fused = value;
  else if (n == "keep_join_images") {
    ctx->opt_keep_join_images = value;
    if (!value) join_keep_clear(ctx);
  } else if (n == "keep_join_images_min_rows") ctx->opt_keep_join_images_min_rows = value;
  else if (n == "index_adopt_join_images") ctx->opt_index_adopt_join_images = value;
  else if (n == "join_second_slices") ctx->opt_join_second_slices = value;
  else if (n == "search_scan_run") ctx->opt_search_scan_run = value;
  else if (n == "search_scan_slices") ctx->opt_search_scan_slices = value;
  else if (n == "join_masked_block") ctx->opt_join_masked_block = value;
  else if (n == "rerank_prune_halves") ctx->opt_rerank_prune_halves = value;
  else if (n == "sharded_split_assembly") ctx->opt_sharded_split_assembly = value;
  else if (n == "join_basis_sample") ctx->opt_join_basis_sample = value;
  else if (n == "keep_open_rows") ctx->opt_keep_open_rows = value;
  else if (n == "release_workspace") {  // an action, not a setting: the context's cached workspace blocks go back to the device now
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->free_blocks) (void)hipFree(kv.second);
    ctx->free_blocks.clear();
  }
  else if (n == "rerank_image_order") ctx->opt_rerank_image_order = value;
  else if (n == "join_open_order") ctx->opt_join_open_order = value;
  else if (n == "join_cone_groups") ctx->opt_join_cone_groups = value;
  else if (n == "join_cone_split_above_e3") ctx->opt_join_cone_split_above_e3 = value;
  else if (n == "debug_label_mod") ctx->opt_debug_label_mod = value;
  else if (n == "signal_tau_bracket") ctx->opt_signal_tau_bracket = value;
  else if (n == "signal_tau_overlap") ctx->opt_signal_tau_overlap = value;
  else if (n == "join_theta_far") ctx->opt_join_theta_far = value;
  else if (n == "join_theta_dip_e3") ctx->opt_join_theta_dip_e3 = value;
  else if (n == "join_order_cover") ctx->opt_join_order_cover = value;
  else if (n == "join_theta_band_slack") ctx->opt_join_theta_band_slack = value;
  else if (n == "rerank_stream_cap") ctx->opt_rerank_stream_cap = std::max<int64_t>(0, value);
  else if (n == "join_order_sub") ctx->opt_join_order_sub = value;
  else if (n == "join_basis_iters") ctx->opt_join_basis_iters = value;
  else if (n == "search_lambda_order") ctx->opt_search_lambda_order = value;
  else if (n == "sharded_agreement") ctx->opt_sharded_agreement = value;
  else if (n == "sharded_inject_failure") ctx->opt_sharded_inject_failure = value;
  else if (n == "search_rot_tail_e3") ctx->opt_search_rot_tail_e3 = value;
  else if (n == "assembly_lanes") ctx->opt_assembly_lanes = value;
  else if (n == "search_stream_quick") ctx->opt_search_stream_quick = value;
  else if (n == "join_prefilter_min_rows") ctx->opt_join_prefilter_min_rows = value;
  else if (n == "search_scan_f16") ctx->opt_search_scan_f16 = value;
  else if (n == "search_scan_sample") ctx->opt_search_scan_sample = value;
  else if (n == "search_scan_rank") ctx->opt_search_scan_rank = value;
  else if (n == "search_scan_cap") ctx->opt_search_scan_cap = value;
  else if (n == "cand_precision") ctx->opt_cand_precision = value == 64 ? 64 : 32;
  else return fail(ctx, AS_ERR_ARG, "unknown option " + n);
  return AS_OK;
}

int64_t as_ctx_get_stat(const as_ctx *ctx, const char *name) {
  if (!ctx || !name) return -1;
  if (!ctx->deferred.empty()) (void)asx::deferred_stats_flush(const_cast<as_ctx *>(ctx));  // counters parked by the last call (common.h)
  if (std::strncmp(name, "handle_cache_", 13) == 0) {  // process-wide: time spent in the device allocator on behalf of handles
    HandleCache &c = handle_cache();
    const std::string n(name + 13);
    if (n == "malloc_us") return (int64_t)c.us_malloc.load();
    if (n == "free_us") return (int64_t)c.us_free.load();
    if (n == "wait_us") return (int64_t)c.us_wait.load();
    if (n == "mallocs") return (int64_t)c.n_malloc.load();
    if (n == "frees") return (int64_t)c.n_free.load();
    if (n == "hits") return (int64_t)c.n_hit.load();
  }
  if (std::strcmp(name, "workspace_malloc_us") == 0) return (int64_t)ctx->ws_malloc_us;
  if (std::strcmp(name, "workspace_mallocs") == 0) return (int64_t)ctx->ws_mallocs;
  auto it = ctx->stats.find(name);
  return it == ctx->stats.end() ? -1 : it->second;
}

int as_profile_enable(as_ctx *ctx, int on) {
  if (!ctx) return AS_ERR_ARG;
  ctx->prof_on = on != 0;
  return AS_OK;
}
int as_profile_reset(as_ctx *ctx) {
  if (!ctx) return AS_ERR_ARG;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  prof_resolve(ctx);
  ctx->prof.clear();
  return AS_OK;
}
int as_profile_get(as_ctx *ctx, const char *name, double *ms_total, uint64_t *launches) {
  if (!ctx || !name) return AS_ERR_ARG;
  (void)hipSetDevice(ctx->device);
  prof_resolve(ctx);
  auto it = ctx->prof.find(name);
  if (ms_total) *ms_total = it == ctx->prof.end() ? 0.0 : it->second.ms;
  if (launches) *launches = it == ctx->prof.end() ? 0 : it->second.launches;
  return AS_OK;
}
int as_profile_names(as_ctx *ctx, char *buf, uint64_t cap) {
  if (!ctx || !buf || cap == 0) return AS_ERR_ARG;
  (void)hipSetDevice(ctx->device);
  prof_resolve(ctx);
  std::string s;
  for (auto &kv : ctx->prof) s += kv.first + "\n";
  std::strncpy(buf, s.c_str(), cap - 1);
  buf[cap - 1] = 0;
  return AS_OK;
}

int as_device_upload(as_ctx *ctx, const void *host, uint64_t bytes, void **dev_out) {
  AS_TRY(check_ctx(ctx));
  if (!dev_out || (bytes && !host)) return fail(ctx, AS_ERR_ARG, "null argument");
  *dev_out = nullptr;
  void *p = nullptr;
  AS_HIP(ctx, handle_malloc(&p, bytes ? bytes : 1));
  const hipError_t e = is_device_ptr(host) ? hipMemcpyAsync(p, host, bytes, hipMemcpyDeviceToDevice, ctx->stream) : copy_h2d(ctx, p, host, bytes);
  if (e != hipSuccess) {
    (void)handle_free(p);
    return fail(ctx, AS_ERR_HIP, std::string("as_device_upload: ") + hipGetErrorString(e));
  }
  *dev_out = p;
  return AS_OK;
}
int as_device_download(as_ctx *ctx, const void *dev, uint64_t bytes, void *host_out) {
  AS_TRY(check_ctx(ctx));
  if (bytes && (!dev || !host_out)) return fail(ctx, AS_ERR_ARG, "null argument");
  AS_HIP(ctx, copy_d2h(ctx, host_out, dev, bytes));
  return AS_OK;
}
void as_device_free(as_ctx *ctx, void *dev) {
  if (!dev) return;
  if (ctx) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
  }
  (void)handle_free(dev);
}

uint64_t as_define_result_k(uint64_t k, uint64_t topk) {
  if (k <= 5) return 3;
  if (k < 10) return 4;
  return topk;
}

// ---- graph --------------------------------------------------------------------------------

int as_knn_cosine(as_ctx *ctx, const double *rows, uint64_t n_nodes, uint64_t dim, uint64_t row_begin,
                  uint64_t row_count, const as_graph_params *params, uint64_t *nbr_idx, double *nbr_dist) {
  AS_TRY(check_ctx(ctx));
  AS_TRY(validate_graph_args(ctx, rows, n_nodes, dim, params));
  if (!nbr_idx || !nbr_dist) return fail(ctx, AS_ERR_ARG, "null output");
  if (row_begin + row_count > n_nodes) return fail(ctx, AS_ERR_ARG, "row range out of bounds");
  ctx->stats["knn_join"] = 0;            // (statistics of THIS call: the paths that do not join leave them alone)
  ctx->stats["knn_join_prefilter"] = 0;
  const uint32_t K = (uint32_t)params->topk + 1;
  DevIn<double> R;
  AS_TRY(R.bind(ctx, rows, n_nodes * dim));
  DevBuf<double> scaled;
  if (params->normalise) {
    AS_TRY(scaled.alloc(ctx, n_nodes * dim));
    AS_TRY(standardise_rows(ctx, R.p, n_nodes, dim, scaled.p));
    R.p = scaled.p;
  }
  DevBuf<double> norms;
  AS_TRY(norms.alloc(ctx, n_nodes));
  AS_TRY(row_norms(ctx, R.p, n_nodes, dim, norms.p));
  DevBuf<uint32_t> idx32;
  AS_TRY(idx32.alloc(ctx, row_count * K));
  DevOut<double> D;
  AS_TRY(D.bind(ctx, nbr_dist, row_count * K));
  DevOut<uint64_t> I;
  AS_TRY(I.bind(ctx, nbr_idx, row_count * K));
  AS_TRY(knn_dispatch(ctx, R.p, norms.p, n_nodes, dim, row_begin, row_count, K, params, idx32.p, D.p));
  hipLaunchKernelGGL(widen_idx_kernel, dim3(cdiv(row_count * K, 256)), dim3(256), 0, ctx->stream, idx32.p,
                     row_count * K, 0ull, I.p);
  AS_HIP(ctx, hipGetLastError());
  AS_TRY(D.finish(ctx));
  AS_TRY(I.finish(ctx));
  return AS_OK;
}

int as_nearest_centroids(as_ctx *ctx, const double *rows, uint64_t n_rows, uint64_t dim, const double *centroids,
                         uint64_t n_centroids, uint64_t *out_idx, double *out_dist2) {
  AS_TRY(check_ctx(ctx));
  if (!rows || !centroids || !out_idx || !out_dist2) return fail(ctx, AS_ERR_ARG, "null argument");
  if (dim == 0 || n_centroids == 0) return fail(ctx, AS_ERR_ARG, "dim and n_centroids must be positive");
  DevIn<double> R, Cn;
  AS_TRY(R.bind(ctx, rows, n_rows * dim));
  AS_TRY(Cn.bind(ctx, centroids, n_centroids * dim));
  DevOut<uint64_t> I;
  DevOut<double> D;
  AS_TRY(I.bind(ctx, out_idx, n_rows));
  AS_TRY(D.bind(ctx, out_dist2, n_rows));
  double qmax2 = 0.0;
  if (ctx->opt_nearest_mfma == 2) {  // (tests: the candidate path behind this entry point too)
    double a = 0.0, b = 0.0;
    AS_TRY(max_row_sqnorm(ctx, R.p, n_rows, dim, &a));
    AS_TRY(max_row_sqnorm(ctx, Cn.p, n_centroids, dim, &b));
    qmax2 = std::max(a, b);
  }
  AS_TRY(nearest_centroids(ctx, R.p, n_rows, dim, Cn.p, n_centroids, I.p, D.p, qmax2));
  AS_TRY(I.finish(ctx));
  AS_TRY(D.finish(ctx));
  return AS_OK;
}

// ---- row-sharded form of the symmetric threshold join (see include/arrowspace_hip.h) ----------------
static int join_args(as_ctx *ctx, const double *rows, uint64_t n, uint64_t d, uint64_t row_begin, uint64_t row_count,
                     const as_graph_params *params) {
  AS_TRY(check_ctx(ctx));
  AS_TRY(validate_graph_args(ctx, rows, n, d, params));
  if (row_begin + row_count > n || row_count == 0) return fail(ctx, AS_ERR_ARG, "row range out of bounds");
  if (params->normalise) return fail(ctx, AS_ERR_UNSUPPORTED, "threshold join: normalise = true takes as_knn_cosine");
  if (!is_device_ptr(rows)) return fail(ctx, AS_ERR_UNSUPPORTED, "threshold join: rows must be resident in HBM");
  return AS_OK;
}

int as_knn_join_thresholds(as_ctx *ctx, const double *rows, uint64_t n_nodes, uint64_t dim, uint64_t row_begin,
                           uint64_t row_count, const as_graph_params *params, float *theta_out) {
  AS_TRY(join_args(ctx, rows, n_nodes, dim, row_begin, row_count, params));
  if (!theta_out || !is_device_ptr(theta_out)) return fail(ctx, AS_ERR_ARG, "theta_out must be a device pointer");
  // first call of a sequence: norms and the packed image of all rows are built here and kept for the other two
  AS_TRY(join_session_open(ctx, rows, n_nodes, dim));
  {  // the looser second thresholds of these rows stay with the session (second pass of as_knn_join_finish)
    void *p = nullptr;
    AS_TRY(ws_alloc(ctx, sizeof(float) * row_count, &p, &ctx->join.theta2_cap));
    ctx->join.theta2 = (float *)p;
    ctx->join.theta2_begin = row_begin;
    ctx->join.theta2_rows = row_count;
  }
  const int rc = knn_join_thresholds(ctx, rows, ctx->join.norms, (uint32_t)n_nodes, (uint32_t)dim, (uint32_t)row_begin,
                                     (uint32_t)row_count, theta_out, ctx->join.packed, ctx->join.theta2,
                                     &ctx->join.theta2_valid, ctx->join.packedh);
  if (rc != AS_OK) join_session_close(ctx);
  return rc;
}

int as_knn_join_applies(as_ctx *ctx, uint64_t n_nodes, uint64_t dim, const as_graph_params *params) {
  if (!ctx || !params || params->normalise || n_nodes >= 0xFFFFFFFFull || dim >= 0xFFFFFFFFull || params->topk + 1 > 64) return 0;  // the join's lists serve K <= 64
  return knn_join_applies(ctx, (uint32_t)n_nodes, (uint32_t)dim) ? 1 : 0;
}

uint32_t as_knn_join_capacity(as_ctx *ctx, const as_graph_params *params) {
  if (!ctx || !params) return 0;
  return 2 * knn_join_cap(ctx, (uint32_t)params->topk + 1, true);  // words per row: a slot is (row id, approximate cosine)
}

int as_knn_join_pairs(as_ctx *ctx, const double *rows, uint64_t n_nodes, uint64_t dim, uint64_t row_begin,
                      uint64_t row_count, const as_graph_params *params, const float *theta_all, uint32_t cap,
                      uint32_t *cnt, uint32_t *cand) {
  AS_TRY(join_args(ctx, rows, n_nodes, dim, row_begin, row_count, params));
  if (!theta_all || !cnt || !cand || !is_device_ptr(theta_all) || !is_device_ptr(cnt) || !is_device_ptr(cand))
    return fail(ctx, AS_ERR_ARG, "theta_all / cnt / cand must be device pointers");
  if (cap < 2 || cap % 2 != 0) return fail(ctx, AS_ERR_ARG, "cap must be a positive even number of words");
  if (!join_session_matches(ctx, rows, n_nodes, dim)) AS_TRY(join_session_open(ctx, rows, n_nodes, dim));
  return knn_join_pairs(ctx, rows, ctx->join.norms, (uint32_t)n_nodes, (uint32_t)dim, (uint32_t)row_begin,
                        (uint32_t)row_count, theta_all, cap / 2, cnt, cand, ctx->join.packed, ctx->join.packed16,
                        ctx->join.packedh);
}

// ---- compact form of the candidate lists for the exchange between ranks --------------------------
// The lists are dense arrays (cap words per row) with a handful of slots used per (row, foreign rank); what crosses
// the links is row_offsets + the used slots only.
__global__ void join_used_kernel(const uint32_t *__restrict__ cnt, uint64_t n, uint32_t slots, uint32_t *__restrict__ used) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) used[i] = min(cnt[i], slots);
}
// one wave per row; pack: dense -> packed, else packed -> dense
__global__ __launch_bounds__(256) void join_repack_kernel(int pack, const uint32_t *__restrict__ cnt, uint64_t n, uint32_t slots,
                                                          const uint64_t *__restrict__ offsets, const uint2 *__restrict__ src,
                                                          uint2 *__restrict__ dst) {
  const int lane = threadIdx.x & 63;
  for (uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += (uint64_t)gridDim.x * 4) {
    const uint32_t used = min(cnt[r], slots);
    const uint64_t o = offsets[r];
    for (uint32_t e = lane; e < used; e += 64) {
      if (pack) dst[o + e] = src[r * slots + e];
      else dst[r * slots + e] = src[o + e];
    }
  }
}

int as_knn_join_pack(as_ctx *ctx, const uint32_t *cnt, const uint32_t *cand, uint64_t n_rows, uint32_t cap,
                     uint64_t *row_offsets, uint32_t *packed) {
  AS_TRY(check_ctx(ctx));
  if (!cnt || !cand || !row_offsets || !packed || cap < 2 || cap % 2 != 0) return fail(ctx, AS_ERR_ARG, "bad argument");
  if (!is_device_ptr(cnt) || !is_device_ptr(cand) || !is_device_ptr(row_offsets) || !is_device_ptr(packed))
    return fail(ctx, AS_ERR_ARG, "cnt / cand / row_offsets / packed must be device pointers");
  if (n_rows == 0) return AS_OK;
  DevBuf<uint32_t> used;
  AS_TRY(used.alloc(ctx, n_rows));
  hipLaunchKernelGGL(join_used_kernel, dim3(cdiv(n_rows, 256)), dim3(256), 0, ctx->stream, cnt, n_rows, cap / 2, used.p);
  AS_TRY(exclusive_scan_u32_u64(ctx, used.p, n_rows, row_offsets));
  hipLaunchKernelGGL(join_repack_kernel, dim3((unsigned)std::min<uint64_t>(cdiv(n_rows, 4), 1u << 20)), dim3(256), 0,
                     ctx->stream, 1, cnt, n_rows, cap / 2, row_offsets, (const uint2 *)cand, (uint2 *)packed);
  AS_HIP(ctx, hipGetLastError());
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));  // `used` goes back to the workspace cache
  return AS_OK;
}

int as_knn_join_unpack(as_ctx *ctx, const uint32_t *cnt, const uint32_t *packed, uint64_t n_rows, uint32_t cap,
                       uint32_t *cand_out) {
  AS_TRY(check_ctx(ctx));
  if (!cnt || !packed || !cand_out || cap < 2 || cap % 2 != 0) return fail(ctx, AS_ERR_ARG, "bad argument");
  if (!is_device_ptr(cnt) || !is_device_ptr(packed) || !is_device_ptr(cand_out))
    return fail(ctx, AS_ERR_ARG, "cnt / packed / cand_out must be device pointers");
  if (n_rows == 0) return AS_OK;
  DevBuf<uint32_t> used;
  DevBuf<uint64_t> offsets;
  AS_TRY(used.alloc(ctx, n_rows));
  AS_TRY(offsets.alloc(ctx, n_rows + 1));
  hipLaunchKernelGGL(join_used_kernel, dim3(cdiv(n_rows, 256)), dim3(256), 0, ctx->stream, cnt, n_rows, cap / 2, used.p);
  AS_TRY(exclusive_scan_u32_u64(ctx, used.p, n_rows, offsets.p));
  hipLaunchKernelGGL(join_repack_kernel, dim3((unsigned)std::min<uint64_t>(cdiv(n_rows, 4), 1u << 20)), dim3(256), 0,
                     ctx->stream, 0, cnt, n_rows, cap / 2, offsets.p, (const uint2 *)packed, (uint2 *)cand_out);
  AS_HIP(ctx, hipGetLastError());
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return AS_OK;
}

int as_knn_join_finish(as_ctx *ctx, const double *rows, uint64_t n_nodes, uint64_t dim, uint64_t row_begin,
                       uint64_t row_count, const as_graph_params *params, const float *theta_all, uint32_t n_src,
                       uint32_t cap, const uint32_t *cnt_src, const uint32_t *cand_src, uint64_t *nbr_idx,
                       double *nbr_dist) {
  AS_TRY(join_args(ctx, rows, n_nodes, dim, row_begin, row_count, params));
  if (!theta_all || !cnt_src || !cand_src || !nbr_idx || !nbr_dist || n_src == 0 || cap < 2 || cap % 2 != 0)
    return fail(ctx, AS_ERR_ARG, "null argument");
  if (!is_device_ptr(theta_all) || !is_device_ptr(cnt_src) || !is_device_ptr(cand_src))
    return fail(ctx, AS_ERR_ARG, "theta_all / cnt_src / cand_src must be device pointers");
  const uint32_t K = (uint32_t)params->topk + 1;
  if (!join_session_matches(ctx, rows, n_nodes, dim)) AS_TRY(join_session_open(ctx, rows, n_nodes, dim));
  int rc = AS_OK;
  {
    DevBuf<uint32_t> idx32;
    DevOut<double> D;
    DevOut<uint64_t> I;
    rc = idx32.alloc(ctx, row_count * K);
    if (rc == AS_OK) rc = D.bind(ctx, nbr_dist, row_count * K);
    if (rc == AS_OK) rc = I.bind(ctx, nbr_idx, row_count * K);
    if (rc == AS_OK) {
      const bool second = ctx->join.theta2_valid && ctx->join.theta2_begin == row_begin && ctx->join.theta2_rows == row_count;
      const JoinSources js{ctx->join.packed, theta_all, n_src, cap / 2, cnt_src, cand_src, second ? ctx->join.theta2 : nullptr,
                           ctx->join.packed16, ctx->join.packedh};
      rc = knn_mfma(ctx, rows, ctx->join.norms, n_nodes, dim, row_begin, row_count, K, params->rectify, idx32.p, D.p, &js);
      if (rc == AS_OK && params->self_in_results)
        rc = apply_self_in_results(ctx, rows, ctx->join.norms, dim, row_begin, row_count, K, params->rectify, idx32.p, D.p);
    }
    if (rc == AS_OK) {
      hipLaunchKernelGGL(widen_idx_kernel, dim3(cdiv(row_count * K, 256)), dim3(256), 0, ctx->stream, idx32.p,
                         row_count * K, 0ull, I.p);
      if (hipGetLastError() != hipSuccess) rc = fail(ctx, AS_ERR_HIP, "widen_idx_kernel launch failed");
    }
    if (rc == AS_OK) rc = D.finish(ctx);
    if (rc == AS_OK) rc = I.finish(ctx);
  }
  (void)hipStreamSynchronize(ctx->stream);
  join_session_close(ctx);  // last call of the sequence
  return rc;
}

int as_laplacian_from_knn_rows(as_ctx *ctx, const uint64_t *nbr_idx, const double *nbr_dist, uint64_t n_nodes,
                               const as_graph_params *params, uint64_t row_begin, uint64_t row_count, as_csr **out) {
  AS_TRY(check_ctx(ctx));
  if (!nbr_idx || !nbr_dist || !params || !out) return fail(ctx, AS_ERR_ARG, "null argument");
  if (row_begin + row_count > n_nodes) return fail(ctx, AS_ERR_ARG, "row range out of bounds");
  if (n_nodes < 2 || n_nodes >= 0xFFFFFFFFull) return fail(ctx, AS_ERR_ARG, "bad n_nodes");
  if (params->topk + 1 > 256) return fail(ctx, AS_ERR_UNSUPPORTED, "topk + 1 must be <= 256");
  const uint32_t K = (uint32_t)params->topk + 1;
  DevIn<uint64_t> I;
  AS_TRY(I.bind(ctx, nbr_idx, n_nodes * K));
  DevIn<double> D;
  AS_TRY(D.bind(ctx, nbr_dist, n_nodes * K));
  DevBuf<uint32_t> idx32;
  AS_TRY(idx32.alloc(ctx, n_nodes * K));
  hipLaunchKernelGGL(narrow_idx_kernel, dim3(cdiv(n_nodes * K, 256)), dim3(256), 0, ctx->stream, I.p, n_nodes * K,
                     idx32.p);
  AS_TRY(laplacian_from_knn(ctx, idx32.p, D.p, n_nodes, K, params, out, row_begin, row_count));
  // A caller's table may hold dist(i, j) != dist(j, i): a mutual pair then keeps each row's OWN weight and L is not bitwise
  // symmetric.  Only graphs whose distances the library computed itself are symmetric by construction (as_build_laplacian, the
  // sharded build: they set the flag again); here the one-pass bitwise check of the item-graph lambda decides (a row block
  // streams all of its entries).
  (*out)->known_symmetric = 0;
  return AS_OK;
}

int as_laplacian_from_knn(as_ctx *ctx, const uint64_t *nbr_idx, const double *nbr_dist, uint64_t n_nodes,
                          const as_graph_params *params, as_csr **out) {
  return as_laplacian_from_knn_rows(ctx, nbr_idx, nbr_dist, n_nodes, params, 0, n_nodes, out);
}

int as_build_laplacian(as_ctx *ctx, const double *rows, uint64_t n_nodes, uint64_t dim,
                       const as_graph_params *params, as_csr **out) {
  AS_TRY(check_ctx(ctx));
  if (!out) return fail(ctx, AS_ERR_ARG, "null output");
  *out = nullptr;
  AS_TRY(validate_graph_args(ctx, rows, n_nodes, dim, params));
  const uint32_t K = (uint32_t)params->topk + 1;
  DevIn<double> R;
  AS_TRY(R.bind(ctx, rows, n_nodes * dim));
  DevBuf<double> scaled;
  if (params->normalise) {
    AS_TRY(scaled.alloc(ctx, n_nodes * dim));
    AS_TRY(standardise_rows(ctx, R.p, n_nodes, dim, scaled.p));
    R.p = scaled.p;
  }
  DevBuf<double> norms, dist;
  DevBuf<uint32_t> idx;
  AS_TRY(norms.alloc(ctx, n_nodes));
  AS_TRY(idx.alloc(ctx, n_nodes * K));
  AS_TRY(dist.alloc(ctx, n_nodes * K));
  AS_TRY(row_norms(ctx, R.p, n_nodes, dim, norms.p));
  const uint64_t order_seq = ctx->join_order_seq;
  AS_TRY(knn_dispatch(ctx, R.p, norms.p, n_nodes, dim, 0, n_nodes, K, params, idx.p, dist.p));
  AS_TRY(laplacian_from_knn(ctx, idx.p, dist.p, n_nodes, K, params, out, 0, n_nodes));
  if (ctx->join_order_seq != order_seq && ctx->join_order_n == n_nodes) {
    // the neighbour search grouped the rows by direction: neighbours sit together in that order, which is what the item-graph
    // lambda kernels want to visit the rows in (as_csr_set_order)
    as_csr *m = *out;
    if (handle_malloc((void **)&m->order, sizeof(uint32_t) * n_nodes) == hipSuccess) {
      AS_HIP(ctx, hipMemcpyAsync(m->order, ctx->join_order, sizeof(uint32_t) * n_nodes, hipMemcpyDeviceToDevice, ctx->stream));
      m->order_state = 1;
    } else {
      (void)hipGetLastError();
      m->order = nullptr;
    }
  }
  return AS_OK;
}

uint64_t as_csr_n(const as_csr *m) { return m ? m->n : 0; }
uint64_t as_csr_cols(const as_csr *m) { return m ? (m->n_cols ? m->n_cols : m->n) : 0; }
uint64_t as_csr_row_begin(const as_csr *m) { return m ? m->row_begin : 0; }
uint64_t as_csr_nnz(const as_csr *m) { return m ? m->nnz : 0; }

int as_csr_export(as_ctx *ctx, const as_csr *m, uint64_t *indptr, uint64_t *indices, double *data) {
  AS_TRY(check_ctx(ctx));
  if (!m) return fail(ctx, AS_ERR_ARG, "null csr");
  auto copy = [&](void *dst, const void *src, size_t bytes) -> hipError_t {
    if (!dst || bytes == 0) return hipSuccess;
    if (is_device_ptr(dst)) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream);
    return copy_d2h(ctx, dst, src, bytes);
  };
  AS_HIP(ctx, copy(indptr, m->indptr, sizeof(uint64_t) * (m->n + 1)));
  AS_HIP(ctx, copy(indices, m->indices, sizeof(uint64_t) * m->nnz));
  AS_HIP(ctx, copy(data, m->data, sizeof(double) * m->nnz));
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return AS_OK;
}

int as_csr_import(as_ctx *ctx, uint64_t n, uint64_t nnz, const uint64_t *indptr, const uint64_t *indices,
                  const double *data, as_csr **out) {
  AS_TRY(check_ctx(ctx));
  if (!indptr || (nnz && (!indices || !data)) || !out) return fail(ctx, AS_ERR_ARG, "null argument");
  as_csr *m = new as_csr();
  m->device = ctx->device;
  m->n = n;
  m->n_cols = n;
  m->nnz = nnz;
  AS_HIP(ctx, handle_malloc((void **)&m->indptr, sizeof(uint64_t) * (n + 1)));
  AS_HIP(ctx, handle_malloc((void **)&m->indices, sizeof(uint64_t) * (nnz ? nnz : 1)));
  AS_HIP(ctx, handle_malloc((void **)&m->data, sizeof(double) * (nnz ? nnz : 1)));
  auto put = [&](void *dst, const void *src, size_t bytes) -> hipError_t {
    if (is_device_ptr(src)) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream);
    return copy_h2d(ctx, dst, src, bytes);
  };
  AS_HIP(ctx, put(m->indptr, indptr, sizeof(uint64_t) * (n + 1)));
  if (nnz) {
    AS_HIP(ctx, put(m->indices, indices, sizeof(uint64_t) * nnz));
    AS_HIP(ctx, put(m->data, data, sizeof(double) * nnz));
  }
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *out = m;
  return AS_OK;
}

void as_csr_free(as_csr *m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  if (m->slab) {
    (void)handle_free(m->slab);
  } else {
    (void)handle_free(m->indptr);
    (void)handle_free(m->indices);
    (void)handle_free(m->data);
  }
  if (m->packed) (void)handle_free(m->packed);
  if (m->packed_upper) (void)handle_free(m->packed_upper);
  if (m->v3_ent) (void)handle_free(m->v3_ent);
  if (m->v3_diag) (void)handle_free(m->v3_diag);
  signal_stream_release(m);
  if (m->order) (void)handle_free(m->order);
  delete m;
}

// ---- lambda-tau ----------------------------------------------------------------------------

static int lambdas_common(as_ctx *ctx, const double *items, uint64_t n, uint64_t F, const as_csr *L, int mode,
                          double arg, bool check_finite, double *out) {
  AS_TRY(check_ctx(ctx));
  if (!items || !L || !out) return fail(ctx, AS_ERR_ARG, "null argument");
  if ((L->n_cols && L->n_cols != L->n) || L->row_begin) return fail(ctx, AS_ERR_UNSUPPORTED, "a row block of a sharded graph is not a lambda graph (as_compute_taumode_signals_sharded takes one)");
  if (F >= 0xFFFFFFFFull) return fail(ctx, AS_ERR_UNSUPPORTED, "n_features too large");
  if (L->n > F) return fail(ctx, AS_ERR_DIM, "graph has more nodes than the items have features");
  if (mode < AS_TAU_FIXED || mode > AS_TAU_PERCENTILE) return fail(ctx, AS_ERR_ARG, "bad tau_mode");
  DevIn<double> X;
  AS_TRY(X.bind(ctx, items, n * F));
  DevOut<double> O;
  AS_TRY(O.bind(ctx, out, n));
  if (F >= (uint64_t)ctx->opt_signal_min_len) {
    // long signals (the rows-as-nodes graph of src/energymaps.rs:270): no per-item kernel holds one in LDS / registers; the
    // signals become columns and the graph is streamed once for all of them (signal_lambda.hip)
    if (check_finite) {
      bool bad = false;
      AS_TRY(any_nonfinite(ctx, X.p, n * F, &bad));
      if (bad) return fail(ctx, AS_ERR_NONFINITE, as_status_message(AS_ERR_NONFINITE));
    }
    AS_TRY(signal_lambdas_rows(ctx, X.p, n, F, const_cast<as_csr *>(L), mode, arg, O.p));
  } else {
    AS_TRY(taumode_lambdas(ctx, X.p, n, F, const_cast<as_csr *>(L), mode, arg, check_finite, O.p));
  }
  AS_TRY(O.finish(ctx));
  return AS_OK;
}

int as_compute_taumode_signals(as_ctx *ctx, const double *items, uint64_t n_items, uint64_t n_features, const as_csr *L, int tau_mode,
                               double tau_arg, double *lambdas_out) {
  AS_TRY(check_ctx(ctx));
  if (!items || !L || !lambdas_out) return fail(ctx, AS_ERR_ARG, "null argument");
  if ((L->n_cols && L->n_cols != L->n) || L->row_begin) return fail(ctx, AS_ERR_UNSUPPORTED, "a row block of a sharded graph: use as_compute_taumode_signals_sharded");
  if (n_features == 0 || n_features >= 0xFFFFFFFFull || n_items >= 0xFFFFFFFFull) return fail(ctx, AS_ERR_ARG, "bad shape");
  if (L->n > n_items) return fail(ctx, AS_ERR_DIM, "graph has more nodes than the signals have entries");
  if (tau_mode < AS_TAU_FIXED || tau_mode > AS_TAU_PERCENTILE) return fail(ctx, AS_ERR_ARG, "bad tau_mode");
  DevIn<double> X;
  AS_TRY(X.bind(ctx, items, n_items * n_features));
  DevOut<double> O;
  AS_TRY(O.bind(ctx, lambdas_out, n_features));
  AS_TRY(signal_lambdas(ctx, X.p, n_items, (uint32_t)n_features, const_cast<as_csr *>(L), tau_mode, tau_arg, O.p));
  AS_TRY(O.finish(ctx));
  return AS_OK;
}

int as_csr_set_order(as_ctx *ctx, as_csr *m, const uint32_t *order) {
  AS_TRY(check_ctx(ctx));
  if (!m) return fail(ctx, AS_ERR_ARG, "null argument");
  if (!order) return csr_set_order(ctx, m, nullptr);
  DevIn<uint32_t> P;
  AS_TRY(P.bind(ctx, order, m->n));
  return csr_set_order(ctx, m, P.p);
}

int as_compute_taumode(as_ctx *ctx, const double *items, uint64_t n_items, uint64_t n_features, const as_csr *L,
                       int tau_mode, double tau_arg, double *lambdas_out) {
  return lambdas_common(ctx, items, n_items, n_features, L, tau_mode, tau_arg, false, lambdas_out);
}

int as_prepare_query_items(as_ctx *ctx, const double *queries, uint64_t nq, uint64_t n_features, const as_csr *L,
                           int tau_mode, double tau_arg, double *lambdas_out) {
  return lambdas_common(ctx, queries, nq, n_features, L, tau_mode, tau_arg, true, lambdas_out);
}

int as_select_tau(as_ctx *ctx, const double *rows, uint64_t n_rows, uint64_t len, int tau_mode, double tau_arg,
                  double *taus_out) {
  AS_TRY(check_ctx(ctx));
  if ((!rows && n_rows * len) || !taus_out) return fail(ctx, AS_ERR_ARG, "null argument");
  if (len == 0) {  // empty slices: every mode but a valid Fixed returns the floor
    std::vector<double> h(n_rows, (tau_mode == AS_TAU_FIXED && std::isfinite(tau_arg) && tau_arg > 0.0) ? tau_arg : kTauFloor);
    if (is_device_ptr(taus_out)) AS_HIP(ctx, copy_h2d(ctx, taus_out, h.data(), sizeof(double) * n_rows));
    else std::memcpy(taus_out, h.data(), sizeof(double) * n_rows);
    return AS_OK;
  }
  DevIn<double> X;
  AS_TRY(X.bind(ctx, rows, n_rows * len));
  DevOut<double> O;
  AS_TRY(O.bind(ctx, taus_out, n_rows));
  if (len >= (uint64_t)ctx->opt_signal_min_len) AS_TRY(signal_taus_rows(ctx, X.p, n_rows, len, tau_mode, tau_arg, O.p));  // an order statistic of a long slice: signal_lambda.hip
  else AS_TRY(select_tau_rows(ctx, X.p, n_rows, len, tau_mode, tau_arg, O.p));
  AS_TRY(O.finish(ctx));
  return AS_OK;
}

// ---- search --------------------------------------------------------------------------------

// (as_ctx::JoinKeep: what a build kept for an index of its rows)
__global__ void count_differing_u64_kernel(const unsigned long long *__restrict__ a, const unsigned long long *__restrict__ b, uint64_t n,
                                           unsigned long long *__restrict__ out) {
  unsigned long long c = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) c += a[i] != b[i] ? 1ull : 0ull;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

int as_index_create(as_ctx *ctx, const double *items, const double *lambdas, uint64_t n_items, uint64_t n_features,
                    as_index **out) {
  AS_TRY(check_ctx(ctx));
  if (!items || !lambdas || !out) return fail(ctx, AS_ERR_ARG, "null argument");
  if (n_items == 0 || n_features == 0 || n_items >= 0xFFFFFFFFull) return fail(ctx, AS_ERR_ARG, "bad index shape");
  as_index *ix = new as_index();
  ix->device = ctx->device;
  ix->n = n_items;
  ix->d = n_features;
  if (is_device_ptr(items)) {
    ix->items = items;  // borrowed: the caller keeps it alive (ArrowSpace.data)
  } else {
    double *p;
    AS_HIP(ctx, handle_malloc((void **)&p, sizeof(double) * n_items * n_features));
    AS_HIP(ctx, copy_h2d(ctx, p, items, sizeof(double) * n_items * n_features));
    ix->items = p;
    ix->own_items = true;
  }
  if (is_device_ptr(lambdas)) {
    ix->lambdas = lambdas;
  } else {
    double *p;
    AS_HIP(ctx, handle_malloc((void **)&p, sizeof(double) * n_items));
    AS_HIP(ctx, copy_h2d(ctx, p, lambdas, sizeof(double) * n_items));
    ix->lambdas = p;
    ix->own_lambdas = true;
  }
  AS_HIP(ctx, handle_malloc((void **)&ix->norms, sizeof(double) * n_items));
  AS_TRY(row_norms(ctx, ix->items, n_items, n_features, ix->norms));
  // the order and the images of the item-graph build of these very rows, where the context kept them ("keep_join_images"): the same
  // device pointer, the same shape and, row by row, the same norms -- the caller's side of it is not to have written to the rows in
  // between (include/arrowspace_hip.h).  They move to the index; the first search then scans the ordered image behind the int8 pass.
  ctx->stats["index_adopted_join_images"] = 0;
  as_ctx::JoinKeep &jk = ctx->join_keep;
  if (jk.complete && ctx->opt_index_adopt_join_images && jk.rows == (const void *)ix->items && jk.n == n_items && jk.d == n_features) {
    DevBuf<unsigned long long> differ;
    AS_TRY(differ.alloc(ctx, 1));
    AS_HIP(ctx, hipMemsetAsync(differ.p, 0, sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(count_differing_u64_kernel, dim3(std::min<uint64_t>(cdiv(n_items, 256), 1024)), dim3(256), 0, ctx->stream,
                       (const unsigned long long *)ix->norms, (const unsigned long long *)jk.norms, n_items, differ.p);
    unsigned long long h = 1;
    AS_HIP(ctx, copy_d2h(ctx, &h, differ.p, sizeof(h)));
    if (h == 0) {
      ix->packedh = jk.packedh;
      ix->f16_err = jk.f16_err;
      ix->scan_perm = jk.perm;
      ix->scan_piv = jk.piv;
      ix->scan_np = jk.np;
      ix->rot_qw = jk.rot_qw;
      ix->packed8 = jk.packed8;
      ix->rot_err = jk.rot_err;
      ix->rot_state = 1;
      ix->z_space = true;
      jk.packedh = jk.packed8 = nullptr;
      jk.perm = nullptr;
      jk.piv = jk.rot_qw = nullptr;
      ctx->stats["index_adopted_join_images"] = 1;
    }
    join_keep_clear(ctx);
  }
  AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *out = ix;
  return AS_OK;
}

void as_index_destroy(as_index *ix) {
  if (!ix) return;
  (void)hipSetDevice(ix->device);
  if (ix->own_items) (void)handle_free((void *)ix->items);
  if (ix->own_lambdas) (void)handle_free((void *)ix->lambdas);
  (void)handle_free(ix->norms);
  if (ix->packed) (void)handle_free(ix->packed);
  if (ix->packed32) (void)handle_free(ix->packed32);
  if (ix->packed16) (void)handle_free(ix->packed16);
  if (ix->scan_sample) (void)handle_free(ix->scan_sample);
  if (ix->scan_sample_lam) (void)handle_free(ix->scan_sample_lam);
  if (ix->lam32) (void)handle_free(ix->lam32);
  if (ix->packedh) (void)handle_free(ix->packedh);
  if (ix->scan_perm) (void)handle_free(ix->scan_perm);
  if (ix->lam32_pos) (void)handle_free(ix->lam32_pos);
  if (ix->scan_piv) (void)handle_free(ix->scan_piv);
  if (ix->packed8) (void)handle_free(ix->packed8);
  if (ix->packed8_nat) (void)handle_free(ix->packed8_nat);
  if (ix->rot_q) (void)handle_free(ix->rot_q);
  if (ix->rot_qw) (void)handle_free(ix->rot_qw);
  delete ix;
}

uint64_t as_index_nitems(const as_index *ix) { return ix ? ix->n : 0; }

// core: device queries/lambdas -> internal (u32 idx, key = -score) tables, nq x k
static int search_core(as_ctx *ctx, const as_index *index, const double *Q, const double *qn, const double *QL,
                       uint64_t nq, uint32_t kk, double alpha, uint32_t *idx32, double *keys) {
  if (ctx->opt_search_mfma && kk <= 64)
    return search_mfma(ctx, const_cast<as_index *>(index), Q, qn, QL, nq, kk, alpha, idx32, keys);
  return search_exact(ctx, index, Q, qn, QL, nullptr, nq, nq, kk, alpha, idx32, keys);
}

int as_search_lambda_aware(as_ctx *ctx, const as_index *index, const double *queries, const double *q_lambdas,
                           uint64_t nq, uint64_t k, double alpha, uint64_t id_offset, uint64_t *idx_out,
                           double *score_out) {
  AS_TRY(check_ctx(ctx));
  if (!index || (nq && (!queries || !q_lambdas))) return fail(ctx, AS_ERR_ARG, "null argument");
  if (nq == 0 || k == 0) return AS_OK;
  if (!idx_out || !score_out) return fail(ctx, AS_ERR_ARG, "null output");
  if (k > 4096) return fail(ctx, AS_ERR_UNSUPPORTED, "k must be <= 4096");
  const uint64_t d = index->d;
  DevIn<double> Q, QL;
  AS_TRY(Q.bind(ctx, queries, nq * d));
  AS_TRY(QL.bind(ctx, q_lambdas, nq));
  // assert_ne!(query.lambda, 0.0), src/core.rs:773-776
  DevBuf<int> flag;
  AS_TRY(flag.alloc(ctx, 1));
  AS_HIP(ctx, hipMemsetAsync(flag.p, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL(any_zero_kernel, dim3(cdiv(nq, 256)), dim3(256), 0, ctx->stream, QL.p, nq, flag.p);
  int h = 0;
  AS_HIP(ctx, copy_d2h(ctx, &h, flag.p, sizeof(int)));
  if (h) return fail(ctx, AS_ERR_ZERO_LAMBDA, as_status_message(AS_ERR_ZERO_LAMBDA));

  const uint32_t kk = (uint32_t)k;
  DevBuf<double> qn, keys;
  DevBuf<uint32_t> idx32;
  AS_TRY(qn.alloc(ctx, nq));
  AS_TRY(keys.alloc(ctx, nq * kk));
  AS_TRY(idx32.alloc(ctx, nq * kk));
  AS_TRY(row_norms(ctx, Q.p, nq, d, qn.p));
  AS_TRY(search_core(ctx, index, Q.p, qn.p, QL.p, nq, kk, alpha, idx32.p, keys.p));
  DevOut<double> S;
  DevOut<uint64_t> I;
  AS_TRY(S.bind(ctx, score_out, nq * kk));
  AS_TRY(I.bind(ctx, idx_out, nq * kk));
  hipLaunchKernelGGL(negate_kernel, dim3(cdiv(nq * kk, 256)), dim3(256), 0, ctx->stream, keys.p, nq * kk, S.p);
  hipLaunchKernelGGL(widen_idx_kernel, dim3(cdiv(nq * kk, 256)), dim3(256), 0, ctx->stream, idx32.p, nq * kk,
                     id_offset, I.p);
  AS_HIP(ctx, hipGetLastError());
  AS_TRY(S.finish(ctx));
  AS_TRY(I.finish(ctx));
  return AS_OK;
}

int as_search_lambda_aware_hybrid(as_ctx *ctx, const as_index *index, const double *queries,
                                  const double *q_lambdas, uint64_t nq, uint64_t k, double alpha, uint64_t id_offset,
                                  uint64_t *idx_out, double *score_out) {
  AS_TRY(check_ctx(ctx));
  if (!index || (nq && (!queries || !q_lambdas))) return fail(ctx, AS_ERR_ARG, "null argument");
  if (nq == 0 || k == 0) return AS_OK;  // src/core.rs:811-813
  if (!idx_out || !score_out) return fail(ctx, AS_ERR_ARG, "null output");
  if (k > 4096) return fail(ctx, AS_ERR_UNSUPPORTED, "k must be <= 4096");
  const uint64_t d = index->d;
  const uint32_t kk = (uint32_t)std::min<uint64_t>(k, index->n);
  DevIn<double> Q, QL;
  AS_TRY(Q.bind(ctx, queries, nq * d));
  AS_TRY(QL.bind(ctx, q_lambdas, nq));
  DevBuf<double> qn, kb, ka;
  DevBuf<uint32_t> ib, ia;
  AS_TRY(qn.alloc(ctx, nq));
  AS_TRY(kb.alloc(ctx, nq * kk));
  AS_TRY(ka.alloc(ctx, nq * kk));
  AS_TRY(ib.alloc(ctx, nq * kk));
  AS_TRY(ia.alloc(ctx, nq * kk));
  AS_TRY(row_norms(ctx, Q.p, nq, d, qn.p));
  AS_TRY(search_core(ctx, index, Q.p, qn.p, QL.p, nq, kk, alpha, ib.p, kb.p));  // lambda-aware top-k
  AS_TRY(search_core(ctx, index, Q.p, qn.p, QL.p, nq, kk, 1.0, ia.p, ka.p));    // alpha = 1: score == cosine exactly
  DevOut<double> S;
  DevOut<uint64_t> I;
  AS_TRY(S.bind(ctx, score_out, nq * k));
  AS_TRY(I.bind(ctx, idx_out, nq * k));
  DevBuf<uint32_t> ui;
  DevBuf<double> us;
  AS_TRY(ui.alloc(ctx, nq * (2 * (uint64_t)kk + 1)));
  AS_TRY(us.alloc(ctx, nq * (2 * (uint64_t)kk + 1)));
  hipLaunchKernelGGL(hybrid_merge_kernel, dim3(cdiv(nq, 64)), dim3(64), 0, ctx->stream, ia.p, ka.p, ib.p, kb.p, nq,
                     kk, (uint32_t)k, id_offset, ui.p, us.p, I.p, S.p);
  AS_HIP(ctx, hipGetLastError());
  AS_TRY(S.finish(ctx));
  AS_TRY(I.finish(ctx));
  return AS_OK;
}

int as_range_search(as_ctx *ctx, const as_index *index, double lambda_q, double eps, uint64_t id_offset,
                    uint64_t *idx_out, double *dist_out, uint64_t cap, uint64_t *n_found) {
  AS_TRY(check_ctx(ctx));
  if (!index || !n_found) return fail(ctx, AS_ERR_ARG, "null argument");
  const uint64_t n = index->n;
  DevBuf<uint32_t> flags;
  DevBuf<uint64_t> pos;
  AS_TRY(flags.alloc(ctx, n));
  AS_TRY(pos.alloc(ctx, n + 1));
  hipLaunchKernelGGL(range_flag_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, index->lambdas, n, lambda_q, eps,
                     flags.p);
  AS_TRY(exclusive_scan_u32_u64(ctx, flags.p, n, pos.p));
  uint64_t found = 0;
  AS_HIP(ctx, copy_d2h(ctx, &found, pos.p + n, sizeof(uint64_t)));
  *n_found = found;
  const uint64_t m = std::min<uint64_t>(found, cap);
  if (m == 0) return AS_OK;
  if (!idx_out || !dist_out) return fail(ctx, AS_ERR_ARG, "null output");
  DevOut<double> D;
  DevOut<uint64_t> I;
  AS_TRY(D.bind(ctx, dist_out, m));
  AS_TRY(I.bind(ctx, idx_out, m));
  hipLaunchKernelGGL(range_scatter_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, index->lambdas, n, lambda_q,
                     flags.p, pos.p, m, id_offset, I.p, D.p);
  AS_HIP(ctx, hipGetLastError());
  AS_TRY(D.finish(ctx));
  AS_TRY(I.finish(ctx));
  return AS_OK;
}

}  // extern "C"
