// common.h -- context, device-buffer staging and launch helpers shared by the kernels of
// libarrowspace_hip.so (gfx950 only; no CPU path).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/arrowspace_hip.h"

namespace asx {

constexpr uint32_t kNoIdx = 0xFFFFFFFFu;
constexpr double kTauFloor = 1e-10;  // src/taumode.rs:84

struct ProfEntry {
  double ms = 0.0;
  uint64_t launches = 0;
};

struct PendingEvent {
  std::string name;
  hipEvent_t start, stop;
};

}  // namespace asx

struct as_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  // options
  int64_t opt_knn_mfma = 1;       // 1: MFMA candidates + exact re-rank, 0: exact scan only
  int64_t opt_search_mfma = 1;    // same for the search scan
  int64_t opt_cand_slack = 8;     // extra candidates kept by the MFMA pass beyond topk+1
  int64_t opt_waves_per_simd = 2; // f32 candidate pass: 2 = 8 waves x 32 rows, 1 = 4 waves x 64 rows
  int64_t opt_debug = 0;          // diagnostics only (timing experiments); 0 in production
  int64_t opt_cand_precision = 32;  // first candidate stage: 32 = f32 MFMA (then f64), 64 = f64 MFMA only
  // stats of the last call
  std::map<std::string, int64_t> stats;
  // profiling
  bool prof_on = false;
  std::map<std::string, asx::ProfEntry> prof;
  std::vector<asx::PendingEvent> pending;
};

struct as_csr {
  int device = 0;
  uint64_t n = 0, nnz = 0;
  uint64_t *indptr = nullptr;   // n+1 (device)
  uint64_t *indices = nullptr;  // nnz (device)
  double *data = nullptr;       // nnz (device)
  // packed (row, col, value) entries for the lambda kernel, built on first use
  void *packed = nullptr;
  bool symmetric_checked = false;
};

struct as_index {
  int device = 0;
  uint64_t n = 0, d = 0;
  const double *items = nullptr;    // n x d (device); owned iff own_items
  const double *lambdas = nullptr;  // n (device); owned iff own_lambdas
  bool own_items = false, own_lambdas = false;
  double *norms = nullptr;          // n exact norms
  double *packed = nullptr;         // fragment-packed unit rows, f64 image (lazy)
  float *packed32 = nullptr;        // fragment-packed unit rows, f32 image (lazy)
};

namespace asx {

inline int fail(as_ctx *ctx, int code, const std::string &msg) {
  if (ctx) ctx->err = msg;
  return code;
}

#define AS_HIP(ctx, expr)                                                                  \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      return asx::fail(ctx, AS_ERR_HIP,                                                    \
                       std::string(#expr) + ": " + hipGetErrorString(_e) + " @" + __FILE__ + \
                           ":" + std::to_string(__LINE__));                                \
    }                                                                                      \
  } while (0)

#define AS_TRY(expr)            \
  do {                          \
    int _rc = (expr);           \
    if (_rc != AS_OK) return _rc; \
  } while (0)

inline bool is_device_ptr(const void *p) {
  if (!p) return false;
  hipPointerAttribute_t a;
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // clear: plain host memory is reported as an error
    return false;
  }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// Stream-ordered device allocation that frees itself.
template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t count = 0;
  hipStream_t s = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  int alloc(as_ctx *ctx, size_t n) {
    release();
    s = ctx->stream;
    count = n;
    if (n == 0) n = 1;
    AS_HIP(ctx, hipMallocAsync((void **)&p, n * sizeof(T), s));
    return AS_OK;
  }
  void release() {
    if (p) (void)hipFreeAsync(p, s);
    p = nullptr;
  }
  T *take() {
    T *q = p;
    p = nullptr;
    return q;
  }
};

// Input that may live on the host: staged to HBM when needed.
template <typename T>
struct DevIn {
  const T *p = nullptr;
  DevBuf<T> tmp;
  int bind(as_ctx *ctx, const T *src, size_t n) {
    if (n == 0 || is_device_ptr(src)) {
      p = src;
      return AS_OK;
    }
    AS_TRY(tmp.alloc(ctx, n));
    AS_HIP(ctx, hipMemcpyAsync(tmp.p, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    p = tmp.p;
    return AS_OK;
  }
};

// Output that may live on the host: computed in HBM, copied back by finish().
template <typename T>
struct DevOut {
  T *p = nullptr;
  T *host = nullptr;
  size_t n = 0;
  DevBuf<T> tmp;
  int bind(as_ctx *ctx, T *dst, size_t count) {
    n = count;
    if (count == 0 || is_device_ptr(dst)) {
      p = dst;
      return AS_OK;
    }
    host = dst;
    AS_TRY(tmp.alloc(ctx, count));
    p = tmp.p;
    return AS_OK;
  }
  // Host destination: D2H + stream sync so the buffer is valid on return.  Device destination:
  // synchronous too while the context runs on its own private stream; once the caller has
  // installed a stream (as_ctx_set_stream) results are ordered on that stream instead.
  int finish(as_ctx *ctx) {
    if (host && n) {
      AS_HIP(ctx, hipMemcpyAsync(host, p, n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
      AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    } else if (ctx->own_stream) {
      AS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return AS_OK;
  }
};

// Scoped kernel-group timer: hipEvents on the context's stream, resolved lazily.
struct ProfScope {
  as_ctx *ctx;
  PendingEvent ev;
  bool active;
  ProfScope(as_ctx *c, const char *name) : ctx(c), active(c->prof_on) {
    if (!active) return;
    ev.name = name;
    if (hipEventCreate(&ev.start) != hipSuccess || hipEventCreate(&ev.stop) != hipSuccess) {
      active = false;
      return;
    }
    (void)hipEventRecord(ev.start, ctx->stream);
  }
  ~ProfScope() {
    if (!active) return;
    (void)hipEventRecord(ev.stop, ctx->stream);
    ctx->pending.push_back(ev);
  }
};

inline void prof_resolve(as_ctx *ctx) {
  for (auto &e : ctx->pending) {
    float ms = 0.f;
    if (hipEventSynchronize(e.stop) == hipSuccess && hipEventElapsedTime(&ms, e.start, e.stop) == hipSuccess) {
      auto &p = ctx->prof[e.name];
      p.ms += ms;
      p.launches += 1;
    }
    (void)hipEventDestroy(e.start);
    (void)hipEventDestroy(e.stop);
  }
  ctx->pending.clear();
}

inline unsigned cdiv(uint64_t a, uint64_t b) { return (unsigned)((a + b - 1) / b); }

// ---- stage entry points implemented in the .hip files ---------------------------------------
// exact arithmetic row norms: norms[i] = sqrt(sum_c x[i][c]^2), sequential (src/core.rs:212-216)
int row_norms(as_ctx *ctx, const double *rows, uint64_t n, uint64_t d, double *norms);

struct KnnOut {
  uint32_t *idx;  // rows x K, kNoIdx when missing
  double *dist;   // rows x K, +inf when missing
};

// exact neighbour scan for the listed rows (row_ids == nullptr: rows row_begin..+row_count)
int knn_exact(as_ctx *ctx, const double *rows, const double *norms, uint64_t n, uint64_t d,
              const uint32_t *row_ids, uint64_t row_begin, uint64_t row_count, uint32_t K,
              int rectify, uint32_t *out_idx, double *out_dist, uint64_t out_stride_rows);

// MFMA candidate pass + exact re-rank + certification + exact fall-back
int knn_mfma(as_ctx *ctx, const double *rows, const double *norms, uint64_t n, uint64_t d,
             uint64_t row_begin, uint64_t row_count, uint32_t K, int rectify, uint32_t *out_idx,
             double *out_dist);

int laplacian_from_knn(as_ctx *ctx, const uint32_t *nbr_idx, const double *nbr_dist, uint64_t n,
                       uint32_t K, const as_graph_params *gp, as_csr **out);

int taumode_lambdas(as_ctx *ctx, const double *items, uint64_t n_items, uint64_t F, as_csr *L,
                    int mode, double arg, bool check_finite, double *lambdas_out);
int select_tau_rows(as_ctx *ctx, const double *rows, uint64_t n_rows, uint64_t len, int mode,
                    double arg, double *taus);

int search_exact(as_ctx *ctx, const as_index *ix, const double *queries, const double *q_norms,
                 const double *q_lambdas, const uint32_t *q_ids, uint64_t nq_sel, uint64_t nq,
                 uint32_t k, double alpha, uint32_t *out_idx, double *out_score);
int search_mfma(as_ctx *ctx, as_index *ix, const double *queries, const double *q_norms,
                const double *q_lambdas, uint64_t nq, uint32_t k, double alpha, uint32_t *out_idx,
                double *out_score);

// device exclusive scan of uint32 -> uint64 (out has n+1 entries, out[n] = total)
int exclusive_scan_u32_u64(as_ctx *ctx, const uint32_t *in, uint64_t n, uint64_t *out);

}  // namespace asx
