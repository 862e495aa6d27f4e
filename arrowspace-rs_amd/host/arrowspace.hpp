// arrowspace.hpp -- C++17 header-only host mirror of the reference's public API for the
// lambda-tau path, over the C ABI of libarrowspace_hip.so (include/arrowspace_hip.h).
//
// The reference (tuned-org-uk/arrowspace-rs v0.18.1) is a Rust crate and there is no Rust
// toolchain in the build image, so the host side above the boundary is written in C++ with the
// reference's names, argument meaning and error behaviour (paths relative to /root/reference):
//   arrowspace::TauMode            src/taumode.rs:75-127
//   arrowspace::GraphParams        src/graph.rs:94-102
//   arrowspace::GraphLaplacian     src/graph.rs:127-135
//   arrowspace::GraphFactory       src/graph.rs:143-204
//   arrowspace::build_laplacian_matrix   src/laplacian.rs:122-178
//   arrowspace::ArrowItem          src/core.rs:84-239
//   arrowspace::ArrowSpace         src/core.rs:366-1006 (prepare_query_item, search_lambda_aware)
//   arrowspace::ArrowSpaceBuilder  src/builder.rs:20-455
// Where the reference panics this mirror throws arrowspace::Panic carrying the same message.
// Everything numeric happens in the HIP library; this header only owns handles and marshals
// std::vector<double> rows.  No CPU fallback exists.
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/arrowspace_hip.h"

namespace arrowspace {

struct Panic : std::runtime_error {
  int status;
  Panic(int s, const std::string &m) : std::runtime_error(m), status(s) {}
};

// ---- context ------------------------------------------------------------------------------
class Context {
 public:
  explicit Context(int device = 0) {
    int rc = as_ctx_create(device, &h_);
    if (rc) throw Panic(rc, as_status_message(rc));
  }
  ~Context() { as_ctx_destroy(h_); }
  Context(const Context &) = delete;
  Context &operator=(const Context &) = delete;
  as_ctx *get() const { return h_; }
  void check(int rc) const {
    if (rc) {
      std::string m = as_last_error(h_);
      throw Panic(rc, m.empty() ? as_status_message(rc) : m);
    }
  }
  static std::shared_ptr<Context> shared(int device = 0) {
    static std::shared_ptr<Context> c = std::make_shared<Context>(device);
    return c;
  }

 private:
  as_ctx *h_ = nullptr;
};

// ---- TauMode (src/taumode.rs:75-82) ---------------------------------------------------------
struct TauMode {
  as_tau_mode kind = AS_TAU_MEDIAN;
  double arg = 0.0;
  static TauMode Fixed(double t) { return {AS_TAU_FIXED, t}; }
  static TauMode Median() { return {AS_TAU_MEDIAN, 0.0}; }
  static TauMode Mean() { return {AS_TAU_MEAN, 0.0}; }
  static TauMode Percentile(double p) { return {AS_TAU_PERCENTILE, p}; }
  // TauMode::select_tau(energies, mode), src/taumode.rs:87-127
  static double select_tau(const std::vector<double> &energies, TauMode mode,
                           const std::shared_ptr<Context> &ctx = Context::shared()) {
    double out = 0.0;
    ctx->check(as_select_tau(ctx->get(), energies.data(), 1, energies.size(), mode.kind, mode.arg, &out));
    return out;
  }
};
constexpr double TAU_FLOOR = 1e-10;  // src/taumode.rs:84

// ---- GraphParams (src/graph.rs:94-102) ------------------------------------------------------
struct GraphParams {
  double eps = 1e-3;
  std::size_t k = 6;
  std::size_t topk = 3;
  double p = 2.0;
  std::optional<double> sigma;
  bool normalise = false;
  bool sparsity_check = false;
  // the two CosinePair variants of SURVEY.md Appendix A (not fields of the reference's struct; defaults = the oracle's rules)
  bool rectify = false;
  bool self_in_results = false;
  as_graph_params to_c() const {
    as_graph_params c{};
    c.eps = eps;
    c.k = k;
    c.topk = topk;
    c.p = p;
    c.sigma = sigma.value_or(0.0);
    c.has_sigma = sigma.has_value();
    c.normalise = normalise;
    c.sparsity_check = sparsity_check;
    c.rectify = rectify ? 1 : 0;
    c.self_in_results = self_in_results ? 1 : 0;
    c.reserved = 0;
    return c;
  }
};

// ---- CSR handle: the sprs::CsMat<f64> counterpart ---------------------------------------------
class CsMat {
 public:
  CsMat(std::shared_ptr<Context> ctx, as_csr *h) : ctx_(std::move(ctx)), h_(h) {}
  ~CsMat() { as_csr_free(h_); }
  CsMat(const CsMat &) = delete;
  CsMat &operator=(const CsMat &) = delete;
  static std::shared_ptr<CsMat> from_parts(const std::vector<uint64_t> &indptr, const std::vector<uint64_t> &indices,
                                           const std::vector<double> &data,
                                           std::shared_ptr<Context> ctx = Context::shared()) {
    as_csr *h = nullptr;
    ctx->check(as_csr_import(ctx->get(), indptr.size() - 1, indices.size(), indptr.data(), indices.data(),
                             data.data(), &h));
    return std::make_shared<CsMat>(ctx, h);
  }
  // (rows, columns): square unless the handle is a row block of a sharded graph
  std::pair<std::size_t, std::size_t> shape() const { return {as_csr_n(h_), as_csr_cols(h_)}; }
  std::size_t row_begin() const { return as_csr_row_begin(h_); }
  std::size_t rows() const { return as_csr_n(h_); }
  std::size_t nnz() const { return as_csr_nnz(h_); }
  as_csr *handle() const { return h_; }
  void fetch() const {
    if (!indptr_.empty()) return;
    indptr_.resize(rows() + 1);
    indices_.resize(nnz());
    data_.resize(nnz());
    ctx_->check(as_csr_export(ctx_->get(), h_, indptr_.data(), indices_.data(), data_.data()));
  }
  const std::vector<uint64_t> &indptr() const { return fetch(), indptr_; }
  const std::vector<uint64_t> &indices() const { return fetch(), indices_; }
  const std::vector<double> &data() const { return fetch(), data_; }
  double get(std::size_t i, std::size_t j) const {
    fetch();
    for (uint64_t e = indptr_[i]; e < indptr_[i + 1]; ++e)
      if (indices_[e] == j) return data_[e];
    return 0.0;
  }

 private:
  std::shared_ptr<Context> ctx_;
  as_csr *h_;
  mutable std::vector<uint64_t> indptr_, indices_;
  mutable std::vector<double> data_;
};

// ---- GraphLaplacian (src/graph.rs:127-135) ---------------------------------------------------
struct GraphLaplacian {
  std::vector<double> init_data;  // the matrix the graph was built from (rows = nodes), row-major
  std::size_t init_rows = 0, init_cols = 0;
  std::shared_ptr<CsMat> matrix;
  std::size_t nnodes = 0;  // item count of the ORIGINAL data, not the matrix dimension
  GraphParams graph_params;
  // src/graph.rs:572-578
  static double sparsity(const CsMat &m) {
    auto s = m.shape();
    return 1.0 - double(m.nnz()) / double(s.first * s.second);
  }
};

inline std::vector<double> flatten(const std::vector<std::vector<double>> &rows, std::size_t &n, std::size_t &d) {
  n = rows.size();
  d = n ? rows[0].size() : 0;
  std::vector<double> flat;
  flat.reserve(n * d);
  for (const auto &r : rows) {
    if (r.size() != d) throw Panic(AS_ERR_ARG, "All rows must have identical number of features");  // core.rs:421-424
    flat.insert(flat.end(), r.begin(), r.end());
  }
  return flat;
}

// build_laplacian_matrix(transposed, params, n_items), src/laplacian.rs:122-178.
// `transposed` rows are the graph NODES.
inline GraphLaplacian build_laplacian_matrix(const std::vector<std::vector<double>> &transposed,
                                             const GraphParams &params, std::optional<std::size_t> n_items = {},
                                             std::shared_ptr<Context> ctx = Context::shared()) {
  std::size_t n, d;
  GraphLaplacian gl;
  gl.init_data = flatten(transposed, n, d);
  gl.init_rows = n;
  gl.init_cols = d;
  as_graph_params c = params.to_c();
  as_csr *h = nullptr;
  ctx->check(as_build_laplacian(ctx->get(), gl.init_data.data(), n, d, &c, &h));
  gl.matrix = std::make_shared<CsMat>(ctx, h);
  gl.nnodes = n_items.value_or(n);
  gl.graph_params = params;
  return gl;
}

inline std::vector<std::vector<double>> transpose(const std::vector<std::vector<double>> &m) {
  if (m.empty()) return {};
  std::vector<std::vector<double>> t(m[0].size(), std::vector<double>(m.size()));
  for (std::size_t i = 0; i < m.size(); ++i)
    for (std::size_t j = 0; j < m[i].size(); ++j) t[j][i] = m[i][j];
  return t;
}

struct GraphFactory {
  // src/graph.rs:149-204: clustered is X x F; the graph nodes are its COLUMNS (features).
  static GraphLaplacian build_laplacian_matrix_from_k_cluster(const std::vector<std::vector<double>> &clustered,
                                                              double eps, std::size_t k, std::size_t topk, double p,
                                                              std::optional<double> sigma_override, bool normalise,
                                                              bool sparsity_check, std::size_t n_items,
                                                              std::shared_ptr<Context> ctx = Context::shared()) {
    if (clustered.size() > n_items) throw Panic(AS_ERR_ARG, "assertion failed: clustered.shape().0 <= n_items");
    GraphParams gp{eps, k, topk, p, sigma_override, normalise, sparsity_check};
    GraphLaplacian gl = build_laplacian_matrix(transpose(clustered), gp, n_items, ctx);
    if (sparsity_check) {
      const double s = GraphLaplacian::sparsity(*gl.matrix);
      if (s > 0.95) throw Panic(AS_ERR_TOO_SPARSE, "Resulting laplacian matrix is too sparse " + std::to_string(s));
    }
    return gl;
  }
};

// ---- ArrowItem (src/core.rs:84-239) -----------------------------------------------------------
// ---- clustering pre-stage (src/clustering.rs) and JL projection (src/reduction.rs) --------------
constexpr uint64_t kClusteringSeed = 128;  // src/clustering.rs:30

struct OptimalK {
  std::size_t k = 0, intrinsic_dim = 0;
  double radius = 0.0;  // squared-distance threshold of the incremental clustering
};
// ClusteringHeuristic::compute_optimal_k, src/clustering.rs:36-72 (seeded: DESIGN.md section 2 on what is statistical)
inline OptimalK compute_optimal_k(const std::vector<double> &flat, std::size_t n, std::size_t f, uint64_t seed = kClusteringSeed,
                                  std::shared_ptr<Context> ctx = Context::shared()) {
  uint64_t k = 0, idd = 0;
  OptimalK out;
  ctx->check(as_compute_optimal_k(ctx->get(), flat.data(), n, f, seed, &k, &out.radius, &idd));
  out.k = k;
  out.intrinsic_dim = idd;
  return out;
}
struct Clustering {
  std::vector<std::vector<double>> centroids;  // X x F
  std::vector<int64_t> assignments;            // -1: dropped as an outlier
  std::vector<uint64_t> sizes;
};
// run_incremental_clustering_with_sampling without a sampler, src/clustering.rs:547-910; batch = 1 is its sequential mode
inline Clustering incremental_clustering(const std::vector<double> &flat, std::size_t n, std::size_t f, std::size_t max_clusters,
                                         double radius, std::size_t batch = 4096,
                                         std::shared_ptr<Context> ctx = Context::shared()) {
  std::vector<double> cents(max_clusters * f, 0.0);
  std::vector<uint64_t> counts(max_clusters, 0);
  Clustering out;
  out.assignments.resize(n);
  uint64_t nc = 0;
  ctx->check(as_incremental_clustering(ctx->get(), flat.data(), n, f, max_clusters, radius, batch, cents.data(), &nc,
                                       out.assignments.data(), counts.data()));
  for (uint64_t c = 0; c < nc; ++c) out.centroids.emplace_back(cents.begin() + c * f, cents.begin() + (c + 1) * f);
  out.sizes.assign(counts.begin(), counts.begin() + nc);
  return out;
}
inline std::size_t compute_jl_dimension(std::size_t n_points, double epsilon) {  // src/reduction.rs:126-140
  return as_jl_dimension(n_points, epsilon);
}
// ImplicitProjection, src/reduction.rs:166-203: a seeded Gaussian projection held in HBM
class ImplicitProjection {
 public:
  std::size_t original_dim, reduced_dim;
  ImplicitProjection(std::size_t original, std::size_t reduced, uint64_t seed, std::shared_ptr<Context> ctx = Context::shared())
      : original_dim(original), reduced_dim(reduced), ctx_(std::move(ctx)) {
    ctx_->check(as_jl_create(ctx_->get(), original, reduced, seed, &h_));
  }
  ~ImplicitProjection() { as_jl_destroy(h_); }
  ImplicitProjection(const ImplicitProjection &) = delete;
  ImplicitProjection &operator=(const ImplicitProjection &) = delete;
  std::vector<double> project(const std::vector<double> &row) const {  // :180-203
    if (row.size() != original_dim) throw Panic(AS_ERR_DIM, "projection: dimension mismatch");
    std::vector<double> out(reduced_dim);
    ctx_->check(as_jl_project(ctx_->get(), h_, row.data(), 1, out.data()));
    return out;
  }
  std::vector<std::vector<double>> project_matrix(const std::vector<std::vector<double>> &rows) const {  // :142-164
    std::size_t n = 0, d = 0;
    std::vector<double> flat;
    for (const auto &r : rows) {
      if (r.size() != original_dim) throw Panic(AS_ERR_DIM, "projection: dimension mismatch");
      flat.insert(flat.end(), r.begin(), r.end());
      ++n;
    }
    (void)d;
    std::vector<double> out(n * reduced_dim);
    if (n) ctx_->check(as_jl_project(ctx_->get(), h_, flat.data(), n, out.data()));
    std::vector<std::vector<double>> res;
    for (std::size_t i = 0; i < n; ++i) res.emplace_back(out.begin() + i * reduced_dim, out.begin() + (i + 1) * reduced_dim);
    return res;
  }

 private:
  std::shared_ptr<Context> ctx_;
  as_jl *h_ = nullptr;
};

struct ArrowItem {
  std::vector<double> item;
  double lambda = 0.0;
  ArrowItem(std::vector<double> v, double l) : item(std::move(v)), lambda(l) {}
  std::size_t len() const { return item.size(); }
  bool is_empty() const { return item.empty(); }
};

// ---- ArrowSpace (src/core.rs:366-1006, lambda-tau subset) ---------------------------------------
class ArrowSpace {
 public:
  std::size_t nitems = 0, nfeatures = 0;
  TauMode taumode;

  ArrowSpace(const std::vector<std::vector<double>> &items, TauMode tm, std::shared_ptr<Context> ctx = Context::shared())
      : taumode(tm), ctx_(std::move(ctx)) {
    if (items.empty()) throw Panic(AS_ERR_ARG, "items cannot be empty");  // src/core.rs:416
    if (items.size() < 2) throw Panic(AS_ERR_ARG, "cannot create a arrowspace of one item only");
    data_ = flatten(items, nitems, nfeatures);
  }
  ~ArrowSpace() { as_index_destroy(index_); }
  ArrowSpace(const ArrowSpace &) = delete;
  ArrowSpace &operator=(const ArrowSpace &) = delete;

  const std::vector<double> &lambdas() const { return lambdas_; }
  // F x F spectral Laplacian (src/core.rs:372), set by GraphFactory::build_spectral_laplacian; when present it is
  // the graph of compute_taumode (src/taumode.rs:195-200) while queries keep gl.matrix (src/core.rs:548)
  std::shared_ptr<CsMat> signals;
  // clustering metadata and projection state (src/core.rs:373-384), set by the builder
  std::size_t n_clusters = 0;
  std::vector<int64_t> cluster_assignments;
  std::vector<uint64_t> cluster_sizes;
  double cluster_radius = 0.0;
  std::shared_ptr<ImplicitProjection> projection_matrix;
  std::size_t reduced_dim = 0;
  std::shared_ptr<Context> context() const { return ctx_; }
  const std::vector<double> &flat() const { return data_; }
  // src/core.rs:509-529
  std::vector<double> project_query(const std::vector<double> &query) const {
    return projection_matrix ? projection_matrix->project(query) : query;
  }
  ArrowItem get_item(std::size_t i) const {  // src/core.rs:586-594
    if (i >= nitems) throw Panic(AS_ERR_ARG, "item index out of bounds");
    return ArrowItem(std::vector<double>(data_.begin() + i * nfeatures, data_.begin() + (i + 1) * nfeatures),
                     lambdas_.empty() ? 0.0 : lambdas_[i]);
  }
  // EigenMaps::compute_taumode, src/eigenmaps.rs:358-383
  void compute_taumode(const GraphLaplacian &gl) {
    std::vector<double> out(nitems);
    ctx_->check(as_compute_taumode(ctx_->get(), data_.data(), nitems, nfeatures, (signals ? signals : gl.matrix)->handle(), taumode.kind,
                                   taumode.arg, out.data()));
    update_lambdas(std::move(out));
  }
  void update_lambdas(std::vector<double> l) {  // src/core.rs:979-1005
    if (l.size() != nitems) throw Panic(AS_ERR_ARG, "lambda count mismatch");
    lambdas_ = std::move(l);
    as_index_destroy(index_);
    index_ = nullptr;
  }
  // src/core.rs:533-549
  double prepare_query_item(const std::vector<double> &item, const GraphLaplacian &gl) const {
    if (item.size() != nfeatures)
      throw Panic(AS_ERR_DIM, "Query dimension " + std::to_string(item.size()) +
                                  " doesn't match index original dimension " + std::to_string(nfeatures));
    double out = 0.0;
    if (projection_matrix) {  // src/core.rs:541-546: the query is projected first
      for (double v : item)
        if (!std::isfinite(v)) throw Panic(AS_ERR_NONFINITE, as_status_message(AS_ERR_NONFINITE));
      const std::vector<double> q = projection_matrix->project(item);
      ctx_->check(as_prepare_query_items(ctx_->get(), q.data(), 1, q.size(), gl.matrix->handle(), taumode.kind, taumode.arg, &out));
      return out;
    }
    ctx_->check(as_prepare_query_items(ctx_->get(), item.data(), 1, nfeatures, gl.matrix->handle(), taumode.kind,
                                       taumode.arg, &out));
    return out;
  }
  // src/core.rs:760-798
  std::vector<std::pair<std::size_t, double>> search_lambda_aware(const ArrowItem &query, std::size_t k, double alpha) {
    if (query.len() != nfeatures) throw Panic(AS_ERR_DIM, "items should be of the same length");
    ensure_index();
    const std::size_t kk = k < nitems ? k : nitems;
    std::vector<uint64_t> idx(kk ? kk : 1);
    std::vector<double> sc(kk ? kk : 1);
    ctx_->check(as_search_lambda_aware(ctx_->get(), index_, query.item.data(), &query.lambda, 1, kk, alpha, 0,
                                       idx.data(), sc.data()));
    std::vector<std::pair<std::size_t, double>> out;
    for (std::size_t i = 0; i < kk; ++i)
      if (idx[i] != UINT64_MAX) out.emplace_back(idx[i], sc[i]);
    return out;
  }
  // src/core.rs:802-928
  std::vector<std::pair<std::size_t, double>> search_lambda_aware_hybrid(const ArrowItem &query, std::size_t k,
                                                                          double alpha) {
    if (k == 0) return {};
    ensure_index();
    std::vector<uint64_t> idx(k);
    std::vector<double> sc(k);
    ctx_->check(as_search_lambda_aware_hybrid(ctx_->get(), index_, query.item.data(), &query.lambda, 1, k, alpha, 0,
                                              idx.data(), sc.data()));
    std::vector<std::pair<std::size_t, double>> out;
    for (std::size_t i = 0; i < k; ++i)
      if (idx[i] != UINT64_MAX) out.emplace_back(idx[i], sc[i]);
    return out;
  }
  // src/core.rs:944-976
  std::vector<std::pair<std::size_t, double>> range_search(const ArrowItem &query, const GraphLaplacian &gl, double eps) {
    double lq = query.lambda;
    if (std::fabs(lq) <= 1e-9) lq = prepare_query_item(query.item, gl);  // relative_eq!(lambda, 0.0, epsilon = 1e-9)
    ensure_index();
    std::vector<uint64_t> idx(nitems);
    std::vector<double> dist(nitems);
    uint64_t found = 0;
    ctx_->check(as_range_search(ctx_->get(), index_, lq, eps, 0, idx.data(), dist.data(), nitems, &found));
    std::vector<std::pair<std::size_t, double>> out;
    for (uint64_t i = 0; i < found; ++i) out.emplace_back(idx[i], dist[i]);
    return out;
  }
  // EigenMaps::search, src/eigenmaps.rs:410-455
  std::vector<std::pair<std::size_t, double>> search(const std::vector<double> &item, const GraphLaplacian &gl,
                                                     std::size_t k, double alpha) {
    return search_lambda_aware(ArrowItem(item, prepare_query_item(item, gl)), k, alpha);
  }

 private:
  void ensure_index() {
    if (lambdas_.empty()) throw Panic(AS_ERR_ARG, "call compute_taumode(...) before search");  // eigenmaps.rs:425-431
    if (!index_)
      ctx_->check(as_index_create(ctx_->get(), data_.data(), lambdas_.data(), nitems, nfeatures, &index_));
  }
  std::shared_ptr<Context> ctx_;
  std::vector<double> data_, lambdas_;
  as_index *index_ = nullptr;
};


// GraphFactory::build_spectral_laplacian (src/graph.rs:212-270): the Laplacian of the Laplacian.  The F x F matrix
// of `gl` is made dense, transposed, and its rows become the nodes of a second build_laplacian_matrix with the
// same GraphParams; the result goes to aspace.signals.
inline void build_spectral_laplacian(ArrowSpace &aspace, const GraphLaplacian &gl) {
  const std::size_t n = gl.matrix->rows();
  std::vector<std::vector<double>> dense_t(n, std::vector<double>(n, 0.0));
  const auto &ip = gl.matrix->indptr();
  const auto &ix = gl.matrix->indices();
  const auto &dv = gl.matrix->data();
  for (std::size_t i = 0; i < n; ++i)
    for (uint64_t e = ip[i]; e < ip[i + 1]; ++e) dense_t[ix[e]][i] = dv[e];  // transposed
  GraphLaplacian sig = build_laplacian_matrix(dense_t, gl.graph_params, aspace.nitems, aspace.context());
  aspace.signals = sig.matrix;
  if (gl.graph_params.sparsity_check && GraphLaplacian::sparsity(*aspace.signals) > 0.95)
    throw Panic(AS_ERR_TOO_SPARSE, "Resulting spectral matrix is too sparse");
}

// ---- ArrowSpaceBuilder (src/builder.rs:20-455, lambda-tau subset) --------------------------------
class ArrowSpaceBuilder {
 public:
  // defaults: src/builder.rs:59-91
  double lambda_eps = 1e-3;
  std::size_t lambda_k = 6, lambda_topk = 3;
  double lambda_p = 2.0;
  std::optional<double> lambda_sigma;
  bool normalise = false, sparsity_check = false, prebuilt_spectral = false;
  TauMode synthesis = TauMode::Median();
  // clustering / projection (src/builder.rs:38-57)
  bool use_dims_reduction = false, deterministic_clustering = false;
  double rp_eps = 0.3;
  uint64_t clustering_seed = kClusteringSeed;
  std::size_t clustering_batch = 4096;
  std::size_t cluster_max_clusters = 0;
  double cluster_radius = 0.0;

  static ArrowSpaceBuilder new_() { return ArrowSpaceBuilder(); }
  // src/builder.rs:109-137
  ArrowSpaceBuilder &with_lambda_graph(double eps, std::size_t k, std::size_t topk, double p,
                                       std::optional<double> sigma_override) {
    lambda_eps = eps;
    lambda_k = k;
    lambda_topk = topk;
    lambda_p = p;
    lambda_sigma = sigma_override;
    return *this;
  }
  ArrowSpaceBuilder &with_synthesis(TauMode tm) { return synthesis = tm, *this; }
  ArrowSpaceBuilder &with_normalisation(bool n) { return normalise = n, *this; }
  ArrowSpaceBuilder &with_sparsity_check(bool f) { return sparsity_check = f, *this; }
  ArrowSpaceBuilder &with_spectral(bool f) { return prebuilt_spectral = f, *this; }  // src/builder.rs:157-162
  // src/builder.rs:164-187: a seed makes the clustering sequential (deterministic); dims reduction = JL projection
  ArrowSpaceBuilder &with_seed(uint64_t seed) { return clustering_seed = seed, deterministic_clustering = true, *this; }
  ArrowSpaceBuilder &with_dims_reduction(bool enable, std::optional<double> eps = std::nullopt) {
    use_dims_reduction = enable;
    if (eps) rp_eps = *eps;
    return *this;
  }
  // EigenMaps::start_clustering, src/eigenmaps.rs:175-290: optimal K + radius, incremental clustering, optional projection
  std::vector<std::vector<double>> start_clustering(ArrowSpace &aspace) {
    auto ctx = aspace.context();
    const OptimalK ok = compute_optimal_k(aspace.flat(), aspace.nitems, aspace.nfeatures, clustering_seed, ctx);
    cluster_max_clusters = ok.k;
    cluster_radius = ok.radius;
    Clustering cl = incremental_clustering(aspace.flat(), aspace.nitems, aspace.nfeatures, ok.k, ok.radius,
                                           deterministic_clustering ? 1 : clustering_batch, ctx);
    aspace.n_clusters = cl.centroids.size();
    aspace.cluster_assignments = std::move(cl.assignments);
    aspace.cluster_sizes = std::move(cl.sizes);
    aspace.cluster_radius = ok.radius;
    return maybe_project(aspace, std::move(cl.centroids));
  }
  std::vector<std::vector<double>> maybe_project(ArrowSpace &aspace, std::vector<std::vector<double>> cents) {
    const std::size_t nf = aspace.nfeatures;
    if (use_dims_reduction && nf > 64) {  // src/eigenmaps.rs:247-280
      const std::size_t target = std::min<std::size_t>(compute_jl_dimension(cents.size(), rp_eps), nf / 2);
      if (target < nf) {
        aspace.projection_matrix = std::make_shared<ImplicitProjection>(nf, target, clustering_seed ? clustering_seed : kClusteringSeed,
                                                                        aspace.context());
        aspace.reduced_dim = target;
        return aspace.projection_matrix->project_matrix(cents);
      }
    }
    return cents;
  }
  // src/builder.rs:225-233
  void define_result_k() { lambda_topk = as_define_result_k(lambda_k, lambda_topk); }

  // src/builder.rs:249-455.  Stage 1: clustering (skipped when the centroids, X x F, are supplied) + optional JL
  // projection; stage 2: eigenmaps (Laplacian over centroids^T); stage 3 (optional): spectral; stage 4: taumode.
  std::pair<std::unique_ptr<ArrowSpace>, GraphLaplacian> build(const std::vector<std::vector<double>> &rows,
                                                               const std::vector<std::vector<double>> &centroids = {},
                                                               std::shared_ptr<Context> ctx = Context::shared()) {
    define_result_k();
    auto aspace = std::make_unique<ArrowSpace>(rows, synthesis, ctx);
    const std::vector<std::vector<double>> cents = centroids.empty() ? start_clustering(*aspace) : maybe_project(*aspace, centroids);
    GraphLaplacian gl = GraphFactory::build_laplacian_matrix_from_k_cluster(
        cents, lambda_eps, lambda_k, lambda_topk, lambda_p, lambda_sigma, normalise,
        sparsity_check, aspace->nitems, ctx);
    if (prebuilt_spectral) build_spectral_laplacian(*aspace, gl);  // stage 3, src/eigenmaps.rs:325-345
    aspace->compute_taumode(gl);
    return {std::move(aspace), std::move(gl)};
  }
};

// ---- row-sharded build, one process per GPU (include/arrowspace_hip.h: as_comm_*) --------------
// The exchange sequence runs inside the library; this only owns the handle.  rows_shard: this rank's rows (host or HBM).
class Comm {
 public:
  static std::array<unsigned char, 128> unique_id() {
    std::array<unsigned char, 128> id{};
    const int rc = as_comm_unique_id(id.data());
    if (rc) throw Panic(rc, as_status_message(rc));
    return id;
  }
  Comm(const std::array<unsigned char, 128> &id, int rank, int world, std::shared_ptr<Context> ctx = Context::shared())
      : ctx_(std::move(ctx)) {
    ctx_->check(as_comm_create_rccl(ctx_->get(), id.data(), rank, world, &h_));
  }
  Comm(const as_comm_ops &ops, int rank, int world, std::shared_ptr<Context> ctx = Context::shared()) : ctx_(std::move(ctx)) {
    ctx_->check(as_comm_create_custom(ctx_->get(), &ops, rank, world, &h_));
  }
  ~Comm() { as_comm_destroy(h_); }
  Comm(const Comm &) = delete;
  Comm &operator=(const Comm &) = delete;
  int rank() const { return as_comm_rank(h_); }
  int world() const { return as_comm_world(h_); }
  std::pair<uint64_t, uint64_t> shard_range(uint64_t n) const {
    uint64_t lo = 0, hi = 0;
    as_shard_range(n, rank(), world(), &lo, &hi);
    return {lo, hi};
  }
  // this rank's row block of the item-graph Laplacian (nullptr when it owns no rows) and nnz of the whole matrix
  std::pair<std::shared_ptr<CsMat>, uint64_t> build_laplacian(const double *rows_shard, uint64_t n_total, uint64_t dim,
                                                             const GraphParams &gp) {
    const as_graph_params c = gp.to_c();
    as_csr *block = nullptr;
    uint64_t nnz = 0;
    ctx_->check(as_build_laplacian_sharded(ctx_->get(), h_, rows_shard, n_total, dim, &c, &block, &nnz));
    return {block ? std::make_shared<CsMat>(ctx_, block) : nullptr, nnz};
  }

 private:
  std::shared_ptr<Context> ctx_;
  as_comm *h_ = nullptr;
};

}  // namespace arrowspace
