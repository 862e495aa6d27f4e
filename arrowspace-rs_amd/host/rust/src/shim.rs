//! shim.rs -- the bodies a maintainer of the reference crate swaps in under `--features hip`.
//!
//! UNVERIFIED: the build image has no rustc / cargo, so this file has never been compiled.  It is written against
//! `hip.rs` (the generated `extern "C"` block, checked against include/arrowspace_hip.h by tests/test_abi.py) and against
//! the signatures of the reference at v0.18.1; the SAME call sequences are what the C++ mirror (host/arrowspace.hpp,
//! tests/cpp/test_host_mirror.cpp) and the Python mirror (arrowspace_amd/api.py) run and test on the GPU.
//!
//! What is replaced, and by which export:
//!   ArrowSpaceBuilder::build            src/builder.rs:249-455     upload once -> stages below on the device copy
//!   EigenMaps::start_clustering         src/eigenmaps.rs:175-290   as_compute_optimal_k + as_incremental_clustering (+ as_jl_*)
//!   EigenMaps::eigenmaps                src/eigenmaps.rs:292-356   as_build_laplacian on centroids^T (GraphFactory semantics)
//!   EigenMaps::compute_taumode          src/eigenmaps.rs:358-383   as_compute_taumode
//!   EigenMaps::search                   src/eigenmaps.rs:410-455   as_prepare_query_items + as_search_lambda_aware
//!   build_laplacian_matrix              src/laplacian.rs:122-178   as_build_laplacian
//!   (several GPUs)                      --                         as_build_laplacian_sharded / as_lambda_stats_sharded /
//!                                                                  as_search_lambda_aware_sharded
#![cfg(feature = "hip")]
#![allow(dead_code)]

use crate::hip::{self, AsCsr, AsCtx, AsGraphParams, AsIndex};
use std::os::raw::c_void;

/// One context per host thread (the library is not re-entrant per context).
pub(crate) fn thread_ctx() -> *mut AsCtx {
    thread_local! {
        static CTX: *mut AsCtx = {
            assert_eq!(unsafe { hip::as_abi_version() }, hip::AS_ABI_VERSION, "libarrowspace_hip ABI mismatch");
            let mut c = std::ptr::null_mut();
            let rc = unsafe { hip::as_ctx_create(0, &mut c) };
            assert_eq!(rc, 0, "as_ctx_create failed: no gfx950 device (there is no CPU fallback in the library)");
            c
        };
    }
    CTX.with(|c| *c)
}

/// The reference panics where the library returns a status; the text is the reference's own.
pub(crate) fn check(ctx: *const AsCtx, rc: i32) {
    if rc != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(hip::as_last_error(ctx)) }.to_string_lossy().into_owned();
        panic!("{}", msg);
    }
}

pub(crate) fn tau_to_c(t: crate::taumode::TauMode) -> (i32, f64) {
    use crate::taumode::TauMode::*;
    match t {
        Fixed(v) => (0, v),
        Median => (1, 0.0),
        Mean => (2, 0.0),
        Percentile(p) => (3, p),
    }
}

pub(crate) fn graph_params_to_c(p: &crate::graph::GraphParams) -> AsGraphParams {
    AsGraphParams {
        eps: p.eps,
        k: p.k as u64,
        topk: p.topk as u64,
        p: p.p,
        sigma: p.sigma.unwrap_or(0.0),
        has_sigma: p.sigma.is_some() as i32,
        normalise: p.normalise as i32,
        sparsity_check: p.sparsity_check as i32,
        rectify: 0,
        self_in_results: 0,
        reserved: 0,
    }
}

/// An item matrix resident in HBM: uploaded once (as_device_upload: pipelined pinned staging), freed on drop.
pub struct DeviceRows {
    pub ptr: *const f64,
    pub n: usize,
    pub d: usize,
}
impl DeviceRows {
    pub fn upload(rows: &[Vec<f64>]) -> Self {
        let (n, d) = (rows.len(), rows.first().map(|r| r.len()).unwrap_or(0));
        let flat: Vec<f64> = rows.iter().flat_map(|r| r.iter().copied()).collect();
        let ctx = thread_ctx();
        let mut dev: *mut c_void = std::ptr::null_mut();
        check(ctx, unsafe { hip::as_device_upload(ctx, flat.as_ptr() as *const c_void, (flat.len() * 8) as u64, &mut dev) });
        DeviceRows { ptr: dev as *const f64, n, d }
    }
}
impl Drop for DeviceRows {
    fn drop(&mut self) {
        unsafe { hip::as_device_free(thread_ctx(), self.ptr as *mut c_void) };
    }
}

/// CSR handle -> sprs::CsMat<f64> (GraphLaplacian.matrix stays a real CsMat for every other user of the crate).
pub(crate) fn csr_to_sprs(ctx: *mut AsCtx, h: *mut AsCsr) -> sprs::CsMat<f64> {
    let (n, cols, nnz) = unsafe { (hip::as_csr_n(h) as usize, hip::as_csr_cols(h) as usize, hip::as_csr_nnz(h) as usize) };
    let (mut indptr, mut indices, mut data) = (vec![0u64; n + 1], vec![0u64; nnz], vec![0f64; nnz]);
    check(ctx, unsafe { hip::as_csr_export(ctx, h, indptr.as_mut_ptr(), indices.as_mut_ptr(), data.as_mut_ptr()) });
    sprs::CsMat::new(
        (n, cols),
        indptr.into_iter().map(|x| x as usize).collect(),
        indices.into_iter().map(|x| x as usize).collect(),
        data,
    )
}

/// build_laplacian_matrix, src/laplacian.rs:122-178: rows of `nodes` are the graph's nodes (host or device pointer).
pub(crate) fn laplacian_of(ctx: *mut AsCtx, nodes: *const f64, n_nodes: usize, dim: usize, p: &crate::graph::GraphParams) -> *mut AsCsr {
    let c = graph_params_to_c(p);
    let mut h = std::ptr::null_mut();
    check(ctx, unsafe { hip::as_build_laplacian(ctx, nodes, n_nodes as u64, dim as u64, &c, &mut h) });
    h
}

/// ArrowSpaceBuilder::build, src/builder.rs:249-455: the same four stages on ONE device copy of the rows.
///   stage 1  clustering: optimal K + radius, incremental clustering (src/eigenmaps.rs:175-290)
///   stage 2  F x F Laplacian over centroids^T (GraphFactory::build_laplacian_matrix_from_k_cluster, src/graph.rs:149-204)
///   stage 4  lambdas of all items (TauMode::compute_taumode_lambdas_parallel, src/taumode.rs:174-312)
/// The device copy of the feature graph, freed with its owner (prepare_query_item reads it for every query).
pub struct DeviceGraph(pub *mut AsCsr);
impl Drop for DeviceGraph {
    fn drop(&mut self) {
        if !self.0.is_null() {
            unsafe { hip::as_csr_free(self.0) };
        }
    }
}

/// Returns what the reference returns, plus what the searches need on the device: the index handle (created here once), the
/// device copy of the rows it borrows, and the device copy of the feature graph (search_on_device takes it for the query's lambda).
pub fn build_on_device(
    builder: &mut crate::builder::ArrowSpaceBuilder,
    rows: Vec<Vec<f64>>,
) -> (crate::core::ArrowSpace, crate::graph::GraphLaplacian, *mut AsIndex, DeviceRows, DeviceGraph) {
    builder.define_result_k();
    let ctx = thread_ctx();
    let dev = DeviceRows::upload(&rows);
    let (n, f) = (dev.n, dev.d);

    // ---- stage 1: clustering pre-stage on the device copy ----
    let seed = builder.clustering_seed.unwrap_or(128);
    let (mut k_opt, mut radius, mut id_est) = (0u64, 0f64, 0u64);
    check(ctx, unsafe { hip::as_compute_optimal_k(ctx, dev.ptr, n as u64, f as u64, seed, &mut k_opt, &mut radius, &mut id_est) });
    let batch: u64 = if builder.deterministic_clustering { 1 } else { 4096 };
    let mut cents = vec![0f64; k_opt as usize * f];
    let (mut assign, mut counts, mut nc) = (vec![-1i64; n], vec![0u64; k_opt as usize], 0u64);
    check(ctx, unsafe {
        hip::as_incremental_clustering(ctx, dev.ptr, n as u64, f as u64, k_opt, radius, batch, cents.as_mut_ptr(), &mut nc,
                                       assign.as_mut_ptr(), counts.as_mut_ptr())
    });
    cents.truncate(nc as usize * f);

    // ---- stage 2: nodes of the graph are the FEATURES: rows of centroids^T (F x X) ----
    let x = nc as usize;
    let mut cent_t = vec![0f64; f * x];
    for c in 0..x {
        for j in 0..f {
            cent_t[j * x + c] = cents[c * f + j];
        }
    }
    let gp = crate::graph::GraphParams {
        eps: builder.lambda_eps, k: builder.lambda_k, topk: builder.lambda_topk, p: builder.lambda_p,
        sigma: builder.lambda_sigma, normalise: builder.normalise, sparsity_check: builder.sparsity_check,
    };
    let lh = laplacian_of(ctx, cent_t.as_ptr(), f, x, &gp);
    let matrix = csr_to_sprs(ctx, lh);
    if builder.sparsity_check {
        let sparsity = 1.0 - matrix.nnz() as f64 / (f * f) as f64;
        assert!(sparsity <= 0.95, "Resulting laplacian matrix is too sparse");  // src/graph.rs:185-193
    }

    // ---- stage 4: lambdas of all items, from the device copy; the index borrows the same copy ----
    let (mode, arg) = tau_to_c(builder.synthesis);
    let mut lambdas = vec![0f64; n];
    check(ctx, unsafe { hip::as_compute_taumode(ctx, dev.ptr, n as u64, f as u64, lh, mode, arg, lambdas.as_mut_ptr()) });
    let gl_dev = DeviceGraph(lh);   // kept: every prepared query reads it (search_on_device)
    let mut ix: *mut AsIndex = std::ptr::null_mut();
    check(ctx, unsafe { hip::as_index_create(ctx, dev.ptr, lambdas.as_ptr(), n as u64, f as u64, &mut ix) });

    let mut aspace = crate::core::ArrowSpace::new(rows, builder.synthesis);
    aspace.update_lambdas(lambdas);
    aspace.n_clusters = x;
    aspace.cluster_radius = radius;
    let gl = crate::graph::GraphLaplacian {
        init_data: smartcore::linalg::basic::matrix::DenseMatrix::from_2d_vec(
            &(0..f).map(|j| cent_t[j * x..(j + 1) * x].to_vec()).collect::<Vec<_>>()).unwrap(),
        matrix,
        nnodes: n,
        graph_params: gp,
    };
    (aspace, gl, ix, dev, gl_dev)
}

/// EigenMaps::search, src/eigenmaps.rs:410-455: query lambda, then the lambda-aware top-k on the kept index handle.
pub fn search_on_device(ix: *mut AsIndex, gl_dev: &DeviceGraph, taumode: crate::taumode::TauMode, item: &[f64], k: usize, alpha: f64)
    -> Vec<(usize, f64)> {
    let ctx = thread_ctx();
    let gl_dev = gl_dev.0 as *const AsCsr;
    let (mode, arg) = tau_to_c(taumode);
    let mut lq = 0f64;
    check(ctx, unsafe { hip::as_prepare_query_items(ctx, item.as_ptr(), 1, item.len() as u64, gl_dev, mode, arg, &mut lq) });
    let kk = k.min(unsafe { hip::as_index_nitems(ix) } as usize);
    let (mut idx, mut sc) = (vec![0u64; kk], vec![0f64; kk]);
    check(ctx, unsafe { hip::as_search_lambda_aware(ctx, ix, item.as_ptr(), &lq, 1, kk as u64, alpha, 0, idx.as_mut_ptr(), sc.as_mut_ptr()) });
    idx.into_iter().zip(sc).filter(|(i, _)| *i != u64::MAX).map(|(i, s)| (i as usize, s)).collect()
}

/// The item graph over several GPUs, one process per GPU: `rows_shard` = this rank's rows.
pub fn build_item_graph_sharded(id128: &[u8; 128], rank: i32, world: i32, rows_shard: &DeviceRows, n_total: usize,
                                p: &crate::graph::GraphParams) -> (sprs::CsMat<f64>, u64, u64) {
    let ctx = thread_ctx();
    let mut comm = std::ptr::null_mut();
    check(ctx, unsafe { hip::as_comm_create_rccl(ctx, id128.as_ptr() as *const c_void, rank, world, &mut comm) });
    let c = graph_params_to_c(p);
    let (mut block, mut nnz) = (std::ptr::null_mut(), 0u64);
    check(ctx, unsafe { hip::as_build_laplacian_sharded(ctx, comm, rows_shard.ptr, n_total as u64, rows_shard.d as u64, &c, &mut block, &mut nnz) });
    // a rank that owns no rows gets no block: an empty 0 x n_total matrix
    let (m, row_begin) = if block.is_null() {
        (sprs::CsMat::new((0, n_total), vec![0usize], Vec::new(), Vec::new()), 0u64)
    } else {
        let rb = unsafe { hip::as_csr_row_begin(block) };
        (csr_to_sprs(ctx, block), rb)   // rows [row_begin, row_begin + m.rows()) x n_total, global column ids
    };
    unsafe {
        if !block.is_null() {
            hip::as_csr_free(block);
        }
        hip::as_comm_destroy(comm);
    }
    (m, row_begin, nnz)
}
