"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol the
header declares, and refuses to run without a gfx950 device (no CPU fallback)."""
import os
import re

import pytest

import arrowspace_amd as A
from arrowspace_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "arrowspace_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(as_[a-z0-9_]+)\s*\(", txt)))


def test_library_is_built():
    assert os.path.exists(A.LIB_PATH), "run __graft_entry__.build() first"


def test_every_declared_symbol_is_exported():
    lib = A.load()
    syms = _header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "header declares %s but the library does not export it" % s
    assert sorted(_lib.EXPORTS) == syms


def test_abi_version_and_messages():
    lib = A.load()
    assert lib.as_abi_version() == 5
    # the reference's panic texts travel through the boundary verbatim
    assert lib.as_status_message(_lib.AS_ERR_ZERO_LAMBDA).decode() == \
        "Lambda of the item is 0.0, prepare the item before searching"  # src/core.rs:773-776
    assert lib.as_status_message(_lib.AS_ERR_NONFINITE).decode().startswith(
        "Query item contains invalid values (NaN or infinity)")  # src/core.rs:534-537
    assert "too sparse" in lib.as_status_message(_lib.AS_ERR_TOO_SPARSE).decode()  # src/graph.rs:185-193
    assert "shape (2,2)" in lib.as_status_message(_lib.AS_ERR_SHAPE).decode()  # src/laplacian.rs:129-134
    assert "another rank" in lib.as_status_message(_lib.AS_ERR_PEER).decode()  # the sharded build's status agreement
    # the Python constants mirror the header's enum, one by one
    import re
    hdr = open(os.path.join(ROOT, "include", "arrowspace_hip.h")).read()
    for name, value in re.findall(r"^\s*(AS_(?:OK|ERR_[A-Z_]+)) = (\d+)", hdr, flags=re.M):
        assert getattr(_lib, name) == int(value), name


def test_define_result_k_needs_no_device():
    lib = A.load()
    assert [lib.as_define_result_k(k, 9) for k in (1, 5, 6, 9, 10, 16)] == [3, 3, 4, 4, 9, 9]  # builder.rs:225-233


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_have_gpu(), reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback():
    with pytest.raises(A.ArrowSpaceError) as e:
        A.Context(0)
    assert e.value.status == _lib.AS_ERR_NO_DEVICE


def test_host_mirror_names():
    # same public names as the reference's API for this path
    for name in ("ArrowSpaceBuilder", "ArrowSpace", "ArrowItem", "GraphLaplacian", "GraphParams", "GraphFactory",
                 "TauMode", "build_laplacian_matrix"):
        assert hasattr(A, name)
    b = A.ArrowSpaceBuilder.new().with_lambda_graph(0.5, 16, 16, 2.0, 0.25).with_synthesis(A.TauMode.Median)
    assert (b.lambda_eps, b.lambda_k, b.lambda_topk, b.lambda_p, b.lambda_sigma) == (0.5, 16, 16, 2.0, 0.25)
    d = A.ArrowSpaceBuilder.new()  # defaults: src/builder.rs:59-91
    assert (d.lambda_eps, d.lambda_k, d.lambda_topk, d.lambda_p, d.lambda_sigma) == (1e-3, 6, 3, 2.0, None)
    d.define_result_k()
    assert d.lambda_topk == 4


def test_rust_extern_block_lists_every_export():
    """the `extern "C"` block a maintainer of the reference crate adds
    (arrowspace-rs_amd/host/rust/src/hip.rs, generated from the header, not compiled here: no rustc in the image)
    declares exactly the header's exports, and the committed file is what the generator produces."""
    import subprocess
    import sys
    rs = open(os.path.join(ROOT, "arrowspace-rs_amd", "host", "rust", "src", "hip.rs")).read()
    declared = sorted(set(re.findall(r"pub fn (as_[a-z0-9_]+)\(", rs)))
    assert declared == _header_symbols()
    assert subprocess.call([sys.executable, os.path.join(ROOT, "scripts", "gen_rust_bindings.py"), "--check"]) == 0
    # the parameter struct has the header's fields in the header's order
    hdr = open(os.path.join(ROOT, "include", "arrowspace_hip.h")).read()
    body = re.search(r"typedef struct \{(.*?)\} as_graph_params;", hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    c_fields = re.findall(r"\b(\w+);", body)
    rs_fields = re.findall(r"pub (\w+):", re.search(r"pub struct AsGraphParams \{(.*?)\n\}", rs, flags=re.S).group(1))
    assert c_fields == rs_fields == [f[0] for f in _lib.GraphParamsC._fields_]


def _header_abi_version():
    hdr = open(os.path.join(ROOT, "include", "arrowspace_hip.h")).read()
    return int(re.search(r"#define\s+AS_ABI_VERSION\s+(\d+)", hdr).group(1))


def test_rust_shim_and_integration_doc_follow_the_header():
    """The hand-written Rust side (host/rust/src/shim.rs, uncompiled: no rustc in the image) and INTEGRATION.md must not drift
    from the header: the expected ABI version comes from ONE generated constant (hip.rs: AS_ABI_VERSION), every `hip::as_*` the
    shim calls and every `as_*(` the document names is an export of the header, and the counts the document quotes are the
    header's.  (Round 4 shipped a shim that asserted version 4 against a version-5 library: src/eigenmaps.rs:93-172 seam.)"""
    want = _header_abi_version()
    syms = set(_header_symbols())
    rsdir = os.path.join(ROOT, "arrowspace-rs_amd", "host", "rust", "src")
    hip_rs = open(os.path.join(rsdir, "hip.rs")).read()
    assert int(re.search(r"pub const AS_ABI_VERSION: c_int = (\d+);", hip_rs).group(1)) == want
    shim = open(os.path.join(rsdir, "shim.rs")).read()
    code = "\n".join(line.split("//")[0] for line in shim.splitlines())
    # the version check reads the generated constant, never a literal
    m = re.search(r"assert_eq!\(\s*unsafe \{ hip::as_abi_version\(\) \}\s*,\s*([^,]+),", code)
    assert m and m.group(1).strip() == "hip::AS_ABI_VERSION", m and m.group(1)
    used = set(re.findall(r"\bhip::(as_[a-z0-9_]+)\s*\(", code))
    assert used and used <= syms, sorted(used - syms)
    # every hip:: item that is not a function is a type or constant hip.rs defines
    for item in set(re.findall(r"\bhip::([A-Z][A-Za-z0-9_]*)", code)):
        assert re.search(r"pub (?:struct|const|type) %s\b" % item, hip_rs), item
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    named = set(re.findall(r"\b(as_[a-z0-9_]+)\s*\(", doc)) | set(re.findall(r"`(as_[a-z0-9_]+)`", doc))
    named = {s_ for s_ in named if not s_.endswith("_")}   # `as_jl_*`-style families are prefixes, checked below
    named -= {"as_ptr", "as_mut_ptr"}   # Rust slice methods in the excerpts, not exports
    assert named <= syms | {"as_graph_params", "as_ctx", "as_csr", "as_index", "as_comm", "as_comm_ops", "as_jl", "as_clustering"}, \
        sorted(named - syms)
    for prefix in re.findall(r"`(as_[a-z0-9_]+_)\*`", doc):
        assert any(s_.startswith(prefix) for s_ in syms), prefix
    for v in re.findall(r"ABI version (\d+)", doc) + re.findall(r"AS_ABI_VERSION: c_int = (\d+)", doc) + \
            re.findall(r"as_abi_version\(\)[^\n]*?==\s*(\d+)", doc):
        assert int(v) == want, "INTEGRATION.md names ABI %s, the header %d" % (v, want)
    for cnt in re.findall(r"\((\d+), ABI version", doc):
        assert int(cnt) == len(syms)
    m = re.search(r"// \.\.\. (\d+) more: see host/rust/src/hip.rs", doc)
    shown = len(set(re.findall(r"pub fn (as_[a-z0-9_]+)\(", doc.split("// ...")[0].split("```rust")[-1])))
    assert m and int(m.group(1)) + shown == len(syms), (m and m.group(1), shown, len(syms))


def test_every_option_the_library_accepts_is_documented_in_the_header():
    """as_ctx_set_option's names (csrc/api.hip) and the option block of include/arrowspace_hip.h are the same set: round 5's review
    found A/B switches that existed only in the source."""
    with open(os.path.join(ROOT, "include", "arrowspace_hip.h")) as f:
        hdr = f.read()
    end = hdr.index("int as_ctx_set_option")
    block = hdr[hdr.rindex("/*", 0, end):end]
    documented = set(re.findall(r'"([a-z0-9_]+)"', block))
    with open(os.path.join(ROOT, "arrowspace-rs_amd", "csrc", "api.hip")) as f:
        api = f.read()
    start = api.index("int as_ctx_set_option")
    body = api[start:api.index("\n}\n", start)]
    accepted = set(re.findall(r'n == "([a-z0-9_]+)"', body))
    assert len(accepted) > 60
    assert accepted - documented == set(), sorted(accepted - documented)
    assert documented - accepted == set(), sorted(documented - accepted)
