"""Runs the C++ host-mirror test program (tests/cpp/test_host_mirror.cpp): the reference's own
tests for the path, restated against arrowspace-rs_amd/host/arrowspace.hpp over the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "arrowspace-rs_amd", "build", "test_host_mirror")


def test_cpp_mirror_is_built():
    assert os.path.exists(EXE), "run __graft_entry__.build()"


@pytest.mark.gpu
def test_cpp_mirror_passes_on_gpu():
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 failures" in r.stdout
