"""CPU-only checks of the C-ABI shared library: it builds for gfx950, loads, exports every symbol
include/gdmix_re.h declares, and its host-only entry points work. No compute call is made."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from gdmix_amd import build, solver

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build_library()
    return solver.load_library()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "gdmix_re.h")).read() + open(os.path.join(ROOT, "include", "gdmix_fe.h")).read()
    declared = set(re.findall(r"GDMIX_API[^;(]*?\b(gdmix_\w+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(solver.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_default_opts_and_struct_layout(lib):
    o = solver._Opts()
    lib.gdmix_re_default_opts(C.byref(o))
    d = solver.SolverOptions()
    assert (o.l2, o.regularize_bias, o.has_intercept, o.m, o.max_iter, o.maxfun, o.maxls) == (1.0, 1, 1, 10, 100, 15000, 20)
    assert (o.ftol, o.pgtol, o.variance_mode, o.threshold) == (1e-12, 1e-5, 0, 1e-4)
    c = d.to_c()
    for f, _ in solver._Opts._fields_:
        assert getattr(c, f) == getattr(o, f), f


def test_workspace_size_is_monotone(lib):
    a = lib.gdmix_re_pack_workspace_bytes(10, 100, 1000)
    b = lib.gdmix_re_pack_workspace_bytes(20, 200, 2000)
    assert 0 < a < b


def test_create_without_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    h = C.c_void_p()
    rc = lib.gdmix_re_create(0, C.byref(h))
    assert rc != 0 and not h.value
    assert lib.gdmix_re_last_error()
    with pytest.raises(solver.GdmixReError):
        solver.REDeviceSolver(0)


from test_oracle_golden import JAVA_HASH_KAT, JAVA_PART_KAT  # noqa: E402


@pytest.mark.parametrize("s,h", JAVA_HASH_KAT)
def test_java_hash_host_entry_point(lib, s, h):
    assert solver.java_string_hash(s) == h


@pytest.mark.parametrize("s,n,pid", JAVA_PART_KAT)
def test_java_partition_host_entry_point(lib, s, n, pid):
    assert solver.java_partition_id(s, n) == pid
