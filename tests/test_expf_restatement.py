"""The device sigmoid uses a restatement of glibc's expf (csrc/expf_glibc.h).
Compiled for the host here and compared with the host libm: bit-identical on
x86-64 CPUs with FMA (glibc's `__expf_fma` variant); on a CPU without FMA glibc
runs the unfused variant and a last-ulp difference in a few results is allowed."""
import ctypes as C
import pathlib
import subprocess

import numpy as np
import pytest

HERE = pathlib.Path(__file__).resolve().parent
SRC = HERE / "support" / "expf_shim.cpp"
SO = HERE / "support" / "libexpf_shim.so"


@pytest.fixture(scope="module")
def shim():
    hdr = HERE.parent / "gnn-mwvc_amd" / "csrc" / "expf_glibc.h"
    if not SO.exists() or SO.stat().st_mtime < max(SRC.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.run(["g++", "-O2", "-mavx2", "-mfma", "-ffp-contract=off", "-fno-builtin-expf",
                        "-std=c++17", "-fPIC", "-shared", "-o", str(SO), str(SRC), "-lm"], check=True)
    L = C.CDLL(str(SO))
    for f in ("expf_restated", "expf_libm", "sigmoid_restated", "sigmoid_libm"):
        getattr(L, f).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    return L


def _run(fn, x):
    out = np.empty_like(x)
    fn(x.ctypes.data, out.ctypes.data, x.size)
    return out


def _has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except OSError:
        return False


def test_expf_matches_libm(shim):
    rng = np.random.default_rng(0)
    parts = [rng.uniform(-20, 20, 8_000_000), rng.uniform(-104, 89, 4_000_000),
             rng.normal(0, 1e-3, 1_000_000), rng.uniform(-1, 1, 3_000_000)]
    x = np.concatenate(parts).astype(np.float32)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 88.0, 88.72283, 88.72284, 89.0, 100.0,
                        -87.0, -88.0, -103.0, -103.9, -104.0, -150.0, 1e-30, -1e-30, 1e-45],
                       dtype=np.float32)
    x = np.concatenate([x, special])
    got, want = _run(shim.expf_restated, x), _run(shim.expf_libm, x)
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    if _has_fma():
        assert same.all(), f"{(~same).sum()} of {x.size} differ, e.g. x={x[~same][:5]}"
    else:
        d = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
        assert d[~np.isnan(want)].max() <= 1


def test_sigmoid_matches_libm(shim):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-12, 12, 6_000_000), rng.normal(0, 2, 4_000_000),
                        np.array([0, -0.0, 30, -30, 90, -90, 104, -104, 1e6, -1e6])]).astype(np.float32)
    got, want = _run(shim.sigmoid_restated, x), _run(shim.sigmoid_libm, x)
    if _has_fma():
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    else:
        d = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
        assert d.max() <= 1
