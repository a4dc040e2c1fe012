"""The one third-party arithmetic seam on the path: cblas_sgemm.

The reference's dot() (reference src/matrix.cpp:106-122) hands every linear
layer to OpenBLAS `cblas_sgemm(RowMajor, NoTrans, NoTrans, m, n, k, 1, A, k, B,
n, 0, C, n)`.  OpenBLAS is un-vendored and unpinned (README.md:24-27,
Makefile:32) and the image has no system copy — but SciPy bundles a genuine
LP64 OpenBLAS 0.3.28 exporting `scipy_cblas_sgemm`.  This test calls THAT
library directly (ctypes, no header, no reference build) and checks that the
oracle's restatement — one sequential-k fmaf chain per output from +0.0f, bias
added afterwards — is bit-identical at all nine shapes of the shipped model,
with 1 and with all BLAS threads.  Skipped when the library is absent.
"""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from oracle import oracle_py

SHAPES = [(5, 32), (32, 32), (32, 16), (35, 32), (32, 32), (32, 16), (35, 32), (32, 16), (16, 1)]


def _find_openblas():
    try:
        import scipy
    except Exception:
        return None
    base = os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs")
    hits = sorted(glob.glob(os.path.join(base, "libscipy_openblas-*.so")))
    return hits[0] if hits else None


@pytest.fixture(scope="module")
def sgemm():
    path = _find_openblas()
    if not path:
        pytest.skip("no bundled LP64 OpenBLAS in this environment")
    lib = C.CDLL(path)
    fn = lib.scipy_cblas_sgemm
    fn.restype = None
    fn.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                   C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int]
    set_threads = getattr(lib, "scipy_openblas_set_num_threads", None)
    return fn, set_threads


@pytest.mark.parametrize("threads", [1, 0])
def test_sgemm_is_sequential_fma_chain(sgemm, oracle_model, threads):
    fn, set_threads = sgemm
    if set_threads is not None:
        set_threads(threads if threads else (os.cpu_count() or 1))
    rng = np.random.default_rng(7)
    params = oracle_model.linear_params()
    rows = 10007
    total = 0
    for (W, b), (k, n) in zip(params, SHAPES):
        assert W.shape == (k, n)
        a = (rng.uniform(-3, 3, size=(rows, k))).astype(np.float32)
        # some exact zeros / negative zeros like ReLU outputs and the unused columns
        a[rng.random(a.shape) < 0.2] = 0.0
        c = np.full((rows, n), np.nan, dtype=np.float32)
        CblasRowMajor, CblasNoTrans = 101, 111
        fn(CblasRowMajor, CblasNoTrans, CblasNoTrans, rows, n, k, 1.0,
           a.ctypes.data, k, W.ctypes.data, n, 0.0, c.ctypes.data, n)
        got = (c + b[None, :]).astype(np.float32)  # the reference's separate bias add
        want = oracle_py.linear_layer(a, W, b)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (k, n)
        total += got.size
    assert total > 1_000_000
