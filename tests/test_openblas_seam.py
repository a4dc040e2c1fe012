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


def test_beta_nonzero_has_no_single_openblas_behaviour(sgemm):
    """dot()'s `beta` (reference src/matrix.cpp:106-122) is never non-zero on the inference path
    (src/gnn_inference.cpp:21 passes 0).  Recorded here because include/gnnvc.h documents gnnvc_sgemm's
    beta != 0 result as fma(beta, C_old, chain): OpenBLAS 0.3.28 itself rounds beta * C in TWO ways depending on the
    problem size — its small-matrix kernels fuse it into the final add (the form the engine and oracle_sgemm use),
    its blocked path scales C first and then adds — so neither form is "the" third-party behaviour."""
    fn, set_threads = sgemm
    if set_threads is not None:
        set_threads(1)
    rng = np.random.default_rng(11)
    seen = set()
    for (m, n, k) in [(1000, 32, 35), (37, 16, 32), (2000, 16, 32), (24, 24, 24)]:
        A = rng.uniform(-2, 2, (m, k)).astype(np.float32)
        B = rng.uniform(-2, 2, (k, n)).astype(np.float32)
        C0 = rng.uniform(-2, 2, (m, n)).astype(np.float32)
        for beta in (-1.7, 0.3):
            c = C0.copy()
            fn(101, 111, 111, m, n, k, 1.0, A.ctypes.data, k, B.ctypes.data, n, beta, c.ctypes.data, n)
            chain = oracle_py.sgemm(A, B)                                   # beta = 0: the sequential-k chain
            fused = oracle_py.sgemm(A, B, C0, beta)                         # fma(beta, C, chain)
            scaled = ((np.float32(beta) * C0).astype(np.float32) + chain).astype(np.float32)   # RN(RN(beta C) + chain)
            is_fused = np.array_equal(c.view(np.uint32), fused.view(np.uint32))
            is_scaled = np.array_equal(c.view(np.uint32), scaled.view(np.uint32))
            assert is_fused or is_scaled, (m, n, k, beta)
            seen.add("fused" if is_fused else "scaled")
        # beta = 1 and 0.5 (exact products) agree in both forms and with OpenBLAS
        c = C0.copy()
        fn(101, 111, 111, m, n, k, 1.0, A.ctypes.data, k, B.ctypes.data, n, 0.5, c.ctypes.data, n)
        assert np.array_equal(c.view(np.uint32), oracle_py.sgemm(A, B, C0, 0.5).view(np.uint32))
    assert seen == {"fused", "scaled"}, seen
