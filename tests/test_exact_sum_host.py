"""The parity-map evaluation of the sequential fp32 neighbour sum (csrc/exact_sum.h, used by the giant-row
kernels) against the plain chain `acc = acc + v[i]` of the reference (src/gnn_inference.cpp:33-36), on the
host: tests/support/exact_sum_host.cpp runs the kernel's window / lane / scan control flow with the 64
lanes as a loop.  Bit-identical results are required on every stream, whatever it holds."""
import ctypes as C
import pathlib
import subprocess

import numpy as np
import pytest

HERE = pathlib.Path(__file__).resolve().parent
SRC = HERE / "support" / "exact_sum_host.cpp"
HDR = HERE.parent / "gnn-mwvc_amd" / "csrc" / "exact_sum.h"
SO = HERE / "support" / "libexact_sum_host.so"


@pytest.fixture(scope="module")
def xs():
    if not SO.exists() or SO.stat().st_mtime < max(SRC.stat().st_mtime, HDR.stat().st_mtime):
        subprocess.run(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-o", str(SO), str(SRC)],
                       check=True)
    L = C.CDLL(str(SO))
    L.xsum_stream.restype = C.c_float
    L.xsum_stream.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    L.xsum_sequential.restype = C.c_float
    L.xsum_sequential.argtypes = [C.c_void_p, C.c_size_t]
    L.xsum_stream_segmented.restype = C.c_float
    L.xsum_stream_segmented.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p]
    return L


def _check(L, v, B=16, want_fast=None):
    _check1(L, v, B, want_fast)
    _check1(L, v, -B, want_fast)     # the floating-point decode of the addends
    # the segmented evaluation (one stream on several waves): whatever binade a segment's map was built for — the estimate
    # the kernels use, one above, one below, a pseudo-random one — the result is the chain's
    v = np.ascontiguousarray(v, dtype=np.float32)
    want = np.float32(L.xsum_sequential(v.ctypes.data, v.size))
    for seg_windows, guess in ((4, 0), (4, 1), (4, 2), (4, 4), (1, 0), (7, 0)):
        st = np.zeros(3, dtype=np.uint64)
        got = np.float32(L.xsum_stream_segmented(v.ctypes.data, v.size, -B, seg_windows, guess, st.ctypes.data))
        assert got.view(np.uint32) == want.view(np.uint32) or (np.isnan(got) and np.isnan(want)), \
            f"segmented ({seg_windows} windows, guess {guess}): {got!r} != {want!r}, n={v.size}, stats={st}"


def _check1(L, v, B, want_fast):
    v = np.ascontiguousarray(v, dtype=np.float32)
    st = np.zeros(3, dtype=np.uint64)
    got = np.float32(L.xsum_stream(v.ctypes.data, v.size, B, st.ctypes.data))
    want = np.float32(L.xsum_sequential(v.ctypes.data, v.size))
    gb, wb = got.view(np.uint32), want.view(np.uint32)
    assert gb == wb or (np.isnan(got) and np.isnan(want)), f"{got!r} ({gb:#x}) != {want!r} ({wb:#x}), n={v.size}, stats={st}"
    if want_fast is not None:   # the integer route did the bulk of the work
        windows = (v.size + 64 * abs(B) - 1) // (64 * abs(B))
        assert st[2] == 0 and st[0] <= windows + want_fast, f"steps {st[0]} for {windows} windows (carries {st[1]})"
    return st


def test_numpy_cumsum_is_the_chain(xs):
    # sanity of the yardstick itself: np.cumsum on float32 is the same sequential chain
    rng = np.random.default_rng(0)
    v = rng.uniform(0, 3, 50_000).astype(np.float32)
    want = np.float32(xs.xsum_sequential(v.ctypes.data, v.size))
    assert np.cumsum(v, dtype=np.float32)[-1].view(np.uint32) == want.view(np.uint32)


@pytest.mark.parametrize("B", [4, 16])
def test_feature_like_streams(xs, B):
    rng = np.random.default_rng(1)
    for n in (1, 2, 63, 64, 65, 1023, 1024, 1025, 5000, 70_000, 700_000):
        # post-ReLU activations: many exact zeros, the rest spread over a few binades
        v = rng.gamma(2.0, 0.7, n).astype(np.float32)
        v[rng.random(n) < 0.4] = 0.0
        _check(xs, v, B, want_fast=64)
    # weights / scale (stage 0 of the driver): W in [20, 120], ws = 120
    w = rng.integers(20, 121, 650_000).astype(np.float32) / np.float32(120.0)
    _check(xs, w, B, want_fast=64)


def test_ties_and_parity(xs):
    rng = np.random.default_rng(2)
    # values that are exact multiples of half an ulp of the running sum: every add is a tie or exact
    for scale in (2.0 ** -3, 1.0, 2.0 ** 7):
        v = (rng.integers(0, 64, 40_000) * np.float32(scale / 16)).astype(np.float32)
        v[0] = np.float32(scale * 4096)   # the accumulator starts far above the addends
        _check(xs, v)
    # constant addend: acc walks through every binade, ties all the way once ulp(acc) = 2 v
    _check(xs, np.full(300_000, 1.0, dtype=np.float32))
    _check(xs, np.full(300_000, 3.0, dtype=np.float32))
    _check(xs, np.full(40_000_000 // 64, 0.1, dtype=np.float32))
    # alternating pattern that flips the parity at every step
    v = np.tile(np.array([1.5, 0.5, 2.5, 1.0], dtype=np.float32) * np.float32(2.0 ** -10), 50_000)
    v[0] = 8191.0
    _check(xs, v)


def test_wide_exponent_range_and_denormals(xs):
    rng = np.random.default_rng(3)
    n = 200_000
    v = (rng.random(n).astype(np.float32) * np.exp2(rng.integers(-149, 20, n)).astype(np.float32)).astype(np.float32)
    _check(xs, v)
    # denormals only, then across the denormal / normal boundary
    d = (rng.integers(0, 1 << 20, 100_000).astype(np.uint32)).view(np.float32)
    _check(xs, d)
    d2 = (rng.integers(0, 1 << 24, 100_000).astype(np.uint32)).view(np.float32)
    _check(xs, d2)
    # large jumps: a huge value in the middle of small ones, and growth up to overflow
    v = rng.uniform(0, 1, 100_000).astype(np.float32)
    v[50_000] = 3.0e30
    _check(xs, v)
    big = np.full(10_000, 3.0e38, dtype=np.float32)
    _check(xs, big)                     # overflows to +inf; the rest is added the plain way
    _check(xs, np.concatenate([v, big, v]))


def test_negative_nan_and_signed_zero(xs):
    rng = np.random.default_rng(4)
    v = rng.uniform(0, 2, 100_000).astype(np.float32)
    v[::997] = -0.0
    _check(xs, v)
    v2 = v.copy()
    v2[1234] = -5.0                    # one negative value: its piece goes the plain way
    v2[77_777] = -1e-3
    _check(xs, v2)
    _check(xs, rng.normal(0, 1, 50_000).astype(np.float32))          # half of the values negative
    v3 = v.copy()
    v3[40_000] = np.nan
    _check(xs, v3)
    v4 = v.copy()
    v4[10] = np.inf
    v4[20_000] = -np.inf
    _check(xs, v4)
    _check(xs, np.zeros(5000, dtype=np.float32))
    _check(xs, np.full(5000, -0.0, dtype=np.float32))
    _check(xs, np.array([], dtype=np.float32))


def test_random_bit_patterns(xs):
    rng = np.random.default_rng(5)
    for _ in range(20):
        n = int(rng.integers(1, 20_000))
        bits = rng.integers(0, 1 << 31, n).astype(np.uint32)          # non-negative floats of any exponent
        bits[(bits >> 23) == 255] &= 0x7F000000                       # keep them finite
        _check(xs, bits.view(np.float32), B=int(rng.choice([4, 8, 16])))


def test_segments_are_taken_in_one_step(xs):
    """On streams like the kernels see (non-negative activations, 200 K addends) nearly every segment after the first few is
    applied as ONE map: the binade estimate from the float sums is right and the segment does not carry."""
    rng = np.random.default_rng(7)
    n = 200_000
    v = rng.gamma(2.0, 0.7, n).astype(np.float32)
    v[rng.random(n) < 0.4] = 0.0
    st = np.zeros(3, dtype=np.uint64)
    got = np.float32(xs.xsum_stream_segmented(v.ctypes.data, v.size, -16, 4, 0, st.ctypes.data))
    assert got.view(np.uint32) == np.float32(xs.xsum_sequential(v.ctypes.data, v.size)).view(np.uint32)
    nseg = (n + 4095) // 4096
    assert st[1] >= nseg - 8, (st, nseg)       # all but a handful (the first, and those where the sum crosses a binade)
    assert st[0] <= 8 * 4 + 16                 # ... so the window walk did little
