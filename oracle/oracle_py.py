"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Importers allowed by the project rules: tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (gnn-mwvc_amd/) never
imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import pathlib
import subprocess

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
_LIB_PATH = _HERE / "liboracle.so"


class _Layer(C.Structure):
    _fields_ = [("kind", C.c_int), ("k", C.c_uint32), ("m", C.c_uint32),
                ("W", C.POINTER(C.c_float)), ("bias", C.POINTER(C.c_float))]


class _Model(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("n_layers", C.c_int),
                ("layers", C.POINTER(_Layer)), ("weight_scale", C.c_float)]


class _Graph(C.Structure):
    _fields_ = [("n", C.c_uint32), ("rowptr", C.c_void_p), ("col", C.c_void_p),
                ("w", C.c_void_p), ("nw", C.c_void_p)]


def build(force: bool = False) -> pathlib.Path:
    if force or not _LIB_PATH.exists() or \
            _LIB_PATH.stat().st_mtime < (_HERE / "gnnvc_oracle.c").stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "liboracle.so"], check=True,
                       capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        L = C.CDLL(str(_LIB_PATH))
        L.oracle_model_parse.restype = C.POINTER(_Model)
        L.oracle_model_parse.argtypes = [C.c_char_p, C.c_size_t]
        L.oracle_model_free.argtypes = [C.POINTER(_Model)]
        L.oracle_model_set_weight_scale.argtypes = [C.POINTER(_Model), C.c_float]
        L.oracle_predict.restype = C.c_int
        L.oracle_predict.argtypes = [C.POINTER(_Model), C.POINTER(_Graph), C.c_uint32,
                                     C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32),
                                     C.c_int, C.c_int]
        L.oracle_graph_layer.argtypes = [C.POINTER(_Graph), C.c_float, C.c_uint32,
                                         C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_linear_layer.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_relu.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p]
        L.oracle_sigmoid.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p]
        L.oracle_neighbourhood_weights.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_void_p]
        L.oracle_score_keys.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_score_keys.restype = None
        L.oracle_reduction_flags.argtypes = [C.POINTER(_Graph), C.c_uint32, C.c_void_p]
        L.oracle_set_cblas_sgemm.argtypes = [C.c_void_p]
        L.oracle_has_cblas_sgemm.restype = C.c_int
        L.oracle_sgemm.restype = None
        L.oracle_sgemm.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32,
                                   C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_uint32]
        L.oracle_num_threads.restype = C.c_int
        L.oracle_set_num_threads.argtypes = [C.c_int]
        # Size the OpenMP team to the CPUs this process may really use: a GPU box hands out a
        # 16-CPU share of a 128-thread host (a quota, not an affinity mask), and 128 workers on
        # 16 CPUs crawl.  Set through the API — libgomp may already be loaded (torch), in which
        # case OMP_NUM_THREADS is no longer read.
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        want = int(os.environ.get("GNNVC_ORACLE_THREADS", "0")) or max(1, min(avail, 16))
        L.oracle_set_num_threads(want)
        _lib = L
    return _lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _graph_struct(g):
    rowptr = np.ascontiguousarray(g.rowptr, dtype=np.uint64)
    col = np.ascontiguousarray(g.col, dtype=np.uint32)
    w = np.ascontiguousarray(g.w, dtype=np.uint32)
    nw = np.ascontiguousarray(g.nw, dtype=np.uint32)
    s = _Graph(g.n, _ptr(rowptr), _ptr(col), _ptr(w), _ptr(nw))
    return s, (rowptr, col, w, nw)


class OracleModel:
    """The reference model parsed by the oracle's own parser."""

    def __init__(self, text: str):
        raw = text.encode()
        self._m = lib().oracle_model_parse(raw, len(raw))
        if not self._m:
            raise ValueError("oracle: malformed model text")

    def __del__(self):
        if getattr(self, "_m", None):
            try:
                lib().oracle_model_free(self._m)
            except TypeError:      # interpreter shutdown: the module's globals are gone, and so is the process in a moment
                pass
            self._m = None

    @property
    def n_layers(self) -> int:
        return self._m.contents.n_layers

    def layer_kinds(self):
        return [self._m.contents.layers[i].kind for i in range(self.n_layers)]

    def linear_params(self):
        """[(W[k,m], bias[m])] for the linear layers, in order."""
        out = []
        for i in range(self.n_layers):
            l = self._m.contents.layers[i]
            if l.kind == 0:
                W = np.ctypeslib.as_array(l.W, shape=(l.k, l.m)).copy()
                b = np.ctypeslib.as_array(l.bias, shape=(l.m,)).copy()
                out.append((W, b))
        return out

    def set_weight_scale(self, ws: float):
        lib().oracle_model_set_weight_scale(self._m, C.c_float(ws))

    def predict(self, g, x: np.ndarray, stop_after: int = -1, parallel_agg: bool = False, as_shipped: bool = False):
        """Returns the [n, width] output after layer `stop_after` (-1 = all).  as_shipped: the reference's
        threading (products through the cblas_sgemm set with use_openblas(), everything else serial)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        x = x.reshape(g.n, x.size // g.n if g.n else 1)
        gs, keep = _graph_struct(g)
        out = np.empty((max(g.n, 1), 35 if x.shape[1] <= 16 else 2 * x.shape[1] + 3),
                       dtype=np.float32)
        wd = C.c_uint32(0)
        rc = lib().oracle_predict(self._m, C.byref(gs), x.shape[1], _ptr(x), _ptr(out),
                                  C.byref(wd), stop_after, (1 if parallel_agg else 0) | (2 if as_shipped else 0))
        if rc != 0:
            raise RuntimeError(f"oracle_predict failed: {rc}")
        del keep
        return out.reshape(-1)[: g.n * wd.value].reshape(g.n, wd.value).copy()

    def scores(self, g, x=None):
        return self.predict(g, g.x() if x is None else x)[:, 0]

    def logits(self, g, x=None):
        """Input of the final sigmoid (output of layer n_layers-2)."""
        return self.predict(g, g.x() if x is None else x, stop_after=self.n_layers - 2)[:, 0]


def graph_layer(g, ws: float, h: np.ndarray) -> np.ndarray:
    h = np.ascontiguousarray(h, dtype=np.float32).reshape(g.n, -1)
    f = h.shape[1]
    out = np.empty((g.n, 2 * f + 3), dtype=np.float32)
    gs, keep = _graph_struct(g)
    lib().oracle_graph_layer(C.byref(gs), C.c_float(ws), f, _ptr(h), _ptr(out), 0)
    del keep
    return out


def linear_layer(h: np.ndarray, W: np.ndarray, bias: np.ndarray) -> np.ndarray:
    h = np.ascontiguousarray(h, dtype=np.float32)
    W = np.ascontiguousarray(W, dtype=np.float32)
    bias = np.ascontiguousarray(bias, dtype=np.float32)
    out = np.empty((h.shape[0], W.shape[1]), dtype=np.float32)
    lib().oracle_linear_layer(h.shape[0], W.shape[0], W.shape[1], _ptr(h), _ptr(W), _ptr(bias),
                              _ptr(out))
    return out


def relu(h: np.ndarray) -> np.ndarray:
    h = np.ascontiguousarray(h, dtype=np.float32)
    out = np.empty_like(h)
    lib().oracle_relu(h.size, _ptr(h), _ptr(out))
    return out


def sigmoid(h: np.ndarray) -> np.ndarray:
    h = np.ascontiguousarray(h, dtype=np.float32)
    out = np.empty_like(h)
    lib().oracle_sigmoid(h.size, _ptr(h), _ptr(out))
    return out


def score_keys(scores: np.ndarray):
    """(min(s, 1 - s), s > 0.5) per score — what the driver's sort and selection loop read."""
    s = np.ascontiguousarray(scores, dtype=np.float32).reshape(-1)
    keys = np.empty_like(s)
    above = np.zeros(s.size, dtype=np.uint8)
    lib().oracle_score_keys(s.size, _ptr(s), _ptr(keys), _ptr(above))
    return keys, above


def reduction_flags(g, max_degree: int = 20) -> np.ndarray:
    """Reduction-rule predicate bits per vertex (see gnnvc_oracle.h)."""
    flags = np.zeros(g.n, dtype=np.uint8)
    gs, keep = _graph_struct(g)
    lib().oracle_reduction_flags(C.byref(gs), max_degree, _ptr(flags))
    del keep
    return flags


def sgemm(A, B, C_in=None, beta: float = 0.0, trans_a: bool = False, trans_b: bool = False) -> np.ndarray:
    """dot() with transposes and beta as include/gnnvc.h documents gnnvc_sgemm (oracle_sgemm)."""
    A = np.ascontiguousarray(A, dtype=np.float32)
    B = np.ascontiguousarray(B, dtype=np.float32)
    m = A.shape[1] if trans_a else A.shape[0]
    k = A.shape[0] if trans_a else A.shape[1]
    n = B.shape[0] if trans_b else B.shape[1]
    out = np.zeros((m, n), dtype=np.float32) if C_in is None else np.ascontiguousarray(C_in, dtype=np.float32).copy()
    lib().oracle_sgemm(int(trans_a), int(trans_b), m, n, k, _ptr(A), A.shape[1], _ptr(B), B.shape[1], C.c_float(beta),
                       _ptr(out), n)
    return out


_openblas = None


def find_openblas():
    """(path, symbol) of an LP64 OpenBLAS cblas_sgemm on this machine: a system libopenblas, or the one SciPy
    bundles (`scipy_cblas_sgemm`); None if neither is present."""
    import ctypes.util
    import glob
    cands = []
    sysname = ctypes.util.find_library("openblas")
    if sysname:
        cands.append((sysname, "cblas_sgemm"))
    try:
        import scipy
        base = os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs")
        cands += [(p, "scipy_cblas_sgemm") for p in sorted(glob.glob(os.path.join(base, "libscipy_openblas-*.so")))]
    except Exception:
        pass
    for path, sym in cands:
        try:
            L = C.CDLL(path)
            if hasattr(L, sym):
                return path, sym
        except OSError:
            continue
    return None


def use_openblas(threads: int = 0):
    """Route predict(as_shipped=True)'s products through a discovered OpenBLAS (what the reference links,
    src/matrix.cpp:5,112-121).  Returns a description string, or None when no OpenBLAS was found (the internal
    fmaf loops stay in place)."""
    global _openblas
    hit = find_openblas()
    if hit is None:
        lib().oracle_set_cblas_sgemm(None)
        return None
    path, sym = hit
    L = C.CDLL(path)
    fn = getattr(L, sym)
    for name in ("scipy_openblas_set_num_threads", "openblas_set_num_threads"):
        if threads and hasattr(L, name):
            getattr(L, name)(threads)
            break
    ver = ""
    for name in ("scipy_openblas_get_config", "openblas_get_config"):
        if hasattr(L, name):
            f = getattr(L, name)
            f.restype = C.c_char_p
            ver = (f() or b"").decode(errors="replace")
            break
    _openblas = (L, fn)   # keep the library alive
    lib().oracle_set_cblas_sgemm(C.cast(fn, C.c_void_p))
    return f"{os.path.basename(path)}:{sym} [{ver.strip()}]"


def num_threads() -> int:
    return lib().oracle_num_threads()


def set_num_threads(n: int) -> None:
    lib().oracle_set_num_threads(n)
