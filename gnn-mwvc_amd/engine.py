"""ctypes binding of libgnnvc_hip.so — the C ABI declared in include/gnnvc.h.

Python here is plumbing for tests and the bench; the product is the shared
library.  There is no CPU fallback: if the library is missing or no HIP device
is present, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
import pathlib
import subprocess

import numpy as np

_PKG = pathlib.Path(__file__).resolve().parent
# GNNVC_LIBRARY: another build of the same library (A/B experiments with compile-time variants); default: the in-tree one
_LIB = pathlib.Path(os.environ["GNNVC_LIBRARY"]).resolve() if os.environ.get("GNNVC_LIBRARY") else _PKG / "libgnnvc_hip.so"

# every symbol include/gnnvc.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "gnnvc_abi_version", "gnnvc_strerror", "gnnvc_last_error", "gnnvc_create", "gnnvc_create_multi", "gnnvc_destroy",
    "gnnvc_set_weight_scale", "gnnvc_set_stream", "gnnvc_set_option", "gnnvc_get_info", "gnnvc_num_layers", "gnnvc_is_fused",
    "gnnvc_in_width", "gnnvc_out_width", "gnnvc_upload_graph", "gnnvc_attach_graph_device", "gnnvc_attach_graph_slice",
    "gnnvc_graph_staging", "gnnvc_staged_columns_ready", "gnnvc_commit_staged_graph",
    "gnnvc_derive_graph_begin", "gnnvc_derive_graph_commit", "gnnvc_graph_row_hashes",
    "gnnvc_forward", "gnnvc_forward_device", "gnnvc_num_stages", "gnnvc_stage_widths",
    "gnnvc_stage_forward_device", "gnnvc_stage_input_ready", "gnnvc_live_columns", "gnnvc_column_counts", "gnnvc_pack_rows", "gnnvc_unpack_rows", "gnnvc_unpack_gathered",
    "gnnvc_push_piece", "gnnvc_unpack_pieces", "gnnvc_get_stream",
    "gnnvc_reduction_flags", "gnnvc_score_keys", "gnnvc_synchronize", "gnnvc_last_forward_ms",
    "gnnvc_graph_layer_forward", "gnnvc_linear_forward", "gnnvc_relu_forward",
    "gnnvc_sigmoid_forward", "gnnvc_sgemm", "gnnvc_stream_sum", "gnnvc_kernel_trace",
]
COL_PAD = 64


class GnnvcError(RuntimeError):
    def __init__(self, code: int, detail: str = ""):
        self.code = code
        super().__init__(f"gnnvc error {code}: {detail}")


def library_path() -> pathlib.Path:
    return _LIB


def build_library(force: bool = False) -> pathlib.Path:
    """hipcc --offload-arch=gfx950 build of csrc/ (cross-compiles without a GPU)."""
    if os.environ.get("GNNVC_LIBRARY"):
        # an external build was asked for: it is the caller's to keep current (make only knows the in-tree library)
        if not _LIB.exists():
            raise FileNotFoundError(f"GNNVC_LIBRARY={_LIB} does not exist")
        return _LIB
    # make decides what is stale (csrc/Makefile lists every source and header the library depends on)
    r = subprocess.run(["make", "-C", str(_PKG / "csrc")] + (["-B"] if force else []), capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libgnnvc_hip.so failed:\n" + r.stdout + r.stderr)
    return _LIB


_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: torch bundles its own libamdhip64; importing torch first
    # makes this library bind to the runtime torch uses (loading ours first leaves torch
    # without a usable device).  Python callers here always sit next to torch.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    if not _LIB.exists():
        raise FileNotFoundError(f"{_LIB} not built: run `make -C {_PKG / 'csrc'}` "
                                "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(str(_LIB))
    vp, u32, u64, f32p = C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p
    L.gnnvc_abi_version.restype = C.c_int
    L.gnnvc_strerror.restype = C.c_char_p
    L.gnnvc_strerror.argtypes = [C.c_int]
    L.gnnvc_last_error.restype = C.c_char_p
    L.gnnvc_last_error.argtypes = [vp]
    L.gnnvc_create.argtypes = [C.POINTER(vp), C.c_char_p, C.c_size_t, C.c_int]
    L.gnnvc_create_multi.argtypes = [C.POINTER(vp), C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.c_int]
    L.gnnvc_destroy.argtypes = [vp]
    L.gnnvc_destroy.restype = None
    L.gnnvc_set_weight_scale.argtypes = [vp, C.c_float]
    L.gnnvc_set_stream.argtypes = [vp, vp]
    L.gnnvc_set_option.argtypes = [vp, C.c_char_p, C.c_long]
    L.gnnvc_get_info.argtypes = [vp, C.c_char_p, C.POINTER(C.c_long)]
    for name in ("gnnvc_num_layers", "gnnvc_is_fused", "gnnvc_in_width", "gnnvc_out_width",
                 "gnnvc_num_stages", "gnnvc_synchronize"):
        getattr(L, name).argtypes = [vp]
    L.gnnvc_stage_widths.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.gnnvc_upload_graph.argtypes = [vp, u32, vp, vp, vp, vp]
    L.gnnvc_attach_graph_device.argtypes = [vp, u32, u64, vp, vp, vp, vp]
    L.gnnvc_attach_graph_slice.argtypes = [vp, u32, u32, u32, u64, vp, vp, vp, vp]
    L.gnnvc_derive_graph_begin.argtypes = [vp, u32, vp, vp, vp]
    L.gnnvc_derive_graph_commit.argtypes = [vp, vp, u64, vp, vp]
    L.gnnvc_graph_row_hashes.argtypes = [vp, vp]
    L.gnnvc_graph_staging.argtypes = [vp, u32, u64] + [C.POINTER(vp)] * 4
    L.gnnvc_staged_columns_ready.argtypes = [vp, u64, u64]
    L.gnnvc_commit_staged_graph.argtypes = [vp]
    L.gnnvc_forward.argtypes = [vp, f32p, f32p, f32p]
    L.gnnvc_forward_device.argtypes = [vp, f32p, f32p, f32p]
    L.gnnvc_stage_forward_device.argtypes = [vp, C.c_int, u32, u32, f32p, f32p, f32p]
    L.gnnvc_reduction_flags.argtypes = [vp, u32, vp]
    L.gnnvc_stage_input_ready.argtypes = [vp, C.c_int, f32p, u32, u32]
    L.gnnvc_score_keys.argtypes = [vp, f32p, u32, vp, vp]
    L.gnnvc_live_columns.argtypes = [vp, f32p, u32, u32, C.POINTER(u32)]
    L.gnnvc_column_counts.argtypes = [vp, f32p, u32, u32, vp]
    L.gnnvc_pack_rows.argtypes = [vp, f32p, u32, u32, u32, u32, u32, f32p, vp, u32, vp]
    L.gnnvc_unpack_rows.argtypes = [vp, f32p, vp, u32, u32, u32, u32, u32, u32, f32p]
    L.gnnvc_unpack_gathered.argtypes = [vp, f32p, u32, u32, u64, u32, u32, u32, u32, u32, u32, u32, u32, u32, f32p]
    L.gnnvc_last_forward_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int]
    L.gnnvc_graph_layer_forward.argtypes = [vp, u32, f32p, f32p]
    L.gnnvc_linear_forward.argtypes = [vp, u32, u32, u32, f32p, f32p, f32p, f32p]
    L.gnnvc_relu_forward.argtypes = [vp, C.c_size_t, f32p, f32p]
    L.gnnvc_sigmoid_forward.argtypes = [vp, C.c_size_t, f32p, f32p]
    L.gnnvc_sgemm.argtypes = [vp, C.c_int, C.c_int, u32, u32, u32, f32p, u32, f32p, u32, C.c_float,
                              f32p, u32]
    L.gnnvc_kernel_trace.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.gnnvc_stream_sum.argtypes = [vp, f32p, u32, u32, C.c_int, f32p]
    for name in ABI_SYMBOLS:
        if name not in ("gnnvc_strerror", "gnnvc_last_error", "gnnvc_destroy"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def default_model_text() -> str:
    return (_PKG / "data" / "mwvc_model.txt").read_text()


def _np_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Engine:
    """One engine = one model on one GPU (mirrors `gnn::model`)."""

    def __init__(self, model_text: str | None = None, device: int = 0, devices=None):
        """devices = [ordinals]: several devices behind this one handle (gnnvc_create_multi); an ordinal may repeat."""
        self._L = load_library()
        self._h = C.c_void_p()
        raw = (model_text if model_text is not None else default_model_text()).encode()
        if devices is not None:
            arr = (C.c_int * len(devices))(*devices)
            rc = self._L.gnnvc_create_multi(C.byref(self._h), raw, len(raw), arr, len(devices))
        else:
            rc = self._L.gnnvc_create(C.byref(self._h), raw, len(raw), device)
        if rc != 0:
            self._h = C.c_void_p()
            raise GnnvcError(rc, self._L.gnnvc_strerror(rc).decode())
        self.n = 0
        self._keep = None

    # -- plumbing
    def _check(self, rc: int):
        if rc != 0:
            raise GnnvcError(rc, self._L.gnnvc_last_error(self._h).decode()
                             or self._L.gnnvc_strerror(rc).decode())

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.gnnvc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- model
    @property
    def num_layers(self) -> int:
        return self._L.gnnvc_num_layers(self._h)

    @property
    def fused(self) -> bool:
        return bool(self._L.gnnvc_is_fused(self._h))

    @property
    def in_width(self) -> int:
        return self._L.gnnvc_in_width(self._h)

    @property
    def out_width(self) -> int:
        return self._L.gnnvc_out_width(self._h)

    @property
    def num_stages(self) -> int:
        return self._L.gnnvc_num_stages(self._h)

    def stage_widths(self, stage: int):
        a, b = C.c_int(), C.c_int()
        self._check(self._L.gnnvc_stage_widths(self._h, stage, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_weight_scale(self, ws: float):
        self._check(self._L.gnnvc_set_weight_scale(self._h, C.c_float(ws)))

    def set_option(self, key: str, value: int):
        self._check(self._L.gnnvc_set_option(self._h, key.encode(), value))

    def get_info(self, key: str) -> int:
        v = C.c_long(0)
        self._check(self._L.gnnvc_get_info(self._h, key.encode(), C.byref(v)))
        return v.value

    def set_stream(self, hip_stream: int | None):
        self._check(self._L.gnnvc_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    # -- graph
    def upload_graph(self, g):
        """g: tools.graphgen.CsrGraph-like (n, rowptr u64, col u32, w u32, nw u32), host arrays."""
        rowptr = np.ascontiguousarray(g.rowptr, dtype=np.uint64)
        col = np.ascontiguousarray(g.col, dtype=np.uint32)
        w = np.ascontiguousarray(g.w, dtype=np.uint32)
        nw = np.ascontiguousarray(g.nw, dtype=np.uint32)
        self._check(self._L.gnnvc_upload_graph(self._h, g.n, _np_ptr(rowptr), _np_ptr(col),
                                               _np_ptr(w), _np_ptr(nw)))
        self.n = g.n

    def upload_graph_staged(self, g, pieces: int = 4):
        """Same hand-off through the engine's page-locked staging (gnnvc_graph_staging ..
        gnnvc_commit_staged_graph): the arrays are written into pinned memory and the column array
        goes out in `pieces` announced pieces."""
        n, nnz = int(g.n), int(g.rowptr[-1]) if g.n else 0
        ptr = [C.c_void_p() for _ in range(4)]
        self._check(self._L.gnnvc_graph_staging(self._h, n, 0, C.byref(ptr[0]), None, C.byref(ptr[2]), C.byref(ptr[3])))

        def view(p, count):
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(count,)) if count else np.zeros(0, np.uint32)
        if n:
            view(ptr[0], n + 1)[:] = np.asarray(g.rowptr, dtype=np.uint64).astype(np.uint32)
            view(ptr[2], n)[:] = g.w
            view(ptr[3], n)[:] = g.nw
        self._check(self._L.gnnvc_graph_staging(self._h, n, nnz, None, C.byref(ptr[1]), None, None))
        col = view(ptr[1], nnz)
        step = max(1, -(-nnz // max(1, pieces)))
        for lo in range(0, nnz, step):
            hi = min(nnz, lo + step)
            col[lo:hi] = g.col[lo:hi]
            if hi < nnz:            # the last piece is left to the commit
                self._check(self._L.gnnvc_staged_columns_ready(self._h, lo, hi - lo))
        self._check(self._L.gnnvc_commit_staged_graph(self._h))
        self.n = n

    def attach_graph_device(self, n: int, nnz: int, rowptr_ptr: int, col_ptr: int, w_ptr: int,
                            nw_ptr: int, keepalive=None):
        self._check(self._L.gnnvc_attach_graph_device(self._h, n, nnz, rowptr_ptr, col_ptr, w_ptr,
                                                      nw_ptr))
        self.n = n
        self._keep = keepalive

    def attach_graph_slice(self, n_global: int, row_lo: int, row_hi: int, nnz_local: int, rowptr_ptr: int, col_ptr: int,
                           w_ptr: int, nw_ptr: int, keepalive=None):
        """This rank's rows [row_lo, row_hi) of a vertex-partitioned graph (device arrays, see gnnvc.h)."""
        self._check(self._L.gnnvc_attach_graph_slice(self._h, n_global, row_lo, row_hi, nnz_local, rowptr_ptr, col_ptr,
                                                     w_ptr, nw_ptr))
        self.n = n_global
        self._keep = keepalive

    def derive_graph(self, g_new, old_row: np.ndarray):
        """Make g_new current by deriving it on the device from the resident graph (gnnvc_derive_graph_begin/_commit):
        old_row[u] = row of the resident graph that vertex u of g_new was, or 0xFFFFFFFF.  Returns the tail counts."""
        n = int(g_new.n)
        old_row = np.ascontiguousarray(old_row, dtype=np.uint32)
        rp = np.ascontiguousarray(np.asarray(g_new.rowptr, dtype=np.uint64).astype(np.uint32))
        tail = np.zeros(max(n, 1), dtype=np.uint32)
        self._check(self._L.gnnvc_derive_graph_begin(self._h, n, _np_ptr(old_row), _np_ptr(rp), _np_ptr(tail)))
        tail = tail[:n]
        col = np.asarray(g_new.col, dtype=np.uint32)
        ends = np.asarray(g_new.rowptr, dtype=np.int64)[1:]
        pieces = [col[e - t: e] for e, t in zip(ends[tail > 0], tail[tail > 0])]
        tails = np.ascontiguousarray(np.concatenate(pieces) if pieces else np.zeros(0, dtype=np.uint32), dtype=np.uint32)
        w = np.ascontiguousarray(g_new.w, dtype=np.uint32)
        nw = np.ascontiguousarray(g_new.nw, dtype=np.uint32)
        self._check(self._L.gnnvc_derive_graph_commit(self._h, _np_ptr(tails), tails.size, _np_ptr(w), _np_ptr(nw)))
        self.n = n
        return tail

    def row_hashes(self) -> np.ndarray:
        out = np.zeros(max(self.n, 1), dtype=np.uint64)
        self._check(self._L.gnnvc_graph_row_hashes(self._h, _np_ptr(out)))
        return out[: self.n]

    # -- forward
    def forward(self, x: np.ndarray, want_logits: bool = True, out=None):
        """Host forward: returns (scores[n, out_width], logits or None).  `out` = (scores, logits) arrays to fill
        instead of new ones (a caller that times the call keeps first-touch page faults out of it that way)."""
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(self.n, self.in_width)
        if out is not None:
            scores, logits = out
            for a in (scores, logits) if want_logits else (scores,):
                if a.dtype != np.float32 or a.shape != (self.n, self.out_width) or not a.flags.c_contiguous:
                    raise ValueError("out arrays must be C-contiguous float32 of shape (n, out_width)")
        else:
            scores = np.empty((self.n, self.out_width), dtype=np.float32)
            logits = np.empty((self.n, self.out_width), dtype=np.float32) if want_logits else None
        self._check(self._L.gnnvc_forward(self._h, _np_ptr(x), _np_ptr(scores),
                                          _np_ptr(logits) if want_logits else None))
        return scores, logits

    def forward_device(self, x_ptr: int, scores_ptr: int, logits_ptr: int = 0):
        self._check(self._L.gnnvc_forward_device(self._h, x_ptr, scores_ptr, logits_ptr or None))

    def stage_forward_device(self, stage: int, row_lo: int, row_hi: int, in_ptr: int, out_ptr: int,
                             logits_ptr: int = 0):
        self._check(self._L.gnnvc_stage_forward_device(self._h, stage, row_lo, row_hi, in_ptr,
                                                       out_ptr, logits_ptr or None))

    def stage_input_ready(self, stage: int, in_ptr: int, row_lo: int, row_hi: int):
        """Announce the complete, final input of `stage` before computing rows [row_lo, row_hi) of it in pieces."""
        self._check(self._L.gnnvc_stage_input_ready(self._h, stage, in_ptr, row_lo, row_hi))

    # -- feature-row codec of the inter-GPU exchange (device pointers)
    def live_columns(self, feat_ptr: int, rows: int, width: int = 16) -> int:
        m = C.c_uint32(0)
        self._check(self._L.gnnvc_live_columns(self._h, feat_ptr, rows, width, C.byref(m)))
        return int(m.value)

    def column_counts(self, feat_ptr: int, rows: int, width: int = 16) -> np.ndarray:
        counts = np.zeros(16, dtype=np.uint64)
        self._check(self._L.gnnvc_column_counts(self._h, feat_ptr, rows, width, _np_ptr(counts)))
        return counts

    def pack_rows(self, feat_ptr: int, row_lo: int, row_hi: int, mask: int, kp: int, dense_ptr: int,
                  flag_ptr: int, exc_ptr: int = 0, exc_cap: int = 0, width: int = 16):
        self._check(self._L.gnnvc_pack_rows(self._h, feat_ptr, width, row_lo, row_hi, mask, kp, dense_ptr,
                                            exc_ptr or None, exc_cap, flag_ptr))

    def unpack_rows(self, dense_ptr: int, row_lo: int, row_hi: int, mask: int, kp: int, feat_ptr: int,
                    exc_ptr: int = 0, exc_cap: int = 0, width: int = 16):
        self._check(self._L.gnnvc_unpack_rows(self._h, dense_ptr, exc_ptr or None, exc_cap, width, row_lo, row_hi,
                                              mask, kp, feat_ptr))

    def reduction_flags(self, max_degree: int = 20) -> np.ndarray:
        flags = np.zeros(self.n, dtype=np.uint8)
        self._check(self._L.gnnvc_reduction_flags(self._h, max_degree, _np_ptr(flags)))
        return flags

    def score_keys(self, scores_ptr: int = 0, n: "int | None" = None):
        """(min(s, 1 - s), s > 0.5) per score, from device scores (default: the last host forward's)."""
        n = self.n if n is None else n
        keys = np.empty(n, dtype=np.float32)
        above = np.zeros(n, dtype=np.uint8)
        self._check(self._L.gnnvc_score_keys(self._h, scores_ptr or None, n, _np_ptr(keys), _np_ptr(above)))
        return keys, above

    def synchronize(self):
        self._check(self._L.gnnvc_synchronize(self._h))

    def last_forward_ms(self):
        total = C.c_float()
        stages = (C.c_float * 8)()
        self._check(self._L.gnnvc_last_forward_ms(self._h, C.byref(total), stages, 8))
        ns = max(self.num_stages, 1)
        return total.value, [stages[i] for i in range(ns)]

    def kernel_trace(self, max_records: int = 256):
        """[(kernel name, ms)] of the last forward_device under option "kernel_trace" (main-stream kernels)."""
        names = (C.c_char_p * max_records)()
        ms = (C.c_float * max_records)()
        cnt = C.c_int(0)
        self._check(self._L.gnnvc_kernel_trace(self._h, max_records, names, ms, C.byref(cnt)))
        return [(names[i].decode(), float(ms[i])) for i in range(min(cnt.value, max_records))]

    # -- layer-level entry points
    def graph_layer(self, h: np.ndarray) -> np.ndarray:
        h = np.ascontiguousarray(h, dtype=np.float32).reshape(self.n, -1)
        f = h.shape[1]
        out = np.empty((self.n, 2 * f + 3), dtype=np.float32)
        self._check(self._L.gnnvc_graph_layer_forward(self._h, f, _np_ptr(h), _np_ptr(out)))
        return out

    def linear(self, h: np.ndarray, W: np.ndarray, bias: np.ndarray) -> np.ndarray:
        h = np.ascontiguousarray(h, dtype=np.float32)
        W = np.ascontiguousarray(W, dtype=np.float32)
        bias = np.ascontiguousarray(bias, dtype=np.float32)
        out = np.empty((h.shape[0], W.shape[1]), dtype=np.float32)
        self._check(self._L.gnnvc_linear_forward(self._h, h.shape[0], W.shape[0], W.shape[1],
                                                 _np_ptr(h), _np_ptr(W), _np_ptr(bias), _np_ptr(out)))
        return out

    def relu(self, h: np.ndarray) -> np.ndarray:
        h = np.ascontiguousarray(h, dtype=np.float32)
        out = np.empty_like(h)
        self._check(self._L.gnnvc_relu_forward(self._h, h.size, _np_ptr(h), _np_ptr(out)))
        return out

    def sigmoid(self, h: np.ndarray) -> np.ndarray:
        h = np.ascontiguousarray(h, dtype=np.float32)
        out = np.empty_like(h)
        self._check(self._L.gnnvc_sigmoid_forward(self._h, h.size, _np_ptr(h), _np_ptr(out)))
        return out

    def stream_sum(self, values: np.ndarray, mode: int = 0) -> np.ndarray:
        """Sequential fp32 sums of the rows of `values` (streams x len) through the giant-row kernels."""
        v = np.ascontiguousarray(values, dtype=np.float32)
        v = v.reshape(1, -1) if v.ndim == 1 else v
        out = np.empty(v.shape[0], dtype=np.float32)
        self._check(self._L.gnnvc_stream_sum(self._h, _np_ptr(v), v.shape[0], v.shape[1], mode, _np_ptr(out)))
        return out

    def sgemm(self, A: np.ndarray, B: np.ndarray, C_in: np.ndarray | None = None, beta: float = 0.0,
              trans_a: bool = False, trans_b: bool = False) -> np.ndarray:
        A = np.ascontiguousarray(A, dtype=np.float32)
        B = np.ascontiguousarray(B, dtype=np.float32)
        m = A.shape[1] if trans_a else A.shape[0]
        k = A.shape[0] if trans_a else A.shape[1]
        n = B.shape[0] if trans_b else B.shape[1]
        out = (np.zeros((m, n), dtype=np.float32) if C_in is None
               else np.ascontiguousarray(C_in, dtype=np.float32).copy())
        self._check(self._L.gnnvc_sgemm(self._h, int(trans_a), int(trans_b), m, n, k, _np_ptr(A),
                                        A.shape[1], _np_ptr(B), B.shape[1], C.c_float(beta),
                                        _np_ptr(out), n))
        return out


class EngineRowCodec:
    """distributed.RowCodec served by the engine's HIP kernels (tensors are device tensors).
    A piece region is a flat fp32 tensor: dense_rows x kp floats, then the exception list
    (4 + 4 * cap 32-bit words)."""

    def __init__(self, engine: "Engine"):
        self.e = engine

    def column_counts(self, feat, n: int):
        return [int(v) for v in self.e.column_counts(feat.data_ptr(), n, feat.shape[1])]

    def pack(self, feat, lo: int, hi: int, pk, region, dense_rows: int, flag) -> None:
        base = region.data_ptr()
        self.e.pack_rows(feat.data_ptr(), lo, hi, pk.mask, pk.kp, base, flag.data_ptr(),
                         base + 4 * dense_rows * pk.kp, pk.cap, feat.shape[1])

    def unpack(self, region, dense_rows: int, lo: int, hi: int, pk, feat) -> None:
        base = region.data_ptr()
        self.e.unpack_rows(base, lo, hi, pk.mask, pk.kp, feat.data_ptr(), base + 4 * dense_rows * pk.kp, pk.cap,
                           feat.shape[1])

    def unpack_gathered(self, buf, world: int, skip_rank: int, dense_rows: int, per: int, off: int, size: int, n: int,
                        pk, feat) -> None:
        """Every peer's region of one all-gathered piece ([rank][piece_words] in `buf`) in one launch."""
        e = self.e
        e._check(e._L.gnnvc_unpack_gathered(e._h, buf.data_ptr(), world, skip_rank, pk.piece_words(dense_rows), dense_rows,
                                            pk.cap, feat.shape[1], per, off, size, n, pk.mask, pk.kp, feat.data_ptr()))
