"""CPU-side checks of the drop-in boundary: the library loads, exports every
symbol include/gnnvc.h declares, and refuses to compute without a GPU (no CPU
fallback).  No compute calls here."""
import ctypes as C
import pathlib
import re

import pytest

import gnn_mwvc_amd as G
from gnn_mwvc_amd.engine import ABI_SYMBOLS

ROOT = pathlib.Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    G.build_library()
    return G.load_library()


def test_header_and_binding_agree():
    hdr = (ROOT / "include" / "gnnvc.h").read_text()
    declared = set(re.findall(r"\b(gnnvc_[a-z_0-9]+)\s*\(", hdr))
    declared.discard("gnnvc_engine")
    assert declared == set(ABI_SYMBOLS)


def test_library_exports_every_symbol(lib):
    for name in ABI_SYMBOLS:
        assert hasattr(lib, name), name
    assert lib.gnnvc_abi_version() == 1


def test_error_strings(lib):
    assert lib.gnnvc_strerror(0) == b"ok"
    for code in (-1, -2, -3, -4, -5):
        assert lib.gnnvc_strerror(code)


def test_null_arguments_are_rejected(lib):
    assert lib.gnnvc_create(None, b"x", 1, 0) == -1
    assert lib.gnnvc_set_weight_scale(None, C.c_float(1.0)) == -1
    assert lib.gnnvc_forward(None, None, None, None) == -1
    lib.gnnvc_destroy(None)  # no-op


def test_no_cpu_fallback(lib, model_text):
    """Without a HIP device the engine refuses to exist — it never computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(G.GnnvcError) as ei:
        G.Engine(model_text)
    assert ei.value.code == -2


def test_malformed_model_is_invalid(lib):
    h = C.c_void_p()
    assert lib.gnnvc_create(C.byref(h), b"Name notanumber Layers", 22, 0) == -1
    assert not h.value
