"""gnn-mwvc_amd — MI355X-native GNN-VC scoring engine (the one hot path of
KennethLangedal/GNN-MWVC: `gnn::model::predict`).

Layout:
  csrc/      HIP kernels (gfx950) + the C ABI implementation -> libgnnvc_hip.so
  host/      C++ mirror of the reference's host interface (matrix, gnn::model, ...)
  engine.py  ctypes binding of the C ABI (include/gnnvc.h) for tests and the bench
  data/      the trained model in the reference's text format
"""
from .engine import (Engine, EngineRowCodec, GnnvcError, build_library, default_model_text, library_path,  # noqa: F401
                     load_library)
