"""Where does a tile kernel's time go on a mid-size graph?  Needs a library built with -DGNNVC_PHASE_PROBE=1
(scratch/experiments/phase_probe.sh builds it and points GNNVC_LIBRARY at it): a wave's first lane stamps the 100 MHz wall clock at
the phases of its tile; this script prints, per probed kernel, when waves start (the launch ramp), how long each phase takes
(median / p90 / max over waves) and when the last wave ends.
python scratch/experiments/phase_probe.py N M [key=value ...]"""
import sys, pathlib, ctypes as C
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
import torch
import gnn_mwvc_amd as G
from gnn_mwvc_amd import engine as E
from tools import graphgen_torch as ggt

dev = torch.device("cuda", 0)
n, m = int(sys.argv[1]), int(sys.argv[2])
opts = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in sys.argv[3:]}
g = ggt.erdos_renyi(n, m, 1, dev)
x = g.x().contiguous()
sc = torch.zeros(g.n, device=dev); lg = torch.zeros(g.n, device=dev)
e = G.Engine(G.default_model_text(), device=0)
for k, v in opts.items():
    e.set_option(k, v)
e.set_weight_scale(g.ws)
e.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
for _ in range(12):
    e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
e.synchronize()
L = E.load_library()
L.gnnvc_debug_probe.argtypes = [C.c_void_p, C.c_int]
nwaves = ((n + 31) // 32 + 128) * 2
names = {1: "k_stage_f16 (stage 1)", 2: "k_stage_f16 (stage 2)", 11: "k_stage_t4 (stage 1)", 12: "k_stage_t4 (stage 2)"}
phase_names = {0: "start", 1: "choice known", 2: "indices staged", 3: "gathered", 4: "dirty rows done", 5: "input tile built", 6: "layers 1-2 done", 7: "end", 8: "layer 3 done", 9: "output tile in LDS", 10: "emitted", 11: "end (stores issued)"}
for kind in (1, 2, 11, 12):
    buf = torch.zeros(nwaves * 16, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    assert L.gnnvc_debug_probe(buf.data_ptr(), kind) == 0
    for _ in range(3):
        buf.zero_()
        torch.cuda.synchronize()
        e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
        e.synchronize()
    t = buf.cpu().numpy().reshape(-1, 16).astype(np.int64)
    live = t[:, 0] != 0
    if not live.any():
        print(f"{names[kind]}: not run")
        continue
    t = t[live]
    t0 = t[:, 0].min()
    print(f"{names[kind]}: {len(t)} waves")
    st = (t[:, 0] - t0) / 100.0
    print(f"   wave starts after the first: median {np.median(st):.2f} us, p90 {np.percentile(st, 90):.2f}, last {st.max():.2f}")
    prev = 0
    for ph in range(1, 16):
        ok = t[:, ph] != 0
        if not ok.any():
            continue
        d = (t[ok, ph] - t[ok, prev]) / 100.0
        at = (t[ok, ph] - t0) / 100.0
        print(f"   {phase_names[prev]:>16s} -> {phase_names[ph]:<16s}: median {np.median(d):6.2f} us, p90 {np.percentile(d, 90):6.2f}, max {d.max():6.2f}"
              f"   | reached: median {np.median(at):6.2f}, last {at.max():6.2f}  ({ok.sum()} waves)")
        prev = ph
assert L.gnnvc_debug_probe(None, 0) == 0
e.set_option("kernel_trace", 1)
e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr()); e.synchronize()
e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr()); e.synchronize()
print("event times (us):", [(k.split("<")[0].strip("("), round(ms * 1e3, 1)) for k, ms in e.kernel_trace(64)])
e.close()
