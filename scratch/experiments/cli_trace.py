"""Per-predict-call timing of the whole ./GNN_VC run behind the HIP engine (GPU box), for the three hand-off modes:
GNNVC_DELTA unset (staged full hand-off), 1 (derived on the device + hash verification), 2 (derived, unverified).

Runs oracle/_ref/GNN_VC_hip_fastio (reference driver + this repo's host mirror + libgnnvc_hip.so + fast file I/O) with
GNNVC_TRACE=1 on a generated graph and sums the trace.
usage: python scratch/experiments/cli_trace.py [n] [m] [seed]
"""
import ctypes as C
import hashlib
import os
import pathlib
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tools import graphgen as gg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 10 * n
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 2
g = gg.erdos_renyi(n, m, seed)
L = C.CDLL(str(ROOT / "gnn-mwvc_amd" / "libgnnvc_metis.so"))
L.gnnvc_host_write_metis.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
with tempfile.TemporaryDirectory() as d:
    path = pathlib.Path(d) / f"er{n}.graph"
    rp = np.ascontiguousarray(g.rowptr, dtype=np.uint64)
    assert L.gnnvc_host_write_metis(str(path).encode(), g.n, rp.ctypes.data, g.col.ctypes.data, g.w.ctypes.data) == 0
    results = {}
    for mode in ("0", "1", "2"):
        env = dict(os.environ, GNNVC_TRACE="1")
        env.pop("GNNVC_DELTA", None)
        if mode != "0":
            env["GNNVC_DELTA"] = mode
        out = pathlib.Path(d) / f"out{mode}"
        t = time.time()
        r = subprocess.run([str(ROOT / "oracle/_ref/GNN_VC_hip_fastio"), str(path), str(out), "0", "-1", "0"],
                           capture_output=True, text=True, env=env)
        wall = time.time() - t
        calls = re.findall(r"gnnvc predict n=(\d+) nnz=(\d+) hand-off=([\d.]+)ms( \(derived on the device\))? forward=([\d.]+)ms", r.stderr)
        hand = sum(float(c[2]) for c in calls)
        fwd = sum(float(c[4]) for c in calls)
        derived = sum(1 for c in calls if c[3])
        results[mode] = (tuple(r.stdout.strip().split(",")[:2]), hashlib.md5(out.read_bytes()).hexdigest())
        print(f"GNNVC_DELTA={mode}: wall {wall:.1f} s rc={r.returncode} {r.stdout.strip()}")
        print(f"   {len(calls)} predict calls, {derived} derived on the device; hand-off total {hand:.1f} ms, forwards total {fwd:.1f} ms")
        for c in calls:
            print(f"      n={c[0]:>8} nnz={c[1]:>9} hand-off {float(c[2]):8.3f} ms{' derived' if c[3] else '        '} forward {float(c[4]):7.3f} ms")
    print("identical results across modes:", len(set(results.values())) == 1, results["0"])
