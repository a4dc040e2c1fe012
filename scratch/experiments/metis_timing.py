"""f-4 at the size it exists for: the metric graph (Erdős–Rényi 10 M vertices / 100 M edges) as a ~1.5 GB METIS text file —
this repo's reader (host/metis_loader.cpp) against the reference's parse_graph (src/GNN_VC.cpp:34-91, through
oracle/_ref/ref_parse.so), and the two result-file writers at 10 M lines.  Build container, CPU only; needs ~25 GB of RAM.
python scratch/experiments/metis_timing.py [n] [m]"""
import ctypes as C
import pathlib
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from tools import graphgen as gg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
t0 = time.time()
g = gg.erdos_renyi(n, m, 10)
print(f"graph built in {time.time() - t0:.0f} s: {g.n} vertices, {g.n_edges} edges", flush=True)
L = C.CDLL(str(ROOT / "gnn-mwvc_amd" / "libgnnvc_metis.so"))
L.gnnvc_host_write_metis.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
L.gnnvc_host_write_cover.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t]
d = pathlib.Path(tempfile.mkdtemp(dir="/tmp"))
path = d / "metric.graph"
rp = np.ascontiguousarray(g.rowptr, dtype=np.uint64)
t0 = time.time()
assert L.gnnvc_host_write_metis(str(path).encode(), g.n, rp.ctypes.data, g.col.ctypes.data, g.w.ctypes.data) == 0
print(f"METIS text written in {time.time() - t0:.1f} s: {path.stat().st_size / 1e9:.2f} GB", flush=True)
del g


def load(lib, fn, *extra):
    nn, mm = C.c_uint32(), C.c_uint64()
    w, p = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)()
    t = time.time()
    rc = getattr(lib, fn)(str(path).encode(), C.byref(nn), C.byref(mm), C.byref(w), C.byref(p), *extra)
    dt = time.time() - t
    assert rc == 0
    pairs = np.ctypeslib.as_array(p, shape=(2 * mm.value,))
    h = int(pairs[::1001].astype(np.uint64).sum())      # a cheap fingerprint of the edge list
    libc = C.CDLL(None)
    libc.free(w)
    libc.free(p)
    return dt, nn.value, mm.value, h


for threads in (8, 1):
    dt, a, b, h = load(L, "gnnvc_host_load_metis", C.c_uint(threads))
    print(f"this repo's reader, {threads} thread(s): {dt:.1f} s  ({a} vertices, {b} edges, fingerprint {h})", flush=True)
subprocess.run(["make", "-C", str(ROOT / "oracle"), "_ref/ref_parse.so"], check=True, capture_output=True)
R = C.CDLL(str(ROOT / "oracle" / "_ref" / "ref_parse.so"))
dt, a, b, h = load(R, "ref_parse")
print(f"reference parse_graph: {dt:.1f} s  ({a} vertices, {b} edges, fingerprint {h})", flush=True)
cover = (np.random.default_rng(1).random(n) < 0.5).astype(np.uint8)
t = time.time()
L.gnnvc_host_write_cover(str(d / "res.out").encode(), cover.ctypes.data, n)
t_ours = time.time() - t
src = d / "w.cpp"
src.write_text('#include <fstream>\n#include <vector>\n#include <cstdlib>\nusing namespace std;\nint main(int c,char**v){size_t n=atol(v[2]);vector<char> s(n);for(size_t i=0;i<n;++i)s[i]=i*2654435761u>>31&1;ofstream os(v[1]);for(size_t u=0;u<n;++u){os<<(s[u]?1:0)<<endl;}return 0;}\n')
subprocess.run(["g++", "-O2", "-o", str(d / "w"), str(src)], check=True)
t = time.time()
subprocess.run([str(d / "w"), str(d / "res2.out"), str(n)], check=True)
print(f"result file, {n} lines: one write {t_ours:.2f} s, `os << .. << endl` per line {time.time() - t:.2f} s", flush=True)
for f in d.iterdir():
    f.unlink()
d.rmdir()
