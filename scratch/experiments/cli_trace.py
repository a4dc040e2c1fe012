"""Per-predict-call timing of the whole ./GNN_VC run behind the HIP engine (GPU box).

Runs oracle/_ref/GNN_VC_hip (reference driver + this repo's host mirror +
libgnnvc_hip.so) with GNNVC_TRACE=1 on a generated graph and prints the trace.
usage: python scratch/experiments/cli_trace.py [n] [m] [seed]
"""
import pathlib
import subprocess
import sys
import tempfile
import time

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tools import graphgen as gg  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 10 * n
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 2
g = gg.erdos_renyi(n, m, seed)
with tempfile.TemporaryDirectory() as d:
    path = pathlib.Path(d) / f"er{n}.graph"
    path.write_text(gg.metis_text(g))
    t = time.time()
    r = subprocess.run([str(ROOT / "oracle/_ref/GNN_VC_hip"), str(path), str(pathlib.Path(d) / "out"), "0", "-1", "0"],
                       capture_output=True, text=True, env={"GNNVC_TRACE": "1", "PATH": "/usr/bin:/bin"})
    print("wall %.2f s  rc=%d  stdout: %s" % (time.time() - t, r.returncode, r.stdout.strip()))
    print(r.stderr.strip())
