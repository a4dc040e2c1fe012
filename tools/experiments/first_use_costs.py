"""Where does a first-use forward spend its extra milliseconds?  (GPU box)

Times host forwards on one engine while features are switched on one at a time
(sorted tiles, long rows), so that each first use (buffer allocation, stream
creation, first launch of a kernel) shows up separately from the steady state.
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import gnn_mwvc_amd as G  # noqa: E402
from tools import graphgen as gg  # noqa: E402


def timed(label, fn):
    t = time.perf_counter()
    fn()
    print(f"{label:48s} {1e3 * (time.perf_counter() - t):8.3f} ms")


e = G.Engine(G.default_model_text(), device=0)
g = gg.erdos_renyi(200000, 600000, 3)
x = g.x()
e.set_option("sorted_tiles", 0)
timed("upload (natural tiles)", lambda: e.upload_graph(g))
timed("forward #1", lambda: e.forward(x))
timed("forward #2", lambda: e.forward(x))
e.set_option("sorted_tiles", 1)
timed("upload (sorted tiles forced)", lambda: e.upload_graph(g))
timed("forward #1 sorted (allocates the tile order)", lambda: e.forward(x))
timed("forward #2 sorted", lambda: e.forward(x))
timed("upload again", lambda: e.upload_graph(g))
timed("forward #1 sorted, buffers already there", lambda: e.forward(x))
h = gg.hub_graph(200000, 600000, 3, 65536, seed=5)
timed("upload hub graph (long rows: aux stream created)", lambda: e.upload_graph(h))
timed("forward #1 hubs", lambda: e.forward(h.x()))
timed("forward #2 hubs", lambda: e.forward(h.x()))
