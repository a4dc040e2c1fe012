"""A fixed panel of random LARGE graphs (0.3 - 4 M vertices, up to ~100 M adjacency entries; device-side generators) over the
families the engine's plan rules were tuned on: sparse and dense Erdős–Rényi, R-MAT, power-law with hubs, nearly
degree-uniform with hubs.  case -> (kind, DeviceCsr); the same case number always gives the same graph.  Used by
tests/test_gpu_perf_guard.py and scratch/experiments/fuzz_large.py."""
import numpy as np

from tools import graphgen_torch as ggt

KINDS = ("er", "rmat", "powerlaw", "er_dense", "uniform_hubs")


def panel_graph(case: int, dev, big: bool = False):
    rng = np.random.default_rng(case)
    kind = str(rng.choice(list(KINDS)))
    s = int(rng.integers(1 << 30))
    if kind == "er":
        n = int(rng.integers(4_000_000, 12_000_000) if big else rng.integers(300_000, 4_000_000))
        g = ggt.erdos_renyi(n, int(n * rng.uniform(3, 14)), s, dev)
    elif kind == "rmat":
        g = ggt.rmat(int(rng.integers(21, 24) if big else rng.integers(17, 22)), int(rng.integers(4, 20)), s, dev)
    elif kind == "powerlaw":
        n = int(rng.integers(3_000_000, 8_000_000) if big else rng.integers(300_000, 3_000_000))
        g = ggt.power_law_hubs(n, float(rng.uniform(6, 20)), float(rng.uniform(2.0, 2.5)), int(rng.integers(0, 9)),
                               int(rng.integers(1000, 200_000)), s, dev)
    elif kind == "uniform_hubs":   # nearly degree-uniform + a few hubs
        n = int(rng.integers(3_000_000, 8_000_000) if big else rng.integers(300_000, 3_000_000))
        g = ggt.power_law_hubs(n, float(rng.uniform(8, 24)), float(rng.uniform(3.5, 5.0)), int(rng.integers(1, 9)),
                               int(rng.integers(5000, 300_000)), s, dev)
    else:
        n = int(rng.integers(1_000_000, 2_500_000) if big else rng.integers(300_000, 900_000))
        g = ggt.erdos_renyi(n, int(n * rng.uniform(30, 60)), s, dev)
    return kind, g


# every per-graph plan off, fixed thresholds: the plain kernels
PLAIN = {"lds_table": 0, "compact_gather": 0, "blocked_stage0": 0, "prune_zero_rows": 0, "giant_segments": 0, "sorted_tiles": 0, "table_tiles": 0,
         "long_row_threshold": 512, "giant_row_threshold": 16384}
