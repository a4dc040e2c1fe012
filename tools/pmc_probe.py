#!/usr/bin/env python3
"""PMC probe: the metric graph, one calibration copy of known size, a few forwards.
Run under `rocprofv3 --pmc ... --output-format csv -d <dir> -- python3 tools/pmc_probe.py`
(counters in their own run, never together with a trace domain).  The calibration
kernel reads and writes exactly CALIB_BYTES so the FETCH_SIZE / WRITE_SIZE scale on
gfx950 can be checked against a known byte count in the same run."""
import argparse
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

CALIB_BYTES = 1 << 30


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--m", type=int, default=100_000_000)
    ap.add_argument("--forwards", type=int, default=6)   # plans are built by the second forward, table tiles fit from the fourth; the last one is steady state
    ap.add_argument("--workload", default="", help="one of bench.py's workloads instead of --n / --m (er10m = the default graph)")
    ap.add_argument("--plain", type=int, default=0, help="1 = every per-graph plan off (the plain kernels' traffic)")
    ap.add_argument("--first", type=int, default=0, help="1 = mark the FIRST forward of a fresh engine instead of the last (score-once traffic)")
    a = ap.parse_args()
    import torch
    import gnn_mwvc_amd as G
    from tools import graphgen_torch as ggt
    dev = torch.device("cuda", 0)
    if a.workload:
        import bench
        g, _ = bench.build_workload(a.workload, ggt, dev)
    else:
        g = ggt.erdos_renyi(a.n, a.m, 10, dev)
    eng = G.Engine(G.default_model_text(), device=0)
    if a.plain:
        from tools import panel_graphs
        for k, v in panel_graphs.PLAIN.items():
            eng.set_option(k, v)
    eng.set_weight_scale(g.ws)
    eng.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(),
                            g.nw.data_ptr(), keepalive=g)
    x = g.x().contiguous()
    sc = torch.zeros(g.n, dtype=torch.float32, device=dev)
    lg = torch.zeros(g.n, dtype=torch.float32, device=dev)
    # calibration: a streaming copy of exactly CALIB_BYTES (reads 1 GiB, writes 1 GiB)
    src = torch.empty(CALIB_BYTES // 4, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    torch.cuda.synchronize()
    torch.add(src, 1.0, out=dst)   # the only 'CUDAFunctorOnSelf_add<float>' kernel of the run
    torch.cuda.synchronize()
    # the marker: the only 'add<double>' kernel of the run sits right in front of the forward the summary takes (its last, the
    # steady state — or its first with --first 1)
    mark = torch.zeros(64, dtype=torch.float64, device=dev)
    for i in range(a.forwards):
        if (a.first and i == 0) or (not a.first and i == a.forwards - 1):
            torch.add(mark, 1.0, out=mark)
            torch.cuda.synchronize()
        eng.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
        eng.synchronize()
        if a.first and i == 0:
            break
    print("probe done", g.n, g.nnz)
    eng.close()


if __name__ == "__main__":
    main()
