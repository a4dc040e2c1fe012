"""Round 4: the F = 1 stage of the metric graph with weights beyond a byte: the 10-bit and 16-bit LDS tables against the byte
table (U[20,120]) and against the column-blocked plan they replace.  usage: r4_wide_table.py"""
import sys, time
sys.path.insert(0, ".")
import torch
import gnn_mwvc_amd as G
from tools import graphgen_torch as ggt
dev = torch.device("cuda", 0)
for hi, label in ((120, "U[20,120]"), (1000, "U[20,1000]"), (60000, "U[20,60000]")):
    g = ggt.erdos_renyi(10_000_000, 100_000_000, 10, dev, 20, hi)
    x = g.x().contiguous()
    sc = torch.zeros(g.n, device=dev); lg = torch.zeros(g.n, device=dev)
    ref = None
    for opts in ({}, {"lds_table": 0}, {"lds_table_bits": 16}):
        if opts.get("lds_table_bits") and hi > 120 and opts["lds_table_bits"] == 16 and hi == 60000:
            continue
        e = G.Engine(G.default_model_text(), device=0)
        for k, v in opts.items(): e.set_option(k, v)
        e.set_option("forward_timing", 2)
        e.set_weight_scale(g.ws)
        e.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
        for _ in range(4): e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
        e.synchronize()
        t = time.perf_counter()
        for _ in range(10): e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
        e.synchronize()
        ms = (time.perf_counter() - t) * 100
        stage = e.last_forward_ms()[1]
        bad = 0 if ref is None else int((lg.view(torch.int32) != ref.view(torch.int32)).sum())
        if ref is None: ref = lg.clone()
        print(f"{label:12s} {str(opts):28s} forward {ms:6.3f} ms  stages {[round(v, 3) for v in stage]}  bits {e.get_info('lds_table_bits')} "
              f"table_ok {e.get_info('lds_table_last_ok')} blocked {e.get_info('blocked_stage0_active')}  mismatches vs first {bad}", flush=True)
        e.close()
    del g, x
    torch.cuda.empty_cache()
