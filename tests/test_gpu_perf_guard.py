"""Performance guard: the engine decides per graph which plans to build (LDS table, compact table, pruned
adjacency, sorted tiles, long / giant thresholds) from rules tuned on four graph families.  Nothing else in the suite would
notice a rule change that makes the DEFAULTS slower than the plain kernels on some family.  This test would: a fixed panel of
20 random large graphs (tools/panel_graphs.py: sparse and dense Erdős–Rényi, R-MAT, power-law with hubs, degree-uniform with
hubs; 0.3 - 4 M vertices), for each the steady state with default options against the same graph with every plan off —
  * logits and scores bit-identical, forward after forward (the first one included);
  * engine defaults <= 1.10 x "all plans off" (+ 25 us of timer slack on forwards of a fraction of a millisecond);
  * a fresh graph's first forward with the defaults <= 1.15 x the plain first forward (round 4; it was 1.35: plans built at
    hand-off are outside it — a large skewed graph's predicted pruned adjacency among them — and what a first forward still
    does inside itself, the filter's marks and look-ups on mid-size skewed graphs, has to pay there).
"""
import time

import pytest

pytestmark = pytest.mark.gpu

CASES = list(range(20))


def _run(G, torch, g, x, opts, dev):
    e = G.Engine(G.default_model_text(), device=0)
    try:
        for k, v in opts.items():
            e.set_option(k, v)
        e.set_weight_scale(g.ws)
        e.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
        sc = torch.zeros(g.n, device=dev)
        lg = torch.zeros(g.n, device=dev)
        torch.cuda.synchronize()
        outs, first_ms = [], 0.0
        for i in range(4):
            t = time.perf_counter()
            e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
            e.synchronize()
            if i == 0:
                first_ms = (time.perf_counter() - t) * 1e3
            outs.append((sc.clone(), lg.clone()))
        best = 1e9
        for _ in range(3):                       # the best of three batches of five: a guard must not trip on a hiccup
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(5):
                e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
            e.synchronize()
            best = min(best, (time.perf_counter() - t) * 200.0)
        return outs, best, first_ms
    finally:
        e.close()


def _first_again(G, torch, g, x, opts, dev):
    """One more fresh engine's first forward (hipMalloc on this stack now and then takes 100+ ms, and a single
    wall-clock shot per configuration would trip on it now and then across 20 cases: the guard compares the better of two)."""
    e = G.Engine(G.default_model_text(), device=0)
    try:
        for k, v in opts.items():
            e.set_option(k, v)
        e.set_weight_scale(g.ws)
        e.attach_graph_device(g.n, g.nnz, g.rowptr.data_ptr(), g.col.data_ptr(), g.w.data_ptr(), g.nw.data_ptr(), keepalive=g)
        sc = torch.zeros(g.n, device=dev)
        lg = torch.zeros(g.n, device=dev)
        torch.cuda.synchronize()
        t = time.perf_counter()
        e.forward_device(x.data_ptr(), sc.data_ptr(), lg.data_ptr())
        e.synchronize()
        return (time.perf_counter() - t) * 1e3
    finally:
        e.close()


@pytest.mark.parametrize("case", CASES)
def test_defaults_are_not_slower_than_the_plain_kernels(case):
    import torch
    import gnn_mwvc_amd as G
    from tools import panel_graphs as pg
    dev = torch.device("cuda", 0)
    kind, g = pg.panel_graph(case, dev)
    x = g.x().contiguous()
    ref, ms_plain, first_plain = _run(G, torch, g, x, pg.PLAIN, dev)
    got, ms, first = _run(G, torch, g, x, {}, dev)
    first_plain = min(first_plain, _first_again(G, torch, g, x, pg.PLAIN, dev))
    first = min(first, _first_again(G, torch, g, x, {}, dev))
    for i, (sc, lg) in enumerate(got):
        assert torch.equal(lg.view(torch.int32), ref[0][1].view(torch.int32)), (case, kind, i)
        assert torch.equal(sc.view(torch.int32), ref[0][0].view(torch.int32)), (case, kind, i)
    assert ms <= 1.10 * ms_plain + 0.025, f"case {case} ({kind}, n {g.n}, nnz {g.nnz}): defaults {ms:.3f} ms vs plans off {ms_plain:.3f} ms"
    assert first <= 1.15 * first_plain + 0.05, f"case {case} ({kind}, n {g.n}, nnz {g.nnz}): first forward {first:.3f} ms vs plain {first_plain:.3f} ms"
    del g, x, ref, got
    torch.cuda.empty_cache()
