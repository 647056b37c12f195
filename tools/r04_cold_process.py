#!/usr/bin/env python3
"""Where a FRESH process spends the filter step (what `filter.py` pays once per graph, submit_job.py:20-21): every stage of
scan_topk on a freshly built graph, host-timed with a synchronisation after each -- first-time code-object loads, first-time
hipMalloc and first uses of torch operators show up in the stage that triggers them.  argv: ddi | ppa (the --synthetic stand-ins'
sizes); env WARM=1 runs the whole step once before (the same stages, warm: the difference is the one-off cost)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t_imp = time.perf_counter()
import torch, eps_amd
from eps_amd import candidates, ops, scan, synth
from eps_amd.graph import CSRGraph
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "ppa"
if which == "ppa":
    g0 = synth.ppa_like(seed=3, device=dev)
    K = 4_000_000
else:
    g0 = synth.rmat_graph(12, 64, 3, dev)
    K = 100_000
torch.cuda.synchronize()
print(f"imports + graph: {time.perf_counter() - t_imp:.2f} s; N {g0.n_rows} nnz {g0.nnz()}")
def stages(g, tag):
    acc = []
    def T(label, fn):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        acc.append((label, (time.perf_counter() - t) * 1e3)); return r
    w = T("node weights (col_sums + 1/log)", lambda: node_weight_table(g, ops.W_AA))
    gs, perm = T("hubs-first copy", lambda: scan.scan_graph(g, build=True))
    T("revpos / symmetry / scalars", lambda: scan.scan_available(g))
    T("score bound", lambda: candidates.fused_scores_fit(g, w))
    T("fixed-point weights", lambda: scan._scan_weights(g, gs, perm, w))
    T("column order", lambda: scan.column_order(gs))
    T("variant, bounds, cuts, window paths", lambda: (scan.screen_variant(gs), scan.screen_tables(gs) if scan.one_pass_available(gs) else None,
                                                     scan.window_paths(gs) if scan.one_pass_available(gs) else None))
    T("screen weights, sum bounds, plan", lambda: scan.screen_weights(g, gs, perm, w) if scan.one_pass_available(gs) else None)
    T("scan_topk (sample, main, select, re-score, rows)", lambda: scan.scan_topk(g, w, K, relabel=True))
    T("scan_topk again", lambda: scan.scan_topk(g, w, K, relabel=True))
    tot = sum(t for _, t in acc[:-1])
    print(f"{tag}: {tot:.1f} ms to the first list")
    for l, t in acc: print(f"   {l:52s} {t:8.2f}")
if os.environ.get("WARM") == "1":
    stages(CSRGraph(g0.rowptr, g0.col, None, g0.n_rows, g0.n_cols), "warm-up pass (cold process)")
    stages(CSRGraph(g0.rowptr, g0.col, None, g0.n_rows, g0.n_cols), "fresh graph object, warm process")
else:
    stages(g0, "cold process")
