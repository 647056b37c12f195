#!/usr/bin/env python3
"""What the per-graph tables of the one-pass scan cost on a FRESH graph object (bench.py's prep_ms, stage by stage; host-timed with
a synchronisation after every stage; second pass = allocator warm).  env: NODES, EDGES"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import candidates, ops, scan, synth
from eps_amd.graph import CSRGraph
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g0 = synth.ppa_like(seed=3, device=dev, n_nodes=int(os.environ.get("NODES", 576289)), n_undirected=int(os.environ.get("EDGES", 21231931)))
w0 = node_weight_table(g0, ops.W_AA)
K = 4_000_000
def T(label, fn, acc):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    acc.append((label, (time.perf_counter() - t) * 1e3)); return r
for it in range(3):
    g = CSRGraph(g0.rowptr, g0.col, None, g0.n_rows, g0.n_cols); w = w0.clone(); acc = []
    gs, perm = T("hubs-first copy (scan_graph)", lambda: scan.scan_graph(g, build=True), acc)
    T("revpos + half paths + symmetry", lambda: scan.is_symmetric(gs), acc)
    T("column order (argsort)", lambda: scan.column_order(gs), acc)
    T("fixed-point weights (relabelled)", lambda: scan._scan_weights(g, gs, perm, w), acc)
    T("max degree", lambda: scan.max_degree(gs), acc)
    T("screen_variant", lambda: scan.screen_variant(gs), acc)
    T("bounds + cuts", lambda: scan.screen_tables(gs), acc)
    T("window paths", lambda: scan.window_paths(gs), acc)
    T("score bound (fused_score_bound)", lambda: candidates.fused_score_bound(g, w), acc)
    T("screen weights, sum bounds, plan", lambda: scan.screen_weights(g, gs, perm, w), acc)
    T("shard + sample columns", lambda: (scan.shard_columns(gs, 0, 1), scan.sample_columns(gs, scan.sample_stride(K), 0, 1)), acc)
    T("total half paths", lambda: scan.total_half_paths(gs), acc)
    tot = sum(t for _, t in acc)
    if it:
        print(f"pass {it}: total {tot:.2f} ms")
        for l, t in acc: print(f"   {l:36s} {t:7.3f}")
    T("first scan on the prepared graph", lambda: scan.scan_topk(g, w, K), acc)
    T("second scan", lambda: scan.scan_topk(g, w, K), acc)
    if it: print(f"   first scan {acc[-2][1]:.2f}, second {acc[-1][1]:.2f}")
