#!/usr/bin/env python3
"""One-shot threshold scan (filter.py --keep_top: tables built for ONE scan) on the ppa-like graph: where the wall time goes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import ops, scan, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
K = 4_000_000
def T(name, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print(f"  {name:34s} {(time.perf_counter() - t) * 1e3:8.2f} ms")
    return r
for rep in range(2):
    g = synth.ppa_like(seed=3, device=dev)
    print("pass", rep)
    w = T("node weights (col sums, 1/log)", lambda: node_weight_table(g, ops.W_AA))
    T("reverse positions", lambda: scan.reverse_positions(g))
    T("symmetry check", lambda: scan.is_symmetric(g))
    T("half paths per column", lambda: scan.half_paths(g))
    T("column order (argsort)", lambda: scan.column_order(g))
    T("max degree / window splits", lambda: (scan.max_degree(g), scan.window_splits(g)))
    fixw = T("fixed-point weights", lambda: scan.fixed_weights(g, w))
    T("scan_topk (bar + scan + select)", lambda: scan.scan_topk(g, w, K))
    g2 = synth.ppa_like(seed=3, device=dev)
    T("scan_topk on a fresh graph (all)", lambda: scan.scan_topk(g2, w, K))
