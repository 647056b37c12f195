#!/usr/bin/env python3
"""One main launch of eps_filter_scan over the whole ppa-like graph at a fixed bar -- the subject of rocprofv3 PMC passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import ops, scan, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
if os.environ.get("RELABEL", "1") == "1":
    g = g.degree_ordered()[0]               # hubs first: the labels bench.py's repeatedly scanned graph runs under
w = node_weight_table(g, ops.W_AA)
fixw = scan.fixed_weights(g, w)
order = scan.column_order(g)
bar = float(os.environ.get("BAR", "3.25"))
for _ in range(int(os.environ.get("REPS", "1"))):
    res = ops.Survivors(64 << 20, bar, dev)
    ops.filter_scan(g.rowptr, g.col, scan.reverse_positions(g), fixw, g.n_rows, order, res, scan.max_degree(g), scan.window_splits(g))
torch.cuda.synchronize()
print("slots, unordered candidates:", res.counts(), "half paths:", int(scan.half_paths(g).sum()))
