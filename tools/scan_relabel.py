#!/usr/bin/env python3
"""Main launch of eps_filter_scan on the ppa-like graph as generated vs relabelled by degree (descending = hubs first,
ascending): the symmetric half scheme gives column v the endpoints u < v, so the labelling decides how the 8.35 G half
paths are spread over the columns (as generated: up to 3.7 M per column; hubs first: at most 0.06 M)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import ops, scan, synth
from eps_amd.graph import CSRGraph, _coalesce
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g0 = synth.ppa_like(seed=3, device=dev)
bar = float(os.environ.get("BAR", "3.25"))

def relabel(g, descending):
    perm = torch.argsort(g.degree(), descending=descending, stable=True)
    inv = torch.empty_like(perm); inv[perm] = torch.arange(g.n_rows, device=dev)
    row, col, _ = g.coo()
    rowptr, c, v = _coalesce(inv[row], inv[col], None, g.n_rows, g.n_cols)
    return CSRGraph(rowptr, c, None, g.n_rows, g.n_cols)

for name, g in (("as generated", g0), ("hubs first", relabel(g0, True)), ("hubs last", relabel(g0, False))):
    w = node_weight_table(g, ops.W_AA)
    fixw = scan.fixed_weights(g, w)
    order = scan.column_order(g)
    hp = scan.half_paths(g)
    rp, sp, md = scan.reverse_positions(g), scan.window_splits(g), scan.max_degree(g)
    ts = []
    for _ in range(4):
        res = ops.Survivors(64 << 20, bar, dev)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.filter_scan(g.rowptr, g.col, rp, fixw, g.n_rows, order, res, md, sp)
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    slots, cands = res.counts()
    k, v = res.valid(slots)
    print(f"{name:14s} launch ms {['%.2f' % t for t in ts]}  half paths {int(hp.sum())} max/column {int(hp.max())}  candidates {cands} survivors {k.numel()} score sum {float(v.double().sum()):.4f}")
