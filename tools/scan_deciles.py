#!/usr/bin/env python3
"""Where eps_filter_scan's time goes by column weight: the heaviest-first order cut into ten groups of equal column
count, each scanned alone (kernel time, half paths, candidates)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import ops, scan, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
w = node_weight_table(g, ops.W_AA)
fixw = scan.fixed_weights(g, w)
order = scan.column_order(g)
hp = scan.half_paths(g)
deg = g.degree()
bar = float(os.environ.get("BAR", "2.14"))
n = order.numel()
tot_t = 0
print("group  columns  half_paths(M)  share  mean_deg  ms   us/column  ns/path")
for d in range(10):
    cols = order[d * n // 10:(d + 1) * n // 10].contiguous()
    p = int(hp[cols.long()].sum())
    ts = []
    for rep in range(3):
        res = ops.Survivors(32 << 20, bar, dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.filter_scan(g.rowptr, g.col, scan.reverse_positions(g), fixw, g.n_rows, cols, res, scan.max_degree(g), scan.window_splits(g)); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    t = min(ts); tot_t += t
    print(f"{d:3d} {cols.numel():8d} {p/1e6:12.1f} {p/int(hp.sum()):6.3f} {float(deg[cols.long()].float().mean()):8.1f} {t:7.2f} "
          f"{t*1e3/cols.numel()*256:8.2f} {t*1e6/max(p,1)*256:7.3f}   (per-CU)")
print("sum of groups", tot_t)
