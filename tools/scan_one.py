#!/usr/bin/env python3
"""One main launch of the production scan kernel over the whole ppa-like graph at a fixed bar -- the subject of rocprofv3 PMC
passes.  KERNEL=pieces (eps_scan_screen, default) | twopass (eps_filter_scan, the r02 kernel); VARIANT, PACKED=0 (no packed pieces),
BAR, REPS, RELABEL."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import ops, scan, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g0 = synth.ppa_like(seed=3, device=dev)
w = node_weight_table(g0, ops.W_AA)
g, perm = (g0.degree_ordered()[:2]) if os.environ.get("RELABEL", "1") == "1" else (g0, None)   # hubs first: bench.py's labels
fixw = scan._scan_weights(g0, g, perm, w)
order = scan.column_order(g)
bar = float(os.environ.get("BAR", "2.378"))
kernel = os.environ.get("KERNEL", "pieces")
if kernel == "pieces":
    sc = scan.screen_weights(g0, g, perm, w); fx32, shift, usable = sc.fx32, sc.shift, sc.usable
    bounds, cuts = scan.screen_tables(g)
    variant = int(os.environ.get("VARIANT", str(ops.SCAN_VARIANT)))
    packed = os.environ.get("PACKED", "1") == "1"
for _ in range(int(os.environ.get("REPS", "1"))):
    res = ops.Survivors(64 << 20, bar, dev)
    if kernel == "pieces":
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        ops.scan_screen(g.rowptr, g.col, scan.reverse_positions(g), fx32, cuts, bounds, g.n_rows, order, shift, res, status, variant, wpaths=scan.window_paths(g),
                        ssum=sc.ssum if packed else None, smax=sc.smax if packed else None, plan=sc.plan if packed else None)
    else:
        ops.filter_scan(g.rowptr, g.col, scan.reverse_positions(g), fixw, g.n_rows, order, res, scan.max_degree(g), scan.window_splits(g))
torch.cuda.synchronize()
print(kernel, "slots, unordered candidates:", res.counts(), "half paths:", scan.total_half_paths(g))
