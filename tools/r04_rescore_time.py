#!/usr/bin/env python3
"""eps_rescore_runs alone on the production step's job: the survivors of the main launch at the bench's bar that pass the pre-filter
(~2.2 M pairs), sorted as scan.rescore_exact sorts them; mean HIP-event ms of 5 calls.  For A/Bs of builds through EPS_LIB_PATH."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import ops, scan, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g0 = synth.ppa_like(seed=3, device=dev)
w = node_weight_table(g0, ops.W_AA)
g, perm = g0.degree_ordered()[:2]
sc = scan.screen_weights(g0, g, perm, w)
bounds, cuts = scan.screen_tables(g)
res = ops.Survivors(48 << 20, float(os.environ.get("BAR", "2.378")), dev, both=True, prefill=False)
status = torch.zeros(1, dtype=torch.int32, device=dev)
ops.scan_screen(g.rowptr, g.col, scan.reverse_positions(g), sc.fx32, cuts, bounds, g.n_rows, scan.column_order(g), sc.shift, res, status,
                scan.screen_variant(g), wpaths=scan.window_paths(g), ssum=sc.ssum, smax=sc.smax, plan=sc.plan)
a, b = sc.lower_params(scan.max_degree(g))
ck, _, n, _, _ = ops.select_compact(res.key, res.val, 2_000_000, res.count_ptr, mode=2, params=(a, b, 4e-6))
keys = ck[:int(n.item())]
by_u = ops.sort_pairs_by_u(keys, 20, int(os.environ.get("VBLOCK", scan.RESCORE_V_BLOCK)))
ops.rescore_runs(g.rowptr, g.col, sc.fixw, g.n_rows, by_u)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ops.rescore_runs(g.rowptr, g.col, sc.fixw, g.n_rows, by_u)
e1.record(); torch.cuda.synchronize()
print(f"{os.environ.get('EPS_LIB_PATH', 'in-tree')}: {by_u.numel()} pairs, eps_rescore_runs {e0.elapsed_time(e1) / 5:.3f} ms")
