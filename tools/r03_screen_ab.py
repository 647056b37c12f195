#!/usr/bin/env python3
"""A/B in one process: eps_filter_scan (r02 kernel) vs eps_scan_screen (one-pass piece kernel, every geometry variant) on the
ppa-like graph at a fixed bar -- kernel times (HIP events), candidate counts, and the survivor lists after exact re-scoring,
which must be identical.  env: NODES / EDGES (graph size), BAR, REPS, RELABEL=0 (scan the graph as labelled), VARIANTS=0,1,2"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import ops, scan, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g0 = synth.ppa_like(seed=3, device=dev, n_nodes=int(os.environ.get("NODES", 576289)), n_undirected=int(os.environ.get("EDGES", 21231931)))
w = node_weight_table(g0, ops.W_AA)
relabel = os.environ.get("RELABEL", "1") == "1"
g, perm = (g0.degree_ordered()[:2]) if relabel else (g0, None)
fixw = scan._scan_weights(g0, g, perm, w)
order = scan.column_order(g)
bar = float(os.environ.get("BAR", "2.378"))
reps = int(os.environ.get("REPS", "4"))
cap = 48 << 20

def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1), r

def sorted_list(keys, vals):
    o = torch.argsort(keys)
    return keys[o], vals[o]

# reference: the two-pass kernel
ts = []
for _ in range(reps):
    res = ops.Survivors(cap, bar, dev)
    t, _ = timed(lambda: ops.filter_scan(g.rowptr, g.col, scan.reverse_positions(g), fixw, g.n_rows, order, res, scan.max_degree(g), scan.window_splits(g)))
    ts.append(t)
slots, ncand = res.counts()
rk, rv = sorted_list(*res.valid(slots))
print(f"eps_filter_scan: {min(ts):.2f} ms (min of {reps}), candidates {ncand}, survivors {rk.numel()}, slots {slots}")

t_tab, _ = timed(lambda: scan.screen_tables(g))
sc = scan.screen_weights(g0, g, perm, w); fx32, shift, usable = sc.fx32, sc.shift, sc.usable
bounds, cuts = scan.screen_tables(g)
print(f"screen tables built in {t_tab:.2f} ms; shift {shift}, usable {usable}, bounds {bounds.tolist()}")
for variant, packed, table in [(int(x.rstrip("pt")), "p" in x, "t" in x) for x in os.environ.get("VARIANTS", "0,1,2,2p,2pt").split(",")]:
    plan = None
    if table:      # suffix p: packed / 16-bit direct pieces (sum bounds), t: the per-graph plan table
        plan = ops.scan_plan(g.rowptr, cuts, scan.window_paths(g), sc.ssum if packed else None, sc.smax if packed else None, bounds, g.n_rows, shift, variant)
    ts = []
    for _ in range(reps):
        res = ops.Survivors(cap, bar, dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        t, _ = timed(lambda: ops.scan_screen(g.rowptr, g.col, scan.reverse_positions(g), fx32, cuts, bounds, g.n_rows, order, shift, res, status, variant, wpaths=scan.window_paths(g),
                                               ssum=sc.ssum if packed else None, smax=sc.smax if packed else None, plan=plan))
        ts.append(t)
    slots, nc2 = res.counts()
    keys, vals = res.valid(slots)
    t_re, (k2, v2) = timed(lambda: scan.rescore_exact(g, sc, keys, torch.tensor([bar], device=dev)))
    m = k2 >= 0
    nk, nv = sorted_list(k2[m], v2[m])
    same = nk.numel() == rk.numel() and torch.equal(nk, rk) and torch.equal(nv, rv)
    print(f"eps_scan_screen variant {variant}{' packed' if packed else ''}{' + plan table' if table else ''}: {min(ts):.2f} ms (min of {reps}; all {[round(x, 2) for x in ts]}), candidates {nc2} "
          f"({'==' if nc2 == ncand else '!='}), screened {keys.numel()} -> exact {nk.numel()}, slots {slots}, status {int(status)}, "
          f"re-scoring {t_re:.2f} ms, identical to eps_filter_scan: {same}")
    if not same and nk.numel() and rk.numel():
        a, b = set(nk.tolist()), set(rk.tolist())
        print("   missing", len(b - a), "extra", len(a - b), "first missing", [(x & 0xFFFFFFFF, x >> 32) for x in list(b - a)[:5]])
