#!/usr/bin/env python3
"""Columns of the bench graph by walked half paths (skipped heads at beta, bar from the sample): how many columns / pieces / paths
sit in columns of <= 256 .. > 65536 walked paths, live columns only (ssum >= bar) -- where the per-piece fixed cost goes."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, eps_amd
from eps_amd import ops, scan, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g0 = synth.ppa_like(seed=3, device=dev)
w = node_weight_table(g0, ops.W_AA)
g, perm = g0.degree_ordered()[:2]
g._cache["scan_calls"] = 2          # (the full-width hub table of a graph that is scanned repeatedly: scan.hub_rows)
sc = scan.screen_weights(g0, g, perm, w)
bar = float(os.environ.get("BAR", 2.876))
scan.HEAD_BETA = float(os.environ.get("BETA", 0.5))
ht = scan.head_tables(g, sc, scan.head_budget(bar * 2.0 ** sc.shift))
wp = ht.wpaths.to(torch.int64).bitwise_and(0xFFFFFFFF).sum(1).cpu().numpy()
pptr = ht.plan[0].cpu().numpy().view(np.uint32).astype(np.int64)
ppc = np.diff(pptr)
ssum = sc.ssum.cpu().numpy().view(np.uint32).astype(np.int64)
live = ssum >= int(bar * 2 ** sc.shift)
deg = (g.rowptr[1:] - g.rowptr[:-1]).cpu().numpy()
edges = [0, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536, 1 << 40]
rows = []
for a, b in zip(edges[:-1], edges[1:]):
    m = live & (wp > a) & (wp <= b)
    rows.append({"paths_in": f"({a}, {b}]", "columns": int(m.sum()), "pieces": int(ppc[m].sum()), "paths": int(wp[m].sum()),
                 "max_degree": int(deg[m].max()) if m.any() else 0, "mean_degree": round(float(deg[m].mean()), 1) if m.any() else 0})
print(json.dumps({"bar": bar, "beta": scan.HEAD_BETA, "live_columns": int(live.sum()), "dead_columns": int((~live).sum()),
                  "pieces_live": int(ppc[live].sum()), "paths_live": int(wp[live].sum()), "empty_live": int((live & (wp == 0)).sum())}))
for r in rows:
    print(json.dumps(r))
