#!/usr/bin/env python3
"""One main launch of the production scan (skipped heads at HEAD_BETA, hub rows, ticket batching, live columns) over the whole
ppa-like graph at a fixed bar, followed by eps_scan_refine -- the subject of the rocprofv3 PMC passes of r06.  env BAR, REPS, BETA."""
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
g._cache["scan_calls"] = 2          # (the full-width hub table of a graph that is scanned repeatedly: scan.hub_rows)
sc = scan.screen_weights(g0, g, perm, w)
bounds, cuts = scan.screen_tables(g)
bar = float(os.environ.get("BAR", "2.8758351802825928"))
scan.HEAD_BETA = float(os.environ.get("BETA", scan.HEAD_BETA))
ht = scan.head_tables(g, sc, scan.head_budget(bar * 2.0 ** sc.shift), wide=scan.SKETCH_PIECES and scan.SKETCH_WIDE)
cols = scan.live_columns(g, sc, ht, 0, 1)
hub = scan.hub_rows(g)
pack = scan.column_pack(g, sc, ht) if os.environ.get('PACK', '1') == '1' else None
for _ in range(int(os.environ.get("REPS", "1"))):
    wk = ops.Survivors(128 << 20, bar, dev, prefill=False)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.scan_screen(g.rowptr, g.col, scan.reverse_positions(g), sc.fx32, cuts, bounds, g.n_rows, cols, sc.shift, wk, status, scan.screen_variant(g) | (ops.SCAN_SKETCH if scan.SKETCH_PIECES else 0) | (ops.SCAN_WIDE if ht.wide else 0),
                    wpaths=ht.wpaths, ssum=sc.ssum, smax=sc.smax, plan=ht.plan, heads=ht.heads, batch_from=scan.batch_from(g, cols), rowrec=sc.rowrec, colrec=scan.column_records(g, sc, cols, ht.plan, ht.heads, ht.live, 'one'), pack=pack)
    res = ops.Survivors(48 << 20, bar, dev, prefill=False)
    ops.scan_refine(wk, ht.heads, hub, sc.fx32, g.rowptr, g.col, g.n_rows, sc.shift, res)
torch.cuda.synchronize()
walked = int(ht.wpaths.to(torch.int64).bitwise_and(0xFFFFFFFF).sum())
print("walked slots", wk.counts()[0], "survivors", res.counts()[0], "status", int(status), "columns", cols.numel(), "walked half paths", walked,
      "rows skipped", int(ht.heads[:, 0].to(torch.int64).sum()), "pieces", int(ht.plan[1].shape[0]))
