#!/usr/bin/env python3
"""Where the production plan's packed pieces sit: columns of one round (<= 256 rows: sketch pieces) vs columns of several."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import ops, scan, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g0 = synth.ppa_like(seed=3, device=dev)
w = node_weight_table(g0, ops.W_AA)
st = {}
scan.scan_topk(g0, w, 4_000_000, stats=st, relabel=True)
scan.scan_topk(g0, w, 4_000_000, stats=st, relabel=True)
g, perm = scan.scan_graph(g0)
sc = scan.screen_weights(g0, g, perm, w)
ht = sc.head_cur
plan = ht.plan
info = plan[1][:, 0].view(torch.int32).to(torch.int64) & 0xFFFFFFFF
kinds, paths = info >> 30, info & 0x3FFFFFFF
pptr = plan[0].view(torch.int32).to(torch.int64) & 0xFFFFFFFF
col_of = torch.searchsorted(pptr[1:].contiguous(), torch.arange(info.numel(), device=dev), right=True)
deg = g.degree()[col_of]
live = torch.zeros(g.n_rows, dtype=torch.bool, device=dev)
live[scan.live_columns(g, sc, ht, 0, 1).long()] = True
lv = live[col_of]
for name, m in (("packed, one round", (kinds == 1) & (deg <= 256) & lv), ("packed, several rounds", (kinds == 1) & (deg > 256) & lv),
                ("direct16", (kinds == 3) & lv), ("direct32", (kinds == 2) & lv)):
    print(f"{name:24s} pieces {int(m.sum()):8d}  paths {int(paths[m].sum()):12d}")
