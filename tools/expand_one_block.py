#!/usr/bin/env python3
"""One production-sized launch of the fused expansion (ppa-like graph, block 3, AA scores) -- the subject of PMC passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import candidates, ops, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
w = node_weight_table(g, ops.W_AA)
lo, hi = list(candidates.column_blocks(g))[3]
order = candidates.heaviest_first(g, lo, hi)
for _ in range(int(os.environ.get("REPS", "2"))):
    r = ops.expand_candidates(g.rowptr, g.col, None, w, g.n_rows, lo, hi, want_cn=False, want_score=True, col_order=order, max_paths=candidates.max_paths_of(g))
torch.cuda.synchronize()
print("candidates", r[1].numel(), "paths", int(candidates.path_counts(g)[lo:hi].sum()))
