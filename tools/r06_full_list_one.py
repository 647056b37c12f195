#!/usr/bin/env python3
"""ONE one-pass list launch over the whole ppa-like bench graph (eps_expand_unit_list through ops.expand_unit) -- the process
tools/r06_pmc_full_list.sh profiles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import candidates, ops, scan, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
w = node_weight_table(g, ops.W_AA)
md, sp = scan.max_degree(g), scan.window_splits(g)
pre, pre_host = candidates.segment_bounds(g)
r = ops.expand_unit(g.rowptr, g.col, w, g.n_rows, 0, g.n_rows, md, sp, want_v=False, col_order=candidates.heaviest_first(g, 0, g.n_rows),
                    colptr_ub=pre, total_ub=int(pre_host[-1]))
torch.cuda.synchronize()
print("candidates %d two-hop paths %d nnz %d status %d" % (int(r.counts.sum()), int(candidates.path_counts(g).sum()), g.nnz(), int(r.status)))
