#!/usr/bin/env python3
"""Whole ppa-like graph, block by block (as the filter stage cuts it: candidates.DEFAULT_BLOCK_PATHS two-hop paths per launch; MAX_PATHS overrides): the candidate list
with Adamic-Adar scores / the list only, through eps_expand_count + eps_expand_fill (expand_score.hip) and through
eps_expand_unit_count + eps_expand_unit_fill (the scan kernel's structure) -- wall time per call incl. the counting pass,
the prefix sum and the one host read of the total; outputs compared bit for bit on every block."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import candidates, ops, scan, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
w = node_weight_table(g, ops.W_AA)
blocks = list(candidates.column_blocks(g, int(os.environ.get("MAX_PATHS", candidates.DEFAULT_BLOCK_PATHS))))
md, sp, mp = scan.max_degree(g), scan.window_splits(g), candidates.max_paths_of(g)
def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = fn(); e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1), r
tot = {"old scored": 0.0, "unit scored": 0.0, "old list": 0.0, "unit list": 0.0}
n_cand = 0
for rep in range(2):
    for k in tot: tot[k] = 0.0
    n_cand = 0
    for lo, hi in blocks:
        order = candidates.heaviest_first(g, lo, hi)
        t, a = timed(lambda: ops.expand_candidates(g.rowptr, g.col, None, w, g.n_rows, lo, hi, want_cn=False, want_v=False, col_order=order, max_paths=mp))
        tot["old scored"] += t
        t, b = timed(lambda: ops.expand_unit(g.rowptr, g.col, w, g.n_rows, lo, hi, md, sp, want_v=False, col_order=order))
        tot["unit scored"] += t
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[4], b[4]), (lo, hi)
        n_cand += a[1].numel()
        del a, b
        t, a = timed(lambda: ops.expand_candidates(g.rowptr, g.col, None, None, g.n_rows, lo, hi, want_cn=False, want_v=False, want_score=False, col_order=order, max_paths=0))
        tot["old list"] += t
        t, b = timed(lambda: ops.expand_unit(g.rowptr, g.col, None, g.n_rows, lo, hi, md, sp, want_score=False, want_v=False, col_order=order))
        tot["unit list"] += t
        assert torch.equal(a[1], b[1])
        del a, b
print(f"{len(blocks)} blocks, {n_cand} candidates, {int(candidates.path_counts(g).sum())} two-hop paths")
for k, v in tot.items():
    print(f"{k:12s} {v:8.1f} ms  {n_cand / v / 1e6:7.1f} G candidates/s")
