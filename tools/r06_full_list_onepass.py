#!/usr/bin/env python3
"""The full list of the ppa-like bench graph through the ONE-PASS list call (ops.expand_unit with colptr_ub -> eps_expand_unit_list):
blocks of MAX_PATHS two-hop paths (default: the whole graph in one launch), HIP events around everything incl. allocation."""
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
blocks = list(candidates.column_blocks(g, int(os.environ.get("MAX_PATHS", 1 << 40))))
pre, pre_host = candidates.segment_bounds(g)
def all_blocks():
    n = 0
    cnts = []
    for lo, hi in blocks:
        ub = (pre[lo:hi + 1] - pre[lo]).contiguous()
        r = ops.expand_unit(g.rowptr, g.col, w, g.n_rows, lo, hi, md, sp, want_v=False, col_order=candidates.heaviest_first(g, lo, hi),
                            colptr_ub=ub, total_ub=int(pre_host[hi] - pre_host[lo]))
        cnts.append((r.counts.sum(), r.status))
        del r
    for c, st in cnts:
        assert int(st) == 0 or os.environ.get('ABL')
        n += int(c)
    return n
for rep in range(3):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); n = all_blocks(); b.record(); torch.cuda.synchronize()
    print(f"one-pass list, {len(blocks)} block(s): {a.elapsed_time(b):7.1f} ms  candidates {n}  = {n / a.elapsed_time(b) / 1e6:.1f} G candidates/s")
