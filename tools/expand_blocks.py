#!/usr/bin/env python3
"""Per-launch timing of the fused expansion over the ppa-like graph: is the scoring pass bound by the total number of
atomics or by the heaviest column of each launch?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import candidates, ops, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
w = node_weight_table(g, ops.W_AA)
max_paths = int(os.environ.get("MAX_PATHS", 1 << 27))
blocks = list(candidates.column_blocks(g, max_paths))
deg = g.rowptr[1:] - g.rowptr[:-1]
paths = candidates.path_counts(g)
lpt = os.environ.get("LPT", "1") == "1"
rows = []
def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = fn(); e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1), r
for lo, hi in blocks:
    order = candidates.heaviest_first(g, lo, hi) if lpt else None
    t_list, r = timed(lambda: ops.expand_candidates(g.rowptr, g.col, None, None, g.n_rows, lo, hi, want_cn=False, want_v=True, want_score=False, col_order=order, max_paths=candidates.max_paths_of(g)))
    t_sc, r2 = timed(lambda: ops.expand_candidates(g.rowptr, g.col, None, w, g.n_rows, lo, hi, want_cn=False, want_v=True, want_score=True, col_order=order, max_paths=candidates.max_paths_of(g)))
    p = paths[lo:hi]
    rows.append((lo, hi, r2[1].numel(), int(p.sum()), int(p.max()), int(deg[lo:hi].max()), t_list, t_sc))
    del r, r2
rows.sort(key=lambda r: -r[-1])
print("lo hi cand paths maxcolpaths maxdeg t_list_ms t_score_ms")
for r in rows[:12] + rows[-5:]:
    print(*r[:6], f"{r[6]:.2f} {r[7]:.2f}")
print("total list %.1f ms  score %.1f ms  blocks %d" % (sum(r[6] for r in rows), sum(r[7] for r in rows), len(rows)))
