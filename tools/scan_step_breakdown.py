#!/usr/bin/env python3
"""Where one scan_topk step of the bench spends its wall time outside the main launch (host-timed with synchronisation
between the stages, so the sum is slightly above the pipelined step)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import ops, scan, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g0 = synth.ppa_like(seed=3, device=dev)
w = node_weight_table(g0, ops.W_AA)
K = 4_000_000
for _ in range(3):
    scan.scan_topk(g0, w, K, relabel=True)
g, perm = scan.scan_graph(g0)
fixw = scan._scan_weights(g0, g, perm, w)
order = scan.column_order(g)
def T(fn, n=20):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3, r
t_all, _ = T(lambda: scan.scan_topk(g0, w, K))
t_bar, bar = T(lambda: scan.estimate_bar(g, fixw, K))
cap = min(2 * int(2 * scan.SAFETY * K) + scan._CHUNK_SLACK, (1 << 32) - 1)
t_main, res = T(lambda: scan._launch(g, fixw, order, bar, cap))
t_counts, (slots, nc) = T(lambda: res.counts())
t_valid, (keys, vals) = T(lambda: res.valid(slots))
t_orig, keys2 = T(lambda: scan._original_keys(keys, perm))
t_sel, _ = T(lambda: scan.select_topk(keys2, vals, K))
print(f"step {t_all:.2f} ms = bar estimate {t_bar:.2f} + main launch (incl. Survivors setup) {t_main:.2f} + counts {t_counts:.2f} + valid {t_valid:.2f} "
      f"+ ids back {t_orig:.2f} + select {t_sel:.2f}   (capacity {cap}, slots {slots}, survivors {keys.numel()})")
