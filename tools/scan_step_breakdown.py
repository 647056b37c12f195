#!/usr/bin/env python3
"""Where one scan_topk step of the bench spends its wall time (r03 flow: one-pass screening kernel + exact re-scoring),
host-timed with synchronisation between the stages, so the sum is slightly above the pipelined step."""
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
sc = scan.screen_weights(g0, g, perm, w); fx32, shift, usable = sc.fx32, sc.shift, sc.usable
screen = sc if usable and scan.one_pass_available(g) else None
order = scan.column_order(g)
def T(fn, n=20):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3, r
t_all, _ = T(lambda: scan.scan_topk(g0, w, K))
t_bar, bar = T(lambda: scan.estimate_bar(g, fixw, K, screen=screen))
slack = scan._CHUNK_SLACK if screen is None else scan._PIECE_SLACK
cap = 2 * int(2 * scan.SAFETY * K) + slack
t_main, res = T(lambda: scan._launch(g, fixw, order, bar, cap, both=True, screen=screen))
t_comp, (ck, cv, nv) = T(lambda: ops.compact_at_least(res.key, res.val, None))
nv = int(nv.item())
t_sort, by_u = T(lambda: torch.sort(((ck[:nv] & 0xFFFFFFFF) << 32) | (ck[:nv] >> 32)).values)
t_resc, vals = T(lambda: ops.rescore_runs(g.rowptr, g.col, sc.fixw, g.n_rows, by_u))
t_re_all, (lk, lv) = T(lambda: scan.rescore_exact(g, sc, ck[:nv], bar))
k2 = (K + 1) // 2
t_kth, cut = T(lambda: ops.kth_largest_dist(lv, k2, 1))
t_cut, (sk, sv, ns) = T(lambda: ops.compact_at_least(lk, lv, cut))
ns = int(ns.item())
t_orig, keys2 = T(lambda: scan._original_keys(sk[:ns], perm))
t_rows, _ = T(lambda: ops.select_rows(keys2.contiguous(), sv[:ns].contiguous(), K, 20))
print(f"step {t_all:.2f} ms = bar estimate {t_bar:.2f} + main launch (incl. Survivors fills) {t_main:.2f} + compaction {t_comp:.2f} + "
      f"[re-scoring {t_re_all:.2f}: of which sort by u {t_sort:.2f}, eps_rescore_runs {t_resc:.2f}] + k-th {t_kth:.2f} + cut compaction {t_cut:.2f} "
      f"+ ids back {t_orig:.2f} + rows (mirror + sorts) {t_rows:.2f}   (capacity {cap}, screened {nv}, selected {ns}; one-pass kernel: {screen is not None})")
