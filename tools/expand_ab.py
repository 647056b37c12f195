#!/usr/bin/env python3
"""Timing: fused expansion vs (tensor-op candidate block + column-run pair kernel) on the bench columns."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import candidates, ops, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
w = node_weight_table(g, ops.W_AA)
lo, hi = 0, 1536
def t(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
ms, r = t(lambda: ops.expand_candidates(g.rowptr, g.col, None, w, g.n_rows, lo, hi))
E = r[1].numel()
print(f"fused expand+score columns [{lo},{hi}): {E} candidates in {ms:.2f} ms -> {E / ms / 1e6:.2f} G cand/s")
ms2, _ = t(lambda: ops.expand_candidates(g.rowptr, g.col, None, None, g.n_rows, lo, hi, want_cn=False, want_score=False))
print(f"fused expand only (candidate list): {ms2:.2f} ms")
def old():
    p = torch.cat([candidates.two_hop_block(g, v, min(v + 128, hi)) for v in range(lo, hi, 128)], 1)
    return p
ms3, p = t(old, 1)
print(f"tensor-op candidate generation: {p.shape[1]} candidates in {ms3:.2f} ms")
lo2, hi2 = 0, 65536
ms4, r4 = t(lambda: ops.expand_candidates(g.rowptr, g.col, None, w, g.n_rows, lo2, hi2), 1)
print(f"fused, columns [0,65536): {r4[1].numel()} candidates in {ms4:.2f} ms -> {r4[1].numel() / ms4 / 1e6:.2f} G cand/s")
