#!/usr/bin/env python3
"""Where the full-scale filter stage spends its time: expansion (count/fill) vs streaming top-K."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import candidates, ops, proposals, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
w = node_weight_table(g, ops.W_AA)
blocks = list(candidates.column_blocks(g))
print("blocks", len(blocks))
def run(do_topk, want_score=True):
    top = proposals.StreamingTopK(4_000_000)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    for lo, hi in blocks:
        pairs, _, sc = candidates.expand_block(g, lo, hi, w if want_score else None, want_score=want_score)
        n += pairs.shape[1]
        if do_topk: top.push(pairs, sc)
    torch.cuda.synchronize(); return time.perf_counter() - t0, n
for name, args in {"expand+score": (False, True), "expand+score+topk": (True, True), "expand only": (False, False)}.items():
    dt, n = run(*args)
    print(f"{name:20s} {dt:.3f} s  {n / dt / 1e9:.2f} G cand/s")
# raw kernels without the python-side stack/long conversions
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
for lo, hi in blocks:
    r = ops.expand_candidates(g.rowptr, g.col, None, w, g.n_rows, lo, hi, want_cn=False, want_v=True, max_paths=candidates.max_paths_of(g))
    n += r[1].numel()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{'ops.expand (u,v,score)':20s} {dt:.3f} s  {n / dt / 1e9:.2f} G cand/s")
