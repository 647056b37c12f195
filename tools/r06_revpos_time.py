#!/usr/bin/env python3
"""Reverse positions of the hubs-first copy of the ppa-like bench graph: searches (eps_reverse_positions_symmetric) vs the sort
(eps_reverse_positions_sorted), HIP events, same outputs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import ops, synth
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
gs = g.degree_ordered()[0]
res = {}
for name, g_ in (("as labelled", g), ("hubs first", gs)):
    for how, thr in (("searches", 1 << 62), ("sorted", 0)):
        ops.REVPOS_SORT_MIN = thr
        best = 1e9
        for rep in range(5):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); r = ops.reverse_positions_symmetric(g_.rowptr, g_.col); b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        res[(name, how)] = r
        print(f"{name:12s} {how:9s} {best:7.3f} ms  flag {int(r[2][0]) & 0xFFFFFFFF} stats {r[2][1:].tolist()}")
    a, b = res[(name, "searches")], res[(name, "sorted")]
    print("  identical:", torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]))
