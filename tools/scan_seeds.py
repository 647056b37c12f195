#!/usr/bin/env python3
"""scan_topk on ppa-like graphs of other seeds, AA and RA weights: as labelled vs hubs first must give the same 4 M rows bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, eps_amd
from eps_amd import ops, scan, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
for seed in (4, 5, 6):
    g = synth.ppa_like(seed=seed, device=dev)
    for mode in (ops.W_AA, ops.W_RA):
        w = node_weight_table(g, mode)
        st = {}
        p0, s0 = scan.scan_topk(g, w, 4_000_000, stats=st)
        g2 = synth.ppa_like(seed=seed, device=dev)          # a fresh graph object: relabelled path
        p1, s1 = scan.scan_topk(g2, w, 4_000_000, relabel=True)
        print(seed, "AA" if mode == ops.W_AA else "RA", st["candidates"], (None if st["bar"] is None else round(float(st["bar"]), 4)), st["survivors"], st["launches"], torch.equal(p0, p1) and torch.equal(s0, s1))
