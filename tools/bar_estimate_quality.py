#!/usr/bin/env python3
"""How good is the bar estimated from every n-th column?  For strides 256 / 512 / 1024 and several sample offsets: the
estimated bar and the number of directed candidates of the WHOLE graph above it, against the target SAFETY x K."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import ops, scan, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g0 = synth.ppa_like(seed=int(os.environ.get("SEED", 3)), device=dev)
w = node_weight_table(g0, ops.W_AA)
K = 4_000_000
for relabel in (True, False):
    g, perm = scan.scan_graph(g0, build=True) if relabel else (g0, None)
    fixw = scan._scan_weights(g0, g, perm, w)
    order = scan.column_order(g)
    # all unordered scores above a low bar once: counts above any estimate come from this list
    res = scan._launch(g, fixw, order, 1.7, 400 << 20)
    slots, _ = res.counts()
    _, vals = res.valid(slots)
    vals = torch.sort(vals).values
    hp = scan.half_paths(g)
    for stride in (256, 512, 1024, 2048):
        outs = []
        for off in (stride // 2, stride // 4, 3 * stride // 4, 1, stride - 1):
            sample = order[off::stride].contiguous()
            bound = int(hp[sample.long()].sum().item())
            r = scan._launch(g, fixw, sample, float("-inf"), 2 * bound + scan._CHUNK_SLACK)
            s, _ = r.counts()
            _, v = r.valid(s)
            m = int(scan.SAFETY * K / 2 / stride) + 1
            bar = float(ops.kth_largest(v, m))
            above = 2 * int(vals.numel() - torch.searchsorted(vals, torch.tensor([bar], device=dev), right=True))
            outs.append(f"{bar:.4f}:{above / (scan.SAFETY * K):.3f}")
        print(("hubs first " if relabel else "as labelled"), "stride", stride, " bar:survivors/target ", "  ".join(outs))
