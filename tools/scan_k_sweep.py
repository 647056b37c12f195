import os, sys, time
"""(tools) scan_topk over the whole ppa-like graph for K from 1 to 4e8: launches, bar, survivors, wall time per call."""
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, eps_amd
from eps_amd import ops, scan, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
w = node_weight_table(g, ops.W_AA)
for K in (1, 100, 50_000_000, 400_000_000):
    st = {}
    torch.cuda.synchronize(); t = time.perf_counter()
    p, s = scan.scan_topk(g, w, K, stats=st, relabel=True)
    torch.cuda.synchronize()
    print(K, p.shape, float(s[0]), float(s[-1]), st, f"{(time.perf_counter()-t)*1e3:.1f} ms", bool((s[:-1] >= s[1:]).all()))
