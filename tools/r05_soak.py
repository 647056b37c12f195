#!/usr/bin/env python3
"""Soak: scan_topk on ONE graph object, 240 calls cycling through AA / CN / RA x K in {1000, 150 k, 4 M, 1.2 M, 40 k} in a shuffled
order (head tables of five bar levels per weight table fight for HEAD_CACHE = 4 slots, the pinned staging ring wraps, tables are
reused across weights) -- every call must return exactly what the first call with its (weights, K) returned."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import ops, scan, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev, n_nodes=int(os.environ.get("NODES", 576289)), n_undirected=int(os.environ.get("EDGES", 21231931)))
W = {"aa": node_weight_table(g, ops.W_AA), "ra": node_weight_table(g, ops.W_RA), "cn": torch.ones(g.n_rows, dtype=torch.float32, device=dev)}
KS = [1000, 150_000, 4_000_000, 1_200_000, 40_000]
jobs = [(k, kind) for k in KS for kind in W] * 16
random.Random(5).shuffle(jobs)
first, bad, t0 = {}, 0, time.perf_counter()
for i, (k, kind) in enumerate(jobs):
    st = {"count": False}
    p, s = scan.scan_topk(g, W[kind], k, stats=st, relabel=True)
    key = (k, kind)
    if key not in first:
        first[key] = (p.clone(), s.clone())
    elif not (torch.equal(p, first[key][0]) and torch.equal(s, first[key][1])):
        bad += 1
        print("MISMATCH at call", i, key, st, flush=True)
torch.cuda.synchronize()
print(f"{len(jobs)} calls, {len(first)} distinct jobs, mismatches {bad}, {time.perf_counter() - t0:.1f} s, "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
