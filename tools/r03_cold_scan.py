#!/usr/bin/env python3
"""What the FIRST scan of a fresh graph object costs (a one-shot `filter.py --keep_top K`), scanned as labelled vs under
hubs-first labels (scan_topk(relabel=True): the relabelled copy, its tables, the one-pass kernel), on fresh ppa-like graphs.
The process's one-off code-object loads are taken out with tiny scans first."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd  # noqa: E401,F401
from eps_amd import candidates, ops, scan, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
K = 4_000_000
sync = torch.cuda.synchronize
g_tiny = synth.rmat_graph(10, 8, 1, dev)
wt = node_weight_table(g_tiny, ops.W_AA)
scan.scan_topk(g_tiny, wt, 1000)
scan.scan_topk(g_tiny.degree_ordered()[0], wt[g_tiny.degree_ordered()[1]].contiguous(), 1000)
if os.environ.get("WARM_BIG", "1") == "1":
    big = synth.rmat_graph(17, 8, 2, dev)             # above RELABEL_MIN_NODES: the relabelled flow's code objects too
    scan.scan_topk(big, node_weight_table(big, ops.W_AA), 100000, relabel=True)
    del big
ref = None
for seed in (3, 4):
    for relabel in (False, True, False, True):
        g = synth.ppa_like(seed=seed, device=dev)
        w = node_weight_table(g, ops.W_AA)
        candidates.fused_scores_fit(g, w)
        r0 = torch.cuda.memory_reserved()
        sync(); t0 = time.perf_counter()
        if relabel:
            g.degree_ordered()
            sync()
        t1 = time.perf_counter()
        p, s = scan.scan_topk(g, w, K, relabel=relabel)
        sync(); t2 = time.perf_counter()
        if ref is None or ref[0] != seed:
            ref = (seed, p.clone(), s.clone())
        same = torch.equal(ref[1], p) and torch.equal(ref[2], s)
        print(f"seed {seed} relabel={relabel}: first scan {1e3*(t2-t0):.1f} ms (degree_ordered {1e3*(t1-t0):.1f}), "
              f"same list as the first run of this seed: {same}; allocator reserved +{(torch.cuda.memory_reserved()-r0)>>20} MiB during it")
        sync(); t0 = time.perf_counter()
        scan.scan_topk(g, w, K, relabel=relabel)
        sync()
        print(f"        second scan of the same graph {1e3*(time.perf_counter()-t0):.1f} ms")
        del g, w
