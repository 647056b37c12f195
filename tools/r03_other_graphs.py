#!/usr/bin/env python3
"""The threshold scan (AA, K = 1 M) on graph families other than the bench's: repeat-scan time as labelled (two-pass kernel) vs
under hubs-first labels (one-pass kernel), geometry chosen, list equality."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd  # noqa: E401,F401
from eps_amd import ops, scan, synth
from eps_amd.graph import CSRGraph
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
K = 1_000_000
sync = torch.cuda.synchronize
cases = [("rmat 2^18 x 32", lambda: synth.rmat_graph(18, 32, 1, dev)), ("rmat 2^20 x 16", lambda: synth.rmat_graph(20, 16, 2, dev)),
         ("rmat 2^21 x 8", lambda: synth.rmat_graph(21, 8, 3, dev)), ("rmat 2^22 x 4", lambda: synth.rmat_graph(22, 4, 4, dev)),
         ("ppa-like half size", lambda: synth.ppa_like(seed=5, device=dev, n_nodes=288144, n_undirected=10615965))]
for name, make in cases:
    g0 = make()
    w = node_weight_table(g0, ops.W_AA)
    res = {}
    for relabel in (False, True):
        g = CSRGraph(g0.rowptr, g0.col, None, g0.n_rows, g0.n_cols)
        st = {}
        p, s = scan.scan_topk(g, w, K, stats=st, relabel=relabel)
        scan.scan_topk(g, w, K, relabel=relabel)
        sync(); t = time.perf_counter()
        for _ in range(5):
            scan.scan_topk(g, w, K, relabel=relabel)
        sync()
        gs, perm = scan.scan_graph(g)
        res[relabel] = (p, s, (time.perf_counter() - t) / 5 * 1e3, scan.screen_variant(gs), st["candidates"], int(scan.half_paths(gs).max()), scan.total_half_paths(gs))
    a, b = res[False], res[True]
    print(f"{name}: N {g0.n_rows}, nnz {g0.nnz()}, {b[6] / 1e9:.2f} G half paths, {a[4] / 1e9:.2f} G unordered candidates; as labelled {a[2]:.1f} ms "
          f"(heaviest column {a[5]}, one-pass variant {a[3]}), hubs first {b[2]:.1f} ms (heaviest column {b[5]}, variant {b[3]}); "
          f"same list {torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])}, same count {a[4] == b[4]}; "
          f"{2 * b[4] / b[2] / 1e6:.0f} G directed candidates/s")
