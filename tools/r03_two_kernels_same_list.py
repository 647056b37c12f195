#!/usr/bin/env python3
"""scan_topk on the bench graph (and a second seed) through the two-pass kernel (scan.ONE_PASS = False) and through the one-pass
kernel under hubs-first labels: the K = 4 M rows and their scores must be identical, bit for bit.  env KS (comma list of K), SEEDS"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd  # noqa: E401,F401
from eps_amd import ops, scan, synth
from eps_amd.graph import CSRGraph
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
for seed in [int(x) for x in os.environ.get('SEEDS', '3,11').split(',')]:
  g0 = synth.ppa_like(seed=seed, device=dev)
  for K in [int(x) for x in os.environ.get('KS', '4000000').split(',')]:
    for name, wk in (("AA", ops.W_AA), ("CN", None), ("RA", ops.W_RA)):
        w = torch.ones(g0.n_rows, dtype=torch.float32, device=dev) if wk is None else node_weight_table(g0, wk)
        scan.ONE_PASS = False
        ga = CSRGraph(g0.rowptr, g0.col, None, g0.n_rows, g0.n_cols)
        pa, sa = scan.scan_topk(ga, w, K)
        assert scan.screen_variant(scan.scan_graph(ga)[0]) is None
        scan.ONE_PASS = True
        gb = CSRGraph(g0.rowptr, g0.col, None, g0.n_rows, g0.n_cols)
        st = {}
        pb, sb = scan.scan_topk(gb, w, K, relabel=True, stats=st)
        for _ in range(2):                           # (the second scan of a graph builds the full-width hub table: time the third)
            scan.scan_topk(gb, w, K, relabel=True)
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        pc, sc3 = scan.scan_topk(gb, w, K, relabel=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        assert torch.equal(pb, pc) and torch.equal(sb, sc3)
        gs, perm = scan.scan_graph(gb)
        sc = scan.screen_weights(gb, gs, perm, w)
        print(f"seed {seed} K {K} {name}: two-pass as labelled vs one-pass hubs-first (variant {scan.screen_variant(gs)}, d_used {sc.d_used}, w_min {sc.w_min:.4f}): "
              f"rows identical {torch.equal(pa, pb)}, scores identical {torch.equal(sa, sb)}, candidates {st['candidates']}, cut {float(sb[-1]):.6f}; "
              f"heads {st.get('heads')} (budget {st.get('head_budget')}), touched {st.get('touched')}, steady step {ms:.2f} ms")
