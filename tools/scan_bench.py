#!/usr/bin/env python3
"""Full ppa-like filter through the threshold scan (scan.scan_topk) vs the r01 streaming path: time + identical top-K."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import candidates, ops, proposals, scan, synth, filter_stage
from eps_amd.heuristics import node_weight_table
ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, default=4_000_000)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--old", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
w = node_weight_table(g, ops.W_AA)
torch.cuda.synchronize(); t0 = time.perf_counter()
scan.reverse_positions(g); scan.column_order(g); fixw = scan.fixed_weights(g, w)
torch.cuda.synchronize(); print(f"per-graph tables: {time.perf_counter() - t0:.3f} s; half paths {int(scan.half_paths(g).sum())}")
def ev():
    return torch.cuda.Event(enable_timing=True)
for rep in range(a.reps):
    st = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pairs, scores = scan.scan_topk(g, w, a.k, stats=st)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"scan_topk: {dt*1e3:.1f} ms  {st}  -> {st['candidates']/dt/1e9:.1f} G cand/s", flush=True)
# kernel-only timing of the main launch at the final bar
bar = float(st["bar"]) if st["bar"] is not None else float("-inf")      # (r03: the bar stays on the device)
order = scan.column_order(g)
for rep in range(3):
    res = ops.Survivors(64 << 20, bar, dev)
    e0, e1 = ev(), ev()
    e0.record(); ops.filter_scan(g.rowptr, g.col, scan.reverse_positions(g), fixw, g.n_rows, order, res, scan.max_degree(g), scan.window_splits(g)); e1.record()
    torch.cuda.synchronize()
    print(f"main launch at bar {bar:.4f}: {e0.elapsed_time(e1):.2f} ms, slots/cands {res.counts()}")
e0, e1 = ev(), ev(); e0.record(); b = scan.estimate_bar(g, fixw, a.k); e1.record(); torch.cuda.synchronize()
print(f"estimate_bar: {e0.elapsed_time(e1):.2f} ms -> {float(b)}")
if a.old:
    class A: pass
    args = A(); args.model = "adamic_ogb"
    class D: pass
    data = D(); data.adj_t = g; data.x = None
    for rep in range(2):
        top = proposals.StreamingTopK(a.k)
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
        with torch.no_grad():
            for v_lo, v_hi, blk, sc in filter_stage.scored_blocks(args, None, data, None, 0, None, bar=top.bar):
                top.push(blk, sc); n += blk.numel()
        op, os_ = top.result()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"r01 streaming path: {dt*1e3:.1f} ms, {n} candidates")
    print("identical pairs:", bool(torch.equal(op, pairs)), " identical scores:", bool(torch.equal(os_, scores)))
