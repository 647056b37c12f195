#!/usr/bin/env python3
"""Why 100 timed steps of bench.py average 14.6 ms and the sustained loop after them 14.2: the same step in loops of 20 / 100 /
100 steps, with and without HIP events around the scan launches, per-loop averages and the slowest steps."""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import ops, scan, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
w = node_weight_table(g, ops.W_AA)
for _ in range(4):
    scan.scan_topk(g, w, 4_000_000, relabel=True)
torch.cuda.synchronize()


def loop(n, events):
    ops.KERNEL_EVENTS, ops.EVENT_NAMES = ([], ()) if events else (None, None)
    ts = []
    m0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
    f0 = torch.cuda.memory_stats().get("num_device_free", 0)
    g0 = [s["collections"] for s in gc.get_stats()]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        a = time.perf_counter()
        scan.scan_topk(g, w, 4_000_000, relabel=True)
        ts.append((time.perf_counter() - a) * 1e3)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    ops.KERNEL_EVENTS, ops.EVENT_NAMES = None, None
    s = sorted(ts)
    print(f"{n:4d} steps, events {events}: {dt:.3f} ms per step; host-side per call median {s[len(s) // 2]:.3f}, max {s[-1]:.3f}, >16 ms: {sum(1 for x in ts if x > 16)}; hipMalloc {torch.cuda.memory_stats().get('num_device_alloc', 0) - m0}, hipFree {torch.cuda.memory_stats().get('num_device_free', 0) - f0}, gc runs {[s['collections'] - a for s, a in zip(gc.get_stats(), g0)]}, slow at {[i for i, x in enumerate(ts) if x > 16][:12]}", flush=True)


for n, ev in ((100, False), (100, True), (100, False), (100, True)):
    loop(n, ev)
gc.disable()
print("gc disabled")
for n, ev in ((100, False), (100, True), (100, False), (100, True)):
    loop(n, ev)
