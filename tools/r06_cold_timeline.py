#!/usr/bin/env python3
"""Where the FIRST scan of a fresh graph object goes (what `cold_ms_per_step` of the bench line and a one-shot filter.py pay):
every ops.* call of scan_topk(relabel=True) on a fresh ppa-like graph bracketed by HIP events on its stream, code objects and
allocator warm (a scan of another fresh graph first).  Prints the calls in order with their stream time and the gaps between."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd  # noqa: E401,F401
from eps_amd import ops, scan, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
K = int(os.environ.get("K", 4_000_000))
for name in ("COLUMN_PACK", "HEADS"):
    if name in os.environ:
        setattr(scan, name, os.environ[name] == "1")
g = synth.ppa_like(seed=4, device=dev)
scan.scan_topk(g, node_weight_table(g, ops.W_AA), K, relabel=True)          # warm: code objects, allocator
del g
log = []
depth = [0]
def wrap(name, fn):
    def inner(*a, **kw):
        if depth[0]:
            return fn(*a, **kw)
        depth[0] += 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        try:
            return fn(*a, **kw)
        finally:
            e1.record()
            log.append((name, e0, e1))
            depth[0] -= 1
    return inner
for name in dir(ops):
    fn = getattr(ops, name)
    if isinstance(fn, types.FunctionType) and not name.startswith("_") and name not in ("device_info", "score_bins", "scan_windows", "tail_state"):
        setattr(ops, name, wrap(name, fn))
g = synth.ppa_like(seed=3, device=dev)
w = node_weight_table(g, ops.W_AA)
from eps_amd import candidates
candidates.fused_scores_fit(g, w)
log.clear()
torch.cuda.synchronize()
s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); s0.record()
scan.scan_topk(g, w, K, relabel=True)
s1.record(); torch.cuda.synchronize()
wall = (time.perf_counter() - t0) * 1e3
print(f"first scan of a fresh graph: host wall {wall:.2f} ms, stream {s0.elapsed_time(s1):.2f} ms")
prev = s0
tot = {}
for name, e0, e1 in log:
    gap, ms = prev.elapsed_time(e0), e0.elapsed_time(e1)
    print(f"  {name:28s} {ms:8.3f} ms   (gap before {gap:7.3f})")
    tot[name] = tot.get(name, 0.0) + ms
    prev = e1
print(f"  tail gap {prev.elapsed_time(s1):.3f} ms")
print("by call:", ", ".join(f"{k} {v:.2f}" for k, v in sorted(tot.items(), key=lambda kv: -kv[1])))
t0 = time.perf_counter()
scan.scan_topk(g, w, K, relabel=True); torch.cuda.synchronize()
print(f"second scan {1e3 * (time.perf_counter() - t0):.2f} ms")
t0 = time.perf_counter()
scan.scan_topk(g, w, K, relabel=True); torch.cuda.synchronize()
print(f"third scan {1e3 * (time.perf_counter() - t0):.2f} ms")
