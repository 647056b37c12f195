#!/usr/bin/env python3
"""GPU timeline of one production filter step (scan.scan_topk on the ppa-like graph, as bench.py runs it): every ops.* call the step
makes is bracketed by HIP events on its stream -- no synchronisation is added, so the step is the pipelined one -- and the time
between one call's end event and the next call's start event is reported as `gap` (torch plumbing + host).  env: STEPS, K"""
import collections, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import ops, scan, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g0 = synth.ppa_like(seed=3, device=dev, n_nodes=int(os.environ.get("NODES", 576289)), n_undirected=int(os.environ.get("EDGES", 21231931)))
kind = os.environ.get('KIND', 'aa')
w = torch.ones(g0.n_rows, dtype=torch.float32, device=dev) if kind == 'cn' else node_weight_table(g0, ops.W_RA if kind == 'ra' else ops.W_AA)
K = int(os.environ.get("K", 4_000_000))
if 'VBLOCK' in os.environ:
    scan.RESCORE_V_BLOCK = int(os.environ['VBLOCK'])
if 'HEAD_MAX_ROWS' in os.environ:
    scan.HEAD_MAX_ROWS = int(os.environ['HEAD_MAX_ROWS'])
if 'HEAD_KEEP_HI' in os.environ:
    scan.HEAD_KEEP = (scan.HEAD_KEEP[0], float(os.environ['HEAD_KEEP_HI']))
if 'HEAD_BETA' in os.environ:
    scan.HEAD_BETA = float(os.environ['HEAD_BETA'])
if 'TAIL_SORT' in os.environ:
    scan.TAIL_SORT = os.environ['TAIL_SORT']
if 'TAIL_DEVICE' in os.environ:
    scan.TAIL_DEVICE = os.environ['TAIL_DEVICE'] == '1'
if 'COLUMN_PACK' in os.environ:
    scan.COLUMN_PACK = os.environ['COLUMN_PACK'] == '1'
if 'DMAX_MARGIN' in os.environ:
    scan.DMAX_MARGIN = int(os.environ['DMAX_MARGIN'])
steps = int(os.environ.get("STEPS", 10))
for _ in range(3):
    scan.scan_topk(g0, w, K, relabel=True)
log = []
def wrap(mod, name, label=None):
    fn = getattr(mod, name)
    def inner(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **kw)
        e1.record()
        log.append((label or name, e0, e1))
        return r
    setattr(mod, name, inner)
for n in ("scan_screen", "scan_refine", "filter_scan", "kth_largest_dist", "compact_at_least", "rescore_runs", "rescore_weighted", "select_rows",
          "select_rows_pairs", "select_compact", "sort_pairs_by_u", "select_splitters", "compact_range", "score_hist", "score_pick_compact", "radix_sort_by_u",
          "radix_sort_rows", "rescore_runs_dev"):
    if hasattr(ops, n):
        wrap(ops, n)
wrap(ops, "Survivors", "Survivors (list fills)")
wrap(torch, "sort", "torch.sort")
torch.cuda.synchronize()
tot, gaps, wall = collections.OrderedDict(), 0.0, 0.0
for _ in range(steps):
    log.clear()
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); s0.record()
    scan.scan_topk(g0, w, K)
    s1.record(); torch.cuda.synchronize()
    wall += (time.perf_counter() - t0) * 1e3
    prev = s0
    for name, e0, e1 in log:
        gaps += prev.elapsed_time(e0)
        tot[name] = tot.get(name, 0.0) + e0.elapsed_time(e1)
        prev = e1
    gaps += prev.elapsed_time(s1)
st = {"count": False}
scan.scan_topk(g0, w, K, stats=st)
print(f"[{kind}] step (host wall) {wall / steps:.3f} ms; survivors {st.get('survivors')}, walked slots {st.get('walked_slots')}, launches {st.get('launches')}, bar {float(st['bar']) if st.get('bar') is not None else None}; on the stream:")
acc = 0.0
for name, t in tot.items():
    print(f"  {name:28s} {t / steps:7.3f} ms")
    acc += t / steps
print(f"  {'gaps (torch ops, host, syncs)':28s} {gaps / steps:7.3f} ms")
print(f"  {'sum':28s} {acc + gaps / steps:7.3f} ms")
