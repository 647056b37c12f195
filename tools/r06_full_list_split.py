#!/usr/bin/env python3
"""The full candidate list of the ppa-like bench graph (bench leg `full_list_every_candidate_scored`), taken apart: per block the
counting launch (eps_expand_unit_count) and the fill launch (eps_expand_unit_fill) timed separately with HIP events, scored and
list-only, and the fill launch alone on an UPPER-BOUND layout (segments sized by min(paths, N): no counting launch, the kernel
reports the counts) -- what a one-pass list would cost on the present kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import _lib, candidates, ops, scan, synth
from eps_amd.heuristics import node_weight_table
from eps_amd.ops import _ptr, _stream, _scan_scratch, fixed_weights

dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
w = node_weight_table(g, ops.W_AA)
blocks = list(candidates.column_blocks(g, int(os.environ.get('MAX_PATHS', candidates.DEFAULT_BLOCK_PATHS))))
md, sp = scan.max_degree(g), scan.window_splits(g)
lib = _lib.load()
fixw = fixed_weights(w)
paths = candidates.path_counts(g)
N = g.n_rows


def ev():
    return torch.cuda.Event(enable_timing=True)


def run(scored: bool, ub: bool):
    t_count = t_fill = 0.0
    n_cand = 0
    slots = 0
    for lo, hi in blocks:
        n_cols = hi - lo
        order = candidates.heaviest_first(g, lo, hi)
        ws = _scan_scratch(dev, int(md))
        counts = torch.zeros(n_cols, dtype=torch.int64, device=dev)
        colptr = torch.zeros(n_cols + 1, dtype=torch.int64, device=dev)
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        if ub:
            torch.cumsum(torch.clamp(paths[lo:hi], max=N), 0, out=colptr[1:])
        else:
            _lib.check(lib.eps_expand_unit_count(_ptr(g.rowptr), _ptr(g.col), None, _ptr(sp), N, g.col.numel(), int(md), lo, hi,
                                                 _ptr(order), _ptr(counts), _ptr(ws), ws.numel() * 8, _stream(dev)), "count")
            torch.cumsum(counts, 0, out=colptr[1:])
        e1.record()
        total = int(colptr[-1].item())
        cu = torch.empty(total, dtype=torch.int32, device=dev)
        sc = torch.empty(total, dtype=torch.float32, device=dev) if scored else None
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        e1.record()
        _lib.check(lib.eps_expand_unit_fill(_ptr(g.rowptr), _ptr(g.col), None, _ptr(fixw) if scored else None, _ptr(sp), N, g.col.numel(),
                                            int(md), lo, hi, _ptr(order), _ptr(colptr), _ptr(counts) if ub else None, _ptr(cu), None,
                                            _ptr(sc), _ptr(status), _ptr(ws), ws.numel() * 8, _stream(dev)), "fill")
        e2.record()
        torch.cuda.synchronize()
        assert int(status.item()) == 0
        t_fill += e1.elapsed_time(e2)
        if not ub:
            t_count += e0.elapsed_time(e1)      # (includes the host read + allocation gap: an upper bound of the launch)
        n_cand += int(counts.sum().item()) if ub else total
        slots += total
        del cu, sc
    return t_count, t_fill, n_cand, slots


# the counting launch alone, events tight around it
def count_only():
    t = 0.0
    for lo, hi in blocks:
        order = candidates.heaviest_first(g, lo, hi)
        ws = _scan_scratch(dev, int(md))
        counts = torch.zeros(hi - lo, dtype=torch.int64, device=dev)
        e0, e1 = ev(), ev()
        e0.record()
        _lib.check(lib.eps_expand_unit_count(_ptr(g.rowptr), _ptr(g.col), None, _ptr(sp), N, g.col.numel(), int(md), lo, hi,
                                             _ptr(order), _ptr(counts), _ptr(ws), ws.numel() * 8, _stream(dev)), "count")
        e1.record()
        torch.cuda.synchronize()
        t += e0.elapsed_time(e1)
    return t


import ctypes
wi, nw = ctypes.c_int64(0), ctypes.c_int64(0)
lib.eps_filter_scan_windows(N, ctypes.byref(wi), ctypes.byref(nw))
print(f"{len(blocks)} blocks, N {N}, nnz {g.nnz()}, two-hop paths {int(paths.sum())}; id windows {nw.value} x {wi.value} ids"
      f" (EPS_FS_MIN_WIN={os.environ.get('EPS_FS_MIN_WIN')}, lib {os.path.basename(_lib.LIB_PATH)})")
QUICK = os.environ.get("QUICK") == "1"
if os.environ.get("QUICK") == "ub":
    for rep in range(2):
        _, tf, n_cand, slots = run(True, True)
        print(f"ub layout, scored, {len(blocks)} blocks: fill {tf:7.1f} ms  candidates {n_cand} slots {slots}")
    sys.exit(0)
if QUICK:
    tc = count_only()
    _, tf, n_cand, _ = run(True, False)
    _, tl, _, _ = run(False, False)
    print(f"quick: count {tc:7.1f} ms  scored fill {tf:7.1f} ms  list fill {tl:7.1f} ms  candidates {n_cand}")
    sys.exit(0)
for rep in range(2):
    tc = count_only()
    print(f"rep {rep}: counting launches alone {tc:7.1f} ms")
    for scored in (True, False):
        for ub in (False, True):
            t_count, t_fill, n_cand, slots = run(scored, ub)
            print(f"rep {rep}: {'scored' if scored else 'list  '} {'upper-bound layout' if ub else 'exact layout      '}: fill launches {t_fill:7.1f} ms"
                  f"  candidates {n_cand}  slots {slots}")
