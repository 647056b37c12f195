#!/usr/bin/env python3
"""A/B of eps_filter_scan builds in ONE process (interleaved, same inputs): python tools/scan_ab.py libA.so libB.so ...
Each library is loaded with ctypes next to the product one; only eps_filter_scan is taken from it."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import ops, scan, synth, _lib
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev, n_nodes=int(os.environ.get('NODES', 576289)), n_undirected=int(os.environ.get('EDGES', 21231931)))
if os.environ.get("RELABEL") == "1":       # hubs first: column v (endpoints u < v) then needs a bitmap of the HEAVIER nodes only
    g = g.degree_ordered()[0]
w = node_weight_table(g, ops.W_AA)
fixw = scan.fixed_weights(g, w)
order = scan.column_order(g)
if os.environ.get("COLMAX"):               # only the columns with id < COLMAX (>= -COLMAX when negative)
    cm = int(os.environ["COLMAX"])
    order = order[order < cm].contiguous() if cm > 0 else order[order >= -cm].contiguous()
    print("columns", order.numel(), "half paths", int(scan.half_paths(g)[order.long()].sum()))
if os.environ.get("HPMAX"):                # only the columns with at most HPMAX half paths (more than -HPMAX when negative)
    hm = int(os.environ["HPMAX"])
    hpv = scan.half_paths(g)[order.long()]
    order = order[hpv <= hm].contiguous() if hm > 0 else order[hpv > -hm].contiguous()
    print("columns", order.numel(), "half paths", int(scan.half_paths(g)[order.long()].sum()))
revpos = scan.reverse_positions(g)
bar = float(os.environ.get("BAR", "2.14"))
libs = []
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.join(ROOT, path) if not os.path.isabs(path) else path)
    lib.eps_filter_scan.restype = ctypes.c_int
    lib.eps_filter_scan.argtypes = _lib.SIGNATURES["eps_filter_scan"][1]
    libs.append((os.path.basename(path), lib))
need = 0
for _, lib in libs:
    lib.eps_filter_scan_workspace_bytes.restype = ctypes.c_int64
    lib.eps_filter_scan_workspace_bytes.argtypes = [ctypes.c_int64]
    need = max(need, lib.eps_filter_scan_workspace_bytes(scan.max_degree(g)))
ws = torch.empty(need // 8 + 1, dtype=torch.int64, device=dev)
_splits = {}
def splits_ptr(lib):
    """each build has its own window geometry: its own split table"""
    lib.eps_filter_scan_windows.argtypes = [ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    w, k = ctypes.c_int64(0), ctypes.c_int64(0)
    assert lib.eps_filter_scan_windows(g.n_rows, ctypes.byref(w), ctypes.byref(k)) == 0
    key = (w.value, k.value)
    if key not in _splits:
        _splits[key] = ops.row_window_splits(g.rowptr, g.col, w.value, k.value)
        print("geometry", key)
    return None if _splits[key] is None else _splits[key].data_ptr()
ref = None
times = {n: [] for n, _ in libs}
for rep in range(int(os.environ.get("REPS", "6"))):
    for name, lib in libs:
        res = ops.Survivors(32 << 20, bar, dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.eps_filter_scan(g.rowptr.data_ptr(), g.col.data_ptr(), revpos.data_ptr(), fixw.data_ptr(), splits_ptr(lib), g.n_rows, g.nnz(),
                                 scan.max_degree(g), order.data_ptr(), order.numel(), res.rec.data_ptr(), ws.data_ptr(), ws.numel() * 8,
                                 torch.cuda.current_stream().cuda_stream)
        e1.record(); torch.cuda.synchronize()
        assert rc == 0, name
        times[name].append(e0.elapsed_time(e1))
        slots, nc = res.counts()
        k, v = res.valid(slots)
        o = torch.argsort(k)
        sig = (k[o], v[o])
        if ref is None:
            ref = sig
        else:
            if os.environ.get('NOCHECK') != '1': assert torch.equal(ref[0], sig[0]) and torch.equal(ref[1], sig[1]), f"{name}: survivors differ"
for name, t in times.items():
    t = sorted(t[1:])
    print(f"{name:40s} median {t[len(t)//2]:.2f} ms  min {t[0]:.2f}  (n={len(t)})")
