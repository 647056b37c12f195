#!/usr/bin/env python3
"""Per-phase cycle shares of the LIST kernels (filter_scan_kernel<FS_COUNT> / <FS_EMIT>, eps_expand_unit_*) over the whole ppa-like
graph as labelled (diagnostic build: tools/make_scan_stamps.py): where the full-list leg's time goes."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["EPS_LIB_PATH"] = os.path.join(ROOT, "tools", "libeps_fsstamp.so")
import torch, eps_amd
from eps_amd import _lib, candidates, ops, scan, synth
from eps_amd.heuristics import node_weight_table
from eps_amd.ops import _ptr, _stream, _scan_scratch, fixed_weights
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
w = node_weight_table(g, ops.W_AA)
fixw = fixed_weights(w)
lib = _lib.load()
lib.eps_debug_scan_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
names = ["hand-out + column setup", "A rounds: describe + mark + histogram", "known edges out + rank tables (B) + the u list", "plan",
         "D total (windows) incl. scans", "  of which D2 / tile scans", "  direct-mode windows (D1+scan)", "  record-mode windows (D1+D2)", "    D2: wait for the records (barrier after D1)",
         "    D2 per tile: request + accumulate", "    D2 per tile: barrier (sums complete)", "    D2 per tile: scan (+ score stores)",
         "    D2 per tile: barrier (zeroed)", "(count) tiles in record mode", "(count) columns with candidates", "(count) single-round columns"]
md, sp, N = scan.max_degree(g), scan.window_splits(g), g.n_rows
blocks = list(candidates.column_blocks(g))
acc = {"count": [0] * 16, "fill scored": [0] * 16, "fill list": [0] * 16}
ms = {"count": 0.0, "fill scored": 0.0, "fill list": 0.0}
def ev(): return torch.cuda.Event(enable_timing=True)
for lo, hi in blocks:
    order = candidates.heaviest_first(g, lo, hi)
    ws = _scan_scratch(dev, int(md))
    counts = torch.zeros(hi - lo, dtype=torch.int64, device=dev)
    colptr = torch.zeros(hi - lo + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize(); lib.eps_debug_scan_stamps(buf, 1)
    a, b = ev(), ev(); a.record()
    _lib.check(lib.eps_expand_unit_count(_ptr(g.rowptr), _ptr(g.col), None, _ptr(sp), N, g.col.numel(), int(md), lo, hi,
                                         _ptr(order), _ptr(counts), _ptr(ws), ws.numel() * 8, _stream(dev)), "count")
    b.record(); torch.cuda.synchronize(); ms["count"] += a.elapsed_time(b)
    lib.eps_debug_scan_stamps(buf, 1)
    for i in range(16): acc["count"][i] += buf[i]
    torch.cumsum(counts, 0, out=colptr[1:])
    total = int(colptr[-1].item())
    cu = torch.empty(total, dtype=torch.int32, device=dev)
    sc = torch.empty(total, dtype=torch.float32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    for kind, scored in (("fill scored", True), ("fill list", False)):
        torch.cuda.synchronize()
        a, b = ev(), ev(); a.record()
        _lib.check(lib.eps_expand_unit_fill(_ptr(g.rowptr), _ptr(g.col), None, _ptr(fixw) if scored else None, _ptr(sp), N, g.col.numel(),
                                            int(md), lo, hi, _ptr(order), _ptr(colptr), None, _ptr(cu), None,
                                            _ptr(sc) if scored else None, _ptr(status), _ptr(ws), ws.numel() * 8, _stream(dev)), "fill")
        b.record(); torch.cuda.synchronize(); ms[kind] += a.elapsed_time(b)
        lib.eps_debug_scan_stamps(buf, 1)
        for i in range(16): acc[kind][i] += buf[i]
    del cu, sc
for kind in acc:
    v = acc[kind]
    tot = sum(v[i] for i in (0, 1, 2, 3, 4))
    print(f"== {kind}: {ms[kind]:.1f} ms (stamped build), wave-0 cycles summed over workgroups {tot}")
    for i, n in enumerate(names):
        print(f"{n:52s} {v[i]:>16d}  {100.0 * v[i] / max(tot, 1):6.2f}%  ~{ms[kind] * v[i] / max(tot, 1):6.1f} ms")
