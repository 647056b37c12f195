#!/usr/bin/env python3
"""Per-phase cycle shares of eps_filter_scan's main launch on the ppa-like graph (diagnostic build: tools/make_scan_stamps.py)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["EPS_LIB_PATH"] = os.path.join(ROOT, "tools", "libeps_fsstamp.so")
import torch, eps_amd
from eps_amd import ops, scan, synth, _lib
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
if os.environ.get("RELABEL", "1") == "1":
    g = g.degree_ordered()[0]               # hubs first: the labels a repeatedly scanned graph runs under
w = node_weight_table(g, ops.W_AA)
fixw = scan.fixed_weights(g, w)
lib = _lib.load()
lib.eps_debug_scan_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
names = ["hand-out + column setup", "A rounds: describe + mark + histogram", "known edges out + rank tables (B)", "plan",
         "D total (windows) incl. scans", "  of which D2 / tile scans", "  direct-mode windows (D1+scan)", "  record-mode windows (D1+D2)", "    D2: wait for the records (barrier after D1)",
         "    D2 per tile: request + accumulate", "    D2 per tile: barrier (sums complete)", "    D2 per tile: scan",
         "    D2 per tile: barrier (zeroed)", "(count) tiles in record mode", "(count) columns with candidates", "(count) single-round columns"]
bar = float(os.environ.get("BAR", "3.25"))
for rep in range(2):
    res = ops.Survivors(64 << 20, bar, dev)
    ops.filter_scan(g.rowptr, g.col, scan.reverse_positions(g), fixw, g.n_rows, scan.column_order(g), res, scan.max_degree(g), scan.window_splits(g))
    torch.cuda.synchronize()
    lib.eps_debug_scan_stamps(buf, 1)
tot = sum(buf[i] for i in (0, 1, 2, 3, 4))
print(f"wave-0 cycles summed over workgroups: {tot}")
for i, n in enumerate(names):
    print(f"{n:45s} {buf[i]:>16d}  {100.0 * buf[i] / tot:6.2f}%")
