#!/usr/bin/env python3
"""Per-phase cycle sums of the fused expansion's fill kernel (diagnostic build tools/libeps_xstamp.so, -DEX_STAMP),
one production-sized launch of the ppa-like graph."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["EPS_LIB_PATH"] = os.path.join(ROOT, "tools", "libeps_xstamp.so")
import torch, eps_amd
from eps_amd import candidates, ops, synth, _lib
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
w = node_weight_table(g, ops.W_AA)
lo, hi = list(candidates.column_blocks(g))[3]
order = candidates.heaviest_first(g, lo, hi)
lib = _lib.load()
lib.eps_debug_expand_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
names = ["hand-out + barrier", "A mark (+ histogram) + known edges out", "B scan", "C emit + byte ranks", "tile plan",
         "D1 bin", "D2 per-tile sums + output", "plan..end of column"]
for want_cn, want_score in ((False, True), (True, True), (False, False)):
    for rep in range(2):
        ops.expand_candidates(g.rowptr, g.col, None, w, g.n_rows, lo, hi, want_cn=want_cn, want_score=want_score, col_order=order, max_paths=candidates.max_paths_of(g))
        torch.cuda.synchronize()
        lib.eps_debug_expand_stamps(buf, 1)
    tot = sum(buf[i] for i in (0, 1, 2, 3, 7))
    print(f"--- want_cn={want_cn} want_score={want_score}: wave-0 cycles summed over workgroups = {tot}")
    for i, n in enumerate(names):
        print(f"{n:40s} {buf[i]:>16d}  {100.0 * buf[i] / tot:6.2f}%")
