import ctypes, os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["EPS_LIB_PATH"] = os.path.join(ROOT, "tools", "libeps_stamp.so")
import torch, eps_amd, bench
from eps_amd import candidates, ops, synth, _lib
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
u, v, _ = bench.build_pairs(g, 0, 1 << 25, candidates, torch)
w = node_weight_table(g, ops.W_AA)
lib = _lib.load()
buf = (ctypes.c_ulonglong * 16)()
ops.pair_scores(g.rowptr, g.col, None, w, g.n_rows, u, v, want_cn=False, grouped=True); torch.cuda.synchronize()
lib.eps_debug_stamps(buf, 1)
ops.pair_scores(g.rowptr, g.col, None, w, g.n_rows, u, v, want_cn=False, grouped=True); torch.cuda.synchronize()
lib.eps_debug_stamps(buf, 1)
names = ["barrier-wait", "bitmap rebuild", "group meta", "ring wait", "units 0-1", "on-demand units", "final flush", "whole wave", "node0+select", "refill issue"]
tot = buf[7]
for i, n in enumerate(names):
    print(f"{n:18s} {buf[i]:>16d}  {100.0 * buf[i] / tot:6.2f}%")
