#!/usr/bin/env python3
"""scan_topk's own statistics (rescored pairs, walked slots, budget ...) for AA / RA / CN on the ppa-like graph.  env KIND, K"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import ops, scan, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g0 = synth.ppa_like(seed=3, device=dev)
K = int(os.environ.get("K", 4_000_000))
_sort = ops.sort_pairs_by_u
_shape = {}


def _wrapped(keys, id_bits=32, v_block_shift=0):
    out = _sort(keys, id_bits, v_block_shift)
    u, v = out >> 32, out & 0xFFFFFFFF
    _shape.clear()
    _shape.update({"pairs": int(out.numel()), "distinct_u": int(torch.unique(u).numel()),
                   **{f"runs_at_2^{s}": int(torch.unique(((v >> s) << 32) | u).numel()) for s in (10, 12, 14, 16)}})
    return out


ops.sort_pairs_by_u = _wrapped
for kind in (sys.argv[1:] or ["aa", "ra", "cn"]):
    w = (torch.ones(g0.n_rows, dtype=torch.float32, device=dev) if kind == "cn" else node_weight_table(g0, {"aa": ops.W_AA, "ra": ops.W_RA}[kind]))
    for i in range(3):
        st = {}
        rows, vals = scan.scan_topk(g0, w, K, stats=st, relabel=True)
    print(kind, json.dumps({k: (v if isinstance(v, (int, float, bool, str)) or v is None else str(v)) for k, v in st.items()}), json.dumps(_shape), flush=True)
