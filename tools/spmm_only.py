#!/usr/bin/env python3
"""A few launches of the SpMM (GCN aggregate, F = 256) and of the fused decode on the ppa-like graph -- the subject of the
HBM-traffic PMC passes (profiles/r01/gnn_traffic_pmc.json)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import ops, synth
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
gn = g.gcn_normalized()
N, H, E = g.n_rows, 256, 1 << 22
gen = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(N, H, generator=gen, device=dev)
b = torch.randn(H, generator=gen, device=dev)
for _ in range(3):
    ops.spmm_csr(gn.rowptr, gn.col, gn.val, x, bias=b, relu=True)
u = torch.randint(0, N, (E,), generator=gen, device=dev, dtype=torch.int32)
v = torch.randint(0, N, (E,), generator=gen, device=dev, dtype=torch.int32)
ws = [torch.randn(H if i < 2 else 1, H, generator=gen, device=dev) / 16 for i in range(3)]
bs = [torch.randn(H if i < 2 else 1, generator=gen, device=dev) for i in range(3)]
for _ in range(3):
    ops.mlp_decode(x, u, v, ws, bs)
torch.cuda.synchronize()
print("nnz", gn.nnz(), "N", N, "E", E)
