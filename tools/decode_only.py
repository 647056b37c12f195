#!/usr/bin/env python3
"""A few launches of the fused MLP decode (H=256, L=2 and 3, 2^22 random edges of the ppa-like id space) and of the
fp32-MFMA GEMM -- the subject of the MFMA-utilisation PMC pass (profiles/r01/decode_mfma_pmc.json)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import eps_amd
from eps_amd import ops
dev = torch.device("cuda:0")
N, H, E = 576289, 256, 1 << 22
gen = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(N, H, generator=gen, device=dev)
u = torch.randint(0, N, (E,), generator=gen, device=dev, dtype=torch.int32)
v = torch.randint(0, N, (E,), generator=gen, device=dev, dtype=torch.int32)
for nl in (2, 3):
    ws = [torch.randn(H if i < nl - 1 else 1, H, generator=gen, device=dev) / 16 for i in range(nl)]
    bs = [torch.randn(H if i < nl - 1 else 1, generator=gen, device=dev) for i in range(nl)]
    for _ in range(3):
        ops.mlp_decode(x, u, v, ws, bs)
w = torch.randn(H, H, generator=gen, device=dev)
for _ in range(3):
    ops.gemm(x, w)
# the layer shape of the ppa recipe with its fused epilogue: [N x 316] x [316 x 256] + bias, ReLU
x316 = torch.randn(N, 316, generator=gen, device=dev)
w316 = torch.randn(H, 316, generator=gen, device=dev)
b316 = torch.randn(H, generator=gen, device=dev)
y = torch.empty(N, H, device=dev)
for _ in range(6):
    ops.gemm(x316, w316, bias=b316, relu=True, out=y)
torch.cuda.synchronize()
print("ok")
