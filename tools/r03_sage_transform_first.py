#!/usr/bin/env python3
"""SAGE forward on the ppa stand-in (58 one-hot features + 256-d embedding = 314 -> 256, L = 3: BASELINE configs[3]'s rank model):
layer 1 narrows, so it aggregates lin_l(x) (transform-first, r03) instead of x (r02).  Time of the whole 3-layer forward and of
layer 1 alone, both orders, and the largest difference between the two results."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import models, synth
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
n = g.n_rows
gen = torch.Generator(device=dev).manual_seed(1)
x = torch.cat([torch.randn(n, 256, generator=gen, device=dev),
               torch.nn.functional.one_hot(torch.randint(0, 58, (n,), generator=gen, device=dev), 58).float()], 1).contiguous()
torch.manual_seed(0)
gnn = models.SAGE(314, 256, 256, 3, 0.0).to(dev).eval()
def T(fn, reps=10):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3, r
out = {}
for tf in (False, True):
    models.SAGEConv.TRANSFORM_FIRST = tf
    with torch.no_grad():
        t_all, h = T(lambda: gnn(x, g))
        t_l1, _ = T(lambda: gnn.convs[0](x, g, relu=True))
    out[tf] = h
    print(f"transform_first={tf}: 3-layer SAGE forward {t_all:.2f} ms, layer 1 (314 -> 256) {t_l1:.2f} ms")
d = (out[True] - out[False]).abs().max().item()
print(f"max |difference| of the final embeddings {d:.3e} (scale {out[False].abs().max().item():.3e})")
