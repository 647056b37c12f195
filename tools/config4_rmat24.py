#!/usr/bin/env python3
"""BASELINE configs[4] on ONE GPU's share: R-MAT scale-24 (16.7 M nodes, 256 M generated edges, symmetrised), CN + AA over
125 M pairs (1/8 of the 1 B): half uniform random, half 2-hop samples.  Checks counts on a sample against the same kernel with the endpoints swapped."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, eps_amd
from eps_amd import ops, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
t0 = time.perf_counter()
g = synth.rmat_graph(scale=24, edge_factor=16, seed=5, device=dev)
torch.cuda.synchronize()
print(f"graph: N={g.n_rows} nnz={g.nnz()} max deg={int(g.degree().max())} built in {time.perf_counter() - t0:.1f} s")
w = node_weight_table(g, ops.W_AA)
gen = torch.Generator(device=dev).manual_seed(1)
E = 125_000_000
half = E // 2
u1 = torch.randint(0, g.n_rows, (half,), generator=gen, device=dev, dtype=torch.int32)
v1 = torch.randint(0, g.n_rows, (half,), generator=gen, device=dev, dtype=torch.int32)
# 2-hop samples: a random stored entry (w, u), then a random neighbour v of w
e = torch.randint(0, g.nnz(), (half,), generator=gen, device=dev)
wnode = g.row_index()[e]
u2 = g.col[e]
deg = g.degree()
off = (torch.rand(half, generator=gen, device=dev) * deg[wnode]).long().clamp(max=deg.max() - 1)
off = torch.minimum(off, deg[wnode] - 1)
v2 = g.col[g.rowptr[wnode] + off]
u = torch.cat([u1, u2]).contiguous(); v = torch.cat([v1, v2]).contiguous()
for name, grouped in (("generic", False),):
    torch.cuda.synchronize(); t1 = time.perf_counter()
    cnt, _, ws = ops.pair_scores(g.rowptr, g.col, None, w, g.n_rows, u, v, want_cn=False, grouped=grouped)
    torch.cuda.synchronize(); dt = time.perf_counter() - t1
    print(f"{name}: {E} pairs in {dt * 1e3:.1f} ms -> {E / dt / 1e9:.2f} G pairs/s; mean CN {cnt.float().mean().item():.3f}; "
          f"mean deg sum {(deg[u.long()] + deg[v.long()]).float().mean().item():.0f}")
# sorted by v (candidate order): the column-run kernel with the HASHED bitmap (N > 2^20)
order = torch.argsort(v.long() * g.n_rows + u.long())
us, vs = u[order].contiguous(), v[order].contiguous()
print("runs long?", ops.v_runs_are_long(vs))
# independent check on a sample, without the CPU oracle (that one checks this kernel in tests/test_gpu_pair_scores.py, on an
# R-MAT graph too): the same pairs with the endpoints swapped take the other staging / search roles in the kernel
sel = torch.randint(0, E, (20000,), generator=gen, device=dev)
c_sw, _, a_sw = ops.pair_scores(g.rowptr, g.col, None, w, g.n_rows, v[sel].contiguous(), u[sel].contiguous(), want_cn=False,
                                grouped=False)
print("sample counts equal with endpoints swapped:", bool(torch.equal(c_sw, cnt[sel])))
