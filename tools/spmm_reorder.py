#!/usr/bin/env python3
"""Does vertex reordering help the SpMM?  Same ppa-like graph, F = 256, three labelings:
as generated (ids randomly permuted), hubs first (descending degree), and hubs first with X in that order too."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import ops, synth
from eps_amd.graph import CSRGraph
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
n, f = g.n_rows, 256
x = torch.randn(n, f, device=dev)
def timed(gr, xx, reps=10):
    y = torch.empty_like(xx)
    for _ in range(3): ops.spmm_csr(gr.rowptr, gr.col, gr.val, xx, out=y)
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in e:
        a.record(); ops.spmm_csr(gr.rowptr, gr.col, gr.val, xx, out=y); b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in e) / reps, y
t0, y0 = timed(g, x)
deg = g.degree()
perm = torch.argsort(deg, descending=True, stable=True)          # new id i = old id perm[i]
inv = torch.empty_like(perm); inv[perm] = torch.arange(n, device=dev)
row, col, _ = g.coo()
g2 = CSRGraph.from_edge_index(torch.stack([inv[row], inv[col]]), None, sparse_sizes=(n, n))
t1, y1 = timed(g2, x[perm].contiguous())
assert torch.allclose(y1[inv], y0, rtol=1e-3, atol=1e-3)
gather = g.nnz() * f * 4 + g.nnz() * 8 + n * f * 4
print(f"as generated      : {t0:.2f} ms  {gather / t0 / 1e9:.2f} TB/s on the gather model")
print(f"hubs first (degree): {t1:.2f} ms  {gather / t1 / 1e9:.2f} TB/s   (top 5 % of the nodes hold {float(deg[perm[:n // 20]].sum()) / g.nnz():.2f} of the entries)")
