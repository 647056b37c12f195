#!/usr/bin/env python3
"""Shape of the scan's work on the ppa-like graph under hubs-first labels (input to the r03 single-pass kernel's geometry):
half paths per column, where the path ENDS lie in id space, how evenly a column's paths spread over equal-mass id windows,
and how many distinct candidates a window of a column holds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import scan, synth
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
if os.environ.get("RELABEL", "1") == "1":
    g = g.degree_ordered()[0]
n = g.n_rows
hp = scan.half_paths(g)
rev = scan.reverse_positions(g).long()
deg = g.degree()
print("N", n, "nnz", g.nnz(), "max degree", int(deg.max()), "half paths", int(hp.sum()))
q = torch.tensor([0.1, 0.25, 0.5, 0.75, 0.9, 0.99, 0.999, 1.0], device=dev)
print("half paths per column, quantiles", [int(x) for x in torch.quantile(hp.float(), q)])
for lim in (1024, 2048, 4096, 8192, 16384, 32768, 65536):
    m = hp <= lim
    print(f"columns with <= {lim:6d} half paths: {float(m.float().mean()):.3f} of columns, {float(hp[m].sum()) / float(hp.sum()):.3f} of paths")
# where the paths end: entry (u, w) of row u is the end of deg(w) - revpos - 1 half paths (the v > u of row w)
col = g.col.long()
ends = torch.zeros(n, dtype=torch.int64, device=dev)
ends.index_add_(0, g.row_index(), deg[col] - rev - 1)
assert int(ends.sum()) == int(hp.sum())
cum = torch.cumsum(ends, 0).double() / float(ends.sum())
for i in (256, 1024, 4096, 8192, 16384, 32768, 65536, 131072, 262144):
    print(f"path ends at ids < {i:7d}: {float(cum[i - 1]):.3f}")
M = int(os.environ.get("WINDOWS", "32"))
# equal-mass windows by stored entries (sum of degrees): boundaries in id space
cdeg = torch.cumsum(deg, 0).double() / float(deg.sum())
bounds = torch.searchsorted(cdeg, torch.arange(1, M, device=dev).double() / M) + 1
bounds = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), bounds, torch.tensor([n], device=dev)])
print("equal-entry window boundaries", bounds.tolist())
pe = torch.zeros(M, dtype=torch.int64, device=dev)
pe.index_add_(0, torch.bucketize(torch.arange(n, device=dev), bounds[1:-1], right=True), ends)
print("share of path ends per window", [round(float(x) / float(ends.sum()), 3) for x in pe])
# per-column spread over the windows + distinct candidates, for sample columns across the weight range
order = scan.column_order(g)
samples = [int(order[i]) for i in (0, 10, 100, 1000, 10000, 50000, 100000, 200000, 300000, 400000, 500000)]
rp = g.rowptr
for v in samples:
    nb = col[rp[v]:rp[v + 1]]
    r = rev[rp[v]:rp[v + 1]]
    starts = rp[nb]
    tot = int(r.sum())
    if tot == 0:
        continue
    idx = torch.repeat_interleave(starts, r) + (torch.arange(tot, device=dev) - torch.repeat_interleave(torch.cumsum(r, 0) - r, r))
    u = col[idx]
    w_of = torch.bucketize(u, bounds[1:-1], right=True)
    pw = torch.bincount(w_of, minlength=M)
    uu = torch.unique(u)
    dw = torch.bincount(torch.bucketize(uu, bounds[1:-1], right=True), minlength=M)
    nz = pw > 0
    print(f"column {v:7d} deg {int(deg[v]):5d} half paths {tot:7d} distinct {uu.numel():7d} | per window: paths max {int(pw.max())} "
          f"mean(nonzero) {float(pw[nz].float().mean()):.0f}; distinct max {int(dw.max())}; paths {pw.tolist()}")
