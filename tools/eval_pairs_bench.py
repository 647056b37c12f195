import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, eps_amd
from eps_amd import ops, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
w = node_weight_table(g, ops.W_AA)
gen = torch.Generator(device=dev).manual_seed(0)
E = int(os.environ.get('E', 1 << 24))
u = torch.randint(0, g.n_rows, (E,), generator=gen, device=dev, dtype=torch.int32)
v = torch.randint(0, g.n_rows, (E,), generator=gen, device=dev, dtype=torch.int32)
row, col, _ = g.coo()
sel = torch.randint(0, row.numel(), (E,), generator=gen, device=dev)
pu, pv = row[sel].to(torch.int32).contiguous(), col[sel].to(torch.int32).contiguous()   # positive-like pairs (stored edges)
def t(fn, it=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / it
deg = g.degree()
for name, (a_, b_) in {"uniform random": (u, v), "stored edges": (pu, pv)}.items():
    ms = t(lambda: ops.pair_scores(g.rowptr, g.col, None, w, g.n_rows, a_, b_, want_cn=False, grouped=False))
    byt = 4 * (deg[a_.long()].sum() + deg[b_.long()].sum()).item() + 48 * E
    print(f"{name}: {E} pairs {ms:.2f} ms = {E / ms / 1e6:.2f} G pairs/s, {byt / ms / 1e9:.2f} TB/s algorithmic")
