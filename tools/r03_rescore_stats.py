#!/usr/bin/env python3
"""Shape of the re-scoring job on the ppa-like graph (hubs-first labels): how many screened pairs reach eps_rescore_runs after
the approximate cut, how long their runs of equal u are, and how long the rows on both sides are."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd  # noqa: E401,F401
from eps_amd import ops, scan, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g0 = synth.ppa_like(seed=3, device=dev)
w = node_weight_table(g0, ops.W_AA)
K = 4_000_000
st = {}
scan.scan_topk(g0, w, K, relabel=True, stats=st)
g, perm = scan.scan_graph(g0)
sc = scan.screen_weights(g0, g, perm, w)
fixw = scan._scan_weights(g0, g, perm, w)
order = scan.column_order(g)
bar = float(st["bar"])
res = scan._launch(g, fixw, order, torch.tensor([bar], device=dev), 48 << 20, both=True, screen=sc)
ck, cv, nv = ops.compact_at_least(res.key, res.val, None)
nv = int(nv.item())
keys, approx = ck[:nv], cv[:nv]
k2 = (K + 1) // 2
cut = torch.topk(approx, min(int(k2 * 1.0), nv)).values[-1]
for name, sel in (("all screened survivors", torch.ones_like(approx, dtype=torch.bool)), ("those at or above the approximate cut", approx >= cut)):
    kk = keys[sel]
    u, v = kk & 0xFFFFFFFF, kk >> 32
    deg = g.degree().long()
    du, dv = deg[u], deg[v]
    uu, cnt = torch.unique(u, return_counts=True)
    q = lambda x: [int(t) for t in torch.quantile(x.double(), torch.tensor([.1, .5, .9, .99], dtype=torch.float64, device=dev)).tolist()]
    long_u = du > 512
    print(f"{name}: {kk.numel()} pairs, {uu.numel()} distinct u (runs): pairs per run p10/50/90/99 {q(cnt)} mean {cnt.double().mean():.1f}; "
          f"deg(u) {q(du)} mean {du.double().mean():.0f}; deg(v) {q(dv)} mean {dv.double().mean():.0f}; "
          f"pairs with deg(u) > 512: {int(long_u.sum())} ({100 * long_u.double().mean():.1f} %); entries of N(v) they stream: {int(dv[long_u].sum()) / 1e6:.0f} M; "
          f"bitmap bits they set (one per run and 256-pair chunk at least): {int(deg[uu[deg[uu] > 512]].sum()) / 1e6:.1f} M")
