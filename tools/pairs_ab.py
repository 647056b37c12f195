#!/usr/bin/env python3
"""A/B timing of the pair-scoring kernels on the bench workload (one process, interleaved rounds)."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import eps_amd
from eps_amd import candidates, ops, synth
from eps_amd.heuristics import node_weight_table
sys.path.insert(0, ROOT)
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=1 << 25)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--variants", default="generic,grouped")
args = ap.parse_args()
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
u, v, _ = bench.build_pairs(g, 0, args.pairs, candidates, torch)
w = node_weight_table(g, ops.W_AA)
deg = g.degree()
print("pairs", u.numel(), "mean du", deg[u.long()].float().mean().item(), "mean dv", deg[v.long()].float().mean().item(),
      "max du", deg[u.long()].max().item(), "runs", int((v[1:] != v[:-1]).sum()) + 1)
res = {}
for r in range(args.rounds):
    for name in args.variants.split(","):
        grouped = name == "grouped"
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = ops.pair_scores(g.rowptr, g.col, None, w, g.n_rows, u, v, want_cn=False, grouped=grouped)
        b.record()
        torch.cuda.synchronize()
        res.setdefault(name, []).append(a.elapsed_time(b))
for k, t in res.items():
    print(k, "ms:", " ".join(f"{x:.2f}" for x in t), " min", min(t))
du = deg[u.long()].float()
print("P(du>256)", (du > 256).float().mean().item(), "P(du>512)", (du > 512).float().mean().item(),
      "P(du>1536)", (du > 1536).float().mean().item(),
      "mean extra units (>512)", ((du - 512).clamp(min=0) / 256).ceil().mean().item(),
      "mean trips x4", ((du - 512).clamp(min=0) / 1024).ceil().mean().item())
# eval-like lists: positive-like (actual edges) and uniform random negatives, through the generic kernel
gen = torch.Generator(device=dev).manual_seed(1)
row, colx, _ = g.coo()
sel = torch.randint(0, row.numel(), (6_000_000,), generator=gen, device=dev)
pos_u, pos_v = row[sel].to(torch.int32).contiguous(), colx[sel].to(torch.int32).contiguous()
neg_u = torch.randint(0, g.n_rows, (3_000_000,), generator=gen, device=dev, dtype=torch.int32)
neg_v = torch.randint(0, g.n_rows, (3_000_000,), generator=gen, device=dev, dtype=torch.int32)
for name, (a_, b_) in {"pos-like 6M": (pos_u, pos_v), "neg-uniform 3M": (neg_u, neg_v)}.items():
    ts = []
    for r in range(4):
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.pair_scores(g.rowptr, g.col, None, w, g.n_rows, a_, b_, want_cn=False, grouped=False); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(f"generic on {name}: min {min(ts):.3f} ms -> {a_.numel() / min(ts) / 1e6:.2f} G pairs/s, mean deg sum "
          f"{(deg[a_.long()] + deg[b_.long()]).float().mean().item():.0f}")
