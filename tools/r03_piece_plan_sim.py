#!/usr/bin/env python3
"""Host-side replay of scan_piece_kernel's planner (csrc/scan_pieces.hip: greedy merge of id windows into DIRECT / HASH pieces)
over all columns of the ppa-like graph under hubs-first labels: how many pieces of each kind, how many paths they carry, and
the same for candidate geometries (bigger hash pieces, other window counts) -- what a change of the table layout would buy
BEFORE writing it.  Vectorised over the columns (<= M + 1 greedy steps)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd  # noqa: E401,F401
from eps_amd import ops, scan, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g0 = synth.ppa_like(seed=3, device=dev)
g, perm, _ = g0.degree_ordered()
bounds, cuts = scan.screen_tables(g)
wp = scan.window_paths(g).to(torch.int64) & 0xFFFFFFFF                     # [N, M]
N, M = wp.shape
b = bounds.to(torch.int64)                                                  # [M + 1]
v = torch.arange(N, device=dev).unsqueeze(1)
ek = torch.cat([torch.zeros(N, 1, dtype=torch.int64, device=dev), torch.cumsum(wp, 1)], 1)      # paths before window k
nbk = torch.cat([torch.zeros(N, 1, dtype=torch.int64, device=dev), cuts.to(torch.int64) & 0xFFFF], 1)   # neighbours of v below b[k]
hi_k = torch.minimum(b.unsqueeze(0).expand(N, M + 1), v)                    # end of the run [.., k) in id space
kv = ((b[1:M].unsqueeze(0) <= v - 1)).sum(1)                                # windows 0 .. kv hold ids below v
deg = g.degree().to(torch.int64)
w = node_weight_table(g0, ops.W_AA)[perm].double()
S = torch.zeros(N, dtype=torch.float64, device=dev).index_add_(0, torch.repeat_interleave(torch.arange(N, device=dev), deg), w[g.col.long()])
lane = torch.arange(M + 1, device=dev).unsqueeze(0)


def plan(direct_ids, hash_paths, hash16_paths=0, span16=1 << 17, s16_max=0.0):
    """-> dict kind -> (pieces, paths); kinds: direct, hash, hash16, part (hash-partitioned single windows: passes counted)"""
    k0 = torch.zeros(N, dtype=torch.int64, device=dev)
    live = (deg > 0) & (torch.arange(N, device=dev) > 0)
    out = {k: [0, 0] for k in ("direct", "hash", "hash16", "part")}
    hist = {k: [] for k in out}
    smax_w = None
    if hash16_paths:
        # largest S(u) inside each window: a 16-bit-sum piece needs min(S(v), max S(u) of its ids) below the 16-bit range
        smax_w = torch.stack([S[int(b[k]):int(b[k + 1])].max() if int(b[k + 1]) > int(b[k]) else S.new_zeros(()) for k in range(M)])
        smax_run = torch.cummax(smax_w.flip(0), 0).values.flip(0)         # (hubs first: S falls with the id; max of a run = its first window)
    for _ in range(M + 2):
        act = live & (k0 <= kv)
        if not bool(act.any()):
            break
        lo = b[k0].unsqueeze(1)
        e0 = ek.gather(1, k0.unsqueeze(1))
        nb0 = nbk.gather(1, k0.unsqueeze(1))
        inr = (lane > k0.unsqueeze(1)) & (lane <= (kv + 1).unsqueeze(1))
        kd = k0 + (inr & (hi_k - lo <= direct_ids)).sum(1)
        keys = (ek - e0) + (nbk - nb0)
        kh = k0 + (inr & (keys <= hash_paths)).sum(1)
        kh16 = k0.clone()
        if hash16_paths:
            ok16 = torch.minimum(S, smax_run[k0.clamp(max=M - 1)]) < s16_max
            kh16 = k0 + (inr & (keys <= hash16_paths) & (hi_k - lo <= span16) & ok16.unsqueeze(1)).sum(1)
        is_d = (kd >= torch.maximum(kh, kh16)) & (kd > k0)
        is_16 = ~is_d & (kh16 > kh) & (kh16 > k0)
        is_h = ~is_d & ~is_16 & (kh > k0)
        is_p = ~is_d & ~is_16 & ~is_h
        k1 = torch.where(is_d, kd, torch.where(is_16, kh16, torch.where(is_h, kh, k0 + 1))).clamp(max=M)
        paths = ek.gather(1, k1.unsqueeze(1)).squeeze(1) - e0.squeeze(1)
        nz = act & (paths > 0)
        for name, m in (("direct", is_d), ("hash", is_h), ("hash16", is_16), ("part", is_p)):
            sel = nz & m
            if name == "part":
                kk = (keys.gather(1, k1.unsqueeze(1)).squeeze(1))[sel]
                parts = torch.ones_like(kk)
                while bool((kk > hash_paths * parts).any()):
                    parts = torch.where(kk > hash_paths * parts, parts * 2, parts)
                parts = torch.where(parts > 1, parts * 2, parts)
                out[name][0] += int(parts.sum())
            else:
                out[name][0] += int(sel.sum())
            out[name][1] += int(paths[sel].sum())
            hist[name].append(paths[sel])
        k0 = torch.where(act, k1, k0)
    return out, {k: torch.cat(x) if x else torch.zeros(0, dtype=torch.int64, device=dev) for k, x in hist.items()}


def plan2(direct_ids, packed_paths, ratio=2270, direct16_ids=0, s16_max=0.0):
    """The product planner (packed pieces for everything hashed; a direct run wins against a longer packed one iff
    (Pp - Pd) * ratio < Pd * Pp), optionally with a second direct kind of `direct16_ids` ids for columns / windows whose sums
    stay below s16_max."""
    k0 = torch.zeros(N, dtype=torch.int64, device=dev)
    live = (deg > 0) & (torch.arange(N, device=dev) > 0)
    out = {k: [0, 0] for k in ("direct", "direct16", "packed", "part")}
    hist = {k: [] for k in out}
    smax_w = torch.stack([S[int(b[k]):int(b[k + 1])].max() if int(b[k + 1]) > int(b[k]) else S.new_zeros(()) for k in range(M)])
    smax_run = torch.cummax(smax_w.flip(0), 0).values.flip(0)
    for _ in range(M + 2):
        act = live & (k0 <= kv)
        if not bool(act.any()):
            break
        lo = b[k0].unsqueeze(1)
        e0 = ek.gather(1, k0.unsqueeze(1))
        nb0 = nbk.gather(1, k0.unsqueeze(1))
        inr = (lane > k0.unsqueeze(1)) & (lane <= (kv + 1).unsqueeze(1))
        kd32 = k0 + (inr & (hi_k - lo <= direct_ids)).sum(1)
        kd16 = kd32.clone()
        if direct16_ids:
            ok16 = torch.minimum(S, smax_run[k0.clamp(max=M - 1)]) < s16_max
            kd16 = torch.maximum(kd32, k0 + (inr & (hi_k - lo <= direct16_ids) & ok16.unsqueeze(1)).sum(1))
        kd = kd16
        keys = (ek - e0) + (nbk - nb0)
        kp = k0 + (inr & (keys <= packed_paths)).sum(1)
        pd = ek.gather(1, kd.clamp(max=M).unsqueeze(1)).squeeze(1) - e0.squeeze(1)
        pp = ek.gather(1, kp.clamp(max=M).unsqueeze(1)).squeeze(1) - e0.squeeze(1)
        take_d = (kd >= kp) & (kd > k0)
        take_d |= (kd > k0) & (kp > kd) & ((pp - pd) * ratio < pd * pp)
        is_16 = take_d & (kd > kd32)
        is_d = take_d & ~is_16
        is_p = ~take_d & (kp > k0)
        is_x = ~take_d & ~is_p
        k1 = torch.where(take_d, kd, torch.where(is_p, kp, k0 + 1)).clamp(max=M)
        paths = ek.gather(1, k1.unsqueeze(1)).squeeze(1) - e0.squeeze(1)
        nz = act & (paths > 0)
        for name, m in (("direct", is_d), ("direct16", is_16), ("packed", is_p), ("part", is_x)):
            sel = nz & m
            out[name][0] += int(sel.sum())
            out[name][1] += int(paths[sel].sum())
            hist[name].append(paths[sel])
        k0 = torch.where(act, k1, k0)
    return out, {k: torch.cat(x) if x else torch.zeros(0, dtype=torch.int64, device=dev) for k, x in hist.items()}


def show(title, res):
    out, hist = res
    tot_p = sum(x[0] for x in out.values())
    tot = sum(x[1] for x in out.values())
    print(f"{title}: {tot_p} pieces (passes), {tot / 1e9:.2f} G paths")
    for k, (n, p) in out.items():
        if n:
            h = hist[k].double()
            q = torch.quantile(h, torch.tensor([0.1, 0.5, 0.9], dtype=torch.float64, device=dev)).tolist() if h.numel() else [0, 0, 0]
            print(f"    {k:7s} {n:9d} pieces, {p / 1e9:6.2f} G paths ({100 * p / tot:4.1f} %), paths per piece: mean {p / n:7.0f}, "
                  f"p10/p50/p90 {q[0]:.0f}/{q[1]:.0f}/{q[2]:.0f}")


print(f"graph: N {N}, M {M}, S(v) quantiles 50/90/99/max: "
      f"{[round(x, 1) for x in torch.quantile(S[:: max(1, N // 1000000)], torch.tensor([.5, .9, .99, 1.0], dtype=torch.float64, device=dev)).tolist()]}")
show("product geometry (4096 slots: direct 8192 ids, hash 2048 paths)", plan(8192, 2048))
show("8192 slots (variant 0)", plan(16384, 4096))
for s16 in (32.0, 64.0, 128.0):
    show(f"+ 4-byte hash slots: 4096 paths, span <= 2^17, min(S(v), max S(u)) < {s16}", plan(8192, 2048, 4096, 1 << 17, s16))
show("+ 4-byte hash slots, LF 3/4: 6144 paths, S < 64", plan(8192, 2048, 6144, 1 << 17, 64.0))
show("+ 4-byte hash slots: 4096 paths, span <= 2^16, S < 64", plan(8192, 2048, 4096, 1 << 16, 64.0))

print()
show("product planner r03b (direct 8192 ids | packed 4096 paths, cost rule 2270)", plan2(8192, 4096))
for s16 in (64.0, 128.0):
    show(f"+ direct16 (16384 ids, two 16-bit sums per word) where min(S(v), max S(u)) < {s16}", plan2(8192, 4096, 2270, 16384, s16))
show("+ direct16 everywhere (upper bound of what it can give)", plan2(8192, 4096, 2270, 16384, 1e30))
show("direct 16384 ids everywhere, ratio 1500", plan2(16384, 4096, 1500))
