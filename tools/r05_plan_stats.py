#!/usr/bin/env python3
"""Pieces of the scan's plan table by kind (with and without skipped heads at a bar): count, paths, ids swept per piece.
usage: r05_plan_stats.py [beta ...]; env BAR, KIND"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import ops, scan, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g0 = synth.ppa_like(seed=3, device=dev)
kind = os.environ.get("KIND", "aa")
w = (torch.ones(g0.n_rows, dtype=torch.float32, device=dev) if kind == "cn" else node_weight_table(g0, {"aa": ops.W_AA, "ra": ops.W_RA}[kind]))
g, perm = g0.degree_ordered()[:2]
g._cache["scan_calls"] = 2          # (the full-width hub table of a graph that is scanned repeatedly: scan.hub_rows)
sc = scan.screen_weights(g0, g, perm, w)
bar = float(os.environ.get("BAR", 2.876))
bounds = scan.screen_tables(g)[0].cpu()


def stats(name, plan, heads=None):
    pptr, recs = plan
    r = recs.cpu().numpy().view("uint32")
    info, y, z, lo = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
    kinds = info >> 30
    paths = (info & 0x3FFFFFFF).astype("int64")
    k1 = (y >> 8) & 0xFF
    import numpy as np
    col_of = np.repeat(np.arange(g.n_rows), np.diff(pptr.cpu().numpy().view("uint32").astype("int64")))
    hi = np.minimum(bounds.numpy()[k1], col_of)
    span = hi.astype("int64") - lo.astype("int64")
    out = {"plan": name, "pieces": int(len(info)), "paths": int(paths.sum())}
    for code, nm in ((3, "direct16"), (2, "direct32"), (1, "packed"), (0, "hash")):
        m = kinds == code
        if m.any():
            out[nm] = {"pieces": int(m.sum()), "paths": int(paths[m].sum()), "paths_per_piece": round(float(paths[m].mean()), 1),
                       "span_mean": round(float(span[m].mean()), 1), "known_edges": int(((z[m] >> 16) - (z[m] & 0xFFFF)).sum())}
    ppc = np.diff(pptr.cpu().numpy().view("uint32").astype("int64"))
    out["columns_with_pieces"] = int((ppc > 0).sum())
    out["pieces_per_column_hist"] = np.bincount(np.minimum(ppc, 12)).tolist()
    ssum = sc.ssum.cpu().numpy().view("uint32").astype("int64")
    thr = int(bar * 2 ** sc.shift)
    dead = ssum < thr
    out["dead_columns"] = int(dead.sum())
    out["pieces_in_dead_columns"] = int(ppc[dead].sum())
    out["paths_in_dead_columns"] = int(paths[dead[col_of]].sum())
    print(json.dumps(out), flush=True)


stats("no heads", sc.plan)
for beta in [float(x) for x in (sys.argv[1:] or ["0.5"])]:
    scan.HEAD_BETA = beta
    ht = scan.head_tables(g, sc, scan.head_budget(bar * 2.0 ** sc.shift))
    stats(f"beta {beta}", ht.plan)
