#!/usr/bin/env python3
"""Skipped heads (csrc/scan_heads.hip) against the plain launch on the ppa-like graph at a fixed bar: kernel ms of the main
launch (HIP events, min / median of REPS), the refine kernel, the walked list's length, the half paths still walked, the pieces of
the plan -- and a digest of the survivor list after exact re-scoring, which must be the same for every beta (0 = no heads).
usage: r05_heads_ab.py [beta ...]; env: BAR, REPS, NODES, EDGES, KIND (aa / cn / ra), HUB (bits per mask row), VARIANT"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import ops, scan, synth
from eps_amd.heuristics import node_weight_table

dev = torch.device("cuda:0")
g0 = synth.ppa_like(seed=3, device=dev, n_nodes=int(os.environ.get("NODES", 576289)), n_undirected=int(os.environ.get("EDGES", 21231931)))
kind = os.environ.get("KIND", "aa")
w = (torch.ones(g0.n_rows, dtype=torch.float32, device=dev) if kind == "cn" else
     node_weight_table(g0, {"aa": ops.W_AA, "ra": ops.W_RA}[kind]))
g, perm = g0.degree_ordered()[:2]
g._cache["scan_calls"] = 2          # (the full-width hub table of a graph that is scanned repeatedly: scan.hub_rows)
if "VARIANT" in os.environ:          # force a table geometry (0: 512 threads / 64 KB, 1: 1024 / 128 KB, 2: 256 / 32 KB)
    g._cache["screen_variant"] = int(os.environ["VARIANT"])
order = scan.column_order(g)
reps = int(os.environ.get("REPS", "5"))
if "HUB" in os.environ:
    ops.HUB_MAX = int(os.environ["HUB"])
    scan.HUB_TABLE_BYTES = 8 << 30
if "BATCH_PATHS" in os.environ:
    scan.BATCH_PATHS = int(os.environ["BATCH_PATHS"])
sc = scan.screen_weights(g0, g, perm, w)
bounds, cuts = scan.screen_tables(g)
variant = scan.screen_variant(g)
if "BAR" in os.environ:
    bar = float(os.environ["BAR"])
else:
    bar = float(scan.estimate_bar(g, sc.fixw, int(os.environ.get("K", 4000000)), screen=sc))
print(json.dumps({"bar": bar, "shift": sc.shift, "half_paths": scan.total_half_paths(g), "pieces": int(sc.plan[1].shape[0])}), flush=True)
bar_t = torch.tensor([bar], device=dev)


def timed(fn):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return r, ts[0], ts[len(ts) // 2]


def digest(res):
    slots, ncand = res.counts()
    keys, vals = res.valid(slots)
    k2, v2 = scan.rescore_exact(g, sc, keys, bar_t)
    m = k2 >= 0
    o = torch.argsort(k2[m])
    return (hashlib.sha256(k2[m][o].cpu().numpy().tobytes() + v2[m][o].cpu().numpy().tobytes()).hexdigest()[:16], int(keys.numel()), int(m.sum()), ncand)


out = []
for beta in [float(x) for x in (sys.argv[1:] or ["0", "0.25", "0.5", "0.625"])]:
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    if beta == 0:
        def run():
            res = ops.Survivors(48 << 20, bar, dev, prefill=False)
            ops.scan_screen(g.rowptr, g.col, scan.reverse_positions(g), sc.fx32, cuts, bounds, g.n_rows, order, sc.shift, res, status, variant,
                            wpaths=scan.window_paths(g), ssum=sc.ssum, smax=sc.smax, plan=sc.plan,
                            batch_from=int(os.environ['BATCH_FROM']) if 'BATCH_FROM' in os.environ else scan.batch_from(g, order),
                            rowrec=None if os.environ.get('ROWREC', '1') == '0' else sc.rowrec)
            return res
        res, tmin, tmed = timed(run)
        d = digest(res)
        row = {"beta": 0, "kernel_min_ms": round(tmin, 3), "kernel_median_ms": round(tmed, 3), "pieces": int(sc.plan[1].shape[0])}
    else:
        scan.HEAD_BETA = beta
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        masks = scan.hub_rows(g)
        torch.cuda.synchronize(); e0.record()
        ht = scan.head_tables(g, sc, scan.head_budget(bar * 2.0 ** sc.shift), wide=os.environ.get('WIDE', '0') == '1')
        e1.record(); torch.cuda.synchronize()
        t_tables = e0.elapsed_time(e1)
        walked_paths = int(ht.wpaths.view(torch.int32).to(torch.int64).bitwise_and(0xFFFFFFFF).sum())
        pack, t_pack = None, 0.0
        if os.environ.get('PACK', '0') == '1':          # (r06: the per-column pack; with COLREC / ROWREC on, the specialised body)
            p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            p0.record()
            pack = ops.scan_column_pack(g.rowptr, g.col, scan.reverse_positions(g), sc.rowrec, ht.plan)
            p1.record(); torch.cuda.synchronize()
            t_pack = p0.elapsed_time(p1)
        cols = scan.live_columns(g, sc, ht, 0, 1) if os.environ.get('LIVE', '0') == '1' else order
        def run():
            wk = ops.Survivors(256 << 20, bar, dev, prefill=False)
            ops.scan_screen(g.rowptr, g.col, scan.reverse_positions(g), sc.fx32, cuts, bounds, g.n_rows, cols, sc.shift, wk, status,
                            variant | (ops.SCAN_SKETCH if os.environ.get('SKETCH', '0') == '1' else 0) | (ops.SCAN_WIDE if os.environ.get('WIDE', '0') == '1' else 0),
                            wpaths=ht.wpaths, ssum=sc.ssum, smax=sc.smax, plan=ht.plan, heads=ht.heads,
                            colrec=None if os.environ.get('COLREC', '1') == '0' else scan.column_records(g, sc, cols, ht.plan, ht.heads, ht.live, ("ab", cols.numel())),
                            batch_from=int(os.environ['BATCH_FROM']) if 'BATCH_FROM' in os.environ else scan.batch_from(g, cols),
                            rowrec=None if os.environ.get('ROWREC', '1') == '0' else sc.rowrec, pack=pack)
            return wk
        wk, tmin, tmed = timed(run)
        def fin():
            res = ops.Survivors(48 << 20, bar, dev, prefill=False)
            ops.scan_refine(wk, ht.heads, masks, sc.fx32, g.rowptr, g.col, g.n_rows, sc.shift, res)
            return res
        res, rmin, rmed = timed(fin)
        d = digest(res)
        hx = ht.heads[:, 0].to(torch.int64)
        row = {"beta": beta, "budget": ht.budget * 2.0 ** -sc.shift, "kernel_min_ms": round(tmin, 3), "kernel_median_ms": round(tmed, 3),
               "refine_min_ms": round(rmin, 3), "tables_ms": round(t_tables, 3), "walked_slots": wk.counts()[0], "walked_paths": walked_paths,
               "walked_share": round(walked_paths / scan.total_half_paths(g), 4), "pieces": int(ht.plan[1].shape[0]), "d_used": ht.d_used,
               "rows_skipped_mean": round(float(hx.float().mean()), 2), "rows_skipped_max": int(hx.max()), "hub_rows": masks.shape[0],
               "pack": pack is not None, "pack_build_ms": round(t_pack, 3), "live_columns_only": cols is not order,
               "generic_body": os.environ.get("EPS_SCAN_GENERIC", "0") == "1", "sketch": os.environ.get("SKETCH", "0") == "1"}
    row.update(status=int(status), digest=d[0], screened=d[1], exact=d[2], counted=d[3])
    out.append(row)
    print(json.dumps(row), flush=True)
print("all lists identical:", len({r["digest"] for r in out}) == 1)
