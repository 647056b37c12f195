#!/usr/bin/env python3
"""configs[0] (ddi-like, N = 4,267, 2.14 M stored entries): the CN candidate list through the dense product (csrc/dense_cn.hip)
against the sparse list kernel, stage by stage (HIP events), and the whole `filter.py --model simple` scoring section."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import argparse, torch, eps_amd
from eps_amd import candidates, datasets, ops, scan, filter_stage
dev = torch.device("cuda:0")
edge_index, edge_weight, split_edge, data = datasets.get_data(argparse.Namespace(dataset="ddi", synthetic=True, use_feature=False))
from eps_amd.graph import add_edges
g = add_edges("ddi", edge_index.to(dev), edge_weight.to(dev), torch.zeros(2, 0, dtype=torch.long, device=dev), data.num_nodes)
n = g.n_rows


def ms(fn, reps=5):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return r, min(ts)


a, t_adj = ms(lambda: ops.dense_adjacency(g.rowptr, g.col, n))
c = torch.empty_like(a)
_, t_gemm = ms(lambda: ops.gemm(a, a, out=c, lower_only=True))
_, t_gemm_full = ms(lambda: ops.gemm(a, a, out=c))
(keys, vals), t_all = ms(lambda: ops.dense_cn_candidates(g.rowptr, g.col, n))
ones = torch.ones(n, dtype=torch.float32, device=dev)
scan.reverse_positions(g); scan.window_splits(g)
def sparse():
    ks, vs = [], []
    for lo, hi in candidates.column_blocks(g):
        r = ops.expand_unit(g.rowptr, g.col, ones, n, lo, hi, scan.max_degree(g), scan.window_splits(g),
                            col_order=candidates.heaviest_first(g, lo, hi), revpos=scan.reverse_positions(g))
        ks.append((r.pairs[1].to(torch.int64) << 32) | r.pairs[0].to(torch.int64)); vs.append(r[4])
    return torch.cat(ks), torch.cat(vs)
(sk, sv), t_sparse = ms(sparse)
_, t_rows = ms(lambda: scan.select_topk(keys, vals, 2 * keys.numel(), n))
(dk, dc), t_directed = ms(lambda: ops.dense_cn_candidates(g.rowptr, g.col, n, directed=True, check_symmetric=True))
_, t_sort16 = ms(lambda: torch.sort(dc.to(torch.int16), descending=True, stable=True))
print(json.dumps({"directed_list_incl_symmetry_check_ms": t_directed, "stable_sort_int16_counts_ms": t_sort16, "rows": int(dk.numel())}))
print(json.dumps({"nodes": n, "nnz": g.nnz(), "unordered_candidates": int(keys.numel()), "dense_ms": {"adjacency": t_adj, "product_lower_tiles": t_gemm,
                  "product_all_tiles": t_gemm_full, "whole_list": t_all}, "sparse_list_ms": t_sparse, "mirror_and_order_rows_ms": t_rows,
                  "TFLOPs_lower": 2 * a.shape[0] ** 3 / 2 / (t_gemm * 1e-3) / 1e12, "same_list": bool(torch.equal(torch.sort(sk).values, keys))}))
os.chdir("/tmp")
for dense in (True, False):
    candidates.DENSE_MIN_DENSITY = 0.03 if dense else 2.0
    for _ in range(2):
        filter_stage.main(["--dataset", "ddi", "--model", "simple", "--checkpoint", "ddi_simple||0|0.pt", "--synthetic"])
    print(json.dumps({"dense": dense, **{k: v for k, v in filter_stage.LAST_TIMING.items()}}))
