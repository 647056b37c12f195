#!/usr/bin/env python3
"""The threshold scan of filter.py's OWN ppa stand-in (datasets.py --synthetic: not bench.py's graph), as labelled vs under
hubs-first labels: geometry chosen, first scan of a fresh graph object and repeat scans, allocator warm."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd  # noqa: E401,F401
from eps_amd import datasets, filter_stage, models, ops, scan
from eps_amd.graph import CSRGraph
args = models.default_model_configs(filter_stage.make_parser().parse_args(
    ["--dataset", "ppa", "--model", "adamic_ogb", "--checkpoint", "x", "--synthetic", "--keep_top", "4000000"]))
args.device = 0
_, _, _, data = datasets.get_data(args)
dev = torch.device("cuda:0")
adj = data.adj_t.to(dev) if hasattr(data.adj_t, "to") else data.adj_t
w = filter_stage.fused_node_weights(args, adj, None)
sync = torch.cuda.synchronize
for relabel in (False, True, False, True):
    g = CSRGraph(adj.rowptr, adj.col, None, adj.n_rows, adj.n_cols)
    sync(); t0 = time.perf_counter()
    st = {}
    scan.scan_topk(g, w, 4_000_000, stats=st, relabel=relabel)
    sync(); t1 = time.perf_counter()
    scan.scan_topk(g, w, 4_000_000, relabel=relabel)
    sync(); t2 = time.perf_counter()
    gs, perm = scan.scan_graph(g)
    print(f"relabel={relabel}: first scan {1e3 * (t1 - t0):.1f} ms, second {1e3 * (t2 - t1):.1f} ms; scanned copy relabelled: {perm is not None}, "
          f"one-pass variant {scan.screen_variant(gs)}, heaviest column {int(scan.half_paths(gs).max())} half paths, "
          f"total {scan.total_half_paths(gs) / 1e9:.2f} G, N {g.n_rows}, nnz {g.nnz()}, candidates {st['candidates']}, launches {st['launches']}")
