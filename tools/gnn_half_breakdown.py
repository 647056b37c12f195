#!/usr/bin/env python3
"""configs[3] filter (ppa stand-in, GCN L=3 H=256), the half scheme block by block: list, decode, cut + keys -- where the
time beyond the decode itself goes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import candidates, datasets, filter_stage, models, ops, scan
cli = ["--num_layers", "3", "--hidden_channels", "256", "--dropout", "0.0", "--batch_size", "65536", "--use_feature", "1",
       "--use_learnable_embedding", "1"]
args = models.default_model_configs(filter_stage.make_parser().parse_args(["--dataset", "ppa", "--model", "gcn", "--checkpoint", "x", "--synthetic"] + cli))
_, _, _, data = datasets.get_data(args)
dev = torch.device("cuda:0")
data = data.to(dev)
torch.manual_seed(0)
model = models.build_model(args, data, dev).eval()
g = data.adj_t
def T(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return (time.perf_counter() - t), r
t_emb, _ = T(lambda: model.embeddings(data.x, g))
revpos, md, sp = scan.reverse_positions(g), scan.max_degree(g), scan.window_splits(g)
tot = {"list": 0.0, "decode": 0.0, "cut+keys": 0.0}
n_pairs = 0
bar = None
with torch.no_grad():
    for lo, hi in candidates.column_blocks(g):
        t, r = T(lambda: ops.expand_unit(g.rowptr, g.col, None, g.n_rows, lo, hi, md, sp, want_score=False, want_v=True,
                                         col_order=candidates.heaviest_first(g, lo, hi), revpos=revpos))
        tot["list"] += t
        pairs = r.pairs
        n_pairs += pairs.shape[1]
        t, sc = T(lambda: model(data.x, pairs, g).reshape(-1))
        tot["decode"] += t
        def cut():
            global bar
            if bar is None:
                bar = ops.kth_largest(sc, min(105000, sc.numel()))
            m = sc >= bar
            p2, s2 = pairs[:, m], sc[m]
            return (p2[1].to(torch.int64) << 32) | p2[0].to(torch.int64), s2
        t, _ = T(cut)
        tot["cut+keys"] += t
print(f"embeddings {t_emb:.2f} s; {n_pairs} unordered pairs; " + "; ".join(f"{k} {v:.2f} s" for k, v in tot.items()),
      f"; decode rate {n_pairs / tot['decode'] / 1e6:.1f} M pairs/s")
