#!/usr/bin/env python3
"""BASELINE configs[1] end to end on the collab stand-in at full size: GCN filter (random-init checkpoint; no trained
weights exist offline) -> 150k proposals -> CN ('simple') rank with Hits@{10,50,100}."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import datasets, filter_stage, models, rank_stage
os.makedirs("/tmp/cfg1", exist_ok=True); os.chdir("/tmp/cfg1")
keys = ["num_layers", "hidden_channels", "dropout", "batch_size", "lr", "epochs", "use_feature", "use_learnable_embedding"]
args = models.default_model_configs(argparse.Namespace(dataset="collab", model="gcn", synthetic=True, **{k: None for k in keys}))
_, _, _, data = datasets.get_data(args)
torch.manual_seed(0)
m = models.build_model(args, data, torch.device("cpu"))
os.makedirs("models", exist_ok=True)
torch.save(m.state_dict(), "models/collab_gcn||0|0.pt")
t0 = time.perf_counter()
f = filter_stage.main(["--dataset", "collab", "--model", "gcn", "--checkpoint", "collab_gcn||0|0.pt", "--synthetic", "--keep_top", "150000"])
t1 = time.perf_counter()
print(f"FILTER (GCN, collab stand-in) wall {t1 - t0:.2f} s -> {f}")
c = rank_stage.main(["--dataset", "collab", "--model", "simple", "--sorted_edge_path", os.path.basename(f), "--num_sorted_edge", "150000", "--runs", "1", "--synthetic"])
print(f"RANK (CN, 150k proposals) wall {time.perf_counter() - t1:.2f} s, curve point {c}")
