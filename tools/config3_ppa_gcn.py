#!/usr/bin/env python3
"""BASELINE configs[3], filter half, at full size on the ppa stand-in: GCN (L=3, H=256, 58 features + 256 embedding,
random-init checkpoint) scores EVERY 2-hop non-edge with the fused MFMA decode; 210k proposals kept."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import candidates, datasets, filter_stage, models
if os.environ.get("MAX_PATHS"):
    candidates.DEFAULT_BLOCK_PATHS = int(os.environ["MAX_PATHS"])      # two-hop paths per column block (A/B of the block size)
os.makedirs("/tmp/cfg3", exist_ok=True); os.chdir("/tmp/cfg3")
cli = ["--num_layers", "3", "--hidden_channels", "256", "--dropout", "0.0", "--batch_size", "65536", "--use_feature", "1",
       "--use_learnable_embedding", "1"]   # the reference has no ppa defaults: flags come from the CLI
p = filter_stage.make_parser()
args = models.default_model_configs(p.parse_args(["--dataset", "ppa", "--model", "gcn", "--checkpoint", "x", "--synthetic"] + cli))
_, _, _, data = datasets.get_data(args)
torch.manual_seed(0)
m = models.build_model(args, data, torch.device("cpu"))
os.makedirs("models", exist_ok=True)
torch.save(m.state_dict(), "models/ppa_gcn||0|0.pt")
t0 = time.perf_counter()
f = filter_stage.main(["--dataset", "ppa", "--model", "gcn", "--checkpoint", "ppa_gcn||0|0.pt", "--synthetic", "--keep_top", "210000"] + cli)
print(f"FILTER (GCN + MLP decode, ppa stand-in, all candidates) wall {time.perf_counter() - t0:.2f} s -> {f}")
