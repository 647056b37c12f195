#!/usr/bin/env python3
"""The published ppa recipe (README.md:11-17; submit_job.py:207-213) on the ppa stand-in at full size:
RA filter over every 2-hop non-edge -> 4,000,000 proposals -> RA rank (Hits@{10,100,200})."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, eps_amd
from eps_amd import filter_stage, rank_stage
os.makedirs("/tmp/ppa_ra", exist_ok=True); os.chdir("/tmp/ppa_ra")
t0 = time.perf_counter()
f = filter_stage.main(["--dataset", "ppa", "--model", "resource_allocation", "--checkpoint", "ppa_resource_allocation||0|0.pt",
                       "--synthetic", "--keep_top", "4000000"])
t1 = time.perf_counter()
print(f"FILTER wall {t1 - t0:.2f} s")
c = rank_stage.main(["--dataset", "ppa", "--model", "resource_allocation", "--sorted_edge_path", os.path.basename(f),
                     "--num_sorted_edge", "4000000", "--runs", "1", "--synthetic"])
print(f"RANK wall {time.perf_counter() - t1:.2f} s curve {c}")
