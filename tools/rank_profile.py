import os, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, eps_amd
from eps_amd import filter_stage, rank_stage
os.makedirs("/tmp/ppa_ra", exist_ok=True); os.chdir("/tmp/ppa_ra")
f = filter_stage.main(["--dataset", "ppa", "--model", "resource_allocation", "--checkpoint", "ppa_resource_allocation||0|0.pt",
                       "--synthetic", "--keep_top", "4000000"])
pr = cProfile.Profile(); pr.enable()
t1 = time.perf_counter()
c = rank_stage.main(["--dataset", "ppa", "--model", "resource_allocation", "--sorted_edge_path", os.path.basename(f),
                     "--num_sorted_edge", "4000000", "--runs", "1", "--synthetic"])
torch.cuda.synchronize()
print(f"RANK wall {time.perf_counter() - t1:.2f} s")
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
