import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, eps_amd
from eps_amd import candidates, ops, proposals, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
w = node_weight_table(g, ops.W_AA)
blocks = list(candidates.column_blocks(g))
candidates.segment_bounds(g); candidates.max_paths_of(g)
def run(cf, topk):
    top = proposals.StreamingTopK(4_000_000)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for lo, hi in blocks:
        blk = candidates.expand_block_lazy(g, lo, hi, w, want_score=True, count_free=cf)
        if topk: top.push(blk, blk.score)
    torch.cuda.synchronize(); return time.perf_counter() - t0
for rep in range(2):
    for cf in (False, True):
        for topk in (False, True):
            print(f"count_free={cf} topk={topk}: {run(cf, topk):.3f} s", flush=True)
