import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, eps_amd
from eps_amd import candidates, ops, synth
from eps_amd.heuristics import node_weight_table
dev = torch.device("cuda:0")
g = synth.ppa_like(seed=3, device=dev)
w = node_weight_table(g, ops.W_AA)
blocks = list(candidates.column_blocks(g))[:8]
mp = candidates.max_paths_of(g)
def run():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for lo, hi in blocks:
        r = ops.expand_candidates(g.rowptr, g.col, None, w, g.n_rows, lo, hi, want_cn=False, want_v=False,
                                  col_order=candidates.heaviest_first(g, lo, hi), max_paths=mp)
        del r
    torch.cuda.synchronize(); return time.perf_counter() - t0
for rep in range(6):
    t = [run() for _ in range(2)]
    print(f"scratch #{rep}: ptr {ops._EXPAND_WS[('cuda', 0)].data_ptr():#x}  {t[0]*1e3:.1f} ms {t[1]*1e3:.1f} ms", flush=True)
    ops._EXPAND_WS.clear(); torch.cuda.empty_cache()
    hog = [torch.empty(1 << 30, dtype=torch.uint8, device=dev) for _ in range(rep * 7)]   # shift where the next scratch lands
    ops._expand_scratch(dev, int(ops._lib.load().eps_expand_workspace_bytes(mp)))
    del hog
