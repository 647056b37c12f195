import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, eps_amd
from eps_amd import ops, synth
dev = torch.device("cuda:0")
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / iters
for n, m in ((576_289, 21_231_931), (288_000, 10_600_000), (144_000, 5_300_000), (72_000, 2_650_000), (36_000, 1_325_000)):
    g = synth.ppa_like(seed=3, device=dev, n_nodes=n, n_undirected=m)
    x = torch.randn(n, 256, device=dev)
    ms = timeit(lambda: ops.spmm_csr(g.rowptr, g.col, None, x, mean=True))
    gath = g.nnz() * 256 * 4 + g.nnz() * 4 + n * 256 * 4
    print(json.dumps({"N": n, "nnz": g.nnz(), "X_MB": n * 1024 / 1e6, "ms": round(ms, 3), "gather_TBps": round(gath / ms / 1e9, 2)}), flush=True)
