"""CPU study #2 (numpy / SciPy): the scheme as the kernel would run it.  Hubs-first labels; column v skips its first x_v rows
(= its heaviest hub neighbours: rows ascend by id), x_v = the longest prefix of row v with ids < n_hub whose weights sum to
<= beta x bar.  Reports the share of half paths still walked, and how many table slots pass the first test
(walked sum >= bar - T(v)) and so need the exact hub term."""
import sys, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, "/root/repo/tools")
import importlib.util
spec = importlib.util.spec_from_file_location("sim1", "/root/repo/tools/r05_tail_drop_sim.py")
src = open("/root/repo/tools/r05_tail_drop_sim.py").read().split("t0 = time.time()")[0]
exec(src)
kind = sys.argv[1] if len(sys.argv) > 1 else "aa"
A = graph(); n = A.shape[0]
deg = np.diff(A.indptr)
order = np.argsort(-deg, kind="stable")
A = A[order][:, order].tocsr(); A.sort_indices()
deg = np.diff(A.indptr).astype(np.int64)
with np.errstate(divide="ignore"):
    wt = {"aa": 1.0 / np.log(deg.astype(np.float64)), "ra": 1.0 / deg.astype(np.float64), "cn": np.ones(n)}[kind]
wt[~np.isfinite(wt)] = 0.0
rows = np.repeat(np.arange(n), deg); col = A.indices.astype(np.int64)
o = np.argsort(col * n + rows, kind="stable")
pos = np.empty(A.nnz, np.int64); pos[o] = np.arange(A.nnz) - A.indptr[col[o]]
cost = pos; total = cost.sum()
hp = np.bincount(rows, weights=cost, minlength=n)
horder = np.argsort(-hp, kind="stable"); stride = 512
sample = np.sort(horder[stride // 2::stride])
D = sp.diags(wt); Af = A.astype(np.float64)
sc = (Af[sample] @ D @ Af).tocoo()
m = sc.col < sample[sc.row]; r, c, d = sc.row[m], sc.col[m], sc.data[m]
edge = np.asarray(A[sample[r], c]).ravel() > 0; r, c, d = r[~edge], c[~edge], d[~edge]
K = 4_000_000
bar = np.sort(d)[-int(2.0 * K / 2 / stride)]
print(kind, "bar", bar, "candidates est", d.size * stride, flush=True)
for nh in (64, 256, 1024, 1 << 30):
    print("hub rows", nh, "carry", cost[col < nh].sum() / total, "of the half paths")
idx_in_row = np.arange(A.nnz) - np.repeat(A.indptr[:-1], deg)
csum = np.cumsum(wt[col]); base = np.concatenate([[0.0], csum])[A.indptr[:-1]]
cw = csum - np.repeat(base, deg)                      # inclusive prefix weight inside row v, in id order
import os
for nh in [int(x) for x in os.environ.get('NH','64,256,1024,1073741824').split(',')]:
    for beta in [float(x) for x in os.environ.get('BETA','0.25,0.5,0.625,0.75,0.875').split(',')]:
        dropped = (cw <= beta * bar) & (col < nh)      # a prefix of the row: both conditions are monotone along it
        walked = np.where(dropped, 0, cost).sum() / total
        T = np.bincount(rows, weights=np.where(dropped, wt[col], 0.0), minlength=n)
        Ad = sp.csr_matrix((np.where(dropped, 0.0, 1.0), A.indices, A.indptr), shape=A.shape)
        ws = np.asarray((Ad[sample] @ D @ Af).tocsr()[r, c]).ravel()
        passes = (ws >= bar - T[sample[r]] - 1e-12).sum()
        print(f"n_hub {nh:>10} beta {beta}: walked {walked:.3f}; first-test passes {passes * stride / 1e6:.1f} M (survivors {int((d >= bar).sum()) * stride / 1e6:.1f} M); "
              f"mean rows dropped {dropped.sum() / n:.1f}", flush=True)
