"""CPU study (numpy / SciPy, no GPU): how many two-hop half paths of the ppa-like stand-in a threshold scan must walk when
every column v may DROP its path-heaviest rows w as long as their weights sum to at most beta x bar (score(u,v) <= walked sum
+ dropped weight: sound), and how many more screened survivors that costs.  A graph of the same generator family as
synth.ppa_like (CPU generator: not the same edges, the same statistics)."""
import sys, time
import numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, "/root/repo")

def rmat_edges(scale, n_edges, seed, a, b, c):
    gen = torch.Generator().manual_seed(seed)
    row = torch.zeros(n_edges, dtype=torch.int64); col = torch.zeros(n_edges, dtype=torch.int64)
    for _ in range(scale):
        p = torch.rand(n_edges, generator=gen)
        right = ((p >= a) & (p < a + b)) | (p >= a + b + c)
        down = p >= a + b
        row = row * 2 + down.to(torch.int64); col = col * 2 + right.to(torch.int64)
    return torch.stack([row, col])

def graph(n=576_289, m=21_231_931, seed=3):
    ei = rmat_edges(20, int(m * 1.003), seed, .45, .22, .22)
    perm = torch.randperm(1 << 20, generator=torch.Generator().manual_seed(seed + 1))
    ei = (perm[ei] % n).numpy()
    ei = ei[:, ei[0] != ei[1]]
    A = sp.coo_matrix((np.ones(ei.shape[1], np.float32), (ei[0], ei[1])), shape=(n, n)).tocsr()
    A = ((A + A.T) > 0).astype(np.float32).tocsr()
    return A

t0 = time.time()
A = graph()
n = A.shape[0]
deg = np.diff(A.indptr)
order = np.argsort(-deg, kind="stable")          # hubs first
inv = np.empty(n, np.int64); inv[order] = np.arange(n)
A = A[order][:, order].tocsr(); A.sort_indices()
deg = np.diff(A.indptr).astype(np.int64)
kind = sys.argv[1] if len(sys.argv) > 1 else "aa"
with np.errstate(divide="ignore"):
    wt = {"aa": 1.0 / np.log(deg.astype(np.float64)), "ra": 1.0 / deg.astype(np.float64), "cn": np.ones(n)}[kind]
wt[~np.isfinite(wt)] = 0.0
wt[deg == 0] = 0.0
print("graph", n, A.nnz, "max deg", deg.max(), "t", round(time.time() - t0, 1), flush=True)
rows = np.repeat(np.arange(n), deg)              # entry e = (v = rows[e], w = col[e])
col = A.indices.astype(np.int64)
# cost of entry (v, w): entries of row w below v  (= position of v in row w)
# position of v in row w: for symmetric A with sorted rows, entry (w, v) has index in row w; compute via sort of (col,rows)
key = col * n + rows
o = np.argsort(key, kind="stable")               # entries sorted by (w, v): this is row-major order of the transposed = same matrix
pos_in_row = np.empty(A.nnz, np.int64)
pos_in_row[o] = np.arange(A.nnz) - A.indptr[col[o]]
cost = pos_in_row                                # half paths through entry (v, w)
total = cost.sum()
print("half paths", total, flush=True)
S = np.bincount(rows, weights=wt[col], minlength=n)

# the bar: exact scores of every 512-th column of the heaviest-first order
hp = np.bincount(rows, weights=cost, minlength=n)
horder = np.argsort(-hp, kind="stable")
stride = 512
sample = np.sort(horder[stride // 2::stride])
D = sp.diags(wt.astype(np.float64))
L = sp.tril(A.astype(np.float64), -1).tocsr()    # L[v, u] for u < v
sc = (A[sample].astype(np.float64) @ D @ A.astype(np.float64)).tocsr()   # [sample, n] scores incl. u >= v and edges
# mask: u < v, non-edge
sc = sc.tocoo()
m = sc.col < sample[sc.row]
r, c, d = sc.row[m], sc.col[m], sc.data[m]
edge = np.asarray(A[sample[r], c]).ravel() > 0
r, c, d = r[~edge], c[~edge], d[~edge]
print("sample candidates", d.size, "of est total", d.size * stride, flush=True)
K = 4_000_000
for safety in (1.0, 2.0):
    mrank = int(safety * K / 2 / stride)
    bar = np.sort(d)[-mrank]
    print(f"bar at safety {safety}: {bar:.4f}  (sample rank {mrank})")
bar = np.sort(d)[-int(2.0 * K / 2 / stride)]
print("S(v) quantiles", np.quantile(S, [.1, .25, .5, .75, .9, .99]))
live = S >= bar
print("columns with S >= bar:", live.sum(), "of", n, "; half paths in those columns", hp[live].sum() / total)
# paths with BOTH endpoints live: entries (v,w) count of u<v in N(w) with S(u)>=bar
# under S-descending labels this is a prefix; here approximate with exact count via cumulative live flags per row
livecum = np.cumsum(live[col]) - live[col]       # exclusive prefix over the entry array (entry order = by row w then u)  -- entries of row w: indices indptr[w]..
# for entry (v, w) the entries of row w below v are indptr[w] .. indptr[w] + pos; count live among them
mirror = A.indptr[col] + pos_in_row              # index of entry (w, v) in row w
lc = np.concatenate([[0], np.cumsum(live[col])])
live_below = lc[mirror] - lc[A.indptr[col]]
print("paths with both endpoints live:", live_below[live[rows]].sum() / total, flush=True)

# tail dropping: per column v, drop rows w in descending cost order while the dropped weight stays <= beta * bar
for beta in (0.0, 0.25, 0.5, 0.75):
    budget = beta * bar
    # sort entries within each column by cost descending
    o2 = np.lexsort((-cost, rows))
    w_sorted = wt[col[o2]]
    csum = np.cumsum(w_sorted)
    start = A.indptr[:-1]
    base = np.concatenate([[0.0], csum])[start]
    cw = csum - np.repeat(base, deg)             # inclusive cumulative weight within the column in drop order
    dropped = cw <= budget
    keepcost = np.where(dropped, 0, np.where(live[rows[o2]], live_below[o2], 0)).sum()
    keepcost_nolive = np.where(dropped, 0, cost[o2]).sum()
    T = np.bincount(rows[o2], weights=np.where(dropped, w_sorted, 0.0), minlength=n)
    # survivors in the sample: walked sum >= bar - T(v)
    drop_flag = np.zeros(A.nnz, bool); drop_flag[o2] = dropped
    Ad = sp.csr_matrix((np.where(drop_flag, 0.0, 1.0), A.indices, A.indptr), shape=A.shape)
    scw = (Ad[sample] @ D @ A.astype(np.float64)).tocsr()
    walked = np.asarray(scw[r, c]).ravel()
    surv = (walked >= bar - T[sample[r]] - 1e-12).sum()
    print(f"beta {beta}: walked paths {keepcost_nolive / total:.3f} of all (with live-endpoint pruning too: {keepcost / total:.3f}); "
          f"sample survivors {surv} (x{stride} = {surv * stride / 1e6:.2f} M; exact above bar {int((d >= bar).sum())})", flush=True)
