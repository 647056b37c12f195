"""CPU oracle for the Filter-and-Rank edge-scoring hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module -- as the checker / the reported CPU baseline, never as the thing
shipped.  Nothing under ``edge-proposal-sets_amd/`` imports it.

Two layers:

* ``libeps_oracle.so`` (``eps_oracle.c``, built by ``oracle/Makefile``): scalar C
  restatement, one thread.
* SciPy/numpy restatements that mirror the reference operation-for-operation
  (``scipy_AA`` <- adamic_utils.py:13-25, ``scipy_RA`` <- train_and_eval.py:195-216,
  ``scipy_CN`` <- models.py:536-542).  These are what ``bench.py`` times as the
  reference's own CPU path (the reference itself cannot travel to the GPU box).

Parity status: pair scores and LinkPredictor decode are PINNED by golden vectors made
by importing the reference (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``).
GCNConv/SAGEConv arithmetic, ``SparseTensor`` graph construction and OGB Hits@K are
third-party (torch_geometric 1.7.0, torch_sparse, ogb 1.3.1; un-vendored, absent from
/root/reference): restated from their published behaviour, **parity unpinned**.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libeps_oracle.so")
_lib = None

W_AA, W_RA = 0, 1


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (seconds)."""
    src = os.path.join(_HERE, "eps_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libeps_oracle.so"])
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_hits_at_k.restype = ctypes.c_double
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _csr(rowptr, col, val):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int32)
    val = None if val is None else np.ascontiguousarray(val, dtype=np.float32)
    return rowptr, col, val


# --------------------------------------------------------------------------- C layer
def col_sums(rowptr, col, val, n: int) -> np.ndarray:
    rowptr, col, val = _csr(rowptr, col, val)
    out = np.empty(n, dtype=np.float32)
    lib().oracle_col_sums_f32(_p(rowptr), _p(col), _p(val), ctypes.c_int64(n), _p(out))
    return out


def node_weights(colsum: np.ndarray, mode: int) -> np.ndarray:
    colsum = np.ascontiguousarray(colsum)
    out = np.empty_like(colsum)
    fn = lib().oracle_node_weights_f32 if colsum.dtype == np.float32 else lib().oracle_node_weights_f64
    assert colsum.dtype in (np.float32, np.float64)
    fn(_p(colsum), ctypes.c_int64(colsum.shape[0]), ctypes.c_int(mode), _p(out))
    return out


def pair_scores(rowptr, col, val, node_w, u, v):
    """-> (count int32[E], cn float32[E], wsum float32[E] or None)."""
    rowptr, col, val = _csr(rowptr, col, val)
    u = np.ascontiguousarray(u, dtype=np.int32)
    v = np.ascontiguousarray(v, dtype=np.int32)
    n = u.shape[0]
    count = np.empty(n, dtype=np.int32)
    cn = np.empty(n, dtype=np.float32)
    ws = None
    if node_w is not None:
        node_w = np.ascontiguousarray(node_w, dtype=np.float32)
        ws = np.empty(n, dtype=np.float32)
    lib().oracle_pair_scores_f32(_p(rowptr), _p(col), _p(val), _p(node_w), _p(u), _p(v),
                                 ctypes.c_int64(n), _p(count), _p(cn), _p(ws))
    return count, cn, ws


def pair_scores_f64(rowptr, col, val, node_w, u, v):
    """float64 accumulate (filter.py:141 RA path) -> (count int32[E], wsum float64[E])."""
    rowptr, col, val = _csr(rowptr, col, val)
    u = np.ascontiguousarray(u, dtype=np.int32)
    v = np.ascontiguousarray(v, dtype=np.int32)
    node_w = np.ascontiguousarray(node_w, dtype=np.float64)
    n = u.shape[0]
    count = np.empty(n, dtype=np.int32)
    ws = np.empty(n, dtype=np.float64)
    lib().oracle_pair_scores_f64(_p(rowptr), _p(col), _p(val), _p(node_w), _p(u), _p(v),
                                 ctypes.c_int64(n), _p(count), _p(ws))
    return count, ws


def spmm_csr(rowptr, col, val, x, bias=None, relu=False, mean=False) -> np.ndarray:
    rowptr, col, val = _csr(rowptr, col, val)
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = rowptr.shape[0] - 1
    f = x.shape[1]
    bias = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    y = np.empty((n, f), dtype=np.float32)
    lib().oracle_spmm_csr_f32(_p(rowptr), _p(col), _p(val), _p(x), ctypes.c_int64(n), ctypes.c_int32(f),
                              _p(bias), ctypes.c_int(int(relu)), ctypes.c_int(int(mean)), _p(y))
    return y


def gcn_norm_values(rowptr, col, val) -> np.ndarray:
    rowptr, col, val = _csr(rowptr, col, val)
    out = np.empty(col.shape[0], dtype=np.float32)
    lib().oracle_gcn_norm_f32(_p(rowptr), _p(col), _p(val), ctypes.c_int64(rowptr.shape[0] - 1), _p(out))
    return out


def mlp_decode(h, u, v, weights: Sequence[np.ndarray], biases: Sequence[np.ndarray], apply_sigmoid=True):
    """LinkPredictor.forward (models.py:478-485), float64 accumulate. -> (logit, out) float32[E]."""
    h = np.ascontiguousarray(h, dtype=np.float32)
    u = np.ascontiguousarray(u, dtype=np.int32)
    v = np.ascontiguousarray(v, dtype=np.int32)
    ws = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
    bs = [np.ascontiguousarray(b, dtype=np.float32) for b in biases]
    L = len(ws)
    dims = np.array([ws[0].shape[1]] + [w.shape[0] for w in ws], dtype=np.int32)
    assert dims[-1] == 1 and dims[0] == h.shape[1]
    wp = (ctypes.c_void_p * L)(*[w.ctypes.data for w in ws])
    bp = (ctypes.c_void_p * L)(*[b.ctypes.data for b in bs])
    n = u.shape[0]
    logit = np.empty(n, dtype=np.float32)
    out = np.empty(n, dtype=np.float32)
    lib().oracle_mlp_decode(_p(h), ctypes.c_int32(h.shape[1]), _p(u), _p(v), ctypes.c_int64(n), wp, bp,
                            _p(dims), ctypes.c_int32(L), ctypes.c_int(int(apply_sigmoid)), _p(logit), _p(out))
    return logit, out


def hits_at_k(pos, neg, k: int) -> float:
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    neg = np.ascontiguousarray(neg, dtype=np.float32)
    return float(lib().oracle_hits_at_k(_p(pos), ctypes.c_int64(pos.shape[0]), _p(neg),
                                        ctypes.c_int64(neg.shape[0]), ctypes.c_int64(k)))


# ------------------------------------------------- SciPy mirrors of the reference ops
def scipy_AA(A, edge_index: np.ndarray, batch_size: int = 2000) -> np.ndarray:
    """Operation-for-operation mirror of adamic_utils.py:13-25 (A: scipy CSR, edge_index [2,E])."""
    multiplier = 1 / np.log(A.sum(0))                      # :15
    multiplier[np.isinf(multiplier)] = 0                   # :16
    A_ = A.multiply(multiplier).tocsr()                    # :17
    scores = []
    E = edge_index.shape[1]
    for s in range(0, E, batch_size):                      # :18-20 DataLoader(range(E), batch_size)
        src, dst = edge_index[0, s:s + batch_size], edge_index[1, s:s + batch_size]
        scores.append(np.array(np.sum(A[src].multiply(A_[dst]), 1)).flatten())   # :22
    out = np.concatenate(scores, 0) if scores else np.zeros(0)
    return out.astype(np.float32)                          # :25 torch.FloatTensor(scores)


def scipy_RA(A, link_list: np.ndarray, batch_size: int = 32768) -> np.ndarray:
    """Mirror of train_and_eval.py:195-216 (link_list is [m,2])."""
    w = 1 / A.sum(axis=0)                                  # :203
    w[np.isinf(w)] = 0                                     # :204
    D = A.multiply(w).tocsr()                              # :205
    link_index = link_list.T                               # :207
    scores = []
    E = link_index.shape[1]
    for s in range(0, E, batch_size):
        src, dst = link_index[0, s:s + batch_size], link_index[1, s:s + batch_size]
        scores.append(np.array(np.sum(A[src].multiply(D[dst]), 1)).flatten())    # :212
    out = np.concatenate(scores, 0) if scores else np.zeros(0)
    return out.astype(np.float32)                          # :216


def scipy_CN(A, edges: np.ndarray) -> np.ndarray:
    """CommonNeighborsPredictor('simple') (models.py:536-542): row-sum of adj[u] .* adj[v]."""
    prod = A[edges[0]].multiply(A[edges[1]])
    return np.asarray(prod.sum(1)).reshape(-1).astype(np.float32)


# ------------------------------------------------ graph construction (rank.py:28-36)
def add_edges_scipy(dataset: str, edge_index: np.ndarray, edge_weight: np.ndarray,
                    extra_edges: np.ndarray, num_nodes: int):
    """rank.py:28-36 with torch_sparse semantics restated [3p, parity unpinned]:
    ``SparseTensor.from_edge_index`` keeps duplicates; ``to_symmetric`` concatenates the
    transpose and coalesces with reduce='sum' (so duplicates are summed); ``fill_value(1.)``
    for every dataset but collab."""
    import scipy.sparse as ssp
    full = np.concatenate([edge_index, extra_edges], axis=1).astype(np.int64)
    w = np.concatenate([np.asarray(edge_weight, dtype=np.float32),
                        np.ones(extra_edges.shape[1], dtype=np.float32)])
    row = np.concatenate([full[0], full[1]])
    col = np.concatenate([full[1], full[0]])
    val = np.concatenate([w, w])
    A = ssp.coo_matrix((val, (row, col)), shape=(num_nodes, num_nodes)).tocsr()  # sums duplicates
    A.sum_duplicates()
    A.sort_indices()
    if dataset != "collab":
        A.data[:] = 1.0
    return A.astype(np.float32)


# ------------------------------------------------ candidates (filter.py:96-109)
def candidates_scipy(A):
    """2-hop non-edges: nonzeros of A@A, minus diagonal, minus known edges (A>0), in the
    reference's order.  torch_sparse.from_scipy on a CSC matrix yields (row, col) sorted by
    column then row [3p; SURVEY probe] -> column-major.  Returns (pairs [E,2] int64, a2 values)."""
    A2 = (A @ A).tocsc()
    A2.setdiag(0)
    A2 = A2.tolil()
    known = (A > 0).tocoo()
    A2[known.row, known.col] = 0
    A2 = A2.tocsc()
    A2.eliminate_zeros()
    A2.sort_indices()
    coo = A2.tocoo()        # CSC -> COO keeps column-major order
    order = np.lexsort((coo.row, coo.col))
    return np.stack([coo.row[order], coo.col[order]], 1).astype(np.int64), coo.data[order]


def candidates_scipy_columns(A, lo: int, hi: int):
    """``candidates_scipy`` restricted to the columns [lo, hi): the same steps of filter.py:96-109 on the column slice
    (A @ A)[:, lo:hi] -- for graphs whose full A @ A does not fit the test box.  -> (pairs [E,2] int64 column-major, values)."""
    Ac = A.tocsc()
    A2 = (A.tocsr() @ Ac[:, lo:hi]).tocoo()
    keep = A2.row != (A2.col + lo)                               # A2.setdiag(0)
    r, c, v = A2.row[keep], A2.col[keep], A2.data[keep]
    known = (Ac[:, lo:hi] > 0).tocoo()                           # A2[known] = 0
    n = A.shape[0]
    kk = np.unique(known.col.astype(np.int64) * n + known.row.astype(np.int64))
    key = c.astype(np.int64) * n + r.astype(np.int64)
    m = ~np.isin(key, kk) & (v != 0)                             # eliminate_zeros
    r, c, v = r[m], c[m], v[m]
    order = np.lexsort((r, c))
    return np.stack([r[order], c[order] + lo], 1).astype(np.int64), v[order]


# ------------------------------------------------ GNN forward restatements
def gcn_norm_dense(A_dense: np.ndarray) -> np.ndarray:
    """D^-1/2 (A with diag := 1) D^-1/2 in float64 (gcn_norm of torch_geometric 1.7.0 [3p] on a SparseTensor: fill_diag(1),
    row sums, pow(-0.5), inf -> 0, scale rows and columns).  The only text of this formula the reference itself holds is the
    commented-out pre-computation at email_data/mlp_common.py:274-280 (= reddit/mlp_common.py:280-286): set_diag -> sum(dim=1)
    -> pow(-0.5) -> inf := 0 -> dis.view(-1, 1) * adj * dis.view(1, -1); tests/test_oracle_golden.py evaluates exactly that
    expression against this function."""
    Ah = A_dense.astype(np.float64).copy()
    np.fill_diagonal(Ah, 1.0)
    deg = Ah.sum(1)
    with np.errstate(divide="ignore"):
        dis = deg ** -0.5
    dis[np.isinf(dis)] = 0
    return dis[:, None] * Ah * dis[None, :]


def gcn_dense_forward(A_dense: np.ndarray, x: np.ndarray, weights, biases) -> np.ndarray:
    """Independent dense-formula check for tiny graphs (float64):
    H' = D^-1/2 (A with diag:=1) D^-1/2 (X W) + b, ReLU between layers, none after the last
    (models.py:181-187; GCNConv per torch_geometric 1.7.0 [3p]; normalisation: ``gcn_norm_dense``, witnessed in-tree by
    email_data/mlp_common.py:274-280). weights[l] is [in,out]."""
    An = gcn_norm_dense(A_dense)
    h = x.astype(np.float64)
    for l, (W, b) in enumerate(zip(weights, biases)):
        h = An @ (h @ W.astype(np.float64)) + b.astype(np.float64)
        if l + 1 < len(weights):
            h = np.maximum(h, 0)
    return h


def sage_dense_forward(A_dense: np.ndarray, x: np.ndarray, w_l, b_l, w_r) -> np.ndarray:
    """SAGEConv stack (models.py:434-440; conv semantics witnessed by models.py:347-349,:358-384
    minus :366): out = lin_l(mean_{j in N(i)} x_j) + lin_r(x_i); mean ignores edge values, no
    self loop; lin_r has no bias.  w_l/w_r are torch Linear layout [out,in]."""
    M = (A_dense != 0).astype(np.float64)
    deg = np.maximum(M.sum(1), 1.0)
    h = x.astype(np.float64)
    for l in range(len(w_l)):
        agg = (M @ h) / deg[:, None]
        h = agg @ w_l[l].astype(np.float64).T + b_l[l].astype(np.float64) + h @ w_r[l].astype(np.float64).T
        if l + 1 < len(w_l):
            h = np.maximum(h, 0)
    return h


def with_self_loops(rowptr, col, val, fill: float = 1.0):
    """torch_sparse.fill_diag [3p]: set every diagonal entry to ``fill`` (insert if missing)."""
    import scipy.sparse as ssp
    n = len(rowptr) - 1
    data = np.ones(len(col), dtype=np.float32) if val is None else np.asarray(val, dtype=np.float32)
    A = ssp.csr_matrix((data, np.asarray(col), np.asarray(rowptr)), shape=(n, n)).tolil()
    A.setdiag(fill)
    A = A.tocsr()
    A.sort_indices()
    return A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data.astype(np.float32)


def gcn_forward_csr(rowptr, col, val, x, weights, biases) -> np.ndarray:
    """Float32 CSR GCN forward in the reference's operation order: transform, then aggregate
    with gcn_norm'ed values, then bias (GCNConv 1.7.0 [3p]); ReLU between layers."""
    rp, ci, va = with_self_loops(rowptr, col, val, 1.0)
    nv = gcn_norm_values(rp, ci, va)
    h = np.asarray(x, dtype=np.float32)
    for l, (W, b) in enumerate(zip(weights, biases)):
        xw = (h @ np.asarray(W, dtype=np.float32)).astype(np.float32)
        h = spmm_csr(rp, ci, nv, xw, bias=b, relu=(l + 1 < len(weights)))
    return h


def sage_forward_csr(rowptr, col, x, w_l, b_l, w_r) -> np.ndarray:
    h = np.asarray(x, dtype=np.float32)
    for l in range(len(w_l)):
        agg = spmm_csr(rowptr, col, None, h, mean=True)
        out = agg @ np.asarray(w_l[l], np.float32).T + np.asarray(b_l[l], np.float32) \
            + h @ np.asarray(w_r[l], np.float32).T
        h = np.maximum(out, 0).astype(np.float32) if l + 1 < len(w_l) else out.astype(np.float32)
    return h


# ------------------------------------------------ sort rule (filter.py:160-161)
def sort_desc_stable(scores: np.ndarray) -> np.ndarray:
    """Declared tie rule: score descending, then candidate index ascending
    (== torch.sort(descending=True, stable=True) on the reference's candidate order)."""
    return np.argsort(-scores.astype(np.float64), kind="stable")
