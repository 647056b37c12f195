"""Common-Neighbours / Adamic-Adar / Resource-Allocation pair scoring on the MI355X.

Drop-ins for the reference's heuristic entry points, same names, argument meaning and return
types:

* ``get_A(adj, num_nodes)``                          <- adamic_utils.py:8-11
* ``AA(A, edge_index, batch_size=2000)``             <- adamic_utils.py:13-25
* ``resource_allocation(adj_matrix, link_list, batch_size=32768)``  <- train_and_eval.py:195-216
* ``common_neighbors(adj, edges)``                   <- models.py:536-542 ('simple')

Where the reference loops over 2000-pair batches on one CPU thread through SciPy, these upload
the pair list once, run ``eps_pair_scores`` (csrc/pair_intersect.hip) over all of it and hand
back the same ``torch.FloatTensor``.  ``batch_size`` is accepted for signature compatibility;
it does not change results (the reference's batching does not either).  There is no CPU path.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from . import ops
from ._lib import EpsError
from .graph import CSRGraph

# pairs scored per launch when streaming a very large candidate list through HBM
_STREAM_CHUNK = 1 << 26


def _default_device() -> torch.device:
    if not torch.cuda.is_available():
        raise EpsError("no HIP device visible: the edge-proposal-sets_amd scoring path has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _as_graph(A, device: Optional[torch.device] = None) -> CSRGraph:
    """Accept what the reference passes as ``A``: the object returned by get_A, or a SciPy matrix
    (filter.py:136-139 builds one by hand, possibly int64)."""
    if isinstance(A, CSRGraph):
        g = A
    elif hasattr(A, "tocsr"):
        g = CSRGraph.from_scipy(A)
    else:
        raise TypeError(f"unsupported adjacency type {type(A)}")
    if not g.device.type == "cuda":
        g = g.to(device or _default_device())
    return g


def check_node_ids(edge_index: torch.Tensor, n_nodes: int, what: str = "edge list") -> None:
    """The kernels index rowptr / the embedding table with the ids they are given: an id outside [0, n_nodes) must stop
    here (the reference raises IndexError in the same case), before it is narrowed to int32 or reaches the device."""
    if edge_index.numel() == 0:
        return
    lo, hi = torch.aminmax(edge_index)
    lo, hi = int(lo), int(hi)
    if lo < 0 or hi >= n_nodes:
        raise EpsError(f"{what}: node ids must lie in [0, {n_nodes}), got [{lo}, {hi}]")


def _as_pairs(edge_index, device: torch.device, n_nodes: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """[2,E] integer tensor/array -> two contiguous int32 device vectors (ids range-checked against ``n_nodes``)."""
    if isinstance(edge_index, np.ndarray):
        edge_index = torch.from_numpy(edge_index)
    if edge_index.dim() != 2 or edge_index.size(0) != 2:
        raise EpsError(f"expected a [2,E] edge list, got {tuple(edge_index.shape)}")
    if n_nodes is not None:
        check_node_ids(edge_index, n_nodes)
    e = edge_index.to(device=device, dtype=torch.int32, non_blocking=True)
    return e[0].contiguous(), e[1].contiguous()


def get_A(adj: CSRGraph, num_nodes: int) -> CSRGraph:
    """adamic_utils.py:8-11.  The reference converts the SparseTensor to a host SciPy CSR (keeping
    edge values); here the adjacency simply stays in HBM -- the returned object is what AA() and
    resource_allocation() take as ``A``."""
    if adj.n_rows != num_nodes or adj.n_cols != num_nodes:
        raise EpsError(f"get_A: adjacency is {adj.sparse_sizes()}, expected [{num_nodes},{num_nodes}]")
    return adj if adj.device.type == "cuda" else adj.to(_default_device())


def node_weight_table(g: CSRGraph, mode: int, f64: bool = False) -> torch.Tensor:
    """K2: mult[w] = 1/log(colsum[w]) (AA) or 1/colsum[w] (RA), inf -> 0; cached on the graph."""
    key = ("node_w", mode, f64)
    if key not in g._cache:
        colsum = ops.col_sums(g.rowptr, g.col, g.val, g.n_cols, f64=f64)
        g._cache[key] = ops.node_weights(colsum, mode, f64=f64)
    return g._cache[key]


def pair_scores_streamed(g: CSRGraph, u: torch.Tensor, v: torch.Tensor, node_w: Optional[torch.Tensor],
                         want_count=False, want_cn=False, grouped=None):
    """Run eps_pair_scores over an arbitrarily long pair list in HBM-sized pieces."""
    n = u.numel()
    if n <= _STREAM_CHUNK:
        return ops.pair_scores(g.rowptr, g.col, g.val, node_w, g.n_rows, u, v, want_count=want_count, want_cn=want_cn,
                               grouped=grouped)
    outs = [[], [], []]
    for s in range(0, n, _STREAM_CHUNK):
        r = ops.pair_scores(g.rowptr, g.col, g.val, node_w, g.n_rows, u[s:s + _STREAM_CHUNK].contiguous(),
                            v[s:s + _STREAM_CHUNK].contiguous(), want_count=want_count, want_cn=want_cn,
                            grouped=grouped)
        for k in range(3):
            if r[k] is not None:
                outs[k].append(r[k])
    return tuple(torch.cat(o) if o else None for o in outs)


def AA(A, edge_index, batch_size: int = 2000, device_out: bool = False):
    """The Adamic-Adar heuristic score (adamic_utils.py:13-25).

    ``A``: adjacency from get_A (or a SciPy CSR).  ``edge_index``: LongTensor [2,E].
    Returns ``(torch.FloatTensor[E], edge_index)`` exactly like the reference; weighted when A
    carries non-unit values (collab): score = sum_w A[u,w] * (A[v,w] / log(colsum[w]))."""
    g = _as_graph(A)
    u, v = _as_pairs(edge_index, g.device, g.n_rows)
    w = node_weight_table(g, ops.W_AA)
    _, _, ws = pair_scores_streamed(g, u, v, w)
    return (ws if device_out else ws.cpu()), edge_index


def resource_allocation(adj_matrix, link_list, batch_size: int = 32768, device_out: bool = False):
    """Resource-Allocation similarity (train_and_eval.py:195-216); ``link_list`` is [m,2].

    An integer-typed SciPy adjacency (filter.py:130-139 builds int64 ones) makes the reference run
    in float64 before the final FloatTensor cast; that case takes the float64-accumulate kernel."""
    f64 = hasattr(adj_matrix, "dtype") and np.issubdtype(np.dtype(adj_matrix.dtype), np.integer)
    g = _as_graph(adj_matrix)
    if isinstance(link_list, np.ndarray):
        link_list = torch.from_numpy(link_list)
    u, v = _as_pairs(link_list.t(), g.device, g.n_rows)
    w = node_weight_table(g, ops.W_RA, f64=f64)
    _, _, ws = pair_scores_streamed(g, u, v, w)
    ws = ws.to(torch.float32)
    return ws if device_out else ws.cpu()


def common_neighbors(adj: CSRGraph, edges: torch.Tensor) -> torch.Tensor:
    """CN(u,v) = sum_w adj[u,w]*adj[v,w] (models.py:536-542); float32 on the adjacency's device."""
    g = _as_graph(adj)
    u, v = _as_pairs(edges, g.device, g.n_rows)
    _, cn, _ = pair_scores_streamed(g, u, v, None, want_cn=True)
    return cn
