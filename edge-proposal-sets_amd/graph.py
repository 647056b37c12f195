"""CSR adjacency resident in HBM -- the build's stand-in for the slice of
``torch_sparse.SparseTensor`` the reference touches on the hot path.

Mirrors the calls the reference makes on ``adj_t``: ``SparseTensor.from_edge_index`` /
``to_symmetric`` / ``fill_value`` (rank.py:32-35), ``.to(device)`` / ``.cpu()`` (rank.py:300,
filter.py:96), ``.coo()`` (adamic_utils.py:9), ``.sparse_sizes()`` / ``.sum(-1)``
(models.py:546-547), ``to_scipy`` / ``from_scipy`` (filter.py:100-105).

Layout in HBM (what the kernels read): ``rowptr`` int64[N+1], ``col`` int32[nnz] ascending
inside each row, ``val`` float32[nnz] or ``None`` when every stored value is 1 (the common
case: every dataset but collab, rank.py:34-35) -- the kernels then skip the value stream
entirely.  The graph is always coalesced: duplicate (row, col) entries are summed, which is
what ``to_symmetric`` (reduce="sum") does in the reference [torch_sparse, third-party].

Construction here is host-side plumbing on torch ops (sort / unique / cumsum on whichever
device the edge list lives on); the scoring kernels live in csrc/.
"""
from __future__ import annotations

import itertools
from typing import Optional, Tuple

import torch

_UID = itertools.count(1)


def _coalesce(row: torch.Tensor, col: torch.Tensor, val: Optional[torch.Tensor], n_rows: int, n_cols: int):
    """Sort by (row, col) and sum duplicates.  -> (rowptr int64, col int32, val float32|None)."""
    row = row.to(torch.int64)
    col = col.to(torch.int64)
    key = row * n_cols + col
    if key.numel() == 0:
        rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=key.device)
        return rowptr, torch.zeros(0, dtype=torch.int32, device=key.device), \
            (None if val is None else torch.zeros(0, dtype=torch.float32, device=key.device))
    if val is None:
        ukey = torch.unique(key)  # sorted
        uval = None
    else:
        ukey, inv = torch.unique(key, return_inverse=True)
        uval = torch.zeros(ukey.numel(), dtype=torch.float32, device=key.device)
        uval.index_add_(0, inv, val.to(torch.float32))
    urow = torch.div(ukey, n_cols, rounding_mode="floor")
    ucol = (ukey - urow * n_cols).to(torch.int32)
    counts = torch.bincount(urow, minlength=n_rows)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=key.device)
    torch.cumsum(counts, 0, out=rowptr[1:])
    return rowptr, ucol, uval


class CSRGraph:
    """Coalesced CSR matrix [n_rows, n_cols]; float32 values or implicit ones."""

    def __init__(self, rowptr: torch.Tensor, col: torch.Tensor, val: Optional[torch.Tensor], n_rows: int,
                 n_cols: int):
        assert rowptr.dtype == torch.int64 and col.dtype == torch.int32
        assert val is None or (val.dtype == torch.float32 and val.numel() == col.numel())
        assert rowptr.numel() == n_rows + 1
        self.rowptr = rowptr.contiguous()
        self.col = col.contiguous()
        self.val = None if val is None else val.contiguous()
        self.n_rows = int(n_rows)
        self.n_cols = int(n_cols)
        self._cache = {}
        self.uid = next(_UID)   # identity for caches keyed on the adjacency (id() can be recycled after a free)

    def weight_cached(self, name: str, node_w: Optional[torch.Tensor], build):
        """Cache of a table derived from this graph AND a weight tensor.  An entry holds the weight tensor ITSELF (a strong
        reference, compared with ``is``) and its version counter -- never its address: a freed temporary's address is handed
        to the next tensor of the same size, again at version 0, and a table built for other weights would be served.  At
        most four weight tensors are remembered per graph and table."""
        entries = self._cache.setdefault(("by_weight", name), [])
        ver = None if node_w is None else node_w._version
        for i, (t, v, val) in enumerate(entries):
            if t is node_w and v == ver:
                if i:
                    entries.insert(0, entries.pop(i))
                return val
        val = build()
        entries.insert(0, (node_w, ver, val))
        del entries[4:]
        return val

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_edge_index(cls, edge_index: torch.Tensor, edge_attr: Optional[torch.Tensor] = None,
                        sparse_sizes=None) -> "CSRGraph":
        """rank.py:32 ``SparseTensor.from_edge_index(full_edge_index, full_edge_weights, sparse_sizes=[N,N])``.
        Duplicates are summed here already (the reference sums them one call later, in to_symmetric)."""
        row, col = edge_index[0], edge_index[1]
        if sparse_sizes is None:
            n = int(edge_index.max().item()) + 1 if edge_index.numel() else 0
            sparse_sizes = (n, n)
        rowptr, c, v = _coalesce(row, col, edge_attr, int(sparse_sizes[0]), int(sparse_sizes[1]))
        return cls(rowptr, c, v, int(sparse_sizes[0]), int(sparse_sizes[1]))

    @classmethod
    def from_scipy(cls, A, device=None, keep_values: bool = True) -> "CSRGraph":
        A = A.tocsr()
        A.sum_duplicates()
        A.sort_indices()
        rowptr = torch.from_numpy(A.indptr.astype("int64"))
        col = torch.from_numpy(A.indices.astype("int32"))
        val = torch.from_numpy(A.data.astype("float32")) if keep_values else None
        g = cls(rowptr, col, val, A.shape[0], A.shape[1])
        if val is not None and bool((val == 1).all()):
            g = cls(rowptr, col, None, A.shape[0], A.shape[1])
        return g.to(device) if device is not None else g

    # ------------------------------------------------------------------ torch_sparse-like API
    def to_symmetric(self) -> "CSRGraph":
        """rank.py:33: concatenate with the transpose and coalesce with reduce='sum'."""
        n = max(self.n_rows, self.n_cols)
        row, col, val = self.coo()
        r2 = torch.cat([row, col.to(torch.int64)])
        c2 = torch.cat([col.to(torch.int64), row])
        # a value-less SparseTensor stays value-less (duplicates just merge) [torch_sparse]
        v2 = None if val is None else torch.cat([val, val])
        rowptr, c, v = _coalesce(r2, c2, v2, n, n)
        g = CSRGraph(rowptr, c, v, n, n)
        if v is not None and bool((v == 1).all()):
            g.val = None
        return g

    def fill_value(self, value: float = 1.0) -> "CSRGraph":
        """rank.py:35 ``adj_t.fill_value(1.)``."""
        if value == 1.0:
            return CSRGraph(self.rowptr, self.col, None, self.n_rows, self.n_cols)
        return CSRGraph(self.rowptr, self.col, torch.full((self.nnz(),), float(value), dtype=torch.float32,
                                                          device=self.col.device), self.n_rows, self.n_cols)

    def set_value(self, value: Optional[torch.Tensor], layout=None) -> "CSRGraph":
        return CSRGraph(self.rowptr, self.col, value, self.n_rows, self.n_cols)

    def has_value(self) -> bool:
        return self.val is not None

    def to(self, device) -> "CSRGraph":
        device = torch.device(device)
        if device == self.rowptr.device:
            return self
        return CSRGraph(self.rowptr.to(device), self.col.to(device), None if self.val is None else self.val.to(device),
                        self.n_rows, self.n_cols)

    def cpu(self) -> "CSRGraph":
        return self.to("cpu")

    def cuda(self, index: int = 0) -> "CSRGraph":
        return self.to(f"cuda:{index}")

    @property
    def device(self) -> torch.device:
        return self.rowptr.device

    def nnz(self) -> int:
        return int(self.col.numel())

    def sparse_sizes(self) -> Tuple[int, int]:
        return (self.n_rows, self.n_cols)

    def size(self, dim: int) -> int:
        return self.sparse_sizes()[dim]

    def row_index(self) -> torch.Tensor:
        """int64 row id of every stored entry (CSR -> COO expansion)."""
        if "row" not in self._cache:
            counts = self.rowptr[1:] - self.rowptr[:-1]
            self._cache["row"] = torch.repeat_interleave(
                torch.arange(self.n_rows, device=self.device, dtype=torch.int64), counts)
        return self._cache["row"]

    def coo(self):
        """adamic_utils.py:9 ``row, col, val = adj.coo()`` (col as int64 like torch_sparse)."""
        return self.row_index(), self.col.to(torch.int64), self.val

    def csr(self):
        return self.rowptr, self.col, self.val

    def values_or_ones(self) -> torch.Tensor:
        return self.val if self.val is not None else torch.ones(self.nnz(), dtype=torch.float32, device=self.device)

    def degree(self) -> torch.Tensor:
        return (self.rowptr[1:] - self.rowptr[:-1])

    def sum(self, dim: int = -1) -> torch.Tensor:
        """Row sums (dim=-1/1) or column sums (dim=0) as float32 (models.py:547 ``adj.sum(-1)``)."""
        v = self.values_or_ones()
        if dim in (-1, 1):
            out = torch.zeros(self.n_rows, dtype=torch.float32, device=self.device)
            out.index_add_(0, self.row_index(), v)
        else:
            out = torch.zeros(self.n_cols, dtype=torch.float32, device=self.device)
            out.index_add_(0, self.col.to(torch.int64), v)
        return out

    def to_scipy(self, layout: str = "csr"):
        import scipy.sparse as ssp
        g = self.cpu()
        A = ssp.csr_matrix((g.values_or_ones().numpy(), g.col.numpy(), g.rowptr.numpy()), shape=(g.n_rows, g.n_cols))
        return A.tocsc() if layout == "csc" else A

    # ------------------------------------------------------------------ derived graphs (cached)
    def with_self_loops(self, fill: float = 1.0) -> "CSRGraph":
        """torch_sparse.fill_diag [third-party]: SET every diagonal entry to ``fill`` (insert if absent)."""
        key = ("loops", float(fill))
        if key not in self._cache:
            row, col, val = self.coo()
            off = row != col
            n = min(self.n_rows, self.n_cols)
            diag = torch.arange(n, device=self.device, dtype=torch.int64)
            r2 = torch.cat([row[off], diag])
            c2 = torch.cat([col[off], diag])
            v_off = self.values_or_ones()[off]
            v2 = torch.cat([v_off, torch.full((n,), float(fill), dtype=torch.float32, device=self.device)])
            rowptr, c, v = _coalesce(r2, c2, v2, self.n_rows, self.n_cols)
            self._cache[key] = CSRGraph(rowptr, c, v, self.n_rows, self.n_cols)
        return self._cache[key]

    def degree_ordered(self):
        """(graph relabelled hubs-first, perm, inv): new id i is old id perm[i], inv[perm] = arange.  The SpMM gathers one
        row of X per stored entry; with the most-referenced rows (the hubs: 5 % of the nodes hold ~37 % of the entries of
        the ppa-like graph) packed at the front of X they stay cache-resident: 6.25 -> 5.60 ms per layer (tools/
        spmm_reorder.py).  Square graphs only; cached."""
        if "deg_order" not in self._cache:
            assert self.n_rows == self.n_cols
            if self.device.type == "cuda" and self.nnz() < 1 << 31 and self.n_rows:
                # Row i of the copy IS row perm[i] with its ids mapped: order the nodes (a stable radix sort of the degrees),
                # gather the rows, sort INSIDE each (eps_node_order + eps_relabel_graph: 9 -> 1.4 ms for the 42.5 M entries of
                # the ppa-like graph against the 64-bit sort of all entries below; and no torch operator on the way -- a
                # one-shot filter.py pays ~10 ms for each one it is the first to use)
                from . import ops
                perm, inv32, rowptr = ops.node_order(rowptr=self.rowptr, relabel=True)
                c, v = ops.relabel_graph(self.rowptr, self.col, self.val, perm, inv32, rowptr)
                inv = inv32.to(torch.int64)
            else:
                perm = torch.argsort(self.degree(), descending=True, stable=True)
                inv = torch.empty_like(perm)
                inv[perm] = torch.arange(self.n_rows, device=self.device)
                row, col, val = self.coo()
                rowptr, c, v = _coalesce(inv[row], inv[col], self.values_or_ones() if val is not None else None,
                                         self.n_rows, self.n_cols)
            self._cache["deg_order"] = (CSRGraph(rowptr, c, v, self.n_rows, self.n_cols), perm, inv)
        return self._cache["deg_order"]

    def gcn_normalized(self) -> "CSRGraph":
        """gcn_norm of torch_geometric 1.7.0 GCNConv [third-party, restated]: A^ = A with diag := 1,
        val' = (val * deg^-1/2[row]) * deg^-1/2[col].  Computed once per adjacency on the GPU
        (eps_gcn_norm) and cached -- the reference recomputes it on every forward call."""
        if "gcn" not in self._cache:
            from . import ops
            g = self.with_self_loops(1.0)
            nv = ops.gcn_norm(g.rowptr, g.col, g.val)
            self._cache["gcn"] = CSRGraph(g.rowptr, g.col, nv, g.n_rows, g.n_cols)
        return self._cache["gcn"]

    def __repr__(self) -> str:
        return (f"CSRGraph(n_rows={self.n_rows}, n_cols={self.n_cols}, nnz={self.nnz()}, "
                f"values={'float32' if self.val is not None else 'implicit 1'}, device={self.device})")


def add_edges(dataset: str, edge_index: torch.Tensor, edge_weight: torch.Tensor, extra_edges: torch.Tensor,
              num_nodes: int) -> CSRGraph:
    """Drop-in for rank.py:28-36: concatenate the extra (proposal) edges with weight 1, build the
    adjacency, symmetrise with duplicate SUM, and reset values to 1 unless the dataset is collab."""
    full_edge_index = torch.cat([edge_index, extra_edges.to(edge_index.device)], dim=-1)
    new_edge_weight = torch.ones(extra_edges.shape[1], dtype=torch.float32, device=edge_index.device)
    full_edge_weights = torch.cat([edge_weight.to(torch.float32).to(edge_index.device), new_edge_weight], 0)
    adj_t = CSRGraph.from_edge_index(full_edge_index, full_edge_weights, sparse_sizes=[num_nodes, num_nodes])
    adj_t = adj_t.to_symmetric()
    if dataset != "collab":
        adj_t = adj_t.fill_value(1.)
    return adj_t
