"""Candidate generation of the filter stage (filter.py:96-109): every 2-hop non-edge (u, v) --
(A @ A)[u,v] != 0, u != v, A[u,v] == 0 -- in the reference's order: column-major (v ascending,
then u ascending), both directions present.

This is SURVEY 8(f) row 1 (the step immediately before the hot path).  Column blocks are streamed,
so the full candidate set never has to exist at once (the reference materialises A @ A on the host:
the memory wall on ppa).  On the GPU a block is produced by the fused expansion kernels
(csrc/expand_score.hip: candidates + common-neighbour counts + weighted scores in one pass over the
2-hop paths; id spaces wider than the LDS bitmap in id windows); only CPU tensors (host-logic tests)
take the tensor-op expansion ``two_hop_block``.
"""
from __future__ import annotations

from typing import Iterator, Optional, Tuple

import torch

from .graph import CSRGraph


def two_hop_block(g: CSRGraph, v_lo: int, v_hi: int) -> torch.Tensor:
    """Candidates whose column v lies in [v_lo, v_hi): int64 [2, E_blk] = (u; v), column-major."""
    dev = g.device
    rowptr, col = g.rowptr, g.col.to(torch.int64)
    n = g.n_rows
    deg = rowptr[1:] - rowptr[:-1]
    vs = torch.arange(v_lo, v_hi, device=dev, dtype=torch.int64)
    cnt1 = deg[vs]
    v1 = torch.repeat_interleave(vs, cnt1)                                   # (v, w): w in N(v)
    off1 = torch.arange(v1.numel(), device=dev) - torch.repeat_interleave(torch.cumsum(cnt1, 0) - cnt1, cnt1)
    w1 = col[rowptr[v1] + off1]
    cnt2 = deg[w1]
    v2 = torch.repeat_interleave(v1, cnt2)                                   # (v, u): u in N(w)
    w2 = torch.repeat_interleave(w1, cnt2)
    off2 = torch.arange(v2.numel(), device=dev) - torch.repeat_interleave(torch.cumsum(cnt2, 0) - cnt2, cnt2)
    u2 = col[rowptr[w2] + off2]
    key = torch.unique(v2 * n + u2)                                          # column-major, sorted, deduplicated
    vv = torch.div(key, n, rounding_mode="floor")
    uu = key - vv * n
    keep = (uu != vv) & ~torch.isin(key, v1 * n + w1)                        # no diagonal, no known edge
    return torch.stack([uu[keep], vv[keep]])


def hip_expand_available(g: CSRGraph, scored: bool = True) -> bool:
    """The fused expansion kernels can take this graph: on the GPU and square (any number of ids: id spaces wider than
    the LDS bitmap are expanded in id windows).  ``scored`` (common-neighbour counts / weighted sums out of the same
    expansion) additionally needs the per-workgroup path buckets of the heaviest column to fit the scratch budget --
    a hub whose neighbours are hubs (RMAT-24) does not: such graphs get their candidates from the expansion and their
    scores from the pair kernels."""
    if g.device.type != "cuda" or g.n_rows != g.n_cols or g.n_rows >= 1 << 31:
        return False
    if not scored:
        return True
    from . import ops
    return ops.expand_workspace_fits(max_paths_of(g))


FUSED_SCORE_LIMIT = float(1 << 22)   # the 2^-40 fixed-point accumulators of the fused kernels hold |sums| < 2^23; keep half


DENSE_MIN_DENSITY = 0.03     # stored entries / N^2 from which the common-neighbour list of a graph goes through the dense product
DENSE_MAX_NODES = 8192       # ... up to this many nodes (two N x N float matrices; N^3 flops)


def dense_cn_suits(g: CSRGraph) -> bool:
    """Unit-valued graph on the GPU, small and dense enough that CN = A A^T as ONE dense product on the matrix cores beats the
    sparse expansion (ogbl-ddi: N = 4,267, density 0.117)."""
    return bool(g.device.type == "cuda" and g.val is None and g.n_rows == g.n_cols and 0 < g.n_rows <= DENSE_MAX_NODES
                and g.nnz() >= DENSE_MIN_DENSITY * g.n_rows * g.n_rows)


def fused_score_bound(g: CSRGraph, node_w: Optional[torch.Tensor]) -> float:
    """Upper bound of every fused score sum_w A[u,w] * A[v,w] * node_w[w] of the graph:
    max_v sum_w |A[v,w]| * |node_w[w]| * max_u |A[u,w]|  (one pass over the stored entries; cached).  Unit-valued graphs
    stay far below the accumulators' range (AA <= 1.45 x degree); collab-like multi-edge weights of 10^3..10^4 do not."""
    def build() -> float:
        if g.device.type == "cuda" and (node_w is None or node_w.dtype == torch.float32):
            # one pass of the library (eps_score_bound: float64 row sums, no prefix-sum differences, no gathers through tensor
            # ops: 1.1 -> ~0.2 ms on the ppa-like graph, and no torch operator a one-shot filter.py would be the first to load)
            from . import ops
            if g.n_rows == 0 or g.nnz() == 0:
                return 0.0
            return float(ops.score_bound(g.rowptr, g.col, g.val, node_w, g.n_rows, g.n_cols).item()) * (1.0 + 1e-9) + 1e-12
        # row sums through a float64 prefix sum over the stored entries (rows are contiguous in CSR): an index_add of doubles
        # over 42 M entries took 150-300 ms on the MI355X (atomics), the prefix sum a few; a bound may carry the 1e-9 of slack
        # that the differences of a long prefix sum need
        col = g.col.to(torch.int64)
        term = torch.ones(g.nnz(), dtype=torch.float64, device=g.device) if node_w is None else node_w[col].abs().to(torch.float64)
        if g.val is not None:
            a32 = g.val.abs()
            colmax = torch.zeros(g.n_cols, dtype=torch.float32, device=g.device)
            colmax.scatter_reduce_(0, col, a32, reduce="amax", include_self=True)      # (a maximum: exact in float32)
            term = term * a32.to(torch.float64) * colmax[col].to(torch.float64)
        if g.n_rows == 0 or g.nnz() == 0:
            return 0.0
        cs = torch.cat([torch.zeros(1, dtype=torch.float64, device=g.device), torch.cumsum(term, 0)])
        rowsum = cs[g.rowptr[1:]] - cs[g.rowptr[:-1]]
        both = torch.stack([rowsum.max(), cs[-1]]).tolist()                               # (one host read)
        # a difference of two prefix values carries the rounding of the TOTAL, not of the row: each prefix value is within
        # ~log2(nnz) x 2^-53 of the total (pairwise / blocked scans; 4x margin), so the slack is proportional to cs[-1]
        return both[0] * (1.0 + 1e-9) + both[1] * max(1, g.nnz()).bit_length() * 2.0 ** -52 * 4.0 + 1e-12
    return g.weight_cached("score_bound", node_w, build)    # (keyed on the tensor itself, not on its recyclable address)


def fused_scores_fit(g: CSRGraph, node_w: Optional[torch.Tensor]) -> bool:
    """Whether the fused kernels' fixed-point sums cannot overflow on this graph (else: score with the pair kernels)."""
    return fused_score_bound(g, node_w) < FUSED_SCORE_LIMIT


def path_counts(g: CSRGraph) -> torch.Tensor:
    """paths(v) = sum_{w in N(v)} deg(w): the 2-hop paths leaving column v = the cost of expanding it (cached)."""
    if "paths" not in g._cache:
        deg = g.rowptr[1:] - g.rowptr[:-1]
        paths = torch.zeros(g.n_rows, dtype=torch.int64, device=g.device)
        paths.index_add_(0, g.row_index(), deg[g.col.to(torch.int64)])
        g._cache["paths"] = paths
    return g._cache["paths"]


def max_paths_of(g: CSRGraph) -> int:
    """Two-hop paths of the heaviest column of the graph (cached): sizes the bucket scratch of the fused expansion."""
    if "max_paths" not in g._cache:
        g._cache["max_paths"] = int(path_counts(g).max().item()) if g.n_rows else 0
    return g._cache["max_paths"]


def heaviest_first(g: CSRGraph, v_lo: int, v_hi: int) -> torch.Tensor:
    """Columns of [v_lo, v_hi) by descending path count (int32, relative to v_lo): the hand-out order of the
    expansion kernels -- a hub column is one workgroup's work for milliseconds and must not start last."""
    return torch.argsort(path_counts(g)[v_lo:v_hi], descending=True, stable=True).to(torch.int32)


class ColumnBlock:
    """Candidates of the columns [v_lo, v_hi) in column-major order WITHOUT a materialised v array: candidate i has
    u = cand_u[i] and v = v_lo + (the column whose colptr range holds i).  The filter stage only ever needs the pairs
    of the few candidates that survive the top-K cut, so the expansion kernel skips one scattered 4-byte store per
    candidate and the block is 4 bytes per candidate smaller.

    ``counts`` is set when colptr is an UPPER-BOUND layout (segments sized by the two-hop path counts, so no counting
    pass ran): the arrays then hold padding between the columns -- cand_u -1, score -inf -- and ``padded`` is True."""

    def __init__(self, v_lo: int, colptr: torch.Tensor, cand_u: torch.Tensor, cn, score, counts=None, survivors=None):
        self.v_lo, self.colptr, self.cand_u, self.cn, self.score, self.counts = v_lo, colptr, cand_u, cn, score, counts
        self.padded = counts is not None
        self.survivors = survivors      # (positions ascending, scores) of the candidates above the kernel's cut, or None

    def numel(self) -> int:
        """Real candidates of the block (synchronises in the padded layout)."""
        return int(self.counts.sum().item()) if self.padded else self.cand_u.numel()

    def select(self, idx: torch.Tensor) -> torch.Tensor:
        """int64 [2, len(idx)] pairs (u; v) of the candidates ``idx`` (positions in the block's arrays)."""
        v = torch.searchsorted(self.colptr[1:], idx, right=True) + self.v_lo
        return torch.stack([self.cand_u[idx].long(), v])

    def valid(self) -> torch.Tensor:
        """Positions of the real candidates, ascending (everything, without padding)."""
        if not self.padded:
            return torch.arange(self.cand_u.numel(), device=self.cand_u.device)
        return torch.nonzero(self.cand_u >= 0).squeeze(1)

    def pairs(self) -> torch.Tensor:
        """All pairs, int64 [2, E] (small graphs / the unsorted full-list path)."""
        return self.select(self.valid())


def segment_bounds(g: CSRGraph):
    """Upper bound of every column's candidate count -- min(two-hop paths, N) -- as an exclusive prefix over the
    columns (int64[N+1], device and host copies, cached): the segment layout of the count-free expansion."""
    if "seg_ub" not in g._cache:
        ub = torch.clamp(path_counts(g), max=g.n_rows)
        pre = torch.zeros(g.n_rows + 1, dtype=torch.int64, device=g.device)
        torch.cumsum(ub, 0, out=pre[1:])
        g._cache["seg_ub"] = (pre, pre.cpu())
    return g._cache["seg_ub"]


def _unit_expand(g: CSRGraph, v_lo: int, v_hi: int, node_w, want_score: bool, want_v: bool):
    """The block through the scan-structured list kernels (``ops.expand_unit``: graphs without stored values, or the list
    only) -- same bits as ``ops.expand_candidates``, about half the time; None when the graph does not qualify."""
    from . import ops, scan
    if not (g.device.type == "cuda" and g.n_rows == g.n_cols and 0 < g.n_rows and g.nnz() < 1 << 30):
        return None
    if want_score and g.val is not None:
        return None
    return ops.expand_unit(g.rowptr, g.col, node_w if want_score else None, g.n_rows, v_lo, v_hi, scan.max_degree(g),
                           scan.window_splits(g), want_score=want_score, want_v=want_v,
                           col_order=heaviest_first(g, v_lo, v_hi))


def expand_block_lazy(g: CSRGraph, v_lo: int, v_hi: int, node_w: Optional[torch.Tensor] = None, want_cn: bool = False,
                      want_score: bool = False, count_free: bool = False, cut=None) -> ColumnBlock:
    """``expand_block`` for consumers that read the pairs of a few candidates only (HIP expansion required).
    ``count_free``: lay the columns out by the upper bound ``segment_bounds`` instead of running the counting pass (one
    walk over the two-hop paths and a host synchronisation less; the arrays are then padded).  Measured on the ppa-like
    graphs: the expansion itself gets 9 % faster, but every later pass over the block (top-K cut) reads paths/candidates
    = 1.3x to 2.8x more entries, which costs as much or more -- so the filter stage keeps the counted layout.
    ``cut=(threshold, capacity)``: let the kernel report the candidates above the streaming top-K's bar (see
    ``ops.expand_candidates``); with ``want_score=False`` the block then has no score array at all."""
    from . import ops
    if not want_cn and not count_free and cut is None:
        r = _unit_expand(g, v_lo, v_hi, node_w, want_score, want_v=False)
        if r is not None:
            return ColumnBlock(v_lo, r[0], r[1], None, r[4])
    kw = {}
    if count_free:
        pre, pre_host = segment_bounds(g)
        kw = dict(colptr_ub=(pre[v_lo:v_hi + 1] - pre[v_lo]).contiguous(), total_ub=int(pre_host[v_hi] - pre_host[v_lo]))
    r = ops.expand_candidates(g.rowptr, g.col, g.val, node_w, g.n_rows, v_lo, v_hi, want_cn=want_cn,
                              want_score=want_score, want_v=False, col_order=heaviest_first(g, v_lo, v_hi),
                              max_paths=max_paths_of(g), cut=cut, **kw)
    return ColumnBlock(v_lo, r[0], r[1], r[3], r[4], counts=r.counts, survivors=r.survivors)


def expand_block(g: CSRGraph, v_lo: int, v_hi: int, node_w: Optional[torch.Tensor] = None, want_cn: bool = False,
                 want_score: bool = False, long_pairs: bool = True):
    """Candidates of columns [v_lo, v_hi) of a SYMMETRIC adjacency, with (optionally) the common-neighbour count
    and sum_w A[u,w]*(A[v,w]*node_w[w]) of every candidate, from ONE fused expansion.
    -> (pairs int64 [2,E] column-major, cn int32[E] | None, score float32[E] | None).  ``long_pairs=False`` leaves the
    pairs in the int32 buffer the kernel wrote (no 8-byte copy of a list that may hold 10^8 candidates)."""
    from . import ops
    fused = hip_expand_available(g) and (node_w is None or not want_score or fused_scores_fit(g, node_w))
    if fused and not want_cn and (g.val is None or not want_score):
        r = _unit_expand(g, v_lo, v_hi, node_w, want_score, want_v=True)
        if r is not None:
            return (r.pairs.long() if long_pairs else r.pairs), None, r[4]
    if fused:
        r = ops.expand_candidates(g.rowptr, g.col, g.val, node_w, g.n_rows, v_lo, v_hi, want_cn=want_cn,
                                  want_score=want_score, col_order=heaviest_first(g, v_lo, v_hi),
                                  max_paths=max_paths_of(g))
        return (r.pairs.long() if long_pairs else r.pairs), r[3], r[4]
    if hip_expand_available(g, scored=False):      # candidates from the expansion kernels, scores from the pair kernels
        r = _unit_expand(g, v_lo, v_hi, None, False, want_v=True)
        if r is None:
            r = ops.expand_candidates(g.rowptr, g.col, None, None, g.n_rows, v_lo, v_hi, want_cn=False, want_score=False,
                                      col_order=heaviest_first(g, v_lo, v_hi), max_paths=0)
        pairs = r.pairs
    else:                                          # CPU tensors (host-logic tests)
        pairs = two_hop_block(g, v_lo, v_hi)
    cn = sc = None
    if (want_cn or want_score) and pairs.shape[1]:
        u, v = pairs[0].to(torch.int32).contiguous(), pairs[1].to(torch.int32).contiguous()
        cnt, _, ws = ops.pair_scores(g.rowptr, g.col, g.val, node_w if want_score else None, g.n_rows, u, v,
                                     want_count=want_cn, want_cn=False, grouped=True)
        cn, sc = cnt, ws
    return (pairs.long() if long_pairs else pairs), cn, sc


# Two-hop paths per launch.  A block never holds more candidates than paths, so every per-block count stays below 2^31; at
# ~0.76 candidates per path a full block is ~1.6e9 candidates = 13 GB of ids + scores on a 288 GB device.  Large blocks matter:
# one column is one workgroup's work from start to end, so a launch is never shorter than its heaviest column (7 M paths =
# ~10 ms on the ppa-like graph) -- at 2^29 paths per launch (r01) the 256 workgroups' average share was a quarter of that
# column and the list kernels spent a quarter of their time waiting for it (scored list of the whole graph: 194 -> 141 ms).
DEFAULT_BLOCK_PATHS = (1 << 31) - 1


def column_blocks(g: CSRGraph, max_paths: int = None) -> Iterator[Tuple[int, int]]:
    """Column ranges whose 2-hop path count stays below ``max_paths`` (bounds a block's memory: a block never
    holds more candidates than paths)."""
    max_paths = int(max_paths or DEFAULT_BLOCK_PATHS)
    cum = torch.cumsum(path_counts(g), 0).cpu()
    n = g.n_rows
    v = 0
    while v < n:
        base = int(cum[v - 1]) if v > 0 else 0
        hi = int(torch.searchsorted(cum, torch.tensor(base + max_paths), right=True))
        hi = min(max(hi, v + 1), n)
        yield v, hi
        v = hi


def iter_candidate_blocks(g: CSRGraph, max_paths: int = None) -> Iterator[Tuple[int, int, torch.Tensor]]:
    """Yield (v_lo, v_hi, pairs[2,E_blk]) over all columns."""
    for lo, hi in column_blocks(g, max_paths):
        yield lo, hi, expand_block(g, lo, hi)[0]


def all_candidates(g: CSRGraph, max_paths: int = None) -> torch.Tensor:
    """The whole candidate list [2,E] (small graphs / tests)."""
    blocks = [b for _, _, b in iter_candidate_blocks(g, max_paths)]
    return torch.cat(blocks, 1) if blocks else torch.zeros((2, 0), dtype=torch.int64, device=g.device)
