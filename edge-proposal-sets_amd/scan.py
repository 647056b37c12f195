"""Exact top-K of the filter stage's candidate set without the candidate list (``filter.py --keep_top K``).

The reference generates every 2-hop non-edge (filter.py:96-109), scores all of them (:113-142), sorts all E rows
(:160-161) -- and rank.py:294 reads the first ``num_sorted_edge``.  Here the whole candidate set is still scored, but
by ``eps_filter_scan`` (csrc/filter_scan.hip), which reports only the candidates above a bar:

1. the bar is ESTIMATED from a column sample: every ``SAMPLE_STRIDE``-th column of the heaviest-first order is scanned
   with no bar at all, and the score that ``SAFETY x K`` candidates of the whole graph are expected to exceed is read
   off the sample (on the device: no host round trip);
2. ONE launch scans all columns against that bar;
3. the result is VERIFIED: if at least K candidates survived, the K best of them under the declared rule (score
   descending, then candidate order ascending) ARE the K best of the whole set -- exact, whatever the estimate was.
   Too few survivors (bar too high) or more than the list holds (bar too low) -> the bar is corrected from what was
   found and step 2 repeats.

Symmetry: the kernel scores each unordered pair {u, v} once (u < v) and the list is mirrored here; candidate order is
the reference's column-major order, i.e. ascending key (v << 32 | u).

Multi-GPU (one process per GPU): rank r scans the columns ``order[r::world]`` of the same heaviest-first order
(balanced to within one column of every weight class); the graph is replicated; the sample and therefore the bar are
computed redundantly and identically on every rank; the survivor lists are all-gathered (a few MB) and the selection
runs on every rank -- no collective on the data path.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import ops
from .graph import CSRGraph

SAMPLE_STRIDE = 256        # every n-th column (heaviest-first order) estimates the bar
SAFETY = 2.0               # aim at SAFETY x K survivors (the estimate has been within 10 % on the ppa-sized graphs; too few -> one more scan)
SMALL_SET = 1 << 25        # candidate sets with at most this many two-hop half paths are scanned without a bar
MAX_K = 1 << 30            # rows a survivor list (< 2^32 slots, handed out in chunks) can be asked for; beyond: block streaming
MAX_LAUNCHES = 8           # estimate -> scan -> correct rounds before giving up (two are the rule)
_CHUNK_SLACK = 8192 * 320  # survivor slots are handed out in chunks of 8192 per workgroup
RELABEL_MIN_NODES = 100_000  # graphs at least this large are scanned under hubs-first labels (see scan_graph)


def scan_available(g: CSRGraph) -> bool:
    """eps_filter_scan can take this graph: on the GPU, square, unit values, SYMMETRIC (the half scheme scores a pair once,
    in the column of its larger endpoint -- on an asymmetric pattern that would silently be a different sum)."""
    return (g.device.type == "cuda" and g.n_rows == g.n_cols and g.val is None and 0 < g.n_rows <= ops.filter_scan_max_nodes()
            and g.nnz() < 1 << 30 and is_symmetric(g))       # (any id space: wider ones are scanned in id windows)


def reverse_positions(g: CSRGraph) -> torch.Tensor:
    """revpos of the graph (cached); the same kernel pass leaves the half-path counts and the symmetry flag in the cache."""
    if "revpos" not in g._cache:
        rev, hp, flag = ops.reverse_positions(g.rowptr, g.col, with_stats=True)
        g._cache["revpos"], g._cache["half_paths"], g._cache["asymmetric_flag"] = rev, hp, flag
    return g._cache["revpos"]


def is_symmetric(g: CSRGraph) -> bool:
    """Whether the stored pattern is symmetric (cached): every entry (v, w) finds v at its position in row w (checked by the
    reverse-positions pass itself)."""
    if "symmetric" not in g._cache:
        reverse_positions(g)
        g._cache["symmetric"] = int(g._cache["asymmetric_flag"].item()) == 0
    return g._cache["symmetric"]


def half_paths(g: CSRGraph) -> torch.Tensor:
    """int64[N]: two-hop paths v - w - u with u < v per column v = the work of scanning it (cached; from the revpos pass)."""
    if "half_paths" not in g._cache:
        reverse_positions(g)
    return g._cache["half_paths"]


def column_order(g: CSRGraph) -> torch.Tensor:
    """int32[N]: all columns, heaviest first (a hub column is one workgroup's work for a long time: it must not start last)."""
    if "scan_order" not in g._cache:
        g._cache["scan_order"] = torch.argsort(half_paths(g), descending=True, stable=True).to(torch.int32)
    return g._cache["scan_order"]


def window_splits(g: CSRGraph):
    """Row split table for the id windows eps_filter_scan uses on this graph (None: one window); cached."""
    if "scan_splits" not in g._cache:
        win_ids, n_win = ops.filter_scan_windows(g.n_rows)
        g._cache["scan_splits"] = ops.row_window_splits(g.rowptr, g.col, win_ids, n_win)
    return g._cache["scan_splits"]


def max_degree(g: CSRGraph) -> int:
    if "max_degree" not in g._cache:
        g._cache["max_degree"] = int(g.degree().max().item()) if g.n_rows else 0
    return g._cache["max_degree"]


def fixed_weights(g: CSRGraph, node_w: torch.Tensor) -> torch.Tensor:
    key = ("fixw", node_w.data_ptr(), node_w._version)
    if key not in g._cache:
        g._cache[key] = ops.fixed_weights(node_w)
    return g._cache[key]


def scan_graph(g: CSRGraph, build: bool = False):
    """(graph to scan, perm): large graphs are scanned under hubs-first labels (``CSRGraph.degree_ordered``: new id i is
    old id perm[i]); perm is None when the graph is scanned as it is.  The symmetric scheme gives column v the endpoints
    u < v, so the labelling decides how the half paths spread over the columns: as generated the ppa-like graph has columns
    of up to 3.7 M half paths (several rounds of row descriptors, several windows of tiles, buckets far beyond the L2);
    hubs first no column has more than 64 k -- the same 8.35 G paths in uniform columns: 47.8 -> 44.8 ms per scan
    (tools/scan_ab.py RELABEL=1).  Scores do not depend on the labels (order-independent fixed-point sums).
    Relabelling sorts the stored entries once (~35 ms for 42.5 M): worth it for a graph that is scanned repeatedly, not
    for one scan -- so the relabelled copy is used when it exists (``build=True`` makes it; the GNN path builds the same
    copy for its SpMM) and a one-shot caller (filter.py) scans the graph as it is."""
    if g.n_rows < RELABEL_MIN_NODES or not (build or "deg_order" in g._cache):
        return g, None
    gs, perm, _ = g.degree_ordered()
    return gs, perm


def _scan_weights(g: CSRGraph, gs: CSRGraph, perm, node_w: torch.Tensor) -> torch.Tensor:
    """Fixed-point weight table in the labels of the scanned graph (cached on the original graph per weight tensor)."""
    if perm is None:
        return fixed_weights(g, node_w)
    key = ("fixw_relabelled", node_w.data_ptr(), node_w._version)
    if key not in g._cache:
        g._cache[key] = ops.fixed_weights(node_w[perm].contiguous())
    return g._cache[key]


def _original_keys(keys: torch.Tensor, perm) -> torch.Tensor:
    """Survivor keys (v << 32 | u, u < v in the scanned graph's labels) -> the same unordered pairs in the original labels."""
    if perm is None:
        return keys
    a, b = perm[keys & 0xFFFFFFFF], perm[keys >> 32]
    return (torch.maximum(a, b) << 32) | torch.minimum(a, b)


def _launch(g, fixw, columns, threshold, capacity, scores_only: bool = False) -> ops.Survivors:
    out = ops.Survivors(capacity, threshold, g.device, scores_only)
    if columns.numel():
        ops.filter_scan(g.rowptr, g.col, reverse_positions(g), fixw, g.n_rows, columns, out, max_degree(g), window_splits(g))
    return out


def estimate_bar(g: CSRGraph, fixw: torch.Tensor, k: int, stride: int = SAMPLE_STRIDE, safety: float = SAFETY):
    """Score bar (1-element float32 device tensor) that about ``safety * k`` directed candidates are expected to exceed,
    from a scan of every ``stride``-th column of the heaviest-first order; ``None`` = no bar (keep everything)."""
    order = column_order(g)
    sample = order[stride // 2::stride].contiguous()         # the middle of every weight stratum, not its heaviest column
    hp = half_paths(g)
    bound = int(hp[sample.long()].sum().item())          # unordered candidates of the sample <= its half paths
    if bound == 0:
        return None
    res = _launch(g, fixw, sample, float("-inf"), 2 * bound + _CHUNK_SLACK, scores_only=True)
    slots, n_cand = res.counts()                         # no bar: every candidate of the sample holds a slot
    m = int(safety * k / 2 / stride) + 1                 # unordered pairs of the SAMPLE above the bar we aim at
    if m >= n_cand or slots > res.capacity:
        return None
    return ops.kth_largest(res.scores(slots), m)         # radix select over the slots as they are (untouched ones: -inf)


def _gather_varlen(t: torch.Tensor, world: int):
    """All ranks' 1-D tensors (different lengths) concatenated in rank order."""
    from . import dist as epd
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    lens = [int(x.item()) for x in epd.all_gather_list(n)]
    mx = max(lens + [1])
    pad = torch.zeros(mx, dtype=t.dtype, device=t.device)
    pad[:t.numel()] = t
    parts = epd.all_gather_list(pad)
    return torch.cat([parts[r][:lens[r]] for r in range(world)])


def select_topk(keys: torch.Tensor, vals: torch.Tensor, k: int, n_nodes: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """The k best DIRECTED candidates (score descending, key ascending) of a list of unordered survivors (key = v << 32 | u,
    u < v; both orientations carry the pair's score).  -> (keys, scores), sorted.  On the device: eps_select_topk_cut / _rows
    (``n_nodes`` bounds the ids: fewer radix passes)."""
    if keys.is_cuda:
        bits = 32 if not n_nodes else max(1, min(32, int(n_nodes - 1).bit_length()))
        return ops.select_topk(keys.contiguous(), vals.contiguous(), k, bits)
    return select_topk_torch(keys, vals, k)


def select_topk_torch(keys: torch.Tensor, vals: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """The same selection in tensor ops (host tensors; the device path is checked against it).  The k-th best directed score
    is the ceil(k/2)-th best unordered one; every pair at or above it is mirrored, ties included, and the declared order
    decides among them."""
    k2 = (k + 1) // 2
    if vals.numel() > k2:
        kth = torch.sort(vals, descending=True).values[k2 - 1] if k2 >= 1 else float("inf")
        m = vals >= kth
        keys, vals = keys[m], vals[m]
    keys = torch.cat([keys, ((keys & 0xFFFFFFFF) << 32) | (keys >> 32)])
    vals = torch.cat([vals, vals])
    o = torch.argsort(keys)                                # candidate order ...
    keys, vals = keys[o], vals[o]
    o = torch.sort(vals, descending=True, stable=True).indices[:k]     # ... kept among equal scores
    return keys[o], vals[o]


def scan_topk(g: CSRGraph, node_w: torch.Tensor, k: int, rank: int = 0, world: int = 1, stats: Optional[dict] = None,
              relabel: bool = False):
    """Exact top-``k`` candidates of the whole graph: (pairs int64 [2,<=k] as (u; v), scores float32), best first.
    ``stats`` (optional dict) receives ``candidates`` (directed candidates scored), ``launches``, ``survivors``, ``bar``.
    ``relabel``: build the hubs-first copy of a large graph if it does not exist yet (see ``scan_graph``)."""
    if not scan_available(g):
        raise ops._lib.EpsError("scan_topk: graph not supported by eps_filter_scan (see scan_available)")
    k = int(k)
    if not 0 < k <= MAX_K:
        raise ops._lib.EpsError(f"scan_topk: k = {k} outside (0, {MAX_K}] (longer lists: the block-streaming filter path)")
    dev = g.device
    g0 = g
    g, perm = scan_graph(g0, relabel)    # from here on g is the graph as scanned; ids go back through perm at the end
    fixw = _scan_weights(g0, g, perm, node_w)
    order = column_order(g)
    mine = order if world == 1 else order[rank::world].contiguous()
    total_half = int(half_paths(g).sum().item())
    launches = 0
    if total_half <= SMALL_SET:
        bar = None
    else:
        bar = estimate_bar(g, fixw, k)
        launches += 1
    expect = (2 * total_half if bar is None else int(2 * SAFETY * k)) // world
    capacity = min(2 * expect + _CHUNK_SLACK, (1 << 32) - 1)
    while True:
        if launches >= MAX_LAUNCHES:
            raise ops._lib.EpsError(f"scan_topk: no usable bar after {launches} launches (k = {k}, capacity {capacity})")
        res = _launch(g, fixw, mine, float("-inf") if bar is None else bar, capacity)
        launches += 1
        slots, n_cand = res.counts()
        keys, vals = res.valid(slots)
        overflow = slots > res.capacity
        n_surv = keys.numel()
        if world > 1:
            import torch.distributed as dist
            from . import dist as epd
            agg = torch.tensor([n_surv, int(overflow), n_cand], dtype=torch.int64, device=dev)
            tot = sum(epd.all_gather_list(agg))
            n_all, any_overflow, n_cand_all = int(tot[0]), int(tot[1]) > 0, int(tot[2])
        else:
            n_all, any_overflow, n_cand_all = n_surv, overflow, n_cand
        if any_overflow:
            # the bar was too low for the list.  What was kept is a subset of the survivors: its (k/2)-th best score
            # is a lower bound of the final bar -- scan again just below it.
            need = max(1, (k // 2) // world)
            if vals.numel() >= need:
                local = ops.kth_largest(vals, need)
            else:
                local = torch.full((1,), float("-inf"), device=dev)
            if world > 1:
                local = torch.stack(epd.all_gather_list(local)).min(0).values
            bar = torch.nextafter(local, torch.full_like(local, float("-inf")))
            capacity = min(4 * capacity, (1 << 32) - 1)
            continue
        if bar is not None and 2 * n_all < min(k, 2 * n_cand_all):
            # fewer than k above the bar: lower it (a quarter of the sample rank each time, then no bar at all)
            bar = None if launches > 3 else estimate_bar(g, fixw, k, safety=SAFETY * 8 ** (launches - 1))
            launches += 1
            if bar is None:
                capacity = min(2 * (2 * total_half // world) + _CHUNK_SLACK, (1 << 32) - 1)
            continue
        break
    keys = _original_keys(keys, perm)
    if world > 1:
        keys, vals = _gather_varlen(keys, world), _gather_varlen(vals, world)
    keys, vals = select_topk(keys, vals, k, g.n_rows)
    if stats is not None:
        stats.update(candidates=2 * n_cand_all, launches=launches, survivors=2 * n_all,
                     bar=None if bar is None else float(bar.item()))
    return torch.stack([keys & 0xFFFFFFFF, keys >> 32]), vals
