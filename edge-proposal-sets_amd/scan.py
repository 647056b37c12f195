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

from typing import List, Optional, Tuple

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


def weight_cache(g: CSRGraph, name: str, node_w: Optional[torch.Tensor], build):
    """Per-graph cache of a table derived from a weight tensor (see ``CSRGraph.weight_cached``)."""
    return g.weight_cached(name, node_w, build)


def fixed_weights(g: CSRGraph, node_w: torch.Tensor) -> torch.Tensor:
    return weight_cache(g, "fixw", node_w, lambda: ops.fixed_weights(node_w))


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
    return weight_cache(g, "fixw_relabelled", node_w, lambda: ops.fixed_weights(node_w[perm].contiguous()))


def _original_keys(keys: torch.Tensor, perm) -> torch.Tensor:
    """Survivor keys (v << 32 | u, u < v in the scanned graph's labels) -> the same unordered pairs in the original labels."""
    if perm is None:
        return keys
    a, b = perm[keys & 0xFFFFFFFF], perm[keys >> 32]
    return (torch.maximum(a, b) << 32) | torch.minimum(a, b)


def _launch(g, fixw, columns, threshold, capacity, scores_only: bool = False, both: bool = False) -> ops.Survivors:
    out = ops.Survivors(capacity, threshold, g.device, scores_only, both)
    if columns.numel():
        ops.filter_scan(g.rowptr, g.col, reverse_positions(g), fixw, g.n_rows, columns, out, max_degree(g), window_splits(g))
    return out


def total_half_paths(g: CSRGraph) -> int:
    """Two-hop half paths of the whole graph (cached host number: an upper bound of its unordered candidates)."""
    if "total_half" not in g._cache:
        g._cache["total_half"] = int(half_paths(g).sum().item())
    return g._cache["total_half"]


def shard_columns(g: CSRGraph, rank: int, world: int) -> torch.Tensor:
    """This rank's columns of the heaviest-first order: order[rank::world] (cached)."""
    if world == 1:
        return column_order(g)
    key = ("scan_shard", rank, world)
    if key not in g._cache:
        g._cache[key] = column_order(g)[rank::world].contiguous()
    return g._cache[key]


def sample_columns(g: CSRGraph, stride: int, rank: int = 0, world: int = 1):
    """(this rank's share of the bar sample, its half paths, the whole sample's half paths) -- cached.  The sample is every
    ``stride``-th column of the heaviest-first order, taken from the middle of each stratum; rank r scans sample[r::world]."""
    key = ("scan_sample", stride, rank, world)
    if key not in g._cache:
        sample = column_order(g)[stride // 2::stride]
        mine = sample[rank::world].contiguous()
        hp = half_paths(g)
        both = torch.stack([hp[mine.long()].sum(), hp[sample.long()].sum()]).tolist()     # one read, once per graph
        g._cache[key] = (mine, int(both[0]), int(both[1]))
    return g._cache[key]


def estimate_bar(g: CSRGraph, fixw: torch.Tensor, k: int, stride: Optional[int] = None, safety: Optional[float] = None,
                 rank: int = 0, world: int = 1):
    """Score bar (1-element float32 device tensor) that about ``safety * k`` directed candidates are expected to reach,
    from a scan of every ``stride``-th column of the heaviest-first order; ``None`` = no bar (keep everything).  The bar is
    INCLUSIVE of the sample's m-th best score (one float below it: the kernel keeps scores strictly above its threshold), so
    a run of tied scores at the estimate passes as a whole.  ``world`` > 1: rank r scans its share of the sample and the m-th
    best of the union comes from all-reduced radix-select histograms (ops.kth_largest_dist) -- same bar on every rank, no
    host round trip after the first call on a graph."""
    stride = SAMPLE_STRIDE if stride is None else int(stride)          # module globals read at CALL time (tests patch them)
    safety = SAFETY if safety is None else float(safety)
    mine, bound_mine, bound_all = sample_columns(g, stride, rank, world)
    if bound_all == 0:
        return None
    m = int(safety * k / 2 / stride) + 1                 # unordered pairs of the SAMPLE above the bar we aim at
    if m >= bound_all:                                   # (candidates <= half paths: the sample cannot hold that many)
        return None
    res = _launch(g, fixw, mine, float("-inf"), min(2 * bound_mine + _CHUNK_SLACK, ops.SURVIVOR_SLOTS_MAX), scores_only=True)
    # no bar: every candidate of the sample holds a slot, untouched slots are -inf (fewer than m candidates -> bar -inf)
    kth = ops.kth_largest_dist(res.val, m, world)
    return torch.nextafter(kth, torch.full_like(kth, float("-inf")))


def _gather_varlen(t: torch.Tensor, world: int):
    """All ranks' 1-D tensors (different lengths) concatenated in rank order: one all-gather of the lengths (one host read),
    one padded all-gather of the data."""
    from . import dist as epd
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    lens = torch.cat(epd.all_gather_list(n)).tolist()
    return epd.gather_ragged(t, lens)


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


def _capacity(slots_wanted: int) -> int:
    cap = int(slots_wanted) + _CHUNK_SLACK
    if cap > ops.SURVIVOR_SLOTS_MAX:
        raise ops._lib.EpsError(f"scan_topk: a survivor list of {cap} slots exceeds the {ops.SURVIVOR_SLOTS_MAX} a launch can "
                                "address (slots are 32-bit positions); use the block-streaming filter path for this set")
    return cap


def scan_topk(g: CSRGraph, node_w: torch.Tensor, k: int, rank: int = 0, world: int = 1, stats: Optional[dict] = None,
              relabel: bool = False):
    """Exact top-``k`` candidates of the whole graph: (pairs int64 [2,<=k] as (u; v), scores float32), best first.
    ``stats`` (optional dict) receives ``candidates`` (directed candidates scored), ``launches``, ``survivors``, ``bar``.
    ``relabel``: build the hubs-first copy of a large graph if it does not exist yet (see ``scan_graph``).

    Per call, in the steady state (tables cached on the graph): sample launch -> bar (device) -> main launch -> job-wide
    ceil(k/2)-th best survivor score (device, ops.kth_largest_dist) -> compaction of the survivors at or above it -> ONE host
    read (slot counter, candidate count, selected count, cut; all-gathered when world > 1) -> the selected pairs of all ranks
    (about k / 2 / world each) gathered -> mirrored + ordered (ops.select_rows)."""
    if not scan_available(g):
        raise ops._lib.EpsError("scan_topk: graph not supported by eps_filter_scan (see scan_available)")
    k = int(k)
    if not 0 < k <= MAX_K:
        raise ops._lib.EpsError(f"scan_topk: k = {k} outside (0, {MAX_K}] (longer lists: the block-streaming filter path)")
    from . import candidates
    if not candidates.fused_scores_fit(g, node_w):
        raise ops._lib.EpsError(f"scan_topk: score bound {candidates.fused_score_bound(g, node_w):.3e} of this graph / weight "
                                "table leaves the 2^-40 fixed-point range of the scan (use the pair kernels)")
    dev = g.device
    g0 = g
    g, perm = scan_graph(g0, relabel)    # from here on g is the graph as scanned; ids go back through perm at the end
    fixw = _scan_weights(g0, g, perm, node_w)
    mine = shard_columns(g, rank, world)
    total_half = total_half_paths(g)
    k2 = (k + 1) // 2                    # the k-th best directed row belongs to the ceil(k/2)-th best unordered pair
    launches = 0
    bar = None
    if total_half > SMALL_SET and SAFETY * k < total_half:
        bar = estimate_bar(g, fixw, k, rank=rank, world=world)
        launches += 1
    capacity = _capacity(2 * ((2 * total_half if bar is None else int(2 * SAFETY * k)) // world))
    neg_inf = torch.full((1,), float("-inf"), device=dev)
    while True:
        if launches >= MAX_LAUNCHES:
            raise ops._lib.EpsError(f"scan_topk: no usable bar after {launches} launches (k = {k}, capacity {capacity})")
        res = _launch(g, fixw, mine, float("-inf") if bar is None else bar, capacity, both=True)
        launches += 1
        # the job-wide cut: the k2-th best survivor score over all ranks (-inf when fewer survived); untouched slots are -inf
        cut = ops.kth_largest_dist(res.val, k2, world)
        sel_k, sel_v, n_sel = ops.compact_at_least(res.key, res.val, cut)
        st = torch.cat([res.rec[[1, 4]], n_sel, cut.view(torch.int32).to(torch.int64)])      # slots, candidates, selected, cut bits
        if world > 1:
            from . import dist as epd
            table = torch.stack(epd.all_gather_list(st)).tolist()                          # the ONE host read of the step
        else:
            table = [st.tolist()]
        slots_r, ncand_r, nsel_r = [t[0] for t in table], [t[1] for t in table], [t[2] for t in table]
        n_cand_all, n_sel_all = sum(ncand_r), sum(nsel_r)
        cut_is_inf = (table[0][3] & 0xFFFFFFFF) == 0xFF800000
        if any(sl > capacity for sl in slots_r):
            # the bar was too low for the list.  What was kept is a subset of the survivors, so the k2-th best score among
            # it (the cut just computed, job-wide) is a lower bound of the final cut: scan again just below it, with more
            # room (a level of tied scores -- common-neighbour counts -- may hold far more pairs than K).
            if not cut_is_inf:
                bar = torch.nextafter(cut, neg_inf)
            capacity = _capacity(4 * (capacity - _CHUNK_SLACK))
            continue
        if bar is not None and n_sel_all < min(k2, n_cand_all):
            # fewer than k above the bar: lower it (a quarter of the sample rank each time, then no bar at all)
            bar = None if launches > 3 else estimate_bar(g, fixw, k, safety=SAFETY * 8 ** (launches - 1), rank=rank, world=world)
            launches += 1
            if bar is None:
                capacity = _capacity(2 * (2 * total_half // world))
            continue
        break
    keys, vals = _original_keys(sel_k[:nsel_r[rank]], perm), sel_v[:nsel_r[rank]]
    if world > 1:
        keys, vals = epd.gather_ragged(keys, nsel_r), epd.gather_ragged(vals, nsel_r)
    bits = max(1, min(32, int(g.n_rows - 1).bit_length()))
    keys, vals = ops.select_rows(keys.contiguous(), vals.contiguous(), k, bits)
    if stats is not None:
        # survivors: DIRECTED rows at or above the job-wide cut (what the selection orders); survivor_slots: list slots the
        # launches handed out (chunks: holes included); bar: None or a 1-element device tensor (float(bar) reads it)
        stats.update(candidates=2 * n_cand_all, launches=launches, survivors=2 * n_sel_all,
                     survivor_slots=sum(min(s_, capacity) for s_ in slots_r), bar=bar)
    return torch.stack([keys & 0xFFFFFFFF, keys >> 32]), vals
