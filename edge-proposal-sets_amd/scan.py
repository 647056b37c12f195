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
PLAN_TABLE = True            # eps_scan_screen reads a per-graph table of every column's pieces instead of planning in the launch
PACKED_PIECES = True         # eps_scan_screen may keep key and sum of a candidate in one table word (see _sum_bounds)
RELABEL_MIN_NODES = 100_000  # graphs at least this large are scanned under hubs-first labels (see scan_graph)
ONE_PASS = True            # score with the one-pass piece kernel (csrc/scan_pieces.hip) where the graph qualifies, else eps_filter_scan
_PIECE_SLACK = 8192 * 1024 + 65536   # eps_scan_screen hands slots out in chunks of <= 32768 per workgroup (<= 1024 workgroups)
MAX_SCREEN_SHIFT = 30
HEADS = True               # under a bar a column does not walk its heaviest hub rows (csrc/scan_heads.hip; see head_tables)
HEAD_BETA = 0.5            # ... as long as their screening weights sum to at most this share of the bar
HEAD_KEEP = (0.3, 0.58)    # a head table is used again while its budget stays within this range of the current bar
HEAD_LIST = 2              # the walked list (slots that pass at bar - T_v) is sized this many times the survivor list (16 M slots of 49 M
                           # on the ppa-like graph; an overflow doubles it and repeats.  6 x was 1.8 GB of first-time hipMalloc in a one-shot run)
HEAD_MIN_PATHS = 1 << 30   # graphs with fewer two-hop half paths are scanned without heads: their tables cost more than they save
HEAD_MAX_ROWS = 1024       # rows a column's head holds at most (eps_scan_refine probes each of them for every slot that passes; 48 cost resource allocation 2 ms of scan for 0.15 of refine)
HEAD_CACHE = 4             # head tables kept per (graph, weight table)
DMAX_MARGIN = 5               # a piece drops low weight bits only down to this many bits below the graph's smallest weight (screen_weights)
FORCE_SHARDED = False         # tests on a 1-GPU box: a job of ONE rank takes the sharded step's control flow (its exchanges over the process
                              # group -- with dist.FORCE_COLLECTIVES through RCCL itself; tests/test_gpu_rccl_world1.py)


def _sharded(world: int) -> bool:
    return world > 1 or FORCE_SHARDED


SKETCH_PIECES = True          # r06: in the main launch (heads, under a bar) the packed pieces of single-round columns keep no keys -- a
                              # count-min sketch of two half tables, non-returning adds, a second look at the paths (csrc/scan_pieces.hip)
SKETCH_MIN_PATHS = 1.5        # ... when the bar is at least this many of the HEAVIEST node weight: a tail candidate then needs several paths to
                              # reach it and most sketch pieces end at their read sweep; resource allocation (weights up to 1 under a bar of 0.085
                              # on the ppa-like graph: one path through a light node passes) looks at nearly every piece again -- 10.3 vs 9.9 ms hashed
SKETCH_WIDE = True            # ... and their head tables carry a plan whose packed pieces hold 8192 paths instead of 4096 (a sketch piece has
                              # no keys to run out of slots for): 1.20 M -> 0.98 M pieces on the ppa-like graph, 7.25 -> 6.75 ms
SKETCH_SET = 0                # slots of a sketch piece's set of reported ids (a power of two <= 64; 0 = the kernel's 128): tests shrink it
LAZY_PLAN = True              # the whole-graph plan table (no skipped heads) is built when a launch first wants it; the bar sample plans itself
ROW_RECORDS = True            # the launch gathers a row's cuts, first entry and weight out of ONE 128-byte line (ops.scan_row_records)
BATCH_MIN_COLUMNS = 1 << 16   # lists shorter than this are handed out one column at a time throughout
BATCH_PATHS = 1 << 13      # columns of a heaviest-first list with fewer half paths are handed out eight per ticket (see batch_from)


def scan_available(g: CSRGraph) -> bool:
    """The threshold scan can take this graph: on the GPU, square, SYMMETRIC (the half scheme scores a pair once, in the
    column of its larger endpoint -- on an asymmetric pattern that would silently be a different sum).  Unit values: either
    kernel.  Stored values (collab, rank.py:32-35): positive, symmetric like the pattern, and a graph the piece kernel suits
    (``screen_variant``) -- eps_filter_scan has no weighted flavour."""
    if not (scan_plausible(g) and is_symmetric(g)):           # (any id space: wider ones are scanned in id windows)
        return False
    return g.val is None or (values_symmetric(g) and screen_variant(scan_graph(g)[0]) is not None)


def scan_plausible(g: CSRGraph) -> bool:
    """The cheap part of ``scan_available`` (no table, no kernel): on the GPU, square, within the node and entry limits.  What a
    caller checks BEFORE it pays for the hubs-first copy of a graph the scan would refuse anyway."""
    return bool(g.device.type == "cuda" and g.n_rows == g.n_cols and 0 < g.n_rows <= ops.filter_scan_max_nodes() and g.nnz() < 1 << 30)


def scan_usable(g: CSRGraph, node_w: torch.Tensor) -> bool:
    """scan_topk will take this (graph, weight table): ``scan_available``, the scores fit the fixed-point range, and -- for a
    graph with stored values -- the one-pass kernel's tables are usable (non-negative weights, 32-bit screening sums)."""
    from . import candidates
    if not (scan_available(g) and candidates.fused_scores_fit(g, node_w)):
        return False
    if g.val is None:
        return True
    gs, perm = scan_graph(g)
    return screen_variant(gs) is not None and screen_weights(g, gs, perm, node_w).usable


def values_symmetric(g: CSRGraph) -> bool:
    """Stored values positive and equal on mirrored entries (cached): entry e = (v, w) has its mirror (w, v) at
    rowptr[w] + revpos[e]."""
    if "values_symmetric" not in g._cache:
        ok = True
        if g.val is not None and g.nnz():
            mirror = g.rowptr[g.col.long()] + reverse_positions(g).long()
            ok = bool(((g.val[mirror] == g.val) & (g.val > 0)).all().item())
        g._cache["values_symmetric"] = ok
    return g._cache["values_symmetric"]


def reverse_positions(g: CSRGraph) -> torch.Tensor:
    """revpos of the graph (cached); the same pass leaves the half-path counts and the symmetry verdict in the cache.
    Built by the symmetric-pattern kernel (one search per unordered stored pair: 4.9 -> ~1.7 ms on the ppa-like graph); a
    pattern that turns out NOT to be symmetric gets the general table (eps_reverse_positions) -- the scan refuses it anyway."""
    if "revpos" not in g._cache:
        rev, hp, info = ops.reverse_positions_symmetric(g.rowptr, g.col)
        # one host read for the verdict AND the graph's scalars the scan asks for later (each used to be a read of its own)
        flag_h, max_deg, max_half, total = info.tolist()
        sym = (flag_h & 0xFFFFFFFF) == 0
        if not sym:
            rev, hp, _ = ops.reverse_positions(g.rowptr, g.col, with_stats=True)
        else:
            g._cache.setdefault("max_degree", int(max_deg))
            g._cache.setdefault("max_half", int(max_half))
            g._cache.setdefault("total_half", int(total))
        g._cache["revpos"], g._cache["half_paths"], g._cache["symmetric"] = rev, hp, sym
    return g._cache["revpos"]


def is_symmetric(g: CSRGraph) -> bool:
    """Whether the stored pattern is symmetric (cached): every entry (v, w) finds v at its position in row w (checked by the
    reverse-positions pass itself).  A relabelled copy answers for its original (symmetry does not depend on the labels), so a
    graph that is scanned under hubs-first labels pays for ONE reverse-positions table, the copy's."""
    if "symmetric" not in g._cache:
        if "deg_order" in g._cache and g._cache["deg_order"][0] is not g:
            g._cache["symmetric"] = is_symmetric(g._cache["deg_order"][0])
        else:
            reverse_positions(g)
    return g._cache["symmetric"]


def half_paths(g: CSRGraph) -> torch.Tensor:
    """int64[N]: two-hop paths v - w - u with u < v per column v = the work of scanning it (cached; from the revpos pass)."""
    if "half_paths" not in g._cache:
        reverse_positions(g)
    return g._cache["half_paths"]


def column_order(g: CSRGraph) -> torch.Tensor:
    """int32[N]: all columns, heaviest first (a hub column is one workgroup's work for a long time: it must not start last)."""
    if "scan_order" not in g._cache:
        hp = half_paths(g)
        g._cache["scan_order"] = (ops.node_order(keys=hp) if hp.is_cuda and hp.numel() else
                                  torch.argsort(hp, descending=True, stable=True).to(torch.int32))
    return g._cache["scan_order"]


def window_splits(g: CSRGraph):
    """Row split table for the id windows eps_filter_scan uses on this graph (None: one window); cached."""
    if "scan_splits" not in g._cache:
        win_ids, n_win = ops.filter_scan_windows(g.n_rows)
        g._cache["scan_splits"] = ops.row_window_splits(g.rowptr, g.col, win_ids, n_win)
    return g._cache["scan_splits"]


def max_half_paths(g: CSRGraph) -> int:
    """Two-hop half paths of the graph's heaviest column (cached host number; the reverse-positions pass leaves it)."""
    if "max_half" not in g._cache:
        hp = half_paths(g)
        if "max_half" not in g._cache:
            g._cache["max_half"] = int(hp.max().item()) if g.n_rows else 0
    return g._cache["max_half"]


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
    Relabelling sorts the stored entries once (9 ms for 42.5 M, tools/r03_cold_scan.py) and the one-pass kernel
    (csrc/scan_pieces.hip) needs the even columns it gives, so even ONE scan repays it (first scan of a fresh ppa-like
    graph 57 ms as labelled, 46 ms relabelled with every table built; every later scan 52 vs 23 ms -- in a process whose
    allocator is cold the relabelled path's extra first-time hipMalloc costs ~30 ms more: profiles/r03/cold_scan.txt):
    filter.py and bench.py pass
    ``build=True`` / ``scan_topk(relabel=True)``; without it the relabelled copy is used when it already exists (the
    GNN path builds the same copy for its SpMM) and the graph is scanned as labelled otherwise."""
    if g.val is not None:                 # stored values: only the piece kernel scans them, and it wants even columns
        gs, perm, _ = g.degree_ordered()
        return gs, perm
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


def screen_tables(g: CSRGraph):
    """(bounds int32[M + 1], cuts int16-bits [N, M]) of the one-pass scan, cached on the graph: the id space cut into M windows
    of equal stored-entry mass (so a column's paths spread evenly over them), and per row the number of entries below each
    boundary."""
    if "screen_tables" not in g._cache:
        if max_degree(g) >= 1 << 16:
            raise ops._lib.EpsError("screen_tables: the cut table counts a row's entries in 16 bits (max degree < 65536)")
        # (small device kernels rather than tensor ops: a one-shot filter.py run pays ~10 ms of code-object loading for every
        #  torch operator it is the first to use)
        bounds = ops.scan_bounds(g.rowptr, g.n_rows)
        g._cache["screen_tables"] = (bounds, ops.scan_cuts(g.rowptr, g.col, bounds))
    return g._cache["screen_tables"]


def window_paths(g: CSRGraph) -> torch.Tensor:
    """int32-bits [N, M]: every column's two-hop half paths per id window (cached per graph): what the scan's planner merges
    into pieces -- 128 bytes per column instead of a cut row per (column, neighbour) in every launch."""
    if "window_paths" not in g._cache:
        g._cache["window_paths"] = ops.scan_window_paths(g.rowptr, g.col, reverse_positions(g), screen_tables(g)[1])
    return g._cache["window_paths"]


_VAL_INFLATION = 1.0 + 2.0 ** -19     # stored values: a term is ceil(A[u,w] x rowf) + 1 with rowf scaled by 2^shift x (1 + 2^-20) and two
                                      # float32 roundings on the way -- the sum may exceed bound x 2^shift by that relative amount


def screen_fits(bound: float, shift: int, max_deg: int, weighted: bool = False) -> bool:
    """Every screening sum of the graph stays below 2^31 - 2 at this shift (bit 31 of a table word marks a known edge): score
    bound -- inflated for stored values, whose per-path terms are rounded products -- plus one rounding unit per term."""
    return bound * (_VAL_INFLATION if weighted else 1.0) * (1 << shift) + max_deg < (1 << 31) - 2


def screen_shift(bound: float, max_deg: int, weighted: bool = False) -> int:
    """Fixed point 2^-shift of the screening sums: the finest one that keeps every sum of the graph below 2^31 (score bound
    plus one rounding unit per term; bit 31 is the kernel's known-edge flag), at most MAX_SCREEN_SHIFT."""
    shift = MAX_SCREEN_SHIFT
    while shift > 0 and not screen_fits(bound, shift, max_deg, weighted):
        shift -= 1
    return shift


SKETCH_PIECE_PATHS = 8192      # paths a sketch piece holds at most (csrc/scan_pieces.hip: wide_paths <= SP_UBITS)


def sketch_safe_shift(f_hi: int) -> int:
    """The finest screening fixed point 2^-shift under which SKETCH_PIECE_PATHS paths of the heaviest node weight (``f_hi``, in the
    2^-40 units of the exact scores; screening weights round up) sum to less than 2^32."""
    shift = MAX_SCREEN_SHIFT
    while shift > 0 and ((max(0, f_hi) >> (40 - shift)) + 2) * SKETCH_PIECE_PATHS >= 1 << 32:
        shift -= 1
    return shift


class Screen:
    """What a launch of the piece kernel and the exact re-scoring of its survivors need, for one (graph, weight table)."""
    __slots__ = ("fx32", "shift", "fixw", "val", "node_w", "usable", "ssum", "smax", "_plan", "_plan_build", "d_used", "w_min", "heads", "head_cur",
                 "rowrec", "vword", "exact", "bar_hint", "w_max_units")

    def __init__(self, fx32, shift, fixw, val, node_w, usable, ssum=None, smax=None, plan=None, d_used=0, w_min=0.0):
        self.fx32, self.shift, self.fixw, self.val, self.node_w, self.usable = fx32, shift, fixw, val, node_w, usable
        self.ssum, self.smax, self._plan = ssum, smax, plan
        self._plan_build = None      # r06: () -> ((pptr, records), bits dropped): the whole-graph plan is built when a launch WITHOUT
                                     # skipped heads first asks for it (see `plan`); a one-shot run under a bar never does
        self.d_used = d_used         # most low bits any packed / 16-bit direct piece drops from the screening weights
        self.w_min = w_min           # smallest weight of a node that can be a common neighbour (0: no relative bound)
        self.heads = {}              # budget (table units) -> HeadTables (see head_tables)
        self.head_cur = None         # the HeadTables the last launch under a bar used
        self.rowrec = None           # ops.scan_row_records: one 128-byte line per node with what the walk gathers per row
        self.vword = None            # the `variant` word the plan was built with (geometry + the limit on dropped weight bits)
        self.w_max_units = float("inf")   # the heaviest node weight in screening units (unit-valued graphs: set by screen_weights)
        self.bar_hint = {}           # (k, stride, safety) -> the bar (host float) the last scan with these settings ended with
        self.exact = False           # screening sums ARE the exact scores (one weight for every node, a multiple of every unit a piece
                                     # may round to -- common neighbours): the survivors need no re-scoring

    @property
    def has_plan(self) -> bool:
        """Launches of this (graph, weight table) read their pieces from plan tables (built or still to be built)."""
        return self._plan is not None or self._plan_build is not None

    @property
    def plan(self):
        """(pptr, records) of the whole graph WITHOUT skipped heads, built on first use (0.7 ms of window paths + 1.0 ms of planning on
        the ppa-like graph, one host read): the launches of a step under a bar bring the plan of their head table, and the bar sample
        plans its ~1000 columns inside the launch (`_launch(sample_wp=...)`)."""
        if self._plan is None and self._plan_build is not None:
            self._plan, self.d_used = self._plan_build()
            self._plan_build = None
        return self._plan

    def lower_bound(self, s: torch.Tensor, max_deg: int) -> torch.Tensor:
        """A lower bound of the exact score of a pair whose screening score is ``s`` (monotone in s).  A path's screening term
        exceeds its exact 2^-40 term by less than (2^d + 1) x 2^-shift: once for the round-up to 2^-shift, once for the d low
        bits a packed / 16-bit direct piece drops (d <= d_used).  A pair has at most max_deg paths -- and, when every node
        weighs at least w_min > 0, at most exact / w_min <= s / w_min of them, which is what keeps the bound tight under the
        coarse weights (3.7 % of s on the ppa-like graph against 13230 x 2^-8 = 52 absolute).  Weighted graphs form a term from
        two roundings (no dropped bits)."""
        a, b = self.lower_params(max_deg)
        return torch.maximum(s - a, s * b)

    def lower_params(self, max_deg: int, d_used: Optional[int] = None):
        """(a, b) with lower_bound(s) = max(s - a, s * b) -- the form eps_select_compact evaluates on the device (mode 2).
        b = 0 without a relative bound: scores are sums of non-negative weights, so 0 is a floor under all of them.
        ``d_used``: the dropped bits of the plan the launch ran with (a head table has its own plan)."""
        d_used = self.d_used if d_used is None else max(int(d_used), 0)
        unit = (2.0 ** d_used + 1.0) * 2.0 ** -self.shift * (2.0 if self.val is not None else 1.0)
        b = 0.0
        if self.w_min > 0.0 and unit < self.w_min:
            b = 1.0 - 1.00001 * unit / self.w_min                                     # (w_min itself is exact to 2^-40)
        return max_deg * unit, b


def _sum_bounds(g: CSRGraph, fx32: torch.Tensor):
    """(ssum int32-bits [N], smax int32-bits [M + 1], min_fx int32-bits [1]) for eps_scan_screen's packed / 16-bit direct
    pieces: ssum[v] = sum of the screening weights over row v -- no pair with endpoint v sums to more --, smax[k] = the largest
    ssum among ids >= bounds[k], min_fx = the smallest screening weight of a node that can be a common neighbour."""
    bounds, _ = screen_tables(g)
    return ops.scan_row_sums(g.rowptr, g.col, fx32, bounds, g.n_rows)


def screen_weights(g0: CSRGraph, g: CSRGraph, perm, node_w: torch.Tensor) -> Screen:
    """Tables of the one-pass scan of ``g`` (the scanned copy of ``g0``), cached per weight tensor and labelling: unit-valued
    graphs get screening weights rounded up to 2^-shift (fx32) and re-score with the 2^-40 fixed-point weights (fixw); graphs
    with stored values form the screening weight per path from (val, float node weights) and re-score with those.
    ``usable`` is False for negative weights or a score bound beyond 32 bits: such tables stay on eps_filter_scan."""
    def build():
        from . import candidates
        bound = candidates.fused_score_bound(g0, node_w)
        shift = screen_shift(bound, 2 * max_degree(g), g.val is not None)
        fits = screen_fits(bound, shift, 2 * max_degree(g), g.val is not None)
        if g.val is not None:
            nw = (node_w if perm is None else node_w[perm]).contiguous()
            return Screen(None, shift, None, g.val, nw, fits and bool((nw >= 0).all().item()))
        fixw = _scan_weights(g0, g, perm, node_w)
        if fixw.numel():
            # (r06) the heaviest screening weight stays below 2^19: 8192 of them -- a whole sketch piece hashed into ONE slot -- cannot
            # wrap a 32-bit slot, so a sketch table's sums are upper bounds whatever collides (one more host read when a weight table's
            # screen is built; Adamic-Adar on the ppa-like graph: 1 / ln 2 x 2^19 -> shift 18)
            # (... when that costs one bit at most: resource allocation -- weights from 1 down to 1 / 13 230 -- would lose three and
            #  every small weight its resolution; its bar lies below its heaviest weight anyway: no sketch pieces there, SKETCH_MIN_PATHS)
            safe = sketch_safe_shift(int(fixw.max().item()))
            if shift - safe <= 1:
                shift = min(shift, safe)
        fx32, bad = ops.scan_screen_weights(fixw, shift)
        # Everything is launched before anything is read back: the verdict on the weights (`bad`), the bits the plan drops and the
        # smallest weight come back in ONE host read at the end (tables built from unusable weights are simply not used).
        one_pass = fits and one_pass_available(g)
        ssum, smax, min_fx = _sum_bounds(g, fx32) if one_pass and PACKED_PIECES else (None, None, None)
        plan, d_word, plan_build = None, None, None
        vword = screen_variant(g)
        dmax_limit = max(0, min(24, shift - 8))                   # (csrc/scan_pieces.hip: packed_dmax)
        if one_pass and PLAN_TABLE:
            # How many low weight bits a piece may drop: a screening score exceeds the exact one by up to 2^d + 1 units per path, and
            # the pre-filter in front of the exact re-scoring (lower_params) is as sharp as that is small next to the smallest
            # weight -- DMAX_MARGIN bits below it (one host read; resource allocation on a graph with hubs weighs 1 / 13 230: at the
            # default d = 13 a tenth of every score was slack and the re-scoring took 13.5 ms, r05)
            lowest = int(min_fx.item()) & 0xFFFFFFFF if min_fx is not None else 0xFFFFFFFF
            if lowest not in (0, 0xFFFFFFFF):
                dmax_limit = min(dmax_limit, max(0, lowest.bit_length() - 1 - DMAX_MARGIN))
                vword = ops.scan_variant_word(screen_variant(g), max(0, lowest.bit_length() - 1 - DMAX_MARGIN))
            # every column's pieces, planned once per (graph, weight table): a launch reads them instead of planning (5 %) -- r06:
            # planned when a launch without skipped heads first wants them (Screen.plan); LAZY_PLAN off: here and now, as in r05
            vword_plan = vword

            def plan_build():
                bounds, cuts = screen_tables(g)
                pptr, recs, dw = ops.scan_plan(g.rowptr, cuts, window_paths(g), ssum, smax, bounds, g.n_rows, shift, vword_plan, with_d=True)
                return (pptr, recs), int(dw.item()) & 0xFFFFFFFF
            if not LAZY_PLAN:
                bounds, cuts = screen_tables(g)
                pptr, recs, d_word = ops.scan_plan(g.rowptr, cuts, window_paths(g), ssum, smax, bounds, g.n_rows, shift, vword, with_d=True)
                plan, plan_build = (pptr, recs), None
        zero = torch.zeros(1, dtype=torch.int32, device=g.device)
        f_lo, f_hi = (fixw.min(), fixw.max()) if fixw.numel() else (torch.zeros((), dtype=torch.int64, device=g.device),) * 2
        bad_h, d_h, min_h, f_lo, f_hi = torch.cat([bad.view(torch.int32).to(torch.int64), (d_word if d_word is not None else zero).to(torch.int64),
                                                   (min_fx if min_fx is not None else zero - 1).to(torch.int64),
                                                   f_lo.view(1), f_hi.view(1)]).tolist()
        bad_h, d_h, min_h = bad_h & 0xFFFFFFFF, d_h & 0xFFFFFFFF, min_h & 0xFFFFFFFF
        usable = fits and bad_h == 0
        rowrec = ops.scan_row_records(screen_tables(g)[1], g.rowptr, fx32) if usable and one_pass and ROW_RECORDS else None
        if not usable:
            ssum = smax = plan = plan_build = None
        d_used, w_min = 0, 0.0
        if ssum is not None:
            # the launch's limit (csrc/scan_pieces.hip: packed_dmax; what a launch that plans itself may drop), or what the plan really uses
            d_used = d_h if plan is not None else dmax_limit
            # fx32 rounds the exact weight x 2^shift UP: one unit less is a floor under every common neighbour's exact weight
            lowest = min_h & 0xFFFFFFFF
            w_min = 0.0 if lowest == 0xFFFFFFFF else max(0, lowest - 1) * 2.0 ** -shift
        sc = Screen(fx32, shift, fixw, None, None, usable, ssum, smax, plan, d_used, w_min)
        sc._plan_build = plan_build if one_pass and PLAN_TABLE and usable else None
        sc.rowrec = rowrec
        sc.vword = vword
        sc.w_max_units = float(f_hi) * 2.0 ** (shift - 40)      # the heaviest node weight in screening units (see SKETCH_MIN_PATHS)
        # One weight for all nodes (common neighbours: 1.0), a whole number of screening units that stays whole under every low bit a
        # piece may drop (csrc/scan_pieces.hip: packed_dmax <= min(24, shift - 8)): a path's screening term is then its exact 2^-40
        # term, a pair's screening sum c x fx x 2^-shift converts to the same float32 as eps_rescore_runs' c x fixw x 2^-40
        # (c < 2^24 paths), and scan_topk skips the re-scoring (2.3 of 15.4 ms on the ppa-like graph).
        down, dm = 40 - shift, max(0, min(24, shift - 8))
        sc.exact = bool(ssum is not None and 0 < f_lo == f_hi and down >= 0 and f_lo % (1 << down) == 0
                        and (f_lo >> down) % (1 << dm) == 0 and (f_lo >> down) < (1 << 31) and EXACT_SCREENING)
        return sc
    # (one entry per labelling: the tables are indexed by the SCANNED graph's node ids)
    return g0.weight_cached("screen_weights" if perm is None else "screen_weights_relabelled", node_w, build)


class HeadTables:
    """What a launch with skipped heads brings (one set per budget): the head table, the window paths and the plan of the rows
    that are still walked, and that plan's dropped weight bits."""
    __slots__ = ("budget", "heads", "wpaths", "plan", "d_used", "live", "n_hub", "pack", "wide")

    def __init__(self, budget, heads, wpaths, plan, d_used, n_hub=0, wide=False):
        self.budget, self.heads, self.wpaths, self.plan, self.d_used = budget, heads, wpaths, plan, d_used
        self.wide = wide             # the plan's packed pieces hold up to 8192 paths: for SKETCH launches only (ops.SCAN_WIDE)
        self.n_hub = n_hub           # hub rows the heads were cut from (a graph's second scan widens the table: rebuilt then)
        self.pack = None             # ops.scan_column_pack of this plan (see column_pack)
        self.live = {}               # (rank, world) -> this rank's columns without the DEAD ones (see live_columns)


HUB_TABLE_BYTES = 2 << 30   # the hub row bitmaps of a graph take at most this much (ppa-like: 16384 rows x 72 KB = 1.2 GB of 288)


HUB_FIRST = 4096            # hub rows of a graph's FIRST scan: a one-shot filter.py run pays ~25 ms of first-time hipMalloc per GB,
                            # more than the 1.3 ms the wider table saves its one launch; the second scan of a graph builds the full table


def hub_rows(g: CSRGraph) -> torch.Tensor:
    """int32-bits [n_hub, words], cached per graph: the adjacency rows of the first n_hub ids (the hubs, under hubs-first
    labels) as bitmaps over the id space -- where eps_scan_refine looks a pair's skipped rows up.  HUB_FIRST rows while the
    graph has been scanned once, ops.HUB_MAX from its second scan on (``_count_scan`` drops the narrow table)."""
    if "hub_rows" not in g._cache:
        most = ops.HUB_MAX if g._cache.get("scan_calls", 0) > 1 else min(ops.HUB_MAX, HUB_FIRST)
        n_hub = min(most, g.n_rows, HUB_TABLE_BYTES // (4 * ops.scan_hub_row_words(g.n_rows)))
        g._cache["hub_rows"] = ops.scan_hub_rows(g.rowptr, g.col, n_hub)
    return g._cache["hub_rows"]


def _count_scan(g: CSRGraph, screen) -> None:
    """One more scan_topk call on this (scanned) graph; at the second, the hub table of the first -- and the head tables built
    against it -- make room for the full-width ones."""
    g._cache["scan_calls"] = g._cache.get("scan_calls", 0) + 1
    if g._cache["scan_calls"] == 2 and "hub_rows" in g._cache and g._cache["hub_rows"].shape[0] < min(ops.HUB_MAX, g.n_rows):
        del g._cache["hub_rows"]
        if screen is not None:
            screen.heads.clear()
            screen.head_cur = None


def head_budget(bar_units: float) -> int:
    """Budget of a column's skipped head for a bar (both in table units): HEAD_BETA x bar, rounded down to four significant
    bits -- bars that differ by a few per cent share a table."""
    b = int(HEAD_BETA * bar_units)
    if b <= 0:
        return 0
    drop = max(0, b.bit_length() - 4)
    return (b >> drop) << drop


def head_tables(g: CSRGraph, screen: Screen, budget: int, wide: bool = False) -> HeadTables:
    """The tables of a launch whose columns skip heads of at most ``budget`` (table units), cached on the Screen: the head table
    (eps_scan_heads), the window paths and the plan table of the walked rows.  ~1.5 ms on the ppa-like graph, once per bar
    level; one host read (the plan's size and dropped bits)."""
    n_hub = hub_rows(g).shape[0]
    # (``wide``: a plan for sketch launches -- its packed pieces hold twice the paths, csrc/scan_pieces.hip -- is a table set of its own)
    budget_key, budget = (budget, bool(wide)), int(budget)
    if budget_key in screen.heads and screen.heads[budget_key].n_hub != n_hub:
        # built against the narrow hub table of the graph's first scan (HUB_FIRST): still correct -- the wider bitmaps are a
        # superset -- but its heads stop at the old width; every weight table of the graph gets the full-width heads
        del screen.heads[budget_key]
    if budget_key not in screen.heads:
        heads = ops.scan_heads(g.rowptr, g.col, screen.fx32, n_hub, budget, HEAD_MAX_ROWS)
        bounds, cuts = screen_tables(g)
        wp = ops.scan_window_paths(g.rowptr, g.col, reverse_positions(g), cuts, heads)
        pptr, recs, d_word = ops.scan_plan(g.rowptr, cuts, wp, screen.ssum, screen.smax, bounds, g.n_rows, screen.shift,
                                           (screen.vword if screen.vword is not None else screen_variant(g)) | (ops.SCAN_WIDE if wide else 0),
                                           with_d=True, heads=heads)
        while len(screen.heads) >= 2 * HEAD_CACHE:
            screen.heads.pop(next(iter(screen.heads)))
        screen.heads[budget_key] = HeadTables(budget, heads, wp, (pptr, recs), int(d_word.item()), n_hub, bool(wide))
    return screen.heads[budget_key]


def live_columns(g: CSRGraph, screen: Screen, ht: HeadTables, rank: int, world: int) -> torch.Tensor:
    """This rank's columns of the heaviest-first order without the DEAD ones: a column whose row sum of screening weights lies
    below the lowest bar the head table serves (budget / HEAD_KEEP[1]) cannot hold a pair that passes -- a quarter of the
    ppa-like graph's columns at K = 4 M, which would otherwise each draw a ticket and load a header to find that out.  (The
    launch checks every column against the bar at hand as well; scan_topk voids a launch whose bar fell below the table's.)"""
    key = (rank, world)
    if key not in ht.live:
        mine = shard_columns(g, rank, world)
        floor = int(ht.budget / HEAD_KEEP[1])
        ssum = screen.ssum.to(torch.int64).bitwise_and(0xFFFFFFFF)
        ht.live[key] = mine[ssum[mine.long()] >= floor].contiguous()
    return ht.live[key]


COLUMN_RECORDS = True         # a launch reads a column's header as one 32-byte record in hand-out order (column_records)


def column_records(g: CSRGraph, screen: Screen, columns: torch.Tensor, plan, heads: Optional[torch.Tensor], cache: dict, key):
    """int32 [len(columns), 8], cached in ``cache[key]``: per column of the hand-out list {v, rowptr[v], degree, head rows, head
    weight, row sum, first plan record, pieces} -- what the kernel's column set-up otherwise reads through a chain of dependent
    loads into five tables (plain gathers here: the list, the plan and the head table are fixed once built)."""
    if not COLUMN_RECORDS or plan is None or columns.numel() == 0:
        return None
    if key not in cache:
        c = columns.long()
        rp = g.rowptr
        pptr = plan[0].to(torch.int64).bitwise_and(0xFFFFFFFF)
        zero = torch.zeros_like(c)
        hx, hy = (heads[c, 0].to(torch.int64), heads[c, 1].to(torch.int64)) if heads is not None else (zero, zero)
        ss = screen.ssum[c].to(torch.int64) if screen.ssum is not None else zero
        rec = torch.stack([c, rp[c], rp[c + 1] - rp[c], hx, hy, ss, pptr[c], pptr[c + 1] - pptr[c]], 1)
        cache[key] = rec.to(torch.int32).contiguous()           # (two's-complement wrap keeps the low 32 bits of every field)
    return cache[key]


COLUMN_PACK = True            # the main launch sets a column up from ONE stream of 32-byte records (ops.scan_column_pack) -- from a
                              # graph's second scan on: 32 B per stored entry (1.4 GB on the ppa-like graph, 1.5 ms to build per head
                              # table) is not what a one-shot filter.py run should pay


def column_pack(g: CSRGraph, screen: Screen, ht: HeadTables) -> Optional[torch.Tensor]:
    """The per-column pack of the head table's plan (cached on it), or None: not for a graph's first scan, not without row records."""
    if not COLUMN_PACK or screen.rowrec is None or g._cache.get("scan_calls", 0) < 2:
        return None
    if ht.pack is None:
        ht.pack = ops.scan_column_pack(g.rowptr, g.col, reverse_positions(g), screen.rowrec, ht.plan)
    return ht.pack


def _heads_for(g: CSRGraph, screen: Screen, bar, wide_for=None) -> Optional[HeadTables]:
    """The head tables for a launch under ``bar`` (1-element device tensor): the set the last launch used, without looking at the
    bar -- the kernel itself refuses a head as heavy as the bar (status bit 2), and scan_topk compares budget and bar after
    the step's one host read --, or, the first time, a set built for the bar (one host read of it)."""
    if screen.head_cur is not None and screen.head_cur.n_hub != hub_rows(g).shape[0]:
        screen.head_cur = None                       # (cut from a narrower hub table than the graph has now)
    if screen.head_cur is None:
        b = float(bar)
        if not (b > 0.0 and b < float("inf")):
            return None
        budget = head_budget(b * 2.0 ** screen.shift)
        if budget <= 0:
            return None
        # (``wide_for(budget)``: whether the launch will run sketch pieces -- then the tables carry the plan made for them; asked
        #  BEFORE anything is built: a first scan does not build a set it will not use)
        screen.head_cur = head_tables(g, screen, budget, bool(wide_for(budget)) if wide_for is not None else False)
    elif wide_for is not None and screen.head_cur.wide != bool(wide_for(screen.head_cur.budget)):
        screen.head_cur = head_tables(g, screen, screen.head_cur.budget, bool(wide_for(screen.head_cur.budget)))
    return screen.head_cur


REWALK_MAX = 0.25            # largest share of re-walked paths (hash-partitioned passes) the one-pass kernel is chosen with


def _rewalk_fraction(g: CSRGraph, variant: int) -> float:
    """Extra path visits of eps_scan_screen on this graph, as a share of its half paths: a single id window that is wider than a
    direct piece AND holds more paths than a hash piece is walked in `parts` hash-partitioned passes.  Read off the plan table
    (planned without sum bounds: two-word hash slots only -- packed pieces can only lower it)."""
    bounds, cuts = screen_tables(g)
    plan = ops.scan_plan(g.rowptr, cuts, window_paths(g), None, None, bounds, g.n_rows, 0, variant)
    again, total = ops.scan_plan_rewalk(plan, variant)
    return again / max(1, total)


def screen_variant(g: CSRGraph):
    """Geometry of eps_scan_screen for this graph, or None when the piece kernel does not suit it (-> eps_filter_scan).
    What decides is how much of the graph the kernel would have to RE-WALK: dense stretches of a column go through direct pieces
    whatever they weigh, sparse ones through hash pieces of bounded capacity, and a single id window that is both wide and heavy
    is walked in hash-partitioned passes.  Under hubs-first labels that share is ~0 for the ppa-like graph (heaviest column
    63 k half paths) and small even for raw R-MAT graphs whose heaviest columns hold 0.3-0.85 M (13 / 48 / 55 ms against
    65 / 175 / 170 ms for the two-pass kernel: profiles/r03/other_graphs.txt); a graph scanned AS LABELLED has columns of millions
    of paths in sparse windows (147-686 ms against 48.7) and stays on the two-pass kernel.  The smallest table that keeps the
    share below REWALK_MAX wins (more workgroups per CU: 17.6 ms vs 19.7 / 32 ms on the ppa-like graph).  Rows must be shorter
    than 2^16 (the cut table is uint16)."""
    if "screen_variant" not in g._cache:
        v = None
        if ONE_PASS and 0 < max_degree(g) < 1 << 16 and g.n_rows:
            # (no column heavier than M two-word hash pieces: nothing can need a partitioned pass -- the usual case, decided
            #  without planning the graph three times)
            if max_half_paths(g) <= 2048 * ops.scan_windows():
                v = 2
            else:
                for variant in (2, 0, 1):
                    if _rewalk_fraction(g, variant) <= REWALK_MAX:
                        v = variant
                        break
        g._cache["screen_variant"] = v
    return g._cache["screen_variant"]


def one_pass_available(g: CSRGraph) -> bool:
    return screen_variant(g) is not None


def _launch(g, fixw, columns, threshold, capacity, scores_only: bool = False, both: bool = False, screen=None,
            heads: Optional[HeadTables] = None, walked_capacity: int = 0, sample_key=None, sketch: bool = False) -> ops.Survivors:
    """``screen`` (a Screen) -> the one-pass kernel (screening scores in ``val``), else eps_filter_scan (exact scores).
    ``heads``: the launch skips the columns' heads (its list -- ``walked_capacity`` slots -- holds walked sums) and
    eps_scan_refine completes them into the list that is returned: the same survivors and scores as without heads, compact;
    ``walked_slots`` (device word) = slots the walked list handed out."""
    # (the piece kernel marks the unused slots of its reservations itself: no fill, readers stop at the slot counter)
    out = ops.Survivors(capacity, threshold, g.device, scores_only, both, prefill=screen is None)
    out.walked_slots = None
    if screen is not None:
        out.status = torch.empty(1, dtype=torch.int32, device=g.device)      # (cleared by eps_scan_screen itself)
    if columns.numel():
        if screen is not None and heads is not None:
            bounds, cuts = screen_tables(g)
            walked = ops.Survivors(walked_capacity, threshold, g.device, prefill=False)
            ops.scan_screen(g.rowptr, g.col, reverse_positions(g), screen.fx32, cuts, bounds, g.n_rows, columns, screen.shift, walked,
                            out.status, screen_variant(g) | ((ops.SCAN_SKETCH | SKETCH_SET << 17) if sketch else 0) | (ops.SCAN_WIDE if heads.wide else 0), None, None, heads.wpaths, screen.ssum, screen.smax,
                            heads.plan, heads.heads,
                            batch_from(g, columns), screen.rowrec,
                            column_records(g, screen, columns, heads.plan, heads.heads, heads.live, ("colrec", columns.data_ptr(), columns.numel())),
                            column_pack(g, screen, heads) if variant_is_main(g, screen) else None)
            ops.scan_refine(walked, heads.heads, hub_rows(g), screen.fx32, g.rowptr, g.col, g.n_rows, screen.shift, out)
            out.rec[4:5].copy_(walked.rec[4:5])              # (candidates the walk touched)
            out.walked_slots = walked.rec[1:2]
        elif screen is not None:
            bounds, cuts = screen_tables(g)
            if sample_key is not None and screen.val is None and screen._plan is None and screen._plan_build is not None:
                # the bar sample of a graph whose whole-graph plan has not been needed yet: the launch plans its ~1000 columns itself
                # (the same planner, the same pieces) from the window paths of THOSE columns -- no 0.7 + 1.0 ms of whole-graph tables
                wp, plan = sample_window_paths(g, columns, sample_key), None
            else:
                wp, plan = window_paths(g), screen.plan
            ops.scan_screen(g.rowptr, g.col, reverse_positions(g), screen.fx32, cuts, bounds, g.n_rows, columns, screen.shift, out,
                            out.status, screen.vword if screen.vword is not None and screen.val is None else screen_variant(g),
                            screen.val, screen.node_w, wp, screen.ssum, screen.smax, plan,
                            None, batch_from(g, columns), screen.rowrec)
        else:
            ops.filter_scan(g.rowptr, g.col, reverse_positions(g), fixw, g.n_rows, columns, out, max_degree(g), window_splits(g))
    return out


def variant_is_main(g: CSRGraph, screen: Screen) -> bool:
    """The launch is the one the specialised body serves (csrc/scan_pieces.hip FULL): 256-thread geometry, plan, records, sum bounds."""
    return bool(screen_variant(g) == 2 and screen.has_plan and screen.rowrec is not None and screen.ssum is not None
                and COLUMN_RECORDS)


def sample_window_paths(g: CSRGraph, columns: torch.Tensor, key) -> torch.Tensor:
    """The window-path rows of the bar sample's columns alone (cached per sample; the table is whole-graph sized, uninitialised elsewhere)."""
    ck = ("sample_wpaths",) + tuple(key)
    if ck not in g._cache:
        g._cache[ck] = ops.scan_window_paths(g.rowptr, g.col, reverse_positions(g), screen_tables(g)[1], columns=columns)
    return g._cache[ck]


def batch_from(g: CSRGraph, columns: torch.Tensor) -> int:
    """How many columns of the heaviest-first list ``columns`` are handed to the workgroups one at a time: those with at least
    BATCH_PATHS half paths; the light rest goes eight per draw (the hand-out is an atomic on one device word, ~11 ns each, and a
    light column is ~10 us of one of 1024 workgroups: singly, the tickets would set the pace).  Cached per list (one host read)."""
    if columns.numel() < BATCH_MIN_COLUMNS:          # (a short list -- the bar sample -- has fewer columns than workgroups to feed)
        return columns.numel()
    # (keyed by the list OBJECT, which the entry keeps alive: a raw device pointer is handed out again once its list is freed --
    #  the live-column lists of evicted head tables -- and would then answer for another list of the same length)
    memo = g._cache.setdefault("batch_from", {})
    key = (id(columns), BATCH_PATHS)
    hit = memo.get(key)
    if hit is None or hit[0] is not columns:
        while len(memo) >= 4 * HEAD_CACHE + 8:
            memo.pop(next(iter(memo)))
        hit = memo[key] = (columns, int((half_paths(g)[columns.long()] >= BATCH_PATHS).sum().item()))
    return hit[1]


def candidate_count(g: CSRGraph, screen: Optional[Screen], fixw, rank: int = 0, world: int = 1) -> int:
    """Unordered 2-hop non-edges of the whole graph (cached): one scan that reports nothing (bar +inf, no skipped heads) and
    counts.  A launch with skipped heads counts only the candidates its walk touches; this is the number the filter covers."""
    if "n_candidates" not in g._cache:
        res = _launch(g, fixw, shard_columns(g, rank, world), float("inf"), 1 << 16, screen=screen)
        n = res.rec[4:5].clone()
        if _sharded(world):
            from . import dist as epd
            n = epd.all_reduce_sum_(n)
        g._cache["n_candidates"] = int(n.item())
    return g._cache["n_candidates"]


def rescore_exact(g: CSRGraph, screen: Screen, keys: torch.Tensor, bar):
    """Exact scores of the screened survivors ``keys`` (v << 32 | u, u < v): float32 of the int64 sum of the 2^-40 fixed-point
    terms over the common neighbours -- bit-identical to eps_filter_scan's / eps_expand_fill's sums.  -> (keys, scores) in
    another order; candidates that do not exceed ``bar`` (a 1-element device tensor or None) get key -1 / score -inf, like
    an untouched slot.
    Unit values: the survivors are pairs of hubs (under hubs-first labels u is the heavier one and recurs in hundreds of
    pairs), so they are sorted by u and go through eps_rescore_runs (one LDS bitmap of N(u) per run of a long row, staged
    short rows otherwise).  Stored values: eps_rescore_weighted (both values of every common neighbour)."""
    if keys.numel() == 0:
        return keys, torch.zeros(0, dtype=torch.float32, device=keys.device)
    if screen.val is not None:
        vals = ops.rescore_weighted(g.rowptr, g.col, screen.val, screen.node_w, g.n_rows, keys)
    else:
        by_u = ops.sort_pairs_by_u(keys, max(1, int(g.n_rows - 1).bit_length()), RESCORE_V_BLOCK)     # (u << 32 | v): runs of equal u
        vals = ops.rescore_runs(g.rowptr, g.col, screen.fixw, g.n_rows, by_u)
        keys = ((by_u & 0xFFFFFFFF) << 32) | (by_u >> 32)
    if bar is not None:
        keep = vals > bar
        keys = torch.where(keep, keys, torch.full_like(keys, -1))
        vals = torch.where(keep, vals, torch.full_like(vals, float("-inf")))
    return keys, vals


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


BAR_FROM_HISTOGRAM = True  # repeated scans of a graph at the same K read the bar estimate off a score histogram around the previous bar
SAMPLE_RANK = 4096         # the sample is thinned (stride doubled) while the rank read off it stays at least this large


def sample_stride(k: int, safety: Optional[float] = None) -> int:
    """Stride of the bar sample for a top-``k`` request: ``SAMPLE_STRIDE``, doubled while the rank m = safety x k / 2 / stride
    that is read off the sample stays >= SAMPLE_RANK (an order statistic of rank m is good to ~1 / sqrt(m): 1.6 % at 4096, far
    inside what SAFETY absorbs) -- a large k does not need a sample that grows with it.  (Module globals read at CALL time:
    tests patch them.)"""
    safety = SAFETY if safety is None else float(safety)
    stride = int(SAMPLE_STRIDE)
    while safety * k / 2 / (2 * stride) >= SAMPLE_RANK and stride < 1 << 16:
        stride *= 2
    return stride


def estimate_bar(g: CSRGraph, fixw: torch.Tensor, k: int, stride: Optional[int] = None, safety: Optional[float] = None,
                 rank: int = 0, world: int = 1, screen=None):
    """Score bar (1-element float32 device tensor) that about ``safety * k`` directed candidates are expected to reach,
    from a scan of every ``stride``-th column of the heaviest-first order; ``None`` = no bar (keep everything).  The bar is
    INCLUSIVE of the sample's m-th best score (one float below it: the kernel keeps scores strictly above its threshold), so
    a run of tied scores at the estimate passes as a whole.  ``world`` > 1: rank r scans its share of the sample and the m-th
    best of the union comes from all-reduced radix-select histograms (ops.kth_largest_dist) -- same bar on every rank, no
    host round trip after the first call on a graph."""
    safety = SAFETY if safety is None else float(safety)
    stride = sample_stride(k, safety) if stride is None else int(stride)
    mine, bound_mine, bound_all = sample_columns(g, stride, rank, world)
    if bound_all == 0:
        return None
    m = int(safety * k / 2 / stride) + 1                 # unordered pairs of the SAMPLE above the bar we aim at
    if m >= bound_all:                                   # (candidates <= half paths: the sample cannot hold that many)
        return None
    slack = _CHUNK_SLACK if screen is None else _PIECE_SLACK
    # (room for every candidate of the sample: at most its half paths -- a list that still overflows only thins the sample)
    res = _launch(g, fixw, mine, float("-inf"), min(bound_mine + slack, ops.SURVIVOR_SLOTS_MAX), scores_only=True, screen=screen,
                  sample_key=("scan_sample", stride, rank, world))
    # no bar: every candidate of the sample holds a slot, untouched slots are -inf (fewer than m candidates -> bar -inf).
    # (one-pass kernel: screening scores, at most a few 2^-shift above the exact ones -- an estimate either way)
    if screen is None:
        kth = ops.kth_largest_dist(res.val, m, world)
        return torch.nextafter(kth, torch.full_like(kth, float("-inf")))
    # One launch (select + "one float below"), and on a job NO round of histogram all-reduces: every rank reads the
    # (m / world)-th best of its own share of the sample -- the same strata, so an estimate of the same quantile -- and the job
    # takes the lowest of them (one all-reduce of a word).  Any bar is a valid bar: the verification after the main launch is
    # what makes the result exact.
    m_loc = m if world == 1 else (m + world - 1) // world
    hint = screen.bar_hint.get((k, stride, float(safety))) if BAR_FROM_HISTOGRAM else None
    if hint is not None and hint > 0.0:
        # r06: a graph that is scanned again reads the bar off a score-bucket histogram around the bar its last scan at this K
        # ended with (two short launches instead of the four-round select): buckets are 2^-8 of a score's distance to the base --
        # half the old bar -- so the value is at most 0.4 % below the exact order statistic: an estimate either way, and the
        # verification after the main launch is what makes the step exact.  A sample whose m-th best fell below the base answers
        # -inf: the main launch then keeps everything, overflows its list and is corrected like any bar that was too low.
        base = screen.bar_hint.get(("base", hint))
        if base is None:
            if len(screen.bar_hint) > 32:
                screen.bar_hint.clear()                  # (K and bars of a long-lived graph object change: start over)
            base = screen.bar_hint[("base", hint)] = torch.full((1,), 0.5 * hint, dtype=torch.float32, device=g.device)
        ops.score_hist(None, res.val, res.count_ptr, base)
        bar = ops.score_pick_compact(None, res.val, res.count_ptr, base, m_loc, mode=1)[4]
    else:
        bar = ops.select_compact(None, res.val, m_loc, res.count_ptr, mode=1, compact=False)[4]
    if _sharded(world):
        from . import dist as epd
        # (a rank whose share of the sample holds fewer than m_loc candidates has no estimate: it does not vote)
        inf = torch.full_like(bar, float("inf"))
        vote = epd.all_reduce_min_(torch.where(torch.isinf(bar), inf, bar))
        bar = torch.where(torch.isinf(vote), -inf, vote)
    return bar


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


TAIL_DEVICE = True           # one rank, under a bar: the step's selections come from score-bucket histograms (csrc/tail_sort.hip)
TAIL_SORT = "library"        # ... and its two orderings from the library sorts ("library": sizes from the host, two host reads per step) or
                             # from the one-launch cooperative radix sorts with device-side sizes ("radix": one host read, but slower on
                             # MI355X -- see scan_topk)
EXACT_SCREENING = True       # uniform weights whose screening sums are exact skip the re-scoring (Screen.exact; tests switch it off to compare)
RESCORE_V_BLOCK = 12         # the survivors are re-scored in blocks of 2^12 consecutive v (then u, then v): the workgroups that run side
                             # by side stream the rows of one block of v out of the L2.  r04 (one hand-out counter): 2^8 3.32, 2^10 2.78,
                             # 2^12 2.60, 2^14 2.77 ms.  r05 (chunks dealt to the XCDs in groups, 16-byte row loads), measured INSIDE the
                             # step, same box: 2^9 2.69, 2^10 2.65, 2^11 2.58, 2^12 2.40, 2^14 2.61, 2^16 2.60, 2^20 2.72; the stand-alone
                             # timing of the same kernel preferred 2^10 (2.45 vs 2.64): the step's L2 arrives full of the scan's tables,
                             # and the step is what counts (profiles/r05/rescore_variants.txt)
                             # Blocks multiply the RUNS of equal u (one LDS bitmap each): eps_sort_pairs_by_u keeps them only while a run
                             # still averages 64 pairs (AA / CN: 500; RA, 168 k distinct u: 1.8 -- 6.7 ms blocked, 2.1 ms by (u, v))
DIST_HIST = True             # a sharded step exchanges score HISTOGRAMS (25 KB per rank) for its cut and its deal plan, not the scores
DIST_ROWS_MIN = 1 << 15      # selected pairs from which the final ordering of a sharded step is dealt over the ranks


def score_splitters(vals: torch.Tensor, world: int) -> torch.Tensor:
    """``world - 1`` scores, descending, that cut a list of scores into ``world`` ranges of about equal size: range r holds the
    scores in [splitter[r], splitter[r - 1]) (open at both ends of the list).  Read off a sorted thinning of the list (every
    n / 32768-th value): every rank holds the same gathered list in the same order, so every rank computes the same splitters
    without a collective; equal scores always fall into one range, so the ranges concatenate to the declared order."""
    n = vals.numel()
    samp = torch.sort(vals[::max(1, n // 32768)], descending=True).values
    idx = (torch.arange(1, world, device=vals.device) * samp.numel()) // world
    return samp[idx.clamp(max=samp.numel() - 1)]


def score_range_counts(vals: torch.Tensor, splitters: torch.Tensor) -> torch.Tensor:
    """int64[world]: how many scores fall into each range of ``score_splitters`` (range 0 = the best scores)."""
    r = (vals.unsqueeze(1) < splitters.unsqueeze(0)).sum(1)              # range of every score: splitters it lies below
    return torch.bincount(r, minlength=splitters.numel() + 1)


def _ordered_rows_distributed(keys: torch.Tensor, vals: torch.Tensor, k: int, bits: int, perm, rank: int, world: int,
                              rows_on: Optional[int] = None):
    """The k best directed rows of the selected pairs (all of them, gathered: the same arrays on every rank), ordered by the
    declared rule, with the ORDERING dealt over the ranks: rank r mirrors and sorts the pairs of score range r (1 / world of the
    rows: filter.py:160-161 sorts all E rows on one host), the sorted chunks are gathered in rank order -- which is the declared
    order, because a score range holds every pair of its scores.  One host read (the ranges' sizes, from one small tensor)."""
    from . import dist as epd
    sp = score_splitters(vals, world)
    counts = score_range_counts(vals, sp).tolist()                       # the one host read: pairs per range, on every rank alike
    lo = sp[rank:rank + 1] if rank < world - 1 else None
    hi = sp[rank - 1:rank] if rank > 0 else None
    mk, mv, _ = ops.compact_between(keys, vals, lo, hi)
    m = counts[rank]
    rk, rv = ops.select_rows(mk[:m].contiguous(), mv[:m].contiguous(), 2 * m, bits, perm)      # all 2 m rows of the range, ordered
    lens = [2 * c for c in counts]
    if rows_on is None:
        rows_k, rows_v = epd.gather_ragged(rk, lens), epd.gather_ragged(rv, lens)
    else:                                  # (the rows are wanted on one rank only: a gather instead of an all-gather)
        rows_k, rows_v = epd.gather_ragged_to(rk, lens, rows_on), epd.gather_ragged_to(rv, lens, rows_on)
        if rows_k is None:
            return None, None
    return rows_k[:k], rows_v[:k]


_XCHG_HEAD = 16                    # int32 words in front of a rank's scores in the step's one exchange (six int64 status words, padded)
_NEG_INF_BITS = -8388608           # float32 -inf as int32


_DEAL_SAMPLE = 1 << 15             # gathered scores the splitters are read off


def _deal_plan(scores_all: torch.Tensor, cut: torch.Tensor, world: int):
    """(splitters float32 [world - 1], counts int64 [world, world]) of the final ordering of a sharded step, from everybody's
    re-scored scores (``scores_all`` [world, room], -inf = none) and the job-wide cut -- device tensors of fixed shape, the same on
    every rank.  Range q holds the selected scores in [splitter[q], splitter[q - 1]); counts[r][q] = selected pairs of rank r in
    range q.  The splitters are order statistics of a fixed-stride sample of the gathered scores (sorted descending: the m
    sample values at or above the cut come first; splitter q = the (q m / world)-th of them) -- equal scores never straddle a
    boundary, so the ranges concatenate to the declared order however uneven they turn out."""
    flat = scores_all.reshape(-1)
    samp = torch.sort(flat[::max(1, flat.numel() // _DEAL_SAMPLE)], descending=True).values
    m = ((samp >= cut) & (samp > float("-inf"))).sum()
    idx = (torch.arange(1, world, device=flat.device) * m) // world
    sp = samp[idx.clamp(max=samp.numel() - 1)]
    sp = torch.where(m > 0, sp, torch.full_like(sp, float("inf")))                      # (no sample above the cut: one range takes all)
    live = (scores_all >= cut) & (scores_all > float("-inf"))
    rng_all = (scores_all.unsqueeze(2) < sp.view(1, 1, -1)).sum(2)                     # range of every gathered score
    cell = torch.arange(world, device=flat.device).unsqueeze(1) * world + rng_all       # (sender, range); a dead score counts nowhere
    cell = torch.where(live, cell, torch.full_like(cell, world * world))
    counts = torch.bincount(cell.reshape(-1), minlength=world * world + 1)[:world * world].view(world, world)
    return sp, counts


def _deal_rows(keys: torch.Tensor, vals: torch.Tensor, sp: torch.Tensor, c: List[List[int]], k: int, bits: int, perm,
               rank: int, world: int, rows_on: Optional[int]):
    """The k best directed rows, ordered by the declared rule, when every rank holds its OWN selected pairs (keys, vals) and the
    deal plan (``_deal_plan``: splitters on the device, the counts table read back with the step's status).  The final ordering
    is dealt over the ranks by score range: ONE all-to-all moves each selected pair to the rank that orders its range (1 / world
    of the pairs arrive per rank, not all of them on every rank as in r04), the rank mirrors and sorts them, and the sorted chunks
    go to ``rows_on`` (or to everyone) in rank order = the declared order.  No host read of its own."""
    from . import dist as epd
    rng = (vals.unsqueeze(1) < sp.unsqueeze(0)).sum(1)
    order = torch.sort(rng, stable=True).indices                                      # own pairs, grouped by destination
    packed = torch.stack([keys[order], vals[order].view(torch.int32).to(torch.int64)], 1).reshape(-1)       # (key, score bits) pairs
    got = epd.all_to_all_ragged(packed, [2 * x for x in c[rank]], [2 * c[r][rank] for r in range(world)]).view(-1, 2)
    mk, mv = got[:, 0].contiguous(), got[:, 1].to(torch.int32).view(torch.float32).contiguous()
    m = mk.numel()
    rk, rv = ops.select_rows(mk, mv, 2 * m, bits, perm)                                # all 2 m rows of the range, ordered
    lens = [2 * sum(c[r][q] for r in range(world)) for q in range(world)]
    if rows_on == "shards":
        # (r06) the rows stay where they were ordered: this rank's chunk is rows [offset, offset + len) of the declared order, cut
        # where the first k rows end -- no gather of 48 MB onto one rank; the proposal file is written as one shard per rank and
        # read back in rank order (proposals.load_proposals)
        offset = sum(lens[:rank])
        keep = max(0, min(lens[rank], k - offset))
        return rk[:keep], rv[:keep]
    if rows_on is None:
        rows_k, rows_v = epd.gather_ragged(rk, lens), epd.gather_ragged(rv, lens)
    else:
        rows_k, rows_v = epd.gather_ragged_to(rk, lens, rows_on), epd.gather_ragged_to(rv, lens, rows_on)
        if rows_k is None:
            return None, None
    return rows_k[:k], rows_v[:k]


def _f32_from_bits(word: int) -> float:
    import struct
    return struct.unpack("<f", struct.pack("<I", int(word) & 0xFFFFFFFF))[0]


def _capacity(slots_wanted: int, slack: Optional[int] = None) -> int:
    cap = int(slots_wanted) + (_CHUNK_SLACK if slack is None else slack)
    if cap > ops.SURVIVOR_SLOTS_MAX:
        raise ops._lib.EpsError(f"scan_topk: a survivor list of {cap} slots exceeds the {ops.SURVIVOR_SLOTS_MAX} a launch can "
                                "address (slots are 32-bit positions); use the block-streaming filter path for this set")
    return cap


def scan_topk(g: CSRGraph, node_w: torch.Tensor, k: int, rank: int = 0, world: int = 1, stats: Optional[dict] = None,
              relabel: bool = False, rows_on: Optional[int] = None):
    """Exact top-``k`` candidates of the whole graph: (pairs int64 [2,<=k] as (u; v), scores float32), best first.
    ``stats`` (optional dict) receives ``candidates`` (directed candidates scored), ``launches``, ``survivors``, ``bar``.
    ``relabel``: build the hubs-first copy of a large graph if it does not exist yet (see ``scan_graph``).
    ``rows_on`` (world > 1): the rank that wants the rows -- every other rank returns (None, None) and the ordered chunks travel
    to that rank alone (the proposal file is written by one rank); default: every rank gets them; ``"shards"``: every rank keeps
    the chunk of the declared order it ordered itself (rank r's rows follow rank r - 1's; together the first k rows) -- nothing is
    gathered, ``stats["shard"]`` = (first row of this rank's chunk, rows of all chunks).

    Per call, in the steady state (tables cached on the graph): sample launch -> bar (device; the lowest of the ranks' own
    estimates; r06: read off a score histogram around the previous bar when the graph was scanned at this K before) -> main launch,
    under a bar with skipped heads + eps_scan_refine (r06: the body compiled for its tables, the per-column pack from the graph's
    second scan on) -> LOCAL pre-filter of this rank's list (r06, one rank under a bar: a score-bucket histogram + pick-and-compact;
    its count -- one word -- is read to size the library sorts: host read 1) -> exact re-scoring of what passed (not for uniform
    weights whose screening sums are exact: Screen.exact) -> job-wide cut (one rank: the lower edge of the bucket of the
    ceil(k/2)-th best exact score; on a job one all-gather of the ranks' score HISTOGRAMS behind their status words, every rank
    derives the same cut and deal plan: ops.score_deal_plan) + compaction -> host read 2 of a status vector (slots, candidates,
    selected, cut, kernel status, the rank's pre-filter threshold, walked slots, the bar; all-gathered when world > 1), which
    VERIFIES the step: enough survivors, no list overflow, the cut at or above every rank's pre-filter threshold, usable heads --
    else the launch repeats with what was learnt -> mirrored + ordered (one rank: ops.select_rows_pairs writes the [2, K] tensor),
    on a job the ordering dealt over the ranks by score range, the rows sent to ``rows_on`` or left where they were ordered
    (ops.select_rows, _deal_rows).  TAIL_SORT = "radix": the same with one-launch cooperative sorts, sizes on the device, ONE host read."""
    if relabel and scan_plausible(g):
        scan_graph(g, build=True)        # (first: the symmetry check below then reads the copy's table, the one the scan needs)
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
    fixw = _scan_weights(g0, g, perm, node_w) if g.val is None else None
    mine = shard_columns(g, rank, world)
    total_half = total_half_paths(g)
    k2 = (k + 1) // 2                    # the k-th best directed row belongs to the ceil(k/2)-th best unordered pair
    screen = None
    if one_pass_available(g):
        sc = screen_weights(g0, g, perm, node_w)
        if sc.usable:
            screen = sc
    if screen is None and g.val is not None:
        raise ops._lib.EpsError("scan_topk: this weighted graph / weight table does not fit the one-pass scan (see scan_available)")
    slack = _CHUNK_SLACK if screen is None else _PIECE_SLACK
    launches = 0
    bar = None
    rescore_all = False                  # (set when a rank's pre-filter threshold turned out to lie above the job-wide cut)
    if total_half > SMALL_SET and SAFETY * k < total_half:
        bar = estimate_bar(g, fixw, k, rank=rank, world=world, screen=screen)
        launches += 1
    # unordered survivors expected: SAFETY x k / 2 over all ranks; the list holds four times that (the estimate has been within
    # a factor 0.35 .. 1.7), or every candidate without a bar
    wanted = (2 * total_half if bar is None else int(2 * SAFETY * k)) // world
    capacity = _capacity(wanted, slack)
    neg_inf = torch.full((1,), float("-inf"), device=dev)
    # skipped heads (csrc/scan_heads.hip): unit-valued graphs with a plan table, under a bar
    use_heads = (HEADS and screen is not None and screen.has_plan and screen.ssum is not None and g.val is None
                 and total_half >= HEAD_MIN_PATHS)
    _count_scan(g, screen)
    head_list, head_trouble, head_stale = HEAD_LIST, 0, 0
    sketch, sketch_void, sketch_ran = SKETCH_PIECES, 0, False
    n_rescored = None
    ht = None
    while True:
        if launches >= MAX_LAUNCHES:
            raise ops._lib.EpsError(f"scan_topk: no usable bar after {launches} launches (k = {k}, capacity {capacity})")
        # sketch pieces (and the plan made for them: packed pieces of twice the paths) when the bar is several of the heaviest weight
        # (not when the screening sums ARE the scores -- Screen.exact, common neighbours: a sketch piece's sums are upper bounds, and
        #  nothing re-scores them there)
        def sketch_for(budget):
            return bool(sketch and screen is not None and screen_variant(g) == 2 and not screen.exact
                        and (int(screen.w_max_units) + 2) * SKETCH_PIECE_PATHS < 1 << 32     # (no slot can wrap: screen_weights saw to it)
                        and budget >= HEAD_BETA * SKETCH_MIN_PATHS * screen.w_max_units)
        ht = (_heads_for(g, screen, bar, lambda budget: sketch_for(budget) and SKETCH_WIDE) if use_heads and bar is not None else None)
        want_sketch = ht is not None and sketch_for(ht.budget)
        walked_cap = max(1, min(int(head_list * capacity), ops.SURVIVOR_SLOTS_MAX)) if ht is not None else 0
        res = _launch(g, fixw, mine if ht is None else live_columns(g, screen, ht, rank, world), float("-inf") if bar is None else bar,
                      capacity, both=True, screen=screen, heads=ht, walked_capacity=walked_cap,
                      sketch=want_sketch)
        launches += 1
        l_keys, l_vals = res.key, res.val
        status, pre_thr = None, None
        rows = None
        fast = (TAIL_DEVICE and not _sharded(world) and screen is not None and g.val is None and bar is not None and not rescore_all
                and k2 + (1 << 16) < (1 << 30))
        if fast:
            # r06: the step's selections come from score-bucket histograms instead of four-round radix selects: the pre-filter
            # threshold and the cut are the lower edges of the buckets that hold the k2-th best screening / exact score (<= the exact
            # order statistics: a few pairs more are re-scored, mirrored and sorted; the K rows are the same -- both are verified
            # below).  The two orderings: TAIL_SORT.
            a, b = screen.lower_params(max_degree(g), None if ht is None else ht.d_used)
            bits = max(1, min(32, int(g.n_rows - 1).bit_length()))
            room = min(res.capacity, 2 * k2 + (1 << 16))
            while TAIL_SORT == "library":
                # The selections from score-bucket histograms (two launches each instead of a four-round cooperative select), the two
                # orderings by the library sorts: those take their sizes from the host, so the pre-filter's count is read back (the
                # second host read of the step is the status vector below).  Measured faster than the one-launch cooperative radix sorts
                # of TAIL_SORT = "radix" on MI355X: a grid-wide hand-over inside a kernel costs ~50 us there, a kernel boundary ~10
                # (profiles/r06/sort_probe.txt).
                ops.score_hist(res.key, res.val, res.count_ptr, bar)
                c_keys, c_vals, n_valid, _, pre_thr = ops.score_pick_compact(res.key, res.val, res.count_ptr, bar, k2, mode=2,
                                                                             params=(a, b, 4e-6), room=room, want_vals=screen.exact)
                nv = int(n_valid.item())                                                         # (one word: sizes the sorts)
                if nv > room:
                    room = min(res.capacity, nv)
                    continue
                if screen.exact:
                    x_keys, x_vals, swap = c_keys[:nv], c_vals[:nv], False
                else:
                    x_keys = ops.sort_pairs_by_u(c_keys[:nv], bits, RESCORE_V_BLOCK)              # (u << 32 | v): runs of equal u
                    x_vals = ops.rescore_runs(g.rowptr, g.col, screen.fixw, g.n_rows, x_keys)
                    swap = True
                ops.score_hist(x_keys, x_vals, None, bar, above=bar)
                sel_k, sel_v, n_sel, cut, _ = ops.score_pick_compact(x_keys, x_vals, None, bar, k2, above=bar, swap_halves=swap, room=max(nv, 1))
                # (the status vector as 32-bit words -- views, ONE cat launch, no dtype conversions: [slots lo hi, candidates lo hi,
                #  n_sel lo hi, cut, kernel status, pre-filter threshold, walked slots lo hi, bar])
                w32 = torch.cat([res.rec[1:2].view(torch.int32), res.rec[4:5].view(torch.int32), n_sel.view(torch.int32),
                                 cut.view(torch.int32), res.status, pre_thr.view(torch.int32),
                                 (res.walked_slots if res.walked_slots is not None else res.rec[1:2]).view(torch.int32),
                                 bar.view(torch.int32)]).tolist()                                 # the host read of the step
                u32 = lambda x: x & 0xFFFFFFFF                                                   # noqa: E731
                table = [[u32(w32[0]) | (w32[1] << 32), u32(w32[2]) | (w32[3] << 32), u32(w32[4]) | (w32[5] << 32), w32[6], w32[7], w32[8],
                          (u32(w32[9]) | (w32[10] << 32)) if res.walked_slots is not None else 0, w32[11]]]
                n_rescored = 0 if screen.exact else nv
                break
            while TAIL_SORT != "library":
                ops.score_hist(res.key, res.val, res.count_ptr, bar)
                c_keys, c_vals, n_valid, _, pre_thr = ops.score_pick_compact(res.key, res.val, res.count_ptr, bar, k2, mode=2,
                                                                             params=(a, b, 4e-6), room=room, want_vals=screen.exact)
                if screen.exact:                     # (screening sums that ARE the exact scores: Screen.exact)
                    x_keys, x_vals, swap = c_keys, c_vals, False
                else:
                    x_keys = ops.radix_sort_by_u(c_keys, n_valid, bits, RESCORE_V_BLOCK)          # (u << 32 | v): runs of equal u
                    x_vals = ops.rescore_runs_dev(g.rowptr, g.col, screen.fixw, g.n_rows, x_keys, n_valid)
                    swap = True
                ops.score_hist(x_keys, x_vals, n_valid, bar, above=bar)
                sel_k, sel_v, n_sel, cut, _ = ops.score_pick_compact(x_keys, x_vals, n_valid, bar, k2, above=bar, swap_halves=swap, room=room)
                r_pairs, r_scores, n_rows = ops.radix_sort_rows(sel_k, sel_v, n_sel, k, bits, perm)
                zero = torch.zeros(1, dtype=torch.int64, device=dev)
                st = torch.cat([res.rec[1:2], res.rec[4:5], n_sel, cut.view(torch.int32).to(torch.int64), res.status.to(torch.int64),
                                pre_thr.view(torch.int32).to(torch.int64), res.walked_slots if res.walked_slots is not None else zero,
                                bar.view(torch.int32).to(torch.int64) if bar is not None else zero, n_valid, n_rows])
                row = st.tolist()                                                                # the host read of the step
                nv = row[8]
                if nv <= room:
                    table = [row[:8]]
                    n_rescored = 0 if screen.exact else nv
                    rows = (r_pairs, r_scores, row[9])
                    break
                room = min(res.capacity, nv)        # (a level of tied scores at the pre-filter threshold holds more than the room: once more)
        elif screen is not None:
            # the one-pass kernel screens with upper bounds: its survivors are re-scored exactly (those that do not exceed the
            # bar after all drop out), so from here on the list holds eps_filter_scan's scores bit for bit.
            # Not all of them need it: a screening score s exceeds the exact one by less than what `screen.lower_bound` takes
            # off (a monotone lower bound of the exact score).  With t = lower(the k_pre-th best screening score of THIS rank's
            # list), every pair that is not re-scored has an exact score < t; the job-wide cut found below (the k2-th best exact
            # score among the re-scored pairs of all ranks) is then final iff it is >= every rank's t -- which the status
            # table verifies.  k_pre = k2 / world: the shards are dealt round-robin out of one heaviest-first order, so a rank's
            # (k2 / world)-th best sits at the job's k2-th best, and lower() leaves ~4 % of margin; world = 1: the r03 rule.
            # One launch: select + threshold + compaction, bounded by the list's slot counter (no collective, no pass over the
            # list's unused capacity).
            a, b = screen.lower_params(max_degree(g), None if ht is None else ht.d_used)
            k_pre = 0 if rescore_all else (k2 + world - 1) // world
            # (the outputs hold 2 x k_pre pairs -- lower() keeps ~10 % more than k_pre -- not a copy of the list's worst-case size;
            #  a level of tied scores at the threshold may hold more: then the call is repeated with room for all of them)
            room = res.capacity if rescore_all else min(res.capacity, 2 * k_pre + (1 << 16))
            room_x = room              # (what the r05 score exchange sends: the same on every rank -- a retry below grows `room` only)
            c_keys, c_vals, n_valid, _, pre_thr = ops.select_compact(l_keys, l_vals, k_pre, res.count_ptr, mode=2, params=(a, b, 4e-6), room=room)
            nv = int(n_valid.item())                                                      # (one word: sizes the re-scoring)
            if nv > room:
                c_keys, c_vals, n_valid, _, pre_thr = ops.select_compact(l_keys, l_vals, k_pre, res.count_ptr, mode=2, params=(a, b, 4e-6),
                                                                         room=min(res.capacity, nv))
                nv = int(n_valid.item())
            if screen.exact:
                l_keys, l_vals = c_keys[:nv], c_vals[:nv]        # (screening sums that ARE the exact scores: Screen.exact)
            else:
                l_keys, l_vals = rescore_exact(g, screen, c_keys[:nv], bar)
            n_rescored = 0 if screen.exact else nv
            status = res.status
        # the job-wide cut: the k2-th best survivor score over all ranks (-inf when fewer survived); untouched slots are -inf
        scores_all = None
        if not fast:                   # (the fast path assembles its own status vector: no conversions on its stream)
            zero = torch.zeros(1, dtype=torch.int64, device=dev)
            st_tail = [status.to(torch.int64) if status is not None else zero,                          # kernel status,
                       pre_thr.view(torch.int32).to(torch.int64) if pre_thr is not None else zero,      # pre-filter threshold bits,
                       res.walked_slots if res.walked_slots is not None else zero,                      # slots of the walked list,
                       bar.view(torch.int32).to(torch.int64) if bar is not None else zero]               # the bar's bits (head launches)
        if fast:
            pass
        elif _sharded(world) and screen is not None and not rescore_all and bar is not None and DIST_HIST:
            # r06 -- ONE exchange for the cut AND the step's status, 25 KB per rank whatever the lists hold: every rank's histogram of
            # its re-scored scores over order-preserving buckets of their distance to the bar (ops.score_hist_into: the bar is the
            # same on every rank) behind six status words, all-gathered.  From the same table every rank then derives, in ONE small
            # launch (ops.score_deal_plan) and without asking again, the job-wide cut -- the lower edge of the bucket that holds the
            # k2-th best score: a few pairs more than k2 are selected, the K rows are the same --, everybody's selection count, the
            # splitters of the final ordering (bucket edges) and who sends how many pairs of which range to whom (_deal_rows).
            # (r05: every rank's SCORES, 2.3 MB per rank, and a replicated select over all of them; r04: four all-reduced histogram
            #  rounds for the cut, one all-gather for the status, two for the selected pairs.)
            from . import dist as epd
            bins = ops.score_bins()
            head = torch.cat([res.rec[1:2], res.rec[4:5]] + st_tail).view(torch.int32)                 # 6 x int64
            send = torch.zeros(_XCHG_HEAD + bins, dtype=torch.int32, device=dev)
            send[:head.numel()] = head
            ops.score_hist_into(l_keys, l_vals, None, bar, send[_XCHG_HEAD:], above=bar)
            got = epd._gather_into(send, world).view(world, _XCHG_HEAD + bins)
            cut, deal_sp, deal_counts, nsel_all = ops.score_deal_plan(got[:, _XCHG_HEAD:], k2, bar)
            sel_k, sel_v, n_sel = ops.compact_at_least(l_keys, l_vals, cut)
            heads6 = got[:, :12].contiguous().view(torch.int64)                                          # [world, 6]
            st_all = torch.cat([heads6[:, :2], nsel_all.unsqueeze(1), cut.view(torch.int32).to(torch.int64).expand(world, 1),
                                heads6[:, 2:]], 1)
            flat = torch.cat([st_all.reshape(-1), deal_counts.reshape(-1)]).tolist()            # the host read of the step
            table = [flat[r * 8:(r + 1) * 8] for r in range(world)]
            deal_c = [flat[8 * world + r * world:8 * world + (r + 1) * world] for r in range(world)]
            scores_all = True                     # (marks "the deal plan is at hand" for the ordering below)
        elif _sharded(world) and screen is not None and not rescore_all:
            # (no bar -- small candidate sets -- or DIST_HIST off: the r05 exchange of the scores themselves.  Its length must be the
            #  same on every rank and hold every rank's list: a level of tied scores at a rank's pre-filter threshold may have grown
            #  that rank's list beyond the room it started with -- ADVICE r05 -- so the ranks agree on the longest list first: one
            #  all-reduced word)
            from . import dist as epd
            room_x = max(room_x, -int(epd.all_reduce_min_(torch.tensor([-int(l_vals.numel())], dtype=torch.int64, device=dev)).item()))
            # ONE exchange for the cut AND the step's status: every rank's re-scored scores (at most `room`, -inf beyond its own) behind
            # six status words, all-gathered.  Each rank then finds the job-wide cut itself -- one select launch over the gathered
            # scores: the same values in the same order everywhere, so the same cut without a broadcast -- and knows every rank's
            # selection count, the splitters of the final ordering and who sends how much to whom (_deal_rows) without asking again.
            # (r04: four all-reduced histogram rounds for the cut, one all-gather for the status, two for the selected pairs.)
            from . import dist as epd
            head = torch.cat([res.rec[1:2], res.rec[4:5]] + st_tail).view(torch.int32)                 # 6 x int64
            send = torch.full((_XCHG_HEAD + room_x,), _NEG_INF_BITS, dtype=torch.int32, device=dev)
            send[:head.numel()] = head
            send[_XCHG_HEAD:_XCHG_HEAD + l_vals.numel()] = l_vals.view(torch.int32)
            got = epd._gather_into(send, world).view(world, _XCHG_HEAD + room_x)
            scores_all = got[:, _XCHG_HEAD:].contiguous().view(torch.float32)
            cut = ops.select_compact(None, scores_all.reshape(-1), k2, compact=False)[3]
            sel_k, sel_v, n_sel = ops.compact_at_least(l_keys, l_vals, cut)
            nsel_all = ((scores_all >= cut) & (scores_all > float("-inf"))).sum(1)
            heads6 = got[:, :12].contiguous().view(torch.int64)                                          # [world, 6]
            st_all = torch.cat([heads6[:, :2], nsel_all.unsqueeze(1), cut.view(torch.int32).to(torch.int64).expand(world, 1),
                                heads6[:, 2:]], 1)
            # (the splitters of the final ordering and the table of who sends how many pairs of which range to whom ride along
            #  in the same host read: fixed shapes, nothing here waits for a size)
            deal_sp, deal_counts = _deal_plan(scores_all, cut, world)
            flat = torch.cat([st_all.reshape(-1), deal_counts.reshape(-1)]).tolist()            # the host read of the step
            table = [flat[r * 8:(r + 1) * 8] for r in range(world)]
            deal_c = [flat[8 * world + r * world:8 * world + (r + 1) * world] for r in range(world)]
        else:
            if not _sharded(world) and screen is not None:
                sel_k, sel_v, n_sel, cut, _ = ops.select_compact(l_keys, l_vals, k2)          # (one launch: select + compaction)
            else:
                cut = ops.kth_largest_dist(l_vals, k2, world)
                sel_k, sel_v, n_sel = ops.compact_at_least(l_keys, l_vals, cut)
            st = torch.cat([res.rec[1:2], res.rec[4:5], n_sel, cut.view(torch.int32).to(torch.int64)] + st_tail)
            if _sharded(world):
                from . import dist as epd
                table = torch.stack(epd.all_gather_list(st)).tolist()                          # the host read of the step
            else:
                table = [st.tolist()]
        slots_r, ncand_r, nsel_r = [t[0] for t in table], [t[1] for t in table], [t[2] for t in table]
        if any(t[4] & 8 for t in table):
            # (status bit 3: a sketch piece had more ids to report than its set holds -- a bar far below what K asks for: the launch is
            #  void, the call goes on with hashed packed pieces)
            sketch = False
            sketch_void += 1
            launches -= 1                                    # (the repeat is the same launch again, not a corrected bar)
            if launches + 2 > MAX_LAUNCHES:
                raise ops._lib.EpsError("scan_topk: launches with sketch pieces kept failing")
            continue
        sketch_ran = any(t[4] & 16 for t in table)            # (status bit 4, informational: sketch pieces ran in this launch)
        if any(t[4] & ~(4 | 16) for t in table):
            raise ops._lib.EpsError("scan_topk: eps_scan_screen reported a full hash table (status %s)" % [t[4] for t in table])
        if ht is not None:
            # The head table was taken without looking at the bar.  A head as heavy as the bar (status bit 2: the bar fell since the
            # table was built) or a walked list that overflowed: the launch is void -- a table for THIS bar is built (head_cur
            # dropped: the next _heads_for reads the bar), the list grows; twice in a row: this call goes on without heads.
            bar_units = _f32_from_bits(table[0][7]) * 2.0 ** screen.shift
            # (... or the bar fell below the lowest one the table -- and its list of live columns -- was built for)
            overflow = any(t[6] > walked_cap for t in table)
            stale = any(t[4] & 4 for t in table) or ht.budget > HEAD_KEEP[1] * bar_units
            void = overflow or stale
            if void or not HEAD_KEEP[0] * bar_units <= ht.budget <= HEAD_KEEP[1] * bar_units:
                screen.head_cur = None                       # (a budget out of range only costs time: the NEXT launch rebuilds)
            if void:
                # (two causes, two remedies -- ADVICE r05: a table built for another bar is simply rebuilt; only an overflowing
                #  walked list grows the list and counts toward giving heads up for this call)
                if overflow:
                    head_trouble += 1
                    head_list *= 2
                else:
                    head_stale += 1
                use_heads = head_trouble < 2 and head_stale < 3
                launches -= 1                                # (the repeat is the same launch again, not a corrected bar)
                if launches + 2 > MAX_LAUNCHES:
                    raise ops._lib.EpsError("scan_topk: launches with skipped heads kept failing")
                continue
        n_cand_all, n_sel_all = sum(ncand_r), sum(nsel_r)
        cut_is_inf = (table[0][3] & 0xFFFFFFFF) == 0xFF800000
        if screen is not None and not rescore_all and not any(sl > capacity for sl in slots_r):
            # the pre-filter was sound iff the cut reaches every rank's threshold (floats compared through their bits on the host).
            # A cut of -inf -- fewer than k2 re-scored pairs job-wide -- is NOT exempt: with uneven shards one rank's
            # pre-filter may have held back pairs that belong to the k2 best while another rank had too few to fill its share, so
            # any finite threshold above a -inf cut sends the step round again with everything re-scored.  Only when no rank
            # filtered at all (every threshold -inf) does -inf mean "fewer than k2 pairs survived": the bar is lowered below.
            cut_f = float("-inf") if cut_is_inf else _f32_from_bits(table[0][3])
            if any(cut_f < _f32_from_bits(t[5]) for t in table):
                rescore_all = True         # (never seen with round-robin shards: a rank's list would have to sit far above the job's)
                launches -= 1              # (the repeat is the same launch again, not a corrected bar)
                if launches + 2 > MAX_LAUNCHES:
                    raise ops._lib.EpsError("scan_topk: the re-scoring pre-filter kept failing")
                continue
        if any(sl > capacity for sl in slots_r):
            # the bar was too low for the list.  What was kept is a subset of the survivors, so the k2-th best score among
            # it (the cut just computed, job-wide) is a lower bound of the final cut: scan again just below it, with more
            # room (a level of tied scores -- common-neighbour counts -- may hold far more pairs than K).
            if not cut_is_inf:
                bar = torch.nextafter(cut, neg_inf)
            wanted *= 4
            capacity = _capacity(wanted, slack)
            continue
        # (a launch with skipped heads counts the candidates its walk touched, not all of them: fewer than k2 selected under a bar
        #  then always means "lower the bar" -- the launch without one counts them all)
        if bar is not None and n_sel_all < (k2 if ht is not None else min(k2, n_cand_all)):
            # fewer than k above the bar: lower it (a quarter of the sample rank each time, then no bar at all)
            bar = None if launches > 3 else estimate_bar(g, fixw, k, safety=SAFETY * 8 ** (launches - 1), rank=rank, world=world,
                                                         screen=screen)
            launches += 1
            if bar is None:
                wanted = 2 * total_half // world
                capacity = _capacity(wanted, slack)
            continue
        break
    if screen is not None and bar is not None and len(table[0]) > 7:
        # (the bar this scan ended with, for the next scan's estimate at the same K: see estimate_bar)
        bar_f = _f32_from_bits(table[0][7])
        if 0.0 < bar_f < float("inf"):
            screen.bar_hint[(k, sample_stride(k), float(SAFETY))] = bar_f
    keys, vals = sel_k[:nsel_r[rank]], sel_v[:nsel_r[rank]]      # (still in the scanned graph's labels: select_rows maps them back)
    bits = max(1, min(32, int(g.n_rows - 1).bit_length()))
    shard_info = None
    if rows is not None:
        keys = vals = None                          # (the device tail has ordered the rows already)
    elif _sharded(world) and scores_all is not None and n_sel_all >= DIST_ROWS_MIN:
        keys, vals = _deal_rows(keys.contiguous(), vals.contiguous(), deal_sp, deal_c, k, bits, perm, rank, world, rows_on)
        if rows_on == "shards":
            lens = [2 * sum(deal_c[r][q] for r in range(world)) for q in range(world)]
            shard_info = (min(sum(lens[:rank]), k), min(sum(lens), k))
    elif _sharded(world):
        from . import dist as epd
        keys, vals = epd.gather_ragged(keys, nsel_r), epd.gather_ragged(vals, nsel_r)
        if n_sel_all >= DIST_ROWS_MIN:
            keys, vals = _ordered_rows_distributed(keys.contiguous(), vals.contiguous(), k, bits, perm, rank, world,
                                                   None if rows_on == "shards" else rows_on)
        elif rows_on is None or rows_on == rank or rows_on == "shards":
            keys, vals = ops.select_rows(keys.contiguous(), vals.contiguous(), k, bits, perm)
        else:
            keys = vals = None
        if rows_on == "shards" and keys is not None:
            # (a short list, or the exchange without a deal plan: every rank holds all rows -- it keeps an even share of them)
            n_rows = keys.numel()
            lo_r, hi_r = rank * n_rows // world, (rank + 1) * n_rows // world
            shard_info = (lo_r, n_rows)
            keys, vals = keys[lo_r:hi_r], vals[lo_r:hi_r]
    elif not _sharded(world) and keys.is_cuda:
        # (one rank: the rows are written as the [2, K] proposal tensor by the sort's last kernel)
        rows = ops.select_rows_pairs(keys.contiguous(), vals.contiguous(), k, bits, perm)
        rows = (rows[0], rows[1], rows[0].shape[1])
        keys = vals = None
    elif rows_on is None or rows_on == rank or world == 1:
        keys, vals = ops.select_rows(keys.contiguous(), vals.contiguous(), k, bits, perm)
    else:
        keys = vals = None
    if stats is not None:
        # survivors: DIRECTED rows at or above the job-wide cut (what the selection orders); survivor_slots: list slots the
        # launches handed out (chunks: holes included); bar: None or a 1-element device tensor (float(bar) reads it)
        # candidates: DIRECTED 2-hop non-edges the filter covered.  A launch with skipped heads only counts those its walk touched
        # (`touched`); the exact number then comes from `candidate_count` (one counting scan per graph, cached) unless the caller
        # put count=False into ``stats`` (then candidates = None).
        touched = 2 * n_cand_all
        if ht is None:
            n_all = touched
            if world == 1:
                g._cache.setdefault("n_candidates", n_cand_all)
        else:
            n_all = 2 * candidate_count(g, screen, fixw, rank, world) if stats.get("count", True) else None
        stats.update(candidates=n_all, touched=touched, launches=launches, survivors=2 * n_sel_all, heads=ht is not None, shard=shard_info,
                     sketch=sketch_ran, sketch_void=sketch_void,
                     head_budget=None if ht is None else ht.budget * 2.0 ** -screen.shift,
                     walked_slots=sum(t[6] for t in table) if ht is not None else None, rescored=n_rescored,
                     survivor_slots=sum(min(s_, capacity) for s_ in slots_r), bar=bar)
    if rows is not None:
        return rows[0][:, :rows[2]], rows[1][:rows[2]]
    if keys is None:
        return None, None
    return torch.stack([keys & 0xFFFFFFFF, keys >> 32]), vals
