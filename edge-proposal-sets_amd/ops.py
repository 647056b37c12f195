"""Thin torch-tensor wrappers over the C ABI (include/eps_abi.h).  Every function requires its
tensors on a HIP device and launches on torch's current stream; nothing here computes on the
CPU."""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Tuple

import torch

from . import _lib

W_AA, W_RA = 0, 1


_EMPTY = {}


def _ptr(t: Optional[torch.Tensor]):
    """Device pointer for the C ABI.  An EMPTY tensor has no storage (data_ptr() == 0), which the ABI would take for a
    missing argument: it gets the address of a small per-device dummy instead (nothing is read through it: the sizes say 0)."""
    if t is None:
        return None
    if t.numel() == 0 and t.is_cuda:
        key = (t.device.type, t.device.index)
        if key not in _EMPTY:
            _EMPTY[key] = torch.zeros(8, dtype=torch.int64, device=t.device)
        return ctypes.c_void_p(_EMPTY[key].data_ptr())
    return ctypes.c_void_p(t.data_ptr())


def _stream(dev: torch.device):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _need_gpu(*tensors: Optional[torch.Tensor], row_strided=()) -> torch.device:
    """All tensors on one HIP device and contiguous; those listed in ``row_strided`` may be 2-D views
    with unit column stride and an arbitrary row stride (passed to the ABI as the leading dimension)."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.EpsError("edge-proposal-sets_amd ops run on the MI355X only: got a CPU tensor "
                                "(there is no CPU fallback; move inputs to 'cuda')")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise _lib.EpsError(f"tensors on different devices: {dev} vs {t.device}")
        if any(t is r for r in row_strided):
            if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
                raise _lib.EpsError("2-D operand must have unit column stride")
        elif not t.is_contiguous():
            raise _lib.EpsError("non-contiguous tensor passed to the C ABI")
    if dev is None:
        raise _lib.EpsError("no tensor given")
    return dev


def _chk(t: Optional[torch.Tensor], dtype: torch.dtype, name: str):
    if t is not None and t.dtype != dtype:
        raise _lib.EpsError(f"{name}: expected {dtype}, got {t.dtype}")


def device_info() -> Tuple[int, str]:
    lib = _lib.load()
    n = ctypes.c_int(0)
    buf = ctypes.create_string_buffer(256)
    _lib.check(lib.eps_device_info(ctypes.byref(n), buf, 256), "eps_device_info")
    return n.value, buf.value.decode()


def col_sums(rowptr, col, val, n_cols: int, f64: bool = False) -> torch.Tensor:
    """Column sums, accumulated in float64 on the device; returned as float32 (rounded once) or, ``f64=True``, as is."""
    dev = _need_gpu(rowptr, col, val)
    _chk(rowptr, torch.int64, "rowptr"); _chk(col, torch.int32, "col"); _chk(val, torch.float32, "val")
    if col.numel() == 0:                        # a graph without entries (its col[] has no storage to point at)
        return torch.zeros(n_cols, dtype=torch.float64 if f64 else torch.float32, device=dev)
    wide = torch.empty(n_cols, dtype=torch.float64, device=dev)
    out = None if f64 else torch.empty(n_cols, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_col_sums(_ptr(rowptr), _ptr(col), _ptr(val), rowptr.numel() - 1, n_cols,
                                            _ptr(wide), _ptr(out), _stream(dev)), "eps_col_sums")
    return wide if f64 else out


def node_weights(colsum: torch.Tensor, mode: int, f64: bool = False) -> torch.Tensor:
    """1/log(colsum) (AA) or 1/colsum (RA), inf -> 0: float32 from float32 sums, or float64 from float64 sums."""
    dev = _need_gpu(colsum)
    _chk(colsum, torch.float64 if f64 else torch.float32, "colsum")
    out = torch.empty(colsum.numel(), dtype=torch.float64 if f64 else torch.float32, device=dev)
    fn = _lib.load().eps_node_weights_f64 if f64 else _lib.load().eps_node_weights
    with torch.cuda.device(dev):
        _lib.check(fn(_ptr(colsum), colsum.numel(), mode, _ptr(out), _stream(dev)), "eps_node_weights")
    return out


GROUPED_MIN_RUN = 256  # mean pairs per run of equal v above which the column-run kernel is used


def v_runs_are_long(v: torch.Tensor) -> bool:
    """True when ``v`` (the reference's column-major candidate order) has runs long enough for the
    column-run kernel to pay: one elementwise pass + a reduction on the device."""
    n = v.numel()
    if n < 4 * GROUPED_MIN_RUN:
        return False
    n_runs = int((v[1:] != v[:-1]).sum().item()) + 1
    return n >= n_runs * GROUPED_MIN_RUN


def pair_scores(rowptr, col, val, node_w, n_nodes: int, u, v, want_count=True, want_cn=True, want_wsum=None,
                grouped=None):
    """-> (count int32[E] | None, cn float32[E] | None, wsum float32/float64[E] | None).
    node_w float64 selects the float64-accumulate kernel (cn is then not produced).
    ``grouped``: True -> column-run kernel (pair list sorted by v), False -> generic kernel, None -> decide
    from the run statistics of ``v``.  Results are identical either way."""
    dev = _need_gpu(rowptr, col, val, node_w, u, v)
    _chk(rowptr, torch.int64, "rowptr"); _chk(col, torch.int32, "col"); _chk(val, torch.float32, "val")
    _chk(u, torch.int32, "u"); _chk(v, torch.int32, "v")
    if u.numel() != v.numel():
        raise _lib.EpsError("u and v differ in length")
    if want_wsum is None:
        want_wsum = node_w is not None
    n = u.numel()
    lib = _lib.load()
    if grouped is None:
        grouped = v_runs_are_long(v)
    count = torch.empty(n, dtype=torch.int32, device=dev) if want_count else None
    with torch.cuda.device(dev):
        if node_w is not None and node_w.dtype == torch.float64:
            ws = torch.empty(n, dtype=torch.float64, device=dev) if want_wsum else None
            fn = lib.eps_pair_scores_grouped_f64 if grouped else lib.eps_pair_scores_f64
            _lib.check(fn(_ptr(rowptr), _ptr(col), _ptr(val), _ptr(node_w), n_nodes, _ptr(u), _ptr(v), n, _ptr(count),
                          _ptr(ws), _stream(dev)), "eps_pair_scores_f64")
            return count, None, ws
        _chk(node_w, torch.float32, "node_w")
        cn = torch.empty(n, dtype=torch.float32, device=dev) if want_cn else None
        ws = torch.empty(n, dtype=torch.float32, device=dev) if want_wsum else None
        fn = lib.eps_pair_scores_grouped if grouped else lib.eps_pair_scores
        _lib.check(fn(_ptr(rowptr), _ptr(col), _ptr(val), _ptr(node_w), n_nodes, _ptr(u), _ptr(v), n, _ptr(count),
                      _ptr(cn), _ptr(ws), _stream(dev)), "eps_pair_scores")
    return count, cn, ws


def expand_max_nodes() -> int:
    return int(_lib.load().eps_expand_max_nodes())


_EXPAND_WS = {}


def _expand_scratch(dev, n_bytes: int) -> torch.Tensor:
    """int64 scratch of the fill kernel (status word + per-workgroup path buckets): one grow-only buffer per device,
    reused by every block of a filter run (a fresh multi-GB allocation per block costs more than the kernel saves).
    Stream-ordered like any other tensor on the current stream, so expansions of one device must not run concurrently
    on several streams."""
    key = (dev.type, dev.index)
    n_words = (n_bytes + 7) // 8
    buf = _EXPAND_WS.get(key)
    if buf is None or buf.numel() < n_words:
        _EXPAND_WS.pop(key, None)
        buf = None                                   # let the old buffer go before the larger one is allocated
        buf = torch.empty(int(n_words * 1.25) + 1024, dtype=torch.int64, device=dev)
        _EXPAND_WS[key] = buf
    return buf


def max_column_paths(rowptr: torch.Tensor, col: torch.Tensor, v_lo: int, v_hi: int) -> int:
    """max over the columns v of [v_lo, v_hi) of sum_{w in N(v)} deg(w): the two-hop paths of the heaviest column, which
    sizes the per-workgroup buckets of the fused expansion."""
    if v_hi <= v_lo:
        return 0
    deg = rowptr[1:] - rowptr[:-1]
    lo, hi = int(rowptr[v_lo].item()), int(rowptr[v_hi].item())
    if hi == lo:
        return 0
    dsum = torch.cumsum(deg[col[lo:hi].long()], 0)
    ends = rowptr[v_lo + 1:v_hi + 1] - lo                      # exclusive end of every column's slice
    upto = torch.where(ends > 0, dsum[(ends - 1).clamp(min=0)], torch.zeros_like(ends))
    per_col = upto - torch.cat([upto.new_zeros(1), upto[:-1]])
    return int(per_col.max().item())


class ExpandResult(tuple):
    """(colptr, cand_u, cand_v, cn, score) of ``expand_candidates``; ``.pairs`` is the int32 [2,E] buffer cand_u and
    cand_v are rows of (None without cand_v), so the (u; v) list exists without a copy; ``.counts`` is set in the
    upper-bound layout (see ``expand_candidates``)."""
    pairs = None
    counts = None
    survivors = None
    status = None


_EXPAND_WS_LIMIT = 96 << 30       # bytes of bucket scratch we are willing to hold on a 288 GB device


def expand_workspace_fits(max_paths: int) -> bool:
    """Whether the bucket scratch for columns of up to ``max_paths`` two-hop paths stays within the budget."""
    return int(_lib.load().eps_expand_workspace_bytes(int(max_paths))) <= _EXPAND_WS_LIMIT


def expand_unit(rowptr, col, node_w, n_nodes: int, v_lo: int, v_hi: int, max_degree: int, splits=None, want_score=True,
                want_v=True, col_order=None, revpos=None, colptr_ub=None, total_ub=None):
    """The candidate list of columns [v_lo, v_hi) of a graph WITHOUT stored values, on the threshold scan's structure
    (eps_expand_unit_count / eps_expand_unit_fill, csrc/filter_scan.hip): same tuple and same bits as
    ``expand_candidates(rowptr, col, None, node_w, ...)`` with ``want_cn=False`` -- (colptr, cand_u, cand_v | None, None,
    score | None) -- at about half the time.  ``node_w`` None with ``want_score``: all-ones weights (the score is the
    common-neighbour count).  ``max_degree`` / ``splits``: the per-graph figures ``filter_scan`` takes.  ``revpos``
    (``reverse_positions``; symmetric pattern) selects the HALF list: column v holds its candidates u < v only.

    ``colptr_ub`` (int64[n_cols + 1] on the device: an exclusive prefix of upper bounds of the columns' candidate counts,
    ``candidates.segment_bounds``) + ``total_ub`` (its last entry as a Python int) select the ONE-PASS list
    (eps_expand_unit_list): no counting launch, no host read before the launch; column v fills the front of its segment, the rest
    of the segment is NOT written, ``.counts`` (int64[n_cols]) holds the real counts and there is no cand_v."""
    dev = _need_gpu(rowptr, col, node_w, col_order, splits, revpos, colptr_ub)
    if colptr_ub is not None:
        _chk(colptr_ub, torch.int64, "colptr_ub")
        if total_ub is None or want_v or colptr_ub.numel() != v_hi - v_lo + 1:
            raise ValueError("expand_unit: the one-pass list takes colptr_ub[n_cols + 1] + total_ub and writes no cand_v")
        _chk(rowptr, torch.int64, "rowptr"); _chk(col, torch.int32, "col"); _chk(node_w, torch.float32, "node_w")
        _chk(col_order, torch.int32, "col_order"); _chk(splits, torch.int32, "splits"); _chk(revpos, torch.int32, "revpos")
        if col_order is not None and col_order.numel() != v_hi - v_lo:
            raise ValueError("col_order must have one entry per column of the range")
        lib = _lib.load()
        with torch.cuda.device(dev):
            ws = _scan_scratch(dev, int(max_degree))
            counts = torch.zeros(v_hi - v_lo, dtype=torch.int64, device=dev)
            cand_u = torch.empty(int(total_ub), dtype=torch.int32, device=dev)
            score = torch.empty(int(total_ub), dtype=torch.float32, device=dev) if want_score else None
            status = torch.zeros(1, dtype=torch.int32, device=dev)       # (the call clears it itself; an empty range makes no call)
            fixw = None
            if want_score:
                fixw = fixed_weights(node_w if node_w is not None else torch.ones(n_nodes, dtype=torch.float32, device=dev))
            if v_hi > v_lo:
                _lib.check(lib.eps_expand_unit_list(_ptr(rowptr), _ptr(col), _ptr(revpos), _ptr(fixw), _ptr(splits), n_nodes, col.numel(),
                                                    int(max_degree), v_lo, v_hi, _ptr(col_order), _ptr(colptr_ub), _ptr(counts),
                                                    _ptr(cand_u), _ptr(score), _ptr(status), _ptr(ws), ws.numel() * 8, _stream(dev)),
                           "eps_expand_unit_list")
        out = ExpandResult((colptr_ub, cand_u, None, None, score))
        out.counts = counts
        out.status = status           # (device word: bit 1 = a bound was too small, bit 2 = a sum left the fixed-point range)
        return out
    _chk(revpos, torch.int32, "revpos")
    _chk(rowptr, torch.int64, "rowptr"); _chk(col, torch.int32, "col"); _chk(node_w, torch.float32, "node_w")
    _chk(col_order, torch.int32, "col_order"); _chk(splits, torch.int32, "splits")
    lib = _lib.load()
    n_cols = v_hi - v_lo
    if col_order is not None and col_order.numel() != n_cols:
        raise ValueError("col_order must have one entry per column of the range")
    with torch.cuda.device(dev):
        ws = _scan_scratch(dev, int(max_degree))
        counts = torch.zeros(n_cols, dtype=torch.int64, device=dev)
        colptr = torch.zeros(n_cols + 1, dtype=torch.int64, device=dev)
        if n_cols:
            _lib.check(lib.eps_expand_unit_count(_ptr(rowptr), _ptr(col), _ptr(revpos), _ptr(splits), n_nodes, col.numel(), int(max_degree),
                                                 v_lo, v_hi, _ptr(col_order), _ptr(counts), _ptr(ws), ws.numel() * 8,
                                                 _stream(dev)), "eps_expand_unit_count")
        torch.cumsum(counts, 0, out=colptr[1:])
        total = int(colptr[-1].item())
        pairs = torch.empty((2 if want_v else 1, total), dtype=torch.int32, device=dev)
        score = torch.empty(total, dtype=torch.float32, device=dev) if want_score else None
        if total:
            fixw = None
            if want_score:
                fixw = fixed_weights(node_w if node_w is not None else torch.ones(n_nodes, dtype=torch.float32, device=dev))
            status = torch.zeros(1, dtype=torch.int32, device=dev)
            _lib.check(lib.eps_expand_unit_fill(_ptr(rowptr), _ptr(col), _ptr(revpos), _ptr(fixw), _ptr(splits), n_nodes, col.numel(),
                                                int(max_degree), v_lo, v_hi, _ptr(col_order), _ptr(colptr), None,
                                                _ptr(pairs[0]), _ptr(pairs[1]) if want_v else None, _ptr(score), _ptr(status),
                                                _ptr(ws), ws.numel() * 8, _stream(dev)), "eps_expand_unit_fill")
            st = int(status.item())
            if st:
                raise _lib.EpsError("expand_unit: " + ("a column outgrew its segment; " if st & 2 else "")
                                    + ("a score left the fixed-point range (|sum| >= 2**23): use the pair kernels" if st & 4 else ""))
    out = ExpandResult((colptr, pairs[0], pairs[1] if want_v else None, None, score))
    out.pairs = pairs if want_v else None
    out.counts = None
    return out


def expand_candidates(rowptr, col, val, node_w, n_nodes: int, v_lo: int, v_hi: int, want_cn=True, want_score=True,
                      want_v=True, col_order=None, max_paths=None, colptr_ub=None, total_ub=None, cut=None, tile_ranks=0):
    """Fused 2-hop expansion of columns [v_lo, v_hi) of a SYMMETRIC adjacency (filter.py:96-109 + scoring).
    -> (colptr int64[n_cols+1], cand_u int32[E], cand_v int32[E] | None, cn int32[E] | None, score float32[E] | None);
    candidates are column-major, u ascending inside a column (the reference's order).  ``col_order`` (int32
    permutation of range(v_hi - v_lo), optional) is the order the columns are handed to the workgroups; the results
    do not depend on it.  ``max_paths`` (optional) is an upper bound of the two-hop paths of any column of the range
    (``max_column_paths``; callers that expand many blocks of one graph pass the cached figure).

    ``colptr_ub`` (int64[n_cols+1] on the device) + ``total_ub`` (its last entry, as a Python int) select the
    UPPER-BOUND layout: no counting pass and no host synchronisation before the launch; column v's candidates fill
    the front of [colptr_ub[v], colptr_ub[v+1]) and the rest of the segment is padded (cand_u -1, score -inf, cn 0).
    The result then carries ``.counts`` (int64[n_cols], real candidates per column) and E == total_ub.

    ``tile_ranks`` (0 = default): candidate ranks per LDS summation pass (eps_expand_fill_tiled); results do not depend on it.

    ``cut=(threshold, capacity)``: the kernel also reports the candidates whose score exceeds ``threshold`` (the
    streaming top-K's current K-th score) -- the result carries ``.survivors`` = (positions int64 ascending, scores) or
    None when more than ``capacity`` qualified.  With a cut, ``want_score=False`` skips the score array altogether."""
    dev = _need_gpu(rowptr, col, val, node_w, col_order, colptr_ub)
    _chk(rowptr, torch.int64, "rowptr"); _chk(col, torch.int32, "col"); _chk(val, torch.float32, "val")
    _chk(node_w, torch.float32, "node_w"); _chk(col_order, torch.int32, "col_order"); _chk(colptr_ub, torch.int64, "colptr_ub")
    lib = _lib.load()
    n_cols = v_hi - v_lo
    if col_order is not None and col_order.numel() != n_cols:
        raise ValueError("col_order must have one entry per column of the range")
    if colptr_ub is not None and (colptr_ub.numel() != n_cols + 1 or total_ub is None):
        raise ValueError("colptr_ub needs n_cols + 1 entries and total_ub")
    with torch.cuda.device(dev):
        counts = torch.empty(n_cols, dtype=torch.int64, device=dev)
        if colptr_ub is None:
            colptr = torch.zeros(n_cols + 1, dtype=torch.int64, device=dev)
            _lib.check(lib.eps_expand_count(_ptr(rowptr), _ptr(col), n_nodes, v_lo, v_hi, _ptr(col_order), _ptr(counts),
                                            _stream(dev)), "eps_expand_count")
            torch.cumsum(counts, 0, out=colptr[1:])
            total = int(colptr[-1].item())
        else:
            colptr, total = colptr_ub, int(total_ub)
        pairs = torch.empty((2 if want_v else 1, total), dtype=torch.int32, device=dev)
        cand_u = pairs[0]
        cand_v = pairs[1] if want_v else None
        cn = torch.empty(total, dtype=torch.int32, device=dev) if want_cn else None
        score = torch.empty(total, dtype=torch.float32, device=dev) if want_score else None
        cut_rec = cut_pos = cut_val = None
        if cut is not None and total:
            thr, cap = float(cut[0]), int(cut[1])
            cut_pos = torch.empty(cap, dtype=torch.int64, device=dev)
            cut_val = torch.empty(cap, dtype=torch.float32, device=dev)
            import struct
            head = struct.unpack("<q", struct.pack("<fI", thr, cap))[0]          # eps_score_cut: threshold, capacity
            cut_rec = torch.tensor([head, 0, cut_pos.data_ptr(), cut_val.data_ptr()], dtype=torch.int64, device=dev)
        scored = want_cn or want_score or cut_rec is not None
        if total:
            if max_paths is None:
                max_paths = max_column_paths(rowptr, col, v_lo, v_hi) if scored else 0
            ws_bytes = int(lib.eps_expand_workspace_bytes(int(max_paths) if scored else 0))
            if ws_bytes > _EXPAND_WS_LIMIT:
                raise _lib.EpsError(f"expand_candidates: a column with {max_paths} two-hop paths needs {ws_bytes >> 30} GiB "
                                    "of bucket scratch; score such graphs with the pair kernels")
            ws = _expand_scratch(dev, ws_bytes)
            _lib.check(lib.eps_expand_fill_tiled(_ptr(rowptr), _ptr(col), _ptr(val), _ptr(node_w), n_nodes, v_lo, v_hi,
                                                 _ptr(col_order), _ptr(colptr),
                                                 _ptr(counts) if colptr_ub is not None else None, _ptr(cand_u), _ptr(cand_v),
                                                 _ptr(cn), _ptr(score), _ptr(cut_rec), _ptr(ws), ws_bytes, int(tile_ranks),
                                                 _stream(dev)), "eps_expand_fill")
            if cut_rec is not None:                      # one read-back for the status word and the survivor count
                both = torch.stack([ws[0], cut_rec[1]]).tolist()
                status, n_cut = both[0] & 0xFFFFFFFF, both[1] & 0xFFFFFFFF
            else:
                status, n_cut = int(ws[0].item()) & 0xFFFFFFFF, 0
            if status:
                raise _lib.EpsError("expand_candidates: " + ("a column had more two-hop paths than max_paths allows; " if status & 1 else "")
                                    + ("a column had more candidates than its colptr_ub segment; " if status & 2 else "")
                                    + ("a score left the fixed-point range (|sum| >= 2**23): use the pair kernels" if status & 4 else ""))
    out = ExpandResult((colptr, cand_u, cand_v, cn, score))
    out.pairs = pairs if want_v else None
    out.counts = counts if colptr_ub is not None else None
    if cut_rec is not None and n_cut <= cut_pos.numel():
        order = torch.argsort(cut_pos[:n_cut])                     # arrival order -> candidate order
        out.survivors = (cut_pos[:n_cut][order], cut_val[:n_cut][order])
    elif cut is not None and not total:
        out.survivors = (torch.zeros(0, dtype=torch.int64, device=dev), torch.zeros(0, dtype=torch.float32, device=dev))
    return out


# ------------------------------------------------------------------ threshold scan of the whole candidate set
def filter_scan_max_nodes() -> int:
    return int(_lib.load().eps_filter_scan_max_nodes())


def reverse_positions(rowptr: torch.Tensor, col: torch.Tensor, with_stats: bool = False):
    """int32[nnz]: for entry e of row v with w = col[e], the number of entries of row w below v (per-graph table).
    ``with_stats``: -> (revpos, half_paths int64[N] = its row sums, asymmetric int32[1] device flag) from the same pass."""
    dev = _need_gpu(rowptr, col)
    _chk(rowptr, torch.int64, "rowptr"); _chk(col, torch.int32, "col")
    n = rowptr.numel() - 1
    out = torch.empty(col.numel(), dtype=torch.int32, device=dev)
    hp = torch.zeros(n, dtype=torch.int64, device=dev) if with_stats else None
    flag = torch.zeros(1, dtype=torch.int32, device=dev) if with_stats else None
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_reverse_positions(_ptr(rowptr), _ptr(col), n, _ptr(out), _ptr(hp), _ptr(flag), _stream(dev)),
                   "eps_reverse_positions")
    return (out, hp, flag) if with_stats else out


REVPOS_SORT_MIN = 1 << 62      # stored entries from which the reverse positions come out of a sort instead of searches: never by default --
                               # measured 3.56 vs 3.37 ms on the ppa-like graph (tools/r06_revpos_time.py, profiles/r06/revpos_sorted.txt)


def reverse_positions_symmetric(rowptr: torch.Tensor, col: torch.Tensor):
    """(revpos, half_paths, asymmetric flag) as ``reverse_positions(with_stats=True)`` for a SYMMETRIC pattern, with one search
    per unordered stored pair (eps_reverse_positions_symmetric); the flag comes back 1 on any other pattern and revpos is then
    not usable."""
    dev = _need_gpu(rowptr, col)
    _chk(rowptr, torch.int64, "rowptr"); _chk(col, torch.int32, "col")
    n = rowptr.numel() - 1
    out = torch.empty(col.numel(), dtype=torch.int32, device=dev)
    hp = torch.empty(n, dtype=torch.int64, device=dev)
    # one small buffer for everything the caller reads back: [asymmetric flag (low word), max degree, max half paths, their sum]
    # (zeros: the library clears the flag as the 32-bit word it is -- the high half of info[0] is nobody's)
    info = torch.zeros(4, dtype=torch.int64, device=dev)
    if col.numel() >= REVPOS_SORT_MIN:
        # (r06: no search at all -- the mirror entries come out of a stable sort of the entry indices by column id)
        lib = _lib.load()
        with torch.cuda.device(dev):
            _keep, wsp, wsb = _aligned_ws(dev, lib.eps_reverse_positions_sorted_workspace_bytes(n, col.numel()))
            _lib.check(lib.eps_reverse_positions_sorted(_ptr(rowptr), _ptr(col), n, col.numel(), max(1, int(n - 1).bit_length()), _ptr(out),
                                                        _ptr(hp), info.data_ptr(), info.data_ptr() + 8, wsp, wsb, _stream(dev)),
                       "eps_reverse_positions_sorted")
        return out, hp, info
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_reverse_positions_symmetric(_ptr(rowptr), _ptr(col), n, col.numel(), _ptr(out), _ptr(hp),
                                                               info.data_ptr(), info.data_ptr() + 8, _stream(dev)),
                   "eps_reverse_positions_symmetric")
    return out, hp, info


def node_order(rowptr: Optional[torch.Tensor] = None, keys: Optional[torch.Tensor] = None, relabel: bool = False):
    """Nodes by descending key (the degrees when ``rowptr`` is given, else ``keys`` int64 >= 0), ties by ascending id
    (eps_node_order: a stable radix sort of the library).  -> order int32[n]; with ``relabel`` -> (perm int64[n], inv int32[n],
    new_rowptr int64[n + 1]): what ``relabel_graph`` needs for the copy under that order."""
    src = rowptr if rowptr is not None else keys
    dev = _need_gpu(src)
    _chk(src, torch.int64, "rowptr / keys")
    n = src.numel() - 1 if rowptr is not None else src.numel()
    lib = _lib.load()
    order = perm = inv = new_rp = None
    if relabel:
        perm = torch.empty(n, dtype=torch.int64, device=dev)
        inv = torch.empty(n, dtype=torch.int32, device=dev)
        new_rp = torch.empty(n + 1, dtype=torch.int64, device=dev)
    else:
        order = torch.empty(n, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _, wsp, wsb = _aligned_ws(dev, lib.eps_node_order_workspace_bytes(n))
        _lib.check(lib.eps_node_order(_ptr(rowptr), _ptr(keys) if rowptr is None else None, n, _ptr(order), _ptr(perm), _ptr(inv),
                                      _ptr(new_rp), wsp, wsb, _stream(dev)), "eps_node_order")
    return (perm, inv, new_rp) if relabel else order


def relabel_graph(rowptr, col, val, perm: torch.Tensor, inv32: torch.Tensor, new_rowptr: torch.Tensor):
    """(col, val) of the copy of a coalesced CSR graph under a node permutation: row i = row perm[i] with ids through inv32,
    sorted inside the row (eps_relabel_graph: gather + segmented radix sort)."""
    dev = _need_gpu(rowptr, col, val, perm, inv32, new_rowptr)
    _chk(rowptr, torch.int64, "rowptr"); _chk(col, torch.int32, "col"); _chk(val, torch.float32, "val")
    _chk(perm, torch.int64, "perm"); _chk(inv32, torch.int32, "inv"); _chk(new_rowptr, torch.int64, "new_rowptr")
    n, nnz = rowptr.numel() - 1, col.numel()
    out_c = torch.empty(nnz, dtype=torch.int32, device=dev)
    out_v = None if val is None else torch.empty(nnz, dtype=torch.float32, device=dev)
    if nnz:
        lib = _lib.load()
        with torch.cuda.device(dev):
            _, wsp, wsb = _aligned_ws(dev, lib.eps_relabel_graph_workspace_bytes(n, nnz, int(val is not None)))
            _lib.check(lib.eps_relabel_graph(_ptr(rowptr), _ptr(col), _ptr(val), _ptr(perm), _ptr(inv32), _ptr(new_rowptr), n, nnz,
                                             max(1, int(n - 1).bit_length()), _ptr(out_c), _ptr(out_v), wsp, wsb, _stream(dev)),
                       "eps_relabel_graph")
    return out_c, out_v


def score_bound(rowptr, col, val, node_w, n_rows: int, n_cols: int) -> torch.Tensor:
    """1-element float64 DEVICE tensor: max over the rows of sum |A[v,w]| |node_w[w]| max_u |A[u,w]| (eps_score_bound)."""
    dev = _need_gpu(rowptr, col, val, node_w)
    _chk(rowptr, torch.int64, "rowptr"); _chk(col, torch.int32, "col"); _chk(val, torch.float32, "val"); _chk(node_w, torch.float32, "node_w")
    out = torch.empty(1, dtype=torch.float64, device=dev)
    ws = torch.empty(n_cols, dtype=torch.int32, device=dev) if val is not None else None
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_score_bound(_ptr(rowptr), _ptr(col), _ptr(val), _ptr(node_w), int(n_rows), int(n_cols), col.numel(),
                                               _ptr(out), _ptr(ws), _stream(dev)), "eps_score_bound")
    return out


def filter_scan_windows(n_nodes: int):
    """(ids per window, number of windows) eps_filter_scan uses for an id space of ``n_nodes``."""
    w, k = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.check(_lib.load().eps_filter_scan_windows(int(n_nodes), ctypes.byref(w), ctypes.byref(k)), "eps_filter_scan_windows")
    return int(w.value), int(k.value)


def row_window_splits(rowptr: torch.Tensor, col: torch.Tensor, win_ids: int, n_win: int) -> Optional[torch.Tensor]:
    """int32[(n_win - 1) * N]: entries of every row below each window boundary (None for a single window)."""
    if n_win <= 1:
        return None
    dev = _need_gpu(rowptr, col)
    n = rowptr.numel() - 1
    out = torch.empty((n_win - 1) * n, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_row_window_splits(_ptr(rowptr), _ptr(col), n, int(win_ids), int(n_win), _ptr(out),
                                                     _stream(dev)), "eps_row_window_splits")
    return out


def fixed_weights(node_w: torch.Tensor) -> torch.Tensor:
    """int64[N]: round(node_w * 2**40), the per-node weights in the scan kernel's fixed point."""
    dev = _need_gpu(node_w)
    _chk(node_w, torch.float32, "node_w")
    out = torch.empty(node_w.numel(), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_fixed_weights(_ptr(node_w), node_w.numel(), _ptr(out), _stream(dev)), "eps_fixed_weights")
    return out


_SCAN_WS = {}
KERNEL_EVENTS = None      # a list while bench.py times kernels: (kernel name, start event, end event, work size) per launch
EVENT_NAMES = None        # None: every library call is bracketed while KERNEL_EVENTS is a list; a collection of names: only those (the
                          # scan launches always are) -- bench.py's timed region carries the dominant kernel's events alone


class _timed:
    """``with _timed(dev, name, size):`` -- HIP events around a library call on its stream while bench.py collects KERNEL_EVENTS."""

    def __init__(self, dev, name, size):
        self.dev, self.name, self.size, self.ev = dev, name, int(size), None

    def __enter__(self):
        if KERNEL_EVENTS is not None and (EVENT_NAMES is None or self.name in EVENT_NAMES):
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.ev[0].record(torch.cuda.current_stream(self.dev))
        return self

    def __exit__(self, *exc):
        if self.ev is not None and exc[0] is None:
            self.ev[1].record(torch.cuda.current_stream(self.dev))
            KERNEL_EVENTS.append((self.name, self.ev[0], self.ev[1], self.size))
        return False


def _scan_scratch(dev, max_degree: int) -> torch.Tensor:
    """Scratch of eps_filter_scan (1 GiB of bucket records on 256 CUs + max_degree weights per workgroup): one grow-only
    buffer per device, stream-ordered."""
    key = (dev.type, dev.index)
    need = (int(_lib.load().eps_filter_scan_workspace_bytes(int(max_degree))) + 7) // 8
    if key not in _SCAN_WS or _SCAN_WS[key].numel() < need:
        _SCAN_WS.pop(key, None)
        _SCAN_WS[key] = torch.empty(need, dtype=torch.int64, device=dev)
    return _SCAN_WS[key]


SURVIVOR_SLOTS_MAX = (1 << 32) - (1 << 20)   # slots are 32-bit positions handed out in chunks: keep a chunk's worth of head-room


_PINNED = {"buf": None, "i": 0}


def _pinned_words(words) -> torch.Tensor:
    """A pinned int64 host tensor holding ``words`` (<= 8), from a ring of 256 slots -- the source of an asynchronous host-to-device
    copy must stay untouched until the stream has run it; 256 copies ahead of the device do not happen in this library."""
    if _PINNED["buf"] is None:
        _PINNED["buf"] = torch.empty((256, 8), dtype=torch.int64).pin_memory()
    _PINNED["i"] = (_PINNED["i"] + 1) % 256
    slot = _PINNED["buf"][_PINNED["i"]]
    n = len(words)
    slot[:n] = torch.tensor(words, dtype=torch.int64)
    return slot[:n]


class Survivors:
    """Device-resident eps_survivors record + its key / val arrays.  ``threshold`` may be a Python float or a 0-dim /
    1-element float32 DEVICE tensor (copied on the stream: no host round trip)."""

    def __init__(self, capacity: int, threshold, device, scores_only: bool = False, both: bool = False, prefill: bool = True):
        """``scores_only``: the caller will read the scores alone (the bar estimate): untouched slots are then recognisable
        in ``val`` (-inf) instead of in ``key`` (-1), so no compaction pass is needed before a k-th-largest query.
        ``both``: both fills (key -1 AND val -inf): the list can go through a k-th-largest query as it is and be compacted
        afterwards (scan_topk's selection).
        ``prefill=False`` (eps_scan_screen only): no fill at all -- that kernel marks the unused slots of its reservations
        (key -1, val -inf) itself, and the list's readers stop at the slot counter (``count_ptr``): a step saves two passes
        over a list that is sized for the worst case."""
        import struct
        self.capacity = int(capacity)
        if not 0 < self.capacity <= SURVIVOR_SLOTS_MAX:
            raise _lib.EpsError(f"Survivors: capacity {capacity} outside (0, {SURVIVOR_SLOTS_MAX}]")
        self.scores_only = bool(scores_only) and not both
        self.prefilled = bool(prefill)
        if not prefill:
            self.key = torch.empty(self.capacity, dtype=torch.int64, device=device)
            self.val = torch.empty(self.capacity, dtype=torch.float32, device=device)
        elif both:
            self.key = torch.full((self.capacity,), -1, dtype=torch.int64, device=device)
            self.val = torch.full((self.capacity,), float("-inf"), dtype=torch.float32, device=device)
        elif scores_only:
            self.key = torch.empty(self.capacity, dtype=torch.int64, device=device)
            self.val = torch.full((self.capacity,), float("-inf"), dtype=torch.float32, device=device)
        else:
            self.key = torch.full((self.capacity,), -1, dtype=torch.int64, device=device)
            self.val = torch.empty(self.capacity, dtype=torch.float32, device=device)
        thr_host = float(threshold) if not isinstance(threshold, torch.Tensor) else 0.0
        head = struct.unpack("<q", struct.pack("<fI", thr_host, self.capacity))[0]
        # (the 40-byte record goes up through a pinned staging slot, asynchronously on the stream: torch.tensor(..., device=) is a
        #  blocking copy, and a filter step builds three of these between its launches)
        self.rec = torch.empty(5, dtype=torch.int64, device=device)
        self.rec.copy_(_pinned_words([head, 0, self.key.data_ptr(), self.val.data_ptr(), 0]), non_blocking=True)
        if isinstance(threshold, torch.Tensor):
            self.rec.view(torch.float32)[0:1].copy_(threshold.reshape(1).to(torch.float32))

    @property
    def count_ptr(self) -> int:
        """Device address of the slot counter (uint64): how many slots the launches have handed out so far."""
        return self.rec.data_ptr() + 8

    def counts(self):
        """(slots handed out, unordered candidates scored) -- one device read-back."""
        c = self.rec[[1, 4]].tolist()
        return int(c[0]), int(c[1])                          # (the slot counter is 64-bit: it cannot wrap under the capacity)

    def scores(self, slots: int) -> torch.Tensor:
        """The first ``slots`` score slots as they are: survivors' scores, -inf in untouched slots (``scores_only``)."""
        assert self.scores_only
        return self.val[:min(int(slots), self.capacity)]

    def valid(self, slots: int):
        """(keys, scores) of the survivors among the first ``slots`` slots (unordered): eps_compact_survivors + one host
        read of the count."""
        assert not self.scores_only
        n = min(int(slots), self.capacity)
        dev = self.key.device
        out_k = torch.empty(n, dtype=torch.int64, device=dev)
        out_v = torch.empty(n, dtype=torch.float32, device=dev)
        n_out = torch.zeros(1, dtype=torch.int64, device=dev)
        lib = _lib.load()
        with torch.cuda.device(dev):
            _, wsp, wsb = _aligned_ws(dev, lib.eps_select_topk_cut_workspace_bytes())
            _lib.check(lib.eps_compact_survivors(_ptr(self.key), _ptr(self.val), n, _ptr(out_k), _ptr(out_v), _ptr(n_out), wsp, wsb,
                                                 _stream(dev)), "eps_compact_survivors")
        m = int(n_out.item())
        return out_k[:m], out_v[:m]


def filter_scan(rowptr, col, revpos, fixw, n_nodes: int, columns: torch.Tensor, out: Survivors, max_degree: int,
                splits: Optional[torch.Tensor] = None) -> None:
    """Launch eps_filter_scan over ``columns`` (int32 ids, hand-out order); survivors accumulate in ``out``.
    ``max_degree``: the longest row of the graph (sizes a scratch table); ``splits``: ``row_window_splits`` of the graph
    when ``filter_scan_windows`` reports more than one id window."""
    dev = _need_gpu(rowptr, col, revpos, fixw, columns, splits)
    _chk(rowptr, torch.int64, "rowptr"); _chk(col, torch.int32, "col"); _chk(revpos, torch.int32, "revpos")
    _chk(fixw, torch.int64, "fixw"); _chk(columns, torch.int32, "columns"); _chk(splits, torch.int32, "splits")
    if revpos.numel() != col.numel() or fixw.numel() != n_nodes:
        raise _lib.EpsError("filter_scan: revpos / fixw do not match the graph")
    ws = _scan_scratch(dev, max_degree)
    with torch.cuda.device(dev):
        ev = None
        if KERNEL_EVENTS is not None:              # bench.py: HIP events around the launch, on the stream it runs on
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(torch.cuda.current_stream(dev))
        _lib.check(_lib.load().eps_filter_scan(_ptr(rowptr), _ptr(col), _ptr(revpos), _ptr(fixw), _ptr(splits), n_nodes,
                                               col.numel(), int(max_degree), _ptr(columns), columns.numel(), _ptr(out.rec), _ptr(ws),
                                               ws.numel() * 8, _stream(dev)), "eps_filter_scan")
        if ev is not None:
            ev[1].record(torch.cuda.current_stream(dev))
            KERNEL_EVENTS.append(("filter_scan_kernel", ev[0], ev[1], int(columns.numel())))


# ------------------------------------------------------------------ one-pass threshold scan (csrc/scan_pieces.hip)
def scan_windows() -> int:
    return int(_lib.load().eps_scan_windows())


def scan_cuts(rowptr: torch.Tensor, col: torch.Tensor, bounds: torch.Tensor) -> torch.Tensor:
    """uint16 table [N, M] (stored as int16 bits): entries of every row below each id-window boundary (per-graph table)."""
    dev = _need_gpu(rowptr, col, bounds)
    _chk(rowptr, torch.int64, "rowptr"); _chk(col, torch.int32, "col"); _chk(bounds, torch.int32, "bounds")
    m = scan_windows()
    if bounds.numel() != m + 1:
        raise _lib.EpsError(f"scan_cuts: bounds must hold {m + 1} boundaries")
    n = rowptr.numel() - 1
    out = torch.empty((n, m), dtype=torch.int16, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_scan_cuts(_ptr(rowptr), _ptr(col), n, _ptr(bounds), _ptr(out), _stream(dev)), "eps_scan_cuts")
    return out


def scan_window_paths(rowptr, col, revpos, cuts, heads: Optional[torch.Tensor] = None, columns: Optional[torch.Tensor] = None) -> torch.Tensor:
    """uint32 table [N, M] (int32 bits): two-hop half paths of every column per id window (per-graph table of the scan).
    ``heads`` (``scan_heads``): the paths of the rows a column still walks.
    ``columns`` (int32 ids; without heads): only these rows are computed, the rest of the table stays uninitialised
    (eps_scan_window_paths_columns: the bar sample of a graph whose whole-graph table has not been needed yet)."""
    dev = _need_gpu(rowptr, col, revpos, cuts, heads, columns)
    _chk(rowptr, torch.int64, "rowptr"); _chk(col, torch.int32, "col"); _chk(revpos, torch.int32, "revpos"); _chk(cuts, torch.int16, "cuts")
    n = rowptr.numel() - 1
    _chk_heads(heads, n)
    out = torch.empty((n, scan_windows()), dtype=torch.int32, device=dev)
    if columns is not None:
        _chk(columns, torch.int32, "columns")
        if heads is not None:
            raise _lib.EpsError("scan_window_paths: a column subset comes without a head table")
        with torch.cuda.device(dev):
            _lib.check(_lib.load().eps_scan_window_paths_columns(_ptr(rowptr), _ptr(col), _ptr(revpos), _ptr(cuts), n, _ptr(columns),
                                                                 columns.numel(), _ptr(out), _stream(dev)), "eps_scan_window_paths_columns")
        return out
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_scan_window_paths(_ptr(rowptr), _ptr(col), _ptr(revpos), _ptr(cuts), n, _ptr(heads), _ptr(out),
                                                     _stream(dev)), "eps_scan_window_paths")
    return out


def _chk_heads(heads: Optional[torch.Tensor], n_nodes: int) -> None:
    _chk(heads, torch.int32, "heads")
    if heads is not None and tuple(heads.shape) != (n_nodes, 2):
        raise _lib.EpsError("heads: expected the int32 [N, 2] table of scan_heads")


HUB_MAX = 16384           # hub rows a graph's bitmap table holds (eps_scan_hub_rows; the library takes up to 65536).  Main launch on the
                          # ppa-like graph with 4096 / 8192 / 16384 / 32768 rows: 11.27 / 10.40 / 10.00 / 9.87 ms (72 KB per row)


def scan_row_records(cuts: torch.Tensor, rowptr: torch.Tensor, fx32: torch.Tensor) -> torch.Tensor:
    """int32-bits [N, 32]: per node ONE 128-byte line -- its 32 cuts, its first entry, its screening weight (eps_scan_row_records):
    what eps_scan_screen gathers per walked row, out of one table instead of three.  Per (graph, weight table)."""
    dev = _need_gpu(cuts, rowptr, fx32)
    _chk(cuts, torch.int16, "cuts"); _chk(rowptr, torch.int64, "rowptr"); _chk(fx32, torch.int32, "fx32")
    n = fx32.numel()
    if cuts.shape[0] != n or rowptr.numel() != n + 1:
        raise _lib.EpsError("scan_row_records: cuts / rowptr / fx32 do not match")
    buf = torch.empty(n * 32 + 32, dtype=torch.int32, device=dev)           # (128-byte aligned start inside the allocation)
    off = (-buf.data_ptr() % 128) // 4
    out = buf[off:off + n * 32].view(n, 32)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_scan_row_records(_ptr(cuts), _ptr(rowptr), _ptr(fx32), n, _ptr(out), _stream(dev)), "eps_scan_row_records")
    return out


def scan_column_pack(rowptr, col, revpos, rowrec: torch.Tensor, plan) -> torch.Tensor:
    """int32 [nnz, 8]: the per-column pack of eps_scan_screen's main launch (eps_scan_column_pack): per stored entry, in CSR order,
    what a column's set-up gathers for that neighbour -- id, first entry, weight, reverse position and the cuts of the column's
    first nine pieces in the neighbour's row.  Built from ``plan`` = (pptr, records) and the row records ``rowrec``."""
    pptr, recs = plan
    dev = _need_gpu(rowptr, col, revpos, rowrec, pptr, recs)
    _chk(rowptr, torch.int64, "rowptr"); _chk(col, torch.int32, "col"); _chk(revpos, torch.int32, "revpos")
    _chk(rowrec, torch.int32, "rowrec"); _chk(pptr, torch.int32, "pptr"); _chk(recs, torch.int32, "plan")
    n_nodes = rowptr.numel() - 1
    if pptr.numel() != n_nodes + 1 or revpos.numel() != col.numel() or rowrec.numel() != n_nodes * 32:
        raise _lib.EpsError("scan_column_pack: the tables do not match the graph")
    pack = torch.empty((max(col.numel(), 1), 8), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_scan_column_pack(_ptr(rowptr), _ptr(col), _ptr(revpos), _ptr(rowrec), _ptr(pptr), _ptr(recs), n_nodes,
                                                    _ptr(pack), _stream(dev)), "eps_scan_column_pack")
    return pack


def scan_heads(rowptr, col, fx32: torch.Tensor, n_hub: int, budget: int, max_rows: int = 65535) -> torch.Tensor:
    """int32-bits [N, 2] (x_v, T_v): per column the longest prefix of its row with ids < ``n_hub`` whose screening weights sum
    to T_v <= ``budget`` (table units) -- the rows eps_scan_screen does not walk under a bar (eps_scan_heads)."""
    dev = _need_gpu(rowptr, col, fx32)
    _chk(rowptr, torch.int64, "rowptr"); _chk(col, torch.int32, "col"); _chk(fx32, torch.int32, "fx32")
    n = rowptr.numel() - 1
    if fx32.numel() != n or not 0 <= int(budget) < 1 << 31:
        raise _lib.EpsError("scan_heads: fx32 does not match the graph, or budget outside [0, 2^31)")
    out = torch.empty((n, 2), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_scan_heads(_ptr(rowptr), _ptr(col), _ptr(fx32), n, int(n_hub), int(budget), int(max_rows), _ptr(out),
                                              _stream(dev)), "eps_scan_heads")
    return out


def scan_hub_row_words(n_nodes: int) -> int:
    return int(_lib.load().eps_scan_hub_row_words(int(n_nodes)))


def scan_hub_rows(rowptr, col, n_hub: int) -> torch.Tensor:
    """int32-bits [n_hub, words]: bit x of row w = "x is a neighbour of hub w" -- the adjacency rows of the first ``n_hub`` ids as
    bitmaps over the id space (eps_scan_hub_rows; per-graph table)."""
    dev = _need_gpu(rowptr, col)
    _chk(rowptr, torch.int64, "rowptr"); _chk(col, torch.int32, "col")
    n = rowptr.numel() - 1
    out = torch.empty((max(int(n_hub), 1), scan_hub_row_words(n)), dtype=torch.int32, device=dev)[:int(n_hub)]
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_scan_hub_rows(_ptr(rowptr), _ptr(col), n, int(n_hub), _ptr(out), _stream(dev)), "eps_scan_hub_rows")
    return out


def scan_refine(walked: "Survivors", heads, hubrows, fx32, rowptr, col, n_nodes: int, shift: int, out: "Survivors") -> None:
    """Complete the walked sums of a launch with skipped heads (eps_scan_refine): every valid slot of ``walked`` gets its pair's
    exact head term added; sums at or above ``out``'s bar are appended to ``out`` (compact, scores in 2^-shift units x 2^-shift)."""
    dev = _need_gpu(heads, hubrows, fx32, rowptr, col)
    _chk_heads(heads, n_nodes); _chk(hubrows, torch.int32, "hubrows"); _chk(fx32, torch.int32, "fx32")
    _chk(rowptr, torch.int64, "rowptr"); _chk(col, torch.int32, "col")
    if hubrows.dim() != 2 or hubrows.shape[1] != scan_hub_row_words(n_nodes) or fx32.numel() != n_nodes:
        raise _lib.EpsError("scan_refine: hubrows / fx32 do not match the graph")
    with torch.cuda.device(dev), _timed(dev, "scan_refine", walked.capacity):
        _lib.check(_lib.load().eps_scan_refine(_ptr(walked.rec), _ptr(heads), _ptr(hubrows), hubrows.shape[0], _ptr(fx32), _ptr(rowptr),
                                               _ptr(col), n_nodes, int(shift), _ptr(out.rec), _stream(dev)), "eps_scan_refine")


def scan_screen_weights(fixw: torch.Tensor, shift: int):
    """(fx32 int32-bits[N], bad int32[1]): the scan's fixed-point weights rounded UP to 2^-shift (at least 1)."""
    dev = _need_gpu(fixw)
    _chk(fixw, torch.int64, "fixw")
    out = torch.empty(fixw.numel(), dtype=torch.int32, device=dev)
    bad = torch.empty(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_scan_screen_weights(_ptr(fixw), fixw.numel(), int(shift), _ptr(out), _ptr(bad), _stream(dev)),
                   "eps_scan_screen_weights")
    return out, bad


def rescore_runs(rowptr, col, fixw: torch.Tensor, n_nodes: int, keys_by_u: torch.Tensor) -> torch.Tensor:
    """float32 exact scores of the pairs ``keys_by_u`` = (u << 32) | v, sorted ascending (eps_rescore_runs; unit values)."""
    dev = _need_gpu(rowptr, col, fixw, keys_by_u)
    _chk(rowptr, torch.int64, "rowptr"); _chk(col, torch.int32, "col"); _chk(fixw, torch.int64, "fixw"); _chk(keys_by_u, torch.int64, "keys")
    out = torch.empty(keys_by_u.numel(), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), _timed(dev, "rescore_runs", keys_by_u.numel()):
        _lib.check(_lib.load().eps_rescore_runs(_ptr(rowptr), _ptr(col), _ptr(fixw), n_nodes, _ptr(keys_by_u), keys_by_u.numel(),
                                                _ptr(out), _stream(dev)), "eps_rescore_runs")
    return out


def rescore_weighted(rowptr, col, val, node_w: torch.Tensor, n_nodes: int, keys: torch.Tensor) -> torch.Tensor:
    """float32 exact scores of the pairs ``keys`` = (a << 32) | b on an adjacency with stored values (eps_rescore_weighted)."""
    dev = _need_gpu(rowptr, col, val, node_w, keys)
    _chk(rowptr, torch.int64, "rowptr"); _chk(col, torch.int32, "col"); _chk(val, torch.float32, "val")
    _chk(node_w, torch.float32, "node_w"); _chk(keys, torch.int64, "keys")
    out = torch.empty(keys.numel(), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_rescore_weighted(_ptr(rowptr), _ptr(col), _ptr(val), _ptr(node_w), n_nodes, _ptr(keys), keys.numel(),
                                                    _ptr(out), _stream(dev)), "eps_rescore_weighted")
    return out


def scan_plan_rewalk(plan, variant: int):
    """(paths walked again in hash-partitioned passes, all paths) of a plan table (eps_scan_plan_rewalk) -- two Python ints."""
    pptr, recs = plan
    dev = _need_gpu(pptr, recs)
    out = torch.empty(2, dtype=torch.int64, device=dev)
    n_rec = int(pptr[-1].item())
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_scan_plan_rewalk(_ptr(recs), n_rec, int(variant), _ptr(out), _stream(dev)), "eps_scan_plan_rewalk")
    re, total = out.tolist()
    return re, total


def scan_bounds(rowptr, n_nodes: int) -> torch.Tensor:
    """int32 [M + 1]: the id windows of equal stored-entry mass eps_scan_cuts / eps_scan_screen work on (eps_scan_bounds)."""
    dev = _need_gpu(rowptr)
    _chk(rowptr, torch.int64, "rowptr")
    out = torch.empty(scan_windows() + 1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_scan_bounds(_ptr(rowptr), n_nodes, _ptr(out), _stream(dev)), "eps_scan_bounds")
    return out


def scan_row_sums(rowptr, col, fx32: torch.Tensor, bounds: torch.Tensor, n_nodes: int):
    """(ssum int32-bits [N], smax int32-bits [M + 1], min_fx int32-bits [1]): every row's sum of screening weights (clamped to
    2^31 - 1), its suffix maxima at the window boundaries, and the smallest screening weight of a node with two neighbours or
    more (-1 = none) -- eps_scan_row_sums."""
    dev = _need_gpu(rowptr, col, fx32, bounds)
    _chk(rowptr, torch.int64, "rowptr"); _chk(col, torch.int32, "col"); _chk(fx32, torch.int32, "fx32"); _chk(bounds, torch.int32, "bounds")
    m = scan_windows()
    buf = torch.empty(n_nodes + 2 * m + 2, dtype=torch.int32, device=dev)       # ssum | smax | min_fx | workspace
    ssum, smax, min_fx, ws = buf[:n_nodes], buf[n_nodes:n_nodes + m + 1], buf[n_nodes + m + 1:n_nodes + m + 2], buf[n_nodes + m + 2:]
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_scan_row_sums(_ptr(rowptr), _ptr(col), _ptr(fx32), _ptr(bounds), n_nodes, _ptr(ssum), _ptr(smax),
                                                 _ptr(min_fx), _ptr(ws), _stream(dev)), "eps_scan_row_sums")
    return ssum, smax, min_fx


SCAN_WIDE = 1 << 24        # eps_scan_plan / eps_scan_screen `variant` word, bit 24: packed pieces of single-round columns hold 2^(bits + 1) paths (sketch launches)
SCAN_SKETCH = 1 << 16      # eps_scan_screen's `variant` word, bit 16: packed pieces of single-round columns run as sketch pieces (r06)


def scan_variant_word(variant: int, dmax: Optional[int] = None) -> int:
    """The ``variant`` argument of eps_scan_plan / eps_scan_screen: geometry in the low byte, (dmax + 1) << 8 above it when the
    caller limits the low weight bits a packed / 16-bit direct piece may drop (include/eps_abi.h)."""
    return int(variant) | ((min(int(dmax), 254) + 1) << 8 if dmax is not None else 0)


def scan_plan(rowptr, cuts, wpaths, ssum, smax, bounds, n_nodes: int, shift: int, variant: int, with_d: bool = False,
              heads: Optional[torch.Tensor] = None):
    """(pptr int32-bits [N + 1], records int32 [P, 4]): eps_scan_screen's per-graph plan table (eps_scan_plan, two passes).
    ``with_d``: also the device word that holds the largest number of weight bits a packed / 16-bit direct piece drops.
    ``heads``: the head table ``wpaths`` was built with (a column's sum bound then leaves its skipped head out)."""
    dev = _need_gpu(rowptr, cuts, wpaths, ssum, smax, bounds, heads)
    _chk_heads(heads, n_nodes)
    _chk(rowptr, torch.int64, "rowptr"); _chk(cuts, torch.int16, "cuts"); _chk(wpaths, torch.int32, "wpaths")
    _chk(ssum, torch.int32, "ssum"); _chk(smax, torch.int32, "smax"); _chk(bounds, torch.int32, "bounds")
    lib = _lib.load()
    counts = torch.empty(n_nodes, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.eps_scan_plan(_ptr(rowptr), _ptr(cuts), _ptr(wpaths), _ptr(ssum), _ptr(smax), _ptr(heads), _ptr(bounds), n_nodes,
                                     int(shift), int(variant), _ptr(counts), None, None, None, _stream(dev)), "eps_scan_plan")
        pptr = torch.zeros(n_nodes + 1, dtype=torch.int32, device=dev)
        if n_nodes:
            torch.cumsum(counts, 0, dtype=torch.int32, out=pptr[1:])
        n_rec = int(pptr[-1].item()) if n_nodes else 0
        recs = torch.empty((max(n_rec, 1), 4), dtype=torch.int32, device=dev)
        d_used = torch.zeros(1, dtype=torch.int32, device=dev) if with_d else None
        _lib.check(lib.eps_scan_plan(_ptr(rowptr), _ptr(cuts), _ptr(wpaths), _ptr(ssum), _ptr(smax), _ptr(heads), _ptr(bounds), n_nodes,
                                     int(shift), int(variant), None, _ptr(pptr), _ptr(recs), _ptr(d_used), _stream(dev)), "eps_scan_plan")
    return (pptr, recs, d_used) if with_d else (pptr, recs)


SCAN_VARIANT = 2          # default workgroup / table geometry of eps_scan_screen (include/eps_abi.h); scan.screen_variant picks per graph


def scan_screen(rowptr, col, revpos, fx32, cuts, bounds, n_nodes: int, columns: torch.Tensor, shift: int, out: "Survivors",
                status: torch.Tensor, variant: Optional[int] = None, val: Optional[torch.Tensor] = None,
                node_w: Optional[torch.Tensor] = None, wpaths: Optional[torch.Tensor] = None,
                ssum: Optional[torch.Tensor] = None, smax: Optional[torch.Tensor] = None, plan=None,
                heads: Optional[torch.Tensor] = None, batch_from: Optional[int] = None, rowrec: Optional[torch.Tensor] = None,
                colrec: Optional[torch.Tensor] = None, pack: Optional[torch.Tensor] = None) -> None:
    """Launch eps_scan_screen over ``columns``; survivors (screening scores) accumulate in ``out``.  ``val`` / ``node_w``
    (float32 stored values / node weights): the weighted flavour (eps_scan_screen_weighted; ``fx32`` unused).
    ``ssum`` / ``smax`` (int32-bits [N] / [M + 1]; unit-valued graphs): per-node sums of fx32 over the row and their
    suffix maxima at the window boundaries -- they let pieces keep key and sum in one table word (include/eps_abi.h).
    ``plan`` = (pptr, records) from ``scan_plan`` built with the same wpaths / ssum / smax / shift / variant.
    ``heads`` (``scan_heads``; with the wpaths / plan built for it): the launch skips every column's head and ``out.val`` holds
    the walked sums as raw bits -- ``scan_refine`` turns that list into the one a launch without heads reports.
    ``batch_from``: ``columns[batch_from:]`` are handed out eight per draw (light columns at the end of a heaviest-first list).
    ``rowrec`` (``scan_row_records``): per node one 128-byte line with its cuts, first entry and weight.
    ``colrec`` (int32 [len(columns), 8]; ``scan.column_records``): the columns' headers in hand-out order.
    ``pack`` (``scan_column_pack``; main launch only): the per-column pack built from this plan table and these row records."""
    pptr, recs = plan if plan is not None else (None, None)
    dev = _need_gpu(rowptr, col, revpos, fx32, cuts, bounds, columns, status, val, node_w, wpaths, ssum, smax, pptr, recs, heads)
    _chk_heads(heads, n_nodes)
    if heads is not None and (pptr is None or val is not None):
        raise _lib.EpsError("scan_screen: a head table comes with the plan table built for it (unit-valued graphs)")
    _chk(pptr, torch.int32, "pptr"); _chk(recs, torch.int32, "plan")
    if pptr is not None and (val is not None or wpaths is None or pptr.numel() != n_nodes + 1):
        raise _lib.EpsError("scan_screen: the plan table does not match the graph (unit-valued graphs with wpaths only)")
    _chk(wpaths, torch.int32, "wpaths"); _chk(ssum, torch.int32, "ssum"); _chk(smax, torch.int32, "smax")
    if (ssum is None) != (smax is None) or (ssum is not None and (ssum.numel() != n_nodes or smax.numel() != scan_windows() + 1)):
        raise _lib.EpsError("scan_screen: ssum / smax do not match the graph")
    if wpaths is not None and tuple(wpaths.shape) != (n_nodes, scan_windows()):
        raise _lib.EpsError("scan_screen: wpaths does not match the graph")
    _chk(rowptr, torch.int64, "rowptr"); _chk(col, torch.int32, "col"); _chk(revpos, torch.int32, "revpos")
    _chk(fx32, torch.int32, "fx32"); _chk(cuts, torch.int16, "cuts"); _chk(bounds, torch.int32, "bounds")
    _chk(columns, torch.int32, "columns"); _chk(status, torch.int32, "status"); _chk(val, torch.float32, "val")
    _chk(node_w, torch.float32, "node_w")
    if revpos.numel() != col.numel() or cuts.shape[0] != n_nodes or (val is None and fx32.numel() != n_nodes):
        raise _lib.EpsError("scan_screen: revpos / fx32 / cuts do not match the graph")
    if val is not None and (val.numel() != col.numel() or node_w is None or node_w.numel() != n_nodes):
        raise _lib.EpsError("scan_screen: val / node_w do not match the graph")
    variant = SCAN_VARIANT if variant is None else int(variant)
    lib = _lib.load()
    with torch.cuda.device(dev):
        ev = None
        if KERNEL_EVENTS is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(torch.cuda.current_stream(dev))
        if val is None:
            _lib.check(lib.eps_scan_screen(_ptr(rowptr), _ptr(col), _ptr(revpos), _ptr(fx32), _ptr(cuts), _ptr(wpaths), _ptr(ssum),
                                           _ptr(smax), _ptr(pptr), _ptr(recs), _ptr(heads), _ptr(rowrec), _ptr(pack), _ptr(bounds), n_nodes,
                                           col.numel(), _ptr(columns), _ptr(colrec), columns.numel(), -1 if batch_from is None else int(batch_from),
                                           int(shift), variant, _ptr(out.rec), _ptr(status), _stream(dev)), "eps_scan_screen")
        else:
            _lib.check(lib.eps_scan_screen_weighted(_ptr(rowptr), _ptr(col), _ptr(val), _ptr(revpos), _ptr(node_w), _ptr(cuts),
                                                    _ptr(wpaths), _ptr(bounds), n_nodes, col.numel(), _ptr(columns), columns.numel(), int(shift),
                                                    variant, _ptr(out.rec), _ptr(status), _stream(dev)), "eps_scan_screen_weighted")
        if ev is not None:
            ev[1].record(torch.cuda.current_stream(dev))
            KERNEL_EVENTS.append(("scan_piece_kernel", ev[0], ev[1], int(columns.numel())))


def spmm_csr(rowptr, col, val, x: torch.Tensor, bias=None, relu=False, mean=False, out=None) -> torch.Tensor:
    dev = _need_gpu(rowptr, col, val, x, bias, out, row_strided=(x, out))
    _chk(rowptr, torch.int64, "rowptr"); _chk(col, torch.int32, "col"); _chk(val, torch.float32, "val")
    _chk(x, torch.float32, "x"); _chk(bias, torch.float32, "bias")
    n_rows = rowptr.numel() - 1
    f = x.shape[1]
    if out is None:
        out = torch.empty((n_rows, f), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_spmm_csr(_ptr(rowptr), _ptr(col), _ptr(val), n_rows, _ptr(x), x.stride(0), f,
                                            _ptr(bias), int(relu), int(mean), _ptr(out), out.stride(0),
                                            _stream(dev)), "eps_spmm_csr")
    return out


def gcn_norm(rowptr, col, val) -> torch.Tensor:
    dev = _need_gpu(rowptr, col, val)
    n_rows = rowptr.numel() - 1
    dis = torch.empty(n_rows, dtype=torch.float32, device=dev)
    out = torch.empty(col.numel(), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_gcn_norm(_ptr(rowptr), _ptr(col), _ptr(val), n_rows, _ptr(dis), _ptr(out),
                                            _stream(dev)), "eps_gcn_norm")
    return out


def gemm(a: torch.Tensor, b_nk: torch.Tensor, bias=None, relu=False, out=None, accumulate=False, lower_only=False) -> torch.Tensor:
    """C = act(a @ b_nk.T + bias (+ C)); b_nk is [N,K] (torch.nn.Linear layout).  ``lower_only``: the product is symmetric and
    only its 128 x 128 tiles on and below the diagonal are computed (the rest of ``out`` is left as it is)."""
    dev = _need_gpu(a, b_nk, bias, out, row_strided=(a, b_nk, out))
    _chk(a, torch.float32, "a"); _chk(b_nk, torch.float32, "b"); _chk(bias, torch.float32, "bias")
    m, k = a.shape
    n = b_nk.shape[0]
    if b_nk.shape[1] != k:
        raise _lib.EpsError(f"gemm: inner dims differ ({k} vs {b_nk.shape[1]})")
    if out is None:
        if accumulate:
            raise _lib.EpsError("gemm: accumulate needs out")
        out = torch.empty((m, n), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_gemm_f32(_ptr(a), a.stride(0), _ptr(b_nk), b_nk.stride(0), _ptr(bias), int(relu) | (2 if lower_only else 0),
                                            int(accumulate), _ptr(out), out.stride(0), m, n, k, _stream(dev)),
                   "eps_gemm_f32")
    return out


def dense_adjacency(rowptr, col, n_nodes: int, pad_to: int = 128) -> torch.Tensor:
    """float32 [Np, Np] (Np = n_nodes rounded up to ``pad_to``): 1.0 at every stored entry, 0 elsewhere (eps_dense_adjacency)."""
    dev = _need_gpu(rowptr, col)
    _chk(rowptr, torch.int64, "rowptr"); _chk(col, torch.int32, "col")
    np_ = (int(n_nodes) + pad_to - 1) // pad_to * pad_to
    a = torch.empty((max(np_, 1), max(np_, 1)), dtype=torch.float32, device=dev)[:np_, :np_]
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_dense_adjacency(_ptr(rowptr), _ptr(col), int(n_nodes), np_, np_, _ptr(a), _stream(dev)), "eps_dense_adjacency")
    return a


def dense_cn_candidates(rowptr, col, n_nodes: int, directed: bool = False, check_symmetric: bool = False, as_rows: bool = False):
    """(keys int64 (v << 32) | u, counts float32) of the 2-hop non-edges of a DENSE unit-valued symmetric graph, column-major (v
    ascending, then u), with their common-neighbour counts: A A^T on the f32 MFMA + a masked read (csrc/dense_cn.hip).  Default:
    every unordered pair once (u < v; lower tiles of the product only); ``directed``: both orientations -- the reference's
    candidate list (filter.py:96-109) in its own order.  ``check_symmetric``: None comes back when A != A^T (a transposed
    compare of the dense matrix: the check costs no table).  ``as_rows``: (rows float32 [E, 3] = (u, v, count), counts) instead
    of keys -- the proposal file's rows written by the kernel itself.  One host read (the list's length; with the check, one more)."""
    dev = _need_gpu(rowptr, col)
    a = dense_adjacency(rowptr, col, n_nodes)
    if check_symmetric and not bool(torch.equal(a, a.t())):
        return None
    c = torch.empty_like(a)
    gemm(a, a, out=c, lower_only=True)               # (symmetric: half the tiles ...
    lib = _lib.load()
    if directed:                                     #  ... and a transposed copy for the readers of whole rows)
        with torch.cuda.device(dev):
            _lib.check(lib.eps_dense_mirror_lower(_ptr(c), n_nodes, c.stride(0), _stream(dev)), "eps_dense_mirror_lower")
    counts = torch.empty(n_nodes + 1, dtype=torch.int64, device=dev)
    counts[n_nodes:] = 0
    with torch.cuda.device(dev):
        _lib.check(lib.eps_dense_candidates(_ptr(a), _ptr(c), n_nodes, a.stride(0), 0 if directed else 1, _ptr(counts), None, None, None,
                                            None, _stream(dev)), "eps_dense_candidates")
        colptr = torch.zeros(n_nodes + 1, dtype=torch.int64, device=dev)
        torch.cumsum(counts[:n_nodes], 0, out=colptr[1:])
        total = int(colptr[-1].item())
        keys = torch.empty((total, 3), dtype=torch.float32, device=dev) if as_rows else torch.empty(total, dtype=torch.int64, device=dev)
        vals = torch.empty(total, dtype=torch.float32, device=dev)
        if total:
            _lib.check(lib.eps_dense_candidates(_ptr(a), _ptr(c), n_nodes, a.stride(0), 0 if directed else 1, None, _ptr(colptr),
                                                None if as_rows else _ptr(keys), _ptr(vals), _ptr(keys) if as_rows else None,
                                                _stream(dev)), "eps_dense_candidates")
    return keys, vals


def mlp_decode(h: torch.Tensor, u, v, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor],
               apply_sigmoid=True) -> torch.Tensor:
    dev = _need_gpu(h, u, v, *weights, *biases)
    _chk(h, torch.float32, "h"); _chk(u, torch.int32, "u"); _chk(v, torch.int32, "v")
    L = len(weights)
    hd = h.shape[1]
    for i, (w, b) in enumerate(zip(weights, biases)):
        _chk(w, torch.float32, f"w{i}"); _chk(b, torch.float32, f"b{i}")
        exp = (1 if i == L - 1 else hd, hd)
        if tuple(w.shape) != exp:
            raise _lib.EpsError(f"mlp_decode: layer {i} weight {tuple(w.shape)} != {exp} "
                                f"(hidden width must equal the embedding width, last layer out=1)")
    n = u.numel()
    out = torch.empty(n, dtype=torch.float32, device=dev)
    wp = (ctypes.c_void_p * L)(*[w.data_ptr() for w in weights])
    bp = (ctypes.c_void_p * L)(*[b.data_ptr() for b in biases])
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_mlp_decode(_ptr(h), h.shape[0], hd, _ptr(u), _ptr(v), n, wp, bp, L,
                                              int(apply_sigmoid), _ptr(out), _stream(dev)), "eps_mlp_decode")
    return out


def kth_largest(x: torch.Tensor, k: int) -> torch.Tensor:
    """The k-th largest value of a float32 device vector (1-element device tensor; no host round trip): radix select."""
    dev = _need_gpu(x)
    _chk(x, torch.float32, "x")
    lib = _lib.load()
    out = torch.empty(1, dtype=torch.float32, device=dev)
    ws = torch.empty((int(lib.eps_kth_largest_workspace_bytes()) + 7) // 8, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.eps_kth_largest_f32(_ptr(x), x.numel(), int(k), _ptr(out), _ptr(ws), _stream(dev)), "eps_kth_largest_f32")
    return out


def kth_largest_dist(x: torch.Tensor, k: int, world: int = 1) -> torch.Tensor:
    """The k-th largest value of the UNION of every rank's float32 device vector ``x`` (lengths may differ, 0 allowed) as a
    1-element device tensor, identical on all ranks; -inf when the union holds fewer than k values.  Radix select in four
    rounds; per round one all-reduce of the 256-bin histogram (1 KiB) -- no host round trip.  ``world`` == 1: no collective."""
    dev = _need_gpu(x)
    _chk(x, torch.float32, "x")
    lib = _lib.load()
    state = torch.empty(int(lib.eps_kth_largest_workspace_bytes()) // 4, dtype=torch.int32, device=dev)
    out = torch.empty(1, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        st = _stream(dev)
        _lib.check(lib.eps_kth_begin(_ptr(state), int(k), st), "eps_kth_begin")
        for shift in (24, 16, 8, 0):
            _lib.check(lib.eps_kth_hist_f32(_ptr(x), x.numel(), _ptr(state), shift, st), "eps_kth_hist_f32")
            from . import dist as epd
            if world > 1 or epd.FORCE_COLLECTIVES:
                epd.all_reduce_sum_(state[4:260])
            _lib.check(lib.eps_kth_pick(_ptr(state), shift, _ptr(out), st), "eps_kth_pick")
    return out


def select_compact(keys: Optional[torch.Tensor], vals: torch.Tensor, k: int, count_ptr: Optional[int] = None, mode: int = 0,
                   params=(0.0, 0.0, 0.0), compact: bool = True, room: Optional[int] = None):
    """Radix select + threshold + compaction of a list on ONE device in one launch (eps_select_compact; the sharded job-wide
    select is ``kth_largest_dist``).  -> (out_keys, out_vals, n_out, kth, thr): the entries with key >= 0 and score >= thr
    compacted to the front of fresh arrays (None, None, None without ``compact``), their number, the k-th largest score and
    the threshold derived from it -- all DEVICE tensors, no host read.  ``count_ptr``: device address of a uint64 that bounds the
    list (``Survivors.count_ptr``).  ``mode`` / ``params``: 0 thr = kth; 1 the largest float below kth; 2 max(kth - a, kth * b) -
    |kth| * c with params (a, b, c).  ``room``: entries the output arrays hold (default: the list's length) -- n_out may come
    back LARGER: the entries beyond ``room`` were counted, not stored (a list sized for the worst case need not be mirrored by
    outputs of that size; the caller repeats the call with more room in the rare case)."""
    dev = _need_gpu(keys, vals)
    _chk(keys, torch.int64, "keys"); _chk(vals, torch.float32, "vals")
    n = vals.numel()
    lib = _lib.load()
    words = (int(lib.eps_select_compact_workspace_bytes()) + 7) // 8
    # (one state per call, from the caching allocator: stream-ordered -- the block is handed out again only behind this launch on
    #  this stream, whatever other streams or however many calls are pending; r04's ring of 16 per device was neither)
    state = torch.empty(words, dtype=torch.int64, device=dev)
    kth = torch.empty(2, dtype=torch.float32, device=dev)
    out_k = out_v = n_out = None
    if compact:
        if keys is None:
            raise _lib.EpsError("select_compact: the compaction needs keys")
        room = n if room is None else max(1, min(int(room), n))
        out_k = torch.empty(room, dtype=torch.int64, device=dev)
        out_v = torch.empty(room, dtype=torch.float32, device=dev)
        n_out = torch.empty(1, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev), _timed(dev, "select_compact", n):
        _lib.check(lib.eps_select_compact(_ptr(keys), _ptr(vals), n, count_ptr, int(k), int(mode), float(params[0]), float(params[1]),
                                          float(params[2]), kth.data_ptr(), kth.data_ptr() + 4, _ptr(out_k), _ptr(out_v),
                                          0 if room is None else int(room), _ptr(n_out), _ptr(state), _stream(dev)), "eps_select_compact")
    return out_k, out_v, n_out, kth[0:1], kth[1:2]


def compact_at_least(keys: torch.Tensor, vals: torch.Tensor, cut: Optional[torch.Tensor]):
    """(keys, vals, n) -- the survivors (key >= 0) with score >= ``cut`` (1-element float32 DEVICE tensor; None: all of them)
    compacted to the front of fresh arrays, ``n`` a 1-element int64 device tensor (no host read)."""
    dev = _need_gpu(keys, vals, cut)
    _chk(keys, torch.int64, "keys"); _chk(vals, torch.float32, "vals"); _chk(cut, torch.float32, "cut")
    n = keys.numel()
    out_k = torch.empty(n, dtype=torch.int64, device=dev)
    out_v = torch.empty(n, dtype=torch.float32, device=dev)
    n_out = torch.empty(1, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_compact_at_least(_ptr(keys), _ptr(vals), n, _ptr(cut), _ptr(out_k), _ptr(out_v), _ptr(n_out),
                                                    _stream(dev)), "eps_compact_at_least")
    return out_k, out_v, n_out


def compact_between(keys: torch.Tensor, vals: torch.Tensor, lo: Optional[torch.Tensor], hi: Optional[torch.Tensor]):
    """(keys, vals, n) -- the entries (key >= 0) with lo <= score < hi (1-element float32 DEVICE tensors; None: open end)
    compacted to the front of fresh arrays, ``n`` a 1-element int64 device tensor (no host read)."""
    dev = _need_gpu(keys, vals, lo, hi)
    _chk(keys, torch.int64, "keys"); _chk(vals, torch.float32, "vals"); _chk(lo, torch.float32, "lo"); _chk(hi, torch.float32, "hi")
    n = keys.numel()
    out_k = torch.empty(n, dtype=torch.int64, device=dev)
    out_v = torch.empty(n, dtype=torch.float32, device=dev)
    n_out = torch.empty(1, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_compact_between(_ptr(keys), _ptr(vals), n, _ptr(lo), _ptr(hi), _ptr(out_k), _ptr(out_v), _ptr(n_out),
                                                   _stream(dev)), "eps_compact_between")
    return out_k, out_v, n_out


def sort_pairs_by_u(keys: torch.Tensor, id_bits: int = 32, v_block_shift: int = 0) -> torch.Tensor:
    """Survivor keys v << 32 | u (u < v, any order) -> u << 32 | v sorted by (u, v), what ``rescore_runs`` wants: two stable
    radix sorts over the id bits (eps_sort_pairs_by_u)."""
    dev = _need_gpu(keys)
    _chk(keys, torch.int64, "keys")
    n = keys.numel()
    out = torch.empty(n, dtype=torch.int64, device=dev)
    if n:
        lib = _lib.load()
        with torch.cuda.device(dev), _timed(dev, "sort_pairs_by_u", n):
            _, wsp, wsb = _aligned_ws(dev, lib.eps_sort_pairs_by_u_workspace_bytes(n))
            _lib.check(lib.eps_sort_pairs_by_u(_ptr(keys), n, int(id_bits), int(v_block_shift), _ptr(out), wsp, wsb, _stream(dev)),
                       "eps_sort_pairs_by_u")
    return out


def select_rows(sel_keys: torch.Tensor, sel_vals: torch.Tensor, k: int, id_bits: int = 32,
                perm: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """The first min(k, 2 m) DIRECTED rows, in the declared order, of m selected unordered pairs (every pair at or above the
    job-wide cut): mirror + stable radix sorts (eps_select_topk_rows).  No host read: m is the arrays' length.
    ``perm`` (int64 [n_nodes]): the pairs are in the labels of a relabelled graph whose id i is the caller's perm[i]; the rows
    come out -- and are ordered -- in the caller's labels."""
    dev = _need_gpu(sel_keys, sel_vals, perm)
    _chk(sel_keys, torch.int64, "sel_keys"); _chk(sel_vals, torch.float32, "sel_vals"); _chk(perm, torch.int64, "perm")
    m, k = sel_keys.numel(), int(k)
    take = min(k, 2 * m)
    out_k = torch.empty(take, dtype=torch.int64, device=dev)
    out_v = torch.empty(take, dtype=torch.float32, device=dev)
    if take:
        lib = _lib.load()
        with torch.cuda.device(dev):
            _, wsp, wsb = _aligned_ws(dev, lib.eps_select_topk_rows_workspace_bytes(m))
            if perm is None:
                _lib.check(lib.eps_select_topk_rows(_ptr(sel_keys), _ptr(sel_vals), m, k, int(id_bits), _ptr(out_k), _ptr(out_v), wsp, wsb,
                                                    _stream(dev)), "eps_select_topk_rows")
            else:
                _lib.check(lib.eps_select_topk_rows_relabelled(_ptr(sel_keys), _ptr(sel_vals), m, k, int(id_bits), _ptr(perm), _ptr(out_k),
                                                               _ptr(out_v), wsp, wsb, _stream(dev)), "eps_select_topk_rows_relabelled")
    return out_k, out_v


# ---- the tail of the filter step with device-side sizes (csrc/tail_sort.hip, r06) ------------------------------------------------
_TAIL_STATE = {}


def tail_state(dev) -> torch.Tensor:
    """The state block of the tail kernels for (device, current stream): zeroed ONCE here -- the kernels that consume it leave it
    zeroed, so a step issues no memset for it."""
    key = (dev.type, dev.index, torch.cuda.current_stream(dev).cuda_stream)
    st = _TAIL_STATE.get(key)
    if st is None:
        st = _TAIL_STATE[key] = torch.zeros((int(_lib.load().eps_tail_state_bytes()) + 7) // 8 + 32, dtype=torch.int64, device=dev)
    return st


def _tail_state_ptr(dev):
    st = tail_state(dev)
    return ctypes.c_void_p(st.data_ptr() + (-st.data_ptr()) % 256)


def tail_state_reset(dev) -> None:
    """Zero the state again (after a failed call that may have left a histogram or a hand-over counter half way)."""
    tail_state(dev).zero_()


def _dev_count(n_dev):
    """A device count argument: None, a raw device address (Survivors.count_ptr), or a 1-element int64 tensor."""
    if n_dev is None or isinstance(n_dev, int):
        return n_dev
    return n_dev.data_ptr()


def score_hist(keys: Optional[torch.Tensor], vals: torch.Tensor, n_dev, base: torch.Tensor, above: Optional[torch.Tensor] = None) -> None:
    """Add the histogram of the list's live scores (buckets of their distance to ``base``, a 1-element float32 device tensor) to
    the tail state (eps_score_hist).  ``n_dev``: device count that bounds the list (None: its length)."""
    dev = _need_gpu(keys, vals, base, above)
    _chk(keys, torch.int64, "keys"); _chk(vals, torch.float32, "vals"); _chk(base, torch.float32, "base"); _chk(above, torch.float32, "above")
    with torch.cuda.device(dev), _timed(dev, "select_compact", vals.numel()):
        _lib.check(_lib.load().eps_score_hist(_ptr(keys), _ptr(vals), vals.numel(), _dev_count(n_dev), _ptr(base), _ptr(above),
                                              _tail_state_ptr(dev), _stream(dev)), "eps_score_hist")


def score_pick_compact(keys: Optional[torch.Tensor], vals: torch.Tensor, n_dev, base: torch.Tensor, k: int, above: Optional[torch.Tensor] = None,
                       mode: int = 0, params=(0.0, 0.0, 0.0), swap_halves: bool = False, room: Optional[int] = None,
                       want_vals: bool = True):
    """The selection behind ``score_hist`` (eps_score_pick_compact): -> (out_keys, out_vals, n_out, kth, thr), all device tensors.
    ``kth`` is the lower edge of the bucket that holds the k-th best live value (<= the exact k-th, by at most 2^-8 of its distance
    to ``base``), ``thr`` derived from it as in ``select_compact``; the live entries with value >= thr are compacted into arrays
    of ``room`` entries (more are counted in n_out, not stored); ``swap_halves`` exchanges the key halves on the way."""
    dev = _need_gpu(keys, vals, base, above)
    n = vals.numel()
    kth = torch.empty(2, dtype=torch.float32, device=dev)
    out_k = out_v = n_out = None
    if keys is not None:
        room = max(1, n if room is None else min(int(room), max(n, 1)))
        out_k = torch.empty(room, dtype=torch.int64, device=dev)
        out_v = torch.empty(room, dtype=torch.float32, device=dev) if want_vals else None
        n_out = torch.empty(1, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev), _timed(dev, "select_compact", n):
        _lib.check(_lib.load().eps_score_pick_compact(_ptr(keys), _ptr(vals), n, _dev_count(n_dev), _ptr(base), _ptr(above), int(k), int(mode),
                                                      float(params[0]), float(params[1]), float(params[2]), int(bool(swap_halves)),
                                                      kth.data_ptr(), kth.data_ptr() + 4, _ptr(out_k), _ptr(out_v),
                                                      0 if out_k is None else int(room), _ptr(n_out), _tail_state_ptr(dev), _stream(dev)),
                   "eps_score_pick_compact")
    return out_k, out_v, n_out, kth[0:1], kth[1:2]


def score_bins() -> int:
    return int(_lib.load().eps_score_bins())


def score_hist_into(keys: Optional[torch.Tensor], vals: torch.Tensor, n_dev, base: torch.Tensor, hist: torch.Tensor,
                    above: Optional[torch.Tensor] = None) -> None:
    """``score_hist`` into an int32 array of the caller's (``score_bins()`` words, zeroed by the caller): the histogram a rank of a
    sharded step sends to the others (eps_score_hist_into)."""
    dev = _need_gpu(keys, vals, base, above, hist)
    _chk(keys, torch.int64, "keys"); _chk(vals, torch.float32, "vals"); _chk(base, torch.float32, "base"); _chk(above, torch.float32, "above")
    _chk(hist, torch.int32, "hist")
    if hist.numel() < score_bins() or not hist.is_contiguous():
        raise _lib.EpsError("score_hist_into: hist must hold score_bins() contiguous words")
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_score_hist_into(_ptr(keys), _ptr(vals), vals.numel(), _dev_count(n_dev), _ptr(base), _ptr(above),
                                                   _ptr(hist), _stream(dev)), "eps_score_hist_into")


def score_deal_plan(hists: torch.Tensor, k: int, base: torch.Tensor):
    """(cut float32[1], splitters float32[world - 1], counts int64[world, world], nsel int64[world]) from the ranks' gathered score
    histograms ``hists`` (int32 [world, >= score_bins()], row-contiguous): eps_score_deal_plan -- device tensors, no host read."""
    dev = _need_gpu(hists, base, row_strided=(hists,))
    _chk(hists, torch.int32, "hists"); _chk(base, torch.float32, "base")
    if hists.dim() != 2 or hists.stride(1) != 1 or hists.shape[1] < score_bins():
        raise _lib.EpsError("score_deal_plan: hists must be [world, >= score_bins()] with unit column stride")
    world = hists.shape[0]
    cut = torch.empty(1, dtype=torch.float32, device=dev)
    sp = torch.empty(max(world - 1, 1), dtype=torch.float32, device=dev)
    counts = torch.empty((world, world), dtype=torch.int64, device=dev)
    nsel = torch.empty(world, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_score_deal_plan(_ptr(hists), hists.stride(0), world, int(k), _ptr(base), _ptr(cut), _ptr(sp), _ptr(counts),
                                                   _ptr(nsel), _stream(dev)), "eps_score_deal_plan")
    return cut, sp[:world - 1], counts, nsel


def radix_sort_by_u(keys: torch.Tensor, n_dev, id_bits: int = 32, v_block_shift: int = 0) -> torch.Tensor:
    """``sort_pairs_by_u`` in one cooperative launch with the list's length read on the device (eps_radix_sort_by_u): the first
    min(*n_dev, len(keys)) keys v << 32 | u -> u << 32 | v in the order eps_rescore_runs wants; the rest of the output is undefined."""
    dev = _need_gpu(keys)
    _chk(keys, torch.int64, "keys")
    n = keys.numel()
    out = torch.empty(n, dtype=torch.int64, device=dev)
    if n:
        lib = _lib.load()
        with torch.cuda.device(dev), _timed(dev, "sort_pairs_by_u", n):
            _, wsp, wsb = _aligned_ws(dev, lib.eps_radix_sort_workspace_bytes(n))
            _lib.check(lib.eps_radix_sort_by_u(_ptr(keys), n, _dev_count(n_dev), int(id_bits), int(v_block_shift), _ptr(out), wsp, wsb,
                                               _tail_state_ptr(dev), _stream(dev)), "eps_radix_sort_by_u")
    return out


def radix_sort_rows(sel_keys: torch.Tensor, sel_vals: torch.Tensor, m_dev, k: int, id_bits: int = 32, perm: Optional[torch.Tensor] = None):
    """``select_rows`` in one cooperative launch with the number of selected pairs read on the device (eps_radix_sort_rows):
    -> (pairs int64 [2, cap] as (u; v), scores float32 [cap], n_rows 1-element int64 device tensor) with cap = min(k, 2 len);
    the first n_rows = min(k, 2 m) columns are the rows of the declared order, the rest undefined."""
    dev = _need_gpu(sel_keys, sel_vals, perm)
    _chk(sel_keys, torch.int64, "sel_keys"); _chk(sel_vals, torch.float32, "sel_vals"); _chk(perm, torch.int64, "perm")
    m_max, k = sel_keys.numel(), int(k)
    cap = min(k, 2 * m_max)
    pairs = torch.empty((2, cap), dtype=torch.int64, device=dev)
    scores = torch.empty(cap, dtype=torch.float32, device=dev)
    n_rows = torch.zeros(1, dtype=torch.int64, device=dev) if cap == 0 else torch.empty(1, dtype=torch.int64, device=dev)
    if cap:
        lib = _lib.load()
        with torch.cuda.device(dev), _timed(dev, "select_rows", m_max):
            _, wsp, wsb = _aligned_ws(dev, lib.eps_radix_sort_workspace_bytes(2 * m_max))
            _lib.check(lib.eps_radix_sort_rows(_ptr(sel_keys), _ptr(sel_vals), m_max, _dev_count(m_dev), k, int(id_bits), _ptr(perm),
                                               _ptr(pairs), cap, _ptr(scores), _ptr(n_rows), wsp, wsb, _tail_state_ptr(dev), _stream(dev)),
                       "eps_radix_sort_rows")
    return pairs, scores, n_rows


def rescore_runs_dev(rowptr, col, fixw: torch.Tensor, n_nodes: int, keys_by_u: torch.Tensor, n_dev: torch.Tensor) -> torch.Tensor:
    """``rescore_runs`` over the first min(*n_dev, len) keys (eps_rescore_runs_dev); the other outputs are undefined."""
    dev = _need_gpu(rowptr, col, fixw, keys_by_u, n_dev)
    _chk(keys_by_u, torch.int64, "keys_by_u"); _chk(fixw, torch.int64, "fixw"); _chk(n_dev, torch.int64, "n_dev")
    out = torch.empty(keys_by_u.numel(), dtype=torch.float32, device=dev)
    if keys_by_u.numel():
        with torch.cuda.device(dev), _timed(dev, "rescore_runs", keys_by_u.numel()):
            _lib.check(_lib.load().eps_rescore_runs_dev(_ptr(rowptr), _ptr(col), _ptr(fixw), int(n_nodes), _ptr(keys_by_u), keys_by_u.numel(),
                                                        _ptr(n_dev), _ptr(out), _stream(dev)), "eps_rescore_runs_dev")
    return out


def select_rows_pairs(sel_keys: torch.Tensor, sel_vals: torch.Tensor, k: int, id_bits: int = 32,
                      perm: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """``select_rows`` with the rows written as the proposal tensor itself: (pairs int64 [2, min(k, 2 m)] as (u; v), scores)
    (eps_select_topk_rows_pairs) -- no tensor ops over the K rows afterwards."""
    dev = _need_gpu(sel_keys, sel_vals, perm)
    _chk(sel_keys, torch.int64, "sel_keys"); _chk(sel_vals, torch.float32, "sel_vals"); _chk(perm, torch.int64, "perm")
    m, k = sel_keys.numel(), int(k)
    take = min(k, 2 * m)
    pairs = torch.empty((2, take), dtype=torch.int64, device=dev)
    out_v = torch.empty(take, dtype=torch.float32, device=dev)
    if take:
        lib = _lib.load()
        with torch.cuda.device(dev), _timed(dev, "select_rows", m):
            _, wsp, wsb = _aligned_ws(dev, lib.eps_select_topk_rows_workspace_bytes(m))
            _lib.check(lib.eps_select_topk_rows_pairs(_ptr(sel_keys), _ptr(sel_vals), m, k, int(id_bits), _ptr(perm), _ptr(pairs), take,
                                                      _ptr(out_v), wsp, wsb, _stream(dev)), "eps_select_topk_rows_pairs")
    return pairs, out_v


_SELECT_WS = {}


def _aligned_ws(dev, n_bytes: int):
    """(tensor, 256-byte aligned pointer, bytes from there) of a grow-only per-device scratch."""
    need = (int(n_bytes) + 7) // 8 + 32
    key = (dev.type, dev.index)
    if key not in _SELECT_WS or _SELECT_WS[key].numel() < need:
        _SELECT_WS.pop(key, None)
        _SELECT_WS[key] = torch.empty(int(need * 1.25), dtype=torch.int64, device=dev)
    ws = _SELECT_WS[key]
    off = (-ws.data_ptr()) % 256
    return ws, ctypes.c_void_p(ws.data_ptr() + off), ws.numel() * 8 - off


def select_topk(keys: torch.Tensor, vals: torch.Tensor, k: int, id_bits: int = 32) -> Tuple[torch.Tensor, torch.Tensor]:
    """The k best DIRECTED rows (score descending, key ascending) of a list of unordered survivors of ``filter_scan``
    (key = v << 32 | u, u < v; no -1 slots): (keys int64, scores float32), sorted.  eps_select_topk_cut (radix select of the
    cut + compaction) -> one host read of the count -> eps_select_topk_rows (mirror + stable radix sorts).  ``id_bits``:
    every node id is below 2**id_bits (fewer sort passes)."""
    dev = _need_gpu(keys, vals)
    _chk(keys, torch.int64, "keys"); _chk(vals, torch.float32, "vals")
    n, k = keys.numel(), int(k)
    if vals.numel() != n:
        raise _lib.EpsError("select_topk: keys and vals differ in length")
    lib = _lib.load()
    sel_k = torch.empty(n, dtype=torch.int64, device=dev)
    sel_v = torch.empty(n, dtype=torch.float32, device=dev)
    n_sel = torch.zeros(1, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        _, wsp, wsb = _aligned_ws(dev, lib.eps_select_topk_cut_workspace_bytes())
        _lib.check(lib.eps_select_topk_cut(_ptr(keys), _ptr(vals), n, k, _ptr(sel_k), _ptr(sel_v), _ptr(n_sel), wsp, wsb,
                                           _stream(dev)), "eps_select_topk_cut")
        m = int(n_sel.item())
        take = min(k, 2 * m)
        out_k = torch.empty(take, dtype=torch.int64, device=dev)
        out_v = torch.empty(take, dtype=torch.float32, device=dev)
        if take:
            _, wsp, wsb = _aligned_ws(dev, lib.eps_select_topk_rows_workspace_bytes(m))
            _lib.check(lib.eps_select_topk_rows(_ptr(sel_k), _ptr(sel_v), m, k, int(id_bits), _ptr(out_k), _ptr(out_v), wsp, wsb,
                                                _stream(dev)), "eps_select_topk_rows")
    return out_k, out_v


def pack_keys(score: torch.Tensor, ids: Optional[torch.Tensor] = None, id_base: int = 0) -> torch.Tensor:
    dev = _need_gpu(score, ids)
    _chk(score, torch.float32, "score"); _chk(ids, torch.int64, "ids")
    # the key holds the id in its low 32 bits: an id the kernel would truncate aliases another candidate's key
    if ids is not None and ids.numel():
        lo, hi = torch.aminmax(ids)
        if int(lo) < 0 or int(hi) >= 1 << 32:
            raise _lib.EpsError(f"pack_keys: ids must lie in [0, 2**32), got [{int(lo)}, {int(hi)}] "
                                "(use shard-relative ids and merge_ranked_lists for longer candidate lists)")
    elif ids is None and (id_base < 0 or id_base + score.numel() > 1 << 32):
        raise _lib.EpsError("pack_keys: id_base + n exceeds the 32-bit id field of the key")
    keys = torch.empty(score.numel(), dtype=torch.int64, device=dev)  # bit pattern of the uint64 key
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_pack_keys(_ptr(score), _ptr(ids), id_base, score.numel(), _ptr(keys),
                                             _stream(dev)), "eps_pack_keys")
    return keys


def unpack_keys(keys: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    dev = _need_gpu(keys)
    n = keys.numel()
    score = torch.empty(n, dtype=torch.float32, device=dev)
    ids = torch.empty(n, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().eps_unpack_keys(_ptr(keys), n, _ptr(score), _ptr(ids), _stream(dev)),
                   "eps_unpack_keys")
    return score, ids
