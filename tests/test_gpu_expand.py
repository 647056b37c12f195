"""GPU parity of the fused candidate-generation + scoring kernels (eps_expand_count / eps_expand_fill) vs the
restated filter.py:96-109 candidate set (oracle.candidates_scipy) and the oracle pair scores."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_pair_files, rel_err

pytestmark = pytest.mark.gpu


def _check(eps, oracle, dev, A, weighted):
    import scipy.sparse as ssp
    n = A.shape[0]
    g = eps.CSRGraph.from_scipy(A, device=dev)
    rp, col = A.indptr.astype(np.int64), A.indices.astype(np.int32)
    val = A.data.astype(np.float32) if weighted else None
    want_pairs, a2 = oracle.candidates_scipy(A)
    w = oracle.node_weights(oracle.col_sums(rp, col, val, n), oracle.W_AA)
    from eps_amd.heuristics import node_weight_table
    wt = node_weight_table(g, eps.ops.W_AA)
    got_u, got_v, got_cn, got_sc = [], [], [], []
    step = max(1, n // 3)
    for lo in range(0, n, step):                 # several column blocks: colptr / offsets per block
        hi = min(n, lo + step)
        colptr, cu, cv, cn, sc = eps.ops.expand_candidates(g.rowptr, g.col, g.val, wt, n, lo, hi)
        assert colptr.numel() == hi - lo + 1 and int(colptr[-1]) == cu.numel()
        got_u.append(cu); got_v.append(cv); got_cn.append(cn); got_sc.append(sc)
    u = torch.cat(got_u).cpu().numpy(); v = torch.cat(got_v).cpu().numpy()
    cn = torch.cat(got_cn).cpu().numpy(); sc = torch.cat(got_sc).cpu().numpy()
    assert np.array_equal(np.stack([u, v], 1), want_pairs), "candidate set / column-major order"
    cnt_o, cn_o, ws_o = oracle.pair_scores(rp, col, val, w, want_pairs[:, 0], want_pairs[:, 1])
    assert np.array_equal(cn, cnt_o), "common-neighbour counts bit-exact"
    _, truth = oracle.pair_scores_f64(rp, col, val, w.astype(np.float64), want_pairs[:, 0], want_pairs[:, 1])
    assert rel_err(sc, truth.astype(np.float32)) <= 1e-6          # fixed-point accumulation: ~exact
    assert rel_err(sc, ws_o) <= 1e-5                               # and within the gate of the float32 oracle
    return len(u)


@pytest.mark.parametrize("path", golden_pair_files(), ids=lambda p: os.path.basename(p)[6:-4])
def test_expand_on_golden_graphs(eps, oracle, dev, path):
    import scipy.sparse as ssp
    d = np.load(path)
    n = len(d["rowptr"]) - 1
    A = ssp.csr_matrix((d["val"], d["col"], d["rowptr"]), shape=(n, n))
    _check(eps, oracle, dev, A, weighted=not bool((d["val"] == 1).all()))


def test_expand_rmat_and_determinism(eps, oracle, dev):
    from eps_amd import synth
    g = synth.rmat_graph(12, 10, 5, "cpu")
    A = g.to_scipy()
    n_cand = _check(eps, oracle, dev, A, weighted=False)
    assert n_cand > 100000
    gd = g.to(dev)
    from eps_amd.heuristics import node_weight_table
    wt = node_weight_table(gd, eps.ops.W_AA)
    a = eps.ops.expand_candidates(gd.rowptr, gd.col, None, wt, gd.n_rows, 0, gd.n_rows)
    b = eps.ops.expand_candidates(gd.rowptr, gd.col, None, wt, gd.n_rows, 0, gd.n_rows)
    assert torch.equal(a[4], b[4]) and torch.equal(a[1], b[1]), "fixed-point accumulation is order-independent"
    # the hand-out order of the columns (heaviest first in the product path) must not show in any output
    from eps_amd import candidates
    order = candidates.heaviest_first(gd, 0, gd.n_rows)
    assert sorted(order.tolist()) == list(range(gd.n_rows))
    assert int(candidates.path_counts(gd)[order[0]]) == int(candidates.path_counts(gd).max())
    c = eps.ops.expand_candidates(gd.rowptr, gd.col, None, wt, gd.n_rows, 0, gd.n_rows, col_order=order)
    rev = order.flip(0).contiguous()
    d = eps.ops.expand_candidates(gd.rowptr, gd.col, None, wt, gd.n_rows, 0, gd.n_rows, col_order=rev)
    for i in range(5):
        assert torch.equal(a[i], c[i]) and torch.equal(a[i], d[i])
    with pytest.raises(ValueError):
        eps.ops.expand_candidates(gd.rowptr, gd.col, None, wt, gd.n_rows, 0, gd.n_rows, col_order=order[:5])
    # every subset of the optional outputs (the scratch layout differs) gives the same arrays
    only_cn = eps.ops.expand_candidates(gd.rowptr, gd.col, None, None, gd.n_rows, 0, gd.n_rows, want_score=False)
    only_sc = eps.ops.expand_candidates(gd.rowptr, gd.col, None, wt, gd.n_rows, 0, gd.n_rows, want_cn=False, want_v=False)
    bare = eps.ops.expand_candidates(gd.rowptr, gd.col, None, None, gd.n_rows, 0, gd.n_rows, want_cn=False, want_score=False)
    assert torch.equal(only_cn[3], a[3]) and only_cn[4] is None and torch.equal(only_cn[1], a[1])
    assert torch.equal(only_sc[4], a[4]) and only_sc[3] is None and only_sc[2] is None and only_sc.pairs is None
    assert torch.equal(bare[1], a[1]) and torch.equal(bare[2], a[2]) and torch.equal(bare.pairs[0], a[1])
    # the lazily paired block of the filter stage: same pairs as the materialised ones
    blk = candidates.expand_block_lazy(gd, 0, gd.n_rows, wt, want_score=True, count_free=True)
    assert torch.equal(blk.pairs(), torch.stack([a[1], a[2]]).long())
    pick = torch.tensor([0, 5, blk.numel() // 2, blk.numel() - 1], device=dev)
    assert torch.equal(blk.select(blk.valid()[pick]), blk.pairs()[:, pick])
    assert blk.padded, "no counting pass: segments sized by the path counts"
    real = blk.valid()
    assert torch.equal(blk.cand_u[real], a[1]) and torch.equal(blk.score[real], a[4])
    assert torch.equal(blk.counts, a[0][1:] - a[0][:-1]) and blk.numel() == a[1].numel()
    pad = torch.ones_like(blk.cand_u, dtype=torch.bool); pad[real] = False
    assert bool((blk.cand_u[pad] == -1).all()) and bool(torch.isinf(blk.score[pad]).all()) and bool((blk.score[pad] < 0).all())
    both = candidates.expand_block_lazy(gd, 0, gd.n_rows, wt, want_score=True, want_cn=True, count_free=True)
    assert torch.equal(both.cn[real], a[3]) and bool((both.cn[pad] == 0).all())
    tight = candidates.expand_block_lazy(gd, 0, gd.n_rows, wt, want_score=True)
    assert not tight.padded and torch.equal(tight.cand_u, a[1]) and torch.equal(tight.score, a[4])
    # a segment layout that is too small for some column is refused, not overrun
    short = (a[0] // 2).contiguous()
    with pytest.raises(eps.EpsError):
        eps.ops.expand_candidates(gd.rowptr, gd.col, None, wt, gd.n_rows, 0, gd.n_rows, colptr_ub=short,
                                  total_ub=int(short[-1]))


def test_expand_matches_pair_kernel_at_scale(eps, dev):
    """ppa-like graph, a block of columns: fused expansion == candidate block + column-run pair kernel."""
    from eps_amd import candidates, synth
    from eps_amd.heuristics import node_weight_table
    g = synth.ppa_like(seed=3, device=dev, n_nodes=200_000, n_undirected=4_000_000)
    wt = node_weight_table(g, eps.ops.W_AA)
    colptr, cu, cv, cn, sc = eps.ops.expand_candidates(g.rowptr, g.col, None, wt, g.n_rows, 1000, 1400)
    pairs = candidates.two_hop_block(g, 1000, 1400)
    assert torch.equal(torch.stack([cu, cv]).long(), pairs)
    cnt, _, ws = eps.ops.pair_scores(g.rowptr, g.col, None, wt, g.n_rows, cu, cv, want_cn=False, grouped=True)
    assert torch.equal(cnt, cn)
    assert rel_err(sc.cpu().numpy(), ws.cpu().numpy()) <= 1e-5


def test_expand_rejects_oversized_id_space(eps, dev):
    """Node ids are int32: an id space of 2^31 is refused before anything is launched (wider-than-LDS id spaces below that
    are expanded in id windows: test_expand_id_range_boundaries)."""
    import ctypes
    from eps_amd import _lib
    rp = torch.zeros(16, dtype=torch.int64, device=dev)
    col = torch.zeros(16, dtype=torch.int32, device=dev)
    cnt = torch.zeros(16, dtype=torch.int64, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())       # noqa: E731
    rc = _lib.load().eps_expand_count(P(rp), P(col), 1 << 31, 0, 10, None, P(cnt), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc != 0 and b"int32" in _lib.load().eps_last_error()
    # an EMPTY graph over an id space wider than the LDS bitmap is fine (and has no candidates)
    n = eps.ops.expand_max_nodes() + 1
    r = eps.ops.expand_candidates(torch.zeros(n + 1, dtype=torch.int64, device=dev), torch.zeros(0, dtype=torch.int32, device=dev),
                                  None, None, n, 0, 10)
    assert r[1].numel() == 0 and int(r[0][-1]) == 0


def _dense_symmetric(n, p, seed, weighted):
    import scipy.sparse as ssp
    rng = np.random.default_rng(seed)
    m = np.triu(rng.random((n, n)) < p, 1)
    vals = (rng.integers(1, 5, (n, n)).astype(np.float32) if weighted else np.ones((n, n), np.float32)) * m
    A = ssp.csr_matrix(vals + vals.T)
    A.sort_indices()
    return A


@pytest.mark.parametrize("weighted", [False, True])
def test_expand_dense_graph_many_paths_per_candidate(eps, oracle, dev, weighted):
    """ddi-like density: every candidate is reached by dozens of paths (long accumulation chains per slot)."""
    A = _dense_symmetric(700, 0.12, 21, weighted)
    n_cand = _check(eps, oracle, dev, A, weighted=weighted)
    assert n_cand > 300_000


def test_expand_many_small_tiles(eps, oracle, dev):
    """The score pass bins the paths of a column by candidate-rank tile (8192 ranks in production, so small graphs
    are single-tile).  Forced down to 256 / 512 ranks here: every column of the 700-node graph then spans several
    tiles, and the outputs must not change."""
    from eps_amd.heuristics import node_weight_table
    A = _dense_symmetric(700, 0.05, 22, True)
    g = eps.CSRGraph.from_scipy(A, device=dev)
    wt = node_weight_table(g, eps.ops.W_AA)
    full = eps.ops.expand_candidates(g.rowptr, g.col, g.val, wt, g.n_rows, 0, g.n_rows)
    assert int((full[0][1:] - full[0][:-1]).max()) > 512
    for tile in (512, 1024):
        got = eps.ops.expand_candidates(g.rowptr, g.col, g.val, wt, g.n_rows, 0, g.n_rows, tile_ranks=tile)
        for i in range(5):
            assert torch.equal(full[i], got[i]), (tile, i)
    with pytest.raises(eps.EpsError):
        eps.ops.expand_candidates(g.rowptr, g.col, g.val, wt, g.n_rows, 0, g.n_rows, tile_ranks=100)
    _check(eps, oracle, dev, A, weighted=True)


def test_expand_rejects_undersized_buckets(eps, dev):
    """max_paths below the heaviest column's path count: the kernel flags it instead of writing past its buckets."""
    from eps_amd import synth
    from eps_amd.heuristics import node_weight_table
    g = synth.rmat_graph(10, 8, 4, dev)
    wt = node_weight_table(g, eps.ops.W_AA)
    true_max = eps.ops.max_column_paths(g.rowptr, g.col, 0, g.n_rows)
    ok = eps.ops.expand_candidates(g.rowptr, g.col, None, wt, g.n_rows, 0, g.n_rows, max_paths=true_max)
    auto = eps.ops.expand_candidates(g.rowptr, g.col, None, wt, g.n_rows, 0, g.n_rows)
    assert torch.equal(ok[4], auto[4]) and torch.equal(ok[3], auto[3])
    with pytest.raises(eps.EpsError):
        eps.ops.expand_candidates(g.rowptr, g.col, None, wt, g.n_rows, 0, g.n_rows, max_paths=true_max // 2)


def test_expand_full_size_properties(eps, dev):
    """BASELINE scale (ppa-like, N = 576,289), one production-sized launch (2^29 two-hop paths, ~4e8 candidates), no
    oracle run: (1) a 2 M sample of the candidates re-scored by the column-run intersection kernel -- CN equal, AA within
    the gate; (2) sum of CN == the paths that end in candidates, counted independently; (3) symmetry: (u,v) and (v,u)
    are both candidates and carry bit-identical scores (integer accumulation does not depend on the path order)."""
    from eps_amd import candidates, synth
    from eps_amd.heuristics import node_weight_table
    g = synth.ppa_like(seed=3, device=dev)
    wt = node_weight_table(g, eps.ops.W_AA)
    lo, hi = next(iter(candidates.column_blocks(g)))
    r = eps.ops.expand_candidates(g.rowptr, g.col, None, wt, g.n_rows, lo, hi,
                                  col_order=candidates.heaviest_first(g, lo, hi), max_paths=candidates.max_paths_of(g))
    colptr, cu, cv, cn, sc = r
    E = cu.numel()
    assert E > 100_000_000 and int(colptr[-1]) == E
    assert bool((cn >= 1).all()) and bool(torch.isfinite(sc).all()) and bool((sc > 0).all())
    gen = torch.Generator(device=dev).manual_seed(3)
    pick = torch.randint(0, E, (2_000_000,), generator=gen, device=dev)
    cnt, _, ws = eps.ops.pair_scores(g.rowptr, g.col, None, wt, g.n_rows, cu[pick].contiguous(), cv[pick].contiguous(),
                                     want_cn=False, grouped=False)
    assert torch.equal(cnt, cn[pick])
    assert rel_err(sc[pick].cpu().numpy(), ws.cpu().numpy()) <= 1e-5
    # (2) every two-hop path of the block ends in a candidate, a neighbour of v, or v itself
    paths = int(candidates.path_counts(g)[lo:hi].sum())
    deg = g.degree()
    rows = torch.repeat_interleave(torch.arange(lo, hi, device=dev), (g.rowptr[lo + 1:hi + 1] - g.rowptr[lo:hi]))
    nb = g.col[g.rowptr[lo]:g.rowptr[hi]].long()
    back_to_v = int(deg[lo:hi].sum())                       # v - w - v, one per neighbour w
    cn_nb, _, _ = eps.ops.pair_scores(g.rowptr, g.col, None, None, g.n_rows, nb.to(torch.int32), rows.to(torch.int32),
                                      want_cn=False)       # paths v - w - u with u a neighbour of v
    assert int(cn.sum(dtype=torch.int64)) == paths - back_to_v - int(cn_nb.sum(dtype=torch.int64))
    # (3) symmetry inside the block's column range
    inside = pick[(cu[pick] >= lo) & (cu[pick] < hi)][:500_000]
    u, v = cu[inside].long(), cv[inside].long()
    seg_lo, seg_hi = colptr[u - lo], colptr[u - lo + 1]          # column u's segment; find v in it (ascending)
    pos = seg_lo.clone()
    step = int((seg_hi - seg_lo).max())
    span = 1
    while span < step:
        span <<= 1
    while span:                                                  # vectorised binary search (lower bound)
        probe = pos + span
        ok = (probe <= seg_hi) & (cu[(probe - 1).clamp(max=E - 1)].long() < v)
        pos = torch.where(ok, probe, pos)
        span >>= 1
    assert bool((pos < seg_hi).all()) and torch.equal(cu[pos].long(), v) and torch.equal(cv[pos].long(), u)
    assert torch.equal(sc[pos], sc[inside]) and torch.equal(cn[pos], cn[inside])


def test_one_pass_list_full_size(eps, dev):
    """BASELINE scale (ppa-like, N = 576,289): the ONE-PASS list (eps_expand_unit_list) over the WHOLE graph in blocks of 2^33 two-hop
    paths.  (1) its per-column counts sum to the candidate set's size as a counting scan reports it independently; (2) on the first
    production-sized block (2^31 paths) the list and the scores are bit-identical to eps_expand_unit_count / _fill AND to
    eps_expand_count / _fill (expand_score.hip: a different kernel); (3) for a dozen columns -- hubs, median degree, tail -- the
    rows are the ORACLE's restated filter.py:96-109 candidates with its pair scores (ids exact, scores <= 1e-5 relative)."""
    from eps_amd import candidates, scan, synth
    from eps_amd.heuristics import node_weight_table
    from oracle import eps_oracle as orc
    g = synth.ppa_like(seed=3, device=dev)
    wt = node_weight_table(g, eps.ops.W_AA)
    n = g.n_rows
    md, sp = scan.max_degree(g), scan.window_splits(g)
    pre, pre_host = candidates.segment_bounds(g)
    deg = g.degree()
    by_deg = torch.argsort(deg, descending=True)
    cols = sorted({int(by_deg[i]) for i in (0, 3, 50, 1000, n // 4, n // 2, 3 * n // 4, n - 1000, n - 2)} | {7, n - 1})
    counts, kept = [], {}
    for lo, hi in candidates.column_blocks(g, 1 << 33):
        ub = (pre[lo:hi + 1] - pre[lo]).contiguous()
        r = eps.ops.expand_unit(g.rowptr, g.col, wt, n, lo, hi, md, sp, want_v=False, col_order=candidates.heaviest_first(g, lo, hi),
                                colptr_ub=ub, total_ub=int(pre_host[hi] - pre_host[lo]))
        assert int(r.status) == 0
        counts.append(r.counts)
        for c in cols:
            if lo <= c < hi:
                a, m = int(ub[c - lo]), int(r.counts[c - lo])
                kept[c] = (r[1][a:a + m].cpu().numpy(), r[4][a:a + m].cpu().numpy())
        del r
    counts = torch.cat(counts)
    assert bool((counts <= pre[1:] - pre[:-1]).all())
    # (1) the whole candidate set: both orientations of every unordered 2-hop non-edge
    fixw = scan.fixed_weights(g, wt)
    assert int(counts.sum()) == 2 * scan.candidate_count(g, None, fixw)
    # (2) block 0 of the production block size, bit for bit, against both two-pass kernels
    lo, hi = next(iter(candidates.column_blocks(g)))
    order = candidates.heaviest_first(g, lo, hi)
    ub = (pre[lo:hi + 1] - pre[lo]).contiguous()
    one = eps.ops.expand_unit(g.rowptr, g.col, wt, n, lo, hi, md, sp, want_v=False, col_order=order, colptr_ub=ub,
                              total_ub=int(pre_host[hi] - pre_host[lo]))
    two = eps.ops.expand_unit(g.rowptr, g.col, wt, n, lo, hi, md, sp, want_v=False, col_order=order)
    assert torch.equal(one.counts, two[0][1:] - two[0][:-1]) and torch.equal(one.counts, counts[lo:hi])
    slot = torch.arange(int(ub[-1]), device=dev)
    seg = torch.searchsorted(ub[1:], slot, right=True)
    real = slot - ub[seg] < one.counts[seg]
    del slot, seg
    assert torch.equal(one[1][real], two[1]) and torch.equal(one[4][real], two[4])
    del one, real
    old = eps.ops.expand_candidates(g.rowptr, g.col, None, wt, n, lo, hi, want_cn=False, want_v=False, col_order=order,
                                    max_paths=candidates.max_paths_of(g))
    assert torch.equal(old[0], two[0]) and torch.equal(old[1], two[1]) and torch.equal(old[4], two[4])
    del old, two
    # (3) the oracle's own candidate set and scores for the sampled columns
    A = g.to_scipy()
    rp, col = A.indptr.astype(np.int64), A.indices.astype(np.int32)
    w = orc.node_weights(orc.col_sums(rp, col, None, n), orc.W_AA)
    for c in cols:
        cand, _ = orc.candidates_scipy_columns(A, c, c + 1)
        _, _, sc = orc.pair_scores(rp, col, None, w, cand[:, 0], cand[:, 1])
        u, s_ = kept[c]
        assert np.array_equal(u, cand[:, 0].astype(np.int32)), f"column {c}: candidates differ from the oracle's"
        assert rel_err(s_, sc) <= 1e-5, f"column {c}"


def test_expand_falls_back_when_buckets_would_not_fit(eps, dev, monkeypatch):
    """A graph whose heaviest column needs more bucket scratch than the budget is not offered to the fused kernels:
    the same block then comes from the tensor-op expansion + the column-run intersection kernel, with equal results."""
    from eps_amd import candidates, synth
    from eps_amd.heuristics import node_weight_table
    g = synth.rmat_graph(11, 8, 6, dev)
    wt = node_weight_table(g, eps.ops.W_AA)
    assert candidates.hip_expand_available(g)
    fused = candidates.expand_block(g, 0, g.n_rows, wt, want_cn=True, want_score=True)
    monkeypatch.setattr(eps.ops, "_EXPAND_WS_LIMIT", 1 << 16)
    assert not candidates.hip_expand_available(g)
    plain = candidates.expand_block(g, 0, g.n_rows, wt, want_cn=True, want_score=True)
    assert torch.equal(fused[0], plain[0]) and torch.equal(fused[1], plain[1])
    assert rel_err(fused[2].cpu().numpy(), plain[2].cpu().numpy()) <= 1e-5


def test_expand_score_cut(eps, dev):
    """eps_score_cut: the kernel's survivor list == the positions whose score exceeds the threshold (ascending after the
    host's sort), with or without the score array, in the counted and in the count-free layout; a list that overflows
    its capacity is reported as missing, not truncated."""
    from eps_amd import candidates, synth
    from eps_amd.heuristics import node_weight_table
    g = synth.rmat_graph(12, 10, 5, dev)
    wt = node_weight_table(g, eps.ops.W_AA)
    full = eps.ops.expand_candidates(g.rowptr, g.col, None, wt, g.n_rows, 0, g.n_rows, want_cn=False)
    thr = float(torch.quantile(full[4][:1_000_000].float(), 0.999))
    want = torch.nonzero(full[4] > thr).squeeze(1)
    assert 50 < want.numel() < 100_000
    for want_score in (True, False):
        r = eps.ops.expand_candidates(g.rowptr, g.col, None, wt, g.n_rows, 0, g.n_rows, want_cn=False,
                                      want_score=want_score, cut=(thr, 1 << 20))
        pos, val = r.survivors
        assert torch.equal(pos, want) and torch.equal(val, full[4][want])
        assert (r[4] is None) == (not want_score)
        if want_score:
            assert torch.equal(r[4], full[4])
    blk = candidates.expand_block_lazy(g, 0, g.n_rows, wt, want_score=False, cut=(thr, 1 << 20), count_free=True)
    pos, val = blk.survivors
    assert blk.padded and torch.equal(blk.select(pos), torch.stack([full[1][want], full[2][want]]).long())
    assert torch.equal(val, full[4][want])
    small = eps.ops.expand_candidates(g.rowptr, g.col, None, wt, g.n_rows, 0, g.n_rows, want_cn=False, want_score=False,
                                      cut=(thr, 8))
    assert small.survivors is None
    none = eps.ops.expand_candidates(g.rowptr, g.col, None, wt, g.n_rows, 0, g.n_rows, want_cn=False, want_score=False,
                                     cut=(float("inf"), 8))
    assert none.survivors[0].numel() == 0


@pytest.mark.parametrize("n", [255, 257, 131072, 131073, 262145, 851968, 851969, 1_100_000, 1_703_937])
def test_expand_id_range_boundaries(eps, dev, n):
    """Node counts around the id-range table's limits (512 ranges of 2^k ids: k changes at 131,073 and 262,145 nodes; a
    single range below 257; 851,968 = eps_expand_max_nodes(), the largest LDS footprint): fused expansion of some columns == tensor-op candidates + column-run intersection kernel."""
    from eps_amd import candidates, synth
    from eps_amd.heuristics import node_weight_table
    g = synth.rmat_graph(max(8, (n - 1).bit_length()), 6, 9, dev, n_nodes=n)
    assert g.n_rows == n
    wt = node_weight_table(g, eps.ops.W_AA)
    lo, hi = (0, n) if n < 1000 else (n - 700, n)           # the last columns: ids in the last, partial range
    colptr, cu, cv, cn, sc = eps.ops.expand_candidates(g.rowptr, g.col, None, wt, n, lo, hi)
    pairs = candidates.two_hop_block(g, lo, hi)
    assert torch.equal(torch.stack([cu, cv]).long(), pairs)
    if cu.numel():
        cnt, _, ws = eps.ops.pair_scores(g.rowptr, g.col, None, wt, n, cu, cv, want_cn=False, grouped=True)
        assert torch.equal(cnt, cn) and rel_err(sc.cpu().numpy(), ws.cpu().numpy()) <= 1e-5


def test_expand_wide_id_space_vs_oracle(eps, oracle, dev):
    """filter.py:96-109 on an id space wider than the LDS bitmap (N = 2^21 + 77: three id windows): candidates of column
    blocks at the start, across a window boundary and at the end == the restated A @ A slice, in the reference's order;
    common-neighbour counts == the A @ A values; scores within the gate; upper-bound (count-free) layout included."""
    from eps_amd import candidates, synth
    from eps_amd.heuristics import node_weight_table
    n = (1 << 21) + 77
    g = synth.rmat_graph(21, 2, 31, "cpu", n_nodes=n)
    A = g.to_scipy()
    g = g.to(dev)
    wt = node_weight_table(g, eps.ops.W_AA)
    rp, col = A.indptr.astype(np.int64), A.indices.astype(np.int32)
    w = oracle.node_weights(oracle.col_sums(rp, col, None, n), oracle.W_AA)
    win = eps.ops.expand_max_nodes()
    deg = np.diff(rp)
    hub = int(np.argmax(deg))
    for lo, hi in ((0, 300), (win - 150, win + 150), (n - 300, n), (hub, hub + 1)):
        want, a2 = oracle.candidates_scipy_columns(A, lo, hi)
        colptr, cu, cv, cn, sc = eps.ops.expand_candidates(g.rowptr, g.col, None, wt, n, lo, hi)
        assert np.array_equal(torch.stack([cu, cv], 1).cpu().numpy(), want), (lo, hi)
        assert np.array_equal(cn.cpu().numpy(), a2.astype(np.int32))
        _, truth = oracle.pair_scores_f64(rp, col, None, w.astype(np.float64), want[:, 0], want[:, 1])
        assert rel_err(sc.cpu().numpy(), truth.astype(np.float32)) <= 1e-6
        blk = candidates.expand_block_lazy(g, lo, hi, wt, want_score=True, count_free=True)
        idx = blk.valid()
        assert np.array_equal(blk.select(idx).t().cpu().numpy(), want) and torch.equal(blk.score[idx], sc)
    assert want.shape[0] > 10000 and int((want[:, 0] >= 2 * win).sum()) > 0 and int((want[:, 0] < win).sum()) > 0


def _unit_vs_expand(eps, g, wt, lo, hi, col_order=None):
    """eps_expand_unit_* (scan-structured list kernels) against eps_expand_* on the same block: same bits."""
    from eps_amd import scan
    md, sp = scan.max_degree(g), scan.window_splits(g)
    want = eps.ops.expand_candidates(g.rowptr, g.col, None, wt, g.n_rows, lo, hi, want_cn=False, col_order=col_order)
    got = eps.ops.expand_unit(g.rowptr, g.col, wt, g.n_rows, lo, hi, md, sp, col_order=col_order)
    assert torch.equal(got[0], want[0]), "colptr"
    assert torch.equal(got[1], want[1]) and torch.equal(got[2], want[2]), "candidate list"
    assert got[3] is None and torch.equal(got[4], want[4]), "scores bit-identical"
    lst = eps.ops.expand_unit(g.rowptr, g.col, None, g.n_rows, lo, hi, md, sp, want_score=False, want_v=False)
    assert torch.equal(lst[0], want[0]) and torch.equal(lst[1], want[1]) and lst[2] is None and lst[4] is None
    # the HALF list (revpos given; symmetric patterns): exactly the u < v part of every column, same score bits
    from eps_amd import candidates as _cand, scan as _scan
    if _scan.is_symmetric(g):
        half = eps.ops.expand_unit(g.rowptr, g.col, wt, g.n_rows, lo, hi, md, sp, col_order=col_order, revpos=_scan.reverse_positions(g))
        keep = want[1] < want[2]
        assert torch.equal(half[1], want[1][keep]) and torch.equal(half[2], want[2][keep]) and torch.equal(half[4], want[4][keep])
        per_col = torch.zeros(hi - lo, dtype=torch.int64, device=want[1].device)
        per_col.index_add_(0, (want[2][keep] - lo).long(), torch.ones(int(keep.sum()), dtype=torch.int64, device=want[1].device))
        assert torch.equal(half[0][1:] - half[0][:-1], per_col)
        if hi > lo:                                  # ... and the one-pass form of the half list (the same bounds hold)
            pre_h = _cand.segment_bounds(g)[0]
            ub_h = (pre_h[lo:hi + 1] - pre_h[lo]).contiguous()
            one_h = eps.ops.expand_unit(g.rowptr, g.col, wt, g.n_rows, lo, hi, md, sp, want_v=False, col_order=col_order,
                                        revpos=_scan.reverse_positions(g), colptr_ub=ub_h, total_ub=int(ub_h[-1]))
            assert int(one_h.status) == 0 and torch.equal(one_h.counts, per_col)
            slot_h = torch.arange(int(ub_h[-1]), device=ub_h.device)
            seg_h = torch.searchsorted(ub_h[1:], slot_h, right=True)
            real_h = slot_h - ub_h[seg_h] < one_h.counts[seg_h]
            assert torch.equal(one_h[1][real_h], half[1]) and torch.equal(one_h[4][real_h], half[4])
    # the ONE-PASS list (eps_expand_unit_list: upper-bound segments, no counting launch): the front of every segment holds the
    # column's candidates and scores -- same bits --, the counts come back, nothing is written behind them
    from eps_amd import candidates as _cand
    pre = _cand.segment_bounds(g)[0]
    ub = (pre[lo:hi + 1] - pre[lo]).contiguous()
    total_ub = int(ub[-1]) if hi > lo else 0
    one = eps.ops.expand_unit(g.rowptr, g.col, wt, g.n_rows, lo, hi, md, sp, want_v=False, col_order=col_order, colptr_ub=ub,
                              total_ub=total_ub)
    assert int(one.status) == 0
    assert torch.equal(one.counts, want[0][1:] - want[0][:-1]), "one-pass counts"
    if total_ub:
        slot = torch.arange(total_ub, device=ub.device)
        seg = torch.searchsorted(ub[1:], slot, right=True)
        real = slot - ub[seg] < one.counts[seg]
        assert int(real.sum()) == int(want[0][-1])
        assert torch.equal(one[1][real], want[1]) and torch.equal(one[4][real], want[4]), "one-pass list: same bits"
    cn = eps.ops.expand_unit(g.rowptr, g.col, None, g.n_rows, lo, hi, md, sp)         # all-ones weights: the CN count
    ref_cn = eps.ops.expand_candidates(g.rowptr, g.col, None, None, g.n_rows, lo, hi, want_cn=True, want_score=False)[3]
    assert torch.equal(cn[4], ref_cn.to(torch.float32))
    return int(want[0][-1])


@pytest.mark.parametrize("path", golden_pair_files(), ids=lambda p: os.path.basename(p)[6:-4])
def test_expand_unit_on_golden_graphs(eps, dev, path):
    from eps_amd.heuristics import node_weight_table
    d = np.load(path)
    n = len(d["rowptr"]) - 1
    g = eps.CSRGraph(torch.from_numpy(d["rowptr"].astype(np.int64)).to(dev), torch.from_numpy(d["col"].astype(np.int32)).to(dev),
                     None, n, n)                                           # the structure only: unit values
    wt = node_weight_table(g, eps.ops.W_AA)
    step = max(1, n // 3)
    for lo in range(0, n, step):
        _unit_vs_expand(eps, g, wt, lo, min(n, lo + step))


def test_expand_unit_rmat_hubs_and_order(eps, dev):
    """R-MAT graph with hub columns (several rounds of row descriptors, several tiles), a hand-out order, partial blocks."""
    from eps_amd import candidates, synth
    from eps_amd.heuristics import node_weight_table
    g = synth.rmat_graph(15, 16, 5, dev)
    wt = node_weight_table(g, eps.ops.W_AA)
    n = g.n_rows
    assert _unit_vs_expand(eps, g, wt, 0, n // 2, col_order=candidates.heaviest_first(g, 0, n // 2)) > 1_000_000
    _unit_vs_expand(eps, g, wt, n // 2, n)
    _unit_vs_expand(eps, g, wt, 17, 18)
    _unit_vs_expand(eps, g, wt, 5, 5)


def test_expand_unit_wide_id_space(eps, dev):
    """An id space wider than the LDS bitmap (N = 1.3 M: id windows): candidates still come out in ascending u."""
    from eps_amd import synth
    from eps_amd.heuristics import node_weight_table
    g = synth.rmat_graph(21, 3, 11, dev, n_nodes=1_300_000)
    assert eps.ops.filter_scan_windows(g.n_rows)[1] > 1
    wt = node_weight_table(g, eps.ops.W_AA)
    assert _unit_vs_expand(eps, g, wt, 400_000, 440_000) > 100_000
    _unit_vs_expand(eps, g, wt, 1_299_000, 1_300_000)


def test_expand_unit_upper_bound_layout_and_status(eps, dev):
    """eps_expand_unit_fill through the C ABI with an UPPER-BOUND colptr (segments of min(paths, N) slots: no counting pass):
    the front of every segment holds the column's candidates, the rest is padded (cand_u -1, score -inf), cand_count receives
    the real counts; segments that are too small raise status bit 1 (value 2)."""
    import ctypes
    from eps_amd import _lib, candidates, scan, synth
    from eps_amd.heuristics import node_weight_table
    g = synth.rmat_graph(13, 10, 9, dev)
    n = g.n_rows
    wt = node_weight_table(g, eps.ops.W_AA)
    lo, hi = 100, 5000
    want = eps.ops.expand_unit(g.rowptr, g.col, wt, n, lo, hi, scan.max_degree(g), scan.window_splits(g))
    counts_want = want[0][1:] - want[0][:-1]
    ub = torch.clamp(candidates.path_counts(g)[lo:hi], max=n)
    colptr = torch.zeros(hi - lo + 1, dtype=torch.int64, device=dev)
    torch.cumsum(ub, 0, out=colptr[1:])
    total = int(colptr[-1])
    cu = torch.full((total,), -7, dtype=torch.int32, device=dev)
    cv = torch.full((total,), -7, dtype=torch.int32, device=dev)
    sc = torch.full((total,), 123.0, dtype=torch.float32, device=dev)
    cnt = torch.full((hi - lo,), -1, dtype=torch.int64, device=dev)
    status = torch.ones(1, dtype=torch.int32, device=dev)
    fixw = eps.ops.fixed_weights(wt)
    ws = eps.ops._scan_scratch(dev, scan.max_degree(g))
    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())       # noqa: E731
    lib = _lib.load()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def fill(colptr_t, cnt_t):
        return lib.eps_expand_unit_fill(P(g.rowptr), P(g.col), None, P(fixw), None, n, g.nnz(), scan.max_degree(g), lo, hi, None,
                                        P(colptr_t), P(cnt_t), P(cu), P(cv), P(sc), P(status), P(ws), ws.numel() * 8, stream)
    assert fill(colptr, cnt) == 0
    torch.cuda.synchronize()
    assert int(status) == 0 and torch.equal(cnt, counts_want)
    real = cu >= 0
    assert torch.equal(cu[real], want[1]) and torch.equal(sc[real], want[4]) and torch.equal(cv[real], want[2])
    assert bool((sc[~real] == float("-inf")).all()) and bool((cu[~real] == -1).all())
    seg = torch.repeat_interleave(torch.arange(hi - lo, device=dev), ub)          # padding sits at the END of each segment
    first_pad = colptr[:-1] + counts_want
    assert bool((real == (torch.arange(total, device=dev) < first_pad[seg])).all())
    # segments one slot too small for the columns that have candidates: flagged, nothing is written past a segment
    small = torch.zeros_like(colptr)
    torch.cumsum(torch.clamp(counts_want - 1, min=0), 0, out=small[1:])
    assert fill(small, None) == 0
    torch.cuda.synchronize()
    assert int(status) & 2
