"""GPU parity of the threshold scan (eps_filter_scan, csrc/filter_scan.hip) and of the exact top-K built on it
(edge-proposal-sets_amd/scan.py) vs the restated filter.py:96-109 candidate set + the oracle pair scores, and vs
the fused expansion kernel (bit-identical scores by construction)."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_pair_files, rel_err

pytestmark = pytest.mark.gpu


def _oracle_candidates(oracle, A):
    """(pairs [E,2] column-major, AA float64-accumulated truth, AA float32 oracle, CN counts) of every 2-hop non-edge."""
    n = A.shape[0]
    rp, col = A.indptr.astype(np.int64), A.indices.astype(np.int32)
    pairs, _ = oracle.candidates_scipy(A)
    w = oracle.node_weights(oracle.col_sums(rp, col, None, n), oracle.W_AA)
    cnt, _, ws = oracle.pair_scores(rp, col, None, w, pairs[:, 0], pairs[:, 1])
    _, truth = oracle.pair_scores_f64(rp, col, None, w.astype(np.float64), pairs[:, 0], pairs[:, 1])
    return pairs, truth, ws, cnt


def _scan_all(eps, g, node_w, thr=float("-inf"), columns=None):
    """Every survivor of one launch as {(u, v): score} with u < v, plus the kernel's candidate count."""
    from eps_amd import scan
    fixw = scan.fixed_weights(g, node_w)
    cols = scan.column_order(g) if columns is None else columns
    cap = 2 * int(scan.half_paths(g)[cols.long()].sum().item()) + 8192 * 300
    res = eps.ops.Survivors(cap, thr, g.device)
    eps.ops.filter_scan(g.rowptr, g.col, scan.reverse_positions(g), fixw, g.n_rows, cols, res, scan.max_degree(g), scan.window_splits(g))
    slots, n_cand = res.counts()
    assert slots <= res.capacity
    keys, vals = res.valid(slots)
    keys, vals = keys.cpu().numpy(), vals.cpu().numpy()
    assert len(np.unique(keys)) == len(keys), "a candidate was reported twice"
    return {(int(k & 0xFFFFFFFF), int(k >> 32)): float(s) for k, s in zip(keys, vals)}, n_cand


def _screen_all(eps, g, node_w, thr=float("-inf"), columns=None, variant=2, packed=False, table=False):
    """The same through the one-pass kernel (eps_scan_screen + exact re-scoring): {(u, v): score}, candidate count.
    ``packed``: hand the kernel the per-node sum bounds, so pieces may keep key and sum in one table word (or two 16-bit sums);
    ``table``: the per-graph plan table (eps_scan_plan) instead of planning inside the launch."""
    from eps_amd import scan
    sc = scan.screen_weights(g, g, None, node_w)
    assert sc.usable
    bounds, cuts = scan.screen_tables(g)
    cols = scan.column_order(g) if columns is None else columns
    cap = 2 * int(scan.half_paths(g)[cols.long()].sum().item()) + scan._PIECE_SLACK
    res = eps.ops.Survivors(cap, thr, g.device)
    status = torch.zeros(1, dtype=torch.int32, device=g.device)
    plan = None
    if table and sc.val is None and g.n_rows:
        plan = eps.ops.scan_plan(g.rowptr, cuts, scan.window_paths(g), sc.ssum if packed else None, sc.smax if packed else None,
                                 bounds, g.n_rows, sc.shift, variant)
        assert int(plan[0][-1]) <= plan[1].shape[0] and bool((plan[0][1:] >= plan[0][:-1]).all())
    eps.ops.scan_screen(g.rowptr, g.col, scan.reverse_positions(g), sc.fx32, cuts, bounds, g.n_rows, cols, sc.shift, res, status,
                        variant, sc.val, sc.node_w, wpaths=scan.window_paths(g),
                        ssum=sc.ssum if packed else None, smax=sc.smax if packed else None, plan=plan)
    assert not packed or sc.val is not None or sc.ssum is not None or not scan.one_pass_available(g)
    slots, n_cand = res.counts()
    assert slots <= res.capacity and int(status) == 0
    keys, approx = res.valid(slots)
    assert torch.unique(keys).numel() == keys.numel(), "a candidate was reported twice"
    k2, v2 = scan.rescore_exact(g, sc, keys, None if thr == float("-inf") else torch.tensor([thr], device=g.device))
    if keys.numel():
        o1, o2 = torch.argsort(keys), torch.argsort(k2.where(k2 >= 0, keys.max() + 1 + torch.arange(k2.numel(), device=k2.device)))
        live = k2 >= 0
        # screening scores are upper bounds of the exact ones (that is what makes the screening lossless)
        exact_of = dict(zip(k2[live].tolist(), v2[live].tolist()))
        for kk, aa in zip(keys[:2000].tolist(), approx[:2000].tolist()):
            if kk in exact_of:
                assert aa >= exact_of[kk] * (1 - 1e-6) - 1e-6
    m = k2 >= 0
    return {(int(k & 0xFFFFFFFF), int(k >> 32)): float(x) for k, x in zip(k2[m].tolist(), v2[m].tolist())}, n_cand


def _unit_graph(d):
    import scipy.sparse as ssp
    n = len(d["rowptr"]) - 1
    A = ssp.csr_matrix((np.ones_like(d["val"]), d["col"], d["rowptr"]), shape=(n, n))
    A.sort_indices()
    return A


@pytest.mark.parametrize("path", golden_pair_files(), ids=lambda p: os.path.basename(p)[6:-4])
def test_scan_without_bar_is_the_whole_candidate_set(eps, oracle, dev, path):
    """No bar: every unordered 2-hop non-edge comes out exactly once, with the fused expansion's score."""
    from eps_amd.heuristics import node_weight_table
    A = _unit_graph(np.load(path))
    g = eps.CSRGraph.from_scipy(A, device=dev)
    wt = node_weight_table(g, eps.ops.W_AA)
    pairs, truth, ws_o, _ = _oracle_candidates(oracle, A)
    got, n_cand = _scan_all(eps, g, wt)
    lower = pairs[:, 0] < pairs[:, 1]
    want = {(int(u), int(v)): t for (u, v), t in zip(pairs[lower], truth[lower])}
    assert n_cand == len(want) and set(got) == set(want)
    if want:
        ks = sorted(want)
        a = np.array([got[k] for k in ks], np.float32)
        assert rel_err(a, np.array([want[k] for k in ks], np.float32)) <= 1e-6
        # bit-identical to the fused expansion's float32 scores, and symmetric
        _, cu, cv, _, sc = eps.ops.expand_candidates(g.rowptr, g.col, None, wt, g.n_rows, 0, g.n_rows, want_cn=False)
        ex = {(int(u), int(v)): float(s) for u, v, s in zip(cu.cpu().numpy(), cv.cpu().numpy(), sc.cpu().numpy())}
        for (u, v), s in got.items():
            assert ex[(u, v)] == s and ex[(v, u)] == s
    # the PRODUCTION kernel (eps_scan_screen + exact re-scoring) meets every reference-generated graph directly: all four
    # variant / packed / plan-table combinations against the oracle truth, not through the two-pass kernel's list
    from eps_amd import scan
    if scan.one_pass_available(g):
        for variant, packed, table in ((2, False, False), (2, True, False), (2, True, True), (0, True, True)):
            got_s, n_s = _screen_all(eps, g, wt, variant=variant, packed=packed, table=table)
            assert n_s == len(want) and set(got_s) == set(want)
            if want:
                assert rel_err(np.array([got_s[k] for k in ks], np.float32), np.array([want[k] for k in ks], np.float32)) <= 1e-6
                assert got_s == got                       # (and bit-identical to the two-pass kernel's scores)


@pytest.mark.parametrize("seed,scale,ef", [(5, 12, 10), (9, 13, 6), (2, 11, 40)])
def test_scan_bar_and_column_subsets(eps, oracle, dev, seed, scale, ef):
    """Survivors == {candidates with score > bar} for several bars; column subsets partition the result; the common-
    neighbour count (unit weights) is exact.  scale 11 / edge factor 40 makes multi-tile, multi-round hub columns."""
    from eps_amd import scan, synth
    from eps_amd.heuristics import node_weight_table
    g = synth.rmat_graph(scale, ef, seed, "cpu")
    A = g.to_scipy()
    g = g.to(dev)
    wt = node_weight_table(g, eps.ops.W_AA)
    pairs, truth, _, cnt = _oracle_candidates(oracle, A)
    lower = pairs[:, 0] < pairs[:, 1]
    full, n_cand = _scan_all(eps, g, wt)
    assert n_cand == int(lower.sum()) and len(full) == n_cand
    sc = np.array(sorted(full.values()), np.float32)
    for q in (0.5, 0.99, 0.9999):
        bar = float(sc[int(q * (len(sc) - 1))])
        got, n2 = _scan_all(eps, g, wt, thr=bar)
        assert n2 == n_cand
        assert got == {k: s for k, s in full.items() if s > bar}
    order = scan.column_order(g)
    parts = [_scan_all(eps, g, wt, columns=order[r::3].contiguous()) for r in range(3)]
    merged = {}
    for d, _ in parts:
        assert not (set(d) & set(merged))
        merged.update(d)
    assert merged == full and sum(n for _, n in parts) == n_cand
    ones = torch.ones(g.n_rows, dtype=torch.float32, device=dev)
    cn_got, _ = _scan_all(eps, g, ones)
    want_cn = {(int(u), int(v)): float(c) for (u, v), c in zip(pairs[lower], cnt[lower])}
    assert cn_got == want_cn
    # the one-pass kernel on the same inputs: whole set, a high bar, a column shard, unit weights
    for variant, packed, table in ((2, False, False), (2, True, False), (2, True, True), (0, True, True)):
        got, n3 = _screen_all(eps, g, wt, variant=variant, packed=packed, table=table)
        assert n3 == n_cand and got == full
        bar = float(sc[int(0.99 * (len(sc) - 1))])
        got, _ = _screen_all(eps, g, wt, thr=bar, variant=variant, packed=packed, table=table)
        assert got == {k: s for k, s in full.items() if s > bar}
        got, n4 = _screen_all(eps, g, wt, columns=order[1::3].contiguous(), variant=variant, packed=packed, table=table)
        assert got == parts[1][0] and n4 == parts[1][1]
        got, _ = _screen_all(eps, g, ones, variant=variant, packed=packed, table=table)
        assert got == want_cn


@pytest.mark.parametrize("heads", [False, True], ids=["default", "heads"])
def test_scan_topk_is_exact(eps, oracle, dev, monkeypatch, heads):
    """scan_topk == the first K rows of the declared order over the full candidate list, on the direct path (small
    set, no bar) and on the estimate -> scan -> verify path, incl. a bar that is too high and one that is too low.
    ``heads``: HEAD_MIN_PATHS = 0 -- with hubs-first labels the production configuration of the full-size step (plan table, row /
    column records, skipped heads + eps_scan_refine) runs on this small graph as well, against the same full list."""
    from eps_amd import scan, synth
    from eps_amd.heuristics import node_weight_table
    if heads:
        monkeypatch.setattr(scan, "HEAD_MIN_PATHS", 0)
    ran_heads = 0
    g = synth.rmat_graph(12, 12, 3, "cpu")
    A = g.to_scipy()
    g = g.to(dev)
    wt = node_weight_table(g, eps.ops.W_AA)
    _, cu, cv, _, sc = eps.ops.expand_candidates(g.rowptr, g.col, None, wt, g.n_rows, 0, g.n_rows, want_cn=False)
    order = torch.sort(sc, descending=True, stable=True).indices           # candidate order is column-major already
    small_set = scan.SMALL_SET
    # as labelled, then under the hubs-first relabelling large graphs are scanned with (ids mapped back, same tie rule)
    for relabel_min in (scan.RELABEL_MIN_NODES, 0):
        monkeypatch.setattr(scan, "RELABEL_MIN_NODES", relabel_min)
        monkeypatch.setattr(scan, "SMALL_SET", small_set)
        monkeypatch.setattr(scan, "SAMPLE_STRIDE", 256)
        monkeypatch.setattr(scan, "SAFETY", 2.0)
        assert (scan.scan_graph(g, build=relabel_min == 0)[1] is None) == (relabel_min > 0)
        for k in (1, 1000, 77777, int(sc.numel()) + 5):
            kk = min(k, sc.numel())
            want_pairs = torch.stack([cu[order[:kk]], cv[order[:kk]]]).long()
            want_sc = sc[order[:kk]]
            st = {}
            pairs, scores = scan.scan_topk(g, wt, k, stats=st)
            assert torch.equal(pairs, want_pairs) and torch.equal(scores, want_sc)
            assert st["candidates"] == sc.numel()
        # force the estimate path (sample -> bar -> verify) with strides that make the estimate poor in both directions
        monkeypatch.setattr(scan, "SMALL_SET", 0)
        for stride, safety in ((7, 3.0), (64, 3.0), (3, 0.02), (5, 500.0)):
            monkeypatch.setattr(scan, "SAMPLE_STRIDE", stride)
            monkeypatch.setattr(scan, "SAFETY", safety)
            k = 20000
            st = {}
            pairs, scores = scan.scan_topk(g, wt, k, stats=st)
            assert torch.equal(pairs, torch.stack([cu[order[:k]], cv[order[:k]]]).long()), (stride, safety, st)
            assert torch.equal(scores, sc[order[:k]])
            ran_heads += int(bool(st["heads"]))
    assert (ran_heads > 0) == heads, "skipped heads are expected exactly when HEAD_MIN_PATHS lets this graph have them"
    # the oracle's float32 scores agree within the gate on the selected rows
    rp, col = A.indptr.astype(np.int64), A.indices.astype(np.int32)
    w = oracle.node_weights(oracle.col_sums(rp, col, None, g.n_rows), oracle.W_AA)
    _, _, ws = oracle.pair_scores(rp, col, None, w, pairs[0].cpu().numpy(), pairs[1].cpu().numpy())
    assert rel_err(scores.cpu().numpy(), ws) <= 1e-5


def test_scan_topk_overflow_and_ties(eps, dev, monkeypatch):
    """The correction branches of scan_topk, forced: a survivor list that is far too small for the bar (slots > capacity ->
    the bar is raised to just below the k2-th best of what was kept and the list grows) and heavily TIED scores (common-
    neighbour counts: one integer level holds far more pairs than k) -- the result is still the first k rows of the declared
    order; and two weight tables on ONE graph never share a cached fixed-point table (the cache is keyed on the tensor)."""
    from eps_amd import scan, synth
    g = synth.rmat_graph(14, 12, 3, dev)           # 15.4 M unordered candidates: more than the list's chunk slack alone holds
    ones = torch.ones(g.n_rows, dtype=torch.float32, device=dev)
    from eps_amd.heuristics import node_weight_table
    aa = node_weight_table(g, eps.ops.W_AA)

    full = {}

    def want(wt, k):
        if id(wt) not in full:
            _, cu, cv, _, sc = eps.ops.expand_candidates(g.rowptr, g.col, None, wt, g.n_rows, 0, g.n_rows, want_cn=False)
            o = torch.sort(sc, descending=True, stable=True).indices[:30000]
            full[id(wt)] = (torch.stack([cu[o], cv[o]]).long(), sc[o])
        return full[id(wt)][0][:, :k], full[id(wt)][1][:k]

    monkeypatch.setattr(scan, "SMALL_SET", 0)
    real = scan.estimate_bar
    calls = []

    def low_bar(*a, **kw):                       # an estimate that is far too low: (nearly) everything survives
        calls.append(1)
        return torch.full((1,), 1e-30, device=dev)
    for wt in (ones, aa):
        for k in (3000, 20001):
            monkeypatch.setattr(scan, "estimate_bar", low_bar)
            monkeypatch.setattr(scan, "SAFETY", 0.05)                      # ... and a list sized for a twentieth of k
            st = {}
            pairs, scores = scan.scan_topk(g, wt, k, stats=st)
            assert st["launches"] >= 3, "the overflow branch did not run"
            wp, ws = want(wt, k)
            assert torch.equal(pairs, wp) and torch.equal(scores, ws)
            monkeypatch.setattr(scan, "estimate_bar", real)
            monkeypatch.setattr(scan, "SAFETY", 2.0)
            pairs, scores = scan.scan_topk(g, wt, k)                     # tied levels through the regular path
            assert torch.equal(pairs, wp) and torch.equal(scores, ws)
    # alternating weight tables (fresh temporaries of the same size: the allocator recycles their addresses)
    for i in range(4):
        wt = (ones * 1.0) if i % 2 == 0 else (aa * 1.0)
        pairs, scores = scan.scan_topk(g, wt, 5000)
        wp, ws = want(ones if i % 2 == 0 else aa, 5000)
        assert torch.equal(pairs, wp) and torch.equal(scores, ws)
        del wt


def test_scan_full_size_properties(eps, dev, oracle):
    """ppa-sized graph (BASELINE configs[2]): the scan over all columns finds exactly the candidates above the bar that
    the fused expansion scores above it on a block of columns, symmetric survivors mirror, and a re-run is
    bit-identical (order-independent fixed-point sums)."""
    from eps_amd import candidates, scan, synth
    from eps_amd.heuristics import node_weight_table
    g = synth.ppa_like(seed=3, device=dev)
    wt = node_weight_table(g, eps.ops.W_AA)
    st = {}
    pairs, scores = scan.scan_topk(g, wt, 1_000_000, stats=st)
    assert pairs.shape == (2, 1_000_000) and bool((scores[:-1] >= scores[1:]).all())
    assert st["candidates"] > 1e9
    bar = float(scores[-1])
    # block check against the fused expansion: columns [lo, hi) -- every candidate above the bar must be in the top list
    lo, hi = 1000, 1400
    blk = candidates.expand_block_lazy(g, lo, hi, wt, want_score=True)
    idx = torch.nonzero(blk.score > bar).squeeze(1)
    want = blk.select(idx)                                                     # (u; v), column-major
    in_blk = (pairs[1] >= lo) & (pairs[1] < hi)
    got = pairs[:, in_blk]
    gk = torch.sort(got[1] * g.n_rows + got[0]).values
    wk = torch.sort(want[1] * g.n_rows + want[0]).values
    assert torch.equal(gk, wk)
    # mirrored rows carry equal scores
    key = pairs[1] * g.n_rows + pairs[0]
    mkey = pairs[0] * g.n_rows + pairs[1]
    pos = torch.searchsorted(torch.sort(key).values, mkey)
    skey, sidx = torch.sort(key)
    inside = pos < key.numel()
    hit = torch.zeros_like(inside)
    hit[inside] = skey[pos[inside]] == mkey[inside]
    assert bool((scores[hit] == scores[sidx[pos[hit]]]).all())
    assert int(hit.sum()) >= pairs.shape[1] - 2                                 # at most the K-th tie loses its mirror
    p2, s2 = scan.scan_topk(g, wt, 1_000_000)
    assert torch.equal(p2, pairs) and torch.equal(s2, scores)
    # the same scan under hubs-first labels (what a repeatedly scanned graph runs on): the PRODUCTION configuration -- plan table,
    # row / column records, skipped heads + refine -- faces the oracle's candidate set and scores itself, not only
    # through equality with the as-labelled run: twice, because a graph's second scan widens the hub table (HUB_FIRST -> HUB_MAX)
    checked = [(pairs, scores, bar)]
    for _ in range(2):
        st3 = {}
        p3, s3 = scan.scan_topk(g, wt, 1_000_000, relabel=True, stats=st3)
        assert scan.scan_graph(g)[1] is not None
        assert st3["heads"], "the relabelled full-size scan is expected to run with skipped heads"
        assert st3["candidates"] == st["candidates"]
        checked.append((p3, s3, float(s3[-1])))
        assert torch.equal(p3, pairs) and torch.equal(s3, scores)
    _full_size_oracle_columns(eps, g, wt, checked)


def _full_size_oracle_columns(eps, g, wt, results):
    """Full-size scans against the ORACLE directly (not against the build's own expansion kernel): for a dozen columns --
    hubs, median-degree, tail -- the restated filter.py:96-109 candidate set (oracle.candidates_scipy_columns) scored by the
    oracle's pair_scores; each result's rows of those columns ((pairs, scores, bar) triples: the as-labelled run and the
    hubs-first runs with skipped heads) must be exactly the oracle's candidates above the bar (ids exact, scores <= 1e-5
    relative).  Rows whose oracle score is within 1e-5 of the bar may fall on either side."""
    from oracle import eps_oracle as orc
    deg = g.degree()
    by_deg = torch.argsort(deg, descending=True)
    n = g.n_rows
    cols = sorted({int(by_deg[i]) for i in (0, 3, 50, 1000, n // 4, n // 2, n // 2 + 1, 3 * n // 4, n - 1000, n - 2)} | {7, n - 1})
    A = g.to_scipy()
    rp, col = A.indptr.astype(np.int64), A.indices.astype(np.int32)
    w = orc.node_weights(orc.col_sums(rp, col, None, n), orc.W_AA)
    truth = {}
    for c in cols:
        cand, _ = orc.candidates_scipy_columns(A, c, c + 1)
        _, _, sc = orc.pair_scores(rp, col, None, w, cand[:, 0], cand[:, 1])
        truth[c] = (cand, sc)
    for pairs, scores, bar in results:
        pu, pv, ps = pairs[0].cpu().numpy(), pairs[1].cpu().numpy(), scores.cpu().numpy()
        for c in cols:
            cand, sc = truth[c]
            sure = sc > bar * (1 + 2e-5)
            maybe = sc > bar * (1 - 2e-5)
            m = pv == c
            got = dict(zip(pu[m].tolist(), ps[m].tolist()))
            want_sure = dict(zip(cand[sure, 0].tolist(), sc[sure].tolist()))
            want_maybe = set(cand[maybe, 0].tolist())
            assert set(want_sure) <= set(got) <= want_maybe, f"column {c}: rows differ from the oracle's"
            for u, s_ in want_sure.items():
                assert abs(got[u] - s_) <= 1e-5 * max(abs(s_), abs(got[u])), (c, u, got[u], s_)


def test_scan_wide_id_space_in_windows(eps, oracle, dev):
    """An id space wider than the LDS bitmap (N = 1.3 M: three id windows): the windowed scan finds, for sampled columns,
    exactly the fused expansion's candidates (u < v) with bit-identical scores; its candidate count equals the oracle's
    for those columns; scan_topk over the whole graph equals the top of an explicit full scoring of those columns' rows."""
    from eps_amd import scan, synth
    from eps_amd.heuristics import node_weight_table
    n = 1_300_000
    g = synth.rmat_graph(21, 1, 13, "cpu", n_nodes=n)
    A = g.to_scipy()
    g = g.to(dev)
    win_ids, n_win = eps.ops.filter_scan_windows(n)
    assert n_win >= 2 and win_ids * n_win >= n
    wt = node_weight_table(g, eps.ops.W_AA)
    deg = g.degree()
    hub = int(torch.argmax(deg))
    cols = sorted({5, hub, win_ids - 1, win_ids, win_ids + 7, 2 * win_ids + 3 if 2 * win_ids + 3 < n else n - 2, n - 1,
                   int(torch.argsort(deg, descending=True)[3])})
    colt = torch.tensor(cols, dtype=torch.int32, device=dev)
    got, n_cand = _scan_all(eps, g, wt, columns=colt)
    want = {}
    for c in cols:
        _, cu, cv, _, sc = eps.ops.expand_candidates(g.rowptr, g.col, None, wt, n, c, c + 1, want_cn=False)
        m = cu < c
        for u, s in zip(cu[m].cpu().numpy(), sc[m].cpu().numpy()):
            want[(int(u), c)] = float(s)
        ref, _ = oracle.candidates_scipy_columns(A, c, c + 1)
        assert int((ref[:, 0] < c).sum()) == int(m.sum())
    assert n_cand == len(want) and got == want
    assert any(u >= win_ids for u, _ in want) and any(u < win_ids for u, _ in want)
    # whole graph: the exact top-K through the estimate path == the K best of a no-bar scan (compared on the device)
    fixw = scan.fixed_weights(g, wt)
    res = eps.ops.Survivors(2 * int(scan.half_paths(g).sum().item()) + 8192 * 300, float("-inf"), dev)
    eps.ops.filter_scan(g.rowptr, g.col, scan.reverse_positions(g), fixw, n, scan.column_order(g), res, scan.max_degree(g),
                        scan.window_splits(g))
    keys, vals = res.valid(res.counts()[0])
    keys = torch.cat([keys, ((keys & 0xFFFFFFFF) << 32) | (keys >> 32)])
    vals = torch.cat([vals, vals])
    o = torch.argsort(keys)
    keys, vals = keys[o], vals[o]
    k = 5000
    o = torch.sort(vals, descending=True, stable=True).indices[:k]
    pairs, scores = scan.scan_topk(g, wt, k)
    assert torch.equal((pairs[1] << 32) | pairs[0], keys[o]) and torch.equal(scores, vals[o])


def _weighted_graphs(eps, dev):
    """(name, graph) with stored values: the collab-like golden graph (integer multi-edge weights) and a weighted R-MAT."""
    import scipy.sparse as ssp
    from eps_amd import synth
    d = np.load([p for p in golden_pair_files() if "collab_like" in p][0])
    n = len(d["rowptr"]) - 1
    A = ssp.csr_matrix((d["val"], d["col"], d["rowptr"]), shape=(n, n))
    yield "collab_like", eps.CSRGraph.from_scipy(A, device=dev)
    ei = synth.rmat_edges(12, 14 << 12, 8, dev)
    ei = ei[:, ei[0] != ei[1]]
    w = torch.randint(1, 6, (ei.shape[1],), generator=torch.Generator(device=dev).manual_seed(2), device=dev).to(torch.float32)
    yield "rmat_weighted", eps.CSRGraph.from_edge_index(ei, w, sparse_sizes=(1 << 12, 1 << 12)).to_symmetric()


def test_scan_weighted_symmetric_graphs(eps, oracle, dev, monkeypatch):
    """The threshold scan on adjacencies WITH stored values (collab keeps its summed multi-edge weights: rank.py:32-35; the
    published collab recipe is an AA filter: submit_job.py:199-205): scan_topk == the first K rows of the declared order over
    the fused expansion's full weighted list (same fixed-point terms: bit-identical), on the no-bar path and through the
    estimated bar; scores within 1e-5 of the oracle's float32 AA; common-neighbour weights (models.py:536-542) exact."""
    from eps_amd import scan
    from eps_amd.heuristics import node_weight_table
    for name, g in _weighted_graphs(eps, dev):
        assert g.val is not None and scan.scan_available(g), name
        A = g.to_scipy()
        rp, col, val = A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data.astype(np.float32)
        for mode in ("aa", "cn"):
            wt = node_weight_table(g, eps.ops.W_AA) if mode == "aa" else torch.ones(g.n_rows, dtype=torch.float32, device=dev)
            assert scan.scan_usable(g, wt)
            _, cu, cv, _, sc = eps.ops.expand_candidates(g.rowptr, g.col, g.val, wt, g.n_rows, 0, g.n_rows, want_cn=False)
            order = torch.sort(sc, descending=True, stable=True).indices
            for small_set in (scan.SMALL_SET, 0):
                monkeypatch.setattr(scan, "SMALL_SET", small_set)
                for k in (1, 777, 20000):
                    kk = min(k, sc.numel())
                    st = {}
                    pairs, scores = scan.scan_topk(g, wt, k, stats=st)
                    assert torch.equal(pairs, torch.stack([cu[order[:kk]], cv[order[:kk]]]).long()), (name, mode, small_set, k)
                    assert torch.equal(scores, sc[order[:kk]]) and st["candidates"] == sc.numel()
            pu, pv = pairs[0].cpu().numpy(), pairs[1].cpu().numpy()
            w_o = oracle.node_weights(oracle.col_sums(rp, col, val, g.n_rows), oracle.W_AA) if mode == "aa" else np.ones(g.n_rows, np.float32)
            _, cn_o, ws_o = oracle.pair_scores(rp, col, val, w_o, pu, pv)
            assert rel_err(scores.cpu().numpy(), ws_o) <= 1e-5
            if mode == "cn":
                assert np.array_equal(scores.cpu().numpy(), cn_o)


def _structured_graphs():
    import scipy.sparse as ssp
    rng = np.random.default_rng(17)

    def sym(n, r, c):
        A = ssp.coo_matrix((np.ones(len(r), dtype=np.float32), (r, c)), shape=(n, n)).tocsr()
        A = ((A + A.T) > 0).astype(np.float32).tocsr()
        A.setdiag(0)
        A.eliminate_zeros()
        A.sort_indices()
        return A
    out = {}
    n = 3001                                          # star: a hub of degree n - 1 (several descriptor rounds), leaf columns
    out["star"] = sym(n, np.zeros(n - 1, dtype=np.int64), np.arange(1, n))
    n = 200                                           # complete graph: no candidate at all
    r, c = np.nonzero(np.ones((n, n)) - np.eye(n))
    out["complete"] = sym(n, r, c)
    n = 1000
    out["path"] = sym(n, np.arange(n - 1), np.arange(1, n))
    n = 130                                           # two cliques joined by one edge, plus isolated nodes
    r, c = np.nonzero(np.ones((60, 60)) - np.eye(60))
    out["two_cliques"] = sym(n, np.concatenate([r, r + 60, [0]]), np.concatenate([c, c + 60, [60]]))
    out["empty"] = sym(50, np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64))
    out["single_edge"] = sym(2, np.array([0]), np.array([1]))
    for n, p in ((31, 0.3), (33, 0.2), (1000, 0.01), (4097, 0.002), (1500, 0.5)):      # the last: > 2^20 records per column
        m = int(n * n * p / 2)
        out[f"er_{n}_{p}"] = sym(n, rng.integers(0, n, m), rng.integers(0, n, m))
    return out


@pytest.mark.parametrize("name", list(_structured_graphs()))
def test_scan_and_unit_lists_on_structured_graphs(eps, dev, name):
    """Edge-case shapes: the threshold scan (no bar, all columns) reports exactly the u < v half of the list the unit list
    kernels write, with the same score bits; the unit list equals eps_expand_fill's (whose parity with the oracle is
    established in test_gpu_expand.py)."""
    from eps_amd import scan
    from eps_amd.heuristics import node_weight_table
    A = _structured_graphs()[name]
    g = eps.CSRGraph.from_scipy(A, device=dev, keep_values=False)
    n = g.n_rows
    wt = node_weight_table(g, eps.ops.W_AA)
    want = eps.ops.expand_candidates(g.rowptr, g.col, None, wt, n, 0, n, want_cn=False)
    got = eps.ops.expand_unit(g.rowptr, g.col, wt, n, 0, n, scan.max_degree(g), scan.window_splits(g))
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]) and torch.equal(got[2], want[2])
    assert torch.equal(got[4], want[4])
    half = want[1] < want[2]
    full = {(int(u), int(v)): float(s) for u, v, s in zip(want[1][half].cpu(), want[2][half].cpu(), want[4][half].cpu())}
    if not scan.scan_available(g):
        assert g.nnz() == 0 or n < 1
        return
    found, n_cand = _scan_all(eps, g, wt)
    assert n_cand == len(full) and found == full
    # the one-pass kernel, every geometry (a star's hub is a multi-round row, the dense graphs overflow a piece's hash table
    # into partitioned passes): the same survivors, bit-identical scores, with and without a bar
    if scan.max_degree(g) < 1 << 16:
        bar = sorted(full.values())[len(full) // 2] if full else 0.0
        for variant, packed, table in ((2, False, False), (2, True, False), (2, True, True), (2, False, True), (0, False, False),
                                       (0, True, True), (1, True, True)):
            got, nc = _screen_all(eps, g, wt, variant=variant, packed=packed, table=table)
            assert nc == len(full) and got == full, (name, variant, packed, table)
            got, nc = _screen_all(eps, g, wt, thr=bar, variant=variant, packed=packed, table=table)
            assert nc == len(full) and got == {k: x for k, x in full.items() if x > bar}, (name, variant, packed, table)


def test_one_pass_scan_on_a_two_million_node_graph(eps, dev):
    """Ids past 2^20 (21 key bits in a packed piece, windows of 10^5 ids): the one-pass kernel under hubs-first labels -- packed /
    16-bit direct pieces, plan table -- returns the list the two-pass kernel returns on the graph as labelled, and counts the same
    candidates."""
    from eps_amd import scan, synth
    from eps_amd.heuristics import node_weight_table
    g = synth.rmat_graph(21, 4, 5, dev)
    w = node_weight_table(g, eps.ops.W_AA)
    st0, st1 = {}, {}
    p0, s0 = scan.scan_topk(g, w, 300_000, stats=st0)
    assert scan.scan_graph(g)[1] is None                      # as labelled: eps_filter_scan
    p1, s1 = scan.scan_topk(g, w, 300_000, stats=st1, relabel=True)
    gs, perm = scan.scan_graph(g)
    sc = scan.screen_weights(g, gs, perm, w)
    assert perm is not None and scan.one_pass_available(gs) and sc.ssum is not None and sc.plan is not None
    assert st0["candidates"] == st1["candidates"] and torch.equal(p0, p1) and torch.equal(s0, s1)


@pytest.mark.parametrize("n,k", [(0, 5), (1, 1), (777, 100), (200_000, 1), (200_000, 65_432), (3_000_000, 1_000_000), (50_000, 60_000)])
def test_select_compact_one_launch(eps, dev, n, k):
    """eps_select_compact (radix select + threshold + compaction in one launch, grid-wide hand-over between the rounds) against
    torch: k-th largest with heavy ties, -inf / key -1 slots that are no values, a device-side bound on the list, the three
    threshold modes, fewer than k values."""
    import struct
    gen = torch.Generator(device=dev).manual_seed(n + k)
    cap = n + 1000                                               # slots past the device-side count hold garbage that must not count
    vals = (torch.randint(0, 5000, (cap,), generator=gen, device=dev).float() / 7.0 - 100.0)      # heavy ties, both signs
    keys = torch.randint(0, 1 << 40, (cap,), generator=gen, device=dev, dtype=torch.int64)
    dead = torch.rand(cap, generator=gen, device=dev) < 0.2
    vals[dead] = float("-inf")
    keys[dead] = -1
    count = torch.tensor([n], dtype=torch.int64, device=dev)
    live_v, live_k = vals[:n][~dead[:n]], keys[:n][~dead[:n]]
    want_kth = float("-inf") if live_v.numel() < k or k == 0 else float(torch.sort(live_v, descending=True).values[k - 1])
    for mode, params in ((0, (0.0, 0.0, 0.0)), (1, (0.0, 0.0, 0.0)), (2, (3.5, 0.9, 4e-6))):
        ok, ov, n_out, kth, thr = eps.ops.select_compact(keys, vals, k, count.data_ptr(), mode=mode, params=params)
        assert float(kth) == want_kth
        if want_kth == float("-inf"):
            want_thr = want_kth
        elif mode == 0:
            want_thr = want_kth
        elif mode == 1:
            want_thr = float(torch.nextafter(torch.tensor(want_kth), torch.tensor(float("-inf"))))
        else:
            f = lambda x: struct.unpack("<f", struct.pack("<f", x))[0]
            low, rel = f(f(want_kth) - f(params[0])), f(f(want_kth) * f(params[1]))
            want_thr = f(max(low, rel) - f(abs(want_kth) * f(params[2])))
        assert float(thr) == pytest.approx(want_thr, rel=1e-6, abs=1e-30) and (mode == 2 or float(thr) == want_thr)
        m = int(n_out)
        keep = live_v >= float(thr)
        assert m == int(keep.sum())
        got = torch.sort(ok[:m]).values
        assert torch.equal(got, torch.sort(live_k[keep]).values)
        o1, o2 = torch.argsort(ok[:m]), torch.argsort(live_k[keep])
        assert torch.equal(ov[:m][o1], live_v[keep][o2]) or live_k[keep].unique().numel() != m     # (duplicate random keys: order free)
    # outputs smaller than the selection: everything is counted, the first `room` entries are stored
    if int(keep.sum()) > 10:
        ok, ov, n_out, _, thr2 = eps.ops.select_compact(keys, vals, k, count.data_ptr(), mode=2, params=(3.5, 0.9, 4e-6), room=10)
        assert int(n_out) == int(keep.sum()) and ok.numel() == 10 and float(thr2) == float(thr)
        assert bool(torch.isin(ok, live_k[keep]).all())
    # selection only, scores alone (the bar estimate's call)
    _, _, _, kth, thr = eps.ops.select_compact(None, vals, k, count.data_ptr(), mode=1, compact=False)
    assert float(kth) == want_kth


def test_window_paths_of_a_column_subset_and_the_lazy_plan(eps, dev, monkeypatch):
    """r06: eps_scan_window_paths_columns fills exactly the rows of the listed columns with what the whole-graph table holds; a step
    under a bar with skipped heads leaves the whole-graph plan table unbuilt (the bar sample plans its own columns inside the
    launch), a launch without heads builds it on demand -- and the rows are the same with LAZY_PLAN on and off."""
    from eps_amd import scan, synth
    from eps_amd.graph import CSRGraph
    from eps_amd.heuristics import node_weight_table
    g0 = synth.rmat_graph(13, 12, 9, dev)
    gs, perm = g0.degree_ordered()[:2]
    bounds, cuts = scan.screen_tables(gs)
    full = eps.ops.scan_window_paths(gs.rowptr, gs.col, scan.reverse_positions(gs), cuts)
    cols = scan.column_order(gs)[3::41].contiguous()
    part = eps.ops.scan_window_paths(gs.rowptr, gs.col, scan.reverse_positions(gs), cuts, columns=cols)
    assert torch.equal(part[cols.long()], full[cols.long()])
    monkeypatch.setattr(scan, "SMALL_SET", 0)
    monkeypatch.setattr(scan, "RELABEL_MIN_NODES", 0)
    monkeypatch.setattr(scan, "HEAD_MIN_PATHS", 0)
    out = {}
    for lazy in (True, False):
        monkeypatch.setattr(scan, "LAZY_PLAN", lazy)
        g = CSRGraph(g0.rowptr, g0.col, None, g0.n_rows, g0.n_cols)
        w = node_weight_table(g, eps.ops.W_AA)
        st = {"count": False}
        out[lazy] = scan.scan_topk(g, w, 20_000, relabel=True, stats=st)
        assert st["heads"]
        sc = scan.screen_weights(g, *scan.scan_graph(g), w)
        assert sc.has_plan and (sc._plan is None) == lazy, "the whole-graph plan is built eagerly only with LAZY_PLAN off"
        st2 = {}
        again = scan.scan_topk(g, w, 20_000, relabel=True, stats=st2)          # (counts the candidates: a launch without heads -> the plan)
        assert sc._plan is not None and st2["candidates"] > 0
        assert torch.equal(again[0], out[lazy][0]) and torch.equal(again[1], out[lazy][1])
    assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1])
