"""GPU parity of the scan's SKIPPED HEADS (csrc/scan_heads.hip; still filter.py:96-142 + :160-161 under --keep_top): the head
table and the hub row bitmaps against numpy, the completed list of a launch with heads against the plain launch's (the same
survivors after exact re-scoring, for every budget), and scan_topk with heads against scan_topk without -- bit-identical rows
for AA / RA / CN weights, including the branches that void a launch (a stale head table, a walked list that overflows)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _weights(eps, g, kind):
    from eps_amd.heuristics import node_weight_table
    if kind == "cn":
        return torch.ones(g.n_rows, dtype=torch.float32, device=g.device)
    return node_weight_table(g, {"aa": eps.ops.W_AA, "ra": eps.ops.W_RA}[kind])


def _hubs_first(g):
    return g.degree_ordered()[:2]


def test_head_table_and_hub_rows_match_numpy(eps, dev):
    from eps_amd import scan, synth
    g0 = synth.rmat_graph(12, 10, 5, dev)
    g, perm = _hubs_first(g0)
    sc = scan.screen_weights(g0, g, perm, _weights(eps, g0, "aa"))
    rp, col = g.rowptr.cpu().numpy(), g.col.cpu().numpy()
    fx = sc.fx32.cpu().numpy().view(np.uint32).astype(np.int64)
    n = g.n_rows
    for n_hub, budget in ((n, int(fx.max()) * 3), (100, int(fx.max()) * 50), (7, 1 << 30), (n, 0), (0, 1 << 20)):
        got = eps.ops.scan_heads(g.rowptr, g.col, sc.fx32, n_hub, budget).cpu().numpy().view(np.uint32)
        want = np.zeros((n, 2), np.int64)
        for v in range(n):
            t = x = 0
            for i in range(rp[v], rp[v + 1]):
                w = col[i]
                if w >= n_hub or t + fx[w] > budget:
                    break
                t += fx[w]
                x += 1
            want[v] = (x, t)
        assert np.array_equal(got.astype(np.int64), want), (n_hub, budget)
    for n_hub in (1, 37, min(n, 4096)):
        rows = eps.ops.scan_hub_rows(g.rowptr, g.col, n_hub).cpu().numpy().view(np.uint32)
        assert rows.shape == (n_hub, eps.ops.scan_hub_row_words(n)) and rows.shape[1] % 4 == 0
        want = np.zeros_like(rows)
        for w in range(n_hub):
            x = col[rp[w]:rp[w + 1]].astype(np.int64)
            np.bitwise_or.at(want[w], x >> 5, (np.uint32(1) << (x & 31).astype(np.uint32)))
        assert np.array_equal(rows, want)


def _plain_list(eps, g, sc, order, bar):
    from eps_amd import scan
    bounds, cuts = scan.screen_tables(g)
    res = eps.ops.Survivors(scan._capacity(2 * scan.total_half_paths(g), scan._PIECE_SLACK), bar, g.device, prefill=False)
    status = torch.zeros(1, dtype=torch.int32, device=g.device)
    eps.ops.scan_screen(g.rowptr, g.col, scan.reverse_positions(g), sc.fx32, cuts, bounds, g.n_rows, order, sc.shift, res, status,
                        scan.screen_variant(g), wpaths=scan.window_paths(g), ssum=sc.ssum, smax=sc.smax, plan=sc.plan)
    assert int(status) == 0
    return res


def _exact(g, sc, res, bar):
    from eps_amd import scan
    slots, _ = res.counts()
    assert slots <= res.capacity
    keys, approx = res.valid(slots)
    assert torch.unique(keys).numel() == keys.numel(), "a candidate was reported twice"
    k2, v2 = scan.rescore_exact(g, sc, keys, torch.tensor([bar], device=g.device))
    m = k2 >= 0
    o = torch.argsort(k2[m])
    # (screening scores are upper bounds of the exact ones: what makes the screening lossless)
    by_key = dict(zip(k2[m].tolist()[:3000], v2[m].tolist()[:3000]))
    for kk, aa in zip(keys.tolist(), approx.tolist()):
        if kk in by_key:
            assert aa >= by_key[kk] * (1 - 1e-6) - 1e-6
    return k2[m][o], v2[m][o]


@pytest.mark.parametrize("kind", ["aa", "ra", "cn"])
def test_refined_list_equals_plain_list(eps, dev, kind):
    """One launch with skipped heads + eps_scan_refine reports the survivors a plain launch reports: identical (pair, exact
    score) lists after re-scoring, for small and large budgets and for a hub table narrower than the heads would like."""
    from eps_amd import scan, synth
    g0 = synth.rmat_graph(13, 12, 7, dev)
    g, perm = _hubs_first(g0)
    w = _weights(eps, g0, kind)
    sc = scan.screen_weights(g0, g, perm, w)
    assert sc.usable and sc.plan is not None
    order = scan.column_order(g)
    bounds, cuts = scan.screen_tables(g)
    # a bar that a few thousand pairs exceed
    _, _, _, _, full = eps.ops.expand_candidates(g.rowptr, g.col, None, w[perm].contiguous(), g.n_rows, 0, g.n_rows, want_cn=False)
    bar = float(torch.sort(full, descending=True).values[min(40000, full.numel() - 1)])
    want_k, want_v = _exact(g, sc, _plain_list(eps, g, sc, order, bar), bar)
    assert want_k.numel() > 1000
    units = bar * 2.0 ** sc.shift
    for n_hub, beta in ((min(4096, g.n_rows), 0.25), (min(4096, g.n_rows), 0.5), (min(4096, g.n_rows), 0.97), (64, 0.5), (0, 0.5)):
        hub = eps.ops.scan_hub_rows(g.rowptr, g.col, n_hub)
        heads = eps.ops.scan_heads(g.rowptr, g.col, sc.fx32, n_hub, int(beta * units))
        wp = eps.ops.scan_window_paths(g.rowptr, g.col, scan.reverse_positions(g), cuts, heads)
        plan = eps.ops.scan_plan(g.rowptr, cuts, wp, sc.ssum, sc.smax, bounds, g.n_rows, sc.shift, scan.screen_variant(g), heads=heads)
        walked = eps.ops.Survivors(scan._capacity(2 * scan.total_half_paths(g), scan._PIECE_SLACK), bar, dev, prefill=False)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        eps.ops.scan_screen(g.rowptr, g.col, scan.reverse_positions(g), sc.fx32, cuts, bounds, g.n_rows, order, sc.shift, walked, status,
                            scan.screen_variant(g), wpaths=wp, ssum=sc.ssum, smax=sc.smax, plan=plan, heads=heads)
        assert int(status) == 0
        if n_hub:
            assert int(heads[:, 0].max()) > 0 and int(wp.sum()) < int(scan.window_paths(g).sum()), "no head was skipped"
        res = eps.ops.Survivors(walked.capacity, bar, dev, prefill=False)
        eps.ops.scan_refine(walked, heads, hub, sc.fx32, g.rowptr, g.col, g.n_rows, sc.shift, res)
        got_k, got_v = _exact(g, sc, res, bar)
        assert torch.equal(got_k, want_k) and torch.equal(got_v, want_v), (kind, n_hub, beta)
    # a head table built for a HIGHER bar: the kernel refuses (status bit 2) instead of reporting a list with holes
    heads = eps.ops.scan_heads(g.rowptr, g.col, sc.fx32, min(4096, g.n_rows), int(1.5 * units))
    if int(heads[:, 1].view(torch.int32).max()) >= int(units) + 1:
        wp = eps.ops.scan_window_paths(g.rowptr, g.col, scan.reverse_positions(g), cuts, heads)
        plan = eps.ops.scan_plan(g.rowptr, cuts, wp, sc.ssum, sc.smax, bounds, g.n_rows, sc.shift, scan.screen_variant(g), heads=heads)
        walked = eps.ops.Survivors(1 << 22, bar, dev, prefill=False)
        eps.ops.scan_screen(g.rowptr, g.col, scan.reverse_positions(g), sc.fx32, cuts, bounds, g.n_rows, order, sc.shift, walked, status,
                            scan.screen_variant(g), wpaths=wp, ssum=sc.ssum, smax=sc.smax, plan=plan, heads=heads)
        assert int(status) & 4


@pytest.mark.parametrize("kind", ["aa", "ra", "cn"])
def test_scan_topk_same_rows_with_and_without_heads(eps, dev, monkeypatch, kind):
    from eps_amd import scan, synth
    g = synth.rmat_graph(14, 12, 3, dev)
    w = _weights(eps, g, kind)
    monkeypatch.setattr(scan, "SMALL_SET", 0)                    # (the estimate -> scan -> verify path: heads need a bar)
    monkeypatch.setattr(scan, "HEAD_MIN_PATHS", 0)               # (... and this graph is smaller than the ones heads are used on)
    monkeypatch.setattr(scan, "RELABEL_MIN_NODES", 0)
    for k in (2000, 150_000):
        monkeypatch.setattr(scan, "HEADS", False)
        st0 = {}
        p0, s0 = scan.scan_topk(g, w, k, stats=st0, relabel=True)
        monkeypatch.setattr(scan, "HEADS", True)
        st1 = {}
        p1, s1 = scan.scan_topk(g, w, k, stats=st1, relabel=True)
        assert torch.equal(p0, p1) and torch.equal(s0, s1)
        assert not st0["heads"] and st1["heads"], (st0, st1)
        assert st1["candidates"] == st0["candidates"] and 0 < st1["touched"] <= st1["candidates"]
        st2 = {"count": False}
        p2, s2 = scan.scan_topk(g, w, k, stats=st2, relabel=True)
        # (the walk of this call may have run with another head table than the last one's: `touched` is whatever it reached)
        assert torch.equal(p0, p2) and st2["candidates"] is None and 0 < st2["touched"] <= st1["candidates"]


@pytest.mark.parametrize("kind", ["aa", "cn"])
def test_sketch_pieces_at_the_kernel(eps, dev, kind):
    """One launch with heads, packed pieces hashed vs as sketch pieces (eps_scan_screen's variant bit 16), on a graph whose id space
    is wider than a direct piece: every id is reported once, every reported sum is an upper bound of the exact score (``_exact``
    checks both), and after exact re-scoring the two lists are identical -- at a high and at a low bar.  With a set of ONE reported
    id per piece (bits 17..23) the low bar voids the launch: status bit 3."""
    from eps_amd import scan, synth
    g0 = synth.rmat_graph(17, 10, 3, dev)
    g, perm = _hubs_first(g0)
    w = _weights(eps, g0, kind)
    sc = scan.screen_weights(g0, g, perm, w)
    assert sc.usable and sc.plan is not None
    order = scan.column_order(g)
    bounds, cuts = scan.screen_tables(g)
    variant = scan.screen_variant(g)
    hub = eps.ops.scan_hub_rows(g.rowptr, g.col, min(4096, g.n_rows))
    voids = ran = 0
    for k in (100_000, 6_000_000, 0):
        # (k = 0: a bar far below what any K of interest asks for -- pieces of the tail report several ids each)
        bar = float(scan.scan_topk(g0, w, k)[1][-1]) * 0.97 if k else {"aa": 1.6, "cn": 6.0}[kind]
        units = bar * 2.0 ** sc.shift
        heads = eps.ops.scan_heads(g.rowptr, g.col, sc.fx32, hub.shape[0], int(0.5 * units))
        wp = eps.ops.scan_window_paths(g.rowptr, g.col, scan.reverse_positions(g), cuts, heads)
        plan = eps.ops.scan_plan(g.rowptr, cuts, wp, sc.ssum, sc.smax, bounds, g.n_rows, sc.shift, variant, heads=heads)
        assert int(((plan[1][:, 0].view(torch.int32).to(torch.int64) & 0xFFFFFFFF) >> 30 == 1).sum()) > 100, "no packed pieces in this plan"
        lists = []
        for word in (variant, variant | eps.ops.SCAN_SKETCH, variant | eps.ops.SCAN_SKETCH | 1 << 17):
            walked = eps.ops.Survivors(1 << 26, bar, dev, prefill=False)
            status = torch.zeros(1, dtype=torch.int32, device=dev)
            eps.ops.scan_screen(g.rowptr, g.col, scan.reverse_positions(g), sc.fx32, cuts, bounds, g.n_rows, order, sc.shift, walked, status,
                                word, wpaths=wp, ssum=sc.ssum, smax=sc.smax, plan=plan, heads=heads)
            st = int(status)
            if word & eps.ops.SCAN_SKETCH and st & 8:        # (a piece had more ids to report than its set holds: the launch is void)
                voids += 1
                continue
            # (bit 4 = sketch pieces ran: never without the request; with it, whenever a live column has a packed piece -- under a
            #  high bar the columns of the id space's tail are dead)
            assert st & ~16 == 0 and (word & eps.ops.SCAN_SKETCH or not st & 16), (hex(word), st)
            ran += (st >> 4) & 1
            res = eps.ops.Survivors(walked.capacity, bar, dev, prefill=False)
            eps.ops.scan_refine(walked, heads, hub, sc.fx32, g.rowptr, g.col, g.n_rows, sc.shift, res)
            lists.append(_exact(g, sc, res, bar))
        assert lists[0][0].numel() > 1000
        for got in lists[1:]:
            assert torch.equal(got[0], lists[0][0]) and torch.equal(got[1], lists[0][1]), (kind, k)
    assert ran >= 1, "no launch ran sketch pieces"
    assert voids >= 1, "a set of one id never filled up"


@pytest.mark.parametrize("kind", ["aa", "ra", "cn"])
def test_sketch_pieces_give_the_same_rows_and_fall_back(eps, dev, monkeypatch, kind):
    """r06: in a launch with heads the packed pieces of single-round columns keep no keys (a count-min sketch: upper bounds only).
    scan_topk with and without them returns bit-identical rows (the kernel's status word says that sketch pieces did run); with a
    set of ONE reported id per piece the second survivor of a piece voids the launch (status bit 3) and the call finishes on
    hashed pieces -- same rows again."""
    from eps_amd import scan, synth
    g = synth.rmat_graph(17, 10, 3, dev)                         # (131 K ids: the tail of the id space is wider than a direct piece)
    w = _weights(eps, g, kind)
    monkeypatch.setattr(scan, "SMALL_SET", 0)
    monkeypatch.setattr(scan, "HEAD_MIN_PATHS", 0)
    monkeypatch.setattr(scan, "RELABEL_MIN_NODES", 0)
    monkeypatch.setattr(scan, "SKETCH_MIN_PATHS", 0.0)           # (whatever the bar is next to the heaviest weight: exactness does not depend on it)
    ran = voids = 0
    for k in (2000, 150_000, 3_000_000, -150_000, -3_000_000):
        monkeypatch.setattr(scan, "SKETCH_SET", 1 if k < 0 else 0)
        k = abs(k)
        monkeypatch.setattr(scan, "SKETCH_PIECES", False)
        st0 = {}
        p0, s0 = scan.scan_topk(g, w, k, stats=st0, relabel=True)
        monkeypatch.setattr(scan, "SKETCH_PIECES", True)
        st1 = {}
        p1, s1 = scan.scan_topk(g, w, k, stats=st1, relabel=True)
        assert torch.equal(p0, p1) and torch.equal(s0, s1), (kind, k)
        assert not st0["sketch"] and st0["sketch_void"] == 0 and st1["heads"] == st0["heads"]
        ran += int(st1["sketch"])
        voids += st1["sketch_void"]
        if kind == "cn":            # (uniform weights: the screening sums are the scores -- Screen.exact -- so upper bounds will not do)
            assert not st1["sketch"] and st1["sketch_void"] == 0
        assert not (st1["sketch"] and st1["sketch_void"]), st1
    # (common neighbours: an exact screen; resource allocation: weights whose heaviest would need three bits less of fixed point for
    #  a slot never to wrap -- no sketch pieces for either, the same rows of course)
    assert ran >= 1 or kind != "aa", "no launch ran sketch pieces"
    # (whether the one-slot set fills up depends on where the survivors sit: test_sketch_pieces_at_the_kernel forces it)


def test_scan_topk_voids_a_launch_with_unusable_heads(eps, dev, monkeypatch):
    """(i) the cached head table was built for a much higher bar (a small k first, then a large one): the kernel refuses it,
    scan_topk builds one for the bar at hand and repeats; (ii) the walked list overflows: the list grows, and after the second
    failure the call finishes without heads.  Rows identical to the run without heads every time."""
    from eps_amd import scan, synth
    g = synth.rmat_graph(14, 12, 3, dev)
    w = _weights(eps, g, "aa")
    monkeypatch.setattr(scan, "SMALL_SET", 0)
    monkeypatch.setattr(scan, "HEAD_MIN_PATHS", 0)
    monkeypatch.setattr(scan, "RELABEL_MIN_NODES", 0)
    monkeypatch.setattr(scan, "HEADS", False)
    want = {k: scan.scan_topk(g, w, k, relabel=True) for k in (200, 400_000)}
    monkeypatch.setattr(scan, "HEADS", True)
    gs, perm = scan.scan_graph(g)
    screen = scan.screen_weights(g, gs, perm, w)
    st = {}
    p, s = scan.scan_topk(g, w, 200, stats=st, relabel=True)
    assert torch.equal(p, want[200][0]) and torch.equal(s, want[200][1]) and st["heads"]
    high = screen.head_cur
    assert high is not None
    screen.head_cur = high                                       # (kept even if the budget check dropped it: the stale table)
    st = {}
    p, s = scan.scan_topk(g, w, 400_000, stats=st, relabel=True)
    assert torch.equal(p, want[400_000][0]) and torch.equal(s, want[400_000][1])
    assert st["heads"] and screen.head_cur is not high and len(screen.heads) >= 2, "the stale table was not replaced"
    # (ii) a walked list of a few hundred slots
    monkeypatch.setattr(scan, "HEAD_LIST", 1e-5)
    st = {}
    p, s = scan.scan_topk(g, w, 400_000, stats=st, relabel=True)
    assert torch.equal(p, want[400_000][0]) and torch.equal(s, want[400_000][1]) and not st["heads"]


def test_full_size_heads_properties(eps, dev):
    """ppa-sized graph (BASELINE configs[2]): with heads the main launch walks well under two thirds of the half paths, and the
    proposal rows are bit-identical to the run without heads."""
    from eps_amd import scan, synth
    g = synth.ppa_like(seed=3, device=dev)
    w = _weights(eps, g, "aa")
    st1 = {"count": False}
    p1, s1 = scan.scan_topk(g, w, 4_000_000, stats=st1, relabel=True)
    assert st1["heads"]
    gs, perm = scan.scan_graph(g)
    ht = scan.screen_weights(g, gs, perm, w).head_cur
    walked = int(ht.wpaths.to(torch.int64).bitwise_and(0xFFFFFFFF).sum())
    assert walked < 0.66 * scan.total_half_paths(gs)
    scan.HEADS = False
    try:
        p0, s0 = scan.scan_topk(g, w, 4_000_000, relabel=True)
    finally:
        scan.HEADS = True
    assert torch.equal(p0, p1) and torch.equal(s0, s1)


@pytest.mark.parametrize("w0", [1.0, 0.5, 0.3])
def test_uniform_weights_skip_the_rescoring_with_the_same_rows(eps, dev, monkeypatch, w0):
    """One weight for every node whose screening sums are exact (common neighbours, models.py:536-542: weight 1; 0.5 likewise;
    0.3 is no whole number of screening units): scan_topk then takes the survivors' scores as they are -- the rows and scores
    must be those of the run that re-scores them, bit for bit, with and without skipped heads."""
    from eps_amd import scan, synth
    g = synth.rmat_graph(14, 12, 3, dev)
    monkeypatch.setattr(scan, "SMALL_SET", 0)
    monkeypatch.setattr(scan, "HEAD_MIN_PATHS", 0)
    monkeypatch.setattr(scan, "RELABEL_MIN_NODES", 0)
    for heads in (False, True):
        monkeypatch.setattr(scan, "HEADS", heads)
        for k in (2000, 150_000):
            monkeypatch.setattr(scan, "EXACT_SCREENING", False)
            wa = torch.full((g.n_rows,), w0, dtype=torch.float32, device=dev)        # (tables are cached per weight TENSOR)
            st0 = {}
            p0, s0 = scan.scan_topk(g, wa, k, stats=st0, relabel=True)
            monkeypatch.setattr(scan, "EXACT_SCREENING", True)
            wb = wa.clone()
            st1 = {}
            p1, s1 = scan.scan_topk(g, wb, k, stats=st1, relabel=True)
            assert torch.equal(p0, p1) and torch.equal(s0, s1)
            gs, perm = scan.scan_graph(g)
            exact = scan.screen_weights(g, gs, perm, wb).exact
            assert exact == (w0 in (1.0, 0.5)) and not scan.screen_weights(g, gs, perm, wa).exact
            assert st0["rescored"] > 0 and (st1["rescored"] == 0) == exact and st1["heads"] == heads


@pytest.mark.parametrize("seed", range(6))
def test_scan_topk_with_heads_on_odd_graph_shapes(eps, dev, monkeypatch, seed):
    """Skipped heads far from the graph they were sized on: a few hubs over a sparse random graph, a clique glued to a star, rows
    that are ALL hubs (n_hub >= N), isolated nodes -- rows and scores of scan_topk with heads equal those without, for AA / RA / CN
    in turn, and the launch did use heads (filter.py:96-142 + :160-161 under --keep_top)."""
    import scipy.sparse as ssp
    from eps_amd import scan
    rng = np.random.default_rng(900 + seed)
    n = int(rng.integers(600, 5000))
    m = int(n * rng.uniform(3, 10))
    r, c = rng.integers(0, n, m), rng.integers(0, n, m)
    hubs = rng.integers(0, n, max(2, n // 150))
    r = np.concatenate([r, np.repeat(hubs, n // 3)]); c = np.concatenate([c, rng.integers(0, n, len(hubs) * (n // 3))])
    if seed % 2:                                                       # a clique of 40 glued to the first hub, and a tail of isolated ids
        q = rng.choice(n - 50, 40, replace=False)
        qa, qb = np.meshgrid(q, q)
        r = np.concatenate([r, qa.ravel(), np.full(40, hubs[0])]); c = np.concatenate([c, qb.ravel(), q])
        keep = (r < n - 50) & (c < n - 50)
        r, c = r[keep], c[keep]
    A = ssp.coo_matrix((np.ones(len(r), dtype=np.float32), (r, c)), shape=(n, n)).tocsr()
    A = ((A + A.T) > 0).astype(np.float32).tocsr()
    A.setdiag(0); A.eliminate_zeros(); A.sort_indices()
    g = eps.CSRGraph.from_scipy(A, device=dev, keep_values=False)
    w = _weights(eps, g, ("aa", "ra", "cn")[seed % 3])
    monkeypatch.setattr(scan, "SMALL_SET", 0)
    monkeypatch.setattr(scan, "HEAD_MIN_PATHS", 0)
    monkeypatch.setattr(scan, "RELABEL_MIN_NODES", 0)
    used = 0
    for k in (500, 20_000):
        monkeypatch.setattr(scan, "HEADS", False)
        p0, s0 = scan.scan_topk(g, w, k, relabel=True)
        monkeypatch.setattr(scan, "HEADS", True)
        st = {"count": False}
        p1, s1 = scan.scan_topk(g, w, k, stats=st, relabel=True)
        assert torch.equal(p0, p1) and torch.equal(s0, s1), (seed, k, st)
        used += bool(st["heads"])
    assert used, "no launch of this graph ran with skipped heads"
