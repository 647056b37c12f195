"""GPU parity of the per-graph table builders of csrc/graph_prep.hip (r04) against the tensor-op / general-kernel paths they
replace: the hubs-first relabelled copy (rank.py / filter.py never relabel -- this is the engine's own layout, so the check is
that the copy IS the same graph), the reverse positions of a symmetric pattern, the score bound."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("valued", [False, True])
@pytest.mark.parametrize("scale,ef,seed", [(10, 8, 1), (13, 12, 4), (8, 40, 9)])
def test_relabel_graph_is_the_same_graph(eps, dev, valued, scale, ef, seed):
    from eps_amd import synth
    from eps_amd.graph import CSRGraph, _coalesce
    g = synth.rmat_graph(scale, ef, seed, dev)
    if valued:
        gen = torch.Generator(device=dev).manual_seed(seed)
        g = CSRGraph(g.rowptr, g.col, torch.rand(g.nnz(), generator=gen, device=dev) + 0.5, g.n_rows, g.n_cols)
    gs, perm, inv = g.degree_ordered()
    # the tensor-op construction (a global sort of all stored entries)
    row, col, val = g.coo()
    rp, c, v = _coalesce(inv[row], inv[col], g.values_or_ones() if val is not None else None, g.n_rows, g.n_cols)
    assert torch.equal(gs.rowptr, rp) and torch.equal(gs.col, c)
    assert (gs.val is None) == (v is None) and (v is None or torch.equal(gs.val, v))
    deg = gs.degree()
    assert bool((deg[:-1] >= deg[1:]).all()) and torch.equal(inv[perm], torch.arange(g.n_rows, device=dev))
    # rows ascend strictly (coalesced), and mapping back gives the original rows
    i = int(perm[5])
    back = torch.sort(perm[gs.col[gs.rowptr[5]:gs.rowptr[6]].long()]).values
    assert torch.equal(back, g.col[g.rowptr[i]:g.rowptr[i + 1]].long())


@pytest.mark.parametrize("how", ["searches", "sorted"])
@pytest.mark.parametrize("scale,ef,seed", [(10, 8, 1), (13, 12, 4), (8, 40, 9)])
def test_reverse_positions_symmetric_vs_general(eps, dev, scale, ef, seed, how, monkeypatch):
    """eps_reverse_positions_symmetric (one search per unordered stored pair) and eps_reverse_positions_sorted (r06: no search -- a stable
    sort of the entry indices by column id) against the general table: same positions, half paths, statistics and flag."""
    from eps_amd import synth
    from eps_amd.graph import CSRGraph
    monkeypatch.setattr(eps.ops, "REVPOS_SORT_MIN", 0 if how == "sorted" else 1 << 62)
    g = synth.rmat_graph(scale, ef, seed, dev)
    r0, h0, f0 = eps.ops.reverse_positions(g.rowptr, g.col, with_stats=True)
    r1, h1, f1 = eps.ops.reverse_positions_symmetric(g.rowptr, g.col)
    assert int(f0) == 0 and int(f1[0]) & 0xFFFFFFFF == 0 and torch.equal(r0, r1) and torch.equal(h0, h1)
    assert f1[1:].tolist() == [int(g.degree().max()), int(h0.max()), int(h0.sum())]
    # diagonal entries are their own mirrors
    gd = g.with_self_loops(1.0)
    gd = CSRGraph(gd.rowptr, gd.col, None, gd.n_rows, gd.n_cols)
    r0, h0, f0 = eps.ops.reverse_positions(gd.rowptr, gd.col, with_stats=True)
    r1, h1, f1 = eps.ops.reverse_positions_symmetric(gd.rowptr, gd.col)
    assert int(f0) == 0 and int(f1[0]) & 0xFFFFFFFF == 0 and torch.equal(r0, r1) and torch.equal(h0, h1)
    # an asymmetric pattern is reported, whichever half the odd entry is in
    for drop_upper in (True, False):
        row, col, _ = g.coo()
        pick = torch.nonzero(row < col if drop_upper else row > col)[7]
        keep = torch.ones(row.numel(), dtype=torch.bool, device=dev)
        keep[pick] = False
        ga = CSRGraph.from_edge_index(torch.stack([row[keep], col[keep]]), None, sparse_sizes=(g.n_rows, g.n_cols))
        _, _, fa = eps.ops.reverse_positions_symmetric(ga.rowptr, ga.col)
        assert int(fa[0]) & 0xFFFFFFFF == 1
        from eps_amd import scan
        assert not scan.is_symmetric(ga)
        assert torch.equal(scan.reverse_positions(ga), eps.ops.reverse_positions(ga.rowptr, ga.col))


@pytest.mark.parametrize("valued", [False, True])
def test_score_bound_vs_tensor_ops(eps, dev, valued):
    from eps_amd import candidates, synth
    from eps_amd.graph import CSRGraph
    from eps_amd.heuristics import node_weight_table
    g = synth.rmat_graph(12, 10, 3, dev)
    if valued:
        gen = torch.Generator(device=dev).manual_seed(5)
        vals = torch.rand(g.nnz(), generator=gen, device=dev) * 9 + 0.1
        row, col, _ = g.coo()
        key_lo, key_hi = torch.minimum(row, col), torch.maximum(row, col)
        vals = (torch.sin((key_lo * 7919 + key_hi).double()).abs() * 9 + 0.1).float()      # symmetric values
        g = CSRGraph(g.rowptr, g.col, vals, g.n_rows, g.n_cols)
    w = node_weight_table(g, eps.ops.W_AA)
    got = candidates.fused_score_bound(g, w)
    colidx = g.col.long()
    term = w[colidx].abs().double()
    if valued:
        colmax = torch.zeros(g.n_cols, device=dev).scatter_reduce_(0, colidx, g.val.abs(), reduce="amax", include_self=True)
        term = term * g.val.abs().double() * colmax[colidx].double()
    rows = torch.zeros(g.n_rows, dtype=torch.float64, device=dev).index_add_(0, g.row_index(), term)
    want = float(rows.max())
    assert want <= got <= want * (1 + 1e-6) + 1e-9
    # no weights: the largest degree (x the values)
    got0 = candidates.fused_score_bound(g, None)
    if not valued:
        assert got0 == pytest.approx(float(g.degree().max()), rel=1e-8)


def test_node_order_is_the_stable_descending_argsort(eps, dev):
    from eps_amd import scan, synth
    g = synth.rmat_graph(13, 12, 4, dev)
    deg = g.degree()
    want = torch.argsort(deg, descending=True, stable=True)
    assert torch.equal(eps.ops.node_order(rowptr=g.rowptr).long(), want)
    perm, inv, rp = eps.ops.node_order(rowptr=g.rowptr, relabel=True)
    assert torch.equal(perm, want) and torch.equal(inv.long()[want], torch.arange(g.n_rows, device=dev))
    assert int(rp[0]) == 0 and torch.equal(rp[1:], torch.cumsum(deg[want], 0))
    hp = scan.half_paths(g)                      # heavy ties at 0 and large values
    assert torch.equal(eps.ops.node_order(keys=hp).long(), torch.argsort(hp, descending=True, stable=True))


def test_warm_up_loads_every_code_object(eps, dev):
    """eps_warm_up launches one empty kernel per translation unit on a private stream (0 = every code object loaded); the host
    wrapper runs it from a daemon thread and is idempotent."""
    assert eps.load().eps_warm_up() == 0
    t = eps._lib.warm_up_async(dev)
    assert eps._lib.warm_up_async(dev) is t
    t.join(timeout=120)
    assert not t.is_alive()


def test_round4_entry_points_on_empty_and_tiny_inputs(eps, dev):
    """The r04 entry points on the inputs the reference's data hands over at the small end: a graph without edges, one node,
    one edge; empty survivor lists.  No launch with an empty grid, outputs of the documented shapes."""
    from eps_amd.graph import CSRGraph
    ops = eps.ops
    i64 = dict(dtype=torch.int64, device=dev)
    # a graph of 5 nodes without edges
    rp = torch.zeros(6, **i64)
    col = torch.zeros(0, dtype=torch.int32, device=dev)
    rev, hp, info = ops.reverse_positions_symmetric(rp, col)
    assert rev.numel() == 0 and hp.tolist() == [0] * 5 and info.tolist() == [0, 0, 0, 0]
    perm, inv, nrp = ops.node_order(rowptr=rp, relabel=True)
    assert perm.tolist() == list(range(5)) and inv.tolist() == list(range(5)) and nrp.tolist() == [0] * 6
    c, v = ops.relabel_graph(rp, col, None, perm, inv, nrp)
    assert c.numel() == 0 and v is None
    assert float(ops.score_bound(rp, col, None, torch.ones(5, device=dev), 5, 5)) == 0.0
    assert ops.node_order(keys=torch.zeros(0, **i64)).numel() == 0
    # one node, no edge; two nodes, one edge
    g1 = CSRGraph(torch.zeros(2, **i64), col, None, 1, 1)
    gs, p, q = g1.degree_ordered()
    assert gs.nnz() == 0 and p.tolist() == [0] and q.tolist() == [0]
    g2 = CSRGraph.from_edge_index(torch.tensor([[0, 1], [1, 0]], device=dev), None, sparse_sizes=(2, 2))
    rev, hp, info = ops.reverse_positions_symmetric(g2.rowptr, g2.col)
    # (revpos = the position of v INSIDE row w: both mirrors sit at place 0 of their one-entry rows)
    assert rev.tolist() == [0, 0] and info.tolist()[0] & 0xFFFFFFFF == 0 and info.tolist()[1] == 1
    gs, p, q = g2.degree_ordered()
    assert gs.col.tolist() == [1, 0] and p.tolist() == [0, 1]
    # empty and one-entry survivor lists
    k0, v0 = torch.zeros(0, **i64), torch.zeros(0, device=dev)
    ok, ov, n, kth, thr = ops.select_compact(k0, v0, 3)
    assert int(n) == 0 and float(kth) == float("-inf")
    ok, ov, n = ops.compact_between(k0, v0, None, None)
    assert int(n) == 0
    assert ops.sort_pairs_by_u(k0).numel() == 0
    one = torch.tensor([(7 << 32) | 3], **i64)
    assert ops.sort_pairs_by_u(one, id_bits=4).tolist() == [(3 << 32) | 7]
    ok, ov, n, kth, thr = ops.select_compact(one, torch.tensor([2.5], device=dev), 1)
    assert int(n) == 1 and ok[:1].tolist() == one.tolist() and float(kth) == 2.5 and float(thr) == 2.5
    ok, ov, n, kth, thr = ops.select_compact(one, torch.tensor([2.5], device=dev), 2)     # fewer entries than k: everything
    assert int(n) == 1 and float(kth) == float("-inf")


def test_round5_entry_points_on_empty_and_tiny_inputs(eps, dev):
    """The r05 entry points at the small end: head tables / hub rows / row records of a graph without edges and of one edge, a
    refine pass over an empty walked list, the dense product on one and two nodes, a launch whose whole column list is handed out in
    batches, column records of an empty list."""
    from eps_amd import scan
    from eps_amd.graph import CSRGraph
    ops = eps.ops
    i64 = dict(dtype=torch.int64, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    # a graph of 5 nodes without edges
    rp = torch.zeros(6, **i64)
    col = torch.zeros(0, **i32)
    fx = torch.ones(5, **i32)
    assert ops.scan_heads(rp, col, fx, 5, 100).tolist() == [[0, 0]] * 5
    assert ops.scan_heads(rp, col, fx, 0, 100).tolist() == [[0, 0]] * 5
    hub = ops.scan_hub_rows(rp, col, 5)
    assert tuple(hub.shape) == (5, ops.scan_hub_row_words(5)) and int(hub.abs().sum()) == 0
    assert tuple(ops.scan_hub_rows(rp, col, 0).shape) == (0, ops.scan_hub_row_words(5))
    cuts = ops.scan_cuts(rp, col, ops.scan_bounds(rp, 5))
    rec = ops.scan_row_records(cuts, rp, fx)
    assert tuple(rec.shape) == (5, 32) and rec.data_ptr() % 128 == 0 and rec[:, 17].tolist() == [1] * 5 and int(rec[:, :17].abs().sum()) == 0
    # one edge: heads under a budget that admits the row / does not; the hub row bitmaps hold both directions
    g2 = CSRGraph.from_edge_index(torch.tensor([[0, 1], [1, 0]], device=dev), None, sparse_sizes=(2, 2))
    fx2 = torch.tensor([3, 5], **i32)
    assert ops.scan_heads(g2.rowptr, g2.col, fx2, 2, 10).tolist() == [[1, 5], [1, 3]]
    assert ops.scan_heads(g2.rowptr, g2.col, fx2, 2, 4).tolist() == [[0, 0], [1, 3]]
    assert ops.scan_heads(g2.rowptr, g2.col, fx2, 1, 10).tolist() == [[0, 0], [1, 3]]      # (node 1 is no hub: row 0 keeps its entry)
    assert ops.scan_hub_rows(g2.rowptr, g2.col, 2)[:, 0].tolist() == [2, 1]
    # refine over an empty walked list, and over one slot whose head term decides
    heads = ops.scan_heads(g2.rowptr, g2.col, fx2, 2, 10)
    hub2 = ops.scan_hub_rows(g2.rowptr, g2.col, 2)
    walked = ops.Survivors(16, 0.0, dev, prefill=False)
    out = ops.Survivors(16, 0.0, dev, prefill=False)
    ops.scan_refine(walked, heads, hub2, fx2, g2.rowptr, g2.col, 2, 0, out)
    assert out.counts()[0] == 0
    # (a path graph 0 - 1 - 2: pair (0, 2) has the common neighbour 1; column 2 skips its row 1 -- weight 7 -- and the walked sum 0
    #  plus the head term reaches a bar of 7, stays below one of 9)
    g3 = CSRGraph.from_edge_index(torch.tensor([[0, 1, 1, 2], [1, 0, 2, 1]], device=dev), None, sparse_sizes=(3, 3))
    fx3 = torch.tensor([1, 7, 1], **i32)
    h3 = ops.scan_heads(g3.rowptr, g3.col, fx3, 3, 7)
    assert h3.tolist() == [[1, 7], [2, 2], [1, 7]]
    hub3 = ops.scan_hub_rows(g3.rowptr, g3.col, 3)
    for bar, kept in ((7.0, 1), (9.0, 0)):
        walked = ops.Survivors(16, bar - 0.5, dev, prefill=False)
        walked.key[0] = (2 << 32) | 0
        walked.val.view(torch.int32)[0] = 0
        walked.rec[1] = 1
        out = ops.Survivors(16, bar - 0.5, dev, prefill=False)
        ops.scan_refine(walked, h3, hub3, fx3, g3.rowptr, g3.col, 3, 0, out)
        assert out.counts()[0] == kept
        if kept:
            assert out.key[:1].tolist() == [(2 << 32) | 0] and out.val[:1].tolist() == [7.0]
    # the dense product on one node, on two nodes (an edge: no candidate), on the path (one candidate each way)
    for g, want in ((CSRGraph(torch.zeros(2, **i64), col, None, 1, 1), []), (g2, []), (g3, [(2, 0, 1.0), (0, 2, 1.0)])):
        rows, cnt = ops.dense_cn_candidates(g.rowptr, g.col, g.n_rows, directed=True, check_symmetric=True, as_rows=True)
        assert [tuple(r) for r in rows.tolist()] == want and cnt.tolist() == [w[2] for w in want]
        keys, cnt = ops.dense_cn_candidates(g.rowptr, g.col, g.n_rows)
        assert keys.tolist() == [(v << 32) | u for u, v, _ in want if u < v]
    # column records of an empty list; every column of a list handed out in batches (batch_from = 0), fewer columns than a batch
    assert scan.column_records(g3, None, torch.zeros(0, **i32), (None, None), None, {}, "k") is None
    from eps_amd import synth
    g = synth.rmat_graph(9, 6, 4, dev)
    w = torch.ones(g.n_rows, device=dev)
    sc = scan.screen_weights(g, g, None, w)
    bounds, cuts = scan.screen_tables(g)
    order = scan.column_order(g)
    got = []
    for cols, bf in ((order, 0), (order[:5].contiguous(), 0), (order, None)):
        res = ops.Survivors(scan._capacity(2 * scan.total_half_paths(g), scan._PIECE_SLACK), float("-inf"), dev, prefill=False)
        status = torch.zeros(1, **i32)
        ops.scan_screen(g.rowptr, g.col, scan.reverse_positions(g), sc.fx32, cuts, bounds, g.n_rows, cols, sc.shift, res, status,
                        scan.screen_variant(g), wpaths=scan.window_paths(g), ssum=sc.ssum, smax=sc.smax, plan=sc.plan, batch_from=bf,
                        rowrec=sc.rowrec, colrec=scan.column_records(g, sc, cols, sc.plan, None, {}, "k"))
        assert int(status) == 0
        k, _ = res.valid(res.counts()[0])
        got.append((torch.sort(k).values, res.counts()[1]))
    assert torch.equal(got[0][0], got[2][0]) and got[0][1] == got[2][1] and 0 < got[1][1] <= got[0][1]


@pytest.mark.parametrize("n_u,shift", [(8, 12), (3, 10), (200000, 12), (1, 12)])
def test_sort_pairs_by_u_keeps_blocks_only_while_runs_are_long(eps, dev, n_u, shift):
    """eps_sort_pairs_by_u with a v block: (v >> shift, u, v) when the runs of equal (block, u) average >= 64 pairs, else (u, v)
    -- eps_rescore_runs (filter.py:113-142's scores, exactly, for the K best) pays one bitmap per run; the decision is made on
    the device.  Either order holds the same pairs; both are checked against tensor ops."""
    from eps_amd import ops
    gen = torch.Generator(device=dev).manual_seed(n_u + shift)
    n = 60000
    u = torch.randint(0, n_u, (n,), generator=gen, device=dev)
    v = u + 1 + torch.randint(0, 1 << 16, (n,), generator=gen, device=dev)
    keys = torch.unique((v << 32) | u)                                   # v << 32 | u, u < v, distinct pairs
    keys = keys[torch.randperm(keys.numel(), generator=gen, device=dev)]
    bits = int(v.max()).bit_length()
    out = ops.sort_pairs_by_u(keys, bits, shift)
    uu, vv = keys & 0xFFFFFFFF, keys >> 32
    runs = torch.unique(((vv >> shift) << 32) | uu).numel()
    blocked = runs * 64 <= keys.numel()
    assert blocked == (n_u <= 8)                                         # (the cases straddle the rule)
    major = (vv >> shift) if blocked else torch.zeros_like(vv)
    o = torch.argsort((major << 52) | (uu << 26) | vv)                  # ids below 2^26 here
    assert torch.equal(out, ((uu << 32) | vv)[o])
    # and no block at all is the plain (u, v) order
    o2 = torch.argsort((uu << 26) | vv)
    assert torch.equal(ops.sort_pairs_by_u(keys, bits, 0), ((uu << 32) | vv)[o2])
