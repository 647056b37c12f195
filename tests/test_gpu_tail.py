"""GPU parity of the step's device-side tail (csrc/tail_sort.hip; filter.py:160-165 for the K rows rank.py:294 reads): the
score-bucket selections against torch's exact order statistics, the two cooperative radix sorts against the r04/r05 library
path (eps_sort_pairs_by_u / eps_select_topk_rows: rocPRIM sorts, themselves checked against torch.sort elsewhere) and against
torch.sort directly, device-side counts that are shorter than the arrays, heavy ties, and scan_topk with the device tail
against scan_topk without it -- bit-identical rows."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _state_is_clean(eps, dev):
    torch.cuda.synchronize()
    return int(eps.ops.tail_state(dev).abs().sum().item()) == 0


@pytest.mark.parametrize("n,k,dist", [(1, 1, "u"), (777, 100, "u"), (200_000, 1, "u"), (200_000, 65_432, "exp"), (3_000_000, 1_000_000, "exp"),
                                      (50_000, 60_000, "u"), (400_000, 150_000, "int")])
def test_score_selection_matches_torch(eps, dev, n, k, dist):
    """kth <= the exact k-th best live value, above it by less than one bucket; the compaction holds exactly the live entries at
    or above thr; counts beyond the room are counted, not stored; the state is left zeroed."""
    g = torch.Generator().manual_seed(n + k)
    base = torch.tensor([2.5], device=dev)
    if dist == "u":
        vals = 2.5 + 10 * torch.rand(n, generator=g)
    elif dist == "exp":
        vals = 2.5 + torch.empty(n).exponential_(1.0, generator=g)
    else:
        vals = torch.randint(3, 40, (n,), generator=g).float()
    keys = torch.randint(0, 1 << 40, (n,), generator=g)
    dead = torch.rand(n, generator=g) < 0.1
    keys[dead] = -1
    vals[dead] = float("-inf")
    vals, keys = vals.to(dev), keys.to(dev)
    n_dev = torch.tensor([max(1, n - n // 7)], dtype=torch.int64, device=dev)
    m = int(n_dev)
    for above in (None, torch.tensor([4.0], device=dev)):
        live = (keys[:m] >= 0) & (vals[:m] > float("-inf"))
        if above is not None:
            live &= vals[:m] > above
        lv = vals[:m][live]
        eps.ops.score_hist(keys, vals, n_dev, base, above)
        ok, ov, n_out, kth, thr = eps.ops.score_pick_compact(keys, vals, n_dev, base, k, above=above)
        cnt = int(n_out)
        if lv.numel() < k:
            assert float(kth) == float("-inf") and cnt == lv.numel()
        else:
            exact = float(torch.sort(lv, descending=True).values[k - 1])
            got = float(kth)
            assert got <= exact
            # one bucket: 2^-8 of the distance to the base in steps of the float bit pattern
            d = int(np.float32(exact).view(np.int32)) - int(np.float32(2.5).view(np.int32))
            lo = np.int32(int(np.float32(exact).view(np.int32)) - max(1, d >> 8) - 1).view(np.float32)
            assert got >= float(lo) or got == float("-inf") and d < 256, (got, exact)
            assert cnt == int((lv >= got).sum()) and cnt >= k
        sel = torch.sort(ok[:cnt]).values
        want = torch.sort(keys[:m][live & (vals[:m] >= thr)]).values
        assert torch.equal(sel, want)
        assert _state_is_clean(eps, dev)
    # a room that is too small: counted, not stored; nothing written past the room
    eps.ops.score_hist(keys, vals, None, base)
    ok, ov, n_out, kth, thr = eps.ops.score_pick_compact(keys, vals, None, base, 0, room=5, swap_halves=True)
    live = (keys >= 0) & (vals > float("-inf"))
    assert int(n_out) == int(live.sum()) and ok.numel() == min(5, n)
    got = ok[:min(5, int(n_out))]
    back = ((got & 0xFFFFFFFF) << 32) | ((got >> 32) & 0xFFFFFFFF)
    assert bool(torch.isin(back, keys[live]).all())
    assert _state_is_clean(eps, dev)


@pytest.mark.parametrize("n,id_bits,shift", [(1, 5, 0), (1000, 10, 0), (70_000, 17, 9), (2_300_000, 20, 12), (300_000, 20, 12), (500_000, 31, 0)])
def test_radix_sort_by_u_matches_library_sort(eps, dev, n, id_bits, shift):
    g = torch.Generator().manual_seed(n)
    hi = (1 << id_bits) - 1
    # few distinct u (long runs: the blocked order is kept) or many (it is dropped): both decisions must agree with the library's
    for distinct_u in (37, max(2, n // 2)):
        u = torch.randint(0, min(hi, distinct_u), (n,), generator=g)
        v = torch.randint(0, hi, (n,), generator=g) + 1
        keys = ((v << 32) | u).to(dev)
        keys = torch.unique(keys)                      # (pairs are unique in a survivor list; equal keys would tie arbitrarily)
        m = keys.numel()
        keys = keys[torch.randperm(m, generator=g).to(dev)]
        want = eps.ops.sort_pairs_by_u(keys, id_bits, shift)
        n_dev = torch.tensor([m], dtype=torch.int64, device=dev)
        got = eps.ops.radix_sort_by_u(keys, n_dev, id_bits, shift)
        if shift == 0:          # (u < 2^31: the swapped key is a non-negative int64 whose order is (u, v unsigned))
            ref = torch.sort(((keys & 0xFFFFFFFF) << 32) | ((keys >> 32) & 0xFFFFFFFF)).values
            assert torch.equal(got, ref), "radix sort differs from torch.sort"
            assert torch.equal(want, ref), "library sort differs from torch.sort"
        assert torch.equal(got, want)
        # a device count shorter than the array
        part = max(1, m - m // 3)
        got = eps.ops.radix_sort_by_u(keys, torch.tensor([part], dtype=torch.int64, device=dev), id_bits, shift)
        assert torch.equal(got[:part], eps.ops.sort_pairs_by_u(keys[:part].contiguous(), id_bits, shift))
        assert _state_is_clean(eps, dev)


@pytest.mark.parametrize("m,nodes,ties", [(1, 10, False), (1000, 300, False), (1000, 300, True), (150_000, 5000, True), (2_000_000, 576_289, False),
                                          (700_000, 1 << 20, True)])
def test_radix_sort_rows_matches_library_and_torch(eps, dev, m, nodes, ties):
    g = torch.Generator().manual_seed(m + nodes)
    id_bits = max(1, int(nodes - 1).bit_length())
    a = torch.randint(0, nodes - 1, (2 * m,), generator=g)
    b = torch.randint(0, nodes - 1, (2 * m,), generator=g)
    lo, hi = torch.minimum(a, b), torch.maximum(a, b) + 1
    keys = torch.unique((hi << 32) | lo)[:m]
    m = keys.numel()
    keys = keys[torch.randperm(m, generator=g)].to(dev)
    vals = (torch.randint(1, 30, (m,), generator=g).float() if ties else 2 + 50 * torch.rand(m, generator=g)).to(dev)
    perm = torch.randperm(nodes, generator=g).to(dev)
    for pm in (None, perm):
        for k in (max(1, m // 3), 2 * m, 2 * m + 5):
            wk, wv = eps.ops.select_rows(keys, vals, k, id_bits, pm)
            pairs, scores, n_rows = eps.ops.radix_sort_rows(keys, vals, torch.tensor([m], dtype=torch.int64, device=dev), k, id_bits, pm)
            take = int(n_rows)
            assert take == min(k, 2 * m) == wk.numel()
            assert torch.equal(pairs[0, :take], wk & 0xFFFFFFFF) and torch.equal(pairs[1, :take], wk >> 32)
            assert torch.equal(scores[:take], wv)
            pp, ps = eps.ops.select_rows_pairs(keys, vals, k, id_bits, pm)       # (the library sorts writing the [2, K] tensor themselves)
            assert pp.shape == (2, take) and torch.equal(pp[0], wk & 0xFFFFFFFF) and torch.equal(pp[1], wk >> 32) and torch.equal(ps, wv)
    # against torch.sort directly (declared order: score descending, then (v, u) ascending), with a device count below the length
    part = max(1, m - m // 4)
    pairs, scores, n_rows = eps.ops.radix_sort_rows(keys, vals, torch.tensor([part], dtype=torch.int64, device=dev), 2 * m, id_bits, None)
    kk, vv = keys[:part], vals[:part]
    rk = torch.cat([kk, ((kk & 0xFFFFFFFF) << 32) | (kk >> 32)])
    rv = torch.cat([vv, vv])
    o = torch.argsort(rk)
    rk, rv = rk[o], rv[o]
    o = torch.sort(rv, descending=True, stable=True).indices
    assert int(n_rows) == 2 * part
    assert torch.equal(pairs[1, :2 * part] << 32 | pairs[0, :2 * part], rk[o]) and torch.equal(scores[:2 * part], rv[o])
    assert _state_is_clean(eps, dev)


def test_rescore_with_a_device_count(eps, dev):
    from eps_amd import scan, synth
    from eps_amd.heuristics import node_weight_table
    g = synth.rmat_graph(13, 12, 4, dev)
    wt = node_weight_table(g, eps.ops.W_AA)
    fixw = scan.fixed_weights(g, wt)
    gen = torch.Generator().manual_seed(1)
    u = torch.randint(0, 200, (50_000,), generator=gen)
    v = torch.randint(200, g.n_rows, (50_000,), generator=gen)
    by_u = torch.sort(torch.unique((u << 32) | v)).values.to(dev)
    want = eps.ops.rescore_runs(g.rowptr, g.col, fixw, g.n_rows, by_u)
    part = by_u.numel() - 1234
    got = eps.ops.rescore_runs_dev(g.rowptr, g.col, fixw, g.n_rows, by_u, torch.tensor([part], dtype=torch.int64, device=dev))
    assert torch.equal(got[:part], want[:part])


@pytest.mark.parametrize("kind", ["aa", "ra", "cn"])
def test_scan_topk_device_tail_is_bit_identical(eps, dev, kind, monkeypatch):
    """scan_topk with the r06 tail -- score-bucket selections with the library sorts ("library", the default) or with the one-launch
    cooperative radix sorts and device-side sizes ("radix") -- == scan_topk on the r05 tail (four-round selects): the same rows and scores, as labelled and under hubs-first labels with skipped heads, incl. a K whose cut level is heavily tied (CN)."""
    from eps_amd import scan, synth
    from eps_amd.heuristics import node_weight_table
    g = synth.rmat_graph(15, 14, 7, dev)
    wt = (torch.ones(g.n_rows, dtype=torch.float32, device=dev) if kind == "cn" else
          node_weight_table(g, {"aa": eps.ops.W_AA, "ra": eps.ops.W_RA}[kind]))
    monkeypatch.setattr(scan, "SMALL_SET", 0)
    monkeypatch.setattr(scan, "RELABEL_MIN_NODES", 0)
    monkeypatch.setattr(scan, "HEAD_MIN_PATHS", 0)
    for relabel in (False, True):
        for k in (1, 5000, 200_001):
            out = {}
            for tail in (None, "library", "radix"):
                monkeypatch.setattr(scan, "TAIL_DEVICE", tail is not None)
                monkeypatch.setattr(scan, "TAIL_SORT", tail or "library")
                st = {}
                pairs, scores = scan.scan_topk(g, wt, k, relabel=relabel, stats=st)
                out[tail] = (pairs, scores, st)
            for tail in ("library", "radix"):
                assert torch.equal(out[tail][0], out[None][0]) and torch.equal(out[tail][1], out[None][1]), (kind, relabel, k, tail)
                assert out[tail][2]["candidates"] == out[None][2]["candidates"]
    assert _state_is_clean(eps, dev)


@pytest.mark.parametrize("world,n,k", [(1, 5000, 100), (2, 100_000, 30_000), (3, 50_000, 200_000), (8, 400_000, 1_000_000), (4, 1000, 0)])
def test_deal_plan_from_gathered_histograms(eps, dev, world, n, k):
    """eps_score_hist_into + eps_score_deal_plan (what every rank of a sharded step computes from the all-gathered histograms):
    the cut is the lower edge of the bucket of the job-wide k-th best score, nsel / counts are exactly what the ranks' lists hold
    at or above the cut per score range, and the ranges are bucket-aligned (equal scores never straddle a splitter)."""
    g = torch.Generator().manual_seed(world * 1000 + n)
    base = torch.tensor([1.75], device=dev)
    bins = eps.ops.score_bins()
    lists = []
    hists = torch.zeros((world, bins + 16), dtype=torch.int32, device=dev)
    for r in range(world):
        m = n // (r + 1)                                                   # uneven shards
        vals = (1.75 + torch.empty(m).exponential_(1.5, generator=g)).to(dev)
        if r % 2:
            vals = torch.floor(vals * 8) / 8                                # tied levels
        vals[torch.rand(m, generator=g).to(dev) < 0.05] = float("-inf")    # dropped slots
        keys = torch.arange(m, device=dev, dtype=torch.int64)
        keys[vals == float("-inf")] = -1
        eps.ops.score_hist_into(keys, vals, None, base, hists[r, 16:], above=base)
        lists.append(vals)
    cut, sp, counts, nsel = eps.ops.score_deal_plan(hists[:, 16:], k, base)
    live = [v[(v > float("-inf")) & (v > base)] for v in lists]
    allv = torch.cat(live)
    cut_f = float(cut)
    if k == 0 or allv.numel() < k:
        assert cut_f == float("-inf")
    else:
        exact = float(torch.sort(allv, descending=True).values[k - 1])
        assert cut_f <= exact and int((allv >= cut_f).sum()) >= k
        d = int(np.float32(exact).view(np.int32)) - int(np.float32(1.75).view(np.int32))
        assert cut_f >= float(np.int32(int(np.float32(exact).view(np.int32)) - max(1, d >> 8) - 1).view(np.float32))
    assert nsel.tolist() == [int((v >= cut_f).sum()) for v in live]
    sp_l = sp.tolist()
    assert all(a >= b for a, b in zip(sp_l, sp_l[1:])), "splitters descend"
    want = torch.zeros((world, world), dtype=torch.int64)
    for r, v in enumerate(live):
        sel = v[v >= cut_f]
        rng = (sel.unsqueeze(1) < sp.unsqueeze(0)).sum(1) if world > 1 else torch.zeros(sel.numel(), dtype=torch.int64, device=dev)
        want[r] = torch.bincount(rng, minlength=world).cpu()
    assert torch.equal(counts.cpu(), want)
    tot = want.sum(0)
    if world > 1 and int(tot.sum()) > 50 * world and not any(r % 2 for r in range(1, world)):
        assert int(tot.max()) <= 2 * int(tot.sum()) // world + 8
