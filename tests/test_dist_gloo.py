"""CPU, world_size 2, gloo: the multi-GPU sharding / all-gather / top-K-merge logic, with the oracle
injected as the scorer (the product path injects the HIP scorers)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pack_keys_cpu(score: torch.Tensor, base: int) -> torch.Tensor:
    """numpy restatement of csrc/topk_keys.hip (test-side only)."""
    f = (score.numpy().astype(np.float32) + np.float32(0.0))
    b = f.view(np.uint32).astype(np.uint64)
    o = np.where(b & 0x80000000, (~b) & 0xFFFFFFFF, b | 0x80000000)
    ids = np.arange(base, base + len(f), dtype=np.uint64)
    key = ((o ^ 0x80000000) << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - ids)
    return torch.from_numpy(key.view(np.int64).copy())


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import eps_amd  # noqa: F401
    from eps_amd import dist as epd, synth
    from oracle import eps_oracle as orc
    r, w, dev = epd.init_from_env(backend="gloo", host_only=True)
    assert (r, w) == (rank, world) and dev.type == "cpu"
    g = synth.rmat_graph(10, 8, 5, "cpu")
    n = g.n_rows
    gen = torch.Generator().manual_seed(7)
    u = torch.randint(0, n, (20000,), generator=gen, dtype=torch.int32)
    v = torch.randint(0, n, (20000,), generator=gen, dtype=torch.int32)
    rp, col = g.rowptr.numpy(), g.col.numpy()
    wts = orc.node_weights(orc.col_sums(rp, col, None, n), orc.W_AA)

    def score(uu, vv):
        return torch.from_numpy(orc.pair_scores(rp, col, None, wts, uu.numpy(), vv.numpy())[2])

    keys, (lo, hi), local = epd.score_pairs_sharded(g.degree(), u, v, score, _pack_keys_cpu, k=500)
    # 1. the merged top-k equals the single-process answer
    full = score(u, v)
    want = torch.topk(_pack_keys_cpu(full, 0), 500, largest=True, sorted=True).values
    assert torch.equal(keys, want)
    # 2. shards are contiguous, disjoint, cover everything, and are work-balanced
    bounds = epd.balanced_bounds(epd.pair_work(g.degree(), u, v), world)
    assert bounds[0] == 0 and bounds[-1] == 20000 and (lo, hi) == (bounds[rank], bounds[rank + 1])
    work = epd.pair_work(g.degree(), u, v)
    share = float(work[lo:hi].sum() / work.sum())
    assert abs(share - 1.0 / world) < 0.02
    # 3. all-gather of a ragged row partition
    x = torch.arange(7 * 3, dtype=torch.float32).view(7, 3)
    b = [0, 4, 7]
    got = epd.all_gather_rows(x[b[rank]:b[rank + 1]].clone(), b)
    assert torch.equal(got, x)
    # 4. row-sharded GNN forward == unsharded (mean-aggregate layer restated with the oracle)
    feats = torch.randn(n, 8, generator=torch.Generator().manual_seed(1))

    def layer(xf, lo_, hi_):
        return torch.from_numpy(orc.spmm_csr(rp, col, None, xf.numpy(), mean=True)[lo_:hi_].copy())

    h = epd.sharded_gnn_forward([layer, layer], feats, n)
    ref = orc.spmm_csr(rp, col, None, orc.spmm_csr(rp, col, None, feats.numpy(), mean=True), mean=True)
    assert np.array_equal(h.numpy(), ref)
    # 5. the collectives of the sharded threshold scan (scan.scan_topk): histogram all-reduce, status all-gather, one padded
    # gather of ragged per-rank lists whose lengths every rank already knows
    hist = torch.arange(256, dtype=torch.int32) * (rank + 1)
    epd.all_reduce_sum_(hist)
    assert torch.equal(hist, torch.arange(256, dtype=torch.int32) * 3)
    st = torch.tensor([10 + rank, 20 + rank, 3 + 4 * rank, 7, 0], dtype=torch.int64)
    table = torch.stack(epd.all_gather_list(st)).tolist()
    assert table == [[10, 20, 3, 7, 0], [11, 21, 7, 7, 0]]
    lens = [t[2] for t in table]
    mine = torch.arange(lens[rank], dtype=torch.int64) + 100 * rank
    buf = torch.cat([mine, torch.full((5,), -7, dtype=torch.int64)])          # (the list sits at the front of a longer buffer)
    got = epd.gather_ragged(buf, lens)
    assert torch.equal(got, torch.cat([torch.arange(3), torch.arange(7) + 100]))
    assert torch.equal(epd.gather_ragged(buf[:0], [0, 0]), buf[:0])
    # 6. r04: the bar vote (minimum of the ranks' estimates; a rank without an estimate does not vote), the one-rank gather of
    # ragged lists, and the final ordering of a sharded step dealt over the ranks by score range -- scan._ordered_rows_distributed
    # with the tensor-op ordering standing in for eps_select_topk_rows: rank 0 ends with exactly the rows one rank orders alone
    from eps_amd import scan
    est = torch.tensor([3.5 - rank], dtype=torch.float32)
    assert float(epd.all_reduce_min_(est.clone())) == 2.5
    one = epd.gather_ragged_to(buf, lens, 1)
    assert (one is None) == (rank != 1) and (rank != 1 or torch.equal(one, got))
    gen2 = torch.Generator().manual_seed(11)
    uu = torch.randint(0, 2000, (30000,), generator=gen2)
    pk = torch.unique(((uu + 1 + torch.randint(0, 2000, (30000,), generator=gen2)) << 32) | uu)
    pv = torch.round(torch.rand(pk.numel(), generator=gen2) * 40) / 8           # levels of tied scores
    k_rows = 2 * pk.numel() - 5
    want_k, want_v = scan.select_topk_torch(pk, pv, k_rows)
    sp = scan.score_splitters(pv, world)
    counts = scan.score_range_counts(pv, sp).tolist()
    lo_s = float(sp[rank]) if rank < world - 1 else float("-inf")
    hi_s = float(sp[rank - 1]) if rank > 0 else float("inf")
    m = (pv >= lo_s) & (pv < hi_s)
    assert int(m.sum()) == counts[rank]
    rk, rv = scan.select_topk_torch(pk[m], pv[m], 2 * counts[rank])
    lens2 = [2 * c for c in counts]
    rows_k, rows_v = epd.gather_ragged_to(rk, lens2, 0), epd.gather_ragged_to(rv, lens2, 0)
    if rank == 0:
        assert torch.equal(rows_k[:k_rows], want_k) and torch.equal(rows_v[:k_rows], want_v)
    else:
        assert rows_k is None and rows_v is None
    torch.save(keys, os.path.join(tmp, f"keys_{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_world2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    k0 = torch.load(tmp_path / "keys_0.pt")
    k1 = torch.load(tmp_path / "keys_1.pt")
    assert torch.equal(k0, k1) and k0.numel() == 500


def test_shard_count_invariance():
    """1 / 2 / 4 / 8 logical shards merge to the identical top-K (no process group needed)."""
    sys.path.insert(0, ROOT)
    import eps_amd  # noqa: F401
    from eps_amd import dist as epd, proposals
    g = torch.Generator().manual_seed(3)
    score = torch.randint(0, 30, (50000,), generator=g).float()      # heavy ties, like CN
    work = torch.rand(50000, generator=g) + 0.1
    ref = None
    for world in (1, 2, 4, 8):
        b = epd.balanced_bounds(work, world)
        lists = [torch.topk(_pack_keys_cpu(score[b[r]:b[r + 1]], b[r]), min(1000, b[r + 1] - b[r])).values
                 for r in range(world)]
        merged = torch.topk(torch.cat(lists), 1000).values
        ref = merged if ref is None else ref
        assert torch.equal(merged, ref)
    ids = (0xFFFFFFFF - (ref.numpy().view(np.uint64) & np.uint64(0xFFFFFFFF))).astype(np.int64)
    want = torch.sort(score, descending=True, stable=True).indices[:1000]
    assert np.array_equal(ids, want.numpy())


def test_streaming_topk_and_ranked_merge_cpu():
    """StreamingTopK over blocks in candidate order == stable descending sort of everything; contiguous shards merged
    in shard order give the same list for 1/2/4/8 shards (no candidate ids carried: works past 2**32 candidates)."""
    sys.path.insert(0, ROOT)
    import eps_amd  # noqa: F401
    from eps_amd import proposals
    g = torch.Generator().manual_seed(9)
    n, k = 40000, 700
    scores = torch.randint(0, 25, (n,), generator=g).float()          # CN-like ties
    pairs = torch.stack([torch.arange(n), torch.arange(n) * 7 % 1000])
    want = torch.sort(scores, descending=True, stable=True).indices[:k]
    for world in (1, 2, 4, 8):
        bounds = [n * r // world for r in range(world + 1)]
        pl, sl = [], []
        for r in range(world):
            top = proposals.StreamingTopK(k)
            for b in range(bounds[r], bounds[r + 1], 1777):            # ragged blocks
                e = min(b + 1777, bounds[r + 1])
                top.push(pairs[:, b:e], scores[b:e])
            p, s = top.result()
            pl.append(p); sl.append(s)
        mp, ms = proposals.merge_ranked_lists(pl, sl, k)
        assert torch.equal(mp[0], want) and torch.equal(ms, scores[want])


def test_streaming_topk_sampled_precut_cpu():
    """Blocks far larger than K are cut at a sampled, verified threshold before sorting: same list as the full stable
    sort, with heavy ties at the cut, int32 pairs in / int64 pairs out, and a degenerate all-equal block."""
    sys.path.insert(0, ROOT)
    import eps_amd  # noqa: F401
    from eps_amd import proposals
    g = torch.Generator().manual_seed(11)
    n, k = 400_000, 1500
    for scores in (torch.randint(0, 12, (n,), generator=g).float(),            # ~33k entries per tie class
                   torch.rand(n, generator=g),
                   torch.ones(n)):
        pairs = torch.stack([torch.arange(n, dtype=torch.int32), (torch.arange(n) * 13 % 977).to(torch.int32)])
        want = torch.sort(scores, descending=True, stable=True).indices[:k]
        top = proposals.StreamingTopK(k)
        assert top._precut(scores, k) is not None or bool((scores == scores[0]).all())
        for b in range(0, n, 150_000):
            top.push(pairs[:, b:b + 150_000], scores[b:b + 150_000])
        p, s = top.result()
        assert p.dtype == torch.int64
        assert torch.equal(p[0], want) and torch.equal(s, scores[want])


def test_streaming_topk_lazy_column_block_cpu():
    """A ColumnBlock (cand_u + colptr, no v array) pushed into StreamingTopK gives the same proposals as the
    materialised [2,E] pairs: v is rebuilt from colptr for the surviving positions only."""
    sys.path.insert(0, ROOT)
    import eps_amd  # noqa: F401
    from eps_amd import candidates, proposals
    g = torch.Generator().manual_seed(5)
    n_cols, v_lo, k = 3000, 17, 900
    counts = torch.randint(0, 120, (n_cols,), generator=g)
    counts[5] = 0; counts[-1] = 0                                            # empty columns inside and at the end
    colptr = torch.zeros(n_cols + 1, dtype=torch.int64); colptr[1:] = torch.cumsum(counts, 0)
    e = int(colptr[-1])
    cand_u = torch.randint(0, 50000, (e,), generator=g, dtype=torch.int32)
    scores = torch.randint(0, 9, (e,), generator=g).float()
    blk = candidates.ColumnBlock(v_lo, colptr, cand_u, None, scores)
    pairs = blk.pairs()
    assert pairs.shape == (2, e) and int(pairs[1].min()) >= v_lo and int(pairs[1].max()) < v_lo + n_cols
    assert torch.equal(torch.bincount(pairs[1] - v_lo, minlength=n_cols), counts)
    idx = torch.tensor([0, 1, e // 2, e - 1])
    assert torch.equal(blk.select(idx), pairs[:, idx])
    a, b = proposals.StreamingTopK(k), proposals.StreamingTopK(k)
    for rep in range(3):                                                     # later pushes run the threshold cut
        a.push(pairs, scores + rep * (rep == 1)); b.push(blk, scores + rep * (rep == 1))
    pa, sa = a.result(); pb, sb = b.result()
    assert torch.equal(pa, pb) and torch.equal(sa, sb)
    # the count-free layout: every column's segment is longer than its candidates, the rest is padding
    slack = torch.randint(0, 7, (n_cols,), generator=g)
    colptr_ub = torch.zeros(n_cols + 1, dtype=torch.int64); colptr_ub[1:] = torch.cumsum(counts + slack, 0)
    e_ub = int(colptr_ub[-1])
    pos = torch.repeat_interleave(colptr_ub[:-1] - colptr[:-1], counts) + torch.arange(e)     # where candidate i lands
    u_pad = torch.full((e_ub,), -1, dtype=torch.int32); u_pad[pos] = cand_u
    s_pad = torch.full((e_ub,), float("-inf")); s_pad[pos] = scores
    padded = candidates.ColumnBlock(v_lo, colptr_ub, u_pad, None, s_pad, counts=counts)
    assert padded.padded and padded.numel() == e and torch.equal(padded.pairs(), pairs)
    for kk in (k, e + 5):                                                    # also K larger than the whole block
        c, d = proposals.StreamingTopK(kk), proposals.StreamingTopK(kk)
        for rep in range(3):
            c.push(pairs, scores + rep * (rep == 1)); d.push(padded, s_pad + rep * (rep == 1))
        pc, sc = c.result(); pd_, sd = d.result()
        assert torch.equal(pc, pd_) and torch.equal(sc, sd)


def _a2a_worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import eps_amd  # noqa: F401
    from eps_amd import dist as epd
    epd.init_from_env(backend="gloo", host_only=True)
    # rank r sends (r + 1) * (q + 1) - (1 if r == q else 0) elements to rank q, each tagged with sender and receiver
    cnt = lambda r, q: (r + 1) * (q + 1) - (1 if r == q else 0)  # noqa: E731
    send = torch.cat([torch.full((cnt(rank, q),), 100 * rank + q, dtype=torch.int64) for q in range(world)])
    got = epd.all_to_all_ragged(send, [cnt(rank, q) for q in range(world)], [cnt(r, rank) for r in range(world)])
    want = torch.cat([torch.full((cnt(r, rank),), 100 * r + rank, dtype=torch.int64) for r in range(world)])
    assert torch.equal(got, want)
    # a rank with nothing to send to anybody
    none = epd.all_to_all_ragged(torch.zeros(0, dtype=torch.int64) if rank == 0 else torch.arange(world, dtype=torch.int64),
                                 [0] * world if rank == 0 else [1] * world, [0 if r == 0 else 1 for r in range(world)])
    assert none.tolist() == [rank] * (world - 1)
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_all_to_all_ragged_gloo(world):
    """dist.all_to_all_ragged (the exchange of the sharded step's selected pairs by score range): per-destination pieces of
    different lengths, empty pieces, a rank that sends nothing -- world 2 and 4 on CPU tensors."""
    mp.spawn(_a2a_worker, args=(world, _free_port()), nprocs=world, join=True)
