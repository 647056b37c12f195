"""CPU: host logic that needs no GPU -- graph construction (rank.py:28-36), candidate order
(filter.py:96-109), model config table / state-dict keys / factory (models.py:578-790), Hits@K."""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

KEYS = ["num_layers", "hidden_channels", "dropout", "batch_size", "lr", "epochs", "use_feature",
        "use_learnable_embedding"]


def test_default_model_configs_match_reference_table(eps):
    from eps_amd import models
    table = json.load(open(os.path.join(GOLDEN, "model_configs.json")))
    for key, want in table.items():
        if key.startswith("override:"):
            continue
        d, m = key.split("/")
        args = argparse.Namespace(dataset=d, model=m, **{k: None for k in KEYS})
        got = models.default_model_configs(args)
        assert {k: getattr(got, k) for k in KEYS} == want, key
    args = argparse.Namespace(dataset="collab", model="simple", num_layers=5, hidden_channels=32, dropout=0.1,
                              batch_size=77, lr=0.5, epochs=3, use_feature=True, use_learnable_embedding=True)
    got = models.default_model_configs(args)
    assert {k: getattr(got, k) for k in KEYS} == table["override:collab/simple"]


def test_state_dict_keys_match_reference(eps):
    from eps_amd import models
    want = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
    assert list(models.LinkPredictor(4, 4, 1, 2, 0.0).state_dict()) == want["LinkPredictor_L2"]
    assert list(models.LinkPredictor(4, 4, 1, 3, 0.0).state_dict()) == want["LinkPredictor_L3"]
    for kind, cls in (("gcn", models.GCN), ("sage", models.SAGE)):
        d = np.load(os.path.join(GOLDEN, f"gnn_stack_{kind}_L3.npz"))
        ref_keys = [k[4:] for k in d.files if k.startswith("sd::")]
        model = models.LinkGNN(torch.nn.Embedding(40, 16), cls(28, 16, 16, 3, 0.5), models.LinkPredictor(16, 16, 1, 3, 0.5))
        assert sorted(model.state_dict()) == sorted(ref_keys)
        model.load_state_dict({k: torch.from_numpy(d["sd::" + k]) for k in ref_keys})  # shapes agree too


def test_gcnconv_loads_pyg2_checkpoint_layout(eps):
    from eps_amd import models
    conv = models.GCNConv(5, 3)
    w = torch.randn(3, 5)
    conv.load_state_dict({"lin.weight": w, "bias": torch.zeros(3)})
    assert torch.equal(conv.weight.data, w.t())


def test_build_model_factory(eps):
    from eps_amd import models
    data = argparse.Namespace(num_nodes=30, x=torch.randn(30, 7))
    args = models.default_model_configs(argparse.Namespace(dataset="collab", model="gcn", **{k: None for k in KEYS}))
    m = models.build_model(args, data, torch.device("cpu"))
    assert isinstance(m, models.LinkGNN) and m.emb.weight.shape == (30, 256)
    assert m.gnn.convs[0].weight.shape == (256 + 7, 256) and len(m.gnn.convs) == 3 and len(m.linkpred.lins) == 3
    args = models.default_model_configs(argparse.Namespace(dataset="ddi", model="simple", **{k: None for k in KEYS}))
    m = models.build_model(args, data, torch.device("cpu"))
    assert isinstance(m, models.CommonNeighborsPredictor) and m.type == "simple" and m.emb is None
    args = models.default_model_configs(argparse.Namespace(dataset="ppa", model="adamic_ogb", **{k: None for k in KEYS}))
    assert models.build_model(args, data, torch.device("cpu"))(None, None, None) is None     # models.py:534-535
    with pytest.raises(NotImplementedError):
        models.build_model(argparse.Namespace(model="dea", use_learnable_embedding=False, use_feature=False,
                                              hidden_channels=8, num_layers=2, dropout=0.0), data, torch.device("cpu"))


def test_negative_sampling_avoids_edges(eps):
    from eps_amd import training
    g = torch.Generator().manual_seed(0)
    ei = torch.randint(0, 50, (2, 600), generator=g)
    neg = training.negative_sampling(ei, 50, 1000)
    assert neg.shape == (2, 1000)
    edge_keys = set((ei[0] * 50 + ei[1]).tolist())
    assert not (set((neg[0] * 50 + neg[1]).tolist()) & edge_keys)


@pytest.mark.parametrize("dataset", ["ddi", "collab"])
def test_add_edges_matches_restatement(eps, oracle, dataset):
    """rank.py:28-36: duplicates summed by to_symmetric; values reset to 1 unless collab."""
    rng = np.random.default_rng(1)
    n = 50
    ei = rng.integers(0, n, (2, 300))
    ei = np.concatenate([ei, ei[::-1][:, :100]], 1)           # some edges listed in both directions
    w = rng.integers(1, 4, ei.shape[1]).astype(np.float32)
    extra = rng.integers(0, n, (2, 40))
    A = oracle.add_edges_scipy(dataset, ei, w, extra, n)
    g = eps.add_edges(dataset, torch.from_numpy(ei), torch.from_numpy(w), torch.from_numpy(extra), n)
    assert np.array_equal(g.rowptr.numpy(), A.indptr) and np.array_equal(g.col.numpy(), A.indices)
    assert np.array_equal(g.values_or_ones().numpy(), A.data)
    assert (g.val is None) == (dataset != "collab")
    B = g.to_scipy()
    assert (B != B.T).nnz == 0


def test_graph_api_subset(eps):
    g = eps.CSRGraph.from_edge_index(torch.tensor([[0, 0, 1, 3], [1, 1, 2, 3]]), torch.tensor([1., 2., 5., 7.]), (4, 4))
    assert g.nnz() == 3 and g.val.tolist() == [3., 5., 7.]                        # duplicates summed
    row, col, val = g.coo()
    assert row.tolist() == [0, 1, 3] and col.tolist() == [1, 2, 3]
    assert g.sum(-1).tolist() == [3., 5., 0., 7.] and g.sum(0).tolist() == [0., 3., 5., 7.]
    s = g.to_symmetric()
    assert s.to_scipy().toarray()[3, 3] == 14.0 and s.to_scipy().toarray()[1, 0] == 3.0
    l = g.with_self_loops(1.0).to_scipy().toarray()
    assert np.array_equal(np.diag(l), np.ones(4)) and l[0, 1] == 3.0               # diagonal SET, not added
    assert g.fill_value(1.).val is None and g.sparse_sizes() == (4, 4)
    r = eps.CSRGraph.from_scipy(g.to_scipy())
    assert torch.equal(r.rowptr, g.rowptr) and torch.equal(r.val, g.val)


def test_candidate_blocks_match_restatement(eps, oracle):
    from eps_amd import candidates, synth
    g = synth.rmat_graph(9, 6, 2, "cpu")
    want, _ = oracle.candidates_scipy(g.to_scipy())
    got = candidates.all_candidates(g, max_paths=15000)          # forces many column blocks
    assert np.array_equal(got.t().numpy(), want)
    # empty graph / isolated nodes
    e = eps.CSRGraph.from_edge_index(torch.zeros((2, 0), dtype=torch.long), None, (5, 5))
    assert candidates.all_candidates(e).shape == (2, 0)


def test_evaluator_hits_semantics(eps, oracle):
    from eps_amd import evaluate
    ev = evaluate.Evaluator("ogbl-ppa")
    assert ev.K == 100 and evaluate.hits["ppa"] == [10, 100, 200] and evaluate.hits["ddi"] == [10, 20, 30]
    g = torch.Generator().manual_seed(0)
    pos, neg = torch.rand(5000, generator=g), torch.rand(3000, generator=g)
    neg[:50] = neg[50:100]                                       # ties among negatives
    for k in (10, 100, 200):
        ev.K = k
        assert ev.eval({"y_pred_pos": pos, "y_pred_neg": neg})[f"hits@{k}"] == oracle.hits_at_k(pos.numpy(), neg.numpy(), k)
    ev.K = 5000
    assert ev.eval({"y_pred_pos": pos, "y_pred_neg": neg})["hits@5000"] == 1.0
    ev.K = 1
    assert ev.eval({"y_pred_pos": torch.tensor([.5]), "y_pred_neg": torch.tensor([.5, .1])})["hits@1"] == 0.0


def test_get_pos_neg_edges_permutation(eps):
    from eps_amd import evaluate
    split = {"valid": {"edge": torch.arange(20).view(10, 2), "edge_neg": torch.arange(40).view(20, 2)}}
    pos, neg = evaluate.get_pos_neg_edges("valid", split, None, 100)
    np.random.seed(123)
    assert torch.equal(pos, split["valid"]["edge"].t()[:, np.random.permutation(10)])
    assert pos.shape == (2, 10) and neg.shape == (2, 20)


def test_rank_stage_helpers(eps):
    """rank.py:260-272 sweep schedule, :222-251 --valid_proposal splice, to_undirected (PyG, restated)."""
    from eps_amd import rank_stage
    ns = argparse.Namespace
    assert rank_stage.sweep_schedule(ns(sweep_num=4, sweep_min=None, sweep_max=None, num_sorted_edge=None)) == [0, 750, 1500, 2250, 3000]
    assert rank_stage.sweep_schedule(ns(sweep_num=2, sweep_min=100, sweep_max=300, num_sorted_edge=7)) == [100, 200, 300]
    assert rank_stage.sweep_schedule(ns(sweep_num=None, sweep_min=None, sweep_max=None, num_sorted_edge=530000)) == [530000]
    assert rank_stage.sweep_schedule(ns(sweep_num=None, sweep_min=None, sweep_max=None, num_sorted_edge=None)) == [0]
    from eps_amd.rank_helpers import to_undirected
    und = to_undirected(torch.tensor([[3, 1, 1], [1, 3, 2]]))
    assert und.t().tolist() == [[1, 2], [1, 3], [2, 1], [3, 1]]
    # splice: both directions of the validation edges on top with score 100000, proposals that are validation edges dropped
    props = torch.tensor([[5., 6., .9], [1., 2., .8], [2., 1., .7], [7., 8., .6]])
    valid = torch.tensor([[1, 2], [3, 4]])
    out = rank_stage.splice_valid_proposals(props, valid)
    assert out[:4, :2].long().tolist() == [[1, 2], [2, 1], [3, 4], [4, 3]] and bool((out[:4, 2] == 100000.0).all())
    assert out[4:, :2].long().tolist() == [[5, 6], [7, 8]]
    top = {tuple(r) for r in out[:4, :2].long().tolist()}
    assert top == {(1, 2), (2, 1), (3, 4), (4, 3)}       # rank.py:249 assertion: the top block covers valid_pos_set


def test_rank_supervision_flags_append_proposals_to_train_set(eps):
    """rank.py:303-304: --only_supervision / --also_supervision append the proposal edges to split_edge['train']['edge']
    (accumulating over sweep points); without either flag the training set is untouched."""
    from eps_amd import rank_stage
    ns = argparse.Namespace
    extra = torch.tensor([[7, 8, 9], [1, 2, 3]])                      # [2,k] like sorted_test_edges[:k,:2].t().long()
    for flags, grows in ((dict(only_supervision=True, also_supervision=False), True),
                         (dict(only_supervision=False, also_supervision=True), True),
                         (dict(only_supervision=False, also_supervision=False), False)):
        split = {'train': {'edge': torch.tensor([[0, 1], [2, 3]])}}
        rank_stage.append_supervision(ns(**flags), split, extra)
        want = [[0, 1], [2, 3]] + ([[7, 1], [8, 2], [9, 3]] if grows else [])
        assert split['train']['edge'].tolist() == want
        rank_stage.append_supervision(ns(**flags), split, extra[:, :1])       # second sweep point: accumulates
        assert split['train']['edge'].tolist() == want + ([[7, 1]] if grows else [])


def test_filter_rank_argument_surface(eps):
    """Same flags as the reference parsers (filter.py:27-47, rank.py:130-163) + the documented extensions."""
    from eps_amd import filter_stage, rank_stage
    f = {a.dest for a in filter_stage.make_parser()._actions}
    assert {"dataset", "model", "checkpoint", "num_layers", "hidden_channels", "dropout", "batch_size", "lr", "epochs",
            "use_feature", "use_learnable_embedding", "device"} <= f
    r = {a.dest for a in rank_stage.make_parser()._actions}
    assert {"dataset", "model", "runs", "sorted_edge_path", "num_sorted_edge", "sweep_max", "sweep_min", "sweep_num",
            "only_supervision", "also_supervision", "gen_dataset_only", "valid_proposal", "out_name", "save_models",
            "num_layers", "hidden_channels", "dropout", "batch_size", "lr", "epochs", "use_feature",
            "use_learnable_embedding", "device", "log_steps", "eval_steps"} <= r


def test_path_counts_and_bucket_sizing_helpers():
    """ops.max_column_paths (sizes the bucket scratch of the fused expansion) == max of candidates.path_counts over any
    column range, incl. ranges that start or end with empty columns; segment_bounds caps the per-column bound at N."""
    import eps_amd  # noqa: F401
    from eps_amd import candidates, ops, synth
    from eps_amd.graph import CSRGraph
    g = synth.rmat_graph(10, 6, 3, "cpu")
    pc = candidates.path_counts(g)
    deg = g.rowptr[1:] - g.rowptr[:-1]
    v = int(torch.argmax(deg))
    want_v = int(deg[g.col[g.rowptr[v]:g.rowptr[v + 1]].long()].sum())
    assert int(pc[v]) == want_v and candidates.max_paths_of(g) == int(pc.max())
    for lo, hi in ((0, g.n_rows), (5, 300), (100, 101), (0, 1), (7, 7)):
        assert ops.max_column_paths(g.rowptr, g.col, lo, hi) == (int(pc[lo:hi].max()) if hi > lo else 0)
    rp = torch.tensor([0, 0, 2, 2, 3, 3], dtype=torch.int64)
    col = torch.tensor([3, 1, 1], dtype=torch.int32)
    small = CSRGraph(rp, col, None, 5, 5)
    d = rp[1:] - rp[:-1]
    for lo, hi in ((0, 5), (0, 1), (1, 2), (2, 5), (4, 5)):
        want = max([int(d[col[rp[c]:rp[c + 1]].long()].sum()) for c in range(lo, hi)] + [0])
        assert ops.max_column_paths(rp, col, lo, hi) == want
    pre, pre_host = candidates.segment_bounds(g)
    ub = pre[1:] - pre[:-1]
    assert torch.equal(ub, torch.clamp(pc, max=g.n_rows)) and torch.equal(pre.cpu(), pre_host) and int(pre[0]) == 0
    order = candidates.heaviest_first(g, 10, 200)
    assert sorted(order.tolist()) == list(range(190)) and bool((pc[10:200][order.long()][:-1] >= pc[10:200][order.long()][1:]).all())
    assert candidates.path_counts(small).tolist() == [0, int(d[3] + d[1]), 0, int(d[1]), 0]


def test_select_topk_tensor_ops_vs_brute_force():
    """scan.select_topk_torch (the host mirror the device selection is checked against): the K best DIRECTED rows of a list of
    unordered pairs, score descending then key ascending -- against a brute-force sort of both orientations."""
    import torch
    from eps_amd import scan
    g = torch.Generator().manual_seed(4)
    for n, k in ((1, 1), (1, 5), (50, 7), (50, 99), (50, 100), (50, 1000), (2000, 1501)):
        u = torch.randint(0, 500, (n,), generator=g)
        v = u + 1 + torch.randint(0, 500, (n,), generator=g)
        keys = torch.unique((v << 32) | u)
        vals = torch.randint(0, 6, (keys.numel(),), generator=g).float() / 3          # many ties
        got_k, got_v = scan.select_topk_torch(keys, vals, k)
        rows = [(-float(s), int(kk)) for kk, s in zip(keys.tolist(), vals.tolist())]
        rows += [(-float(s), ((kk & 0xFFFFFFFF) << 32) | (kk >> 32)) for kk, s in zip(keys.tolist(), vals.tolist())]
        rows.sort()
        want = rows[:k]
        assert got_k.tolist() == [r[1] for r in want] and got_v.tolist() == [-r[0] for r in want]


def test_original_keys_after_relabelling():
    """Survivor keys in the scanned graph's labels -> the same unordered pairs in the original labels (larger id in the high word)."""
    import torch
    from eps_amd import scan
    perm = torch.tensor([4, 0, 3, 1, 2])                   # new id i is old id perm[i]
    keys = torch.tensor([(3 << 32) | 1, (4 << 32) | 0, (2 << 32) | 1])      # (u', v') = (1,3), (0,4), (1,2)
    out = scan._original_keys(keys, perm)
    assert out.tolist() == [(1 << 32) | 0, (4 << 32) | 2, (3 << 32) | 0]
    assert scan._original_keys(keys, None) is keys


def test_weight_keyed_cache_holds_the_tensor_not_its_address(eps):
    """CSRGraph.weight_cached (scan tables derived from a weight tensor): an entry is found again for the SAME tensor object at
    the same version, never for another tensor that happens to live at a recycled address; at most four tables are kept."""
    g = eps.CSRGraph.from_edge_index(torch.tensor([[0, 1], [1, 0]]), None, sparse_sizes=(2, 2))
    calls = []

    def build(tag):
        calls.append(tag)
        return tag
    w1 = torch.ones(2)
    assert g.weight_cached("t", w1, lambda: build("a")) == "a" and g.weight_cached("t", w1, lambda: build("x")) == "a"
    w1.add_(1.0)                                          # in-place update: a new version, a new table
    assert g.weight_cached("t", w1, lambda: build("b")) == "b"
    ptr = w1.data_ptr()
    del w1
    w2 = torch.ones(2)                                    # the allocator may hand out the same address again
    assert g.weight_cached("t", w2, lambda: build("c")) == "c", (ptr, w2.data_ptr())
    for i in range(6):
        g.weight_cached("t", torch.ones(2) * i, lambda i=i: build(i))
    assert len(g._cache[("by_weight", "t")]) == 4
    assert g.weight_cached("other", None, lambda: build("none")) == "none" and g.weight_cached("other", None, lambda: build("z")) == "none"


def test_scan_host_rules(eps):
    """Host-side rules of the threshold scan that need no device: the screening fixed point leaves bit 31 free, the survivor
    list refuses sizes its 32-bit slot positions cannot address, the exact selection in tensor ops."""
    from eps_amd import ops, scan
    for bound, deg in ((2646.0, 13230), (1.0, 3), (3.9e6, 1000), (0.0, 0)):
        sh = scan.screen_shift(bound, deg)
        assert 0 <= sh <= scan.MAX_SCREEN_SHIFT and bound * (1 << sh) + deg < (1 << 31) - 2
        assert sh == scan.MAX_SCREEN_SHIFT or bound * (1 << (sh + 1)) + deg >= (1 << 31) - 2
    assert scan._capacity(1000, 0) == 1000 and scan._capacity(10, 5) == 15
    with pytest.raises(ops._lib.EpsError):
        scan._capacity(ops.SURVIVOR_SLOTS_MAX + 1, 0)
    keys = torch.tensor([(5 << 32) | 1, (7 << 32) | 2, (9 << 32) | 3], dtype=torch.int64)
    vals = torch.tensor([2.0, 3.0, 2.0])
    k, v = scan.select_topk_torch(keys, vals, 3)
    assert k.tolist() == [(2 << 32) | 7, (7 << 32) | 2, (1 << 32) | 5] and v.tolist() == [3.0, 3.0, 2.0]


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("ties", [False, True])
def test_distributed_final_ordering_concatenates_to_the_declared_order(world, ties):
    """scan._ordered_rows_distributed's host logic on CPU tensors: score ranges from ``score_splitters`` partition the selected
    pairs (equal scores never straddle a boundary), their sizes match ``score_range_counts``, and the per-range orderings --
    mirror, key ascending, score descending: ``select_topk_torch`` -- concatenate in rank order to exactly the rows a single
    rank orders: byte-identical for 1, 2, 4 and 8 logical shards."""
    from eps_amd import scan
    gen = torch.Generator().manual_seed(17 + world)
    n = 50_000
    u = torch.randint(0, 3000, (n,), generator=gen)
    v = u + 1 + torch.randint(0, 3000, (n,), generator=gen)
    keys = torch.unique((v << 32) | u)
    n = keys.numel()
    vals = torch.rand(n, generator=gen) * 5
    if ties:
        vals = torch.round(vals * 3) / 3                       # a handful of distinct scores: whole levels of ties
    k = 2 * n - 7
    want_k, want_v = scan.select_topk_torch(keys, vals, k)
    sp = scan.score_splitters(vals, world)
    assert sp.numel() == world - 1 and bool((sp[:-1] >= sp[1:]).all())
    counts = scan.score_range_counts(vals, sp).tolist()
    assert sum(counts) == n
    chunks_k, chunks_v = [], []
    for r in range(world):
        lo = float(sp[r]) if r < world - 1 else float("-inf")
        hi = float(sp[r - 1]) if r > 0 else float("inf")
        m = (vals >= lo) & (vals < hi)
        assert int(m.sum()) == counts[r]
        rk, rv = scan.select_topk_torch(keys[m], vals[m], 2 * int(m.sum()))
        chunks_k.append(rk)
        chunks_v.append(rv)
    got_k, got_v = torch.cat(chunks_k)[:k], torch.cat(chunks_v)[:k]
    assert torch.equal(got_k, want_k) and torch.equal(got_v, want_v)


@pytest.mark.parametrize("world,ties", [(2, False), (4, True), (8, False), (3, True)])
def test_deal_plan_partitions_the_selection_for_one_all_to_all(world, ties):
    """scan._deal_plan / _deal_rows' host logic on CPU tensors (the sharded step's one exchange, r05): from everybody's gathered
    scores and the cut every rank derives the same splitters and the same table counts[r][q] of who sends how many selected pairs
    of range q; emulating the all-to-all with slices, the per-range orderings concatenate in rank order to exactly the rows a
    single rank orders -- uneven shards, padding, ties at the cut and at the splitters included."""
    from eps_amd import scan
    gen = torch.Generator().manual_seed(5 + world)
    sizes = [int(x) for x in torch.randint(50, 4000, (world,), generator=gen)]
    sizes[0] = 7                                               # a rank with next to nothing
    room = max(sizes) + 100
    keys_r, vals_r = [], []
    base = 0
    for n in sizes:
        u = torch.arange(base, base + n)
        keys_r.append(((u + 100_000) << 32) | u)
        v = torch.rand(n, generator=gen) * 5
        vals_r.append(torch.round(v * 4) / 4 if ties else v)
        base += n
    scores_all = torch.full((world, room), float("-inf"))
    for r in range(world):
        scores_all[r, :sizes[r]] = vals_r[r]
    all_k, all_v = torch.cat(keys_r), torch.cat(vals_r)
    for k2 in (1, sum(sizes) // 3, sum(sizes), sum(sizes) + 50):
        srt = torch.sort(all_v, descending=True).values
        cut = srt[k2 - 1:k2] if k2 <= srt.numel() else torch.tensor([float("-inf")])
        sp, counts = scan._deal_plan(scores_all, cut, world)
        n_sel = int((all_v >= cut).sum())
        assert sp.numel() == world - 1 and bool((sp[:-1] >= sp[1:]).all()) and int(counts.sum()) == n_sel
        c = counts.tolist()
        # every rank groups its own selected pairs by range; range q's owner receives the pieces in rank order
        recv_k, recv_v = [[] for _ in range(world)], [[] for _ in range(world)]
        for r in range(world):
            m = vals_r[r] >= cut
            sk, sv = keys_r[r][m], vals_r[r][m]
            rng = (sv.unsqueeze(1) < sp.unsqueeze(0)).sum(1)
            assert torch.bincount(rng, minlength=world).tolist() == c[r]
            for q in range(world):
                recv_k[q].append(sk[rng == q])
                recv_v[q].append(sv[rng == q])
        k = 2 * n_sel - 3 if n_sel > 2 else 2 * n_sel
        chunks = [scan.select_topk_torch(torch.cat(recv_k[q]), torch.cat(recv_v[q]), 2 * sum(c[r][q] for r in range(world)))
                  for q in range(world)]
        got_k, got_v = torch.cat([ck for ck, _ in chunks])[:k], torch.cat([cv for _, cv in chunks])[:k]
        want_k, want_v = scan.select_topk_torch(all_k[all_v >= cut], all_v[all_v >= cut], k)
        assert torch.equal(got_k, want_k) and torch.equal(got_v, want_v), (world, ties, k2)


def test_sharded_proposal_files_read_back_in_rank_order(tmp_path):
    """filter.py --shard_proposals leaves <file>.shard{r}of{N} (each rank the chunk of the sorted list it ordered); the reader that
    rank.py goes through (proposals.load_proposals, rank.py:219 + :294) concatenates them in rank order, prefers a plain file when
    one exists, and refuses a list with a missing shard."""
    import torch
    from eps_amd import proposals
    rows = torch.cat([torch.arange(30, dtype=torch.float32).view(10, 3), torch.arange(30, 60, dtype=torch.float32).view(10, 3)])
    path = str(tmp_path / "x_sorted_edges.pt")
    parts = [rows[:7], rows[7:7], rows[7:]]                           # (an empty shard in the middle)
    for r, part in enumerate(parts):
        proposals.save_sorted_edges_shard(path, part, r, 3)
    assert torch.equal(proposals.load_sorted_edges(path), rows)
    assert torch.equal(proposals.load_proposals(path, 9), rows[:9, :2].t().long())
    import os
    os.remove(proposals.shard_path(path, 1, 3))
    with pytest.raises(FileNotFoundError):
        proposals.load_sorted_edges(path)
    proposals.save_sorted_edges(path, rows[:5])
    assert torch.equal(proposals.load_sorted_edges(path), rows[:5])   # (a plain file wins)


def test_sketch_safe_shift_keeps_a_whole_piece_below_2_32():
    """scan.sketch_safe_shift: under the returned fixed point SKETCH_PIECE_PATHS paths of the heaviest weight (2^-40 units, rounded
    up to screening units) sum to less than 2^32 -- and one bit finer they would not, unless the screen's own maximum stops it."""
    import math
    from eps_amd import scan
    for w in (1.0 / math.log(2.0), 1.0, 0.25, 7.5e-5, 1000.0, 22.0):
        f_hi = int(round(w * 2.0 ** 40))
        s = scan.sketch_safe_shift(f_hi)
        assert 0 <= s <= scan.MAX_SCREEN_SHIFT
        units = (f_hi >> (40 - s)) + 2                       # (what screen_weights rounds up to, and one to spare)
        assert units * scan.SKETCH_PIECE_PATHS < 1 << 32, (w, s)
        if s < scan.MAX_SCREEN_SHIFT:
            assert ((f_hi >> (40 - s - 1)) + 2) * scan.SKETCH_PIECE_PATHS >= 1 << 32, (w, s)
    assert scan.sketch_safe_shift(int(round(2.0 ** 40 / math.log(2.0)))) == 18      # Adamic-Adar: 1 / ln 2 is the heaviest weight
