"""GPU parity of the model facades (reference signatures) vs the golden gnn_stack vectors (reference layer
loops over the oracle conv) and vs the oracle's CSR float32 restatement on a larger seeded graph."""
import argparse
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _load(eps, kind, L, dev):
    from eps_amd import models
    d = np.load(os.path.join(GOLDEN, f"gnn_stack_{kind}_L{L}.npz"))
    n, fin = d["x"].shape
    H = d["sd::emb.weight"].shape[1]
    cls = models.GCN if kind == "gcn" else models.SAGE
    model = models.LinkGNN(torch.nn.Embedding(n, H), cls(fin + H, H, H, L, 0.5), models.LinkPredictor(H, H, 1, L, 0.5))
    model.load_state_dict({k[4:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("sd::")})
    model = model.to(dev).eval()
    import scipy.sparse as ssp
    adj = eps.CSRGraph.from_scipy(ssp.csr_matrix(d["A"]), device=dev)
    return d, model, adj


@pytest.mark.parametrize("kind,L", [("gcn", 2), ("gcn", 3), ("sage", 2), ("sage", 3)])
def test_linkgnn_matches_reference_stack(eps, dev, kind, L):
    d, model, adj = _load(eps, kind, L, dev)
    x = torch.from_numpy(d["x"]).to(dev)
    edges = torch.from_numpy(d["edges"].astype(np.int64)).to(dev)
    out = model(x, edges, adj)                                   # reference call shape: model(x, edges, adj_t)
    assert out.shape == (edges.shape[1], 1)
    h = model.embeddings(x, adj)
    assert float((h.cpu() - torch.from_numpy(d["h"])).abs().max()) <= 1e-5 * max(1.0, float(np.abs(d["h"]).max()))
    assert rel_err(out.squeeze(1).cpu().numpy(), d["prob"]) <= TOL
    # second batch reuses the cached embeddings; changing a parameter invalidates them
    assert model.embeddings(x, adj) is h
    with torch.no_grad():
        model.emb.weight.add_(1.0)
    assert model.embeddings(x, adj) is not h


@pytest.mark.parametrize("tag", ["plain", "weighted"])
def test_sage_conv_matches_reference_witness(eps, dev, tag):
    """HIP SAGEConv (mean-mode SpMM + two GEMMs) against the reference's own vendored conv code
    (models.SAGEConv2.forward, models.py:358-384; fixture made by oracle/gen_golden.py): ``out_1hop`` is the class with
    its extra hop (:366) skipped == SAGEConv; ``out_2hop`` is the class as written (two mean aggregations)."""
    import scipy.sparse as ssp
    from eps_amd import models, ops
    d = np.load(os.path.join(GOLDEN, f"sageconv_ref_{tag}.npz"))
    fin, fout = d["x"].shape[1], d["lin_l_weight"].shape[0]
    conv = models.SAGEConv(fin, fout)
    conv.load_state_dict({"lin_l.weight": torch.from_numpy(d["lin_l_weight"]), "lin_l.bias": torch.from_numpy(d["lin_l_bias"]),
                          "lin_r.weight": torch.from_numpy(d["lin_r_weight"])})
    conv = conv.to(dev).eval()
    adj = eps.CSRGraph.from_scipy(ssp.csr_matrix(d["A"]), device=dev)
    x = torch.from_numpy(d["x"]).to(dev)
    out = conv(x, adj).cpu().numpy()
    assert float(np.abs(out - d["out_1hop"]).max()) <= TOL * max(1.0, float(np.abs(d["out_1hop"]).max()))
    xp = torch.zeros((x.shape[0], 12), device=dev)
    xp[:, :fin] = x                                              # float4 rows for the bare SpMM
    agg2 = ops.spmm_csr(adj.rowptr, adj.col, None, ops.spmm_csr(adj.rowptr, adj.col, None, xp, mean=True), mean=True)[:, :fin]
    out2 = ops.gemm(agg2.contiguous(), conv.lin_l.weight.detach(), bias=conv.lin_l.bias.detach())
    out2 = ops.gemm(x, conv.lin_r.weight.detach(), out=out2, accumulate=True).cpu().numpy()
    assert float(np.abs(out2 - d["out_2hop"]).max()) <= TOL * max(1.0, float(np.abs(d["out_2hop"]).max()))


def test_sage_conv_transform_first(eps, oracle, dev):
    """A SAGE layer that narrows (ppa: 58 features + 256-d embedding -> 256) aggregates lin_l(x) instead of x -- the mean
    commutes with the linear map (models.py:358-384: lin_l(mean_j x_j) + lin_r(x_i)).  Same result as the aggregate-first
    order and as the oracle's dense float64 conv, within the float32 gate; also as rows [lo, hi) of a row shard."""
    from eps_amd import models, synth
    g = synth.rmat_graph(12, 10, 4, dev)
    n, fin, fout = g.n_rows, 314, 256
    gen = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(n, fin, generator=gen, device=dev)
    torch.manual_seed(3)
    conv = models.SAGEConv(fin, fout).to(dev).eval()
    assert conv.TRANSFORM_FIRST and fin > fout
    a = conv(x, g, relu=True)
    conv.TRANSFORM_FIRST = False
    b = conv(x, g, relu=True)
    conv.TRANSFORM_FIRST = True
    scale = max(1.0, float(b.abs().max()))
    assert float((a - b).abs().max()) <= TOL * scale
    A = g.to_scipy().astype(np.float64)
    deg = np.maximum(np.asarray(A.sum(1)).ravel(), 1.0)
    xs = x.cpu().numpy().astype(np.float64)
    want = (A @ xs) / deg[:, None] @ conv.lin_l.weight.detach().cpu().numpy().astype(np.float64).T \
        + conv.lin_l.bias.detach().cpu().numpy().astype(np.float64) + xs @ conv.lin_r.weight.detach().cpu().numpy().astype(np.float64).T
    want = np.maximum(want, 0.0)
    assert float(np.abs(a.cpu().numpy() - want).max()) <= TOL * max(1.0, float(np.abs(want).max()))
    rows = conv.forward_rows(x, g, 1000, 3000, relu=True)
    assert torch.equal(rows, a[1000:3000])


def test_linkpredictor_reference_signature(eps, dev):
    from eps_amd import models
    d = np.load(os.path.join(GOLDEN, "linkpred_H256_L3.npz"))
    lp = models.LinkPredictor(256, 256, 1, 3, 0.5)
    lp.load_state_dict({f"lins.{i}.{k}": torch.from_numpy(d[f"{k[0]}{i}"]) for i in range(3) for k in ("weight", "bias")})
    lp = lp.to(dev).eval()
    h = torch.from_numpy(d["h"]).to(dev)
    e = torch.from_numpy(d["edges"].astype(np.int64)).to(dev)
    out = lp(h[e[0]], h[e[1]])                                   # models.py:506 call shape
    assert out.shape == (e.shape[1], 1)
    assert rel_err(out.squeeze(1).cpu().numpy(), d["prob"]) <= TOL


@pytest.mark.parametrize("kind", ["gcn", "sage"])
def test_gnn_vs_oracle_larger_graph(eps, oracle, dev, kind):
    """Weighted (collab-like) graph, in=70 (not a multiple of 4), H=64, L=3."""
    from eps_amd import models, synth
    g = synth.rmat_graph(11, 10, 9, "cpu")
    gen = torch.Generator().manual_seed(2)
    val = torch.randint(1, 5, (g.nnz(),), generator=gen).float()
    A = g.to_scipy(); A.data[:] = val.numpy(); A = ((A + A.T) / 2).tocsr(); A.sort_indices()
    adj = eps.CSRGraph.from_scipy(A, device=dev)
    n = g.n_rows
    x = torch.randn(n, 70, generator=gen)
    cls = models.GCN if kind == "gcn" else models.SAGE
    torch.manual_seed(4)
    net = cls(70, 64, 64, 3, 0.0)
    if kind == "gcn":
        for c in net.convs:
            torch.nn.init.normal_(c.bias, std=0.1)
    net = net.to(dev).eval()
    h = net(x.to(dev), adj).cpu().numpy()
    sd = {k: v.cpu().numpy() for k, v in net.state_dict().items()}
    if kind == "gcn":
        ref = oracle.gcn_forward_csr(A.indptr, A.indices, A.data.astype(np.float32), x.numpy(),
                                     [sd[f"convs.{i}.weight"] for i in range(3)], [sd[f"convs.{i}.bias"] for i in range(3)])
    else:
        ref = oracle.sage_forward_csr(A.indptr, A.indices, x.numpy(), [sd[f"convs.{i}.lin_l.weight"] for i in range(3)],
                                      [sd[f"convs.{i}.lin_l.bias"] for i in range(3)],
                                      [sd[f"convs.{i}.lin_r.weight"] for i in range(3)])
    assert float(np.abs(h - ref).max()) <= 2e-5 * max(1.0, float(np.abs(ref).max()))


def test_cn_predictor_simple_and_adamic(eps, oracle, dev):
    from eps_amd import models
    d = np.load(os.path.join(GOLDEN, "pairs_collab_like.npz"))
    n = len(d["rowptr"]) - 1
    adj = eps.CSRGraph(torch.from_numpy(d["rowptr"]), torch.from_numpy(d["col"]), torch.from_numpy(d["val"]), n, n).to(dev)
    edges = torch.from_numpy(d["pairs"].astype(np.int64)).to(dev)
    m = models.CommonNeighborsPredictor(None, 0, 8, 8, 2, 0.0, model_type="simple").eval()
    out = m(None, edges, adj)
    assert out.shape == (edges.shape[1],) and np.array_equal(out.cpu().numpy(), d["cn"])
    # 'adamic' flavour (models.py:547-554): sum over common w of 1/log(rowsum+1e-6), sigmoid, values ignored
    m2 = models.CommonNeighborsPredictor(None, 0, 8, 8, 2, 0.0, model_type="adamic").eval()
    got = m2(None, edges, adj).cpu().numpy()
    deg = np.add.reduceat(d["val"], d["rowptr"][:-1].clip(max=len(d["val"]) - 1)) if False else None
    import scipy.sparse as ssp
    A = ssp.csr_matrix((d["val"], d["col"], d["rowptr"]), shape=(n, n))
    w = (1.0 / np.log(np.asarray(A.sum(1)).reshape(-1).astype(np.float32) + np.float32(1e-6))).astype(np.float32)
    _, _, ws = oracle.pair_scores(d["rowptr"], d["col"], None, w, d["pairs"][0], d["pairs"][1])
    ref = 1.0 / (1.0 + np.exp(-ws.astype(np.float64)))
    assert float(np.abs(got - ref).max()) <= 1e-5


def test_eval_harness_and_hits(eps, oracle, dev):
    """test_adamic / test_resource_allocation / test (train_and_eval.py) end to end on a small synthetic split:
    Hits@K identical to the oracle's scores pushed through the restated Hits@K."""
    from eps_amd import evaluate, models, synth
    g = synth.rmat_graph(11, 8, 21, "cpu")
    n = g.n_rows
    gen = torch.Generator().manual_seed(5)
    row, col, _ = g.coo()
    und = row < col
    ei = torch.stack([row[und], col[und]])
    perm = torch.randperm(ei.shape[1], generator=gen)
    n_val, n_test = 400, 400
    valid, test_e, train = ei[:, perm[:n_val]], ei[:, perm[n_val:n_val + n_test]], ei[:, perm[n_val + n_test:]]
    split = {"train": {"edge": train.t()}, "eval_train": {"edge": train.t()[:n_val]},
             "valid": {"edge": valid.t(), "edge_neg": torch.randint(0, n, (1500, 2), generator=gen)},
             "test": {"edge": test_e.t(), "edge_neg": torch.randint(0, n, (1500, 2), generator=gen)}}
    adj = eps.add_edges("ppa", train, torch.ones(train.shape[1]), torch.zeros(2, 0, dtype=torch.long), n).to(dev)
    full = eps.add_edges("ppa", train, torch.ones(train.shape[1]), torch.cat([valid, valid.flip(0)], 1), n).to(dev)
    data = argparse.Namespace(x=None, adj_t=adj, full_adj_t=full, num_nodes=n, edge_index=train)
    args = argparse.Namespace(dataset="ppa", model="adamic_ogb")
    ev = evaluate.evaluators["ppa"]
    res = evaluate.test_adamic(None, data, split, ev, 1024, args, dev)

    def oracle_scores(graph, edges, mode):
        rp, ci = graph.rowptr.cpu().numpy(), graph.col.cpu().numpy()
        w = oracle.node_weights(oracle.col_sums(rp, ci, None, n), mode)
        return oracle.pair_scores(rp, ci, None, w, edges[:, 0].numpy(), edges[:, 1].numpy())[2]

    for mode, fn, name in ((oracle.W_AA, evaluate.test_adamic, "adamic_ogb"),
                           (oracle.W_RA, evaluate.test_resource_allocation, "resource_allocation")):
        args.model = name
        res = fn(None, data, split, ev, 1024, args, dev)
        pv, nv = oracle_scores(adj, split["valid"]["edge"], mode), oracle_scores(adj, split["valid"]["edge_neg"], mode)
        pt, nt = oracle_scores(full, split["test"]["edge"], mode), oracle_scores(full, split["test"]["edge_neg"], mode)
        for K in (10, 100, 200):
            tr, va, te = res[f"Hits@{K}"]
            assert va == oracle.hits_at_k(pv, nv, K) and te == oracle.hits_at_k(pt, nt, K)
    # model-driven loop (train_and_eval.test) with the CN predictor
    args.model = "simple"
    m = models.CommonNeighborsPredictor(None, 0, 8, 8, 2, 0.0, model_type="simple").eval()
    res = evaluate.test(m, data, split, ev, 1024, args, dev)
    cn = lambda gr, e: oracle.pair_scores(gr.rowptr.cpu().numpy(), gr.col.cpu().numpy(), None, None,  # noqa: E731
                                          e[:, 0].numpy(), e[:, 1].numpy())[1]
    assert res["Hits@100"][1] == oracle.hits_at_k(cn(adj, split["valid"]["edge"]), cn(adj, split["valid"]["edge_neg"]), 100)
    assert res["Hits@100"][2] == oracle.hits_at_k(cn(full, split["test"]["edge"]), cn(full, split["test"]["edge_neg"]), 100)


def test_proposal_sort_rule_and_file(eps, dev, tmp_path):
    from eps_amd import proposals
    g = torch.Generator().manual_seed(1)
    edges = torch.randint(0, 5000, (2, 30000), generator=g)
    scores = torch.randint(0, 20, (30000,), generator=g).float()          # CN-like: massive ties
    t = proposals.sorted_edges_tensor(edges.to(dev), scores.to(dev))
    ref_order = torch.sort(scores, descending=True, stable=True).indices
    assert t.dtype == torch.float32 and t.shape == (30000, 3)
    assert torch.equal(t[:, :2].cpu().long(), edges[:, ref_order].t()) and torch.equal(t[:, 2].cpu(), scores[ref_order])
    path = str(tmp_path / "x_sorted_edges.pt")
    proposals.save_sorted_edges(path, t)
    back = proposals.load_proposals(path, 100)
    assert back.shape == (2, 100) and back.dtype == torch.int64 and torch.equal(back, edges[:, ref_order[:100]])
    # top-k keys of two shards merge to the global top-k
    k0 = proposals.top_k_keys(scores[:15000].to(dev), 100, id_base=0)
    k1 = proposals.top_k_keys(scores[15000:].to(dev), 100, id_base=15000)
    merged = proposals.merge_top_k([k0, k1], 100)
    _, ids = eps.ops.unpack_keys(merged)
    assert torch.equal(ids.cpu(), ref_order[:100])


@pytest.mark.parametrize("kind", ["gcn", "sage"])
def test_row_sharded_last_layer_equals_full_forward(eps, dev, kind):
    """dist path: the last layer computed per row block (virtual ranks 0..3 on one GPU) and concatenated == the
    unsharded forward, bit for bit (same kernels, same per-row arithmetic)."""
    from eps_amd import models, synth
    g = synth.rmat_graph(11, 8, 6, dev)
    torch.manual_seed(3)
    net = (models.GCN if kind == "gcn" else models.SAGE)(70, 64, 64, 3, 0.0).to(dev).eval()
    x = torch.randn(g.n_rows, 70, device=dev)
    full = net(x, g)
    world = 4
    parts = []
    for rank in range(world):
        parts.append(net.forward_sharded(x, g, rank, world, lambda local, bounds: local))
    assert torch.equal(torch.cat(parts, 0), full)


@pytest.mark.parametrize("kind", ["gcn", "sage"])
def test_hubs_first_relabelling_is_transparent(eps, dev, kind, monkeypatch):
    """LinkGNN.embeddings on the degree-ordered relabelling of the graph (large graphs: the SpMM's gathers stay cache-
    resident) returns the embeddings in the caller's node order, equal to the plain run up to the summation order."""
    from eps_amd import models, synth
    g = synth.rmat_graph(12, 12, 4, dev)
    n = g.n_rows
    gen = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(n, 20, generator=gen, device=dev)
    cls = models.GCN if kind == "gcn" else models.SAGE
    torch.manual_seed(3)
    model = models.LinkGNN(torch.nn.Embedding(n, 12), cls(32, 32, 32, 3, 0.0), models.LinkPredictor(32, 32, 1, 3, 0.0)).to(dev).eval()
    plain = model.embeddings(x, g).clone()
    monkeypatch.setattr(models, "REORDER_MIN_NODES", 1)
    model._h_key = None
    gp, perm, inv = g.degree_ordered()
    assert torch.equal(perm[inv], torch.arange(n, device=dev)) and bool((gp.degree()[:-1] >= gp.degree()[1:]).all())
    relab = model.embeddings(x, g)
    assert float((relab - plain).abs().max()) <= 2e-5 * max(1.0, float(plain.abs().max()))
