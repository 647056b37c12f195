#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json by IMPORTING the reference (in the build container only).

Run:  python oracle/gen_golden.py            (needs /root/reference; never runs on the GPU box)

The reference is pure Python; its hot-path arithmetic lives in adamic_utils.AA,
train_and_eval.resource_allocation and models.LinkPredictor, all importable here once the
missing third-party modules (ogb, torch_geometric, torch_sparse) are stubbed.  Only the
emitted vectors (inputs + expected outputs) are committed -- no reference source travels.

Fixtures
  pairs_<graph>.npz  CSR graph + pair list + AA (f32) + RA (f32 A) + RA (int64 A -> f64 math)
                     + CN via the reference's own SciPy expression with unit multiplier.
  linkpred_*.npz     seeded LinkPredictor weights, embeddings h, edges -> probabilities.
  gnn_stack_*.npz    reference GCN/SAGE/LinkGNN layer loops (models.py:181-187, :434-440,
                     :500-506) driven over the oracle's conv restatement injected as the
                     GCNConv/SAGEConv stub: pins loop structure, not conv arithmetic.
  sageconv_ref_*.npz the reference's vendored SAGEConv2.forward (models.py:358-384) run over a stand-in
                     SparseTensor/matmul, with and without its extra hop (:366): pins the SAGE conv arithmetic.
  model_configs.json default_model_configs output for every (dataset, model).
  state_dict_keys.json
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import scipy.sparse as ssp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import eps_oracle as orc  # noqa: E402


# ----------------------------------------------------------------------------- shims
class EI:
    """numpy-backed stand-in for a LongTensor edge list: SciPy >= 1.12 refuses torch tensors as
    fancy indices at adamic_utils.py:22 / train_and_eval.py:212."""

    def __init__(self, a):
        self.a = np.asarray(a)

    def size(self, d):
        return self.a.shape[d]

    def t(self):
        return EI(self.a.T)

    def __getitem__(self, k):
        r, idx = k
        return self.a[r, idx.numpy() if hasattr(idx, "numpy") else np.asarray(idx)]


def install_stubs():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    mod("ogb")
    lp = mod("ogb.linkproppred")

    class Evaluator:
        def __init__(self, name):
            self.K = None

        def eval(self, d):
            return {}

    lp.Evaluator = Evaluator
    lp.PygLinkPropPredDataset = object
    mod("torch_geometric")
    u = mod("torch_geometric.utils")
    u.negative_sampling = None
    u.to_undirected = None
    mod("torch_geometric.transforms")
    nn = mod("torch_geometric.nn")
    conv = mod("torch_geometric.nn.conv")
    typing_ = mod("torch_geometric.typing")
    typing_.OptPairTensor = typing_.Adj = typing_.Size = object

    class MessagePassing(torch.nn.Module):
        """Minimal torch_geometric 1.7.0 MessagePassing [3p]: keeps ``aggr`` and, for a sparse adjacency, routes
        ``propagate`` to the subclass's fused ``message_and_aggregate`` -- the only route the reference's vendored
        SAGEConv2 (models.py:358-384) takes with a SparseTensor.  ``SKIP_ROOTLESS`` makes a propagate whose input is
        ``(out, None)`` an identity: that is models.py:366, the extra hop that distinguishes SAGEConv2 from PyG's
        SAGEConv, so with the switch on the reference's own forward IS the SAGEConv arithmetic."""
        SKIP_ROOTLESS = False

        def __init__(self, *a, aggr="add", **k):
            super().__init__()
            self.aggr = aggr

        def propagate(self, edge_index, size=None, **kwargs):
            x = kwargs["x"]
            if MessagePassing.SKIP_ROOTLESS and isinstance(x, tuple) and x[1] is None:
                return x[0]
            return self.message_and_aggregate(edge_index, x)

    conv.MessagePassing = MessagePassing
    ts = mod("torch_sparse")

    class SparseTensor:
        """Pattern + optional values of a square sparse matrix [3p stand-in]: just what models.py:379-380 touches."""

        def __init__(self, csr, has_value=True):
            self.csr, self.has_value = csr, has_value

        def set_value(self, value, layout=None):
            assert value is None
            return SparseTensor(self.csr, has_value=False)

    def matmul(src, other, reduce="sum"):
        """torch_sparse.matmul [3p]: rows reduced over their STORED entries; 'mean' divides the sum by the number of
        stored entries of the row (empty rows give 0)."""
        A = src.csr.astype(np.float64)
        if not src.has_value:
            A = A.copy()
            A.data[:] = 1.0
        out = A @ other.detach().numpy().astype(np.float64)
        if reduce == "mean":
            cnt = np.diff(src.csr.indptr).astype(np.float64)
            out = out / np.maximum(cnt, 1.0)[:, None]
        else:
            assert reduce in ("sum", "add")
        return torch.from_numpy(out.astype(np.float32))

    ts.SparseTensor = SparseTensor
    ts.matmul = matmul
    ts.sum = None

    # Conv stubs: parameters laid out like torch_geometric 1.7.0 [3p] (GCNConv.weight [in,out] +
    # bias; SAGEConv.lin_l (bias) / lin_r (no bias)); arithmetic = oracle restatement on a dense
    # adjacency carried in `adj_t`.
    class GCNConv(torch.nn.Module):
        def __init__(self, i, o):
            super().__init__()
            self.weight = torch.nn.Parameter(torch.empty(i, o))
            self.bias = torch.nn.Parameter(torch.empty(o))
            self.reset_parameters()

        def reset_parameters(self):
            torch.nn.init.xavier_uniform_(self.weight)
            torch.nn.init.normal_(self.bias, std=0.1)

        def forward(self, x, adj_t):
            out = orc.gcn_dense_forward(adj_t.numpy(), x.detach().numpy(),
                                        [self.weight.detach().numpy()], [self.bias.detach().numpy()])
            return torch.from_numpy(out.astype(np.float32))

    class SAGEConv(torch.nn.Module):
        def __init__(self, i, o):
            super().__init__()
            self.lin_l = torch.nn.Linear(i, o, bias=True)
            self.lin_r = torch.nn.Linear(i, o, bias=False)

        def reset_parameters(self):
            self.lin_l.reset_parameters()
            self.lin_r.reset_parameters()

        def forward(self, x, adj_t):
            out = orc.sage_dense_forward(adj_t.numpy(), x.detach().numpy(),
                                         [self.lin_l.weight.detach().numpy()],
                                         [self.lin_l.bias.detach().numpy()],
                                         [self.lin_r.weight.detach().numpy()])
            return torch.from_numpy(out.astype(np.float32))

    class _Dummy(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    nn.GCNConv, nn.SAGEConv, nn.TAGConv, nn.JumpingKnowledge = GCNConv, SAGEConv, _Dummy, _Dummy


# ----------------------------------------------------------------------------- graphs
def sym_csr(rows, cols, n, weights=None):
    rows, cols = np.asarray(rows), np.asarray(cols)
    keep = rows != cols
    rows, cols = rows[keep], cols[keep]
    w = np.ones(len(rows), dtype=np.float32) if weights is None else np.asarray(weights, np.float32)[keep]
    A = ssp.coo_matrix((w, (rows, cols)), shape=(n, n)).tocsr()
    A = (A + A.T).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    if weights is None:
        A.data[:] = 1.0
    return A.astype(np.float32)


def rmat_edges(scale, n_edges, rng, a=0.57, b=0.19, c=0.19):
    r = np.zeros(n_edges, dtype=np.int64)
    cc = np.zeros(n_edges, dtype=np.int64)
    for _ in range(scale):
        p = rng.random(n_edges)
        right = (p >= a) & (p < a + b) | (p >= a + b + c)
        down = p >= a + b
        r = r * 2 + down
        cc = cc * 2 + right
    return r, cc


def make_graphs():
    rng = np.random.default_rng(20240601)
    g = {}
    n = 500
    m = rng.random((n, n)) < 0.02
    r, c = np.nonzero(np.triu(m, 1))
    g["er500"] = sym_csr(r, c, n)
    r, c = rmat_edges(10, 8000, rng)
    g["rmat10"] = sym_csr(r, c, 1 << 10)
    r, c = rmat_edges(12, 40000, rng)
    g["rmat12"] = sym_csr(r, c, 1 << 12)
    g["star50"] = sym_csr(np.zeros(49, int), np.arange(1, 50), 50)
    r, c = np.nonzero(np.triu(np.ones((20, 20)), 1))
    g["clique20"] = sym_csr(r, c, 20)
    g["path30"] = sym_csr(np.arange(29), np.arange(1, 30), 30)
    # isolated nodes (0..4), degree-1 leaves, one hub
    r = np.concatenate([np.full(20, 5), np.arange(30, 40)])
    c = np.concatenate([np.arange(6, 26), np.arange(40, 50)])
    g["isolated_deg1"] = sym_csr(r, c, 60)
    # collab-like: power-law + integer weights 1..5, duplicates summed by symmetrisation
    r, c = rmat_edges(9, 6000, rng)
    w = rng.integers(1, 6, size=len(r)).astype(np.float32)
    g["collab_like"] = sym_csr(r, c, 1 << 9, w)
    return g


def make_pairs(A, rng, n_rand=1500):
    n = A.shape[0]
    rand = rng.integers(0, n, (2, n_rand))
    coo = A.tocoo()
    k = min(300, coo.nnz)
    sel = rng.choice(coo.nnz, k, replace=False) if coo.nnz else np.zeros(0, int)
    adjacent = np.stack([coo.row[sel], coo.col[sel]])
    selfp = np.tile(rng.integers(0, n, 50), (2, 1))
    cand, _ = orc.candidates_scipy(A)
    if len(cand) > 2500:
        cand = cand[np.sort(rng.choice(len(cand), 2500, replace=False))]
    dup = rand[:, :100]
    return np.concatenate([rand, adjacent, selfp, cand.T, dup], axis=1).astype(np.int64)


# ----------------------------------------------------------------------------- emitters
def emit_pairs(adamic_utils, train_and_eval):
    rng = np.random.default_rng(7)
    for name, A in make_graphs().items():
        pairs = make_pairs(A, rng)
        aa, _ = adamic_utils.AA(A, EI(pairs))                                  # adamic_utils.py:13-25
        ra32 = train_and_eval.resource_allocation(A, EI(pairs.T), batch_size=1024)       # eval path f32 A
        Ai = ssp.csr_matrix((np.ones(A.nnz, dtype=np.int64), A.indices, A.indptr), shape=A.shape)
        ra64 = train_and_eval.resource_allocation(Ai, EI(pairs.T), batch_size=8192)      # filter.py:130-141
        # CN through the reference's own SciPy expression (adamic_utils.py:22) with multiplier == 1
        cn = np.array(np.sum(A[pairs[0]].multiply(A[pairs[1]]), 1)).flatten().astype(np.float32)
        np.savez_compressed(
            os.path.join(OUT, f"pairs_{name}.npz"),
            rowptr=A.indptr.astype(np.int64), col=A.indices.astype(np.int32), val=A.data.astype(np.float32),
            pairs=pairs.astype(np.int32), aa=aa.numpy(), ra_f32=ra32.numpy(), ra_i64=ra64.numpy(), cn=cn)
        print(f"pairs_{name}: N={A.shape[0]} nnz={A.nnz} E={pairs.shape[1]} "
              f"mean CN={cn.mean():.2f} max AA={aa.max():.3f}")


def emit_linkpred(models):
    for seed, (tag, H, L, n_nodes, n_edges) in enumerate([("H256_L2", 256, 2, 64, 600), ("H256_L3", 256, 3, 64, 600),
                                                          ("H8_L3", 8, 3, 40, 300), ("H64_L2", 64, 2, 50, 300)]):
        torch.manual_seed(11 + seed)
        lp = models.LinkPredictor(H, H, 1, L, 0.5)
        lp.eval()
        h = torch.randn(n_nodes, H) * 0.7
        edges = torch.randint(0, n_nodes, (2, n_edges))
        with torch.no_grad():
            prob = lp(h[edges[0]], h[edges[1]]).squeeze(1)
            x = h[edges[0]] * h[edges[1]]
            for lin in lp.lins[:-1]:
                x = torch.relu(lin(x))
            logit = lp.lins[-1](x).squeeze(1)
        d = {"h": h.numpy(), "edges": edges.numpy().astype(np.int32), "prob": prob.numpy(), "logit": logit.numpy()}
        for i, lin in enumerate(lp.lins):
            d[f"w{i}"] = lin.weight.detach().numpy()
            d[f"b{i}"] = lin.bias.detach().numpy()
        np.savez_compressed(os.path.join(OUT, f"linkpred_{tag}.npz"), **d)
        print(f"linkpred_{tag}: prob range [{prob.min():.4f}, {prob.max():.4f}]")


def emit_gnn_stack(models):
    rng = np.random.default_rng(3)
    n, fin, H = 40, 12, 16
    M = np.triu(rng.random((n, n)) < 0.15, 1)
    A = (M | M.T).astype(np.float32)
    A[3, :] = 0
    A[:, 3] = 0  # one isolated node
    adj = torch.from_numpy(A)
    for kind in ["gcn", "sage"]:
        for L in (2, 3):
            torch.manual_seed(100 + L)
            gnn = (models.GCN if kind == "gcn" else models.SAGE)(fin + H, H, H, L, 0.5)
            linkpred = models.LinkPredictor(H, H, 1, L, 0.5)
            emb = torch.nn.Embedding(n, H)
            model = models.LinkGNN(emb, gnn, linkpred)
            model.eval()
            x = torch.randn(n, fin)
            edges = torch.randint(0, n, (2, 200))
            with torch.no_grad():
                h = gnn(torch.cat([emb.weight, x], 1), adj)
                prob = model(x, edges, adj).squeeze(1)
            d = {"A": A, "x": x.numpy(), "edges": edges.numpy().astype(np.int32),
                 "h": h.numpy(), "prob": prob.detach().numpy().astype(np.float32)}
            for k, v in model.state_dict().items():
                d["sd::" + k] = v.numpy()
            np.savez_compressed(os.path.join(OUT, f"gnn_stack_{kind}_L{L}.npz"), **d)
            print(f"gnn_stack_{kind}_L{L}: keys={list(model.state_dict().keys())}")


def emit_sageconv_witness(models):
    """The reference vendors ONE conv implementation on this path: SAGEConv2 (models.py:317-384), PyG's SAGEConv plus a
    second hop (:366).  Running its own ``forward`` over the stand-in SparseTensor / matmul pins what the restatement
    otherwise takes from PyG's documentation: aggregate = MEAN over stored neighbours (:336), edge values dropped
    (:379), no self loop, ``lin_l`` (with bias) on the aggregate, ``lin_r`` (no bias, :349) on the root, summed
    (:367-371).  ``out_1hop``: line :366 skipped (== SAGEConv); ``out_2hop``: the class as written."""
    conv_mod = sys.modules["torch_geometric.nn.conv"]
    rng = np.random.default_rng(11)
    graphs = {}
    n = 60
    M = np.triu(rng.random((n, n)) < 0.12, 1)
    A = (M | M.T).astype(np.float32)
    A[5, :] = 0
    A[:, 5] = 0                                    # isolated node: mean over nothing = 0
    graphs["plain"] = A
    W = A * rng.integers(1, 6, size=A.shape).astype(np.float32)
    graphs["weighted"] = np.maximum(W, W.T)        # collab-like values: must not change the mean
    for tag, Ad in graphs.items():
        torch.manual_seed(5)
        fin, fout = 10, 8
        conv = models.SAGEConv2(fin, fout)
        x = torch.randn(Ad.shape[0], fin)
        adj = sys.modules["torch_sparse"].SparseTensor(ssp.csr_matrix(Ad))
        with torch.no_grad():
            conv_mod.MessagePassing.SKIP_ROOTLESS = True
            out1 = conv(x, adj)
            conv_mod.MessagePassing.SKIP_ROOTLESS = False
            out2 = conv(x, adj)
        np.savez_compressed(os.path.join(OUT, f"sageconv_ref_{tag}.npz"), A=Ad, x=x.numpy(),
                            lin_l_weight=conv.lin_l.weight.detach().numpy(), lin_l_bias=conv.lin_l.bias.detach().numpy(),
                            lin_r_weight=conv.lin_r.weight.detach().numpy(), out_1hop=out1.numpy(), out_2hop=out2.numpy())
        print(f"sageconv_ref_{tag}: |out_1hop| max {out1.abs().max():.3f}, |out_2hop - out_1hop| max "
              f"{(out2 - out1).abs().max():.3f}")


def emit_configs(models):
    datasets = ["ddi", "collab", "ppa", "reddit", "twitch", "fb", "email"]
    names = ['sage', 'sage2', 'gcn', 'dea', 'dea_512', 'mlpcos', 'simplecos', 'adamic', 'simple', 'adamic_ogb',
             'resource_allocation', 'katz', 'ensemble_gcn_sage']
    keys = ["num_layers", "hidden_channels", "dropout", "batch_size", "lr", "epochs", "use_feature",
            "use_learnable_embedding"]
    table = {}
    for d in datasets:
        for m in names:
            args = argparse.Namespace(dataset=d, model=m, **{k: None for k in keys})
            out = models.default_model_configs(args)
            table[f"{d}/{m}"] = {k: getattr(out, k) for k in keys}
    # CLI override wins when not None (models.py:774-778); heuristics still force the flags (:783-785)
    args = argparse.Namespace(dataset="collab", model="simple", num_layers=5, hidden_channels=32, dropout=0.1,
                              batch_size=77, lr=0.5, epochs=3, use_feature=True, use_learnable_embedding=True)
    out = models.default_model_configs(args)
    table["override:collab/simple"] = {k: getattr(out, k) for k in keys}
    with open(os.path.join(OUT, "model_configs.json"), "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    lp2 = models.LinkPredictor(4, 4, 1, 2, 0.0)
    lp3 = models.LinkPredictor(4, 4, 1, 3, 0.0)
    mlp = models.MLP(4, 4, 1, 3, 0.0)
    with open(os.path.join(OUT, "state_dict_keys.json"), "w") as f:
        json.dump({"LinkPredictor_L2": list(lp2.state_dict()), "LinkPredictor_L3": list(lp3.state_dict()),
                   "MLP_L3": list(mlp.state_dict())}, f, indent=1)
    print(f"model_configs: {len(table)} entries")


def main():
    os.makedirs(OUT, exist_ok=True)
    install_stubs()
    import adamic_utils
    import train_and_eval
    import models
    adamic_utils.tqdm = lambda x: x
    train_and_eval.tqdm = lambda x: x
    emit_pairs(adamic_utils, train_and_eval)
    emit_linkpred(models)
    emit_gnn_stack(models)
    emit_sageconv_witness(models)
    emit_configs(models)


if __name__ == "__main__":
    main()
