"""CPU: pin the oracle (C restatement + SciPy mirrors) to the golden vectors produced by the
imported reference (oracle/gen_golden.py).  CN exact; AA/RA <= 1e-6 relative (the GPU gate is 1e-5)."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as ssp

from conftest import GOLDEN, golden_pair_files, rel_err


@pytest.mark.parametrize("path", golden_pair_files(), ids=lambda p: os.path.basename(p)[6:-4])
def test_pair_scores_match_reference(oracle, path):
    d = np.load(path)
    rp, col, val, pairs = d["rowptr"], d["col"], d["val"], d["pairs"]
    n = len(rp) - 1
    cs = oracle.col_sums(rp, col, val, n)
    w_aa = oracle.node_weights(cs, oracle.W_AA)
    w_ra = oracle.node_weights(cs, oracle.W_RA)
    cnt, cn, aa = oracle.pair_scores(rp, col, val, w_aa, pairs[0], pairs[1])
    _, _, ra = oracle.pair_scores(rp, col, val, w_ra, pairs[0], pairs[1])
    assert np.array_equal(cn, d["cn"]), "CN must be bit-exact"
    unit = bool((val == 1).all())
    if unit:
        assert np.array_equal(cnt.astype(np.float32), d["cn"])
    assert rel_err(aa, d["aa"]) <= 1e-6
    assert rel_err(ra, d["ra_f32"]) <= 1e-6
    # filter.py:130-141 flavour: integer adjacency -> float64 math -> FloatTensor
    cs64 = oracle.col_sums(rp, col, None, n).astype(np.float64)
    w64 = oracle.node_weights(cs64, oracle.W_RA)
    _, ra64 = oracle.pair_scores_f64(rp, col, None, w64, pairs[0], pairs[1])
    assert rel_err(ra64.astype(np.float32), d["ra_i64"]) <= 1e-6


@pytest.mark.parametrize("path", golden_pair_files(), ids=lambda p: os.path.basename(p)[6:-4])
def test_scipy_mirror_is_bit_exact(oracle, path):
    """The SciPy mirror (what bench.py times as the reference CPU path) reproduces the imported
    reference bit for bit."""
    d = np.load(path)
    n = len(d["rowptr"]) - 1
    A = ssp.csr_matrix((d["val"], d["col"], d["rowptr"]), shape=(n, n))
    pairs = d["pairs"].astype(np.int64)
    with np.errstate(divide="ignore"):
        assert np.array_equal(oracle.scipy_AA(A, pairs), d["aa"])
        assert np.array_equal(oracle.scipy_RA(A, pairs.T, batch_size=1024), d["ra_f32"])
    assert np.array_equal(oracle.scipy_CN(A, pairs), d["cn"])


@pytest.mark.parametrize("tag", ["H256_L2", "H256_L3", "H8_L3", "H64_L2"])
def test_decode_matches_reference(oracle, tag):
    d = np.load(os.path.join(GOLDEN, f"linkpred_{tag}.npz"))
    L = sum(1 for k in d.files if k.startswith("w"))
    ws = [d[f"w{i}"] for i in range(L)]
    bs = [d[f"b{i}"] for i in range(L)]
    logit, prob = oracle.mlp_decode(d["h"], d["edges"][0], d["edges"][1], ws, bs)
    assert rel_err(prob, d["prob"]) <= 1e-5
    assert np.abs(logit - d["logit"]).max() <= 1e-5 * max(1.0, np.abs(d["logit"]).max())


@pytest.mark.parametrize("kind,L", [("gcn", 2), ("gcn", 3), ("sage", 2), ("sage", 3)])
def test_gnn_stack_structure(oracle, kind, L):
    """Layer-loop structure (ReLU placement, no activation after the last layer, emb-first concat,
    decode) as executed by the reference's own GCN/SAGE/LinkGNN.forward over the oracle's conv."""
    d = np.load(os.path.join(GOLDEN, f"gnn_stack_{kind}_L{L}.npz"))
    sd = {k[4:]: d[k] for k in d.files if k.startswith("sd::")}
    A = d["A"]
    x = np.concatenate([sd["emb.weight"], d["x"]], 1)       # models.py:503-504: embedding first
    if kind == "gcn":
        h = oracle.gcn_dense_forward(A, x, [sd[f"gnn.convs.{i}.weight"] for i in range(L)],
                                     [sd[f"gnn.convs.{i}.bias"] for i in range(L)])
    else:
        h = oracle.sage_dense_forward(A, x, [sd[f"gnn.convs.{i}.lin_l.weight"] for i in range(L)],
                                      [sd[f"gnn.convs.{i}.lin_l.bias"] for i in range(L)],
                                      [sd[f"gnn.convs.{i}.lin_r.weight"] for i in range(L)])
    assert np.abs(h - d["h"]).max() <= 1e-4 * max(1.0, np.abs(d["h"]).max())
    # CSR float32 restatement agrees with the dense float64 formula
    Acsr = ssp.csr_matrix(A)
    Acsr.sort_indices()
    if kind == "gcn":
        h32 = oracle.gcn_forward_csr(Acsr.indptr, Acsr.indices, Acsr.data, x,
                                     [sd[f"gnn.convs.{i}.weight"] for i in range(L)],
                                     [sd[f"gnn.convs.{i}.bias"] for i in range(L)])
    else:
        h32 = oracle.sage_forward_csr(Acsr.indptr, Acsr.indices, x,
                                      [sd[f"gnn.convs.{i}.lin_l.weight"] for i in range(L)],
                                      [sd[f"gnn.convs.{i}.lin_l.bias"] for i in range(L)],
                                      [sd[f"gnn.convs.{i}.lin_r.weight"] for i in range(L)])
    assert np.abs(h32 - h).max() <= 1e-4 * max(1.0, np.abs(h).max())
    ws = [sd[f"linkpred.lins.{i}.weight"] for i in range(L)]
    bs = [sd[f"linkpred.lins.{i}.bias"] for i in range(L)]
    _, prob = oracle.mlp_decode(d["h"], d["edges"][0], d["edges"][1], ws, bs)
    assert rel_err(prob, d["prob"]) <= 1e-5


def test_gcn_norm_matches_the_in_tree_witness(oracle):
    """The GCN normalisation against the only text of it the reference holds -- the commented-out pre-computation at
    email_data/mlp_common.py:274-280 (= reddit/mlp_common.py:280-286): ``adj_t.set_diag()`` (every diagonal entry := 1),
    ``deg = adj_t.sum(dim=1)``, ``deg.pow(-0.5)``, ``inf := 0``, ``dis.view(-1, 1) * adj_t * dis.view(1, -1)`` -- evaluated
    here exactly as written, on dense torch tensors, incl. weighted entries, an existing diagonal entry and an empty row."""
    import torch
    rng = np.random.default_rng(3)
    n = 23
    A = (rng.random((n, n)) < 0.2) * rng.integers(1, 4, (n, n))
    A = (A + A.T).astype(np.float32)
    A[5, :] = 0; A[:, 5] = 0                    # an isolated node (its row is the self loop alone)
    A[7, 7] = 3.0                               # a stored diagonal entry: set_diag overwrites it with 1
    adj_t = torch.from_numpy(A.copy())
    adj_t.fill_diagonal_(1.0)                                          # adj_t = data.adj_t.set_diag()
    deg = adj_t.sum(dim=1).to(torch.float)                             # deg = adj_t.sum(dim=1).to(torch.float)
    deg_inv_sqrt = deg.pow(-0.5)                                       # deg_inv_sqrt = deg.pow(-0.5)
    deg_inv_sqrt[deg_inv_sqrt == float('inf')] = 0                     # deg_inv_sqrt[deg_inv_sqrt == float('inf')] = 0
    want = deg_inv_sqrt.view(-1, 1) * adj_t * deg_inv_sqrt.view(1, -1)
    got = oracle.gcn_norm_dense(A)
    assert np.abs(got - want.numpy().astype(np.float64)).max() <= 1e-6
    # ... and the CSR float32 restatement the HIP path is tested against uses the same matrix
    Acsr = ssp.csr_matrix(A)
    Acsr.sort_indices()
    x = rng.standard_normal((n, 4)).astype(np.float32)
    W = np.eye(4, dtype=np.float32)
    h = oracle.gcn_forward_csr(Acsr.indptr, Acsr.indices, Acsr.data, x, [W], [np.zeros(4, np.float32)])
    assert np.abs(h - got @ x.astype(np.float64)).max() <= 1e-5


@pytest.mark.parametrize("tag", ["plain", "weighted"])
def test_sage_conv_matches_reference_witness(oracle, tag):
    """The SAGE conv arithmetic against the reference's OWN conv code: models.SAGEConv2.forward (models.py:358-384) run
    by oracle/gen_golden.py with its extra hop (:366) skipped (``out_1hop`` == PyG SAGEConv) and as written (``out_2hop``).
    Pins mean-over-stored-neighbours, values ignored, no self loop, lin_l bias / lin_r no bias, for the dense float64
    formula and for the CSR float32 restatement (C spmm) the GPU tests compare against."""
    d = np.load(os.path.join(GOLDEN, f"sageconv_ref_{tag}.npz"))
    A, x = d["A"], d["x"]
    wl, bl, wr = [d["lin_l_weight"]], [d["lin_l_bias"]], [d["lin_r_weight"]]
    h = oracle.sage_dense_forward(A, x, wl, bl, wr)
    assert np.abs(h - d["out_1hop"]).max() <= 1e-5 * max(1.0, np.abs(d["out_1hop"]).max())
    Acsr = ssp.csr_matrix(A)
    Acsr.sort_indices()
    h32 = oracle.sage_forward_csr(Acsr.indptr, Acsr.indices, x, wl, bl, wr)
    assert np.abs(h32 - d["out_1hop"]).max() <= 1e-5 * max(1.0, np.abs(d["out_1hop"]).max())
    # the class as written: the aggregate of the aggregate goes through lin_l, the root term is unchanged
    agg1 = oracle.spmm_csr(Acsr.indptr, Acsr.indices, None, x.astype(np.float32), mean=True)
    agg2 = oracle.spmm_csr(Acsr.indptr, Acsr.indices, None, agg1, mean=True)
    out2 = agg2 @ wl[0].T + bl[0] + x @ wr[0].T
    assert np.abs(out2 - d["out_2hop"]).max() <= 1e-5 * max(1.0, np.abs(d["out_2hop"]).max())
    if tag == "weighted":      # edge values never enter the mean (models.py:379 set_value(None))
        p = np.load(os.path.join(GOLDEN, "sageconv_ref_plain.npz"))
        assert np.array_equal(p["out_1hop"], d["out_1hop"]) and np.array_equal((p["A"] != 0), (A != 0))


def test_hits_at_k(oracle):
    rng = np.random.default_rng(0)
    pos = rng.random(1000).astype(np.float32)
    neg = rng.random(500).astype(np.float32)
    for k in (1, 10, 100, 500):
        kth = np.sort(neg)[::-1][k - 1]
        assert oracle.hits_at_k(pos, neg, k) == pytest.approx(float((pos > kth).mean()))
    assert oracle.hits_at_k(pos, neg, 501) == 1.0           # fewer negatives than K -> 1.0
    # strict '>' : a positive tied with the K-th negative is NOT a hit
    assert oracle.hits_at_k(np.array([0.5], np.float32), np.array([0.5, 0.1], np.float32), 1) == 0.0
    # permutation invariance
    assert oracle.hits_at_k(pos[::-1].copy(), rng.permutation(neg), 10) == oracle.hits_at_k(pos, neg, 10)


def test_candidates_order_and_cn(oracle):
    """filter.py:96-109 restated: column-major order, both directions, no diagonal, no known edges;
    the discarded A^2 value equals CN(u,v)."""
    d = np.load(os.path.join(GOLDEN, "pairs_er500.npz"))
    n = len(d["rowptr"]) - 1
    A = ssp.csr_matrix((d["val"], d["col"], d["rowptr"]), shape=(n, n))
    pairs, a2 = oracle.candidates_scipy(A)
    key = pairs[:, 1] * n + pairs[:, 0]
    assert np.all(np.diff(key) > 0), "column-major, strictly ascending"
    assert not np.any(pairs[:, 0] == pairs[:, 1])
    assert A[pairs[:, 0], pairs[:, 1]].sum() == 0
    s = set(map(tuple, pairs.tolist()))
    assert all((v, u) in s for u, v in list(s)[:2000])
    cnt, _, _ = oracle.pair_scores(d["rowptr"], d["col"], None, None, pairs[:, 0], pairs[:, 1])
    assert np.array_equal(cnt, a2.astype(np.int32))
    assert cnt.min() >= 1


def test_candidates_column_slice_equals_full(oracle):
    """The column-block form used for graphs too wide for a full A @ A agrees with the full restatement."""
    rng = np.random.default_rng(4)
    n = 300
    M = np.triu(rng.random((n, n)) < 0.03, 1)
    A = ssp.csr_matrix((M | M.T).astype(np.float32))
    full, vals = oracle.candidates_scipy(A)
    for lo, hi in ((0, n), (0, 1), (17, 140), (250, 300)):
        p, v = oracle.candidates_scipy_columns(A, lo, hi)
        m = (full[:, 1] >= lo) & (full[:, 1] < hi)
        assert np.array_equal(p, full[m]) and np.array_equal(v, vals[m])


def test_model_configs_fixture_shape():
    with open(os.path.join(GOLDEN, "model_configs.json")) as f:
        table = json.load(f)
    assert table["ddi/gcn"]["batch_size"] == 65536 and table["ddi/simple"]["batch_size"] == 1024
    assert table["ppa/gcn"]["hidden_channels"] is None      # no ppa block in the reference
