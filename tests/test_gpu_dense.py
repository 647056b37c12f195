"""GPU parity of the dense / SpMM kernels through the C ABI vs the oracle (float32, tolerances stated)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu


def _close(a, b, rtol=1e-5, scale=None):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    scale = scale if scale is not None else max(1.0, float(np.abs(b).max()))
    return float(np.abs(a - b).max()) <= rtol * scale


@pytest.mark.parametrize("m,n,k", [(128, 128, 32), (300, 256, 384), (1000, 256, 58), (77, 40, 19), (4096, 256, 256),
                                   (5, 1, 7)])
def test_gemm_vs_float64(eps, dev, m, n, k):
    """eps_gemm_f32 (f32-input MFMA == fmaf chain): error vs float64 bounded by k * eps * sum|a||b|."""
    g = torch.Generator().manual_seed(m * 7 + n + k)
    a = torch.randn(m, k, generator=g)
    b = torch.randn(n, k, generator=g)
    bias = torch.randn(n, generator=g)
    ref = a.double() @ b.double().t() + bias.double()
    out = eps.ops.gemm(a.to(dev), b.to(dev), bias=bias.to(dev))
    bound = (a.abs().double() @ b.abs().double().t() + bias.abs().double()) * (k + 2) * 2.0 ** -24
    assert bool(((out.cpu().double() - ref).abs() <= bound + 1e-30).all())
    out_r = eps.ops.gemm(a.to(dev), b.to(dev), bias=bias.to(dev), relu=True)
    assert torch.equal(out_r, out.clamp_min(0))
    # accumulate: C += A B^T (SAGEConv lin_r added onto lin_l)
    acc = eps.ops.gemm(a.to(dev), b.to(dev), out=out.clone(), accumulate=True)
    assert _close(acc.cpu().numpy(), (2 * ref - bias.double()).numpy(), rtol=2e-5, scale=float(bound.max() * 2 ** 24 / (k + 2)))


def test_gemm_strided_rows(eps, dev):
    g = torch.Generator().manual_seed(5)
    big = torch.randn(200, 100, generator=g).to(dev)
    a = big[:, :58]  # lda = 100, K = 58 (ppa feature width): non-contiguous view with row stride
    assert a.stride(0) == 100
    b = torch.randn(64, 58, generator=g).to(dev)
    lib_out = torch.empty(200, 64, device=dev)
    import ctypes
    from eps_amd import _lib
    rc = _lib.load().eps_gemm_f32(ctypes.c_void_p(a.data_ptr()), 100, ctypes.c_void_p(b.data_ptr()), 58, None, 0, 0,
                                  ctypes.c_void_p(lib_out.data_ptr()), 64, 200, 64, 58,
                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    ref = a.double() @ b.double().t()
    assert _close(lib_out.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, scale=float(ref.abs().max()))


@pytest.mark.parametrize("f", [256, 64, 37, 300])
@pytest.mark.parametrize("mode", ["sum", "sum_unit", "mean"])
def test_spmm_vs_oracle(eps, oracle, dev, f, mode):
    import scipy.sparse as ssp
    rng = np.random.default_rng(f)
    n = 3000
    A = ssp.random(n, n, density=0.01, random_state=rng, dtype=np.float32, format="csr")
    A.data[:] = rng.integers(1, 4, A.nnz)
    A[5, :] = 0; A.eliminate_zeros()          # an empty row
    hub = np.zeros(n, np.float32); hub[rng.choice(n, 900, replace=False)] = 1
    A = ssp.vstack([A[:7], ssp.csr_matrix(hub), A[8:]]).tocsr(); A.sort_indices()   # one long row (> 64 * 4)
    x = rng.standard_normal((n, f)).astype(np.float32)
    bias = rng.standard_normal(f).astype(np.float32)
    val = None if mode != "sum" else A.data
    ref = oracle.spmm_csr(A.indptr, A.indices, val, x, bias=bias, relu=True, mean=(mode == "mean"))
    rp = torch.from_numpy(A.indptr.astype(np.int64)).to(dev)
    ci = torch.from_numpy(A.indices.astype(np.int32)).to(dev)
    vv = None if val is None else torch.from_numpy(val).to(dev)
    out = eps.ops.spmm_csr(rp, ci, vv, torch.from_numpy(x).to(dev), bias=torch.from_numpy(bias).to(dev), relu=True,
                           mean=(mode == "mean"))
    # sequential ascending-neighbour float32 accumulation on both sides (fmaf vs mul+add differ by <= 1 ulp/term)
    rowabs = oracle.spmm_csr(A.indptr, A.indices, val, np.abs(x), mean=(mode == "mean"))
    assert float(np.abs(out.cpu().numpy() - ref).max()) <= 1e-5 * max(1.0, float(rowabs.max()))
    assert out[5].abs().max().item() == pytest.approx(float(np.maximum(bias, 0).max()), rel=1e-6)


def test_gcn_norm_vs_oracle(eps, oracle, dev):
    d = np.load(os.path.join(GOLDEN, "pairs_collab_like.npz"))
    rp, ci, va = oracle.with_self_loops(d["rowptr"], d["col"], d["val"], 1.0)
    ref = oracle.gcn_norm_values(rp, ci, va)
    n = len(rp) - 1
    g = eps.CSRGraph(torch.from_numpy(d["rowptr"]), torch.from_numpy(d["col"]), torch.from_numpy(d["val"]), n, n).to(dev)
    gn = g.gcn_normalized()
    assert np.array_equal(gn.rowptr.cpu().numpy(), rp) and np.array_equal(gn.col.cpu().numpy(), ci)
    assert rel_err(gn.val.cpu().numpy(), ref) <= 1e-6


@pytest.mark.parametrize("tag", ["H256_L2", "H256_L3", "H64_L2", "H8_L3"])
def test_decode_golden(eps, dev, tag):
    """eps_mlp_decode vs the imported reference's LinkPredictor output (probabilities AND logits:
    sigmoid flattens relative error, so the logit is the sharper check)."""
    d = np.load(os.path.join(GOLDEN, f"linkpred_{tag}.npz"))
    L = sum(1 for k in d.files if k.startswith("w"))
    ws = [torch.from_numpy(d[f"w{i}"]).to(dev) for i in range(L)]
    bs = [torch.from_numpy(d[f"b{i}"]).to(dev) for i in range(L)]
    h = torch.from_numpy(d["h"]).to(dev)
    u = torch.from_numpy(d["edges"][0]).to(dev); v = torch.from_numpy(d["edges"][1]).to(dev)
    prob = eps.ops.mlp_decode(h, u, v, ws, bs)
    logit = eps.ops.mlp_decode(h, u, v, ws, bs, apply_sigmoid=False)
    assert rel_err(prob.cpu().numpy(), d["prob"]) <= 1e-5
    assert float(np.abs(logit.cpu().numpy() - d["logit"]).max()) <= 1e-5 * max(1.0, float(np.abs(d["logit"]).max()))


@pytest.mark.parametrize("H,L,E", [(256, 3, 100_003), (256, 2, 64), (128, 3, 5000), (32, 1, 999), (96, 4, 1000), (16, 2, 777), (100, 3, 500)])
def test_decode_vs_oracle_seeded(eps, oracle, dev, H, L, E):
    g = torch.Generator().manual_seed(H + L)
    n = 2000
    h = torch.randn(n, H, generator=g)
    ws = [torch.randn(H if i < L - 1 else 1, H, generator=g) / H ** 0.5 for i in range(L)]
    bs = [torch.randn(H if i < L - 1 else 1, generator=g) * 0.1 for i in range(L)]
    u = torch.randint(0, n, (E,), generator=g, dtype=torch.int32)
    v = torch.randint(0, n, (E,), generator=g, dtype=torch.int32)
    Eo = min(E, 3000)  # the float64 oracle is O(E * H^2): check a prefix and the ragged tail
    sel = np.r_[0:Eo - 100, E - 100:E]
    logit_o, prob_o = oracle.mlp_decode(h.numpy(), u.numpy()[sel], v.numpy()[sel], [w.numpy() for w in ws],
                                        [b.numpy() for b in bs])
    dw = [w.to(dev) for w in ws]; db = [b.to(dev) for b in bs]
    prob = eps.ops.mlp_decode(h.to(dev), u.to(dev), v.to(dev), dw, db).cpu().numpy()
    logit = eps.ops.mlp_decode(h.to(dev), u.to(dev), v.to(dev), dw, db, apply_sigmoid=False).cpu().numpy()
    assert rel_err(prob[sel], prob_o) <= 1e-5
    assert float(np.abs(logit[sel] - logit_o).max()) <= 2e-5 * max(1.0, float(np.abs(logit_o).max()))


def test_pack_keys_order(eps, dev):
    """Declared tie rule: score descending, then candidate id ascending."""
    g = torch.Generator().manual_seed(0)
    score = torch.randint(0, 50, (100000,), generator=g).float()
    score[::7] = -score[::7]; score[3] = 0.0; score[4] = -0.0
    keys = eps.ops.pack_keys(score.to(dev), id_base=1000)
    order = torch.argsort(keys, descending=True)
    ref = torch.sort(score, descending=True, stable=True).indices
    assert torch.equal(order.cpu(), ref)
    s2, ids = eps.ops.unpack_keys(keys)
    assert torch.equal(s2.cpu(), score + 0.0) and torch.equal(ids.cpu(), torch.arange(100000) + 1000)


def test_full_size_properties_spmm_decode(eps, dev):
    """Size-independent checks at the BASELINE scale (ppa-like graph, F = H = 256), no oracle run:
    SpMM of an all-ones matrix = row sums (exact); mean mode of ones = 1 (0 for empty rows); the GCN-normalised
    adjacency maps sqrt(deg+1) to itself; linearity; decode is symmetric in (u,v) bit for bit and stays in (0,1)."""
    from eps_amd import synth
    g = synth.ppa_like(seed=3, device=dev)
    n = g.n_rows
    ones = torch.ones(n, 256, device=dev)
    y = eps.ops.spmm_csr(g.rowptr, g.col, None, ones)
    deg = g.degree().float()
    assert torch.equal(y[:, 0], deg) and torch.equal(y[:, 255], deg)
    ym = eps.ops.spmm_csr(g.rowptr, g.col, None, ones, mean=True)
    assert torch.equal(ym[:, 7], (deg > 0).float())
    gn = g.gcn_normalized()
    s = torch.sqrt(deg + 1.0)                                  # D^1/2 1 is the fixed vector of D^-1/2 (A+I) D^-1/2
    x = s[:, None].repeat(1, 4).contiguous()
    z = eps.ops.spmm_csr(gn.rowptr, gn.col, gn.val, x)
    # sequential float32 accumulation (the reference kernel's order): error grows like row length x 2^-24
    assert bool(((z[:, 0] - s).abs() <= 2.0 ** -23 * (deg + 2.0) * s).all())
    gen = torch.Generator(device=dev).manual_seed(0)
    a = torch.randn(n, 64, generator=gen, device=dev)
    b = torch.randn(n, 64, generator=gen, device=dev)
    lhs = eps.ops.spmm_csr(gn.rowptr, gn.col, gn.val, a + b)
    rhs = eps.ops.spmm_csr(gn.rowptr, gn.col, gn.val, a) + eps.ops.spmm_csr(gn.rowptr, gn.col, gn.val, b)
    assert float((lhs - rhs).abs().max()) <= 1e-4
    h = torch.randn(n, 256, generator=gen, device=dev)
    ws = [torch.randn(256, 256, generator=gen, device=dev) / 16, torch.randn(256, 256, generator=gen, device=dev) / 16,
          torch.randn(1, 256, generator=gen, device=dev) / 16]
    bs = [torch.randn(256, generator=gen, device=dev), torch.randn(256, generator=gen, device=dev), torch.randn(1, generator=gen, device=dev)]
    E = 3_000_001
    u = torch.randint(0, n, (E,), generator=gen, device=dev, dtype=torch.int32)
    v = torch.randint(0, n, (E,), generator=gen, device=dev, dtype=torch.int32)
    p1 = eps.ops.mlp_decode(h, u, v, ws, bs)
    p2 = eps.ops.mlp_decode(h, v, u, ws, bs)
    assert torch.equal(p1, p2)                                  # Hadamard product commutes exactly
    assert float(p1.min()) >= 0.0 and float(p1.max()) <= 1.0 and p1.isfinite().all()
    # permuting the edge list permutes the outputs (tiles / workgroups are independent)
    perm = torch.randperm(E, generator=gen, device=dev)
    assert torch.equal(eps.ops.mlp_decode(h, u[perm].contiguous(), v[perm].contiguous(), ws, bs), p1[perm])


def test_full_size_properties_collab_shape(eps, dev, monkeypatch):
    """BASELINE configs[1] at FULL size (collab stand-in: N = 235,868, summed integer multi-edge weights, 128 features +
    256-d embedding = 384 -> H = 256, L = 3), size-independent checks, no oracle run: the SpMM WITH stored values of an
    all-ones matrix = the weighted row sums (exact: small integers); the GCN-normalised weighted adjacency fixes
    sqrt(weighted degree + 1); the K = 384 layer GEMM against float64 on sampled rows; a full GCN forward (finite, right shape);
    the fused decode is symmetric in (u, v) bit for bit; and the weighted threshold scan's top
    proposals equal the fused expansion's on a block of columns."""
    import argparse
    from eps_amd import candidates, datasets, models, scan
    from eps_amd.heuristics import node_weight_table
    monkeypatch.delenv("EPS_SYNTH_SCALE", raising=False)
    args = models.default_model_configs(argparse.Namespace(dataset="collab", model="gcn", synthetic=True, num_layers=None,
                                                           hidden_channels=None, dropout=None, batch_size=None, lr=None, epochs=None,
                                                           use_feature=None, use_learnable_embedding=None))
    _, _, _, data = datasets.get_data(args)
    data = data.to(dev)
    g = data.adj_t
    n = g.n_rows
    assert n == 235_868 and g.val is not None and data.x.shape[1] == 128
    wdeg = torch.zeros(n, device=dev).index_add_(0, g.row_index(), g.val)
    y = eps.ops.spmm_csr(g.rowptr, g.col, g.val, torch.ones(n, 256, device=dev))
    assert torch.equal(y[:, 0], wdeg) and torch.equal(y[:, 255], wdeg)
    gn = g.gcn_normalized()
    s = torch.sqrt(wdeg + 1.0)
    z = eps.ops.spmm_csr(gn.rowptr, gn.col, gn.val, s[:, None].repeat(1, 4).contiguous())
    assert bool(((z[:, 0] - s).abs() <= 2.0 ** -22 * (g.degree().float() + 2.0) * s).all())
    gen = torch.Generator(device=dev).manual_seed(4)
    a = torch.randn(n, 384, generator=gen, device=dev)
    wt = torch.randn(256, 384, generator=gen, device=dev) / 20
    bias = torch.randn(256, generator=gen, device=dev)
    c = eps.ops.gemm(a, wt, bias=bias, relu=True)
    rows = torch.randint(0, n, (512,), generator=gen, device=dev)
    want = torch.relu(a[rows].double() @ wt.double().t() + bias.double())
    assert float((c[rows].double() - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))
    torch.manual_seed(0)
    model = models.build_model(args, data, dev).eval()
    with torch.no_grad():
        h = model.embeddings(data.x, g)
        assert h.shape == (n, 256) and bool(h.isfinite().all())
        lp = model.linkpred
        ws = [l.weight.detach() for l in lp.lins]
        bs = [l.bias.detach() for l in lp.lins]
        E = 2_000_003
        u = torch.randint(0, n, (E,), generator=gen, device=dev, dtype=torch.int32)
        v = torch.randint(0, n, (E,), generator=gen, device=dev, dtype=torch.int32)
        p1 = eps.ops.mlp_decode(h, u, v, ws, bs)
        assert torch.equal(p1, eps.ops.mlp_decode(h, v, u, ws, bs)) and float(p1.min()) >= 0.0 and float(p1.max()) <= 1.0
    # the weighted threshold scan (AA, the published collab filter: submit_job.py:199-205) against the fused expansion
    w = node_weight_table(g, eps.ops.W_AA)
    assert scan.scan_usable(g, w)
    k = 150_000
    pairs, scores = scan.scan_topk(g, w, k)
    assert pairs.shape == (2, k) and bool((scores[:-1] >= scores[1:]).all())
    bar = float(scores[-1])
    lo, hi = 5000, 9000
    blk = candidates.expand_block_lazy(g, lo, hi, w, want_score=True)
    idx = torch.nonzero(blk.score > bar).squeeze(1)
    want_p = blk.select(idx)
    in_blk = (pairs[1] >= lo) & (pairs[1] < hi) & (scores > bar)
    got = pairs[:, in_blk]
    assert torch.equal(torch.sort(got[1] * n + got[0]).values, torch.sort(want_p[1] * n + want_p[0]).values)
    got_s = dict(zip((got[1] * n + got[0]).tolist(), scores[in_blk].tolist()))
    want_s = dict(zip((want_p[1] * n + want_p[0]).tolist(), blk.score[idx].tolist()))
    assert got_s == want_s                                     # the same fixed-point terms: bit-identical scores


def test_kth_largest_radix_select(eps, dev):
    """eps_kth_largest_f32 == the k-th entry of a descending sort, incl. ties, negatives, +-0, infinities and tiny arrays."""
    gen = torch.Generator(device=dev).manual_seed(3)
    for n in (1, 2, 257, 100_003, 3_000_000):
        x = torch.randn(n, generator=gen, device=dev)
        if n > 1000:
            x[::7] = x[3]                                  # many ties
            x[5], x[6], x[8], x[9] = 0.0, -0.0, float("inf"), float("-inf")
        ref = torch.sort(x, descending=True).values
        for k in sorted({1, 2, n // 3 + 1, n // 2 + 1, n}):
            if k > n:
                continue
            got = eps.ops.kth_largest(x, k)
            assert got.shape == (1,) and float(got) == float(ref[k - 1]), (n, k)
    # every value in ONE bin in every round (the wave-aggregated histogram path), and two values that differ in the last bit
    const = torch.full((200_001,), 1.5, device=dev)
    assert float(eps.ops.kth_largest(const, 1)) == 1.5 and float(eps.ops.kth_largest(const, 200_001)) == 1.5
    two = const.clone()
    two[::3] = torch.nextafter(torch.tensor(1.5), torch.tensor(2.0)).item()
    ref = torch.sort(two, descending=True).values
    for k in (1, int((two > 1.5).sum()), int((two > 1.5).sum()) + 1, 200_001):
        assert float(eps.ops.kth_largest(two, k)) == float(ref[k - 1]), k
    with pytest.raises(eps.EpsError):
        eps.ops.kth_largest(torch.zeros(4, device=dev), 5)


@pytest.mark.parametrize("n,k", [(1, 1), (5, 3), (1000, 7), (1000, 2000), (1000, 5000), (200_000, 40_001), (3_000_000, 1_000_000)])
def test_select_topk_matches_tensor_ops(eps, dev, n, k):
    """eps_select_topk (cut by radix select, mirror, two stable radix sorts) against the same selection in tensor ops: heavy
    ties (scores from a small set), k odd / above the list / below it."""
    from eps_amd import scan
    g = torch.Generator().manual_seed(n + k)
    u = torch.randint(0, 1 << 19, (n,), generator=g)
    v = u + 1 + torch.randint(0, 1 << 19, (n,), generator=g)
    keys = torch.unique((v << 32) | u)                                   # distinct unordered pairs, u < v
    vals = (torch.randint(0, 50, (keys.numel(),), generator=g).float() / 7).contiguous()
    want_k, want_v = scan.select_topk_torch(keys, vals, k)
    for bits in (32, 21):                                                 # ids <= 2^20: the key sort may skip the other bits
        got_k, got_v = eps.ops.select_topk(keys.to(dev), vals.to(dev), k, bits)
        assert torch.equal(got_k.cpu(), want_k) and torch.equal(got_v.cpu(), want_v)


def test_select_topk_random_sizes_and_id_widths(eps, dev):
    from eps_amd import scan
    g = torch.Generator().manual_seed(99)
    for trial in range(40):
        bits = int(torch.randint(3, 31, (1,), generator=g))
        n = int(torch.randint(1, 30000, (1,), generator=g))
        k = int(torch.randint(1, 3 * n + 2, (1,), generator=g))
        u = torch.randint(0, (1 << bits) - 1, (n,), generator=g)
        v = u + 1 + (torch.rand(n, generator=g) * ((1 << bits) - 1 - u).double()).long()
        v = torch.clamp(v, max=(1 << bits) - 1)
        ok = v > u
        keys = torch.unique((v[ok] << 32) | u[ok])
        if keys.numel() == 0:
            continue
        vals = (torch.randint(0, 9, (keys.numel(),), generator=g).float() - 3.5).contiguous()      # ties, negative scores too
        want_k, want_v = scan.select_topk_torch(keys, vals, k)
        for b in (bits, 32):
            got_k, got_v = eps.ops.select_topk(keys.to(dev), vals.to(dev), k, b)
            assert torch.equal(got_k.cpu(), want_k) and torch.equal(got_v.cpu(), want_v), (trial, bits, b, n, k)


@pytest.mark.parametrize("n,p", [(300, 0.2), (1000, 0.1), (1031, 0.04), (129, 0.5)])
def test_dense_common_neighbours_match_oracle_and_sparse_path(eps, oracle, dev, n, p):
    """csrc/dense_cn.hip (configs[0]: a dense graph's CN list = A A^T on the f32 MFMA, lower tiles only, + a masked read): the
    unordered candidates u < v of every column in ascending order with their counts -- bit-exact against the oracle's
    candidate set (filter.py:96-109 restated) and counts (models.py:536-542), and against the sparse list kernel."""
    import scipy.sparse as ssp
    from eps_amd import candidates, ops, scan
    rng = np.random.default_rng(n)
    M = ssp.random(n, n, density=p / 2, random_state=rng, format="csr")
    A = ((M + M.T) > 0).astype(np.float32).tocsr()
    A.setdiag(0)
    A.eliminate_zeros()
    A.sort_indices()
    g = eps.CSRGraph.from_scipy(A, device=dev)
    g.val = None
    assert candidates.dense_cn_suits(g)
    keys, vals = ops.dense_cn_candidates(g.rowptr, g.col, g.n_rows)
    pairs, _ = oracle.candidates_scipy(A)
    lower = pairs[:, 0] < pairs[:, 1]
    want_pairs = pairs[lower]                                    # column-major: v ascending, then u
    cnt = oracle.pair_scores(A.indptr.astype(np.int64), A.indices.astype(np.int32), None, None, want_pairs[:, 0], want_pairs[:, 1])[1]
    k = keys.cpu().numpy()
    assert np.array_equal(k >> 32, want_pairs[:, 1]) and np.array_equal(k & 0xFFFFFFFF, want_pairs[:, 0])
    assert np.array_equal(vals.cpu().numpy(), cnt.astype(np.float32))
    # both orientations, the reference's own order (filter.py:96-109: column-major, v then u), None for an asymmetric pattern
    dk, dvals = ops.dense_cn_candidates(g.rowptr, g.col, g.n_rows, directed=True, check_symmetric=True)
    cnt_all = oracle.pair_scores(A.indptr.astype(np.int64), A.indices.astype(np.int32), None, None, pairs[:, 0], pairs[:, 1])[1]
    dkn = dk.cpu().numpy()
    assert np.array_equal(dkn >> 32, pairs[:, 1]) and np.array_equal(dkn & 0xFFFFFFFF, pairs[:, 0])
    assert np.array_equal(dvals.cpu().numpy(), cnt_all.astype(np.float32))
    rws, rc = ops.dense_cn_candidates(g.rowptr, g.col, g.n_rows, directed=True, as_rows=True)
    assert torch.equal(rc, dvals) and np.array_equal(rws.cpu().numpy(), np.stack([pairs[:, 0], pairs[:, 1], cnt_all], 1).astype(np.float32))
    B = A.tolil(copy=True)
    i, j = np.argwhere(A.toarray() == 0)[5]
    if i != j:
        B[i, j] = 1.0
        gb = eps.CSRGraph.from_scipy(B.tocsr(), device=dev)
        assert ops.dense_cn_candidates(gb.rowptr, gb.col, gb.n_rows, directed=True, check_symmetric=True) is None
    # the sparse list kernel: same pairs, same bits
    ones = torch.ones(n, dtype=torch.float32, device=dev)
    r = ops.expand_unit(g.rowptr, g.col, ones, n, 0, n, scan.max_degree(g), scan.window_splits(g), revpos=scan.reverse_positions(g))
    sk = (r.pairs[1].to(torch.int64) << 32) | r.pairs[0].to(torch.int64)
    o = torch.argsort(sk)
    assert torch.equal(sk[o], keys) and torch.equal(r[4][o], vals)
    # the triangle flag of the product: tiles above the diagonal are left alone
    a = ops.dense_adjacency(g.rowptr, g.col, n)
    c = torch.full_like(a, -7.0)
    ops.gemm(a, a, out=c, lower_only=True)
    full = ops.gemm(a, a)
    np_ = a.shape[0]
    tile = torch.arange(np_, device=dev) // 128
    low = tile.unsqueeze(1) >= tile.unsqueeze(0)
    assert torch.equal(c[low], full[low]) and bool((c[~low] == -7.0).all())
