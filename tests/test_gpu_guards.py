"""GPU: range / determinism guards of the weight prologue and of the fused fixed-point paths:
column sums past 2^24, and adjacency weights so large that a fused score would leave the accumulators' range."""
import numpy as np
import pytest
import scipy.sparse as ssp
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def test_column_sums_exact_past_2_24(eps, oracle, dev):
    """Integer-valued weights whose column sums exceed 2^24 (float32 atomics stop being exact there): the float64 sums
    are exact, the float32 table is their single rounding, the float64 RA table is exactly 1/sum, and two runs agree."""
    from eps_amd.heuristics import node_weight_table
    rng = np.random.default_rng(5)
    n = 3000
    r = rng.integers(0, n, 60000)
    c = rng.integers(0, 40, 60000)                       # 40 hub columns collect ~1500 entries each
    w = rng.integers(20000, 70000, 60000).astype(np.float32)
    keep = r != c
    A = ssp.coo_matrix((w[keep], (r[keep], c[keep])), shape=(n, n)).tocsr()
    A = (A + A.T).tocsr()
    A.sum_duplicates(); A.sort_indices()
    exact = np.asarray(A.astype(np.float64).sum(0)).reshape(-1)
    assert exact.max() > 2 ** 25 and np.all(exact == np.round(exact))
    g = eps.CSRGraph.from_scipy(A, device=dev)
    s64 = eps.ops.col_sums(g.rowptr, g.col, g.val, n, f64=True).cpu().numpy()
    s32 = eps.ops.col_sums(g.rowptr, g.col, g.val, n).cpu().numpy()
    assert np.array_equal(s64, exact), "float64 column sums are exact for integer-valued weights"
    assert np.array_equal(s32, exact.astype(np.float32))
    assert np.array_equal(eps.ops.col_sums(g.rowptr, g.col, g.val, n, f64=True).cpu().numpy(), s64)
    ra = node_weight_table(g, eps.ops.W_RA, f64=True).cpu().numpy()
    with np.errstate(divide="ignore"):
        want = 1.0 / exact
    want[np.isinf(want)] = 0
    assert np.array_equal(ra, want)
    # the float32 AA table follows the reference's float32 prologue (adamic_utils.py:15-16) on the rounded sums
    aa = node_weight_table(g, eps.ops.W_AA).cpu().numpy()
    assert rel_err(aa, oracle.node_weights(exact.astype(np.float32), oracle.W_AA)) <= 1e-6


def _heavy_collab_like(seed, n=400, m=6000, lo=1000, hi=10000):
    rng = np.random.default_rng(seed)
    r, c = rng.integers(0, n, m), rng.integers(0, n, m)
    w = rng.integers(lo, hi, m).astype(np.float32)
    keep = r != c
    A = ssp.coo_matrix((w[keep], (r[keep], c[keep])), shape=(n, n)).tocsr()
    A = (A + A.T).tocsr()
    A.sum_duplicates(); A.sort_indices()
    return A.astype(np.float32)


def test_fused_score_bound_sends_heavy_weights_to_the_pair_kernels(eps, oracle, dev, tmp_path, monkeypatch):
    """collab-like weights of 10^3..10^4 (rank.py:32-35 sums multi-edges): A[u,w]*A[v,w]*mult[w] reaches 10^7 > 2^23, the
    range of the fused kernels' 2^-40 fixed point.  The bound check must see that, the filter stage must score such a
    graph with the float32 pair kernels instead, and the scores must match the oracle."""
    import argparse
    from eps_amd import candidates, filter_stage
    from eps_amd.heuristics import node_weight_table
    A = _heavy_collab_like(3)
    g = eps.CSRGraph.from_scipy(A, device=dev)
    wt = node_weight_table(g, eps.ops.W_AA)
    n = g.n_rows
    rp, col, val = A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data
    w = oracle.node_weights(oracle.col_sums(rp, col, val, n), oracle.W_AA)
    pairs, _ = oracle.candidates_scipy(A)
    _, truth = oracle.pair_scores_f64(rp, col, val, w.astype(np.float64), pairs[:, 0], pairs[:, 1])
    assert truth.max() > 2 ** 23, "the graph really crosses the fixed-point range"
    bound = candidates.fused_score_bound(g, wt)
    assert bound >= truth.max() and not candidates.fused_scores_fit(g, wt)
    # a light graph of the same shape stays fused
    Al = _heavy_collab_like(3, lo=1, hi=6)
    gl = eps.CSRGraph.from_scipy(Al, device=dev)
    assert candidates.fused_scores_fit(gl, node_weight_table(gl, eps.ops.W_AA))
    # the stage's block generator falls back to the pair kernels and is right
    args = argparse.Namespace(model="adamic_ogb")
    data = argparse.Namespace(adj_t=g, x=None)
    got_p, got_s = [], []
    for _, _, blk, sc in filter_stage.scored_blocks(args, None, data, None):
        assert isinstance(blk, torch.Tensor), "pair-kernel path yields explicit pair tensors"
        got_p.append(blk); got_s.append(sc)
    got_p, got_s = torch.cat(got_p, 1).cpu().numpy(), torch.cat(got_s).cpu().numpy()
    assert np.array_equal(got_p.T, pairs)
    assert rel_err(got_s, truth.astype(np.float32)) <= 1e-5


def test_scan_refuses_asymmetric_pattern(eps, dev):
    """The threshold scan's half scheme needs a symmetric pattern: an asymmetric one is refused (scan_available False,
    scan_topk raises) instead of being scored as something else."""
    from eps_amd import scan, synth
    from eps_amd.graph import CSRGraph
    g = synth.rmat_graph(10, 8, 3, dev)
    assert scan.is_symmetric(g) and scan.scan_available(g)
    row, col, _ = g.coo()
    keep = ~((row == row[5]) & (col == col[5]))                  # drop ONE direction of one edge
    h = CSRGraph.from_edge_index(torch.stack([row[keep], col[keep].long()]), None, sparse_sizes=(g.n_rows, g.n_cols))
    assert h.nnz() == g.nnz() - 1
    assert not scan.is_symmetric(h) and not scan.scan_available(h)
    with pytest.raises(eps.EpsError):
        scan.scan_topk(h, torch.ones(h.n_rows, device=dev), 10)


def test_scan_topk_refuses_unusable_k(eps, dev):
    from eps_amd import scan, synth
    g = synth.rmat_graph(9, 6, 3, dev)
    w = torch.ones(g.n_rows, device=dev)
    for k in (0, -3, scan.MAX_K + 1):
        with pytest.raises(eps.EpsError):
            scan.scan_topk(g, w, k)
    pairs, scores = scan.scan_topk(g, w, scan.MAX_K)            # more rows than candidates: all of them, sorted
    assert pairs.shape[1] == scores.numel() > 0 and bool((scores[:-1] >= scores[1:]).all())
