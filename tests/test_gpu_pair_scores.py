"""GPU parity: eps_pair_scores (through the C ABI) vs the golden vectors of the imported reference
and vs the oracle on seeded inputs.  CN/count bit-exact; AA/RA within 1e-5 relative."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_pair_files, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5  # north_star: AA / RA float scores within 1e-5 relative


def _graph(eps, d, dev, with_val=True):
    val = torch.from_numpy(d["val"]) if with_val and not bool((d["val"] == 1).all()) else None
    n = len(d["rowptr"]) - 1
    return eps.CSRGraph(torch.from_numpy(d["rowptr"]), torch.from_numpy(d["col"]), val, n, n).to(dev)


@pytest.mark.parametrize("path", golden_pair_files(), ids=lambda p: os.path.basename(p)[6:-4])
def test_golden_facades(eps, dev, path):
    """AA / resource_allocation / common_neighbors with the reference's signatures."""
    d = np.load(path)
    g = _graph(eps, d, dev)
    pairs = torch.from_numpy(d["pairs"].astype(np.int64))
    aa, ei = eps.AA(eps.get_A(g, g.n_rows), pairs)
    assert ei is pairs and aa.dtype == torch.float32 and aa.device.type == "cpu"
    assert rel_err(aa.numpy(), d["aa"]) <= TOL
    ra = eps.resource_allocation(g, pairs.t(), batch_size=1024)
    assert rel_err(ra.numpy(), d["ra_f32"]) <= TOL
    cn = eps.common_neighbors(g, pairs.to(dev))
    assert np.array_equal(cn.cpu().numpy(), d["cn"]), "CN must be bit-exact"


@pytest.mark.parametrize("path", golden_pair_files(), ids=lambda p: os.path.basename(p)[6:-4])
def test_golden_int64_adjacency_ra(eps, dev, path):
    """filter.py:130-141: SciPy int64 adjacency -> float64 math -> FloatTensor."""
    import scipy.sparse as ssp
    d = np.load(path)
    n = len(d["rowptr"]) - 1
    Ai = ssp.csr_matrix((np.ones(len(d["col"]), dtype=np.int64), d["col"], d["rowptr"]), shape=(n, n))
    ra = eps.resource_allocation(Ai, torch.from_numpy(d["pairs"].astype(np.int64)).t(), batch_size=8192)
    assert rel_err(ra.numpy(), d["ra_i64"]) <= TOL


def _random_graph(rng, n, avg_deg, weighted=False, hub=None):
    m = int(n * avg_deg / 2)
    # skewed endpoints: square of uniform -> low ids are hubs
    r = (rng.random(m) ** 2 * n).astype(np.int64)
    c = rng.integers(0, n, m)
    if hub:
        r = np.concatenate([r, np.zeros(hub, dtype=np.int64)])
        c = np.concatenate([c, rng.choice(n, hub, replace=False)])
    import scipy.sparse as ssp
    w = rng.integers(1, 4, len(r)).astype(np.float32) if weighted else np.ones(len(r), np.float32)
    keep = r != c
    A = ssp.coo_matrix((w[keep], (r[keep], c[keep])), shape=(n, n)).tocsr()
    A = (A + A.T).tocsr()
    A.sum_duplicates(); A.sort_indices()
    if not weighted:
        A.data[:] = 1
    return A.astype(np.float32)


@pytest.mark.parametrize("n,deg,weighted,hub", [(3000, 40, False, None), (3000, 40, True, None),
                                                (20000, 12, False, 15000),   # hub row >> LDS pass, in-place path
                                                (6000, 700, False, None)])   # rows > 1024: multi-pass staging
def test_vs_oracle_seeded(eps, oracle, dev, n, deg, weighted, hub):
    rng = np.random.default_rng(n + deg)
    A = _random_graph(rng, n, deg, weighted, hub)
    g = eps.CSRGraph.from_scipy(A, device=dev)
    E = 70001  # ragged: not a multiple of 64
    pairs = rng.integers(0, n, (2, E)).astype(np.int32)
    pairs[:, :500] = np.stack([np.zeros(500, np.int32), rng.integers(0, n, 500)])  # hub rows
    pairs[:, 500:600] = pairs[0, 500:600]                                            # self pairs
    rp, col = A.indptr.astype(np.int64), A.indices.astype(np.int32)
    val = A.data if weighted else None
    cs = oracle.col_sums(rp, col, val, n)
    w = oracle.node_weights(cs, oracle.W_AA)
    cnt_o, cn_o, ws_o = oracle.pair_scores(rp, col, val, w, pairs[0], pairs[1])
    u, v = torch.from_numpy(pairs[0]).to(dev), torch.from_numpy(pairs[1]).to(dev)
    from eps_amd.heuristics import node_weight_table
    wt = node_weight_table(g, eps.ops.W_AA)
    assert rel_err(wt.cpu().numpy(), w) <= 1e-6
    for grouped in (False, True):       # the column-run kernel must be right on an UNSORTED list too
        cnt, cn, ws = eps.ops.pair_scores(g.rowptr, g.col, g.val, wt, n, u, v, grouped=grouped)
        assert np.array_equal(cnt.cpu().numpy(), cnt_o)
        assert np.array_equal(cn.cpu().numpy(), cn_o)
        assert rel_err(ws.cpu().numpy(), ws_o) <= TOL
    assert cnt_o.max() > 0
    # sorted by v (the reference's candidate order): long runs -> auto-selects the column-run kernel
    order = np.lexsort((pairs[0], pairs[1] % 37))   # 37 distinct columns -> runs of ~1900 pairs
    ps = pairs[:, order].copy(); ps[1] = ps[1] % 37
    cnt_s, cn_s, ws_s = oracle.pair_scores(rp, col, val, w, ps[0], ps[1])
    us, vs = torch.from_numpy(ps[0]).to(dev), torch.from_numpy(ps[1]).to(dev)
    assert eps.ops.v_runs_are_long(vs) and not eps.ops.v_runs_are_long(v)
    cnt, cn, ws = eps.ops.pair_scores(g.rowptr, g.col, g.val, wt, n, us, vs)
    assert np.array_equal(cnt.cpu().numpy(), cnt_s) and np.array_equal(cn.cpu().numpy(), cn_s)
    # this list pairs the 15000-neighbour hub with itself: a 15000-term float32 sum, where the ORACLE's sequential
    # accumulation is itself ~7e-5 off.  Gate the kernel against the float64-accumulated value instead, and the
    # float32 oracle only where both agree (short sums).
    _, truth = oracle.pair_scores_f64(rp, col, val, w.astype(np.float64), ps[0], ps[1])
    assert rel_err(ws.cpu().numpy(), truth.astype(np.float32)) <= TOL
    short = cnt_s <= 512
    assert rel_err(ws.cpu().numpy()[short], ws_s[short]) <= TOL
    w64 = oracle.node_weights(cs.astype(np.float64), oracle.W_RA)
    _, ws64_o = oracle.pair_scores_f64(rp, col, val, w64, ps[0], ps[1])
    _, _, ws64 = eps.ops.pair_scores(g.rowptr, g.col, g.val, torch.from_numpy(w64).to(dev), n, us, vs, grouped=True)
    assert rel_err(ws64.cpu().numpy(), ws64_o) <= 1e-12


def test_grouped_hashed_bitmap_large_id_space(eps, oracle, dev):
    """N > 2^20 nodes: the LDS bitmap is hashed (w & mask) and every positive is verified in row v."""
    from eps_amd import synth
    g = synth.rmat_graph(scale=21, edge_factor=2, seed=11, device=dev)
    n = g.n_rows
    assert n > (1 << 20)
    rng = np.random.default_rng(0)
    deg = g.degree().cpu().numpy()
    hubs = np.argsort(-deg)[:8].astype(np.int32)                    # columns with large neighbourhoods
    u = rng.integers(0, n, 40000).astype(np.int32)
    u[:4000] = rng.choice(np.argsort(-deg)[:2000], 4000)            # some high-degree rows too
    v = np.repeat(hubs, 5000)
    rp, col = g.rowptr.cpu().numpy(), g.col.cpu().numpy()
    w = oracle.node_weights(oracle.col_sums(rp, col, None, n), oracle.W_AA)
    cnt_o, _, ws_o = oracle.pair_scores(rp, col, None, w, u, v)
    cnt, _, ws = eps.ops.pair_scores(g.rowptr, g.col, None, torch.from_numpy(w).to(dev), n, torch.from_numpy(u).to(dev),
                                     torch.from_numpy(v).to(dev), want_cn=False, grouped=True)
    assert np.array_equal(cnt.cpu().numpy(), cnt_o) and cnt_o.sum() > 0
    # hub x hub intersections are thousands of terms long: gate against the float64-accumulated value
    # (the float32 sequential oracle is itself ~1e-4 off there), and against the float32 oracle on short sums
    _, truth = oracle.pair_scores_f64(rp, col, None, w.astype(np.float64), u, v)
    assert rel_err(ws.cpu().numpy(), truth.astype(np.float32)) <= TOL
    short = cnt_o <= 512
    assert rel_err(ws.cpu().numpy()[short], ws_o[short]) <= TOL


def test_empty_and_tiny(eps, dev):
    g = eps.add_edges("ddi", torch.tensor([[0, 1], [1, 2]]), torch.ones(2), torch.zeros(2, 0, dtype=torch.long), 5).to(dev)
    aa, _ = eps.AA(g, torch.zeros(2, 0, dtype=torch.long))
    assert aa.shape == (0,)
    cn = eps.common_neighbors(g, torch.tensor([[0, 3, 4, 0], [2, 4, 4, 0]]))
    assert cn.cpu().tolist() == [1.0, 0.0, 0.0, 1.0]      # isolated nodes, self pair
    with pytest.raises(eps.EpsError):
        eps.ops.pair_scores(g.rowptr, g.col, None, None, 5, torch.zeros(3, dtype=torch.int32, device=dev),
                            torch.zeros(2, dtype=torch.int32, device=dev))


def test_full_size_properties(eps, dev):
    """Size-independent checks at a BASELINE-like scale (no oracle run): symmetry CN(u,v)==CN(v,u),
    CN(u,u)==deg(u), AA symmetric within tolerance, CN <= min degree, linearity in the value scale."""
    from eps_amd import synth
    g = synth.rmat_graph(scale=17, edge_factor=16, seed=3, device=dev)
    n = g.n_rows
    gen = torch.Generator(device="cpu").manual_seed(1)
    E = 2_000_000
    u = torch.randint(0, n, (E,), generator=gen, dtype=torch.int32).to(dev)
    v = torch.randint(0, n, (E,), generator=gen, dtype=torch.int32).to(dev)
    from eps_amd.heuristics import node_weight_table
    w = node_weight_table(g, eps.ops.W_AA)
    c1, _, a1 = eps.ops.pair_scores(g.rowptr, g.col, None, w, n, u, v, want_cn=False)
    c2, _, a2 = eps.ops.pair_scores(g.rowptr, g.col, None, w, n, v, u, want_cn=False)
    assert torch.equal(c1, c2)
    assert rel_err(a1.cpu().numpy(), a2.cpu().numpy()) <= TOL
    deg = g.degree().to(torch.int32)
    assert bool((c1 <= torch.minimum(deg[u.long()], deg[v.long()])).all())
    cs, _, _ = eps.ops.pair_scores(g.rowptr, g.col, None, None, n, u, u, want_cn=False)
    assert torch.equal(cs, deg[u.long()])
    # linearity: scaling every stored value by 2 scales CN by 4 (exact in float32: powers of two)
    g2 = g.fill_value(2.0)
    _, cn2, _ = eps.ops.pair_scores(g2.rowptr, g2.col, g2.val, None, n, u[:200000], v[:200000])
    assert torch.equal(cn2, 4.0 * c1[:200000].to(torch.float32))


def test_rmat24_share_vs_oracle(eps, oracle, dev):
    """BASELINE configs[4] on one GPU's share: R-MAT scale 24 (16.7 M nodes, ~520 M stored entries, hub degree ~4e5),
    CN + AA over 2^24 pairs -- half uniform, half 2-hop samples -- through the generic kernel and, sorted by v, through the
    column-run kernel (hashed bitmap: N > 2^20); a 20,000-pair sample is checked against the CPU ORACLE (counts exact,
    AA within the gate), all pairs against each other."""
    from eps_amd import synth
    from eps_amd.heuristics import node_weight_table
    g = synth.rmat_graph(scale=24, edge_factor=16, seed=5, device=dev)
    assert g.n_rows == 1 << 24 and g.nnz() > 400_000_000
    w = node_weight_table(g, eps.ops.W_AA)
    gen = torch.Generator(device=dev).manual_seed(1)
    half = 1 << 23
    u1 = torch.randint(0, g.n_rows, (half,), generator=gen, device=dev, dtype=torch.int32)
    v1 = torch.randint(0, g.n_rows, (half,), generator=gen, device=dev, dtype=torch.int32)
    e = torch.randint(0, g.nnz(), (half,), generator=gen, device=dev)      # 2-hop: a stored entry (w,u), a neighbour v of w
    deg = g.degree()
    wnode = torch.searchsorted(g.rowptr, e, right=True) - 1
    u2 = g.col[e]
    off = torch.minimum((torch.rand(half, generator=gen, device=dev) * deg[wnode]).long(), deg[wnode] - 1)
    v2 = g.col[g.rowptr[wnode] + off]
    u, v = torch.cat([u1, u2]).contiguous(), torch.cat([v1, v2]).contiguous()
    cnt, _, ws = eps.ops.pair_scores(g.rowptr, g.col, None, w, g.n_rows, u, v, want_cn=False, grouped=False)
    assert float(cnt.float().mean()) > 0.5
    order = torch.argsort(v.long() * g.n_rows + u.long())
    cg, _, wg = eps.ops.pair_scores(g.rowptr, g.col, None, w, g.n_rows, u[order].contiguous(), v[order].contiguous(),
                                    want_cn=False, grouped=True)
    assert torch.equal(cg, cnt[order]) and rel_err(wg.cpu().numpy(), ws[order].cpu().numpy()) <= 1e-5
    # the oracle on a sample (heaviest pairs included: the top of the degree-sum order + random ones)
    heavy = torch.argsort(deg[u.long()] + deg[v.long()], descending=True)[:2000]
    sel = torch.cat([heavy, torch.randint(0, u.numel(), (18000,), generator=gen, device=dev)])
    rp, col = g.rowptr.cpu().numpy(), g.col.cpu().numpy()
    wo = oracle.node_weights(oracle.col_sums(rp, col, None, g.n_rows), oracle.W_AA)
    assert rel_err(w.cpu().numpy(), wo) <= 1e-6
    us, vs = u[sel].cpu().numpy(), v[sel].cpu().numpy()
    co, _, ao = oracle.pair_scores(rp, col, None, wo, us, vs)
    assert np.array_equal(cnt[sel].cpu().numpy(), co), "common-neighbour counts bit-exact vs the oracle"
    got = ws[sel].cpu().numpy()
    light = co <= 1000
    assert light.sum() > 15000 and rel_err(got[light], ao[light]) <= 1e-5
    # hub x hub pairs sum up to ~4e5 float32 terms: there the float32 oracle itself is ~1e-4 off the float64 sum (any
    # summation order is); both are held to the float32 accumulation bound n * 2^-24 around the float64-accumulated value
    _, truth = oracle.pair_scores_f64(rp, col, None, wo.astype(np.float64), us, vs)
    bound = np.maximum(1e-5, co * 2.0 ** -24) * np.abs(truth) + 1e-30
    assert np.all(np.abs(got - truth) <= bound) and np.all(np.abs(ao - truth) <= bound)
    assert co.max() > 100_000
