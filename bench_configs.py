"""Secondary legs of bench.py: one bounded, driver-run measurement per BASELINE.json config that is not the headline
(configs[2] is bench.py's main line), each through the drop-in entry points (filter.py / rank.py mirrors) on the seeded
stand-ins of SURVEY 8(d), with the roofline unit of its dominant kernel.  Everything the CLIs print goes to a log buffer:
the bench prints ONE JSON line.

  configs[0]  ogbl-ddi Common-Neighbours filter + rank (full size; the reference runs it on the CPU -- there is no CPU product
              path here, by the rules of the build)
  configs[1]  ogbl-collab GCN filter -> CN rank, 150 k proposals (weighted graph, in = 128 + 256, H = 256, L = 3)
  configs[3]  ogbl-ppa GCN filter with the fused MLP decode over a column shard of >= 1 G candidates, 210 k proposals
  configs[4]  R-MAT scale-24, one GPU's share of the 1 B pairs: 125 M pairs, CN + AA
  + when $EPS_DATA_ROOT/ppa.pt exists: the headline scan and Hits@100 of the published RA -> RA recipe on the real graph.
"""
import argparse
import contextlib
import io
import os
import tempfile
import time

MFMA_F32_PEAK_TF = 157.3
HBM_PEAK_GBPS = 8000.0


@contextlib.contextmanager
def _quiet_cwd():
    """A scratch working directory (the CLIs write models/ and filtered_edges/ relative to it) with stdout captured."""
    old = os.getcwd()
    buf = io.StringIO()
    with tempfile.TemporaryDirectory(prefix="eps_bench_") as d:
        os.chdir(d)
        try:
            with contextlib.redirect_stdout(buf):
                yield buf
        finally:
            os.chdir(old)


def _sync(torch):
    torch.cuda.synchronize()


def _checkpoint(torch, models, datasets, filter_stage, dataset, model, cli):
    """A seeded random-init checkpoint under models/ (no trained weights exist offline): the filter loads it like a real one."""
    p = filter_stage.make_parser()
    args = models.default_model_configs(p.parse_args(["--dataset", dataset, "--model", model, "--checkpoint", "x", "--synthetic"] + cli))
    _, _, _, data = datasets.get_data(args)
    torch.manual_seed(0)
    m = models.build_model(args, data, torch.device("cpu"))
    os.makedirs("models", exist_ok=True)
    name = f"{dataset}_{model}||0|0.pt"
    torch.save(m.state_dict(), os.path.join("models", name))
    return name, args, data


def leg_config0(torch):
    """ddi-like, full size: CN ('simple') filter over all ~16 M candidates -> [E,3] file -> CN rank with Hits@20."""
    from eps_amd import filter_stage, rank_stage
    with _quiet_cwd() as log:
        t0 = time.perf_counter()
        f = filter_stage.main(["--dataset", "ddi", "--model", "simple", "--checkpoint", "ddi_simple||0|0.pt", "--synthetic"])
        _sync(torch)
        t1 = time.perf_counter()
        rows = torch.load(f).shape[0]
        curves = rank_stage.main(["--dataset", "ddi", "--model", "simple", "--sorted_edge_path", os.path.basename(f),
                                  "--num_sorted_edge", "100000", "--runs", "1", "--synthetic"])
        _sync(torch)
        t2 = time.perf_counter()
    return {"workload": "configs[0] ddi-like S1 (N=4,267, full size): filter.py --model simple (all candidates, full [E,3] file) -> "
                        "rank.py --model simple, 100 k proposals",
            "filter_s": t1 - t0, "candidates": rows, "candidates_per_s": rows / (t1 - t0), "rank_s": t2 - t1,
            "rank_curve": _jsonable(curves), "note": "wall clock incl. stand-in generation and file I/O; on the GPU (the reference runs "
                                                     "this config on the CPU)"}


def leg_config1(torch):
    """collab-like weighted graph, full size: GCN filter (in = 128 features + 256-d embedding, H = 256, L = 3) -> 150 k
    proposals -> CN rank."""
    from eps_amd import datasets, filter_stage, models, rank_stage
    with _quiet_cwd() as log:
        name, args, data = _checkpoint(torch, models, datasets, filter_stage, "collab", "gcn", [])
        t0 = time.perf_counter()
        f = filter_stage.main(["--dataset", "collab", "--model", "gcn", "--checkpoint", name, "--synthetic", "--keep_top", "150000"])
        _sync(torch)
        t1 = time.perf_counter()
        curves = rank_stage.main(["--dataset", "collab", "--model", "simple", "--sorted_edge_path", os.path.basename(f),
                                  "--num_sorted_edge", "150000", "--runs", "1", "--synthetic"])
        _sync(torch)
        t2 = time.perf_counter()
        seen = [l for l in log.getvalue().splitlines() if l.startswith("using ") and " edges; scored in " in l]
    return {"workload": "configs[1] collab-like S2 (N=235,868, weighted, 128 features + 256-d embedding, H=256, L=3): filter.py --model gcn "
                        "--keep_top 150000 -> rank.py --model simple --num_sorted_edge 150000",
            "filter_s": t1 - t0, "rank_s": t2 - t1, "filter_log": seen[-1] if seen else None, "rank_curve": _jsonable(curves),
            "note": "wall clock incl. stand-in generation, checkpoint load and file I/O"}


def leg_config3(torch, min_candidates=1_000_000_000):
    """ppa stand-in: GCN embeddings (58 one-hot features + 256-d embedding -> 256, L = 3) once, then the HALF candidate list of
    column blocks decoded by the fused MFMA LinkPredictor until >= ``min_candidates`` directed candidates were scored."""
    from eps_amd import candidates, datasets, filter_stage, models, ops, scan
    cli = ["--num_layers", "3", "--hidden_channels", "256", "--dropout", "0.0", "--batch_size", "65536", "--use_feature", "1",
           "--use_learnable_embedding", "1"]
    with _quiet_cwd():
        name, args, data = _checkpoint(torch, models, datasets, filter_stage, "ppa", "gcn", cli)
        dev = torch.device("cuda", torch.cuda.current_device())
        data = data.to(dev)
        model = models.build_model(args, data, dev)
        model.load_state_dict(torch.load(os.path.join("models", name), map_location=dev))
        model.eval()
        g = data.adj_t
        with torch.no_grad():
            _sync(torch)
            t0 = time.perf_counter()
            model.embeddings(data.x, g)
            _sync(torch)
            t_emb = time.perf_counter() - t0
            revpos, md, sp = scan.reverse_positions(g), scan.max_degree(g), scan.window_splits(g)
            seen, t_list, t_dec = 0, 0.0, 0.0
            for v_lo, v_hi in candidates.column_blocks(g):
                _sync(torch)
                t1 = time.perf_counter()
                r = ops.expand_unit(g.rowptr, g.col, None, g.n_rows, v_lo, v_hi, md, sp, want_score=False, want_v=True,
                                    col_order=candidates.heaviest_first(g, v_lo, v_hi), revpos=revpos)
                pairs = r.pairs
                _sync(torch)
                t2 = time.perf_counter()
                sc = model(data.x, pairs, g).reshape(-1)
                _sync(torch)
                t3 = time.perf_counter()
                t_list += t2 - t1
                t_dec += t3 - t2
                seen += 2 * pairs.shape[1]
                del r, pairs, sc
                if seen >= min_candidates:
                    break
    h, layers = 256, 3
    flops = (seen / 2) * (h + 2 * h * h * (layers - 1) + 2 * h)          # each unordered pair decoded once
    t = t_list + t_dec
    return {"workload": "configs[3] ppa stand-in (N=576,289, 58 features + 256-d embedding, H=256, L=3 GCN + L=3 LinkPredictor): GCN "
                        "forward once, then candidate generation + fused MFMA decode over column blocks until >= 1 G directed "
                        "candidates (each unordered pair decoded once: the decode is symmetric)",
            "directed_candidates": seen, "embeddings_s": t_emb, "list_s": t_list, "decode_s": t_dec,
            "candidates_per_s": seen / t, "bound": "mfma", "TFLOPs_incl_list_generation": flops / t / 1e12,
            "frac_incl_list_generation": flops / t / 1e12 / MFMA_F32_PEAK_TF, "TFLOPs_decode_only": flops / t_dec / 1e12,
            "frac_decode_only": flops / t_dec / 1e12 / MFMA_F32_PEAK_TF}


def leg_config4(torch, n_pairs=125_000_000):
    """R-MAT scale-24 (16.7 M nodes, 256 M generated edges, symmetrised): one GPU's eighth of the 1 B pairs -- half uniform
    random, half 2-hop samples -- through the generic pair kernel: CN count + AA sum."""
    from eps_amd import ops, synth
    from eps_amd.heuristics import node_weight_table
    dev = torch.device("cuda", torch.cuda.current_device())
    t0 = time.perf_counter()
    g = synth.rmat_graph(scale=24, edge_factor=16, seed=5, device=dev)
    _sync(torch)
    t_graph = time.perf_counter() - t0
    w = node_weight_table(g, ops.W_AA)
    gen = torch.Generator(device=dev).manual_seed(1)
    half = n_pairs // 2
    u1 = torch.randint(0, g.n_rows, (half,), generator=gen, device=dev, dtype=torch.int32)
    v1 = torch.randint(0, g.n_rows, (half,), generator=gen, device=dev, dtype=torch.int32)
    e = torch.randint(0, g.nnz(), (half,), generator=gen, device=dev)
    wnode = g.row_index()[e]
    u2 = g.col[e]
    deg = g.degree()
    off = torch.minimum((torch.rand(half, generator=gen, device=dev) * deg[wnode]).long(), deg[wnode] - 1)
    v2 = g.col[g.rowptr[wnode] + off]
    u, v = torch.cat([u1, u2]).contiguous(), torch.cat([v1, v2]).contiguous()
    del u1, v1, u2, v2, e, wnode, off
    ops.pair_scores(g.rowptr, g.col, None, w, g.n_rows, u[:1 << 20].contiguous(), v[:1 << 20].contiguous(), want_cn=False, grouped=False)
    _sync(torch)
    t1 = time.perf_counter()
    cnt, _, ws = ops.pair_scores(g.rowptr, g.col, None, w, g.n_rows, u, v, want_cn=False, grouped=False)
    _sync(torch)
    dt = time.perf_counter() - t1
    du, dv, cn = int(deg[u.long()].sum()), int(deg[v.long()].sum()), int(cnt.sum())
    alg = 4 * (du + dv) + 4 * cn + 48 * u.numel()               # 4 (d_u + d_v) + 4 CN + 32 + 8 + 8 per pair
    return {"workload": "configs[4] R-MAT scale-24 (N=16,777,216, nnz=%d, max degree %d): one GPU's share, %d pairs (half uniform, half "
                        "2-hop samples), CN count + AA sum, eps_pair_scores" % (g.nnz(), int(deg.max()), u.numel()),
            "graph_build_s": t_graph, "pairs": u.numel(), "kernel_s": dt, "pairs_per_s": u.numel() / dt, "bound": "hbm",
            "algorithmic_bytes": alg, "GBps": alg / dt / 1e9, "frac": alg / dt / 1e9 / HBM_PEAK_GBPS,
            "mean_degree_sum": (du + dv) / u.numel(), "mean_cn": cn / u.numel()}


def leg_real_ppa(torch, keep_top):
    """$EPS_DATA_ROOT/ppa.pt present: the headline scan on the REAL ogbl-ppa training graph and Hits@100 of the published
    recipe RA filter -> RA rank at 4 M proposals (README.md:11-17, submit_job.py:207-213)."""
    root = os.environ.get("EPS_DATA_ROOT")
    if not root or not os.path.exists(os.path.join(root, "ppa.pt")):
        return None
    from eps_amd import datasets, filter_stage, ops, rank_stage, scan
    from eps_amd.heuristics import node_weight_table
    out = {"workload": "real ogbl-ppa from $EPS_DATA_ROOT/ppa.pt"}
    try:
        args = argparse.Namespace(dataset="ppa", synthetic=False, use_feature=False)
        _, _, _, data = datasets.get_data(args)
        dev = torch.device("cuda", torch.cuda.current_device())
        g = data.to(dev).adj_t
        w = node_weight_table(g, ops.W_AA)
        st = {}
        scan.scan_topk(g, w, keep_top, stats=st, relabel=True)
        _sync(torch)
        t0 = time.perf_counter()
        for _ in range(5):
            scan.scan_topk(g, w, keep_top, stats=st)
        _sync(torch)
        dt = (time.perf_counter() - t0) / 5
        out.update(n_nodes=g.n_rows, nnz=g.nnz(), candidates=st["candidates"], ms_per_step=dt * 1e3, value=st["candidates"] / dt)
        with _quiet_cwd() as log:
            f = filter_stage.main(["--dataset", "ppa", "--model", "resource_allocation", "--checkpoint", "ppa_resource_allocation||0|0.pt",
                                   "--keep_top", "4000000"])
            curves = rank_stage.main(["--dataset", "ppa", "--model", "resource_allocation", "--sorted_edge_path", os.path.basename(f),
                                      "--num_sorted_edge", "4000000", "--runs", "1"])
        out["ra_filter_ra_rank_curve"] = _jsonable(curves)
        out["published_hits_at_100"] = 53.24
    except Exception as exc:       # a malformed data file must not cost the bench its line
        out["error"] = f"{type(exc).__name__}: {exc}"
    return out


def _jsonable(x):
    """Curves come back as nested lists with 0-dim tensors: plain numbers for the JSON line."""
    if isinstance(x, (list, tuple)):
        return [_jsonable(y) for y in x]
    if isinstance(x, dict):
        return {str(k): _jsonable(v) for k, v in x.items()}
    if hasattr(x, "item") and getattr(x, "numel", lambda: 2)() == 1:
        return x.item()
    if isinstance(x, (int, float, str, bool)) or x is None:
        return x
    return str(x)


def run_all(torch, keep_top):
    legs = {}
    for name, fn in (("config0_ddi_cn", leg_config0), ("config1_collab_gcn_cn", leg_config1), ("config3_ppa_gcn_decode", leg_config3),
                     ("config4_rmat24_share", leg_config4)):
        t0 = time.perf_counter()
        try:
            legs[name] = fn(torch)
        except Exception as exc:
            legs[name] = {"error": f"{type(exc).__name__}: {exc}"}
        legs[name]["leg_wall_s"] = time.perf_counter() - t0
        torch.cuda.empty_cache()
    real = leg_real_ppa(torch, keep_top)
    if real is not None:
        legs["real_ppa"] = real
    return legs
