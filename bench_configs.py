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


def _gpu_ms(torch, fn, reps=1):
    """Mean HIP-event milliseconds of ``fn`` on the current stream (one warm call first)."""
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        r = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, r


def _cpu_gnn_filter_baseline(torch, model, x_in, g, edges, layers_timed=None, batch=65536):
    """SURVEY 8(d) / BASELINE.md 2: the reference's GNN filter on the host cores.  Its scoring loop (filter.py:113-121) calls
    the WHOLE model for every batch of 65,536 candidates -- LinkGNN.forward recomputes the GNN embeddings each time
    (models.py:505) -- so one batch costs one GNN forward + one decode: FAITHFUL = batch / (t_gnn + t_decode); FAIR = the same
    loop with the embeddings computed once = batch / t_decode.  Host libraries as its CPU run would use them: SciPy CSR x dense
    for the aggregation (one thread), BLAS for the dense layers (the threads numpy's BLAS takes).  ``layers_timed``: time only
    the first n GCN layers and scale to the model's depth (a bounded sample: a full ppa-sized forward is ~40 s on one SpMM
    thread)."""
    import numpy as np
    import scipy.sparse as ssp
    n = g.n_rows
    gh = g.cpu()
    A = gh.to_scipy().astype(np.float32)
    An = (A + ssp.identity(n, dtype=np.float32, format="csr")).tocsr()
    An.setdiag(1.0)                                   # gcn_norm: the diagonal is SET to 1 (fill_diag), not added to
    with np.errstate(divide="ignore"):
        dis = 1.0 / np.sqrt(np.asarray(An.sum(1)).ravel())
    dis[np.isinf(dis)] = 0
    An = (ssp.diags(dis) @ An @ ssp.diags(dis)).tocsr().astype(np.float32)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    ws = [sd[k] for k in sorted(sd) if k.startswith("gnn.convs.") and k.endswith(".weight")]
    bs = [sd[k] for k in sorted(sd) if k.startswith("gnn.convs.") and k.endswith(".bias")]
    h = x_in.detach().cpu().numpy()
    n_layers = len(ws)
    do = n_layers if layers_timed is None else min(layers_timed, n_layers)
    t0 = time.perf_counter()
    for i in range(do):
        h = An @ (h @ ws[i]) + bs[i]
        if i < n_layers - 1:
            h = np.maximum(h, 0)
    t_gnn = (time.perf_counter() - t0) * (n_layers / do)
    if do < n_layers:                                 # (the decode below only needs SOME embeddings of the right shape to be timed)
        h = np.ascontiguousarray(h[:, :ws[-1].shape[1]]) if h.shape[1] >= ws[-1].shape[1] else np.pad(h, ((0, 0), (0, ws[-1].shape[1] - h.shape[1])))
    lw = [sd[k] for k in sorted(sd) if k.startswith("linkpred.lins.") and k.endswith(".weight")]
    lb = [sd[k] for k in sorted(sd) if k.startswith("linkpred.lins.") and k.endswith(".bias")]
    ub, vb = edges[0][:batch].cpu().numpy(), edges[1][:batch].cpu().numpy()
    t0 = time.perf_counter()
    z = h[ub] * h[vb]
    for i, (W, b) in enumerate(zip(lw, lb)):
        z = z @ W.T + b
        if i < len(lw) - 1:
            z = np.maximum(z, 0)
    z = 1 / (1 + np.exp(-z))
    t_dec = time.perf_counter() - t0
    return {"kind": "port", "unit": "edges/s", "gnn_forward_s": t_gnn, "decode_batch_s": t_dec, "batch": int(len(ub)),
            "faithful_value": len(ub) / (t_gnn + t_dec), "fair_value": len(ub) / t_dec, "cores": {"spmm": 1, "blas": os.cpu_count()},
            "sample": f"one GNN forward ({do} of {n_layers} layers timed" + (", scaled to the depth" if do < n_layers else "") +
                      f") + the decode of one batch of {len(ub)} candidates: what ONE iteration of filter.py:113-121 costs on the host; "
                      "faithful = embeddings recomputed per batch (models.py:505), fair = computed once",
            "full_ok": do == n_layers, "probabilities_checked": None}


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
        first = dict(filter_stage.LAST_TIMING)
        # (once more: the first run of a kind in this process also loads the code objects of every kernel it is the first to use)
        filter_stage.main(["--dataset", "ddi", "--model", "simple", "--checkpoint", "ddi_simple||0|1.pt", "--synthetic"])
        _sync(torch)
        ft = dict(filter_stage.LAST_TIMING)
        ft["first_run_scored_s"] = first.get("scored_s")
        t1b = time.perf_counter()
        rows = torch.load(f).shape[0]
        curves = rank_stage.main(["--dataset", "ddi", "--model", "simple", "--sorted_edge_path", os.path.basename(f),
                                  "--num_sorted_edge", "100000", "--runs", "1", "--synthetic"])
        _sync(torch)
        t2 = time.perf_counter() - (t1b - t1)
    tm = dict(ft)
    return {"workload": "configs[0] ddi-like S1 (N=4,267, full size): filter.py --model simple (all candidates, full [E,3] file) -> "
                        "rank.py --model simple, 100 k proposals",
            "candidates": rows, "scored_s": tm.get("scored_s"), "gpu_ms": tm.get("gpu_ms"), "first_run_scored_s": tm.get("first_run_scored_s"),
            "candidates_per_s": rows / max(tm.get("scored_s") or (t1 - t0), 1e-9),
            # r05: the list comes out of ONE dense product A A^T on the f32 MFMA (csrc/dense_cn.hip): the tiles on and below the
            # diagonal of a 4352^3 product, against the 157.3 TF fp32 matrix peak over the WHOLE scoring section
            "bound": "mfma", "product_flops": 4352 ** 3,
            "TFLOPs_over_section": 4352 ** 3 / max((tm.get("gpu_ms") or 1e9) * 1e-3, 1e-9) / 1e12,
            "frac": 4352 ** 3 / max((tm.get("gpu_ms") or 1e9) * 1e-3, 1e-9) / 1e12 / 157.3,
            "rows_GBps": rows * 12 / max((tm.get("gpu_ms") or 1e9) * 1e-3, 1e-9) / 1e9,
            "wall_s": {"filter_cli": t1 - t0, "rank_cli": t2 - t1},
            "rank_curve": _jsonable(curves),
            "note": "scored_s / gpu_ms: the filter's scoring section (graph on the device -> all rows ordered on the device) on the host "
                    "clock and between two HIP events, SECOND run in this process (first_run_scored_s: the first, which also loads the "
                    "code objects of the kernels it is the first to use); wall_s: the whole CLI incl. stand-in generation, [E,3] file write / read and "
                    "the rank stage's training-free evaluation.  r05: a dense graph's common-neighbour list = A A^T on the matrix cores "
                    "(lower tiles, mirrored), a masked read in the reference's candidate order and ONE stable sort by the integer "
                    "count -- about twenty launches, 2.3 ms (r04: column blocks through the sparse kernels, 7.5 ms in hundreds of "
                    "launches); `frac` = the product's flops over the whole section against the fp32 MFMA peak (the product itself: "
                    "0.9-1.1 ms = 77-93 TF, profiles/r05/dense_cn.txt).  The reference runs this config on the CPU; there is no CPU "
                    "product path here"}


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
        first = dict(filter_stage.LAST_TIMING)
        # (once more: the first run of a kind in this process also loads the code objects of every kernel it is the first to use)
        import shutil
        shutil.copy(os.path.join("models", name), os.path.join("models", "collab_gcn||0|1.pt"))
        filter_stage.main(["--dataset", "collab", "--model", "gcn", "--checkpoint", "collab_gcn||0|1.pt", "--synthetic", "--keep_top", "150000"])
        _sync(torch)
        ft = dict(filter_stage.LAST_TIMING)
        ft["first_run_scored_s"] = first.get("scored_s")
        t1b = time.perf_counter()
        curves = rank_stage.main(["--dataset", "collab", "--model", "simple", "--sorted_edge_path", os.path.basename(f),
                                  "--num_sorted_edge", "150000", "--runs", "1", "--synthetic"])
        _sync(torch)
        t2 = time.perf_counter() - (t1b - t1)
        seen = [l for l in log.getvalue().splitlines() if l.startswith("using ") and " edges; scored in " in l]
        # the reference's CPU path for this filter, on the same model and graph (bounded: one forward + one batch)
        dev = torch.device("cuda", torch.cuda.current_device())
        data = data.to(dev)
        model = models.build_model(args, data, dev)
        model.load_state_dict(torch.load(os.path.join("models", name), map_location=dev))
        model.eval()
        g = data.adj_t
        x_in = torch.cat([model.emb.weight.detach(), data.x], 1) if getattr(model, "emb", None) is not None else data.x
        gen = torch.Generator(device=dev).manual_seed(3)
        edges = torch.randint(0, g.n_rows, (2, 65536), generator=gen, device=dev)
        cpu = _cpu_gnn_filter_baseline(torch, model, x_in, g, edges)
    h, layers = 256, 3
    n_c = ft.get("candidates") or 0
    flops = (n_c / 2) * (h + 2 * h * h * (layers - 1) + 2 * h)           # each unordered pair decoded once
    gms = ft.get("gpu_ms") or 0.0
    return {"workload": "configs[1] collab-like S2 (N=235,868, weighted, 128 features + 256-d embedding, H=256, L=3): filter.py --model gcn "
                        "--keep_top 150000 -> rank.py --model simple --num_sorted_edge 150000",
            "candidates": n_c, "scored_s": ft.get("scored_s"), "gpu_ms": gms, "first_run_scored_s": ft.get("first_run_scored_s"),
            "candidates_per_s": n_c / max(ft.get("scored_s") or 1e9, 1e-9),
            "bound": "mfma", "TFLOPs_decode_over_gpu_ms": flops / max(gms * 1e-3, 1e-9) / 1e12,
            "frac": flops / max(gms * 1e-3, 1e-9) / 1e12 / MFMA_F32_PEAK_TF,
            "wall_s": {"filter_cli": t1 - t0, "rank_cli": t2 - t1}, "filter_log": seen[-1] if seen else None,
            "rank_curve": _jsonable(curves), "cpu_baseline": cpu,
            "gpu_over_cpu": {"faithful": n_c / max(ft.get("scored_s") or 1e9, 1e-9) / cpu["faithful_value"],
                             "fair": n_c / max(ft.get("scored_s") or 1e9, 1e-9) / cpu["fair_value"]},
            "note": "scored_s / gpu_ms: the filter's scoring section (GCN embeddings once, candidate blocks, fused MFMA decode of each "
                    "unordered pair, streaming top-K) on the host clock / between HIP events, SECOND run in this process "
                    "(first_run_scored_s: the first, code-object loads included); frac = decode flops of that section over "
                    "gpu_ms against the fp32 MFMA peak (the section also holds the embeddings and the list generation); wall_s: whole "
                    "CLIs incl. stand-in generation, checkpoint and file I/O"}


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
            # ---- the CPU path of the same filter (bounded: one GCN layer timed, scaled to three; one batch decoded)
            x_in = torch.cat([model.emb.weight.detach(), data.x], 1)
            gen = torch.Generator(device=dev).manual_seed(3)
            cpu = _cpu_gnn_filter_baseline(torch, model, x_in, g, torch.randint(0, g.n_rows, (2, 65536), generator=gen, device=dev),
                                           layers_timed=1)
            # ---- the RANK half of configs[3] (rank.py --model sage on ppa): one evaluation pass of train_and_eval.test
            # (train_and_eval.py:98-136): SAGE embeddings once, then the LinkPredictor decode of every evaluation edge -- ppa's
            # splits hold 3 M valid + 3 M test positives, 3 M shared negatives per split and the eval_train sample: ~21 M decodes per
            # epoch of evaluation -- and the AA heuristic over the same lists (test_adamic, :160-173: 6.06 M positive, 3 M negative)
            sage = models.SAGE(x_in.shape[1], 256, 256, 3, 0.0).to(dev).eval()
            lp = models.LinkPredictor(256, 256, 1, 3, 0.0).to(dev).eval()
            ms_sage, hs = _gpu_ms(torch, lambda: sage(x_in, g), reps=3)
            n_eval = 21_000_000
            eu = torch.randint(0, g.n_rows, (n_eval,), generator=gen, device=dev, dtype=torch.int32)
            ev = torch.randint(0, g.n_rows, (n_eval,), generator=gen, device=dev, dtype=torch.int32)
            lws = [l.weight.detach() for l in lp.lins]
            lbs = [l.bias.detach() for l in lp.lins]
            ms_dec, _ = _gpu_ms(torch, lambda: ops.mlp_decode(hs, eu, ev, lws, lbs), reps=3)
            from eps_amd.heuristics import node_weight_table
            w_aa = node_weight_table(g, ops.W_AA)
            row, colx, _ = g.coo()
            sel = torch.randint(0, row.numel(), (6_060_000,), generator=gen, device=dev)
            pu, pv = row[sel].to(torch.int32).contiguous(), colx[sel].to(torch.int32).contiguous()        # positive-like: stored edges
            nu, nv = eu[:3_000_000].contiguous(), ev[:3_000_000].contiguous()                              # negatives: uniform pairs
            ms_pos, _ = _gpu_ms(torch, lambda: ops.pair_scores(g.rowptr, g.col, None, w_aa, g.n_rows, pu, pv, want_cn=False, grouped=False), reps=3)
            ms_neg, _ = _gpu_ms(torch, lambda: ops.pair_scores(g.rowptr, g.col, None, w_aa, g.n_rows, nu, nv, want_cn=False, grouped=False), reps=3)
            deg = g.degree()
            b_pos = 4 * int(deg[pu.long()].sum() + deg[pv.long()].sum()) + 48 * pu.numel()
            b_neg = 4 * int(deg[nu.long()].sum() + deg[nv.long()].sum()) + 48 * nu.numel()
            fl_dec = float(n_eval) * (256 + 2 * 256 * 256 * 2 + 2 * 256)
            rank_half = {"what": "rank.py --model sage on the ppa stand-in, one evaluation pass (train_and_eval.py:98-136) + the AA heuristic "
                                 "over the evaluation lists (:160-173); random-init weights, synthetic lists of the splits' sizes",
                         "sage_forward_ms": ms_sage, "decode_edges": n_eval, "decode_ms": ms_dec, "decode_edges_per_s": n_eval / ms_dec * 1e3,
                         "decode_TFLOPs": fl_dec / ms_dec / 1e9, "decode_frac_mfma": fl_dec / ms_dec / 1e9 / MFMA_F32_PEAK_TF,
                         "aa_pos_pairs": pu.numel(), "aa_pos_ms": ms_pos, "aa_pos_GBps_8d": b_pos / ms_pos / 1e6,
                         "aa_pos_frac_hbm": b_pos / ms_pos / 1e6 / HBM_PEAK_GBPS,
                         "aa_neg_pairs": nu.numel(), "aa_neg_ms": ms_neg, "aa_neg_GBps_8d": b_neg / ms_neg / 1e6,
                         "aa_neg_frac_hbm": b_neg / ms_neg / 1e6 / HBM_PEAK_GBPS}
    h, layers = 256, 3
    flops = (seen / 2) * (h + 2 * h * h * (layers - 1) + 2 * h)          # each unordered pair decoded once
    t = t_list + t_dec
    return {"workload": "configs[3] ppa stand-in (N=576,289, 58 features + 256-d embedding, H=256, L=3 GCN + L=3 LinkPredictor): GCN "
                        "forward once, then candidate generation + fused MFMA decode over column blocks until >= 1 G directed "
                        "candidates (each unordered pair decoded once: the decode is symmetric)",
            "directed_candidates": seen, "embeddings_s": t_emb, "list_s": t_list, "decode_s": t_dec,
            "candidates_per_s": seen / t, "bound": "mfma", "TFLOPs_incl_list_generation": flops / t / 1e12,
            "frac_incl_list_generation": flops / t / 1e12 / MFMA_F32_PEAK_TF, "TFLOPs_decode_only": flops / t_dec / 1e12,
            "frac_decode_only": flops / t_dec / 1e12 / MFMA_F32_PEAK_TF,
            "gpu_ms": {"embeddings": t_emb * 1e3, "list": t_list * 1e3, "decode": t_dec * 1e3},
            "cpu_baseline": cpu,
            "gpu_over_cpu": {"faithful": seen / t / cpu["faithful_value"], "fair": seen / t / cpu["fair_value"]},
            "rank_half": rank_half}


def leg_full_list(torch):
    """The literal filter.py:113-165 on the bench graph: EVERY candidate gets its exact score, written out in candidate order
    (no bar, no top-K), in ONE pass over the two-hop paths: eps_expand_unit_list (r06) over column blocks of < 2^33 paths --
    column-major, both orientations, u ascending, float32 of the exact 2^-40 fixed-point sums; segments sized by
    min(two-hop paths, N) per column (no counting launch, no host read before a launch), the counts come back with the list.
    The list itself is 8 bytes per candidate (u + score; v is the segment that holds the slot): 102 GB leave the chip for
    12.7 G candidates, which is why the production path never writes it.  The r02-r05 two-pass form (eps_expand_unit_count +
    _fill, exact layout) is timed beside it."""
    from eps_amd import candidates, ops, scan, synth
    from eps_amd.heuristics import node_weight_table
    dev = torch.device("cuda", torch.cuda.current_device())
    g = synth.ppa_like(seed=3, device=dev)
    w = node_weight_table(g, ops.W_AA)
    md, sp = scan.max_degree(g), scan.window_splits(g)
    blocks = list(candidates.column_blocks(g, 1 << 33))
    blocks2 = list(candidates.column_blocks(g))
    pre, pre_host = candidates.segment_bounds(g)

    def one_pass():
        parts = []
        for v_lo, v_hi in blocks:
            ub = (pre[v_lo:v_hi + 1] - pre[v_lo]).contiguous()
            r = ops.expand_unit(g.rowptr, g.col, w, g.n_rows, v_lo, v_hi, md, sp, want_score=True, want_v=False,
                                col_order=candidates.heaviest_first(g, v_lo, v_hi), colptr_ub=ub,
                                total_ub=int(pre_host[v_hi] - pre_host[v_lo]))
            parts.append(torch.cat([r.counts.sum().view(1), r.status.view(1).to(torch.int64)]))
            del r
        t = torch.stack(parts).cpu()                  # ONE host read, after the last launch: the counts and the status words
        if int(t[:, 1].sum()):
            raise RuntimeError("full-list leg: status %s" % t[:, 1].tolist())
        return int(t[:, 0].sum())

    def two_pass():
        n = 0
        for v_lo, v_hi in blocks2:
            r = ops.expand_unit(g.rowptr, g.col, w, g.n_rows, v_lo, v_hi, md, sp, want_score=True, want_v=False,
                                col_order=candidates.heaviest_first(g, v_lo, v_hi))
            n += int(r[1].numel())
            del r
        return n

    def timed(fn):
        fn()
        _sync(torch)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record()
        n = fn()
        b.record()
        _sync(torch)
        return n, a.elapsed_time(b), time.perf_counter() - t0

    n2, ms2, _ = timed(two_pass)
    torch.cuda.empty_cache()
    n_cand, ms, wall = timed(one_pass)
    assert n_cand == n2
    paths = int(candidates.path_counts(g).sum().item())
    abytes = 4 * paths + 8 * n_cand + 24 * g.nnz()      # every two-hop path read ONCE + the list + the graph's own arrays
    return {"workload": "full list (filter.py:113-165 as written): every 2-hop non-edge of the ppa-like bench graph with its exact AA "
                        "score, in candidate order, one pass, %d launches" % len(blocks),
            "kernel": "filter_scan_kernel<FS_EMIT> (eps_expand_unit_list)", "candidates": n_cand,
            "two_hop_paths": paths, "gpu_ms": ms, "wall_s": wall, "candidates_per_s": n_cand / ms * 1e3, "bound": "hbm",
            "algorithmic_bytes": abytes, "GBps": abytes / ms / 1e6, "frac": abytes / ms / 1e6 / HBM_PEAK_GBPS,
            "floor_ms_at_peak": abytes / HBM_PEAK_GBPS / 1e6,
            "slots_allocated": int(pre_host[-1]),
            "two_pass": {"gpu_ms": ms2, "candidates_per_s": n2 / ms2 * 1e3, "blocks": len(blocks2),
                         "kernel": "filter_scan_kernel<FS_COUNT> + <FS_EMIT> (eps_expand_unit_count / _fill): the r02-r05 leg"},
            "note": "gpu_ms: HIP events around all launches incl. allocation and the ONE host read at the end (counts + status); "
                    "algorithmic bytes = the one-pass model of VERDICT r04 #4: 4 B x two-hop paths + 8 B x candidates written + "
                    "24 B x nnz.  Inside a column the kernel still walks its paths twice (mark, then rank + bucket) and moves a "
                    "4-byte record per path through a per-workgroup scratch that stays in L2 / MALL; what went away is the "
                    "counting LAUNCH (20.7 ms) and the padding.  profiles/r06/full_list_*.txt: where the launch's time goes "
                    "(stamps, ablations) and what was tried (id windows for direct accumulation: slower; LDS-staged u stores: "
                    "slower -- the launch is bound by instruction issue at 4 waves per SIMD, not by bytes).  The 102 GB of rows "
                    "alone are 12.8 ms at the HBM peak; no production path writes them (rank.py:294 reads K rows)"}


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
                     ("full_list_every_candidate_scored", leg_full_list), ("config4_rmat24_share", leg_config4)):
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
