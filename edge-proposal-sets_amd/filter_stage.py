"""The filter stage (drop-in for filter.py:26-166): score every 2-hop non-edge with the filter model,
order the candidates, save the proposal file.

Same command line (filter.py:27-47), same checkpoint-name protocol ``spec|edges|num|run.pt`` (:72-76), same
output ``filtered_edges/{spec}_{edges}_{num}_{run}_sorted_edges.pt`` = float32 [E,3] rows (u, v, score)
(:160-165).  Differences, all on purpose:
  * candidates are generated in column blocks on the device and streamed to the scoring kernels; A @ A is
    never materialised on the host;
  * GNN filters compute the node embeddings once, not once per scoring batch;
  * the order is the declared rule (score desc, candidate index asc) instead of an unstable sort;
  * extensions: ``--synthetic`` (seeded stand-in data), ``--keep_top K`` (save only the best K rows -- rank.py
    never reads past ``num_sorted_edge``).
"""
from __future__ import annotations

import argparse
import time
from pathlib import Path

import torch

from . import _lib, candidates, ops, proposals, scan
from .datasets import get_data
from .graph import CSRGraph, add_edges
from .heuristics import node_weight_table, pair_scores_streamed
from .models import build_model, default_model_configs


def make_parser():
    parser = argparse.ArgumentParser(description='filter stage (MI355X)')
    parser.add_argument('--dataset', type=str, required=True)
    parser.add_argument('--model', type=str, required=True)
    parser.add_argument('--checkpoint', type=str, required=True)
    parser.add_argument('--num_layers', type=int)
    parser.add_argument('--hidden_channels', type=int)
    parser.add_argument('--dropout', type=float)
    parser.add_argument('--batch_size', type=int)
    parser.add_argument('--lr', type=float)
    parser.add_argument('--epochs', type=int)
    parser.add_argument('--use_feature', type=bool)
    parser.add_argument('--use_learnable_embedding', type=bool)
    parser.add_argument('--device', type=int, default=None)
    # extensions
    parser.add_argument('--dist_backend', type=str, default=None, choices=[None, 'nccl', 'gloo'])
    parser.add_argument('--synthetic', action="store_true", default=False)
    parser.add_argument('--keep_top', type=int, default=0)
    parser.add_argument('--shard_proposals', action='store_true',
                        help='sharded runs (torchrun) under --keep_top: every rank writes the chunk of the sorted list it ordered '
                             '(<file>.shard{r}of{N}; rank.py reads the shards in rank order) instead of sending its rows to rank 0')
    return parser


def train_only_graph(split_edge, num_nodes: int, device) -> CSRGraph:
    """filter.py:130-139: the RA filter scores on a graph built from the TRAIN edges only (both directions,
    integer ones, duplicates summed by the csr_matrix constructor), ignoring proposal edges and adj_t."""
    e = split_edge['train']['edge'].t().to(device)
    both = torch.cat([e, e.flip(0)], 1)
    g = CSRGraph.from_edge_index(both, torch.ones(both.shape[1], device=device), (num_nodes, num_nodes))
    if g.val is not None and bool((g.val == 1).all()):      # every undirected edge listed once (ddi, ppa): unit graph
        g = g.fill_value(1.0)
    return g


def score_block(args, model, data, pairs: torch.Tensor, ra_graph=None) -> torch.Tensor:
    """Scores of one candidate block [2,E_blk] (int64, on device) -> float32 [E_blk] on device."""
    u = pairs[0].to(torch.int32).contiguous()
    v = pairs[1].to(torch.int32).contiguous()
    if args.model == "adamic_ogb":                       # filter.py:122-126
        g = data.adj_t
        return pair_scores_streamed(g, u, v, node_weight_table(g, ops.W_AA), grouped=True)[2]   # column-major blocks
    if args.model == "resource_allocation":              # filter.py:127-142 (float64 math, FloatTensor out)
        w = node_weight_table(ra_graph, ops.W_RA, f64=True)
        return pair_scores_streamed(ra_graph, u, v, w, grouped=True)[2].to(torch.float32)
    return model(data.x, pairs, data.adj_t).reshape(-1)  # filter.py:116-121


def rank_column_range(g: CSRGraph, rank: int, world: int):
    """Contiguous column shard of this rank, balanced by 2-hop path counts (the cost of generating + scoring a column)."""
    if world == 1:
        return 0, g.n_rows
    from . import dist as epd
    b = epd.balanced_bounds(epd.column_work(g.rowptr, g.col), world)
    return b[rank], b[rank + 1]


def fused_node_weights(args, g: CSRGraph, ra_graph):
    """Per-node weights that make the fused kernels' sum_w A[u,w]*A[v,w]*node_w[w] the filter model's score, or None
    when the model does not score on the candidate graph itself (GNN filters; RA after proposal edges were added).
    AA: filter.py:122-126; CN 'simple': :116-121 with models.py:536-542 (unit weights); RA: :130-141 scores on the
    train-only graph -- when no proposal edges were added that IS the candidate graph (unit values), with 1/deg formed
    in float64 like the reference's int64 adjacency makes it, then rounded once to float32."""
    if args.model == "adamic_ogb":
        return node_weight_table(g, ops.W_AA)
    if args.model == "simple":
        return torch.ones(g.n_rows, dtype=torch.float32, device=g.device)
    if (args.model == "resource_allocation" and ra_graph is not None and g.val is None and ra_graph.val is None
            and g.nnz() == ra_graph.nnz() and torch.equal(g.rowptr, ra_graph.rowptr) and torch.equal(g.col, ra_graph.col)):
        return node_weight_table(ra_graph, ops.W_RA, f64=True).to(torch.float32)
    return None


GNN_HALF = True            # GNN filters under --keep_top on symmetric graphs: decode every unordered pair once (gnn_half_topk)
GNN_PRUNE_SLACK = 1 << 20  # gnn_half_topk re-cuts its kept pairs to the running bar when it holds more than 4 x ceil(K/2) + this


def gnn_half_topk(args, model, data, keep: int, rank: int, world: int):
    """The ``keep`` best proposals of a GNN filter, decoding each unordered candidate pair ONCE.

    The reference scores both orientations of a pair (filter.py:96-121), but LinkPredictor decodes h_u * h_v
    (models.py:478-485): the product commutes bit for bit, so score(u, v) == score(v, u).  On a symmetric pattern column v
    therefore lists only its candidates u < v (eps_expand_unit_* with revpos: half the list), the decoder runs on those
    (half the MFMA work -- the decode is all of this filter's time: 39.5 -> 20 s on the ppa stand-in), the pairs whose score
    reaches the running ceil(keep/2)-th best are kept, and the final selection mirrors them and orders the rows by the
    declared rule (score descending, then candidate order), exactly as the threshold scan's does (scan.select_topk).
    -> (pairs int64 [2,<=keep], scores float32), candidates seen (both orientations counted)."""
    g = data.adj_t
    dev = g.device
    revpos, md, sp = scan.reverse_positions(g), scan.max_degree(g), scan.window_splits(g)
    col_lo, col_hi = rank_column_range(g, rank, world)
    blocks = [(max(lo, col_lo), min(hi, col_hi)) for lo, hi in candidates.column_blocks(g) if lo < col_hi and hi > col_lo]
    k2 = (keep + 1) // 2
    keys_l, vals_l, held, bar, n_seen = [], [], 0, None, 0
    if world > 1 and hasattr(model, "embeddings"):
        model.embeddings(data.x, data.adj_t)         # the row-sharded forward holds a collective: every rank reaches it

    def prune():
        nonlocal keys_l, vals_l, held, bar
        keys, vals = torch.cat(keys_l), torch.cat(vals_l)
        if vals.numel() > k2:
            bar = ops.kth_largest(vals, k2)
            m = vals >= bar
            keys, vals = keys[m], vals[m]
        keys_l, vals_l, held = [keys], [vals], keys.numel()

    for v_lo, v_hi in blocks:
        r = ops.expand_unit(g.rowptr, g.col, None, g.n_rows, v_lo, v_hi, md, sp, want_score=False, want_v=True,
                            col_order=candidates.heaviest_first(g, v_lo, v_hi), revpos=revpos)
        pairs = r.pairs                                          # int32 [2, E]: (u; v), u < v
        if pairs.shape[1] == 0:
            continue
        n_seen += 2 * pairs.shape[1]
        sc = model(data.x, pairs, data.adj_t).reshape(-1)
        if bar is not None:
            m = sc >= bar
            pairs, sc = pairs[:, m], sc[m]
        keys_l.append((pairs[1].to(torch.int64) << 32) | pairs[0].to(torch.int64))
        vals_l.append(sc)
        held += sc.numel()
        if held > 4 * k2 + GNN_PRUNE_SLACK:
            prune()
    if keys_l:
        prune()
        keys, vals = keys_l[0], vals_l[0]
    else:
        keys, vals = torch.zeros(0, dtype=torch.int64, device=dev), torch.zeros(0, dtype=torch.float32, device=dev)
    if world > 1:
        keys, vals = scan._gather_varlen(keys, world), scan._gather_varlen(vals, world)
        n_seen = _sum_over_ranks(n_seen)
    keys, vals = scan.select_topk(keys, vals, keep, g.n_rows)
    return torch.stack([keys & 0xFFFFFFFF, keys >> 32]), vals, n_seen


def _sum_over_ranks(x: int) -> int:
    from . import dist as epd
    t = torch.tensor([x], dtype=torch.int64, device=torch.device("cuda", torch.cuda.current_device()))
    return int(sum(int(v.item()) for v in epd.all_gather_list(t)))


CUT_CAPACITY = 1 << 23     # survivors per block the expansion kernel may report (96 MB); more -> the block is redone in full


def scored_blocks(args, model, data, ra_graph, col_lo: int = 0, col_hi: int = None, bar=None):
    """(v_lo, v_hi, pairs, scores) per column block.  Heuristic filters whose scoring graph IS the candidate graph
    (AA: filter.py:122-126; CN 'simple': :116-121 with models.py:536-542) come out of the fused expansion already
    scored; RA scores on the train-only graph (filter.py:130-141) and GNN filters decode the block's pairs."""
    g = data.adj_t
    col_hi = g.n_rows if col_hi is None else col_hi
    blocks = [(max(lo, col_lo), min(hi, col_hi)) for lo, hi in candidates.column_blocks(g) if lo < col_hi and hi > col_lo]
    node_w = fused_node_weights(args, g, ra_graph) if candidates.hip_expand_available(g) else None
    if node_w is not None and not candidates.fused_scores_fit(g, node_w):
        # weights so large that a score could leave the fused kernels' fixed-point range: float32 / float64 pair kernels
        print(f'fused scoring disabled: score bound {candidates.fused_score_bound(g, node_w):.3e} >= 2^22')
        node_w = None
    if node_w is None:
        for v_lo, v_hi in blocks:
            pairs = candidates.expand_block(g, v_lo, v_hi)[0]      # HIP expansion (list only); scores from score_block
            yield v_lo, v_hi, pairs, (score_block(args, model, data, pairs, ra_graph) if pairs.shape[1] else None)
        return
    print(f'fused candidate generation + scoring ({args.model})')
    for v_lo, v_hi in blocks:
        # once the streaming top-K holds K proposals (``bar()`` is its K-th score) only candidates above that bar matter:
        # the kernel reports them directly and the block's score array is neither written nor scanned
        thr = bar() if bar is not None else None
        blk = None
        if thr is not None:
            # (nothing scans the block afterwards, so the count-free segment layout costs nothing and saves the counting pass)
            blk = candidates.expand_block_lazy(g, v_lo, v_hi, node_w, want_score=False, cut=(thr, CUT_CAPACITY),
                                               count_free=True)
            if blk.survivors is None:
                blk = None                           # more survivors than the list holds: score the block in full
        if blk is None:
            blk = candidates.expand_block_lazy(g, v_lo, v_hi, node_w, want_score=True)
        yield v_lo, v_hi, blk, blk.score


LAST_TIMING = {}      # the last run's scoring section: {"scored_s": host stopwatch, "gpu_ms": HIP events on the stream, "candidates": n}


class _Stopwatch:
    """The scoring section of a run on both clocks: host wall (what the stage prints) and a pair of HIP events on the current
    stream around the same section (bench_configs.py reports both: `wall_s` of a CLI leg also holds stand-in generation,
    checkpoint and file I/O, which say nothing about the engine)."""

    def __init__(self, device):
        self.device = device
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e1 = torch.cuda.Event(enable_timing=True)
        self.e0.record(torch.cuda.current_stream(device))
        self.t0 = time.perf_counter()

    def stop(self, candidates: int) -> float:
        self.e1.record(torch.cuda.current_stream(self.device))
        torch.cuda.synchronize(self.device)
        dt = time.perf_counter() - self.t0
        LAST_TIMING.clear()
        LAST_TIMING.update(scored_s=dt, gpu_ms=self.e0.elapsed_time(self.e1), candidates=int(candidates))
        return dt


def run(args) -> str:
    args = default_model_configs(args)
    print(args)
    Path("filtered_edges").mkdir(exist_ok=True)
    if not torch.cuda.is_available():
        raise RuntimeError("filter stage needs a HIP device: the scoring path has no CPU fallback")
    # one process per GPU under torchrun (WORLD_SIZE > 1): candidate COLUMNS are sharded, the graph is replicated, the
    # only exchange is the final merge (top-K lists under --keep_top, else the shards of the full list)
    from . import dist as epd
    rank, world, dist_dev = epd.init_from_env(args.dist_backend, args.device)
    device = dist_dev if world > 1 else torch.device(f'cuda:{args.device or 0}')
    _lib.warm_up_async(device)           # (code objects load in the background while the dataset is read on the host)
    ops._pinned_words([0])               # (... and the pinned staging ring of the survivor records is allocated now, not inside the timed section)

    edge_index, edge_weight, split_edge, data = get_data(args)
    data = data.to(device)
    model = build_model(args, data, device)
    print(f'using model {model}')
    use_params = sum(p.numel() for p in model.parameters() if p.requires_grad) > 0
    print('using params?', use_params)
    if use_params:
        model.load_state_dict(torch.load(f'models/{args.checkpoint}', map_location=device))

    parts = args.checkpoint.split("|")
    spec, sorted_edge_path, num_sorted_edge = parts[0], parts[1], int(parts[2])
    run_id = parts[3].split(".")[0]
    if sorted_edge_path:
        print("Loading corresponding extra edges from ", sorted_edge_path)
        extra_edges = proposals.load_proposals(f"filtered_edges/{sorted_edge_path}.pt", num_sorted_edge)
        assert extra_edges.size(0) == 2 and extra_edges.size(1) == num_sorted_edge
    else:
        extra_edges = torch.zeros([2, 0], dtype=torch.long)
    data.adj_t = add_edges(args.dataset, edge_index.to(device), edge_weight.to(device), extra_edges.to(device),
                           data.num_nodes)
    model.eval()
    ra_graph = train_only_graph(split_edge, data.num_nodes, device) if args.model == "resource_allocation" else None

    _lib.warm_up_join()                  # (the background loads are done -- or given up on -- before anything is timed)
    watch = _Stopwatch(device)
    keep = int(args.keep_top)
    if keep == 0 and args.model == "simple" and world == 1 and candidates.dense_cn_suits(data.adj_t):
        # configs[0]: a small DENSE graph (ddi: N = 4,267, 11.7 % of all pairs are edges).  Common-neighbour counts = A A^T, ONE
        # product on the matrix cores (exact: counts < 2^24), the reference's candidate list a masked read of it in its own
        # column-major order, and the file's order ONE stable sort by the integer count -- no per-graph table, a dozen launches
        # (csrc/dense_cn.hip; the sparse path took 7.5 ms of GPU time in hundreds of launches for the same 16 M rows)
        g = data.adj_t
        with torch.no_grad():
            got = ops.dense_cn_candidates(g.rowptr, g.col, g.n_rows, directed=True, check_symmetric=True, as_rows=True)
        if got is not None:                      # (an asymmetric pattern takes the general path below)
            rows, cnt = got
            rows = rows[torch.sort(cnt.to(torch.int16), descending=True, stable=True).indices]  # counts <= N - 2 < 2^15
            n_seen = int(rows.shape[0])
            dt = watch.stop(n_seen)
            print('dense common-neighbour product (A A^T on the f32 MFMA)')
            print(f'using {n_seen} edges; scored in {dt:.2f} s ({n_seen / max(dt, 1e-9):.3e} candidate edges/s incl. generation)')
            return _save(args, spec, sorted_edge_path, num_sorted_edge, run_id, rank, world, rows)
    if 0 < keep <= scan.MAX_K and scan.scan_plausible(data.adj_t):
        # (the hubs-first copy first: the symmetry check of scan_available then reads the COPY's reverse positions -- the table
        #  the scan needs anyway -- instead of building one for the graph as labelled, 3-5 ms on a ppa-sized graph)
        scan.scan_graph(data.adj_t, build=True)
    scan_w = fused_node_weights(args, data.adj_t, ra_graph) if 0 < keep <= scan.MAX_K and scan.scan_available(data.adj_t) else None
    if scan_w is not None and not scan.scan_usable(data.adj_t, scan_w):
        scan_w = None                    # sums could leave the scan's fixed-point range: the pair kernels score this graph
    if scan_w is not None:
        # --keep_top with a heuristic filter on a symmetric graph (unit-valued, or collab's summed multi-edge weights): one
        # threshold scan of the whole candidate set (csrc/scan_pieces.hip / filter_scan.hip) instead of candidate blocks +
        # streaming top-K; every rank ends with the same list
        # (count=False: a launch with skipped heads counts the candidates its walk TOUCHES; the exact size of the candidate set
        #  would be one more scan of the graph -- a fifth of this one-shot run -- for a number that is only printed)
        st = {"count": False}
        with torch.no_grad():
            # (the file is written by rank 0: the ordered rows travel there alone)
            shards = world > 1 and bool(getattr(args, "shard_proposals", False))
            best_pairs, best_scores = scan.scan_topk(data.adj_t, scan_w, keep, rank, world, stats=st, relabel=True,
                                                     rows_on="shards" if shards else (0 if world > 1 else None))
        n_seen = st["candidates"] if st["candidates"] is not None else st["touched"]
        dt = watch.stop(n_seen)
        bar = None if st["bar"] is None else float(st["bar"])
        print(f'threshold scan ({args.model}): bar {bar}, {st["survivors"]} survivors, {st["launches"]} launches'
              + (f', heads skipped under a budget of {st["head_budget"]:.4f}' if st.get("heads") else ''))
        print(f'using {"at least " if st["candidates"] is None else ""}{n_seen} edges; scored in {dt:.2f} s '
              f'({n_seen / max(dt, 1e-9):.3e} candidate edges/s incl. generation)'
              + (' -- every 2-hop non-edge is covered by the bound; the count is of those the walk touched' if st["candidates"] is None else ''))
        return _save(args, spec, sorted_edge_path, num_sorted_edge, run_id, rank, world,
                     None if best_pairs is None else torch.cat([best_pairs.t().to(torch.float32), best_scores.unsqueeze(1)], 1),
                     sharded=shards)
    full_w = (fused_node_weights(args, data.adj_t, ra_graph)
              if keep == 0 and data.adj_t.val is None and scan.scan_available(data.adj_t) else None)
    if (full_w is not None and candidates.fused_scores_fit(data.adj_t, full_w)
            and int(scan.half_paths(data.adj_t).sum().item()) < 1 << 29):      # (unordered pairs <= half paths: lists that fit)
        # the whole [E,3] file of a heuristic filter on a unit-valued symmetric graph: scores are symmetric too, so the list
        # kernels write each unordered pair once (column v: its candidates u < v), and the rows -- both orientations -- come
        # out of the same mirror + declared-order sort the threshold scan's selection uses
        g = data.adj_t
        col_lo, col_hi = rank_column_range(g, rank, world)
        keys_l, vals_l = [], []
        with torch.no_grad():
            for lo, hi in candidates.column_blocks(g):
                lo, hi = max(lo, col_lo), min(hi, col_hi)
                if lo >= hi:
                    continue
                r = ops.expand_unit(g.rowptr, g.col, full_w, g.n_rows, lo, hi, scan.max_degree(g), scan.window_splits(g),
                                    col_order=candidates.heaviest_first(g, lo, hi), revpos=scan.reverse_positions(g))
                keys_l.append((r.pairs[1].to(torch.int64) << 32) | r.pairs[0].to(torch.int64))
                vals_l.append(r[4])
        keys = torch.cat(keys_l) if keys_l else torch.zeros(0, dtype=torch.int64, device=device)
        vals = torch.cat(vals_l) if vals_l else torch.zeros(0, dtype=torch.float32, device=device)
        if world > 1:
            keys, vals = scan._gather_varlen(keys, world), scan._gather_varlen(vals, world)
        if keys.numel():
            rows_k, rows_v = scan.select_topk(keys, vals, 2 * keys.numel(), g.n_rows)
            n_seen = rows_k.numel()
            dt = watch.stop(n_seen)
            print(f'using {n_seen} edges; scored in {dt:.2f} s ({n_seen / max(dt, 1e-9):.3e} candidate edges/s incl. generation)')
            rows = torch.stack([(rows_k & 0xFFFFFFFF).to(torch.float32), (rows_k >> 32).to(torch.float32), rows_v], 1)
            return _save(args, spec, sorted_edge_path, num_sorted_edge, run_id, rank, world, rows)
    from .models import LinkGNN
    if (GNN_HALF and 0 <= keep <= scan.MAX_K and isinstance(model, LinkGNN) and data.adj_t.device.type == "cuda"
            and data.adj_t.n_rows == data.adj_t.n_cols and data.adj_t.nnz() < 1 << 30 and scan.is_symmetric(data.adj_t)
            and (keep > 0 or int(scan.half_paths(data.adj_t).sum().item()) < 1 << 29)):     # (the whole file: lists that fit)
        with torch.no_grad():
            best_pairs, best_scores, n_seen = gnn_half_topk(args, model, data, keep if keep else scan.MAX_K, rank, world)
        dt = watch.stop(n_seen)
        print(f'GNN filter, each unordered pair decoded once ({args.model})')
        print(f'using {n_seen} edges; scored in {dt:.2f} s ({n_seen / max(dt, 1e-9):.3e} candidate edges/s incl. generation)')
        return _save(args, spec, sorted_edge_path, num_sorted_edge, run_id, rank, world,
                     torch.cat([best_pairs.t().to(torch.float32), best_scores.unsqueeze(1)], 1))
    col_lo, col_hi = rank_column_range(data.adj_t, rank, world)
    n_seen = 0
    all_pairs, all_scores = [], []
    top = proposals.StreamingTopK(keep) if keep else None
    if world > 1 and hasattr(model, "embeddings"):
        # the row-sharded GNN forward holds a collective: every rank must reach it, whatever its column shard yields
        with torch.no_grad():
            model.embeddings(data.x, data.adj_t)
    with torch.no_grad():
        for v_lo, v_hi, pairs, score in scored_blocks(args, model, data, ra_graph, col_lo, col_hi,
                                                      bar=top.bar if keep else None):
            n_blk = pairs.numel() if (not isinstance(pairs, torch.Tensor)) else pairs.shape[1]
            if n_blk == 0:
                continue
            if keep:
                top.push(pairs, score)          # blocks arrive in candidate (column-major) order
            elif not isinstance(pairs, torch.Tensor):     # candidates.ColumnBlock, possibly padded
                idx = pairs.valid()
                all_pairs.append(pairs.select(idx))
                all_scores.append(score[idx])
            else:
                all_pairs.append(pairs)
                all_scores.append(score)
            n_seen += n_blk
    dt = watch.stop(n_seen)
    print(f'using {n_seen} edges; scored in {dt:.2f} s ({n_seen / max(dt, 1e-9):.3e} candidate edges/s incl. generation)')

    if keep:
        best_pairs, best_scores = top.result()
        best_pairs, best_scores = best_pairs.to(device), best_scores.to(device)
        if world > 1:
            # ranks hold contiguous column ranges in rank order: gather the (padded) per-rank lists -- K x 20 B each --
            # and stable-merge them in rank order
            n_mine = torch.tensor([best_scores.numel()], dtype=torch.int64, device=device)
            n_all = epd.all_gather_list(n_mine)
            pad_s = torch.full((keep,), float("-inf"), dtype=torch.float32, device=device)
            pad_p = torch.zeros((2, keep), dtype=torch.int64, device=device)
            pad_s[:best_scores.numel()] = best_scores
            pad_p[:, :best_pairs.shape[1]] = best_pairs
            gs = epd.all_gather_list(pad_s)
            gp = epd.all_gather_list(pad_p)
            cnt = [int(x.item()) for x in n_all]
            best_pairs, best_scores = proposals.merge_ranked_lists([gp[r][:, :cnt[r]] for r in range(world)],
                                                                   [gs[r][:cnt[r]] for r in range(world)], keep)
        sorted_edges = torch.cat([best_pairs.t().to(torch.float32), best_scores.unsqueeze(1)], 1)
    else:
        pairs = torch.cat(all_pairs, 1) if all_pairs else torch.zeros((2, 0), dtype=torch.int64, device=device)
        scores = torch.cat(all_scores) if all_scores else torch.zeros(0, dtype=torch.float32, device=device)
        if world > 1:
            # the full list: ranks hold contiguous column ranges in rank order, so the shards concatenated in rank order ARE
            # the single-process candidate order -- one variable-length all-gather (20 B per candidate; sized for lists that
            # fit one GPU, which is what a file of all [E,3] rows presupposes), then the same sort on every rank
            pairs = torch.stack([scan._gather_varlen(pairs[0].contiguous(), world),
                                 scan._gather_varlen(pairs[1].contiguous(), world)])
            scores = scan._gather_varlen(scores, world)
        sorted_edges = proposals.sorted_edges_tensor(pairs, scores)          # filter.py:160-161
    return _save(args, spec, sorted_edge_path, num_sorted_edge, run_id, rank, world, sorted_edges)


def _save(args, spec, sorted_edge_path, num_sorted_edge, run_id, rank, world, sorted_edges, sharded: bool = False) -> str:
    filename = f'filtered_edges/{spec}_{sorted_edge_path}_{num_sorted_edge}_{run_id}_sorted_edges.pt'
    if sharded:
        # (every rank holds the chunk of the sorted list it ordered: rank r's rows follow rank r - 1's -- proposals.load_sorted_edges)
        import os
        if rank == 0 and os.path.exists(filename):
            os.remove(filename)                      # (a list of an earlier, unsharded run would shadow the shards)
        proposals.save_sorted_edges_shard(filename, sorted_edges, rank, world)
        print(f"rank {rank}: {sorted_edges.shape[0]} rows to {proposals.shard_path(filename, rank, world)}")
    elif rank == 0:
        print(sorted_edges)
        proposals.save_sorted_edges(filename, sorted_edges)                    # filter.py:164-165
        print("Saving to ", filename)
    if world > 1:
        torch.distributed.barrier()
    return filename


def main(argv=None):
    return run(make_parser().parse_args(argv))
