"""The rank stage (drop-in for rank.py:129-391): evaluate a rank model on the graph augmented with the top-k
proposal edges, over the reference's sweep schedule, and write the curve points.

Same flags (rank.py:130-163), same proposal consumption (``[:k,:2].t().long()``, :294), same adjacency
construction (:300, :307-314), same eval dispatch (:337-349), same stdout lines (:364-369), ``models/`` checkpoints
(:356-361) and ``curves/`` files (:381-385).  Parametrised rank models are trained per run like rank.py:317-333
(training.py: torch autograd over the HIP SpMM), or evaluated from ``--load_model <state_dict.pt>``.
"""
from __future__ import annotations

import argparse
import os
from datetime import datetime
from pathlib import Path

import torch

from .datasets import get_data
from .evaluate import evaluators, hits, test, test_adamic, test_resource_allocation
from .graph import add_edges
from .runlog import Logger
from .models import build_model, default_model_configs
from .rank_helpers import to_undirected
from .training import train


def make_parser():
    parser = argparse.ArgumentParser(description='rank stage (MI355X)')
    parser.add_argument('--dataset', type=str, required=True)
    parser.add_argument('--model', type=str)
    parser.add_argument('--runs', type=int, default=10)
    parser.add_argument('--sorted_edge_path', type=str, default="")
    parser.add_argument('--num_sorted_edge', type=int)
    parser.add_argument('--sweep_max', type=int)
    parser.add_argument('--sweep_min', type=int)
    parser.add_argument('--sweep_num', type=int)
    parser.add_argument('--only_supervision', action="store_true", default=False)
    parser.add_argument('--also_supervision', action="store_true", default=False)
    parser.add_argument('--gen_dataset_only', action="store_true", default=False)
    parser.add_argument('--valid_proposal', action="store_true", default=False)
    parser.add_argument('--out_name', type=str)
    parser.add_argument('--save_models', action="store_true", default=False)
    parser.add_argument('--num_layers', type=int)
    parser.add_argument('--hidden_channels', type=int)
    parser.add_argument('--dropout', type=float)
    parser.add_argument('--batch_size', type=int)
    parser.add_argument('--lr', type=float)
    parser.add_argument('--epochs', type=int)
    parser.add_argument('--use_feature', type=bool)
    parser.add_argument('--use_learnable_embedding', type=bool)
    parser.add_argument('--device', type=int, default=0)
    parser.add_argument('--log_steps', type=int, default=1)
    parser.add_argument('--eval_steps', type=int, default=1)
    # extensions
    parser.add_argument('--synthetic', action="store_true", default=False)
    parser.add_argument('--load_model', type=str, default="")
    return parser


def splice_valid_proposals(sorted_test_edges: torch.Tensor, valid_pos: torch.Tensor) -> torch.Tensor:
    """rank.py:222-251 (--valid_proposal): both directions of every validation edge go on top with score 100000, and
    proposal rows that are validation edges (either direction) are dropped.  Vectorised (the reference walks the
    proposal file row by row in Python); the reference builds the top block from a Python set, whose iteration order
    is arbitrary -- here it is sorted, which gives the same adjacency."""
    v = valid_pos.long()
    n = int(max(v.max().item() if v.numel() else 0, sorted_test_edges[:, :2].max().item() if sorted_test_edges.numel() else 0)) + 1
    both = torch.cat([v, v.flip(1)], 0)
    vkeys = torch.unique(both[:, 0] * n + both[:, 1])                     # directed keys of both directions, sorted
    top = torch.stack([torch.div(vkeys, n, rounding_mode="floor"), vkeys % n], 1).to(torch.float64)
    top = torch.cat([top, torch.full((top.shape[0], 1), 100000.0, dtype=torch.float64)], 1)
    p = sorted_test_edges.to(torch.float64)
    if p.shape[1] == 2:
        p = torch.cat([p, torch.zeros((p.shape[0], 1), dtype=torch.float64)], 1)
    pkeys = p[:, 0].long() * n + p[:, 1].long()
    keep = ~torch.isin(pkeys, vkeys)                                       # vkeys holds both directions already
    return torch.cat([top, p[keep, :3]], 0)


def append_supervision(args, split_edge, extra_edges: torch.Tensor) -> None:
    """rank.py:303-304: with ``--only_supervision`` / ``--also_supervision`` the proposal edges become additional
    training POSITIVES (appended to ``split_edge['train']['edge']``; like the reference, the append accumulates over the
    sweep points of one invocation)."""
    if args.only_supervision or args.also_supervision:
        tr = split_edge['train']['edge']
        split_edge['train']['edge'] = torch.cat((tr, extra_edges.t().to(tr.device, tr.dtype)))


def sweep_schedule(args):
    """rank.py:260-272."""
    index_ends = []
    if args.sweep_num:
        if args.sweep_min is None:
            args.sweep_min = 0
        if args.sweep_max is None:
            args.sweep_max = (args.sweep_num - 1) * 1000
        for i in range(args.sweep_num + 1):
            index_ends.append(args.sweep_min + int(i * (args.sweep_max - args.sweep_min) / args.sweep_num))
    elif args.num_sorted_edge:
        index_ends.append(args.num_sorted_edge)
    else:
        index_ends.append(0)
    return index_ends


def _default_out_name(args) -> str:
    suffix = ("_onlys" if args.only_supervision else "_alsos" if args.also_supervision else
              "_validproposal" if args.valid_proposal else "")
    return f"{args.dataset}_{args.model}{suffix}"


def _load_proposals(args, split_edge) -> torch.Tensor:
    """The proposal file of the filter stage (rank.py:214-257): float rows (u, v, score), best first."""
    if not args.sorted_edge_path:
        return torch.zeros(42, 2)
    from . import proposals                      # (a sharded filter run may have left <file>.shard{r}of{N}: read back in rank order)
    rows = proposals.load_sorted_edges(f"filtered_edges/{args.sorted_edge_path}")
    print('sorted test edges', rows.size())
    return splice_valid_proposals(rows, split_edge['valid']['edge']) if args.valid_proposal else rows


_HEURISTIC_EVAL = {"adamic_ogb": test_adamic, "resource_allocation": test_resource_allocation}


def _evaluate(args, model, data, split_edge, evaluator, device):
    """{Hits@K: (train, valid, test)} of the model on the current graphs (the dispatch of rank.py:337-349)."""
    if args.model == "katz":
        raise NotImplementedError("katz (sparse inverse) is outside the accelerated path")
    if args.model in _HEURISTIC_EVAL:
        return _HEURISTIC_EVAL[args.model](model, data, split_edge, evaluator, args.batch_size, args, device)
    return test(model, data, split_edge, evaluator, args.batch_size or (1 << 16), args, device)


class _SweepPoint:
    """One point of the sweep: the graphs with the ``index_end`` best proposals added (rank.py:294-314), the per-K loggers of its
    runs, and -- per run -- the model-selection state: the MIDDLE K's validation score decides which checkpoint is kept
    (rank.py:356-361) and which evaluation becomes the run's curve point (rank.py:376-379)."""

    def __init__(self, args, index_end: int, ks, proposals, data, split_edge, ei_dev, ew_dev, device):
        self.args, self.index_end = args, int(index_end)
        self.select_key = f"Hits@{ks[1]}"
        self.loggers = {f'Hits@{k}': Logger(args.runs, args) for k in ks}
        extra = proposals[:self.index_end, :2].t().long()
        assert extra.size(0) == 2 and extra.size(1) == self.index_end
        if not args.only_supervision:
            data.adj_t = add_edges(args.dataset, ei_dev, ew_dev, extra.to(device), data.num_nodes)
        append_supervision(args, split_edge, extra)
        data.full_adj_t = data.adj_t
        if args.dataset in ("collab", "email", "reddit"):       # validation edges join the graph the TEST edges are scored on
            with_valid = torch.cat([extra, to_undirected(split_edge['valid']['edge'].t())], dim=-1)
            data.full_adj_t = add_edges(args.dataset, ei_dev, ew_dev, with_valid.to(device), data.num_nodes)
        self.best_valid = 0

    def start_run(self) -> None:
        self.best_valid = 0

    def record(self, run_i: int, results) -> None:
        for key, triple in results.items():
            self.loggers[key].add_result(run_i, triple)

    def is_best_logged(self, results) -> bool:
        """True when this LOGGED evaluation's selection score does not fall behind the run's best logged one (rank.py:352-361
        looks at the score only on epochs it logs)."""
        valid = results[self.select_key][1] if self.select_key in results else None
        if valid is None or valid < self.best_valid:
            return False
        self.best_valid = valid
        return True

    def checkpoint_name(self, run_i) -> str:
        return f'{self.args.out_name}|{self.args.sorted_edge_path.split(".")[0]}|{self.index_end}|{run_i}.pt'

    def curve_point(self, run_i: int):
        evals = 100 * torch.tensor(self.loggers[self.select_key].results[run_i])
        best = evals[:, 1].argmax().item()
        return [self.index_end, evals[best, 1], evals[best, 2]]

    def print_run(self, run_i: int) -> None:
        for key, lg in self.loggers.items():
            print(key)
            lg.print_statistics(run_i)

    def print_all(self) -> None:
        for key, lg in self.loggers.items():
            print(key)
            lg.print_statistics()


def _print_epoch(results, run_i: int, epoch: int, loss: float) -> None:
    for key, (train_hits, valid_hits, test_hits) in results.items():
        print(key)
        print(f'Run: {run_i + 1:02d}, Epoch: {epoch:02d}, Loss: {loss:.4f}, '
              f'Train: {100 * train_hits:.2f}%, Valid: {100 * valid_hits:.2f}%, '
              f'Test: {100 * test_hits:.2f}%')
    print('---')


def run(args):
    args = default_model_configs(args)
    print(args)
    if not torch.cuda.is_available():
        raise RuntimeError("rank stage needs a HIP device: the scoring path has no CPU fallback")
    device = torch.device(f'cuda:{args.device}')
    from . import _lib
    _lib.warm_up_async(device)           # (code objects load in the background while the dataset is read on the host)
    Path("curves").mkdir(exist_ok=True)
    Path("models").mkdir(exist_ok=True)
    assert not (args.only_supervision and args.also_supervision)
    if args.out_name is None:
        args.out_name = _default_out_name(args)

    edge_index, edge_weight, split_edge, data = get_data(args)
    if args.gen_dataset_only:
        return []
    if args.model is None:
        raise ValueError("Model not specified")
    data = data.to(device)
    model = build_model(args, data, device)
    print(f'using model {model}')
    evaluator, ks = evaluators[args.dataset], hits[args.dataset]
    print("Evaluating at hits: ", ks)
    proposals = _load_proposals(args, split_edge)
    index_ends = sweep_schedule(args)
    print(f"Scheduled extra edges sweep: {index_ends} x {args.runs}")

    # what a run does with the model: heuristics are evaluated once; a parametrised model is trained from scratch every run
    # (rank.py:317-333) unless --load_model brings its weights, in which case it is evaluated once as well
    has_params = sum(p.numel() for p in model.parameters() if p.requires_grad) > 0
    trains = has_params and not args.load_model
    if has_params and args.load_model:
        model.load_state_dict(torch.load(args.load_model, map_location=device))
    if not has_params:
        args.epochs = 1
    n_epochs = (args.epochs or 1) if trains or not has_params else 1
    ei_dev, ew_dev = edge_index.to(device), edge_weight.to(device)
    _lib.warm_up_join()

    curves = []
    for index_end in index_ends:
        print('---------------------')
        print(f'Using {index_end} highest scoring edges')
        print('---------------------')
        point = _SweepPoint(args, index_end, ks, proposals, data, split_edge, ei_dev, ew_dev, device)
        for run_i in range(args.runs):
            point.start_run()
            optimizer = None
            if trains or not has_params:
                model.reset_parameters()                                          # rank.py:318
            if trains:
                optimizer = torch.optim.Adam(model.parameters(), lr=args.lr)      # rank.py:322
            for epoch in range(1, n_epochs + 1):
                loss = (train(model, data, args.dataset, split_edge, optimizer, args.batch_size, has_params, args.model, device)
                        if trains else -1)
                if epoch % args.eval_steps:
                    continue
                results = _evaluate(args, model, data, split_edge, evaluator, device)
                point.record(run_i, results)
                if epoch % args.log_steps == 0:
                    if point.is_best_logged(results) and args.save_models and has_params:
                        torch.save(model.state_dict(), os.path.join('models', point.checkpoint_name(run_i)))
                    _print_epoch(results, run_i, epoch, loss)
            point.print_run(run_i)
            curve_point = point.curve_point(run_i)
            stamp = datetime.now().strftime('%Y-%m-%d-%H:%M:%S')
            filename = f'{args.out_name}|{args.sorted_edge_path.split(".")[0]}|{index_end}|{stamp}.pt'
            print(curve_point)
            print("Saving curve to ", filename)
            torch.save(curve_point, os.path.join('curves', filename))
            curves.append(curve_point)
        point.print_all()
    return curves


def main(argv=None):
    return run(make_parser().parse_args(argv))
