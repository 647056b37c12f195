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
        args.out_name = args.dataset + "_" + str(args.model)
        if args.only_supervision:
            args.out_name += "_onlys"
        elif args.also_supervision:
            args.out_name += "_alsos"
        elif args.valid_proposal:
            args.out_name += "_validproposal"

    edge_index, edge_weight, split_edge, data = get_data(args)
    if args.gen_dataset_only:
        return []
    if args.model is None:
        raise ValueError("Model not specified")
    data = data.to(device)
    model = build_model(args, data, device)
    print(f'using model {model}')
    evaluator = evaluators[args.dataset]
    K = hits[args.dataset]
    print("Evaluating at hits: ", K)

    if args.sorted_edge_path:
        sorted_test_edges = torch.load(f"filtered_edges/{args.sorted_edge_path}")
        print('sorted test edges', sorted_test_edges.size())
        if args.valid_proposal:
            sorted_test_edges = splice_valid_proposals(sorted_test_edges, split_edge['valid']['edge'])
    else:
        sorted_test_edges = torch.zeros(42, 2)

    index_ends = sweep_schedule(args)
    print(f"Scheduled extra edges sweep: {index_ends} x {args.runs}")
    use_params = sum(p.numel() for p in model.parameters() if p.requires_grad) > 0
    trained = use_params and not args.load_model     # rank.py:317-333: reset + train every run
    if use_params and args.load_model:
        model.load_state_dict(torch.load(args.load_model, map_location=device))
    ei_dev, ew_dev = edge_index.to(device), edge_weight.to(device)

    curves = []
    for index_end in index_ends:
        loggers = {f'Hits@{k}': Logger(args.runs, args) for k in K}
        print('---------------------')
        print(f'Using {index_end} highest scoring edges')
        print('---------------------')
        extra_edges = sorted_test_edges[:int(index_end), :2].t().long()
        assert extra_edges.size(0) == 2 and extra_edges.size(1) == index_end
        if not args.only_supervision:
            data.adj_t = add_edges(args.dataset, ei_dev, ew_dev, extra_edges.to(device), data.num_nodes)
        append_supervision(args, split_edge, extra_edges)
        if args.dataset in ["collab", "email", "reddit"]:
            val_edge_index = to_undirected(split_edge['valid']['edge'].t())
            full_extra_edges = torch.cat([extra_edges, val_edge_index], dim=-1)
            data.full_adj_t = add_edges(args.dataset, ei_dev, ew_dev, full_extra_edges.to(device), data.num_nodes)
        else:
            data.full_adj_t = data.adj_t

        curve_point = []
        for run_i in range(args.runs):
            optimizer = None
            if not use_params:
                model.reset_parameters()
                args.epochs = 1
            elif trained:
                model.reset_parameters()                                    # rank.py:318
                optimizer = torch.optim.Adam(model.parameters(), lr=args.lr)   # rank.py:322
            highest_eval = 0
            for epoch in range(1, 1 + (args.epochs or 1)):
                loss = -1
                if trained:
                    loss = train(model, data, args.dataset, split_edge, optimizer, args.batch_size, use_params, args.model,
                                 device)
                if epoch % args.eval_steps == 0:
                    if args.model == "adamic_ogb":
                        results = test_adamic(model, data, split_edge, evaluator, args.batch_size, args, device)
                    elif args.model == "resource_allocation":
                        results = test_resource_allocation(model, data, split_edge, evaluator, args.batch_size, args,
                                                           device)
                    elif args.model == "katz":
                        raise NotImplementedError("katz (sparse inverse) is outside the accelerated path")
                    else:
                        results = test(model, data, split_edge, evaluator, args.batch_size or (1 << 16), args, device)
                    for key, result in results.items():
                        loggers[key].add_result(run_i, result)
                    if epoch % args.log_steps == 0:
                        for key, result in results.items():
                            train_hits, valid_hits, test_hits = result
                            if key == f"Hits@{K[1]}" and valid_hits >= highest_eval:
                                highest_eval = valid_hits
                                if args.save_models and use_params:          # rank.py:356-361
                                    fn = f'{args.out_name}|{args.sorted_edge_path.split(".")[0]}|{index_end}|{run_i}.pt'
                                    torch.save(model.state_dict(), os.path.join('models', fn))
                            print(key)
                            print(f'Run: {run_i + 1:02d}, Epoch: {epoch:02d}, Loss: {loss:.4f}, '
                                  f'Train: {100 * train_hits:.2f}%, Valid: {100 * valid_hits:.2f}%, '
                                  f'Test: {100 * test_hits:.2f}%')
                        print('---')
                if use_params and not trained:
                    break  # a loaded model is evaluated once
            for key in loggers.keys():
                print(key)
                loggers[key].print_statistics(run_i)
                if key == f"Hits@{K[1]}":                          # model selection on the MIDDLE K (rank.py:376)
                    result = 100 * torch.tensor(loggers[key].results[run_i])
                    argmax = result[:, 1].argmax().item()
                    curve_point = [index_end, result[argmax, 1], result[argmax, 2]]
            stamp = datetime.now().strftime('%Y-%m-%d-%H:%M:%S')
            filename = f'{args.out_name}|{args.sorted_edge_path.split(".")[0]}|{index_end}|{stamp}.pt'
            print(curve_point)
            print("Saving curve to ", filename)
            torch.save(curve_point, os.path.join('curves', filename))
            curves.append(curve_point)
        for key in loggers.keys():
            print(key)
            loggers[key].print_statistics()
    return curves


def main(argv=None):
    return run(make_parser().parse_args(argv))
