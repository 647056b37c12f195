"""Evaluation harness of the rank stage with the reference's names and signatures
(train_and_eval.py:11-29, :98-156, :158-193, :218-270; adamic_utils.py:27-68), on the HIP scoring
path.  ``Evaluator`` restates ogb 1.3.1's Hits@K [third-party, parity unpinned]:
``kth = topk(y_pred_neg, K)[-1]; hits = mean(y_pred_pos > kth)``; 1.0 when there are fewer than K
negatives; strict ``>``.
"""
from __future__ import annotations

import numpy as np
import torch

from .heuristics import AA, get_A, resource_allocation


class Evaluator:
    """ogb.linkproppred.Evaluator for the Hits@K datasets (ddi: K=20, collab: 50, ppa: 100)."""
    _DEFAULT_K = {"ogbl-ddi": 20, "ogbl-collab": 50, "ogbl-ppa": 100}

    def __init__(self, name: str):
        self.name = name
        self.K = self._DEFAULT_K.get(name, 50)

    def eval(self, input_dict):
        pos, neg = input_dict["y_pred_pos"], input_dict["y_pred_neg"]
        if isinstance(pos, np.ndarray):
            pos, neg = torch.from_numpy(pos), torch.from_numpy(neg)
        if len(neg) < self.K:
            return {f"hits@{self.K}": 1.0}
        kth = torch.topk(neg.reshape(-1), self.K)[0][-1]
        hitsK = float(torch.sum(pos.reshape(-1) > kth).cpu()) / len(pos)
        return {f"hits@{self.K}": hitsK}


evaluators = {                                     # train_and_eval.py:11-19
    "collab": Evaluator(name='ogbl-collab'),
    "reddit": Evaluator(name='ogbl-collab'),
    "ddi": Evaluator(name='ogbl-ddi'),
    "ppa": Evaluator(name='ogbl-ppa'),
    "email": Evaluator(name='ogbl-ddi'),
    "twitch": Evaluator(name='ogbl-ddi'),
    "fb": Evaluator(name='ogbl-collab'),
}
hits = {                                           # train_and_eval.py:20-29
    "collab": [10, 50, 100],
    "reddit": [10, 50, 100],
    "ppa": [10, 100, 200],
    "ddi": [10, 20, 30],
    "email": [10, 20, 30],
    "twitch": [10, 50, 100],
    "fb": [10, 20, 30],
}


def _hits_table(dataset, evaluator, pos_train_pred, pos_valid_pred, neg_valid_pred, pos_test_pred, neg_test_pred):
    results = {}
    for K in hits[dataset]:
        evaluator.K = K
        train_hits = evaluator.eval({'y_pred_pos': pos_train_pred, 'y_pred_neg': neg_valid_pred})[f'hits@{K}']
        valid_hits = evaluator.eval({'y_pred_pos': pos_valid_pred, 'y_pred_neg': neg_valid_pred})[f'hits@{K}']
        test_hits = evaluator.eval({'y_pred_pos': pos_test_pred, 'y_pred_neg': neg_test_pred})[f'hits@{K}']
        results[f'Hits@{K}'] = (train_hits, valid_hits, test_hits)
    return results


def _score_all(model, x, edge_list, adj, batch_size):
    """One of the five scoring loops of train_and_eval.py:108-136.  ``edge_list`` is [E,2]; batches only bound
    the size of a launch (results do not depend on them)."""
    preds = []
    n = edge_list.size(0)
    step = max(int(batch_size), 1 << 22)
    for s in range(0, n, step):
        edge = edge_list[s:s + step].t()
        preds.append(model(x, edge, adj).reshape(-1))
    return torch.cat(preds, 0).cpu() if preds else torch.zeros(0)


@torch.no_grad()
def test(model, data, split_edge, evaluator, batch_size, args, device):
    """train_and_eval.py:98-156.  pos_train / pos_valid / neg_valid are scored on ``data.adj_t``;
    pos_test / neg_test on ``data.full_adj_t`` (:129, :135)."""
    model.eval()
    e = {k: split_edge[k] for k in ('eval_train', 'valid', 'test')}
    pos_train_pred = _score_all(model, data.x, e['eval_train']['edge'].to(device), data.adj_t, batch_size)
    pos_valid_pred = _score_all(model, data.x, e['valid']['edge'].to(device), data.adj_t, batch_size)
    neg_valid_pred = _score_all(model, data.x, e['valid']['edge_neg'].to(device), data.adj_t, batch_size)
    pos_test_pred = _score_all(model, data.x, e['test']['edge'].to(device), data.full_adj_t, batch_size)
    neg_test_pred = _score_all(model, data.x, e['test']['edge_neg'].to(device), data.full_adj_t, batch_size)
    return _hits_table(args.dataset, evaluator, pos_train_pred, pos_valid_pred, neg_valid_pred, pos_test_pred,
                       neg_test_pred)


def get_pos_neg_edges(split, split_edge, edge_index, num_nodes, percent=100):
    """adamic_utils.py:27-68, 'edge' branch for valid/test: [2,E] pos/neg lists after the seed-123
    permutation (order-invariant for Hits@K; kept so per-edge outputs line up with the reference)."""
    pos_edge = split_edge[split]['edge'].t()
    neg_edge = split_edge[split]['edge_neg'].t()
    np.random.seed(123)
    num_pos = pos_edge.size(1)
    perm = np.random.permutation(num_pos)[:int(percent / 100 * num_pos)]
    pos_edge = pos_edge[:, perm]
    np.random.seed(123)
    num_neg = neg_edge.size(1)
    perm = np.random.permutation(num_neg)[:int(percent / 100 * num_neg)]
    neg_edge = neg_edge[:, perm]
    return pos_edge, neg_edge


def test_adamic(model, data, split_edge, evaluator, batch_size, args, device):
    """train_and_eval.py:158-193."""
    assert args.model == "adamic_ogb"
    A_eval = get_A(data.adj_t, data.num_nodes)
    A = get_A(data.full_adj_t, data.num_nodes)
    pos_val_edge, neg_val_edge = get_pos_neg_edges('valid', split_edge, data.edge_index, data.num_nodes)
    pos_test_edge, neg_test_edge = get_pos_neg_edges('test', split_edge, data.edge_index, data.num_nodes)
    pos_train_pred = torch.ones(split_edge['train']['edge'].size(0))
    pos_valid_pred, _ = AA(A_eval, pos_val_edge)
    neg_valid_pred, _ = AA(A_eval, neg_val_edge)
    pos_test_pred, _ = AA(A, pos_test_edge)
    neg_test_pred, _ = AA(A, neg_test_edge)
    return _hits_table(args.dataset, evaluator, pos_train_pred, pos_valid_pred, neg_valid_pred, pos_test_pred,
                       neg_test_pred)


def test_resource_allocation(model, data, split_edge, evaluator, batch_size, args, device):
    """train_and_eval.py:218-270."""
    A_eval = get_A(data.adj_t, data.num_nodes)
    A = get_A(data.full_adj_t, data.num_nodes)
    batch_size = 1024
    pos_valid_pred = resource_allocation(A_eval, split_edge['valid']['edge'], batch_size=batch_size)
    neg_valid_pred = resource_allocation(A_eval, split_edge['valid']['edge_neg'], batch_size=batch_size)
    pos_test_pred = resource_allocation(A, split_edge['test']['edge'])
    neg_test_pred = resource_allocation(A, split_edge['test']['edge_neg'])
    pos_train_pred = torch.ones(split_edge['train']['edge'].size(0))
    return _hits_table(args.dataset, evaluator, pos_train_pred, pos_valid_pred, neg_valid_pred, pos_test_pred,
                       neg_test_pred)
