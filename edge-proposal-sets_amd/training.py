"""Training loop of the rank stage (train_and_eval.py:31-96), for the GCN / SAGE + LinkPredictor models.

Outside the scored hot path but needed to produce the checkpoints filter.py consumes and to
run rank.py with a parametrised model: torch autograd drives it, the GNN aggregate is the HIP SpMM (custom autograd
Function in models.py), dense layers and the pairwise decode run on torch in training mode.
"""
from __future__ import annotations

import torch

from .rank_helpers import to_undirected


def negative_sampling(edge_index: torch.Tensor, num_nodes: int, num_neg_samples: int) -> torch.Tensor:
    """torch_geometric.utils.negative_sampling [third-party, restated]: ``num_neg_samples`` node pairs drawn uniformly
    that are NOT edges of ``edge_index`` (rejection against the sorted edge keys)."""
    dev = edge_index.device
    keys = torch.unique(edge_index[0] * num_nodes + edge_index[1])
    out = []
    have = 0
    while have < num_neg_samples:
        n = int((num_neg_samples - have) * 1.2) + 16
        cand = torch.randint(0, num_nodes * num_nodes, (n,), device=dev)
        pos = torch.searchsorted(keys, cand).clamp(max=keys.numel() - 1)
        cand = cand[keys[pos] != cand] if keys.numel() else cand
        out.append(cand)
        have += cand.numel()
    k = torch.cat(out)[:num_neg_samples]
    return torch.stack([torch.div(k, num_nodes, rounding_mode="floor"), k % num_nodes])


def train(model, data, dataset_name, split_edge, optimizer, batch_size, use_params, model_str, device):
    """train_and_eval.py:31-96: one epoch over the training edges; -log(pos) - log(1-neg) on a batch of positive
    edges (both directions) and as many negatives; gradient clipping at 1.0; Adam step."""
    model.train()
    pos_train_edge = split_edge['train']['edge'].to(device)
    row, col, _ = data.adj_t.coo()
    edge_index = torch.stack([col, row], dim=0)
    total_loss = total_examples = 0
    perm_all = torch.randperm(pos_train_edge.size(0), device=device)        # DataLoader(..., shuffle=True)
    for s in range(0, perm_all.numel(), batch_size):
        perm = perm_all[s:s + batch_size]
        if use_params:
            optimizer.zero_grad()
        pos_edge = to_undirected(pos_train_edge[perm].t())
        if model_str in ['gcn', 'sage']:
            if dataset_name in ["collab"]:
                neg_edge = torch.randint(0, data.num_nodes, pos_edge.size(), dtype=torch.long, device=pos_edge.device)
            else:
                neg_edge = negative_sampling(edge_index, data.num_nodes, pos_edge.size(1))
        else:
            neg_dst = torch.randint(0, data.num_nodes, (pos_edge.size(1),), dtype=torch.long, device=pos_edge.device)
            neg_edge = torch.stack([pos_edge[0], neg_dst])
        out = model(data.x, torch.cat([pos_edge, neg_edge], 1), data.adj_t).squeeze()
        pos_out = out[:pos_edge.size(1)]
        neg_out = out[pos_edge.size(1):]
        loss = -torch.log(pos_out + 1e-8).mean() - torch.log(1 - neg_out + 1e-8).mean()
        if use_params:
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            optimizer.step()
        num_examples = pos_out.size(0)
        total_loss += loss.item() * num_examples
        total_examples += num_examples
    return total_loss / max(total_examples, 1)
