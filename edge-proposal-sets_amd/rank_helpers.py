"""Small graph helpers shared by the rank stage and the training loop."""
import torch


def to_undirected(edge_index: torch.Tensor) -> torch.Tensor:
    """torch_geometric.utils.to_undirected [third-party, restated]: both directions, coalesced (sorted, unique)."""
    both = torch.cat([edge_index, edge_index.flip(0)], 1)
    n = int(both.max()) + 1 if both.numel() else 1
    key = torch.unique(both[0] * n + both[1])
    return torch.stack([torch.div(key, n, rounding_mode="floor"), key % n])
