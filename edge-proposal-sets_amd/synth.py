"""Seeded synthetic graphs shaped like the BASELINE configs (S1..S5).  No dataset or
network is available on either box, so every benchmark / full-size test input is generated here,
on the device, from a seed."""
from __future__ import annotations

import torch

from .graph import CSRGraph


def rmat_edges(scale: int, n_edges: int, seed: int, device, a=0.57, b=0.19, c=0.19) -> torch.Tensor:
    """Graph500-style R-MAT edge list [2, n_edges] (int64) over 2**scale nodes."""
    gen = torch.Generator(device=device).manual_seed(seed)
    row = torch.zeros(n_edges, dtype=torch.int64, device=device)
    col = torch.zeros(n_edges, dtype=torch.int64, device=device)
    for _ in range(scale):
        p = torch.rand(n_edges, generator=gen, device=device)
        right = ((p >= a) & (p < a + b)) | (p >= a + b + c)
        down = p >= a + b
        row = row * 2 + down.to(torch.int64)
        col = col * 2 + right.to(torch.int64)
    return torch.stack([row, col])


def rmat_graph(scale: int, edge_factor: int, seed: int, device, n_nodes: int | None = None,
               permute: bool = True, **abc) -> CSRGraph:
    """Symmetric, deduplicated, loop-free unit-weight R-MAT graph.  ``n_nodes`` (<= 2**scale) folds ids
    with a modulo so that non-power-of-two node counts (ppa: 576,289) keep the degree skew."""
    n = n_nodes or (1 << scale)
    ei = rmat_edges(scale, edge_factor * (1 << scale), seed, device, **abc)
    if permute:  # break the id <-> degree correlation of raw R-MAT
        gen = torch.Generator(device=device).manual_seed(seed + 1)
        perm = torch.randperm(1 << scale, generator=gen, device=device)
        ei = perm[ei]
    if n != (1 << scale):
        ei = ei % n
    ei = ei[:, ei[0] != ei[1]]
    g = CSRGraph.from_edge_index(ei, None, sparse_sizes=(n, n)).to_symmetric()
    return g


def ppa_like(seed: int = 3, device="cuda", n_nodes: int = 576_289, n_undirected: int = 21_231_931) -> CSRGraph:
    """S3: ppa-sized graph (N=576,289; ~21.2 M undirected edges -> nnz ~42.5 M, avg degree ~74) with
    R-MAT skew softened (a=.45,b=.22,c=.22) so the maximum degree lands in the low thousands like ppa's
    (~3.2 k) instead of Graph500's tens of thousands."""
    scale = 20
    draw = int(n_undirected * 1.003)  # head-room for duplicates / loops removed by coalescing
    ei = rmat_edges(scale, draw, seed, device, a=0.45, b=0.22, c=0.22)
    gen = torch.Generator(device=device).manual_seed(seed + 1)
    perm = torch.randperm(1 << scale, generator=gen, device=device)
    ei = perm[ei] % n_nodes
    ei = ei[:, ei[0] != ei[1]]
    return CSRGraph.from_edge_index(ei, None, sparse_sizes=(n_nodes, n_nodes)).to_symmetric()
