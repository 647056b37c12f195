"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI).

The reference has no distributed code at all; this is new design, shaped by the path:

* candidate pairs are independent -> the pair list (or the candidate COLUMN range) is cut into
  contiguous shards balanced by a work estimate; the CSR graph and the weight table are
  replicated (ppa 0.35 GB, RMAT-24 2.2 GB << 288 GB); CN / AA / RA need no data-path collective.
* GNN filters: every layer's output rows are partitioned over the ranks and exchanged with ONE
  all-gather per layer (N x H x 4 B; ppa at H=256: 590 MB total, 74 MB per rank) -- the only
  bulk exchange on the path.  xGMI is a full mesh of point-to-point links, so the gather is issued
  as a single large collective, not bucketed.
* the final proposal set is a top-K merge of per-rank sorted key lists (K x 8 B per rank); keys carry
  the global candidate index, so the merged result does not depend on the number of shards.

Scoring callables are injected (``score_fn``), so the sharding / merge logic is testable on CPU with
the gloo backend while the product path passes the HIP scorers.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


# Tests on a 1-GPU box: run every helper's collective even in a job of ONE rank (they normally return early), so that the calls --
# dtypes, shapes, split lists -- go through RCCL itself (`init_process_group("nccl", world_size=1)`); scan.FORCE_SHARDED sends the
# sharded filter step through them.  tests/test_gpu_rccl_world1.py
FORCE_COLLECTIVES = False


def init_from_env(backend: Optional[str] = None, device_index: Optional[int] = None,
                  host_only: bool = False) -> Tuple[int, int, torch.device]:
    """Join the job described by RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).  -> (rank, world, device).
    ``backend``: "nccl" (RCCL over xGMI; the default on a GPU box) or "gloo" (collectives staged through host memory).
    ``device_index``: the HIP device of this rank; default LOCAL_RANK (whatever the backend: ``torchrun --nproc-per-node N
    filter.py --dist_backend gloo`` runs rank r on cuda:r).  RCCL needs one device per rank; with gloo several ranks may share
    a device (``filter.py --dist_backend gloo --device 0``: how a 1-GPU box runs the multi-rank path).
    ``host_only``: a job of CPU tensors also on a box with GPUs (the gloo tests of the sharding / merge logic)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available() and not host_only:
        n_dev = torch.cuda.device_count()
        index = local if device_index is None else int(device_index)
        if not 0 <= index < n_dev:
            raise RuntimeError(f"init_from_env: rank {rank} wants HIP device {index}, the box has {n_dev} "
                               "(pass --device to share one, or start fewer ranks per node)")
        device = torch.device("cuda", index)
        torch.cuda.set_device(device)
    else:
        device = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if device.type == "cuda" else "gloo")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, device


def all_gather_list(t: torch.Tensor) -> List[torch.Tensor]:
    """Every rank's copy of an equally shaped tensor, in rank order.  RCCL moves device tensors directly; gloo (CPU
    jobs and the one-device test hook) gathers through host memory."""
    rank, world = world_info()
    if world == 1 and not FORCE_COLLECTIVES:
        return [t]
    if dist.get_backend() == "gloo" and t.device.type != "cpu":
        parts = [torch.empty_like(t, device="cpu") for _ in range(world)]
        dist.all_gather(parts, t.cpu())
        return [p.to(t.device) for p in parts]
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t.contiguous())
    return parts


def all_reduce_min_(t: torch.Tensor) -> torch.Tensor:
    """In-place minimum over the ranks of a small device tensor (the ranks' bar estimates); transport as ``all_reduce_sum_``."""
    rank, world = world_info()
    if world == 1 and not FORCE_COLLECTIVES:
        return t
    if dist.get_backend() == "gloo" and t.device.type != "cpu":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MIN)
        t.copy_(h)
        return t
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return t


def all_reduce_sum_(t: torch.Tensor) -> torch.Tensor:
    """In-place sum over the ranks of a small device tensor (histograms, counters).  RCCL reduces device memory directly and
    stays on the stream; gloo (the one-device test hook) goes through host memory."""
    rank, world = world_info()
    if world == 1 and not FORCE_COLLECTIVES:
        return t
    if dist.get_backend() == "gloo" and t.device.type != "cpu":
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
        return t
    dist.all_reduce(t)
    return t


def gather_ragged(t: torch.Tensor, lens: Sequence[int]) -> torch.Tensor:
    """All ranks' 1-D tensors concatenated in rank order when every rank already KNOWS all lengths (they travelled with an
    earlier exchange): one padded all-gather, no length round trip, no host synchronisation."""
    rank, world = world_info()
    if world == 1 and not FORCE_COLLECTIVES:
        return t[:lens[0]]
    mx = max(max(lens), 1)
    pad = torch.zeros(mx, dtype=t.dtype, device=t.device)
    pad[:lens[rank]] = t[:lens[rank]]
    out = _gather_into(pad, world).view(world, mx)
    return torch.cat([out[r, :lens[r]] for r in range(world)])


def gather_ragged_to(t: torch.Tensor, lens: Sequence[int], dst: int) -> Optional[torch.Tensor]:
    """``gather_ragged`` onto ONE rank: ``dst`` gets the concatenation in rank order, every other rank None (the proposal file is
    written by one rank -- filter.py:160-165 -- so the K rows need not travel to all of them: at N = 8 an all-gather of the 4 M
    rows puts 48 MB on every rank's links, a gather 42 MB on one rank's seven)."""
    rank, world = world_info()
    if world == 1 and not FORCE_COLLECTIVES:
        return t[:lens[0]]
    mx = max(max(lens), 1)
    pad = torch.zeros(mx, dtype=t.dtype, device=t.device)
    pad[:lens[rank]] = t[:lens[rank]]
    via_host = dist.get_backend() == "gloo" and t.device.type != "cpu"
    src = pad.cpu() if via_host else pad
    parts = [torch.empty_like(src) for _ in range(world)] if rank == dst else None
    dist.gather(src, parts, dst=dst)
    if rank != dst:
        return None
    out = torch.cat([parts[r][:lens[r]] for r in range(world)])
    return out.to(t.device) if via_host else out


def all_to_all_ragged(t: torch.Tensor, send_counts: Sequence[int], recv_counts: Sequence[int]) -> torch.Tensor:
    """One all-to-all of a 1-D tensor cut into per-destination pieces: this rank sends ``t[sum(send_counts[:q]) : ...]`` to rank q
    and receives ``recv_counts[r]`` elements from rank r, concatenated in rank order.  Every rank knows both count lists already
    (no length round trip).  RCCL moves device memory directly; gloo (the one-device test hook) goes through host memory."""
    rank, world = world_info()
    if world == 1 and not FORCE_COLLECTIVES:
        return t[:send_counts[0]]
    via_host = dist.get_backend() == "gloo" and t.device.type != "cpu"
    src = (t.cpu() if via_host else t).contiguous()
    out = torch.empty(int(sum(recv_counts)), dtype=src.dtype, device=src.device)
    dist.all_to_all_single(out, src[:int(sum(send_counts))], [int(x) for x in recv_counts], [int(x) for x in send_counts])
    return out.to(t.device) if via_host else out


def world_info() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def balanced_bounds(work: torch.Tensor, world: int) -> List[int]:
    """Cut [0, len(work)) into ``world`` contiguous shards of (nearly) equal total work.
    -> world+1 boundaries.  Deterministic; every rank computes the same cut."""
    n = work.numel()
    if n == 0:
        return [0] * (world + 1)
    cum = torch.cumsum(work.to(torch.float64), 0)
    total = float(cum[-1])
    targets = torch.tensor([total * r / world for r in range(1, world)], dtype=torch.float64, device=work.device)
    cuts = torch.searchsorted(cum, targets, right=False).cpu().tolist() if world > 1 else []
    bounds = [0] + [min(int(c) + 1, n) for c in cuts] + [n]
    for i in range(1, len(bounds)):  # monotone
        bounds[i] = max(bounds[i], bounds[i - 1])
    return bounds


def pair_work(deg: torch.Tensor, u: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """Work estimate of scoring (u,v): d_u + d_v adjacency entries."""
    return (deg[u.long()] + deg[v.long()] + 16).to(torch.float32)


def shard_pairs(deg: torch.Tensor, u: torch.Tensor, v: torch.Tensor, rank: int, world: int) -> Tuple[int, int]:
    b = balanced_bounds(pair_work(deg, u, v), world)
    return b[rank], b[rank + 1]


def column_work(rowptr: torch.Tensor, col: torch.Tensor) -> torch.Tensor:
    """2-hop paths per column v: sum_{w in N(v)} deg(w) -- the cost of generating + scoring column v."""
    deg = rowptr[1:] - rowptr[:-1]
    n = deg.numel()
    rows = torch.repeat_interleave(torch.arange(n, device=rowptr.device), deg)
    out = torch.zeros(n, dtype=torch.float64, device=rowptr.device)
    out.index_add_(0, rows, deg[col.long()].to(torch.float64))
    return out


# ------------------------------------------------------------------ collectives
def _gather_into(t: torch.Tensor, world: int) -> torch.Tensor:
    """all_gather_into_tensor along dim 0; device tensors go through host memory when the transport is gloo."""
    via_host = dist.get_backend() == "gloo" and t.device.type != "cpu"
    src = t.cpu() if via_host else t
    out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src)
    return out.to(t.device) if via_host else out


def all_gather_rows(local: torch.Tensor, bounds: Sequence[int]) -> torch.Tensor:
    """All-gather a row-partitioned [N,F] matrix: rank r holds rows [bounds[r], bounds[r+1]).  One collective
    (ragged shards are padded to the largest)."""
    rank, world = world_info()
    if world == 1 and not FORCE_COLLECTIVES:
        return local
    sizes = [bounds[r + 1] - bounds[r] for r in range(world)]
    mx = max(sizes)
    f = local.shape[1]
    pad = local
    if local.shape[0] < mx:
        pad = torch.zeros((mx, f), dtype=local.dtype, device=local.device)
        pad[:local.shape[0]] = local
    out = _gather_into(pad.contiguous(), world)
    if all(s == mx for s in sizes):
        return out
    return torch.cat([out[r * mx: r * mx + sizes[r]] for r in range(world)], 0)


def all_gather_keys(keys: torch.Tensor, k: int) -> List[torch.Tensor]:
    """Gather every rank's (<= k) sorted int64 keys.  The true lengths travel with the lists (every int64 bit pattern
    is a legal key, so no value can serve as a padding sentinel)."""
    rank, world = world_info()
    if world == 1 and not FORCE_COLLECTIVES:
        return [keys]
    n = min(int(keys.numel()), k)
    pad = torch.zeros((k + 1,), dtype=torch.int64, device=keys.device)
    pad[0] = n
    pad[1:n + 1] = keys[:n]
    out = _gather_into(pad, world).view(world, k + 1)
    lens = out[:, 0].cpu().tolist()
    return [out[r, 1:1 + int(lens[r])] for r in range(world)]


def merge_topk(local_keys: torch.Tensor, k: int) -> torch.Tensor:
    """Global top-k keys (descending), identical on every rank and independent of the sharding."""
    allk = torch.cat(all_gather_keys(local_keys, k))
    kk = min(k, allk.numel())
    return torch.topk(allk, kk, largest=True, sorted=True).values


# ------------------------------------------------------------------ sharded pipelines
def score_pairs_sharded(deg: torch.Tensor, u: torch.Tensor, v: torch.Tensor,
                        score_fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor],
                        pack_fn: Callable[[torch.Tensor, int], torch.Tensor], k: int):
    """Each rank scores its work-balanced contiguous slice of the (replicated) pair list, keeps its local top-k
    as keys tagged with GLOBAL pair indices, and the lists are merged.  -> (merged keys, (lo, hi), local scores)."""
    rank, world = world_info()
    lo, hi = shard_pairs(deg, u, v, rank, world)
    scores = score_fn(u[lo:hi].contiguous(), v[lo:hi].contiguous())
    keys = pack_fn(scores, lo)
    kk = min(k, keys.numel())
    local = torch.topk(keys, kk, largest=True, sorted=True).values if kk else keys[:0]
    return merge_topk(local, k), (lo, hi), scores


def sharded_gnn_forward(layers: Sequence[Callable[[torch.Tensor, int, int], torch.Tensor]], x_full: torch.Tensor,
                        n_rows: int) -> torch.Tensor:
    """Row-sharded GNN forward: ``layers[l](x_full, lo, hi)`` returns rows [lo,hi) of layer l's output given the
    FULL input; after every layer the row blocks are exchanged with one all-gather.  Returns the full final
    embeddings on every rank (what the decode kernel gathers from)."""
    rank, world = world_info()
    bounds = [n_rows * r // world for r in range(world + 1)]
    x = x_full
    for layer in layers:
        local = layer(x, bounds[rank], bounds[rank + 1])
        x = all_gather_rows(local, bounds)
    return x
