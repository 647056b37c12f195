"""GCN / SAGE / LinkPredictor / LinkGNN / CommonNeighborsPredictor with the reference's class
names, constructor signatures, ``forward`` signatures and state-dict keys (models.py:163-187,
:417-440, :461-485, :487-506, :508-575, :578-670, :673-790) -- the forward passes run on the
hand-written HIP kernels (csrc/) instead of torch_geometric / torch_sparse / cuBLAS.

Scoring signature kept: ``model(x, edges[2,B], adj_t) -> scores`` ([B,1] for LinkGNN, [B] for
CommonNeighborsPredictor('simple')).

Scope: the scoring loops (filter.py:113-121, train_and_eval.py:108-136: ``model.eval()`` / ``torch.no_grad()``)
run entirely on the HIP kernels.  Training (train_and_eval.py:31-96) runs on torch autograd
with the HIP SpMM as a custom autograd Function (forward and backward) and the dense layers on torch.

What differs from the reference on purpose: ``LinkGNN`` computes the node embeddings ``h`` ONCE per
(parameters, x, adjacency) and reuses them for every scoring batch; the reference re-runs the whole
GNN for each batch (models.py:505 called from filter.py:118).  Results are identical in eval mode.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F

from . import ops
from .graph import CSRGraph
from . import heuristics


class _SpMM(torch.autograd.Function):
    """Differentiable CSR x dense aggregate for TRAINING (train_and_eval.py:31-96): forward and backward both run
    eps_spmm_csr.  The adjacency must be structurally symmetric with symmetric values (every adjacency the reference
    builds is: rank.py:33), so the transposed product of the backward pass is a product with the same matrix:
      sum : Y = A X          ->  dX = A dY
      mean: Y = D^-1 A X     ->  dX = A (D^-1 dY)        (unit values)"""

    @staticmethod
    def forward(ctx, x, graph, mean):
        ctx.graph, ctx.mean = graph, mean
        return ops.spmm_csr(graph.rowptr, graph.col, None if mean else graph.val, x.contiguous(), mean=mean)

    @staticmethod
    def backward(ctx, grad_out):
        g = ctx.graph
        grad_out = grad_out.contiguous()
        if ctx.mean:
            inv = 1.0 / g.degree().clamp(min=1).to(torch.float32)
            grad_out = (grad_out * inv[:, None]).contiguous()
        return ops.spmm_csr(g.rowptr, g.col, None if ctx.mean else g.val, grad_out), None, None


def _pad4_full(x: torch.Tensor):
    """-> (view [N,K], full [N,ld]) with ld = K rounded up to a multiple of 4 floats and zero pad columns:
    16-B aligned rows let the GEMM / SpMM kernels use 16-byte loads; K itself is unchanged."""
    n, k = x.shape
    ld = (k + 3) // 4 * 4
    if ld == k and x.is_contiguous() and x.data_ptr() % 16 == 0:
        return x, x
    buf = torch.zeros((n, ld), dtype=torch.float32, device=x.device)
    buf[:, :k] = x
    return buf[:, :k], buf


def _pad4(x: torch.Tensor) -> torch.Tensor:
    if x.dim() == 2 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0:
        return x
    return _pad4_full(x)[0]


# ----------------------------------------------------------------------------------- convs
class GCNConv(torch.nn.Module):
    """torch_geometric 1.7.0 GCNConv [third-party, restated]: out = D^-1/2 (A with diag := 1) D^-1/2 (x W) + b.
    Parameters as in PyG 1.7: ``weight`` [in,out] (glorot), ``bias`` [out] (zeros).  Checkpoints written by
    PyG >= 2.0 (``lin.weight`` [out,in]) load too."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = torch.nn.Parameter(torch.empty(in_channels, out_channels))
        self.bias = torch.nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        a = math.sqrt(6.0 / (self.in_channels + self.out_channels))  # glorot
        with torch.no_grad():
            self.weight.uniform_(-a, a)
            self.bias.zero_()

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        k_new = prefix + "lin.weight"
        if k_new in state_dict and prefix + "weight" not in state_dict:
            state_dict[prefix + "weight"] = state_dict.pop(k_new).t()
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, x: torch.Tensor, adj_t: CSRGraph, relu: bool = False) -> torch.Tensor:
        gn = adj_t.gcn_normalized()                      # cached per adjacency (eps_gcn_norm)
        if torch.is_grad_enabled() and self.training:    # training: HIP SpMM inside autograd, dense part on torch
            out = _SpMM.apply(x @ self.weight, gn, False) + self.bias
            return F.relu(out) if relu else out
        with torch.no_grad():
            return self._forward_hip(x, gn, relu)

    def _forward_hip(self, x, gn, relu, rows=None):
        w_nk = self.weight.detach().t().contiguous()     # [out,in]: the GEMM takes Linear layout
        xw = ops.gemm(_pad4(x), w_nk)                    # transform ...
        rowptr = gn.rowptr if rows is None else gn.rowptr[rows[0]:rows[1] + 1]   # row block: absolute offsets into col
        return ops.spmm_csr(rowptr, gn.col, gn.val, xw, bias=self.bias.detach(), relu=relu)   # ... then aggregate

    @torch.no_grad()
    def forward_rows(self, x: torch.Tensor, adj_t: CSRGraph, lo: int, hi: int, relu: bool = False) -> torch.Tensor:
        """Rows [lo, hi) of the layer output from the FULL input (multi-GPU row sharding, dist.py)."""
        return self._forward_hip(x, adj_t.gcn_normalized(), relu, rows=(lo, hi))

    def __repr__(self):
        return f"GCNConv({self.in_channels}, {self.out_channels})"


class SAGEConv(torch.nn.Module):
    """torch_geometric 1.7.0 SAGEConv [third-party; semantics witnessed in-tree by models.py:347-349,
    :358-384]: out = lin_l(mean_{j in N(i)} x_j) + lin_r(x_i); mean ignores edge values, no self loop;
    lin_r has no bias."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin_l = torch.nn.Linear(in_channels, out_channels, bias=True)
        self.lin_r = torch.nn.Linear(in_channels, out_channels, bias=False)

    def reset_parameters(self):
        self.lin_l.reset_parameters()
        self.lin_r.reset_parameters()

    def forward(self, x: torch.Tensor, adj_t: CSRGraph, relu: bool = False) -> torch.Tensor:
        if torch.is_grad_enabled() and self.training:
            out = self.lin_l(_SpMM.apply(x, adj_t, True)) + self.lin_r(x)
            return F.relu(out) if relu else out
        with torch.no_grad():
            return self._forward_hip(x, adj_t, relu)

    TRANSFORM_FIRST = True     # aggregate lin_l(x) instead of x when the layer narrows (see _forward_hip)

    def _forward_hip(self, x, adj_t, relu, rows=None):
        k = x.shape[1]
        x, x_full = _pad4_full(x)                      # pad columns are zero: aggregate the padded width (float4 path)
        rowptr = adj_t.rowptr if rows is None else adj_t.rowptr[rows[0]:rows[1] + 1]
        x_rows = x if rows is None else x[rows[0]:rows[1]]
        if self.TRANSFORM_FIRST and self.in_channels > self.out_channels and self.out_channels % 4 == 0:
            # the mean commutes with lin_l: mean_j(x_j) W^T == mean_j(x_j W^T).  A layer that narrows (ppa: 58 features +
            # 256-d embedding = 314 -> 256; collab: 384 -> 256) then gathers ONE 1-KiB row of z = x W_l^T per neighbour
            # instead of a 256-column pass plus a partial-row pass over x; lin_l's bias rides in the SpMM's epilogue.
            z = ops.gemm(x, self.lin_l.weight.detach())
            out = ops.spmm_csr(rowptr, adj_t.col, None, z, bias=self.lin_l.bias.detach(), mean=True)
        else:
            agg = ops.spmm_csr(rowptr, adj_t.col, None, x_full, mean=True)[:, :k]
            out = ops.gemm(agg, self.lin_l.weight.detach(), bias=self.lin_l.bias.detach())
        return ops.gemm(x_rows, self.lin_r.weight.detach(), out=out, accumulate=True, relu=relu)

    @torch.no_grad()
    def forward_rows(self, x: torch.Tensor, adj_t: CSRGraph, lo: int, hi: int, relu: bool = False) -> torch.Tensor:
        """Rows [lo, hi) of the layer output from the FULL input (multi-GPU row sharding, dist.py)."""
        return self._forward_hip(x, adj_t, relu, rows=(lo, hi))

    def __repr__(self):
        return f"SAGEConv({self.in_channels}, {self.out_channels})"


class _ConvStack(torch.nn.Module):
    conv_cls = None

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout):
        super().__init__()
        self.convs = torch.nn.ModuleList()
        self.convs.append(self.conv_cls(in_channels, hidden_channels))
        for _ in range(num_layers - 2):
            self.convs.append(self.conv_cls(hidden_channels, hidden_channels))
        self.convs.append(self.conv_cls(hidden_channels, out_channels))
        self.dropout = dropout

    def reset_parameters(self):
        for conv in self.convs:
            conv.reset_parameters()

    def forward(self, x, adj_t):
        # models.py:181-187 / :434-440: ReLU (+ dropout, identity in eval) after every layer but the last;
        # the ReLU rides in the producing kernel's epilogue.
        for conv in self.convs[:-1]:
            x = conv(x, adj_t, relu=True)
            x = F.dropout(x, p=self.dropout, training=self.training)   # identity in eval mode
        return self.convs[-1](x, adj_t)


    @torch.no_grad()
    def forward_sharded(self, x, adj_t, rank: int, world: int, gather):
        """Multi-GPU forward (inference): the last layer -- the one whose output every rank needs in full for the
        decode -- is computed for this rank's row block only and exchanged with ONE all-gather (``gather(local,
        bounds)``, dist.all_gather_rows: N x H x 4 B over xGMI); the layers before it are replicated (tens of ms at
        these sizes: cheaper than an all-gather per layer)."""
        for conv in self.convs[:-1]:
            x = conv(x, adj_t, relu=True)
        n = adj_t.n_rows
        bounds = [n * r // world for r in range(world + 1)]
        local = self.convs[-1].forward_rows(x, adj_t, bounds[rank], bounds[rank + 1])
        return gather(local, bounds)


class GCN(_ConvStack):
    """models.py:163-187."""
    conv_cls = GCNConv


class SAGE(_ConvStack):
    """models.py:417-440."""
    conv_cls = SAGEConv


REORDER_MIN_NODES = 100_000      # graphs from this size on run their GNN layers on the hubs-first relabelling


# ----------------------------------------------------------------------------------- decode
class LinkPredictor(torch.nn.Module):
    """models.py:461-485: Hadamard -> (L-1) x [Linear, ReLU, dropout] -> Linear(hidden, out) -> sigmoid."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout):
        super().__init__()
        self.lins = torch.nn.ModuleList()
        self.lins.append(torch.nn.Linear(in_channels, hidden_channels))
        for _ in range(num_layers - 2):
            self.lins.append(torch.nn.Linear(hidden_channels, hidden_channels))
        self.lins.append(torch.nn.Linear(hidden_channels, out_channels))
        self.dropout = dropout

    def reset_parameters(self):
        for lin in self.lins:
            lin.reset_parameters()

    @torch.no_grad()
    def decode(self, h: torch.Tensor, edges: torch.Tensor, apply_sigmoid: bool = True) -> torch.Tensor:
        """Fused gather + MLP + sigmoid over edges [2,B] (eps_mlp_decode) -> float32 [B].  Inference only."""
        heuristics.check_node_ids(edges, h.shape[0], "decode edges")   # the kernel gathers h[u], h[v] unchecked
        e = edges.to(device=h.device, dtype=torch.int32)
        ws = [lin.weight.detach().contiguous() for lin in self.lins]
        bs = [lin.bias.detach().contiguous() for lin in self.lins]
        return ops.mlp_decode(h.contiguous(), e[0].contiguous(), e[1].contiguous(), ws, bs, apply_sigmoid)

    def forward(self, x_i: torch.Tensor, x_j: torch.Tensor) -> torch.Tensor:
        """Reference signature: two gathered [B,H] blocks -> [B,1].  (LinkGNN uses decode(), which gathers
        inside the kernel instead of materialising the two blocks.)"""
        if torch.is_grad_enabled() and self.training:   # models.py:478-485 on autograd
            x = x_i * x_j
            for lin in self.lins[:-1]:
                x = F.dropout(F.relu(lin(x)), p=self.dropout, training=True)
            return torch.sigmoid(self.lins[-1](x))
        b = x_i.shape[0]
        h2 = torch.cat([x_i, x_j], 0)
        idx = torch.arange(b, device=x_i.device, dtype=torch.int32)
        return self.decode(h2, torch.stack([idx, idx + b])).unsqueeze(1)


class LinkGNN(torch.nn.Module):
    """models.py:487-506."""

    def __init__(self, emb, gnn, linkpred):
        super().__init__()
        self.gnn = gnn
        self.linkpred = linkpred
        self.emb = emb
        self._h_key = None
        self._h = None

    def reset_parameters(self):
        self.gnn.reset_parameters()
        self.linkpred.reset_parameters()
        if self.emb is not None:
            self.emb.reset_parameters()
        self._h_key = None

    @torch.no_grad()
    def embeddings(self, x: Optional[torch.Tensor], adj: CSRGraph) -> torch.Tensor:
        """h = gnn([emb.weight || x], adj), embedding FIRST (models.py:501-505); cached per
        (parameter versions, x, adjacency) -- the reference recomputes it for every scoring batch."""
        key = (adj.uid, None if x is None else (x.data_ptr(), x._version),
               tuple((p.data_ptr(), p._version) for p in self.parameters()))
        if key != self._h_key:
            if x is None:
                xin = self.emb.weight.detach()
            elif self.emb is not None:
                xin = torch.cat([self.emb.weight.detach(), x], dim=1)
            else:
                xin = x
            from . import dist as epd
            rank, world = epd.world_info()
            # large graphs run the layers on the hubs-first relabelling (graph.degree_ordered: the SpMM's gathers hit the
            # cache more often); the embeddings come back in the caller's node order
            relabel = adj.n_rows >= REORDER_MIN_NODES and adj.n_rows == adj.n_cols
            if relabel:
                adj_run, perm, inv = adj.degree_ordered()
                xin = xin[perm].contiguous()
            else:
                adj_run = adj
            if world > 1 and hasattr(self.gnn, "forward_sharded"):
                h = self.gnn.forward_sharded(xin, adj_run, rank, world, epd.all_gather_rows)
            else:
                h = self.gnn(xin, adj_run)
            self._h = h[inv].contiguous() if relabel else h
            self._h_key = key
        return self._h

    def forward(self, x, edges, adj):
        if torch.is_grad_enabled() and self.training:   # models.py:500-506 on autograd (no caching while training)
            if x is None:
                x = self.emb.weight
            elif self.emb is not None:
                x = torch.cat([self.emb.weight, x], dim=1)
            h = self.gnn(x, adj)
            return self.linkpred(h[edges[0]], h[edges[1]])
        h = self.embeddings(x, adj)
        return self.linkpred.decode(h, edges).unsqueeze(1)


class CommonNeighborsPredictor(torch.nn.Module):
    """models.py:508-575.  'simple' (CN) and 'adamic' run on eps_pair_scores; 'adamic_ogb',
    'resource_allocation', 'katz' return None exactly like the reference (those heuristics are evaluated by
    AA()/resource_allocation(), not by the module).  The cosine variants ('mlpcos', 'simplecos') are not
    part of the accelerated path."""

    def __init__(self, emb, in_channels, hidden_channels, out_channels, num_layers, dropout, model_type='weighted'):
        super().__init__()
        assert model_type in ['mlpcos', 'simplecos', 'adamic', 'simple', 'adamic_ogb', "resource_allocation", 'katz']
        self.type = model_type
        self.mlp = torch.nn.Identity()
        self.emb = emb

    def reset_parameters(self):
        if self.emb is not None:
            self.emb.reset_parameters()

    @torch.no_grad()
    def forward(self, x, edges, adj):
        if self.type in ['adamic_ogb', "resource_allocation", 'katz']:
            return None                                                      # models.py:534-535
        if self.type == 'simple':
            return heuristics.common_neighbors(adj, edges)                   # models.py:536-542
        if self.type == 'adamic':
            # models.py:547-554: weight 1/log(rowsum(adj) + 1e-6) per common neighbour (edge values of the two
            # rows are NOT used: only the indices of the product), no inf guard, then sigmoid
            key = ("adamic_model_w",)
            if key not in adj._cache:
                adj._cache[key] = (1.0 / torch.log(adj.sum(-1) + 1e-6)).contiguous()
            g = adj if adj.val is None else adj.fill_value(1.0)
            heuristics.check_node_ids(edges, g.n_rows)
            e = edges.to(device=adj.device, dtype=torch.int32)
            _, _, ws = ops.pair_scores(g.rowptr, g.col, None, adj._cache[key], g.n_rows, e[0].contiguous(),
                                       e[1].contiguous(), want_count=False, want_cn=False)
            return torch.sigmoid(ws)
        raise NotImplementedError(f"CommonNeighborsPredictor('{self.type}') is outside the accelerated path")


# ----------------------------------------------------------------------------------- factory
DECODE_MAX_HIDDEN = 256          # csrc/mlp_decode.hip: hdim % 4 == 0 && hdim <= 256
_MODELS = ['sage', 'sage2', 'gcn', 'dea', 'dea_512', 'mlpcos', 'simplecos', 'adamic', 'simple', 'adamic_ogb',
           "resource_allocation", 'katz', 'ensemble_gcn_sage']
_HEURISTICS = ['mlpcos', 'simplecos', 'adamic', 'simple', 'adamic_ogb', 'katz', "resource_allocation"]


def build_model(args, data, device):
    """models.py:578-670 for the models on the accelerated path (gcn, sage, heuristics)."""
    assert args.model in _MODELS
    emb = None
    if args.use_learnable_embedding:
        emb = torch.nn.Embedding(data.num_nodes, args.hidden_channels).to(device)
    input_dim = 0
    if args.use_learnable_embedding:
        input_dim += args.hidden_channels
    if args.use_feature:
        input_dim += data.x.shape[1]
    if args.model in ('sage', 'gcn'):
        hc = args.hidden_channels
        if hc is None or hc % 4 != 0 or hc > DECODE_MAX_HIDDEN:
            # eps_mlp_decode keeps a 64-edge tile of width H in LDS: fail here, before any training time is spent
            raise ValueError(f"--hidden_channels {hc}: the fused decode kernel takes a multiple of 4 up to "
                             f"{DECODE_MAX_HIDDEN} (the reference's ogbl defaults are 256)")
        gnn_cls = SAGE if args.model == 'sage' else GCN
        gnn = gnn_cls(input_dim, args.hidden_channels, args.hidden_channels, args.num_layers, args.dropout).to(device)
        linkpred = LinkPredictor(args.hidden_channels, args.hidden_channels, 1, args.num_layers,
                                 args.dropout).to(device)
        return LinkGNN(emb, gnn, linkpred)
    if args.model in _HEURISTICS:
        return CommonNeighborsPredictor(emb, input_dim, args.hidden_channels, args.hidden_channels, args.num_layers,
                                        args.dropout, model_type=args.model).to(device)
    raise NotImplementedError(f"model '{args.model}' (sage2 / dea / ensemble) is outside the accelerated path "
                              "(SURVEY 2.1: alternative / experimental models, not in the north star)")


# per-dataset defaults, one row per (dataset group, model group): restates the table of models.py:673-790
_GNNS = ('sage', 'sage2', 'gcn', 'dea', 'dea_512', 'ensemble_gcn_sage')
_KEYS = ["num_layers", "hidden_channels", "dropout", "batch_size", "lr", "epochs", "use_feature",
         "use_learnable_embedding"]


def _defaults_for(dataset: str, model: str) -> dict:
    d = dict.fromkeys(_KEYS)

    def put(**kw):
        d.update(kw)

    if dataset == 'ddi':
        put(use_feature=False, use_learnable_embedding=True, batch_size=64 * 1024)
        if model in _GNNS:
            put(num_layers=2, hidden_channels=256, dropout=0.5, lr=0.005, epochs=200)
            if model in ('dea', 'dea_512'):
                put(num_layers=3, epochs=400)
                if model == 'dea_512':
                    put(hidden_channels=512)
        if model in ('mlpcos', 'simplecos'):
            put(num_layers=2, hidden_channels=256, dropout=0.5, lr=0.005, epochs=200)
        if model in ('simple', 'simplecos'):
            put(batch_size=1024)
            if model == 'simplecos':
                put(use_feature=True)
    if dataset == 'collab':
        put(use_feature=True, use_learnable_embedding=True, batch_size=16 * 1024)
        if model in ('sage', 'sage2', 'gcn', 'dea', 'dea_512'):
            put(num_layers=3, hidden_channels=256, dropout=0.0, lr=0.001, epochs=200)
            if model in ('dea', 'dea_512'):
                put(num_layers=4, epochs=400)
                if model == 'dea_512':
                    put(hidden_channels=512)
        if model in ('mlpcos', 'simplecos'):
            put(num_layers=3, hidden_channels=256, dropout=0.0, lr=0.00001, epochs=400)
    if dataset in ('reddit', 'twitch', 'fb'):
        put(use_feature=True, use_learnable_embedding=True, batch_size=64 * 1024)
        if model in ('sage', 'sage2', 'gcn'):
            put(num_layers=3, hidden_channels=256, dropout=0.0, lr=0.005, epochs=200)
        if model in ('mlpcos', 'simplecos'):
            put(batch_size=1024, num_layers=3, hidden_channels=256, dropout=0.0, lr=0.001, epochs=10)
    if dataset == 'email':
        put(use_feature=False, use_learnable_embedding=True, batch_size=16 * 1024)
        if model in ('sage', 'sage2', 'gcn'):
            put(num_layers=3, hidden_channels=300, dropout=0.0, lr=0.001, epochs=200)
        if model in ('mlpcos', 'simplecos'):
            put(num_layers=3, hidden_channels=256, dropout=0.0, batch_size=1024, lr=0.00004, epochs=30)
    return d


def default_model_configs(args):
    """models.py:673-790: fill every model flag the command line left at None from the per-(dataset, model)
    table; a CLI value wins (:774-778); heuristic models force use_feature / use_learnable_embedding off
    (:783-785).  ppa has no block in the reference: its flags stay None unless given on the CLI."""
    defaults = _defaults_for(args.dataset, args.model)
    for attr in _KEYS:
        if getattr(args, attr) is None:
            setattr(args, attr, defaults[attr])
    if args.model in ['adamic', 'simple', 'adamic_ogb', "resource_allocation", 'katz', "ensemble_gcn_sage"]:
        args.use_feature = False
        args.use_learnable_embedding = False
    if args.model == 'simplecos' and args.dataset != "email":
        args.use_learnable_embedding = False
    return args
