"""edge-proposal-sets_amd -- MI355X-native Filter-and-Rank edge-scoring engine.

Drop-in for the hot path of CUAI/Edge-Proposal-Sets (filter.py / rank.py scoring loops,
adamic_utils.AA, train_and_eval.resource_allocation, models.GCN / SAGE / LinkPredictor /
LinkGNN / CommonNeighborsPredictor): Python host code on PyTorch-ROCm calling hand-written HIP
(csrc/*.hip -> libeps_hip.so) through the C ABI in include/eps_abi.h.

The directory name carries a hyphen, so import it through the ``eps_amd`` shim at the repo
root (``import eps_amd``), which registers this package under that name.
"""
from . import _lib  # noqa: F401
from ._lib import EpsError, build, load  # noqa: F401
from .graph import CSRGraph, add_edges  # noqa: F401
from . import ops  # noqa: F401
from .heuristics import AA, common_neighbors, get_A, resource_allocation  # noqa: F401

__all__ = ["EpsError", "build", "load", "CSRGraph", "add_edges", "ops", "AA", "resource_allocation", "get_A",
           "common_neighbors"]
