"""CPU: the C-ABI library builds, loads, and exports every symbol include/eps_abi.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "eps_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(eps_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_surface():
    syms = declared_symbols()
    for s in ["eps_pair_scores", "eps_pair_scores_f64", "eps_spmm_csr", "eps_mlp_decode", "eps_gemm_f32",
              "eps_col_sums", "eps_node_weights", "eps_gcn_norm", "eps_pack_keys", "eps_unpack_keys",
              "eps_last_error", "eps_version"]:
        assert s in syms


def test_library_exports_every_declared_symbol(eps):
    lib = eps.load()
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} declared in eps_abi.h but not exported"
    assert lib.eps_version() == eps._lib.ABI_VERSION == 7


def test_python_signatures_cover_header(eps):
    assert sorted(eps._lib.SIGNATURES) == declared_symbols()


def test_argument_validation_without_gpu(eps):
    """EINVAL paths return before any HIP call, so they are checkable on a CPU-only box."""
    lib = eps.load()
    rc = lib.eps_pair_scores(None, None, None, None, 10, None, None, 5, None, None, None, None)
    assert rc == -1 and b"null" in lib.eps_last_error()
    rc = lib.eps_mlp_decode(None, 0, 102, None, None, 0, None, None, 2, 1, None, None)
    assert rc == -1 and b"hdim" in lib.eps_last_error()
    rc = lib.eps_spmm_csr(None, None, None, -1, None, 0, 0, None, 0, 0, None, 0, None)
    assert rc == -1


def test_ops_refuse_cpu_tensors(eps):
    import torch
    g = eps.add_edges("ddi", torch.tensor([[0, 1], [1, 2]]), torch.ones(2), torch.zeros(2, 0, dtype=torch.long), 3)
    with pytest.raises(eps.EpsError):
        eps.ops.pair_scores(g.rowptr, g.col, None, None, 3, torch.zeros(1, dtype=torch.int32),
                            torch.zeros(1, dtype=torch.int32))
    if not torch.cuda.is_available():
        with pytest.raises(eps.EpsError):
            eps.AA(g, torch.tensor([[0], [2]]))


def test_missing_library_fails_loudly(eps, monkeypatch, tmp_path):
    """No HIP library -> the loader raises (there is no CPU fallback to slide onto)."""
    from eps_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libeps_hip.so"))
    with pytest.raises(_lib.EpsError, match="no CPU fallback"):
        _lib.load()


def test_product_package_never_imports_the_oracle():
    import glob
    for f in glob.glob(os.path.join(ROOT, "edge-proposal-sets_amd", "*.py")) + [os.path.join(ROOT, n) for n in ("filter.py", "rank.py", "eps_amd.py")]:
        src = open(f).read()
        assert "import oracle" not in src and "from oracle" not in src and "eps_oracle" not in src, f


def test_graft_entry_build():
    """The driver's "does it build" check: __graft_entry__.build() compiles the library, the oracle and the example host and
    verifies the ABI version the Python side expects (an incremental make here)."""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    entry = importlib.import_module("__graft_entry__")
    entry.build()
