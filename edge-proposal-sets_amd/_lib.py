"""ctypes binding of libeps_hip.so (include/eps_abi.h).  No fallback: if the HIP library is
missing or a tensor is not on a GPU, the call raises -- the product path never routes through
a CPU implementation."""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EPS_LIB_PATH") or os.path.join(_HERE, "libeps_hip.so")  # override: kernel A/B experiments
CSRC = os.path.join(_HERE, "csrc")

_c = ctypes
_vp, _i64, _i32, _int = _c.c_void_p, _c.c_int64, _c.c_int32, _c.c_int

# name -> (restype, argtypes); mirrors include/eps_abi.h one to one
SIGNATURES = {
    "eps_version": (_int, []),
    "eps_last_error": (_c.c_char_p, []),
    "eps_device_info": (_int, [_c.POINTER(_int), _c.c_char_p, _int]),
    "eps_warm_up": (_int, []),
    "eps_col_sums": (_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "eps_node_weights": (_int, [_vp, _i64, _int, _vp, _vp]),
    "eps_node_weights_f64": (_int, [_vp, _i64, _int, _vp, _vp]),
    "eps_pair_scores": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "eps_pair_scores_f64": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp]),
    "eps_pair_scores_grouped": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "eps_pair_scores_grouped_f64": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp]),
    "eps_expand_max_nodes": (_int, []),
    "eps_expand_count": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "eps_expand_fill": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "eps_expand_fill_tiled": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "eps_expand_workspace_bytes": (_i64, [_i64]),
    "eps_filter_scan_max_nodes": (_i64, []),
    "eps_filter_scan_workspace_bytes": (_i64, [_i64]),
    "eps_reverse_positions": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "eps_relabel_graph_workspace_bytes": (_i64, [_i64, _i64, _i32]),
    "eps_relabel_graph": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _i64, _vp]),
    "eps_reverse_positions_symmetric": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "eps_reverse_positions_sorted_workspace_bytes": (_i64, [_i64, _i64]),
    "eps_reverse_positions_sorted": (_int, [_vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "eps_node_order_workspace_bytes": (_i64, [_i64]),
    "eps_node_order": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "eps_score_bound": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "eps_fixed_weights": (_int, [_vp, _i64, _vp, _vp]),
    "eps_filter_scan_windows": (_int, [_i64, _c.POINTER(_i64), _c.POINTER(_i64)]),
    "eps_row_window_splits": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "eps_filter_scan": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _i64, _vp]),
    "eps_scan_windows": (_i32, []),
    "eps_rescore_runs": (_int, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp]),
    "eps_scan_cuts": (_int, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "eps_scan_screen_weights": (_int, [_vp, _i64, _i32, _vp, _vp, _vp]),
    "eps_scan_window_paths": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "eps_scan_window_paths_columns": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp]),
    "eps_scan_screen": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp]),
    "eps_scan_column_pack": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "eps_scan_row_records": (_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "eps_scan_heads": (_int, [_vp, _vp, _vp, _i64, _i32, _c.c_uint32, _i32, _vp, _vp]),
    "eps_scan_hub_rows": (_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "eps_scan_hub_row_words": (_i64, [_i64]),
    "eps_scan_refine": (_int, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "eps_scan_bounds": (_int, [_vp, _i64, _vp, _vp]),
    "eps_scan_row_sums": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "eps_scan_plan_rewalk": (_int, [_vp, _i64, _i32, _vp, _vp]),
    "eps_scan_plan": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "eps_scan_screen_weighted": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "eps_rescore_weighted": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp]),
    "eps_expand_unit_count": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _vp]),
    "eps_expand_unit_fill": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                    _i64, _vp]),
    "eps_expand_unit_list": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "eps_spmm_csr": (_int, [_vp, _vp, _vp, _i64, _vp, _i64, _i32, _vp, _int, _int, _vp, _i64, _vp]),
    "eps_gcn_norm": (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "eps_gemm_f32": (_int, [_vp, _i64, _vp, _i64, _vp, _int, _int, _vp, _i64, _i64, _i32, _i32, _vp]),
    "eps_dense_adjacency": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "eps_dense_candidates": (_int, [_vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "eps_dense_mirror_lower": (_int, [_vp, _i64, _i64, _vp]),
    "eps_mlp_decode": (_int, [_vp, _i64, _i32, _vp, _vp, _i64, _c.POINTER(_vp), _c.POINTER(_vp), _i32, _int, _vp, _vp]),
    "eps_kth_largest_workspace_bytes": (_i64, []),
    "eps_kth_largest_f32": (_int, [_vp, _i64, _i64, _vp, _vp, _vp]),
    "eps_select_topk_rows_relabelled": (_int, [_vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _vp]),
    "eps_select_topk_rows_pairs": (_int, [_vp, _vp, _i64, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _i64, _vp]),
    "eps_sort_pairs_by_u_workspace_bytes": (_i64, [_i64]),
    "eps_sort_pairs_by_u": (_int, [_vp, _i64, _i32, _i32, _vp, _vp, _i64, _vp]),
    "eps_compact_between": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "eps_select_compact_workspace_bytes": (_i64, []),
    "eps_select_compact": (_int, [_vp, _vp, _i64, _vp, _i64, _i32, _c.c_float, _c.c_float, _c.c_float, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "eps_kth_begin": (_int, [_vp, _i64, _vp]),
    "eps_kth_hist_f32": (_int, [_vp, _i64, _vp, _i32, _vp]),
    "eps_kth_pick": (_int, [_vp, _i32, _vp, _vp]),
    "eps_compact_at_least": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "eps_select_topk_cut_workspace_bytes": (_i64, []),
    "eps_compact_survivors": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp]),
    "eps_select_topk_cut": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp]),
    "eps_select_topk_rows_workspace_bytes": (_i64, [_i64]),
    "eps_select_topk_rows": (_int, [_vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _i64, _vp]),
    "eps_tail_state_bytes": (_i64, []),
    "eps_score_bins": (_i32, []),
    "eps_score_hist_into": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "eps_score_deal_plan": (_int, [_vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "eps_score_hist": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "eps_score_pick_compact": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _i64, _i32, _c.c_float, _c.c_float, _c.c_float, _i32, _vp, _vp, _vp,
                                      _vp, _i64, _vp, _vp, _vp]),
    "eps_radix_sort_workspace_bytes": (_i64, [_i64]),
    "eps_radix_sort_by_u": (_int, [_vp, _i64, _vp, _i32, _i32, _vp, _vp, _i64, _vp, _vp]),
    "eps_radix_sort_rows": (_int, [_vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp]),
    "eps_rescore_runs_dev": (_int, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    "eps_pack_keys": (_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "eps_unpack_keys": (_int, [_vp, _i64, _vp, _vp, _vp]),
}

ABI_VERSION = 7        # include/eps_abi.h EPS_ABI_VERSION
_lib = None
_load_lock = threading.RLock()


class EpsError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 into libeps_hip.so (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "-s", "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j8"])
    if not os.path.exists(LIB_PATH):
        raise EpsError(f"build did not produce {LIB_PATH}")
    return LIB_PATH


def load() -> ctypes.CDLL:
    """Load libeps_hip.so.  torch is imported first so that the HIP runtime torch ships
    (same soname, libamdhip64.so.7) is the one the library binds to -- device pointers are
    only meaningful inside one runtime."""
    global _lib
    if _lib is not None:
        return _lib
    with _load_lock:             # (the warm-up thread and the main thread may both arrive here first: one dlopen, one signature pass)
        return _load_locked()


def _load_locked() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (loads torch's libamdhip64 before ours resolves it)
    if not os.path.exists(LIB_PATH):
        raise EpsError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       f"or `make -C {CSRC}`.  There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == ABI drift, fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.eps_version() != ABI_VERSION:
        raise EpsError(f"libeps_hip.so ABI version {lib.eps_version()} != {ABI_VERSION}: rebuild (make -C {CSRC})")
    _lib = lib
    return lib


_warm = None


def warm_up_async(device=None):
    """Start loading the library's code objects (eps_warm_up) -- and the handful of torch operators the filter step uses -- in a
    daemon thread, and return at once.  A fresh process pays 10-30 ms per larger code object at the first launch of one of its
    kernels; filter.py / rank.py are one process per graph (submit_job.py:20-21) and spend their first hundreds of
    milliseconds reading the dataset on the host: the loads ride along.  Idempotent; errors are left for the first real call
    to report."""
    global _warm
    if _warm is not None:
        return _warm
    import atexit
    import torch

    def work():
        try:
            if not torch.cuda.is_available():
                return
            dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
            # (a stream of its own: nothing the main thread launches on the default stream queues behind the warm-up, and the
            #  thread waits for ITS stream only -- never for the device)
            side = torch.cuda.Stream(dev)
            with torch.cuda.device(dev), torch.cuda.stream(side):
                load().eps_warm_up()
                # (torch's own code objects: the few element-wise / copy operators between the library calls of a step)
                z = torch.zeros(4, dtype=torch.int64, device=dev)
                f = torch.full((4,), float("-inf"), dtype=torch.float32, device=dev)
                torch.cat([z, z[1:2] + 1]).tolist()
                torch.stack([z & 3, z >> 1]); f.view(torch.int32).to(torch.int64); torch.empty(4, device=dev)[:2].contiguous()
                torch.tensor([1, 2], dtype=torch.int64, device=dev)
                side.synchronize()
        except Exception:       # noqa: BLE001  (a warm-up must never be the thing that fails a run)
            pass

    _warm = threading.Thread(target=work, name="eps-warm-up", daemon=True)
    _warm.start()
    atexit.register(warm_up_join)        # (the interpreter must not tear down while the thread is inside the HIP runtime)
    return _warm


def warm_up_join(timeout: float = 30.0) -> None:
    """Wait for the warm-up thread (no-op without one): before a timed section, and at interpreter exit."""
    t = _warm
    if t is not None and t.is_alive():
        t.join(timeout)


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().eps_last_error().decode("utf-8", "replace")
        raise EpsError(f"{what} failed (rc={rc}): {msg}")
