"""The compiled host of examples/filter_step.cpp -- the filter step on the C ABI alone, no torch in the process -- against
scan.scan_topk on the same graph."""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("scale,k", [(9, 1500), (11, 40_000), (11, 10**9)])
def test_compiled_host_matches_python_host(eps, dev, tmp_path, scale, k):
    from eps_amd import scan, synth
    from eps_amd.heuristics import node_weight_table
    exe = os.path.join(ROOT, "examples", "filter_step")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
    g = synth.rmat_graph(scale, 8, 4, dev)
    rp, col = g.rowptr.cpu().numpy().astype(np.int64), g.col.cpu().numpy().astype(np.int32)
    gpath, opath = str(tmp_path / "graph.bin"), str(tmp_path / "out.bin")
    with open(gpath, "wb") as f:
        f.write(struct.pack("<qq", g.n_rows, len(col)))
        f.write(rp.tobytes())
        f.write(col.tobytes())
    env = dict(os.environ)
    out = subprocess.run([exe, gpath, str(k), opath], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    raw = open(opath, "rb").read()
    rows = struct.unpack("<q", raw[:8])[0]
    rec = np.frombuffer(raw[8:], dtype=np.dtype([("key", "<i8"), ("val", "<f4")]))
    assert len(rec) == rows
    pairs, scores = scan.scan_topk(g, node_weight_table(g, eps.ops.W_AA), k)
    assert rows == pairs.shape[1]
    want_key = (pairs[1] << 32 | pairs[0]).cpu().numpy()
    assert np.array_equal(rec["key"], want_key) and np.array_equal(rec["val"], scores.cpu().numpy())
