"""The sharded filter step's collectives through RCCL itself, as far as a 1-GPU box allows: a job of ONE rank on the "nccl" backend.

RCCL refuses two ranks on one device, so the multi-rank tests (tests/test_gpu_multirank.py) use gloo as their transport and the first
real N > 1 run is the driver's.  What CAN be checked here is everything but the wire: that ``init_process_group("nccl", device_id=...)``
comes up the way bench.py and dist.init_from_env call it, and that every collective of the sharded step -- its dtypes, shapes, split
lists, in-place reductions -- is accepted and returns the right values: ``dist.FORCE_COLLECTIVES`` makes the helpers call the backend
in a one-rank job too, ``scan.FORCE_SHARDED`` sends scan_topk down the sharded control flow, and the rows must equal the one-rank
fast path's.  Runs in a child process (a process group is process-wide state)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, os.environ["EPS_ROOT"])
import torch
import torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)                       # bench.py / dist.init_from_env
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
seen = [None]
dist.all_gather_object(seen, {"rank": dist.get_rank(), "device": torch.cuda.get_device_name(dev)})       # bench.py's self-check
assert seen[0]["rank"] == 0
dist.barrier()
t = torch.tensor([12.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)                             # bench.py: the slowest rank's time
assert float(t) == 12.5

import eps_amd
from eps_amd import dist as epd, ops, scan, synth
from eps_amd.heuristics import node_weight_table
epd.FORCE_COLLECTIVES = True

# every helper with the dtypes the step sends
i64 = torch.arange(7, dtype=torch.int64, device=dev)
f32 = torch.linspace(0, 1, 5, device=dev)
i32 = torch.arange(11, dtype=torch.int32, device=dev)
assert torch.equal(epd.all_gather_list(i64)[0], i64) and torch.equal(epd.all_gather_list(f32)[0], f32)
assert torch.equal(epd.all_reduce_min_(f32.clone()), f32) and torch.equal(epd.all_reduce_sum_(i64.clone()), i64)
assert torch.equal(epd.all_reduce_sum_(i32.clone()), i32)
assert torch.equal(epd._gather_into(i32, 1), i32)
assert torch.equal(epd.gather_ragged(i64, [5]), i64[:5]) and torch.equal(epd.gather_ragged(f32, [3]), f32[:3])
assert torch.equal(epd.gather_ragged_to(i64, [7], 0), i64)
assert torch.equal(epd.all_to_all_ragged(i64, [6], [6]), i64[:6])
assert float(ops.kth_largest_dist(f32, 2, 1)) == 0.75

# (count what reaches the backend: the equalities below must not be reached on a path that skipped it)
calls = {}
def counted(name):
    fn = getattr(dist, name)
    def wrapper(*a, **kw):
        calls[name] = calls.get(name, 0) + 1
        return fn(*a, **kw)
    setattr(dist, name, wrapper)
for name in ("all_gather_into_tensor", "all_to_all_single", "gather", "all_reduce", "all_gather"):
    counted(name)

# the sharded step itself, one rank, over the process group
g = synth.rmat_graph(15, 16, 5, dev)
w = node_weight_table(g, ops.W_AA)
K = 30000
scan.SMALL_SET = 0                                                    # the estimated-bar path (a sample launch, a bar, the exchanges)
ref_p, ref_s = scan.scan_topk(g, w, K)
assert ref_p.shape == (2, K)
scan.FORCE_SHARDED = True
for hist in (True, False):
    for rows_min in (0, 1 << 15):
        for rows_on in (None, 0, "shards"):
            scan.DIST_HIST, scan.DIST_ROWS_MIN = hist, rows_min
            st = {}
            p, s = scan.scan_topk(g, w, K, 0, 1, stats=st, rows_on=rows_on)
            assert torch.equal(p, ref_p) and torch.equal(s, ref_s), (hist, rows_min, rows_on)
# no bar at all (small candidate sets): the r05 exchange of the scores
scan.SMALL_SET = 1 << 40
p, s = scan.scan_topk(g, w, K, 0, 1)
assert torch.equal(p, ref_p) and torch.equal(s, ref_s)
assert all(calls.get(n, 0) > 0 for n in ("all_gather_into_tensor", "all_to_all_single", "gather", "all_reduce")), calls
dist.barrier()
dist.destroy_process_group()
print("RCCL world-1 OK", calls)
'''


def test_sharded_step_over_rccl_in_a_one_rank_job(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               EPS_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL world-1 OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
