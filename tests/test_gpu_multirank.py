"""Two ranks on ONE GPU (``--dist_backend gloo --device 0``: gloo as the transport, both ranks on cuda:0): the multi-rank
control flow of the filter stage -- column sharding by work, the sharded threshold scan with its survivor all-gather
(unit-valued graphs), per-rank streaming top-K + rank-ordered merge (weighted graphs), row-sharded last GNN layer +
all-gather -- on the real kernels, against the single-process result; and bench.py's N > 1 paths as child processes."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, workdir, argv):
    sys.path.insert(0, ROOT)
    os.chdir(workdir)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import eps_amd  # noqa: F401
    from eps_amd import candidates, filter_stage
    candidates.DEFAULT_BLOCK_PATHS = 30_000          # several blocks per rank: the in-kernel cut runs too
    filter_stage.main(argv + ["--dist_backend", "gloo", "--device", "0"])
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("model,dataset", [("adamic_ogb", "collab"), ("gcn", "collab"), ("adamic_ogb", "ddi"),
                                           ("simple", "ddi"), ("resource_allocation", "ddi")])
def test_filter_two_ranks_one_gpu(eps, dev, tmp_path, model, dataset, monkeypatch):
    from eps_amd import datasets, filter_stage, models, scan
    os.chdir(tmp_path)
    if dataset == "ddi":           # unit-valued graph: the threshold scan, sharded by columns; force the estimated-bar path
        monkeypatch.setenv("EPS_SYNTH_SCALE", "0.5")
        monkeypatch.setattr(scan, "SMALL_SET", 0)
        return _scan_case(tmp_path, model)
    extra = []
    if model == "gcn":   # a seeded random-init checkpoint both runs load
        extra = ["--num_layers", "2", "--hidden_channels", "32", "--dropout", "0.0", "--batch_size", "4096",
                 "--use_feature", "1", "--use_learnable_embedding", "1"]
        args = models.default_model_configs(filter_stage.make_parser().parse_args(
            ["--dataset", "collab", "--model", "gcn", "--checkpoint", "x", "--synthetic"] + extra))
        _, _, _, data = datasets.get_data(args)
        torch.manual_seed(0)
        os.makedirs("models", exist_ok=True)
        torch.save(models.build_model(args, data, torch.device("cpu")).state_dict(), "models/collab_gcn||0|0.pt")
        torch.save(torch.load("models/collab_gcn||0|0.pt"), "models/collab_gcn||0|1.pt")
    argv = lambda run: ["--dataset", "collab", "--model", model, "--checkpoint", f"collab_{model}||0|{run}.pt",  # noqa: E731
                        "--synthetic", "--keep_top", "700"] + extra
    single = torch.load(filter_stage.main(argv(0)))
    mp.spawn(_rank_main, args=(2, _free_port(), str(tmp_path), argv(1)), nprocs=2, join=True)
    multi = torch.load(f"filtered_edges/collab_{model}__0_1_sorted_edges.pt")
    assert single.shape == (700, 3)
    assert torch.equal(single[:, :2], multi[:, :2]), "same proposals in the same order"
    if model == "adamic_ogb":
        assert torch.equal(single, multi)
    else:   # the row-sharded last layer sums in the same order: bit-identical embeddings, hence scores
        assert torch.equal(single, multi)


def _scan_rank_main(rank, world, port, workdir, argv):
    os.environ["EPS_SYNTH_SCALE"] = "0.5"
    sys.path.insert(0, ROOT)
    import eps_amd  # noqa: F401
    from eps_amd import scan
    scan.SMALL_SET = 0
    _rank_main(rank, world, port, workdir, argv)


def _scan_case(tmp_path, model):
    from eps_amd import filter_stage
    argv = lambda run: ["--dataset", "ddi", "--model", model, "--checkpoint", f"ddi_{model}||0|{run}.pt",  # noqa: E731
                        "--synthetic", "--keep_top", "5000"]
    single = torch.load(filter_stage.main(argv(0)))
    mp.spawn(_scan_rank_main, args=(2, _free_port(), str(tmp_path), argv(1)), nprocs=2, join=True)
    multi = torch.load(f"filtered_edges/ddi_{model}__0_1_sorted_edges.pt")
    assert single.shape == (5000, 3) and torch.equal(single, multi)


@pytest.mark.parametrize("model,dataset", [("adamic_ogb", "collab"), ("simple", "ddi")])
def test_filter_two_ranks_full_list(eps, dev, tmp_path, model, dataset, monkeypatch):
    """Without --keep_top the reference writes ALL [E,3] rows: two ranks score their column shards, the shards are gathered in
    rank order (= candidate order) and sorted -- the file equals the single-process one bit for bit."""
    from eps_amd import filter_stage
    os.chdir(tmp_path)
    monkeypatch.setenv("EPS_SYNTH_SCALE", "0.25")
    argv = lambda run: ["--dataset", dataset, "--model", model, "--checkpoint", f"{dataset}_{model}||0|{run}.pt",  # noqa: E731
                        "--synthetic"]
    single = torch.load(filter_stage.main(argv(0)))
    mp.spawn(_scaled_rank_main, args=(2, _free_port(), str(tmp_path), argv(1)), nprocs=2, join=True)
    multi = torch.load(f"filtered_edges/{dataset}_{model}__0_1_sorted_edges.pt")
    assert single.shape[0] > 10_000 and torch.equal(single, multi)


def _scaled_rank_main(rank, world, port, workdir, argv):
    os.environ["EPS_SYNTH_SCALE"] = "0.25"
    _rank_main(rank, world, port, workdir, argv)


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_child_process(dev, tmp_path, scaling):
    """bench.py --gpus 2 as the driver launches it (python -m torch.distributed.run, fresh child processes), on one GPU
    with gloo: both scaling modes print ONE JSON line with the contract's keys and a whole-job candidate count."""
    import json
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--scaling", scaling, "--nodes", "30000", "--edges", "600000", "--keep_top", "20000", "--no-cpu", "--no-legs",
           "--backend", "gloo", "--one-device", "--sustain", "0.3"]
    out = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line
    assert line["n_gpus"] == 2 and line["scaling"] == scaling and line["value"] > 0
    per_graph = line["config"]["candidates_per_step_all_ranks"] // (2 if scaling == "weak" else 1)
    assert per_graph > 0 and 0 < line["roofline"]["frac"] <= 1
    assert (line["weak_value"] is not None) == (scaling == "strong")
    assert line["replicated_ms"] <= line["ms_per_step"] and len(line["sharded_ms_per_rank"]) == 2
    assert line["sustained"]["steps"] >= 2 and line["sustained"]["value"] > 0


def test_bench_self_launch_from_plain_shell(dev, tmp_path):
    """`python bench.py --gpus 2` with NO torchrun around it (how a person -- or a driver that does not wrap it -- starts
    it): the parent starts the two ranks itself, never touches a GPU, and passes the one JSON line through.  Strong scaling is
    the default for N > 1; the line carries per-rank kernel times, serial_ms and the weak figure as a secondary key."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--nodes", "30000",
           "--edges", "600000", "--keep_top", "20000", "--no-cpu", "--no-legs", "--backend", "gloo", "--one-device", "--sustain", "0.3"]
    out = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0 and line["weak_value"] > 0
    assert len(line["kernel_ms_per_rank"]) == 2 and "serial_ms" in line and "prep_ms" in line
    assert 0 < line["roofline"]["frac"] <= 1 and line["value_unordered_pairs_per_s"] * 2 == pytest.approx(line["value"])


def _relabelled_scan_rank_main(rank, world, port, workdir, dist_rows_min=0, rows_on=None, kind="aa", k=30000, dist_hist=True):
    sys.path.insert(0, ROOT)
    os.chdir(workdir)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import eps_amd  # noqa: F401
    from eps_amd import dist as epd, ops, scan, synth
    from eps_amd.heuristics import node_weight_table
    epd.init_from_env("gloo", 0)
    scan.RELABEL_MIN_NODES = 0
    scan.HEAD_MIN_PATHS = 0                          # (skipped heads on this small graph too: the sharded step runs them)
    scan.SMALL_SET = 0                               # the estimated-bar path
    scan.DIST_ROWS_MIN = dist_rows_min               # 0: the final ordering is dealt over the ranks by score range
    scan.DIST_HIST = dist_hist                       # the step's one exchange: score histograms (r06) or the scores themselves (r05)
    dev = torch.device("cuda:0")
    g = synth.rmat_graph(13, 10, 21, dev)
    w = torch.ones(g.n_rows, dtype=torch.float32, device=dev) if kind == "cn" else node_weight_table(g, ops.W_AA)
    st = {}
    pairs, scores = scan.scan_topk(g, w, k, rank, world, relabel=True, rows_on=rows_on, stats=st)
    assert scan.scan_graph(g)[1] is not None
    assert (pairs is None) == (rows_on is not None and rows_on != "shards" and rank != rows_on)
    torch.save(None if pairs is None else (pairs.cpu(), scores.cpu(), st.get("shard")), f"scan_rank{rank}.pt")
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world,dist_rows_min,rows_on", [(2, 0, None), (4, 0, None), (2, 1 << 30, None), (3, 0, 0), (2, 1 << 30, 1),
                                                         (2, 0, "shards"), (3, 0, "shards"), (4, 0, "shards"), (2, 1 << 30, "shards")])
def test_scan_two_ranks_hubs_first_labels(eps, dev, tmp_path, world, dist_rows_min, rows_on):
    """The sharded threshold scan on the hubs-first relabelled copy (what bench.py --scaling strong runs): every rank ends
    with the single-process list as scanned under the original labels, bit for bit -- with the final ordering dealt over the
    ranks by score range (2 and 4 ranks on this one GPU) and with every rank ordering all rows itself; ``rows_on``: the rows
    gathered onto one rank alone (what filter.py and bench.py ask for: the file is written by rank 0)."""
    from eps_amd import scan, synth
    from eps_amd.heuristics import node_weight_table
    g = synth.rmat_graph(13, 10, 21, dev)
    want_p, want_s = scan.scan_topk(g, node_weight_table(g, eps.ops.W_AA), 30000)
    assert scan.scan_graph(g)[1] is None
    mp.spawn(_relabelled_scan_rank_main, args=(world, _free_port(), str(tmp_path), dist_rows_min, rows_on), nprocs=world, join=True)
    if rows_on == "shards":
        # (r06) every rank keeps the chunk of the declared order it ordered: the chunks, in rank order, ARE the single-process rows
        parts = [torch.load(os.path.join(tmp_path, f"scan_rank{r}.pt")) for r in range(world)]
        total = parts[0][2][1]
        at = 0
        for p, s, shard in parts:
            assert shard == (at, total)
            at += p.shape[1]
        assert at == total == want_p.shape[1]
        assert torch.equal(torch.cat([p for p, _, _ in parts], 1), want_p.cpu())
        assert torch.equal(torch.cat([s for _, s, _ in parts]), want_s.cpu())
        return
    for r in range(world):
        got = torch.load(os.path.join(tmp_path, f"scan_rank{r}.pt"))
        if rows_on is not None and r != rows_on:          # (the rows travel to one rank only)
            assert got is None
            continue
        p, s, _ = got
        assert torch.equal(p, want_p.cpu()) and torch.equal(s, want_s.cpu())


@pytest.mark.parametrize("dist_hist", [True, False], ids=["histograms", "scores"])
@pytest.mark.parametrize("world", [2, 3])
def test_scan_ranks_tied_levels_beyond_the_room(eps, dev, tmp_path, world, dist_hist):
    """common-neighbour counts tie by the hundred thousand, so a rank's pre-filter threshold level holds far more pairs
    than the 2 k_pre + 65536 entries its outputs start with and that rank repeats the pre-filter with more room -- the step's
    exchange must still have one length on every rank.  With the histogram exchange (fixed 25 KB) and with the r05 score
    exchange (the ranks agree on the longest list first): the single-process rows on every rank."""
    from eps_amd import scan, synth
    g = synth.rmat_graph(13, 10, 21, dev)
    ones = torch.ones(g.n_rows, dtype=torch.float32, device=dev)
    want_p, want_s = scan.scan_topk(g, ones, 3000)
    mp.spawn(_relabelled_scan_rank_main, args=(world, _free_port(), str(tmp_path), 0, None, "cn", 3000, dist_hist), nprocs=world, join=True)
    for r in range(world):
        p, s, _ = torch.load(os.path.join(tmp_path, f"scan_rank{r}.pt"))
        assert torch.equal(p, want_p.cpu()) and torch.equal(s, want_s.cpu())


@pytest.mark.parametrize("gpus", [1, 2])
def test_bench_failure_still_prints_one_json_line(dev, tmp_path, gpus):
    """A rank that fails inside the bench (here: an impossible --keep_top) leaves ONE parseable JSON line with "error" and a
    non-zero exit code -- alone, and under the launcher with a second rank."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "1", "--warmup", "0", "--nodes", "5000",
           "--edges", "60000", "--keep_top", "0", "--no-cpu", "--no-legs", "--backend", "gloo", "--one-device", "--sustain", "0"]
    for attempt in range(3):       # (a rendezvous that loses the race for its port fails before any rank runs: once more)
        out = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{") and l.rstrip().endswith("}")]
        if lines:
            break
    assert out.returncode != 0, (out.stdout[-1000:], out.stderr[-2000:])
    assert lines, (out.stdout[-1000:], out.stderr[-2000:])
    line = json.loads(lines[-1])
    assert line["value"] is None and "EpsError" in line["error"] and line["n_gpus"] == gpus


def _uneven_scan_rank_main(rank, world, port, workdir, k):
    sys.path.insert(0, ROOT)
    os.chdir(workdir)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import eps_amd  # noqa: F401
    from eps_amd import dist as epd, ops, scan, synth
    from eps_amd.heuristics import node_weight_table
    epd.init_from_env("gloo", 0)
    scan.SMALL_SET = 1 << 40                         # no bar: every candidate is a survivor

    def lopsided(g, r, w):                           # rank 0 scans nine tenths of the heaviest-first order, rank 1 the light rest
        order = scan.column_order(g)
        cut = order.numel() * 9 // 10
        return (order[:cut] if r == 0 else order[cut:]).contiguous()
    scan.shard_columns = lopsided
    dev = torch.device("cuda:0")
    g = synth.rmat_graph(11, 8, 5, dev)
    pairs, scores = scan.scan_topk(g, node_weight_table(g, ops.W_AA), k, rank, world)
    torch.save((pairs.cpu(), scores.cpu()), f"uneven_rank{rank}.pt")
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("k_of_all", [0.98, 1.0, 1.5])
def test_scan_two_ranks_uneven_shards_without_bar(eps, dev, tmp_path, k_of_all):
    """uneven shards, no bar, k at or beyond the candidate count.  Rank 1 holds far fewer survivors than its share of
    k, so the union of the locally pre-filtered lists can fall short of k2 while enough candidates exist: the job-wide cut comes
    back -inf next to a finite pre-filter threshold, and the step must go round again with everything re-scored instead of
    returning a short list.  Both ranks end with the single-process rows."""
    from eps_amd import scan, synth
    from eps_amd.heuristics import node_weight_table
    g = synth.rmat_graph(11, 8, 5, dev)
    w = node_weight_table(g, eps.ops.W_AA)
    st = {}
    scan.scan_topk(g, w, 10, stats=st)
    k = int(st["candidates"] * k_of_all)
    want_p, want_s = scan.scan_topk(g, w, k)
    assert want_p.shape[1] == min(k, st["candidates"])
    mp.spawn(_uneven_scan_rank_main, args=(2, _free_port(), str(tmp_path), k), nprocs=2, join=True)
    for r in range(2):
        p, s = torch.load(os.path.join(tmp_path, f"uneven_rank{r}.pt"))
        assert torch.equal(p, want_p.cpu()) and torch.equal(s, want_s.cpu())
