"""Two ranks on ONE GPU (gloo as the transport, EPS_DIST_ONE_DEVICE=1): the multi-rank control flow of the filter stage
-- column sharding by work, per-rank streaming top-K, rank-ordered merge, row-sharded last GNN layer + all-gather -- on
the real kernels, against the single-process result."""
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
                      MASTER_PORT=str(port), EPS_DIST_ONE_DEVICE="1")
    import eps_amd  # noqa: F401
    from eps_amd import candidates, filter_stage
    candidates.DEFAULT_BLOCK_PATHS = 30_000          # several blocks per rank: the in-kernel cut runs too
    filter_stage.main(argv)
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("model", ["adamic_ogb", "gcn"])
def test_filter_two_ranks_one_gpu(eps, dev, tmp_path, model):
    from eps_amd import datasets, filter_stage, models
    os.chdir(tmp_path)
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
