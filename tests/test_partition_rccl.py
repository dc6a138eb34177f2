"""The partitioned path over RCCL (torch.distributed backend "nccl"), one rank per GPU -- what the driver's
multi-GPU bench runs.  Needs at least two devices, so it is skipped on the one-GPU test box and runs on the
8-GPU node; the same workers run there over gloo on one GPU (test_partition_gpu.py)."""
import pytest
import torch

from test_partition_gpu import _train_worker, _worker, run_ranks

pytestmark = pytest.mark.gpu


def _world():
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"RCCL needs one GPU per rank; {n} device(s) visible")
    return min(n, 4)          # at most 6 processes may share a box's GPUs


def test_rccl_forward_matches_single_gpu(device):
    world = _world()
    results = run_ranks(_worker, world, backend="nccl")
    for rank, r in results.items():
        assert r["backend"] == "nccl" and r["world"] == world
        assert r["own"] <= 1e-5 and r["items"] <= 1e-5 and r["full"] <= 1e-5 and r["worst_row"] <= 1e-5, (rank, r)
    assert abs(sum(r["share"] for r in results.values()) - 1.0) < 1e-6


def test_rccl_training_step_matches_single_gpu(device):
    world = _world()
    results = run_ranks(_train_worker, world, backend="nccl")
    for rank, r in results.items():
        assert r["backend"] == "nccl" and r["world"] == world
        assert r["bpr"] <= 1e-5 and r["reg"] <= 1e-5 and r["own"] <= 1e-5 and r["items"] <= 1e-5, (rank, r)
