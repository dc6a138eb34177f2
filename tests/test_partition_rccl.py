"""The partitioned path over RCCL (torch.distributed backend "nccl"), one rank per GPU -- what the driver's
multi-GPU bench runs.  The two-rank tests need at least two devices, so they are skipped on the one-GPU test box and
run on the 8-GPU node; the same workers run there over gloo on one GPU (test_partition_gpu.py).  What DOES run on one
GPU: a partition of ONE rank forced through the backend (LGCN_COMM_FORCE=1) -- RCCL refuses two ranks on one device, but
a one-rank communicator is legal, and everything on torch's side of the collectives is the same: eager communicator
creation with ``device_id``, asynchronous all-reduce handles on views of the tables, ``wait()`` on the launch stream,
``all_gather_into_tensor``, the barrier, and ProcessGroupNCCL's watchdog thread working beside the HIP-graph capture of
the recorded trainer (thread-local capture mode)."""
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
        assert r["recorded_same"], (rank, r)
    assert len({r["recorded"] for r in results.values()}) == 1          # every rank replays, or none does
    assert abs(sum(r["share"] for r in results.values()) - 1.0) < 1e-6


def test_rccl_training_step_matches_single_gpu(device):
    world = _world()
    results = run_ranks(_train_worker, world, backend="nccl")
    for rank, r in results.items():
        assert r["backend"] == "nccl" and r["world"] == world
        assert r["bpr"] <= 1e-5 and r["reg"] <= 1e-5 and r["own"] <= 1e-5 and r["items"] <= 1e-5, (rank, r)


def test_one_rank_goes_through_the_rccl_process_group(device, monkeypatch):
    """Forward, both training paths, sharded Adam and the recorded trainer of a ONE-rank partition with every collective
    issued to the real backend (identities there): the same checks as the two-rank workers."""
    monkeypatch.setenv("LGCN_COMM_FORCE", "1")             # read at import time by the spawned worker
    fwd = run_ranks(_worker, 1, backend="nccl")[0]
    assert fwd["backend"] == "nccl" and fwd["world"] == 1
    assert fwd["own"] <= 1e-5 and fwd["items"] <= 1e-5 and fwd["full"] <= 1e-5 and fwd["worst_row"] <= 1e-5, fwd
    assert fwd["recorded"] and fwd["recorded_same"], fwd    # the forward with its collectives as ONE replayed HIP graph
    r = run_ranks(_train_worker, 1, backend="nccl")[0]
    assert r["backend"] == "nccl" and r["world"] == 1
    assert r["bpr"] <= 1e-5 and r["reg"] <= 1e-5 and r["own"] <= 1e-5 and r["items"] <= 1e-5, r
    assert r["seeded_bpr"] <= 1e-5 and r["seeded_own"] <= 1e-5 and r["seeded_items"] <= 1e-5 and r["seeded_node"], r
    assert r["trainer_graph_equals_eager"] and r["trainer_graph_launches"] >= 2, r
    assert r["trainer_full_equals_eager"] and r["trainer_full_launches"] == 1, r      # ONE graph, the collectives inside
    assert r["trainer_own"] <= 1e-5 and r["trainer_items"] <= 1e-5 and r["trainer_bpr"] <= 1e-5, r
    assert r["adam_own"] <= 1e-5 and r["adam_items"] <= 1e-5, r
