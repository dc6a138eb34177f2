"""bench.py's output contract, as the driver consumes it (one JSON line from rank 0)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

REQUIRED = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int,
            "ms_per_step": float, "higher_is_better": bool, "scaling": str, "dtype": str, "data": str,
            "config": dict, "roofline": dict}


def check_line(text, n_gpus, steps, warmup):
    lines = [l for l in text.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, text
    d = json.loads(lines[0])
    for key, typ in REQUIRED.items():
        assert key in d and isinstance(d[key], typ), (key, d.get(key))
    assert "vs_baseline" in d and d["vs_baseline"] is None            # BASELINE.md publishes no number
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup
    assert d["metric"].startswith("edges propagated/sec") and d["unit"] == "edges/s" and d["dtype"] == "f32"
    assert d["scaling"] == "strong" and d["data"] == "synthetic" and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["achieved"] > 0 and "traffic" in r
    assert d["value"] > 0 and d["ms_per_step"] > 0
    return d


def test_single_gpu_line(device):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "small", "--steps", "3",
                          "--warmup", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = check_line(out.stdout, 1, 3, 1)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "edges/s" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]


def test_two_rank_launch_as_the_driver_does_it(device):
    """python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 (gloo: two ranks share one GPU)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "small", "--backend", "gloo"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = check_line(out.stdout, 2, 2, 1)
    assert "cpu_baseline" not in d                                    # rank 0 at N = 1 only
    assert d["roofline"]["launches_timed"] == 2 * 3


def test_plain_python_launch_with_gpus_2_starts_its_own_ranks(device):
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE (how the driver ran BENCH): bench.py starts
    the ranks itself as child processes and relays rank 0's line with the children's exit code."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--config", "small", "--backend", "gloo"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = check_line(out.stdout, 2, 2, 1)
    assert "cpu_baseline" not in d


def test_the_multi_gpu_code_path_rehearsed_with_one_rank_over_rccl(device):
    """`bench.py --rehearse-partition`: the N > 1 code path -- nccl process group with device_id, PartitionedPropagator, the
    forward recorded with its collectives as one HIP graph, barrier fences, MAX over ranks -- as a partition of ONE rank over
    the real backend (RCCL refuses two ranks on one device); stdout still carries exactly one line (the communicator's
    banner goes to stderr)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rehearse-partition", "--steps", "3", "--warmup", "1",
                          "--config", "small"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(out.stdout.strip().splitlines()) == 1, out.stdout
    d = check_line(out.stdout, 1, 3, 1)
    assert "recorded as one HIP graph (collectives included)" in d["config"]["parallelism"], d["config"]["parallelism"]
    assert "cpu_baseline" not in d and d["roofline"]["launches_timed"] == 3 * 3
