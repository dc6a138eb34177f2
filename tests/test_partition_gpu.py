"""The partitioned path with the REAL HipOps: two ranks share cuda:0 and exchange over gloo (RCCL
refuses two ranks on one device; the driver's 8-GPU run uses backend "nccl" through the same code).
Checked against the single-GPU HIP result and, through it, the oracle."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gnn_ecommerce_amd as lg
        from gnn_ecommerce_amd import synth
        from gnn_ecommerce_amd.partition import PartitionedPropagator
        dev = torch.device("cuda:0")
        g = synth.make_bipartite(20000, 1500, 150000, seed=3)
        ei, ew = g.coo(dev)
        n, dim, alphas = g.num_nodes, 64, (0.4, 0.3, 0.2, 0.1)
        x0 = synth.xavier_table(n, dim, 2, dev)
        single = lg.propagate_sum(x0, lg.PropGraph(ei, ew, n), alphas)
        pp = PartitionedPropagator(ei, ew, g.n_users, g.n_items, rank, world)
        out = pp.propagate_sum(x0, alphas)
        full = pp.gather_users(out.clone())
        torch.cuda.synchronize()

        def rel(a, b):
            return ((a.double() - b.double()).norm() / b.double().norm()).item()

        lo, hi = pp.ranges[rank]
        q.put((rank, {"own": rel(out[lo:hi], single[lo:hi]), "items": rel(out[g.n_users:], single[g.n_users:]),
                      "full": rel(full, single), "worst_row": ((full - single).norm(dim=1) / single.norm(dim=1)).max().item(),
                      "share": pp.local_nnz / g.nnz}))
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_match_single_gpu(device):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, r in results.items():
        assert r["own"] <= 1e-5 and r["items"] <= 1e-5 and r["full"] <= 1e-5 and r["worst_row"] <= 1e-5, (rank, r)
        assert 0.45 <= r["share"] <= 0.55
