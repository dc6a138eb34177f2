"""The N>1 path on CPU ranks over gloo: partition, per-hop exchange, Horner sequencing, gather.

The arithmetic is supplied by the oracle-based test double (tests/cpu_ops.py); what is under test is
gnn_ecommerce_amd/partition.py itself -- the same object the GPU ranks run with HipOps over RCCL."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    """A fresh rendezvous FILE (no TCP port to race for); the name is kept for the call sites."""
    import tempfile
    fd, path = tempfile.mkstemp(prefix="lgcn_rdzv_")
    os.close(fd)
    os.unlink(path)
    return path


def _worker(rank, world, port, layers, dim, alpha_kind, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    try:
        from cpu_ops import CpuOps
        from oracle import lightgcn_oracle as oracle
        from gnn_ecommerce_amd import synth
        from gnn_ecommerce_amd.partition import PartitionedPropagator
        g = synth.make_bipartite(600, 90, 5000, seed=7)
        ei, ew = g.coo()
        n = g.num_nodes
        x0 = synth.xavier_table(n, dim, 1)
        alpha = oracle.default_alpha(layers) if alpha_kind == "uniform" else torch.linspace(0.5, 0.1, layers + 1)
        want = oracle.get_embedding(x0, alpha, ei, ew, layers)
        pp = PartitionedPropagator(ei, ew, g.n_users, g.n_items, rank, world, ops=CpuOps())
        out = pp.propagate_sum(x0, alpha.tolist())
        lo, hi = pp.ranges[rank]

        def rel(a, b):
            return ((a.double() - b.double()).norm() / b.double().norm()).item()

        res = {"own_users": rel(out[lo:hi], want[lo:hi]) if hi > lo else 0.0,
               "items": rel(out[g.n_users:], want[g.n_users:]),
               "ranges": pp.ranges, "local_nnz": pp.local_nnz}
        full = pp.gather_users(out.clone())
        res["gathered"] = rel(full, want)
        # every rank must hold bit-identical item rows after the exchange
        items = [torch.empty_like(out[g.n_users:]) for _ in range(world)]
        dist.all_gather(items, out[g.n_users:].contiguous())
        res["items_identical"] = all(torch.equal(items[0], t) for t in items)
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,layers,dim,alpha_kind", [(2, 3, 64, "uniform"), (3, 2, 90, "ramp"), (2, 1, 16, "ramp"), (8, 3, 64, "uniform")])
def test_partitioned_propagate_matches_single_process_oracle(world, layers, dim, alpha_kind):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, layers, dim, alpha_kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ranges = results[0]["ranges"]
    assert ranges[0][0] == 0 and ranges[-1][1] == 600
    assert sum(r["local_nnz"] for r in results.values()) == 2 * 5000          # shards partition the edges
    for rank, r in results.items():
        assert r["ranges"] == ranges
        assert r["own_users"] <= 1e-5 and r["items"] <= 1e-5 and r["gathered"] <= 1e-5, (rank, r)
        assert r["items_identical"]


def _train_worker(rank, world, port, seeded, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    try:
        from cpu_ops import CpuOps
        from oracle import lightgcn_oracle as oracle
        from gnn_ecommerce_amd import synth
        from gnn_ecommerce_amd.partition import PartitionedPropagator, partitioned_bpr_loss
        g = synth.make_bipartite(400, 70, 3000, seed=11)
        ei, ew = g.coo()
        n, dim, layers, decay, batch = g.num_nodes, 32, 3, 1e-4, 64
        alpha = oracle.default_alpha(layers)
        w0 = synth.xavier_table(n, dim, 4)
        gen = torch.Generator().manual_seed(5)
        users = torch.randperm(g.n_users, generator=gen)[:batch]
        if world == 3:
            users = users[users < 150]            # leave the last rank without a single own triple
            batch = users.numel()
        pos = torch.randint(0, g.n_items, (batch,), generator=gen) + g.n_users
        neg = torch.randint(0, g.n_items, (batch,), generator=gen) + g.n_users
        # single-process reference (src/train_lightgcn.py:137-146 through the oracle)
        wr = w0.clone().requires_grad_(True)
        _, bpr, reg, loss = oracle.train_step_loss(wr, alpha, ei, ew, users, pos, neg, layers, decay)
        loss.backward()
        # partitioned
        pp = PartitionedPropagator(ei, ew, g.n_users, g.n_items, rank, world, ops=CpuOps())
        w = w0.clone().requires_grad_(True)
        from gnn_ecommerce_amd import propagate
        propagate.SEED_ROWS_FACTOR = 0 if seeded else 10 ** 9
        local, gbpr, greg = partitioned_bpr_loss(pp, w, alpha.tolist(), users, pos, neg, decay,
                                                 pair_scores=None if seeded else oracle.pair_scores)
        local.backward()
        lo, hi = pp.ranges[rank]

        def rel(a, b):
            return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()

        foreign = torch.ones(g.n_users, dtype=torch.bool)
        foreign[lo:hi] = False
        # PartitionedTrainer: the same step without autograd + Adam over the owned rows, two steps, against
        # torch.optim.Adam on the single-process loss
        from gnn_ecommerce_amd.trainer import PartitionedTrainer
        ref_w = torch.nn.Parameter(w0.clone())
        ref_opt = torch.optim.Adam([ref_w], lr=0.01)
        wt = w0.clone()
        trainer = PartitionedTrainer(pp, wt, alpha.tolist(), lr=0.01, decay=decay, batch=batch)
        trainer_stats = []
        for k in range(2):
            perm = torch.roll(torch.arange(batch), k)
            ref_opt.zero_grad()
            _, r_bpr, r_reg, r_loss = oracle.train_step_loss(ref_w, alpha, ei, ew, users[perm], pos, neg, layers, decay)
            r_loss.backward()
            ref_opt.step()
            st = trainer.step(users[perm], pos, neg)
            trainer_stats.append(max(abs(st[0].item() - r_bpr.item()) / abs(r_bpr.item()), abs(st[1].item() - r_reg.item()) / abs(r_reg.item())))
        own = torch.zeros(n, dtype=torch.bool)
        own[lo:hi] = True
        own[g.n_users:] = True
        trainer_rows = rel((wt - w0)[own], (ref_w.detach() - w0)[own])
        trainer_foreign = bool(torch.equal(wt[:g.n_users][foreign], w0[:g.n_users][foreign]))
        q.put((rank, {"bpr": abs(gbpr.item() - bpr.item()) / abs(bpr.item()),
                      "trainer_stats": max(trainer_stats), "trainer_rows": trainer_rows, "trainer_foreign": trainer_foreign,
                      "reg": abs(greg.item() - reg.item()) / abs(reg.item()),
                      "grad_own_users": rel(w.grad[lo:hi], wr.grad[lo:hi]) if hi > lo else 0.0,
                      "grad_items": rel(w.grad[g.n_users:], wr.grad[g.n_users:]),
                      "grad_foreign_zero": bool((w.grad[:g.n_users][foreign] == 0).all()),
                      "own_triples": int(((users >= lo) & (users < hi)).sum())}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("seeded", [True, False], ids=["seeded_node", "dense_backward"])
@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_training_step_matches_single_process_gradients(world, seeded):
    """Row (e) of the hot-path contract end to end: forward, pair routing to the user's owner, BPR + regulariser,
    backward on A^T with the item-gradient all-reduce -- against the oracle's autograd on one process."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, seeded, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if world == 3:
        assert results[2]["own_triples"] == 0           # the collective path without any own pair
    for rank, r in results.items():
        assert r["bpr"] <= 1e-5 and r["reg"] <= 1e-5, (rank, r)
        assert r["grad_own_users"] <= 1e-5 and r["grad_items"] <= 1e-5, (rank, r)
        assert r["grad_foreign_zero"], (rank, r)
        # two trainer steps: the update of the rows a rank owns equals torch.optim.Adam's on the single-process loss
        assert r["trainer_stats"] <= 1e-5 and r["trainer_rows"] <= 1e-4 and r["trainer_foreign"], (rank, r)
