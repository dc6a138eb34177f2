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


def _init(rank, world, port, backend):
    """gloo: every rank on cuda:0.  nccl (= RCCL): one rank per GPU, as the driver launches bench.py."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device(f"cuda:{rank}" if backend == "nccl" else "cuda:0")
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dev


def _worker(rank, world, port, q, backend="gloo"):
    dev = _init(rank, world, port, backend)
    try:
        import gnn_ecommerce_amd as lg
        from gnn_ecommerce_amd import synth
        from gnn_ecommerce_amd.partition import PartitionedPropagator
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
                      "share": pp.local_nnz / g.nnz, "world": dist.get_world_size(), "backend": dist.get_backend()}))
    finally:
        dist.destroy_process_group()


def run_ranks(target, world, backend="gloo"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        results = dict(q.get(timeout=300) for _ in range(world))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(results) == list(range(world))
    return results


def test_two_ranks_one_gpu_match_single_gpu(device):
    results = run_ranks(_worker, 2)
    for rank, r in results.items():
        assert r["own"] <= 1e-5 and r["items"] <= 1e-5 and r["full"] <= 1e-5 and r["worst_row"] <= 1e-5, (rank, r)
        assert 0.45 <= r["share"] <= 0.55


def _train_worker(rank, world, port, q, backend="gloo"):
    dev = _init(rank, world, port, backend)
    try:
        import gnn_ecommerce_amd as lg
        from gnn_ecommerce_amd import synth
        from gnn_ecommerce_amd.partition import PartitionedPropagator, partitioned_bpr_loss
        g = synth.make_bipartite(20000, 1500, 150000, seed=3)
        ei, ew = g.coo(dev)
        n, dim, layers, decay, batch = g.num_nodes, 64, 3, 1e-4, 1024
        w0 = synth.xavier_table(n, dim, 2, dev)
        gen = torch.Generator().manual_seed(9)
        users = torch.randperm(g.n_users, generator=gen)[:batch].to(dev)
        pos = (torch.randint(0, g.n_items, (batch,), generator=gen) + g.n_users).to(dev)
        neg = (torch.randint(0, g.n_items, (batch,), generator=gen) + g.n_users).to(dev)
        # single-GPU model, the step of src/train_lightgcn.py:137-146
        model = lg.LightGCN(n, dim, layers)
        model.load_state_dict({"alpha": model.alpha, "embedding.weight": w0.cpu()})
        model.to(dev)
        labels = torch.stack((torch.cat([users, users]), torch.cat([pos, neg])))
        out = model(ei, labels, ew)
        bpr = model.recommendation_loss(out[:batch], out[batch:], 0) * batch
        w = model.embedding.weight
        reg = 0.5 * (w[users].norm().pow(2) + w[pos].norm().pow(2) + w[neg].norm().pow(2)) / batch * decay
        (bpr + reg).backward()
        ref_grad = w.grad
        # partitioned, real HipOps
        pp = PartitionedPropagator(ei, ew, g.n_users, g.n_items, rank, world)
        wp = w0.clone().requires_grad_(True)
        local, gbpr, greg = partitioned_bpr_loss(pp, wp, [0.25] * 4, users, pos, neg, decay)
        local.backward()
        torch.cuda.synchronize()
        lo, hi = pp.ranges[rank]

        def rel(a, b):
            return ((a.double() - b.double()).norm() / b.double().norm()).item()

        q.put((rank, {"bpr": abs(gbpr.item() - bpr.item()) / abs(bpr.item()),
                      "reg": abs(greg.item() - reg.item()) / abs(reg.item()),
                      "own": rel(wp.grad[lo:hi], ref_grad[lo:hi]), "items": rel(wp.grad[g.n_users:], ref_grad[g.n_users:]),
                      "world": dist.get_world_size(), "backend": dist.get_backend()}))
    finally:
        dist.destroy_process_group()


def test_two_ranks_training_step_matches_single_gpu(device):
    results = run_ranks(_train_worker, 2)
    for rank, r in results.items():
        assert r["bpr"] <= 1e-5 and r["reg"] <= 1e-5 and r["own"] <= 1e-5 and r["items"] <= 1e-5, (rank, r)


def test_exchange_hook_of_the_c_abi_drives_a_partitioned_hop(device):
    """include/lgconv_hip.h: lgc_hop_exchange = item step -> caller's exchange callback on the item block -> user step,
    called through ctypes the way a non-Python host would, with the callback standing in for ncclAllReduce: it adds the
    other rank's partial item block (both 'ranks' live in this process).  Result = the single-GPU hop on rank 0's rows."""
    import ctypes
    import gnn_ecommerce_amd as lg
    from gnn_ecommerce_amd import _native, synth
    from gnn_ecommerce_amd.partition import PartitionedPropagator
    g = synth.make_bipartite(6000, 700, 50000, seed=5)
    ei, ew = g.coo(device)
    n, nu, dim = g.num_nodes, g.n_users, 64
    x = synth.xavier_table(n, dim, 1, device)
    r = synth.xavier_table(n, dim, 2, device)
    want = lg.PropGraph(ei, ew, n).forward_op.apply(x, torch.empty_like(x), a=0.5, r=r, b=0.25)
    pp = [PartitionedPropagator(ei, ew, nu, g.n_items, rank, 2) for rank in range(2)]
    other = torch.zeros_like(x)
    pp[1].item_op.apply(x, other, a=0.5)                       # rank 1's partial sums (rank 0 carries the epilogue term)
    y = torch.full_like(x, float("nan"))
    seen = {}

    def exchange(block, rows, row_stride, d, stream, user):
        seen.update(block=block, rows=rows, row_stride=row_stride, dim=d, stream=stream)
        y[nu:] += other[nu:]                                   # in stream order on the current stream
        return 0

    cb = _native.EXCHANGE_FN(exchange)
    lib = _native.load()
    item_c, user_c = pp[0].item_op.c_struct(dim, False), pp[0].user_op.c_struct(dim, False)
    code = lib.lgc_hop_exchange(ctypes.byref(item_c), ctypes.byref(user_c), n, x.data_ptr(), x.stride(0), y.data_ptr(),
                                y.stride(0), r.data_ptr(), r.stride(0), 0.5, 0.25, dim, nu, g.n_items, cb, None,
                                _native.stream_of(device))
    assert code == 0
    torch.cuda.synchronize()
    assert seen["block"] == y[nu:].data_ptr() and seen["rows"] == g.n_items and seen["row_stride"] == dim and seen["dim"] == dim
    lo, hi = pp[0].ranges[0]

    def rel(a, b):
        return ((a.double() - b.double()).norm() / b.double().norm()).item()

    # rank 0 adds b * r to the item block it contributes (the item step of a rank-0 operator applies the epilogue)
    assert rel(y[nu:], want[nu:]) <= 1e-5 and rel(y[lo:hi], want[lo:hi]) <= 1e-6
    assert torch.isnan(y[hi:nu]).all()                         # the other rank's users are not touched
    # a failing callback aborts the hop with its code
    bad = _native.EXCHANGE_FN(lambda *a: 7)
    assert lib.lgc_hop_exchange(ctypes.byref(item_c), ctypes.byref(user_c), n, x.data_ptr(), x.stride(0), y.data_ptr(),
                                y.stride(0), None, 0, 1.0, 0.0, dim, nu, g.n_items, bad, None, _native.stream_of(device)) == 7
    assert lib.lgc_hop_exchange(ctypes.byref(item_c), ctypes.byref(user_c), n, x.data_ptr(), x.stride(0), y.data_ptr(),
                                y.stride(0), None, 0, 1.0, 0.0, dim, nu, g.n_items + 1, cb, None, None) == -1
