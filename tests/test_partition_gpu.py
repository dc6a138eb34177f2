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
    # graphs this small would keep the full last item step (Operator.listed_rows_pay): the listed-rows exchange is what is tested
    os.environ.setdefault("LGCN_LISTED_ROWS_MAX_SHARE", "1e9")
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
        # the forward recorded as one HIP graph with its collectives (capturable over RCCL; over gloo every rank falls back to
        # the eager forward together): the same rows either way, and again after the input changed in place
        from gnn_ecommerce_amd.partition import RecordedForward
        rec = RecordedForward(pp, x0, alphas)
        again = rec().clone()
        x0.mul_(0.5)
        half = rec().clone()
        x0.mul_(2.0)
        torch.cuda.synchronize()
        same = (torch.equal(again[lo:hi], out[lo:hi]) and torch.equal(again[g.n_users:], out[g.n_users:])
                and rel(half[lo:hi], 0.5 * out[lo:hi]) <= 1e-6 and rel(half[g.n_users:], 0.5 * out[g.n_users:]) <= 1e-6)
        q.put((rank, {"recorded": rec.recorded, "recorded_same": bool(same),
                      "own": rel(out[lo:hi], single[lo:hi]), "items": rel(out[g.n_users:], single[g.n_users:]),
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
        assert r["recorded_same"] and not r["recorded"], (rank, r)      # gloo cannot be captured: both ranks went eager, together


def _uses_node(fn, name, depth=6):
    """Whether an autograd graph contains a node whose class name holds ``name`` (searched a few levels deep)."""
    if fn is None or depth == 0:
        return False
    return name in type(fn).__name__ or any(_uses_node(nxt, name, depth - 1) for nxt, _ in fn.next_functions)


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
        ref_grad = w.grad.clone()
        from gnn_ecommerce_amd.optim import Adam as HipAdam
        HipAdam([w], lr=0.005).step()                       # the single-GPU step's dense Adam
        ref_w = w.detach()
        # partitioned, real HipOps
        from gnn_ecommerce_amd import partition
        pp = PartitionedPropagator(ei, ew, g.n_users, g.n_items, rank, world)
        lo, hi = pp.ranges[rank]

        def rel(a, b):
            return ((a.double() - b.double()).norm() / b.double().norm()).item()

        res = {"world": dist.get_world_size(), "backend": dist.get_backend()}
        for seeded in (True, False):
            partition.SEEDED_STEP = seeded
            lg.propagate.SEED_ROWS_FACTOR = 0                   # the seeded node whatever the table size (21,500 rows here)
            wp = w0.clone().requires_grad_(True)
            local, gbpr, greg = partitioned_bpr_loss(pp, wp, [0.25] * 4, users, pos, neg, decay, zero_foreign_rows=not seeded)
            res[("seeded_" if seeded else "") + "node"] = _uses_node(local.grad_fn, "PartitionedStep")
            local.backward()
            torch.cuda.synchronize()
            tag = "seeded_" if seeded else ""
            res.update({tag + "bpr": abs(gbpr.item() - bpr.item()) / abs(bpr.item()),
                        tag + "reg": abs(greg.item() - reg.item()) / abs(reg.item()),
                        tag + "own": rel(wp.grad[lo:hi], ref_grad[lo:hi]),
                        tag + "items": rel(wp.grad[g.n_users:], ref_grad[g.n_users:])})
            if seeded:
                # sharded Adam: only the rows this rank owns; they must equal the single-GPU dense step's rows, and the
                # rows other ranks own must not have been touched
                HipAdam([wp], lr=0.005, row_ranges=pp.owned_row_ranges()).step()
                torch.cuda.synchronize()
                res["adam_own"] = rel(wp.detach()[lo:hi] - w0[lo:hi], ref_w[lo:hi] - w0[lo:hi])
                res["adam_items"] = rel(wp.detach()[g.n_users:] - w0[g.n_users:], ref_w[g.n_users:] - w0[g.n_users:])
                foreign = torch.ones(g.n_users, dtype=torch.bool, device=dev)
                foreign[lo:hi] = False
                res["adam_foreign_untouched"] = bool(torch.equal(wp.detach()[:g.n_users][foreign], w0[:g.n_users][foreign]))
        # PartitionedTrainer: the same launches without autograd, eager and as replayed HIP graphs (one graph per stretch of
        # work between two collectives): four steps each from the same start, bit for bit the same table; the first step
        # equals the single-GPU step's update on the rows this rank owns
        from gnn_ecommerce_amd.trainer import PartitionedTrainer
        lg.propagate.SEED_ROWS_FACTOR = 0
        tables = {}
        for graphs in (False, True, "full"):
            wt = w0.clone()
            tr = PartitionedTrainer(pp, wt, [0.25] * 4, lr=0.005, decay=decay, batch=batch, graphs=graphs, warmup=1)
            stats = []
            for k in range(4):
                stats.append(tr.step(torch.roll(users, k), pos, neg).clone())
                if k == 0:
                    first = wt.clone()
            torch.cuda.synchronize()
            tables[graphs] = (wt, first, torch.stack(stats), tr.graph_launches)
        res["trainer_graph_equals_eager"] = bool(torch.equal(tables[True][0], tables[False][0])
                                                 and torch.equal(tables[True][2], tables[False][2]))
        res["trainer_graph_launches"] = tables[True][3]
        # graphs="full": the whole step as ONE graph with the collectives inside over nccl; elsewhere every rank falls back to
        # the segmented recording together -- the same table either way
        res["trainer_full_equals_eager"] = bool(torch.equal(tables["full"][0], tables[False][0])
                                                and torch.equal(tables["full"][2], tables[False][2]))
        res["trainer_full_launches"] = tables["full"][3]
        first = tables[False][1]
        res["trainer_own"] = rel(first[lo:hi] - w0[lo:hi], ref_w[lo:hi] - w0[lo:hi])
        res["trainer_items"] = rel(first[g.n_users:] - w0[g.n_users:], ref_w[g.n_users:] - w0[g.n_users:])
        res["trainer_bpr"] = abs(tables[False][2][0, 0].item() - bpr.item()) / abs(bpr.item())
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_two_ranks_training_step_matches_single_gpu(device):
    results = run_ranks(_train_worker, 2)
    for rank, r in results.items():
        assert r["bpr"] <= 1e-5 and r["reg"] <= 1e-5 and r["own"] <= 1e-5 and r["items"] <= 1e-5, (rank, r)
        assert r["seeded_bpr"] <= 1e-5 and r["seeded_reg"] <= 1e-5 and r["seeded_own"] <= 1e-5 and r["seeded_items"] <= 1e-5, (rank, r)
        assert r["seeded_node"] and not r["node"], (rank, r)
        assert r["trainer_graph_equals_eager"] and r["trainer_graph_launches"] >= 2, (rank, r)
        assert r["trainer_full_equals_eager"] and r["trainer_full_launches"] >= 2, (rank, r)     # gloo: segments, agreed
        assert r["trainer_own"] <= 1e-5 and r["trainer_items"] <= 1e-5 and r["trainer_bpr"] <= 1e-5, (rank, r)
        assert r["adam_own"] <= 1e-5 and r["adam_items"] <= 1e-5 and r["adam_foreign_untouched"], (rank, r)


def test_exchange_hook_of_the_c_abi_drives_a_partitioned_hop(device):
    """include/lgconv_hip.h: lgc_hop_exchange = item step -> caller's exchange callback on the item block -> user step,
    called through ctypes the way a non-Python host would, BOTH ranks through the hook with the same (r, b) -- the
    `item_epilogue` flag says which rank's item step carries the b * r term into the summed block (without it
    the exchanged rows would hold a A x + world * b r).  The callbacks stand in for ncclAllReduce: rank 1 runs first and
    parks its partial block, rank 0's callback adds it.  Result = the single-GPU hop on each rank's rows."""
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
    lib = _native.load()
    ys = [torch.full_like(x, float("nan")) for _ in range(2)]
    parked = {}
    seen = {}

    def exchange_rank1(block, rows, row_stride, d, stream, user):
        parked["partial"] = ys[1][nu:].clone()                 # what rank 1 contributes to the sum
        return 0

    def exchange_rank0(block, rows, row_stride, d, stream, user):
        seen.update(block=block, rows=rows, row_stride=row_stride, dim=d, stream=stream)
        ys[0][nu:] += parked["partial"]                        # the all-reduce, in stream order on the current stream
        return 0

    def hop(rank, cb, epilogue, y, rows=g.n_items, rr=r):
        item_c, user_c = pp[rank].item_op.c_struct(dim, False), pp[rank].user_op.c_struct(dim, False)
        return lib.lgc_hop_exchange(ctypes.byref(item_c), ctypes.byref(user_c), n, x.data_ptr(), x.stride(0), y.data_ptr(),
                                    y.stride(0), None if rr is None else rr.data_ptr(), 0 if rr is None else rr.stride(0),
                                    0.5, 0.25, dim, nu, rows, epilogue, cb, None, _native.stream_of(device))

    cb1, cb0 = _native.EXCHANGE_FN(exchange_rank1), _native.EXCHANGE_FN(exchange_rank0)
    assert hop(1, cb1, 0, ys[1]) == 0                           # same r and b on both ranks; rank 1 leaves the epilogue out
    assert hop(0, cb0, 1, ys[0]) == 0
    torch.cuda.synchronize()
    assert seen["block"] == ys[0][nu:].data_ptr() and seen["rows"] == g.n_items and seen["row_stride"] == dim and seen["dim"] == dim

    def rel(a, b):
        return ((a.double() - b.double()).norm() / b.double().norm()).item()

    lo, hi = pp[0].ranges[0]
    assert rel(ys[0][nu:], want[nu:]) <= 1e-5 and rel(ys[0][lo:hi], want[lo:hi]) <= 1e-6
    assert torch.isnan(ys[0][hi:nu]).all()                     # the other rank's users are not touched
    lo1, hi1 = pp[1].ranges[1]
    assert rel(ys[1][lo1:hi1], want[lo1:hi1]) <= 1e-6           # every rank applies the epilogue of its OWN user rows
    # rank 1's partial block carries no b * r term
    only_sum = pp[1].item_op.apply(x, torch.empty_like(x), a=0.5)
    assert torch.equal(parked["partial"], only_sum[nu:])
    # a failing callback aborts the hop with its code; a block beyond the table is refused
    bad = _native.EXCHANGE_FN(lambda *a: 7)
    assert hop(0, bad, 1, ys[0], rr=None) == 7
    assert hop(0, cb0, 1, ys[0], rows=g.n_items + 1, rr=None) == -1


class _ThreadComm:
    """Stand-in for partition.Comm when the ranks of a partition are THREADS of one process sharing one GPU and one
    stream: a collective = barrier, rank 0 sums the ranks' tensors on the device, barrier, every rank copies the sum.
    All launches go to the same stream in the order the barriers impose, so no device synchronisation is needed."""

    def __init__(self, rank, world, shared):
        self.rank, self.world, self.shared = rank, world, shared

    def _sum(self, t):
        sh = self.shared
        sh["slots"][self.rank] = t
        sh["barrier"].wait()
        if self.rank == 0:
            total = sh["slots"][0].clone()
            for other in sh["slots"][1:]:
                total += other
            sh["total"] = total
        sh["barrier"].wait()
        t.copy_(sh["total"])
        sh["barrier"].wait()

    def start(self, block):
        self._sum(block)
        return None

    def wait(self, handle):
        pass

    def reduce_now(self, t):
        self._sum(t)


def _run_thread_ranks(world, fn):
    import threading
    shared = {"slots": [None] * world, "barrier": threading.Barrier(world), "total": None}
    results, errors = [None] * world, []

    def body(rank):
        try:
            torch.cuda.set_device(0)
            results[rank] = fn(rank, _ThreadComm(rank, world, shared))
        except BaseException as exc:                      # a failing rank must not leave the others at a barrier
            errors.append((rank, exc))
            shared["barrier"].abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    real = [e for e in errors if not isinstance(e[1], threading.BrokenBarrierError)]
    assert not real, real
    return results


def test_eight_ranks_at_full_size_on_one_gpu(device):
    """BASELINE.json configs[3] and configs[4] with everything but RCCL: the 8-way partition of the full-size graph
    (1.64 M users x 54.6 k items, 20.3 M edges; D=64, K=3), its eight ranks as threads of this process on ONE GPU, every
    collective a device-side sum across the ranks -- the real kernels on the real slices (each rank's item step runs the
    2-band sweep of its own user range), against the single-GPU path: get_embedding on every rank's rows, then one
    training step (B=1024: seeded node, per-hop exchange, Adam over the rows a rank owns) against the single-GPU step."""
    import gnn_ecommerce_amd as lg
    from gnn_ecommerce_amd import synth
    from gnn_ecommerce_amd.partition import PartitionedPropagator, step_backward, step_forward
    from gnn_ecommerce_amd.optim import Adam as HipAdam
    world, dim, layers, batch, decay = 8, 64, 3, 1024, 1e-4
    g = synth.make_bipartite(**synth.CONFIG_COSMETICS, seed=0)
    ei, ew = g.coo(device)
    n, nu = g.num_nodes, g.n_users
    alphas = [0.25] * 4
    w0 = synth.xavier_table(n, dim, 0, device)
    gen = torch.Generator().manual_seed(3)
    users = torch.randint(0, nu, (batch,), generator=gen).to(device)
    pos = (torch.randint(0, g.n_items, (batch,), generator=gen) + nu).to(device)
    neg = (torch.randint(0, g.n_items, (batch,), generator=gen) + nu).to(device)
    # single GPU: forward, then the reference's step
    model = lg.LightGCN(n, dim, layers)
    model.load_state_dict({"alpha": model.alpha, "embedding.weight": w0.cpu()})
    model.to(device)
    with torch.no_grad():
        single = model.get_embedding(ei, ew)
    labels = torch.stack((torch.cat([users, users]), torch.cat([pos, neg])))
    out = model(ei, labels, ew)
    bpr = model.recommendation_loss(out[:batch], out[batch:], 0) * batch
    reg = lg.regularization_loss(model.embedding.weight, batch, users, pos, neg, decay)
    (bpr + reg).backward()
    ref_grad = model.embedding.weight.grad.clone()
    HipAdam([model.embedding.weight], lr=0.005).step()
    ref_w = model.embedding.weight.detach()
    pps = [PartitionedPropagator(ei, ew, nu, g.n_items, r, world) for r in range(world)]
    assert sum(pp.local_nnz for pp in pps) == g.nnz and all(pp.item_op.sweep_cols is not None for pp in pps)

    def rel(a, b):
        return ((a.double() - b.double()).norm() / b.double().norm()).item()

    def rank_body(rank, comm):
        pp = pps[rank]
        pp.comm = comm
        lo, hi = pp.ranges[rank]
        with torch.no_grad():
            emb = pp.propagate_sum(w0, alphas)
        res = {"own": rel(emb[lo:hi], single[lo:hi]), "items": rel(emb[nu:], single[nu:]),
               "worst": ((emb[lo:hi] - single[lo:hi]).norm(dim=1) / single[lo:hi].norm(dim=1)).max().item()}
        # the two halves of partition._PartitionedStep, called directly: autograd's engine runs every rank's backward on ONE
        # worker thread per device, where a rank waiting at a barrier would block the other seven
        with torch.no_grad():
            local, bpr_l, reg_u, reg_i, saved = step_forward(pp, w0, tuple(alphas), users, pos, neg, decay)
            grad = step_backward(pp, saved, tuple(alphas), None, decay / batch, zero_foreign=False)
            part = torch.stack([bpr_l, reg_u])
            comm.reduce_now(part)
        gbpr, greg = part[0], part[1] + reg_i
        res.update(bpr=abs(gbpr.item() - bpr.item()) / abs(bpr.item()), reg=abs(greg.item() - reg.item()) / abs(reg.item()),
                   g_own=rel(grad[lo:hi], ref_grad[lo:hi]), g_items=rel(grad[nu:], ref_grad[nu:]))
        wp = torch.nn.Parameter(w0.clone())
        wp.grad = grad
        HipAdam([wp], lr=0.005, row_ranges=pp.owned_row_ranges()).step()
        res.update(w_own=rel(wp.detach()[lo:hi] - w0[lo:hi], ref_w[lo:hi] - w0[lo:hi]),
                   w_items=rel(wp.detach()[nu:] - w0[nu:], ref_w[nu:] - w0[nu:]))
        return res

    results = _run_thread_ranks(world, rank_body)
    torch.cuda.synchronize()
    for rank, r in enumerate(results):
        print(f"rank {rank}: " + "  ".join(f"{k} {v:.1e}" for k, v in r.items()))
        assert r["own"] <= 1e-5 and r["items"] <= 1e-5 and r["worst"] <= 1e-5, (rank, r)
        assert r["bpr"] <= 1e-5 and r["reg"] <= 1e-5 and r["g_own"] <= 1e-5 and r["g_items"] <= 1e-5, (rank, r)
        assert r["w_own"] <= 1e-4 and r["w_items"] <= 1e-4, (rank, r)


def test_the_multi_gpu_training_harness_runs_two_ranks_on_one_gpu(device):
    """tools/train_dist.py (configs[4] on N GPUs: one process per GPU, PartitionedTrainer) rehearsed with two ranks on this
    one GPU over gloo, eager and recorded: one JSON line from rank 0, a finite decreasing-from-ln2 loss, the same loss from
    the recorded step."""
    import json
    import subprocess
    lines = []
    for extra in ([], ["--graphs"], ["--graphs", "full"]):       # (over gloo "full" falls back to the segments, on every rank)
        proc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_dist.py"), "--gpus", "2", "--backend", "gloo",
                               "--config", "small", "--steps", "6", "--warmup", "4", "--batch", "256"] + extra,
                              capture_output=True, text=True, timeout=600)
        assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
        found = [json.loads(ln) for ln in proc.stdout.splitlines() if ln.startswith("{")]
        assert len(found) == 1
        lines.append(found[0])
    for line in lines:
        assert line["n_gpus"] == 2 and line["unit"] == "steps/s" and line["value"] > 0
        assert 0.0 < line["loss"]["bpr"] < 0.6932 and line["loss"]["reg"] > 0
    assert abs(lines[0]["loss"]["total"] - lines[1]["loss"]["total"]) <= 1e-6 * lines[0]["loss"]["total"]
    assert lines[2]["loss"]["total"] == lines[1]["loss"]["total"] and "declined" in lines[2]["config"]["trainer"]


def test_recorded_forward_replays_and_falls_back_cleanly(device):
    """partition.RecordedForward on one rank (no process group): the forward as one replayed HIP graph gives the eager rows,
    follows in-place changes of its input, and a capture that fails half way (a fault injected while capturing) leaves the
    thread on a healthy stream with the eager forward."""
    from gnn_ecommerce_amd import synth
    from gnn_ecommerce_amd.partition import PartitionedPropagator, RecordedForward
    g = synth.make_bipartite(20000, 1500, 150000, seed=3)
    ei, ew = g.coo(device)
    x0 = synth.xavier_table(g.num_nodes, 64, 2, device)
    alphas = (0.4, 0.3, 0.2, 0.1)
    pp = PartitionedPropagator(ei, ew, g.n_users, g.n_items, 0, 1)
    want = pp.propagate_sum(x0, alphas).clone()
    rec = RecordedForward(pp, x0, alphas)
    assert rec.recorded and rec.error is None
    assert torch.equal(rec(), want)
    x0.mul_(2.0)
    doubled = rec().clone()
    x0.mul_(0.5)
    assert ((doubled - 2.0 * want).norm() / want.norm()).item() <= 1e-6 and torch.equal(rec(), want)
    real = pp.propagate_sum

    def flaky(*a, **k):
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("injected while capturing")
        return real(*a, **k)

    pp.propagate_sum = flaky
    broken = RecordedForward(pp, x0, alphas)
    assert not broken.recorded and "injected" in broken.error
    assert not torch.cuda.is_current_stream_capturing()
    assert torch.equal(broken(), want)                                  # eager, on a healthy stream
    pp.propagate_sum = real
    assert RecordedForward(pp, x0, alphas).recorded                     # and the next capture works again
