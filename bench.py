#!/usr/bin/env python3
"""Benchmark of the LightGCN propagation hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One *step* = one ``get_embedding``: LAYERS sparse hops over the whole graph with the layer sum
fused (the work src/lightgcn.py:91-99 does per mini-batch and per serving request).
Workload at every N: BASELINE.json configs[1] -- synthetic cosmetics-scale graph, 1,639,358 users
x 54,571 items, 10,157,408 pairs = 20,314,816 directed edges, 3 layers, emb_dim 64, fp32.
N > 1 (launched by torch.distributed.run, one rank per GPU): the SAME graph, users partitioned by
nnz-balanced ranges, items replicated, one RCCL all-reduce of the [n_items, D] block per hop
(strong scaling).

metric  = edges propagated per second per LGConv layer = nnz * LAYERS / t_step, whole job.
roofline = algorithmic bytes of one hop (B_min, SURVEY.md 8d: nnz*8 + (N+1)*4 + 2*N*D*4) divided
           by the mean duration of one hop's launches, measured with events on the launch stream
           inside the timed region, against 8 TB/s HBM3E.
cpu_baseline = the reference-semantics CPU path (oracle/, a port of the reference's PyG route:
           unsorted COO, per-layer gcn_norm, gather -> scale -> index_add_) timed on this host.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# dmabuf IPC for RCCL / multi-process GPU work on this pool: must be in the environment before HIP initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist

LAYERS, DIM, SEED = 3, 64, 0
# Thread sweep of the CPU baseline on the GPU box's host (2 x EPYC 9575F, 256 hardware threads visible; round 1,
# gpurun_out/cpu_threads.log): 8 -> 7.5, 16/32 -> 7.9, 64 -> 8.5, 128 -> 6.7, 256 -> 3.5 M edges/s per layer.
# The route is memory-bound; 64 threads is the fastest setting, so that is the default (capped by the cores present).
CPU_THREADS_DEFAULT = 64
HBM_PEAK = 8.0e12  # B/s, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--layers", type=int, default=LAYERS)
    ap.add_argument("--dim", type=int, default=DIM)
    ap.add_argument("--config", choices=["cosmetics", "small"], default="cosmetics")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-deep", action="store_true", help="skip the configs[2] (D=90, K=5) measurement that rides "
                    "along on the default single-GPU run")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step measurements (configs[4]'s step on one "
                    "GPU at D=64/K=3 and D=90/K=5, and its one-step CPU baseline) of the default single-GPU run")
    ap.add_argument("--no-rank-local", action="store_true", help="skip the rank-local measurement (one rank of an 8-way "
                    "partition on this GPU, collectives stubbed)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads for the CPU baseline (0 = auto)")
    ap.add_argument("--rehearse-partition", action="store_true", help="ONE GPU: run the N > 1 code path (process group with "
                    "device_id, PartitionedPropagator, RecordedForward, barrier fences, MAX over ranks) as a partition of one rank "
                    "whose collectives go to a real one-rank nccl group (LGCN_COMM_FORCE) -- a rehearsal of the multi-GPU launch, "
                    "not a measurement of the fabric")
    ap.add_argument("--no-graph", action="store_true", help="N > 1: the eager forward instead of the forward recorded as one HIP "
                    "graph with its collectives (partition.RecordedForward)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only "
                    "to rehearse the N > 1 path with several ranks on ONE GPU)")
    return ap.parse_args()


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            names = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")]
        return f"{names[0]} ({len(names)} hardware threads visible)" if names else "unknown CPU"
    except OSError:
        return "unknown CPU"


def cpu_baseline(graph, layers, dim, threads=0):
    """Reference-semantics CPU path on a bounded sample (SURVEY.md 8d protocol): the full K-layer propagate of the
    same graph, one warm-up layer then best of 3 passes (about 25 s of host time at full size)."""
    from oracle import lightgcn_oracle as oracle
    from gnn_ecommerce_amd import synth
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    cores = threads if threads > 0 else min(avail, CPU_THREADS_DEFAULT)
    torch.set_num_threads(cores)
    ei, ew = graph.coo()
    w0 = synth.xavier_table(graph.num_nodes, dim, SEED)
    times = []
    with torch.no_grad():
        oracle.lgconv(w0, ei, ew)                     # warm-up (allocator, thread pool)
        for _ in range(3):
            t0 = time.perf_counter()
            oracle.get_embedding(w0, oracle.default_alpha(layers), ei, ew, layers)
            times.append(time.perf_counter() - t0)
    dt = min(times)
    out = {"value": graph.nnz * layers / dt, "unit": "edges/s", "cores": cores, "kind": "port",
           "sample": f"full {layers}-layer get_embedding of the same graph ({graph.nnz} edges, D={dim}), torch "
                     f"{torch.__version__} CPU fp32 on {cpu_model()}, {cores} threads, best of 3 passes "
                     f"({', '.join('%.2f' % t for t in times)} s) after a 1-layer warm-up"}
    # a STRONGER comparator that is not the reference's route -- torch's own CSR SpMM (MKL)
    # on the same normalised values, graph conversion excluded, K hops after one warm-up hop.
    with torch.no_grad():
        val = oracle.gcn_norm(ei, ew, graph.num_nodes)
        a = torch.sparse_coo_tensor(torch.stack((ei[1], ei[0])), val, (graph.num_nodes,) * 2).coalesce().to_sparse_csr()
        x = a @ w0
        t0 = time.perf_counter()
        for _ in range(layers):
            x = a @ x
        dt2 = time.perf_counter() - t0
    out["stronger_comparator"] = {"what": "torch CSR SpMM (MKL) on precomputed values; NOT the reference's path",
                                  "value": graph.nnz * layers / dt2, "unit": "edges/s", "cores": cores,
                                  "seconds": round(dt2, 3)}
    return out


def training_step_bench(graph, ei, ew, dim, layers, steps, warmup, batch=1024, lr=0.005, decay=1e-4):
    """BASELINE.json configs[4]'s step on ONE GPU: the loop body of src/train_lightgcn.py:130-147 (zero_grad -> labels ->
    forward -> bpr * size + reg -> backward -> Adam) with B = 1024 on the full graph, (u, i+, i-) drawn uniformly with a
    seeded generator (the pandas sampler is out of scope).  The calls are the reference's own --
    ``model(edge_index, labels, edge_weight)``, ``recommendation_loss``, ``regularization_loss``, ``loss.backward()``,
    ``optimizer.step()``, three ``.item()`` -- on the drop-in classes."""
    import gnn_ecommerce_amd as lg
    from gnn_ecommerce_amd.optim import Adam as HipAdam
    dev = ei.device
    torch.manual_seed(SEED)
    model = lg.LightGCN(graph.num_nodes, dim, layers).to(dev)
    opt = HipAdam(model.parameters(), lr)
    gen = torch.Generator().manual_seed(SEED)

    def step():
        opt.zero_grad()
        u = torch.randint(0, graph.n_users, (batch,), generator=gen).to(dev)
        p = (torch.randint(0, graph.n_items, (batch,), generator=gen) + graph.n_users).to(dev)
        n = (torch.randint(0, graph.n_items, (batch,), generator=gen) + graph.n_users).to(dev)
        labels = torch.stack((torch.cat([u, u]), torch.cat([p, n])))
        out = model(ei, labels, ew)
        bpr = model.recommendation_loss(out[:batch], out[batch:], 0) * batch
        reg = lg.regularization_loss(model.embedding.weight, batch, u, p, n, decay)
        loss = bpr + reg
        loss.backward()
        opt.step()
        return bpr.item(), reg.item(), loss.item()

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        vals = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    del model, opt
    return {"value": 1.0 / dt, "unit": "steps/s", "ms_per_step": dt * 1e3, "steps": steps, "warmup": warmup, "loss": vals[2]}


def training_step_cpu(graph, dim, layers, threads=0, batch=1024, lr=0.005, decay=1e-4):
    """The same step on this host through the oracle's restatement of the reference route (unsorted COO, per-layer
    gcn_norm, gather -> scale -> index_add_, autograd, dense torch Adam): ONE step, untimed allocations included."""
    from oracle import lightgcn_oracle as oracle
    from gnn_ecommerce_amd import synth
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    cores = threads if threads > 0 else min(avail, CPU_THREADS_DEFAULT)
    torch.set_num_threads(cores)
    ei, ew = graph.coo()
    w = torch.nn.Parameter(synth.xavier_table(graph.num_nodes, dim, SEED))
    opt = torch.optim.Adam([w], lr)
    gen = torch.Generator().manual_seed(SEED)
    u = torch.randint(0, graph.n_users, (batch,), generator=gen)
    p = torch.randint(0, graph.n_items, (batch,), generator=gen) + graph.n_users
    n = torch.randint(0, graph.n_items, (batch,), generator=gen) + graph.n_users
    t0 = time.perf_counter()
    opt.zero_grad()
    loss = oracle.train_step_loss(w, oracle.default_alpha(layers), ei, ew, u, p, n, layers, decay)[3]
    loss.backward()
    opt.step()
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"ONE step (forward + backward + dense Adam, B={batch}, D={dim}, K={layers}) of the same graph on {cpu_model()}, "
                      f"{cores} threads: {dt:.1f} s"}


def rank_local_bench(graph, ei, ew, dim, layers, world, ranks, single_hop_s):
    """What ONE rank of a ``world``-way partition computes per hop, measured on this one GPU with every collective
    stubbed out (local work only: what RCCL adds can only be measured on a multi-GPU node) -- and the ceiling it puts on
    the speed-up over one GPU.  Also the rank-local training step (partition.partitioned_bpr_loss + Adam over the rows
    the rank owns), same stub."""
    import torch.distributed as dist
    from gnn_ecommerce_amd import partition, synth
    from gnn_ecommerce_amd.optim import Adam as HipAdam
    real_all_reduce = dist.all_reduce
    dist.all_reduce = lambda *a, **k: None
    try:
        dev = ei.device
        alphas = tuple([1.0 / (layers + 1)] * (layers + 1))
        x0 = synth.xavier_table(graph.num_nodes, dim, SEED, dev)
        hop_us, hop_mean_us, train_ms, train_graph_ms = [], [], [], []
        for r in ranks:
            pp = partition.PartitionedPropagator(ei, ew, graph.n_users, graph.n_items, r, world)
            for _ in range(3):
                pp.propagate_sum(x0, alphas)
            torch.cuda.synchronize()
            reps = 20
            times = []
            for _ in range(5):                      # five batches of ten back-to-back forwards (like the headline's timed
                t0 = time.perf_counter()            # loop), the MEDIAN batch is reported: one allocator stall (a fresh
                for _ in range(10):                 # 434 MB block: tens of ms) otherwise shows up as +400 us per hop
                    pp.propagate_sum(x0, alphas)    # (seen once in four runs)
                torch.cuda.synchronize()
                times.append((time.perf_counter() - t0) / 10)
            hop_us.append(statistics.median(times) / layers * 1e6)
            hop_mean_us.append(sum(times) / len(times) / layers * 1e6)
            w = torch.nn.Parameter(x0.clone())
            opt = HipAdam([w], 0.005, row_ranges=pp.owned_row_ranges())
            gen = torch.Generator().manual_seed(SEED)

            def step():
                opt.zero_grad()
                u = torch.randint(0, graph.n_users, (1024,), generator=gen).to(dev)
                p = (torch.randint(0, graph.n_items, (1024,), generator=gen) + graph.n_users).to(dev)
                n = (torch.randint(0, graph.n_items, (1024,), generator=gen) + graph.n_users).to(dev)
                local, gbpr, greg = partition.partitioned_bpr_loss(pp, w, alphas, u, p, n, 1e-4, zero_foreign_rows=False)
                local.backward()
                opt.step()
                return gbpr.item(), greg.item()

            for _ in range(3):
                step()
            torch.cuda.synchronize()
            times = []
            for _ in range(reps):
                t0 = time.perf_counter()
                step()                               # ends in two .item() syncs
                times.append(time.perf_counter() - t0)
            train_ms.append(statistics.median(times) * 1e3)
            # the same step as trainer.PartitionedTrainer records it: ONE HIP graph, the (stubbed) collectives inside
            from gnn_ecommerce_amd.trainer import PartitionedTrainer
            tr = PartitionedTrainer(pp, x0.clone(), alphas, lr=0.005, decay=1e-4, batch=1024, graphs="full")
            ids = [(torch.randint(0, graph.n_users, (1024,), generator=gen).to(dev),
                    (torch.randint(0, graph.n_items, (1024,), generator=gen) + graph.n_users).to(dev),
                    (torch.randint(0, graph.n_items, (1024,), generator=gen) + graph.n_users).to(dev)) for _ in range(4)]
            for k in range(4):
                tr.step(*ids[k % 4])
            torch.cuda.synchronize()
            times = []
            for k in range(reps):
                t0 = time.perf_counter()
                tr.step(*ids[k % 4]).tolist()        # one host sync per step
                times.append(time.perf_counter() - t0)
            train_graph_ms.append(statistics.median(times) * 1e3)
            del pp, w, opt, tr
        worst = max(hop_us)
        return {"world": world, "ranks_measured": list(ranks), "us_per_hop": worst, "us_per_hop_by_rank": hop_us,
                "us_per_hop_mean_by_rank": hop_mean_us, "ceiling_x": single_hop_s * 1e6 / worst,
                "train_ms_per_step": max(train_graph_ms), "train_ms_per_step_autograd": max(train_ms),
                "what": "local work of one rank per hop / per training step on this one GPU, every all-reduce stubbed out; "
                        "us_per_hop: median of five batches of ten back-to-back forwards (mean beside it); training: medians of 20 "
                        "individually timed steps; ceiling_x = this run's "
                        "single-GPU hop time / us_per_hop (no exchange cost in it); train_ms_per_step = "
                        "trainer.PartitionedTrainer with the whole step recorded as ONE HIP graph (graphs='full': over RCCL the "
                        "collectives are captured with it; against a real one-rank nccl group 0.77 ms vs 1.01 recorded between the "
                        "collectives vs 1.02-1.07 eager, profiles/r04k), "
                        "train_ms_per_step_autograd = partitioned_bpr_loss + backward() + optim.Adam(row_ranges)"}
    finally:
        dist.all_reduce = real_all_reduce


import contextlib


@contextlib.contextmanager
def c_stdout_to_stderr():
    """RCCL prints a version banner with printf when a communicator is created; this process's stdout carries ONE JSON line.
    While the block runs, file descriptor 1 points at stderr; the C library's buffer is flushed before it is restored."""
    import ctypes
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:                                     # noqa: BLE001
            pass
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def exchange_machinery_bench(graph, ei, ew, dim, layers, world, rank):
    """What torch's collectives cost a rank's hop BEFORE any byte moves, measured on this one GPU: RCCL refuses two ranks on
    one device, but a ONE-rank nccl group is legal and its all-reduce is an identity -- so rank ``rank`` of a ``world``-way
    partition runs its forward with every per-hop all-reduce issued to such a group (event on the launch stream, RCCL's
    stream waits, the collective, event, the launch stream waits), eagerly and as partition.RecordedForward replays it (one
    HIP graph, the collectives inside).  Never fatal: any failure comes back as {"error": ...}."""
    import datetime
    import socket
    import torch.distributed as dist
    from gnn_ecommerce_amd import partition, synth
    made_group = False
    try:
        dev = ei.device
        if not dist.is_initialized():
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                port = s.getsockname()[1]
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev,
                                    timeout=datetime.timedelta(seconds=60))
            made_group = True
        alphas = tuple([1.0 / (layers + 1)] * (layers + 1))
        x0 = synth.xavier_table(graph.num_nodes, dim, SEED, dev)
        pp = partition.PartitionedPropagator(ei, ew, graph.n_users, graph.n_items, rank, world)
        pp.comm = partition.Comm(world, None)             # active: its all-reduces go to the default (one-rank) group
        rec = partition.RecordedForward(pp, x0, alphas)
        out = {}
        for name, fn in (("eager", lambda: pp.propagate_sum(x0, alphas)), ("recorded", rec)):
            times = []
            for rnd in range(4):
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10):
                    fn()
                e.record()
                torch.cuda.synchronize()
                if rnd:
                    times.append(s.elapsed_time(e) / (10 * layers) * 1e3)
            out[name] = statistics.median(times)
        return {"world": world, "rank": rank, "us_per_hop_eager_collectives": out["eager"],
                "us_per_hop_recorded_with_collectives": out["recorded"], "recorded": rec.recorded, "capture_error": rec.error,
                "what": "the same rank's forward with its per-hop all-reduce issued to a REAL one-rank nccl process group "
                        "(an identity: the collective's machinery alone, no byte moves) -- eagerly, and captured once with "
                        "its collectives into one HIP graph and replayed (partition.RecordedForward, what bench.py --gpus N "
                        "times); medians of three batches of ten back-to-back forwards; compare rank_local.us_per_hop "
                        "(collectives stubbed out)"}
    except Exception as exc:                                  # noqa: BLE001 -- a rider: never fatal for the bench line
        return {"error": f"{type(exc).__name__}: {exc}"[:300]}
    finally:
        if made_group:
            try:
                dist.destroy_process_group()
            except Exception:                                 # noqa: BLE001
                pass


def workload_name(args, world: int) -> str:
    """Which BASELINE.json configuration the arguments describe."""
    if args.config != "cosmetics":
        return "configs[0]-sized plumbing graph"
    if world > 1:
        return "configs[3]"
    return "configs[2]" if (args.layers, args.dim) == (5, 90) else "configs[1]"


def kernel_source_hash() -> str:
    """sha256 over the files that decide which bytes a hop moves: the kernels and the work plans."""
    import hashlib
    h = hashlib.sha256()
    for rel in ("gnn-ecommerce_amd/csrc/lgconv_hip.hip", "gnn-ecommerce_amd/graph.py", "gnn-ecommerce_amd/propagate.py"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes through
    torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1), relay their output -- rank 0 prints the
    JSON line -- and hand back their exit code.  The parent never initialises HIP and never exec()s."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(self_launch(args.gpus))   # plain `python bench.py --gpus N`: nothing has touched the GPU yet
        args.gpus = world
    dev = torch.device(f"cuda:{local_rank % max(torch.cuda.device_count(), 1)}")
    torch.cuda.set_device(dev)
    partitioned = world > 1 or args.rehearse_partition      # the multi-rank code path (with ONE rank when rehearsed)
    if args.rehearse_partition and world == 1:
        import socket
        os.environ["LGCN_COMM_FORCE"] = "1"                 # read when gnn_ecommerce_amd.partition is imported (below)
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if partitioned:
        with c_stdout_to_stderr():                         # the communicator's banner does not belong on stdout
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(args.backend)

    import gnn_ecommerce_amd as lg
    from gnn_ecommerce_amd import propagate, synth

    cfg = synth.CONFIG_COSMETICS if args.config == "cosmetics" else synth.CONFIG_SMALL
    t0 = time.perf_counter()
    graph = synth.make_bipartite(**cfg, seed=SEED)
    t_gen = time.perf_counter() - t0
    n, nnz = graph.num_nodes, graph.nnz
    ei, ew = graph.coo(dev)

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if not partitioned:
        pg = lg.PropGraph(ei, ew, n)
        torch.cuda.synchronize()
        t_build = time.perf_counter() - t0
        # the work plans are built lazily by the first hop: force them here so that their cost is reported instead of
        # hiding in the warm-up (tile classes on the device; the item half's band-sweep plan: D2H copy, host planner, upload)
        t0 = time.perf_counter()
        pg.prepare(args.dim)
        t_plan = time.perf_counter() - t0
        make_step = lambda x0, alphas: (lambda: propagate.propagate_sum(x0, pg, alphas))
        parallelism = "single"
    else:
        from gnn_ecommerce_amd.partition import PartitionedPropagator, RecordedForward
        pp = PartitionedPropagator(ei, ew, graph.n_users, graph.n_items, rank, world)
        torch.cuda.synchronize()
        t_build, t_plan = time.perf_counter() - t0, None
        recorded_steps = []
        # the forward of one fixed table recorded once -- before the timed region -- as ONE HIP graph, the per-hop all-reduces
        # and their waits included as graph edges (each eager collective costs a rank two cross-stream hand-offs, +23 us on a
        # 94 us hop: profiles/r04k); every rank falls back to the eager forward together if any rank cannot capture
        if args.no_graph:
            make_step = lambda x0, alphas: (lambda: pp.propagate_sum(x0, alphas))
        else:
            def make_step(x0, alphas):
                rec = RecordedForward(pp, x0, alphas)
                recorded_steps.append(rec)
                return rec
        parallelism = (f"user-range x{world}, items replicated, all-reduce [n_items,D]/hop over {args.backend}"
                       + ("" if args.backend == "nccl" else " (rehearsal, not RCCL)")
                       + (" -- ONE rank going through the real process group: a rehearsal of the multi-GPU launch, the "
                          "collectives are identities" if args.rehearse_partition and world == 1 else ""))
    keep_coo = (ei, ew)
    del ei, ew

    def fence():
        if partitioned:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(dim, layers, steps, warmup):
        """W untimed steps, then exactly K timed steps between two fences; per-hop events on the launch stream."""
        alphas = tuple([1.0 / (layers + 1)] * (layers + 1))
        x0 = synth.xavier_table(n, dim, SEED, dev)
        step = make_step(x0, alphas)
        for _ in range(warmup):
            step()
        propagate.HOP_EVENT_LOG = []
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
        hop_events, propagate.HOP_EVENT_LOG = propagate.HOP_EVENT_LOG, None
        if partitioned:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = tmax.item()
        hop_ms = [s.elapsed_time(e) for s, e in hop_events]
        hop_mean_s = sum(hop_ms) / max(len(hop_ms), 1) * 1e-3
        if not hop_ms:          # a replayed graph carries no per-hop events: the timed region's wall time per hop
            hop_ms, hop_mean_s = [0.0] * (steps * layers), elapsed / (steps * layers)
        bmin = synth.algorithmic_bytes_per_layer(n, nnz, dim)
        achieved = bmin / hop_mean_s if hop_mean_s > 0 else 0.0
        return {"elapsed": elapsed, "hop_mean_s": hop_mean_s, "launches": len(hop_ms), "bmin": bmin, "achieved": achieved}

    def traffic_of(dim):
        """HBM-side bytes per hop from separate rocprofv3 PMC passes (profiles/collect.sh): quoted only while the kernel
        source they were measured on is the one running now, otherwise null.  Not measured by THIS process."""
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if not partitioned and os.path.isfile(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("kernel_source_sha256") == kernel_source_hash():
                return tj.get(f"{args.config}_d{dim}"), tj.get("source", ""), tj.get("measured_on", "unrecorded")
        return None, "no PMC measurement for this kernel source (run profiles/collect.sh)", None

    def roofline_of(m, dim, kernel_note):
        traffic, note, where = traffic_of(dim)
        return {"bound": "hbm", "achieved": m["achieved"] / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": m["achieved"] / HBM_PEAK, "traffic": traffic, "traffic_source": note,
                "traffic_measured_on": where, "kernel": kernel_note,
                "algorithmic_bytes_per_launch": m["bmin"], "launch_ms": m["hop_mean_s"] * 1e3,
                "launches_timed": m["launches"]}

    main_m = measure(args.dim, args.layers, args.steps, args.warmup)
    # configs[2] (same graph, 5 layers, emb_dim 90) rides along on the default single-GPU run: it is the configuration
    # furthest below its roofline, so the driver's own clock sees it too
    deep_m = None
    if not partitioned and args.config == "cosmetics" and (args.layers, args.dim) == (LAYERS, DIM) and not args.no_deep:
        t0 = time.perf_counter()
        pg.prepare(90)
        deep_plan = time.perf_counter() - t0
        deep_steps = max(5, args.steps // 2)
        deep_m = measure(90, 5, deep_steps, args.warmup)
    default_run = not partitioned and args.config == "cosmetics" and (args.layers, args.dim) == (LAYERS, DIM)
    train_lines, rank_local = [], None
    if default_run and not args.no_train:
        for t_dim, t_layers, t_steps, with_cpu in ((64, 3, max(20, args.steps), True), (90, 5, max(10, args.steps // 2), False)):
            t = training_step_bench(graph, keep_coo[0], keep_coo[1], t_dim, t_layers, t_steps, args.warmup)
            entry = {"workload": f"configs[4]'s step on ONE GPU: B=1024, emb_dim {t_dim}, {t_layers} LGConv layers, forward + backward "
                                 "(seeded) + Adam (optim.Adam, one pass) on the full graph, uniform seeded triples"
                                 + ("" if (t_dim, t_layers) == (64, 3) else " -- the reference's own hyper-parameters "
                                    "(src/train_lightgcn.py:47-53)"),
                     "dtype": "f32", **t}
            if with_cpu and not args.no_cpu_baseline:
                entry["cpu_baseline"] = training_step_cpu(graph, t_dim, t_layers, args.cpu_threads)
            train_lines.append(entry)
    if default_run and not args.no_rank_local:
        rank_local = rank_local_bench(graph, keep_coo[0], keep_coo[1], args.dim, args.layers, 8, (0, 5), main_m["hop_mean_s"])
        with c_stdout_to_stderr():
            rank_local["exchange_machinery"] = exchange_machinery_bench(graph, keep_coo[0], keep_coo[1], args.dim, args.layers, 8, 5)
    del keep_coo
    if rank == 0:
        elapsed = main_m["elapsed"]
        line = {
            "metric": "edges propagated/sec per LGConv layer",
            "value": nnz * args.layers * args.steps / elapsed,
            "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{workload_name(args, world)}: {graph.n_users} users x {graph.n_items} items, "
                                   f"{nnz} directed edges, {args.layers} LGConv layers, emb_dim {args.dim}, "
                                   "get_embedding (K hops + fused layer sum)",
                       "parallelism": parallelism + ("" if not partitioned else (
                           ", forward recorded as one HIP graph (collectives included)" if recorded_steps and recorded_steps[0].recorded
                           else ", eager forward" + (f" (capture declined: {recorded_steps[0].error})" if recorded_steps and recorded_steps[0].error else ""))),
                       "seed": SEED,
                       "graph_build_s": round(t_build, 3),
                       "plan_build_s": None if t_plan is None else round(t_plan, 3),
                       "synth_gen_s": round(t_gen, 1)},
            "roofline": roofline_of(main_m, args.dim,
                                    "one hop = one LGConv layer: item step k_sweep + k_sweep_combine, user step "
                                    "k_apply_fused (one launch: tile classes 8|16|32 through the DPP row kernel + chunked rows > 32 entries)"),
        }
        if deep_m is not None:
            line["other_configs"] = [{
                "workload": f"configs[2]: same graph, 5 LGConv layers, emb_dim 90, get_embedding",
                "value": nnz * 5 * deep_steps / deep_m["elapsed"], "unit": "edges/s", "steps": deep_steps,
                "warmup": args.warmup, "ms_per_step": deep_m["elapsed"] / deep_steps * 1e3, "dtype": "f32",
                "plan_build_s": round(deep_plan, 3),
                "roofline": roofline_of(deep_m, 90, "one hop: k_sweep_wide + k_sweep_combine, k_apply_fused<2> (two DPP rows "
                                                    "per table row, 96-float internal row stride)")}]
        if train_lines:
            line.setdefault("other_configs", []).extend(train_lines)
        if rank_local is not None:
            line["rank_local"] = rank_local
        if not partitioned and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(graph, args.layers, args.dim, args.cpu_threads)
        print(json.dumps(line), flush=True)
    if partitioned:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
