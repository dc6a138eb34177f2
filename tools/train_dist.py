#!/usr/bin/env python3
"""BASELINE.json configs[4] on N GPUs of one node: the mini-batch step of src/train_lightgcn.py:130-147 on the user-range
partition (partition.PartitionedPropagator + trainer.PartitionedTrainer), one process per GPU over RCCL.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        tools/train_dist.py --steps 30 [--graphs] [--dim 90 --layers 5]
    python tools/train_dist.py --gpus N ...          # starts the N ranks itself (child processes), like bench.py

Every rank draws the same seeded global batch (B triples, uniform), runs its part of the step (forward on the
partition, BPR + regulariser of its own triples, seeded backward with the per-hop item-block exchange, Adam over the rows
it owns) and the loop is timed like bench.py times its steps: W untimed steps, a barrier + synchronize, K steps, a barrier
+ synchronize, MAX over ranks; rank 0 prints ONE JSON line (steps/s of the whole job).  ``--backend gloo`` rehearses the
path with several ranks on ONE GPU (RCCL refuses that); the number then says nothing about the fabric.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="without a launcher: start this many ranks as child processes")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--config", choices=["cosmetics", "small"], default="cosmetics")
    ap.add_argument("--graphs", nargs="?", const="segments", default=None, choices=["segments", "full"],
                    help="record the step as HIP graphs: segments = one graph per stretch of work between two collectives; "
                         "full = the whole step as ONE graph with the collectives inside (nccl only; falls back to segments)")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--rehearse", action="store_true", help="ONE GPU: the same code path as a partition of ONE rank whose "
                    "collectives go to a real one-rank process group (LGCN_COMM_FORCE) -- a rehearsal of the launch, not of the fabric")
    return ap.parse_args()


def self_launch(n: int) -> int:
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]     # a rank ignores --gpus (WORLD_SIZE is set)
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1:
        sys.exit(self_launch(args.gpus))           # nothing has touched the GPU yet; the parent never exec()s
    import torch
    import torch.distributed as dist
    rank, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    grouped = world > 1 or args.rehearse
    if args.rehearse and world == 1:
        import socket
        os.environ["LGCN_COMM_FORCE"] = "1"                 # read when gnn_ecommerce_amd.partition is imported (below)
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    dev = torch.device(f"cuda:{local_rank % max(torch.cuda.device_count(), 1)}")
    torch.cuda.set_device(dev)
    if grouped:
        # (the communicator's printf banner goes to stderr: stdout carries the one JSON line)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(args.backend)
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    from gnn_ecommerce_amd import synth
    from gnn_ecommerce_amd.partition import PartitionedPropagator
    from gnn_ecommerce_amd.trainer import PartitionedTrainer

    g = synth.make_bipartite(**(synth.CONFIG_COSMETICS if args.config == "cosmetics" else synth.CONFIG_SMALL), seed=0)
    ei, ew = g.coo(dev)
    pp = PartitionedPropagator(ei, ew, g.n_users, g.n_items, rank, world)
    w = synth.xavier_table(g.num_nodes, args.dim, 0, dev)
    alphas = [1.0 / (args.layers + 1)] * (args.layers + 1)
    tr = PartitionedTrainer(pp, w, alphas, lr=0.005, decay=1e-4, batch=args.batch, graphs={None: False, "segments": True, "full": "full"}[args.graphs])
    gen = torch.Generator().manual_seed(0)             # the same stream of batches on every rank

    def step():
        u = torch.randint(0, g.n_users, (args.batch,), generator=gen).to(dev)
        p = (torch.randint(0, g.n_items, (args.batch,), generator=gen) + g.n_users).to(dev)
        n = (torch.randint(0, g.n_items, (args.batch,), generator=gen) + g.n_users).to(dev)
        return tr.step(u, p, n)

    def fence():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        stats = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stats = step()
    fence()
    elapsed = time.perf_counter() - t0
    if grouped:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
    bpr, reg, loss = stats.tolist()
    if rank == 0:
        print(json.dumps({
            "metric": "training steps/s (fwd+bwd+Adam)", "value": args.steps / elapsed, "unit": "steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[4]: BPR training step, B={args.batch}, emb_dim {args.dim}, {args.layers} LGConv layers, "
                                   f"{g.n_users} users x {g.n_items} items, {g.nnz} directed edges",
                       "parallelism": f"user-range x{world}, items replicated, [n_items, D] all-reduce per hop (the forward's last "
                                      f"layer: [2B, D]) over {args.backend}" + ("" if args.backend == "nccl" or world == 1
                                                                              else " (rehearsal, not RCCL)"),
                       "trainer": "PartitionedTrainer, " + ({"segments": "recorded as HIP graphs between the collectives", "full": "recorded as "
                                   f"{tr.graph_launches} HIP graph(s), collectives inside" + (f" (one graph declined: {tr.full_error})" if tr.full_error else "")}
                                  .get(args.graphs, "eager")),
                       "own_users": [pp.u0, pp.u1]},
            "loss": {"bpr": bpr, "reg": reg, "total": loss}}), flush=True)
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
