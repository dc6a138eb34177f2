#!/usr/bin/env python3
"""Compute-only time of ONE rank of an N-way partitioned propagate on a single GPU (the all-reduce is
stubbed out), to separate local work from the exchange when judging multi-GPU scaling.
    python tools/rank_compute.py [world] [ranks...]"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from gnn_ecommerce_amd import synth
from gnn_ecommerce_amd.partition import PartitionedPropagator

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ranks = [int(a) for a in sys.argv[2:]] or [0, world // 2, world - 1]
dist.all_reduce = lambda *a, **k: None          # measurement stub: local work only
dev = torch.device("cuda:0")
g = synth.make_bipartite(**synth.CONFIG_COSMETICS, seed=0)
ei, ew = g.coo(dev)
x0 = synth.xavier_table(g.num_nodes, 64, 0, dev)
alphas = (0.25, 0.25, 0.25, 0.25)
for r in ranks:
    pp = PartitionedPropagator(ei, ew, g.n_users, g.n_items, r, world)
    for _ in range(3):
        pp.propagate_sum(x0, alphas)
    ts, host = [], []
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pp.propagate_sum(x0, alphas)
        t1 = time.perf_counter()                       # everything enqueued: the host's share of a hop
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 3 * 1e6)
        host.append((t1 - t0) / 3 * 1e6)
    print(f"world {world} rank {r}: users [{pp.u0},{pp.u1}) local nnz {pp.local_nnz}  compute {statistics.median(ts):.1f} us/hop"
          f"  host enqueue {statistics.median(host):.1f} us/hop")
    del pp
    if os.environ.get("RANK_COMPUTE_GRAPH") == "1":
        # the same forward recorded as ONE HIP graph (no collective inside: they are stubbed) and replayed back to back, against
        # the eager forward issued back to back: what recording the compute segments of a forward could buy (VERDICT r3 #1c)
        pp = PartitionedPropagator(ei, ew, g.n_users, g.n_items, r, world)
        for _ in range(3):
            pp.propagate_sum(x0, alphas)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            pp.propagate_sum(x0, alphas)
        res = {}
        for name, fn in (("eager", lambda: pp.propagate_sum(x0, alphas)), ("recorded", graph.replay)) * 3:
            torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                fn()
            e.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(s.elapsed_time(e) / 30 * 1e3)
        print(f"world {world} rank {r}: ten forwards back to back, us/hop: " + "; ".join(f"{k} {' / '.join(f'{v:.1f}' for v in vs)}" for k, vs in res.items()))
        del pp, graph
