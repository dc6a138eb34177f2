#!/usr/bin/env python3
"""BASELINE.json configs[4] on one GPU: the mini-batch step of src/train_lightgcn.py:130-147
(zero_grad -> labels -> forward -> bpr*size + reg -> backward -> Adam) on the cosmetics-scale graph.
(u, i+, i-) are drawn uniformly with a seeded generator (the pandas sampler is out of scope)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnn_ecommerce_amd as lg
from gnn_ecommerce_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--dim", type=int, default=64); ap.add_argument("--layers", type=int, default=3)
ap.add_argument("--steps", type=int, default=10); ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--sampler", choices=["uniform", "device", "python"], default="uniform",
                help="uniform: seeded uniform triples (SURVEY 8d); device: TripleSampler (batch_loader contract on "
                     "the GPU); python: the oracle restatement of the reference's batch_loader on the host")
ap.add_argument("--adam", choices=["default", "fused", "hip"], default="default",
                help="torch.optim.Adam(fused=False|True), or hip: gnn_ecommerce_amd.optim.Adam (one pass, lgc_adam_step)")
ap.add_argument("--reg", choices=["caller", "routed"], default="routed",
                help="caller: upstream's regularization_loss on plain torch ops (three dense [N, D] gradients); routed: "
                     "gnn_ecommerce_amd.regularization_loss (same value, gradient added inside the scoring node)")
ap.add_argument("--cpu-reference", type=int, default=0, metavar="THREADS",
                help="also time the same step on the host with THREADS threads through the oracle's restatement of "
                     "the reference route (comparison only: BASELINE.json configs[4] quotes steps/s against it); "
                     "one warm-up step, one timed step")
ap.add_argument("--world", type=int, default=1, help="> 1: the rank-local step of a WORLD-way partition on this one GPU, "
                "every collective stubbed out (partition.partitioned_bpr_loss: seeded node, optim.Adam over the rows the "
                "rank owns) -- local work only, what the exchange adds can only be measured on a multi-GPU node")
ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--dense-partition", action="store_true", help="with --world: the dense partitioned backward (round 3)")
ap.add_argument("--trainer", choices=["autograd", "eager", "graphs", "full"], default="autograd",
                help="with --world: autograd = partitioned_bpr_loss + backward() + optim.Adam(row_ranges); eager / graphs = "
                     "trainer.PartitionedTrainer (the same launches without autograd / recorded once as HIP graphs between "
                     "the collectives and replayed)")
args = ap.parse_args()
dev = torch.device("cuda:0")
g = synth.make_bipartite(**synth.CONFIG_COSMETICS, seed=0)
ei, ew = g.coo(dev)
if args.world > 1 or args.trainer != "autograd":     # --world 1 --trainer graphs: the single-GPU step as recorded launches
    import torch.distributed as dist
    from gnn_ecommerce_amd import partition
    from gnn_ecommerce_amd.optim import Adam as HipAdam
    dist.all_reduce = lambda *a, **k: None                    # measurement stub: local work only
    partition.SEEDED_STEP = not args.dense_partition
    pp = partition.PartitionedPropagator(ei, ew, g.n_users, g.n_items, args.rank, args.world)
    w = torch.nn.Parameter(synth.xavier_table(g.num_nodes, args.dim, 0, dev))
    alphas = [1.0 / (args.layers + 1)] * (args.layers + 1)
    popt = (HipAdam([w], 0.005, row_ranges=pp.owned_row_ranges()) if args.adam == "hip"
            else torch.optim.Adam([w], 0.005, fused=(args.adam == "fused")))
    pgen = torch.Generator().manual_seed(0)
    if args.trainer != "autograd":
        from gnn_ecommerce_amd.trainer import PartitionedTrainer
        tr = PartitionedTrainer(pp, w.detach(), alphas, lr=0.005, decay=1e-4, batch=args.batch, graphs={"graphs": True, "full": "full"}.get(args.trainer, False))
    def pstep():
        if args.trainer != "autograd":
            u = torch.randint(0, g.n_users, (args.batch,), generator=pgen).to(dev)
            p = (torch.randint(0, g.n_items, (args.batch,), generator=pgen) + g.n_users).to(dev)
            n = (torch.randint(0, g.n_items, (args.batch,), generator=pgen) + g.n_users).to(dev)
            st = tr.step(u, p, n).tolist()                    # one host sync per step (bpr, reg, loss)
            return st[0], st[1]
        popt.zero_grad()
        u = torch.randint(0, g.n_users, (args.batch,), generator=pgen).to(dev)
        p = (torch.randint(0, g.n_items, (args.batch,), generator=pgen) + g.n_users).to(dev)
        n = (torch.randint(0, g.n_items, (args.batch,), generator=pgen) + g.n_users).to(dev)
        local, gbpr, greg = partition.partitioned_bpr_loss(pp, w, alphas, u, p, n, 1e-4, zero_foreign_rows=args.adam != "hip")
        local.backward()
        popt.step()
        return gbpr.item(), greg.item()                       # the host syncs of train_lightgcn.py:149-151
    for _ in range(3): vals = pstep()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps): vals = pstep()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"metric": "rank-local training step, collectives stubbed (B=%d)" % args.batch, "world": args.world,
                      "rank": args.rank, "users": [pp.u0, pp.u1], "ms_per_step": dt * 1e3, "dim": args.dim, "layers": args.layers,
                      "adam": args.adam, "seeded": partition.SEEDED_STEP, "trainer": args.trainer, "bpr_local_sum": vals[0]}))
    sys.exit(0)
model = lg.LightGCN(g.num_nodes, args.dim, args.layers).to(dev)
if args.adam == "hip":
    from gnn_ecommerce_amd.optim import Adam as HipAdam
    opt = HipAdam(model.parameters(), 0.005)
else:
    opt = torch.optim.Adam(model.parameters(), 0.005, fused=(args.adam == 'fused'))
gen = torch.Generator().manual_seed(0)
purchase = g.weight == 1.0                      # positives = purchases, as pos_item_list (src/utils_v2.py:64-73)
pu, pi = g.user[purchase], g.item[purchase] + g.n_users
if args.sampler == "device":
    from gnn_ecommerce_amd.sampler import TripleSampler
    sampler = TripleSampler.from_pairs(g.n_users, g.n_items, pu, pi, pu, pi, dev, seed=0)
elif args.sampler == "python":
    import random
    from oracle import lightgcn_oracle as oracle
    import numpy as np
    order = np.argsort(pu, kind="stable"); su, si = pu[order], pi[order]
    cuts = np.flatnonzero(np.diff(su)) + 1
    lists = dict(zip(su[np.concatenate([[0], cuts])].tolist(), [a.tolist() for a in np.split(si, cuts)]))
    user_order, rng = list(lists), random.Random(0)
def batch():
    if args.sampler == "device":
        return sampler.sample(args.batch)
    if args.sampler == "python":
        u, p, n = oracle.batch_loader(user_order, lists, lists, args.batch, g.n_users, g.n_items, rng)
        return u.to(dev), p.to(dev), n.to(dev)
    u = torch.randint(0, g.n_users, (args.batch,), generator=gen)
    p = torch.randint(0, g.n_items, (args.batch,), generator=gen) + g.n_users
    n = torch.randint(0, g.n_items, (args.batch,), generator=gen) + g.n_users
    return u.to(dev), p.to(dev), n.to(dev)
def step():
    opt.zero_grad()
    u, p, n = batch()
    labels = torch.stack((torch.cat([u, u]), torch.cat([p, n])))
    out = model(ei, labels, ew)
    size = len(u)
    bpr = model.recommendation_loss(out[:size], out[size:], 0) * size
    w = model.embedding.weight
    if args.reg == "routed":
        reg = lg.regularization_loss(w, size, u, p, n, 1e-4)
    else:
        reg = 0.5 * (w[u].norm().pow(2) + w[p].norm().pow(2) + w[n].norm().pow(2)) / size * 1e-4
    loss = bpr + reg
    loss.backward()
    opt.step()
    return bpr.item(), reg.item(), loss.item()          # the three host syncs of train_lightgcn.py:149-151
for _ in range(3): vals = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(args.steps): vals = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
line = {"metric": "training steps/s (fwd+bwd+Adam, B=%d)" % args.batch, "value": 1 / dt, "ms_per_step": dt * 1e3,
        "dim": args.dim, "layers": args.layers, "sampler": args.sampler, "adam": args.adam, "reg": args.reg, "loss": vals[2]}
if args.cpu_reference > 0:          # the reference route on the host: unsorted COO, per-layer gcn_norm, autograd, dense Adam
    from oracle import lightgcn_oracle as oracle
    torch.set_num_threads(args.cpu_reference)
    ei_c, ew_c = g.coo()
    w = torch.nn.Parameter(synth.xavier_table(g.num_nodes, args.dim, 0))
    alpha = oracle.default_alpha(args.layers)
    copt = torch.optim.Adam([w], 0.005)
    cgen = torch.Generator().manual_seed(0)
    def cpu_step():
        copt.zero_grad()
        u = torch.randint(0, g.n_users, (args.batch,), generator=cgen)
        p = torch.randint(0, g.n_items, (args.batch,), generator=cgen) + g.n_users
        n = torch.randint(0, g.n_items, (args.batch,), generator=cgen) + g.n_users
        loss = oracle.train_step_loss(w, alpha, ei_c, ew_c, u, p, n, args.layers, 1e-4)[3]
        loss.backward()
        copt.step()
        return loss.item()
    cpu_step()
    t0 = time.perf_counter(); cpu_step(); cdt = time.perf_counter() - t0
    line["cpu_reference"] = {"steps_per_s": 1 / cdt, "s_per_step": cdt, "threads": args.cpu_reference, "kind": "port",
                             "speedup": cdt / dt}
print(json.dumps(line))
