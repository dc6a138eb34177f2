#!/usr/bin/env python3
"""How much of a training step is host time?  Enqueue time (no sync) of forward / backward / optimizer vs the step's
wall time with the caller's three .item() syncs.  Usage: python tools/step_cpu_time.py [fused]"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnn_ecommerce_amd as lg
from gnn_ecommerce_amd import synth

dev = torch.device("cuda:0")
g = synth.make_bipartite(**synth.CONFIG_COSMETICS, seed=0)
ei, ew = g.coo(dev)
model = lg.LightGCN(g.num_nodes, 64, 3).to(dev)
opt = torch.optim.Adam(model.parameters(), 0.005, fused=(len(sys.argv) > 1 and sys.argv[1] == "fused"))
gen = torch.Generator().manual_seed(0)
B = 1024
rec = {k: [] for k in ("batch", "forward", "loss", "backward", "opt", "sync", "total")}
for it in range(25):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    opt.zero_grad()
    u = torch.randint(0, g.n_users, (B,), generator=gen).to(dev)
    p = (torch.randint(0, g.n_items, (B,), generator=gen) + g.n_users).to(dev)
    n = (torch.randint(0, g.n_items, (B,), generator=gen) + g.n_users).to(dev)
    labels = torch.stack((torch.cat([u, u]), torch.cat([p, n])))
    t.append(time.perf_counter())
    out = model(ei, labels, ew)
    t.append(time.perf_counter())
    bpr = model.recommendation_loss(out[:B], out[B:], 0) * B
    w = model.embedding.weight
    reg = 0.5 * (w[u].norm().pow(2) + w[p].norm().pow(2) + w[n].norm().pow(2)) / B * 1e-4
    loss = bpr + reg
    t.append(time.perf_counter())
    loss.backward()
    t.append(time.perf_counter())
    opt.step()
    t.append(time.perf_counter())
    vals = (bpr.item(), reg.item(), loss.item())
    t.append(time.perf_counter())
    if it >= 5:
        for k, a, b in zip(("batch", "forward", "loss", "backward", "opt", "sync"), t[:-1], t[1:]):
            rec[k].append((b - a) * 1e3)
        rec["total"].append((t[-1] - t[0]) * 1e3)
print({k: round(statistics.median(v), 3) for k, v in rec.items()})
