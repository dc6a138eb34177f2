#!/usr/bin/env python3
"""Latency of one serving request (torchserve/lightgcn_handler.py:73-96 -> LightGCN.recommendK, k=20) on the
cosmetics-scale graph: with the propagated table reused across requests (default) and recomputed per
request as upstream does."""
import json, os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnn_ecommerce_amd as lg
from gnn_ecommerce_amd import synth

dev = torch.device("cuda:0")
g = synth.make_bipartite(**synth.CONFIG_COSMETICS, seed=0)
ei, ew = g.coo(dev)
model = lg.LightGCN(g.num_nodes, 64, 3).to(dev).eval()
out = {}
for n_users in (1, 64):
    users = list(range(7, 7 + n_users))
    seen = torch.zeros(n_users, g.n_items)
    seen_dev = seen.to(dev)                      # a caller that keeps its interaction rows on the device
    model.cache_recommend_embeddings = True
    ts = []
    with torch.no_grad():
        for _ in range(23):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            model.recommendK(ei, ew, g.n_users, g.n_items, seen_dev, users, 20)
            ts.append((time.perf_counter() - t0) * 1e3)
    out[f"users{n_users}_reuse_devmask_ms"] = round(statistics.median(ts[3:]), 3)
    for reuse in (True, False):
        model.cache_recommend_embeddings = reuse
        ts = []
        with torch.no_grad():
            for _ in range(13):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                model.recommendK(ei, ew, g.n_users, g.n_items, seen, users, 20)
                ts.append((time.perf_counter() - t0) * 1e3)
        out[f"users{n_users}_{'reuse' if reuse else 'recompute'}_ms"] = round(statistics.median(ts[3:]), 3)
print(json.dumps(out))
