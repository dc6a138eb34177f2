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
# the select kernel alone (HIP events): dense device mask and the list form
from gnn_ecommerce_amd.propagate import SeenLists, mask_topk
for rows in (1, 64, 1024):
    sc = torch.randn(rows, g.n_items, device=dev)
    dense = (torch.rand(rows, g.n_items, device=dev) < 0.001).float()
    nz = torch.nonzero(dense)
    ptr = torch.zeros(rows + 1, dtype=torch.int64, device=dev)
    ptr[1:] = torch.cumsum(dense.sum(dim=1).long(), 0)
    lists = SeenLists(ptr, nz[:, 1].contiguous(), torch.arange(rows, device=dev))
    for name, m in (("dense", dense), ("lists", lists)):
        for _ in range(3):
            mask_topk(sc, m, 20)
        ts = []
        for _ in range(20):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); mask_topk(sc, m, 20); e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e3)
        out[f"mask_topk_rows{rows}_{name}_us"] = round(statistics.median(ts), 1)
print(json.dumps(out))
