#!/usr/bin/env python3
"""End to end on the serving side (SURVEY.md 8a a-S + 8f N4): raw-id interaction CSV -> ingest (relabel, COO, CSR on the
device) -> ONE persisted graph file -> handler-shaped worker (stub TorchServe context) -> requests.
    python tools/serve_demo.py [--full]      (--full: the cosmetics-scale graph, else 20k x 2k)"""
import argparse
import json
import os
import statistics
import sys
import tempfile
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pandas as pd
import torch

import gnn_ecommerce_amd as lg
from gnn_ecommerce_amd import ingest, serving, synth

ap = argparse.ArgumentParser()
ap.add_argument("--full", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = synth.CONFIG_COSMETICS if args.full else dict(n_users=20000, n_items=2000, n_pairs=200000)
g = synth.make_bipartite(**cfg, seed=0)
rng = np.random.default_rng(0)
raw_users = np.sort(rng.choice(10 ** 9, g.n_users, replace=False))        # raw ids as the shop's event log has them
raw_items = np.sort(rng.choice(10 ** 7, g.n_items, replace=False))
out = {}
with tempfile.TemporaryDirectory() as d:
    csv = os.path.join(d, "interactions.csv")
    pd.DataFrame({"user_id": raw_users[g.user], "item_id": raw_items[g.item], "weight": g.weight}).to_csv(csv, index=False)
    t0 = time.perf_counter()
    inter = ingest.read_interactions_csv(csv)
    out["csv_parse_relabel_s"] = round(time.perf_counter() - t0, 2)
    t0 = time.perf_counter()
    ingest.save_serving_graph(os.path.join(d, serving.GRAPH_FILE), inter, device=dev)
    torch.cuda.synchronize()
    out["build_and_save_s"] = round(time.perf_counter() - t0, 2)
    out["graph_file_mb"] = round(os.path.getsize(os.path.join(d, serving.GRAPH_FILE)) / 1e6, 1)
    model = lg.LightGCN(inter.n_users + inter.n_items, 64, 3)
    torch.save({"model_state_dict": model.state_dict(), "hyperparams": {"latent_dim": 64, "n_layers": 3}},
               os.path.join(d, "LightGCN_best.pt"))
    ctx = types.SimpleNamespace(manifest={"model": {"serializedFile": "LightGCN_best.pt"}},
                                system_properties={"model_dir": d, "gpu_id": 0})
    t0 = time.perf_counter()
    h = serving.RecommendHandler()
    h.initialize(ctx)
    torch.cuda.synchronize()
    out["worker_initialize_s"] = round(time.perf_counter() - t0, 2)
    first = h.handle([{"body": [7]}])
    ts = []
    for r in range(30):
        t0 = time.perf_counter()
        resp = h.handle([{"body": [7 + r]}])
        ts.append((time.perf_counter() - t0) * 1e3)
    out["request_1_user_ms"] = round(statistics.median(ts[5:]), 3)
    ts = []
    for r in range(15):
        t0 = time.perf_counter()
        h.handle([{"body": list(range(r, r + 64))}])
        ts.append((time.perf_counter() - t0) * 1e3)
    out["request_64_users_ms"] = round(statistics.median(ts[3:]), 3)
    out["response_shape"] = [len(first), list(first[0]), len(first[0]["items"]), len(first[0]["items"][0])]
print(json.dumps(out))
