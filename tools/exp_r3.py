#!/usr/bin/env python3
"""Round-3 A/B harness: the two halves of one full-size hop and the whole K-layer forward, for several builds of the
library and several environment knobs, interleaved in ONE process (boxes differ by +-1.5 %).

    python tools/exp_r3.py [name=lib.so[,ENV=val...]] ...      (name "base" = the product library)

A variant is a library path (optional) plus environment variables that the library reads at launch time
(e.g. LGCN_SWEEP_DEPTH).  Every variant's results are checked against the first.
Env: EXP_DIM (64), EXP_LAYERS (3), EXP_ROUNDS (5)."""
import ctypes
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import gnn_ecommerce_amd as lg
from gnn_ecommerce_amd import _native, graph, propagate, synth


def cached_graph(seed=0):
    path = f"/tmp/lgcn_cosmetics_seed{seed}.npz"
    if os.path.exists(path):
        z = np.load(path)
        return synth.BipartiteGraph(int(z["n_users"]), int(z["n_items"]), z["user"], z["item"], z["weight"])
    g = synth.make_bipartite(**synth.CONFIG_COSMETICS, seed=seed)
    np.savez(path, n_users=g.n_users, n_items=g.n_items, user=g.user, item=g.item, weight=g.weight)
    return g


def load_variant(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, (restype, argtypes) in _native.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = restype, argtypes
    return lib


def timed(fn, reps=7, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return statistics.median(ts)


def main():
    specs = sys.argv[1:] or ["base"]
    dim, layers, rounds = int(os.environ.get("EXP_DIM", 64)), int(os.environ.get("EXP_LAYERS", 3)), int(os.environ.get("EXP_ROUNDS", 5))
    dev = torch.device("cuda:0")
    g = cached_graph()
    ei, ew = g.coo(dev)
    pg = lg.PropGraph(ei, ew, g.num_nodes)
    user_op, item_op = pg.halves()
    x = synth.xavier_table(g.num_nodes, dim, 0, dev)
    y = torch.empty_like(x)
    alphas = tuple([1.0 / (layers + 1)] * (layers + 1))
    variants = []
    for spec in specs:
        name, _, rest = spec.partition("=")
        parts = rest.split(",") if rest else []
        lib_path = next((p for p in parts if p.endswith(".so")), None)
        env = dict(p.split("=", 1) for p in parts if "=" in p and not p.endswith(".so"))
        variants.append((name, load_variant(lib_path) if lib_path else _native.load(), env))
    res = {n: {"user": [], "user+r": [], "item": [], "hop": []} for n, _, _ in variants}
    ref = {}
    for r in range(rounds):
        for name, lib, env in variants:
            _native._lib = lib
            saved = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            entry = "edge"
            uop = user_op
            t_u = timed(lambda: uop.apply(x, y))
            yu = y[:g.n_users:997].clone()
            t_ur = timed(lambda: uop.apply(x, y, a=0.25, r=x, b=0.25))
            t_i = timed(lambda: item_op.apply(x, y))
            yi = y[g.n_users:].clone()
            if entry == "edge":
                t_h = timed(lambda: propagate.propagate_sum(x, pg, alphas), reps=5) / layers
            else:
                t_h = float("nan")
            for key, val in (("user", yu), ("item", yi)):
                if key not in ref:
                    ref[key] = val.double()
                else:
                    err = ((val.double() - ref[key]).norm() / ref[key].norm()).item()
                    assert err < 2e-6, (name, key, err)
            for k, v in (("user", t_u), ("user+r", t_ur), ("item", t_i), ("hop", t_h)):
                res[name][k].append(v)
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    print(f"{'variant':>18} {'user':>8} {'user+r':>8} {'item':>8} {'hop':>8}   us, median of {rounds} rounds (item = sweep + combine)")
    for name, _, _ in variants:
        m = {k: statistics.median(v) for k, v in res[name].items()}
        print(f"{name:>18} {m['user']:8.1f} {m['user+r']:8.1f} {m['item']:8.1f} {m['hop']:8.1f}", flush=True)


if __name__ == "__main__":
    main()
