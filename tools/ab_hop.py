#!/usr/bin/env python3
"""A/B builds of liblgconv_hip.so on the full-size propagate, interleaved in ONE process
(cdna_hip_programming.md rule 24).  Usage:  python tools/ab_hop.py name=path.so [name=path.so ...]
Optional env: AB_DIM (64), AB_LAYERS (3), AB_ROUNDS (7), LGCN_SHORT_MAX, LGCN_CHUNK_LEN."""
import ctypes
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gnn_ecommerce_amd as lg
from gnn_ecommerce_amd import _native, propagate, synth


def load_variant(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, (restype, argtypes) in _native.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = restype, argtypes
    return lib


def main():
    variants = [a.split("=", 1) for a in sys.argv[1:]] or [["default", _native.LIB_PATH]]
    libs = {name: load_variant(path) for name, path in variants}
    dim, layers, rounds = int(os.environ.get("AB_DIM", 64)), int(os.environ.get("AB_LAYERS", 3)), int(os.environ.get("AB_ROUNDS", 7))
    dev = torch.device("cuda:0")
    g = synth.make_bipartite(**synth.CONFIG_COSMETICS, seed=0)
    ei, ew = g.coo(dev)
    pg = lg.PropGraph(ei, ew, g.num_nodes)
    x0 = synth.xavier_table(g.num_nodes, dim, 0, dev)
    alphas = tuple([1.0 / (layers + 1)] * (layers + 1))
    ref = None
    times = {n: [] for n in libs}
    for r in range(rounds + 1):
        for name, lib in libs.items():
            _native._lib = lib
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                out = propagate.propagate_sum(x0, pg, alphas)
            e.record()
            torch.cuda.synchronize()
            if r > 0:
                times[name].append(s.elapsed_time(e) / 5 / layers * 1e3)
            if ref is None:
                ref = out.clone()
            else:
                err = ((out.double() - ref.double()).norm() / ref.double().norm()).item()
                assert err < 1e-6, (name, err)
    for name, t in times.items():
        print(f"{name:>16}: us/hop median {statistics.median(t):8.1f}  min {min(t):8.1f}  "
              f"-> {g.nnz / statistics.median(t) / 1e3:6.2f} G edges/s/layer")


if __name__ == "__main__":
    main()
