#!/usr/bin/env python3
"""One configuration of the item half of a full-size hop, N times (for rocprofv3 passes).
Usage: python tools/run_item_half.py chunk|sweep [key=value ...sweep cfg] [iters=5]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gnn_ecommerce_amd as lg
from gnn_ecommerce_amd import synth
from gnn_ecommerce_amd.graph import Operator, SweepPlan


def main():
    kind = sys.argv[1]
    kv = dict(a.split("=") for a in sys.argv[2:])
    iters = int(kv.pop("iters", 5))
    cfg = {k: int(v) for k, v in kv.items()}
    dim = 64
    dev = torch.device("cuda:0")
    g = synth.make_bipartite(**synth.CONFIG_COSMETICS, seed=0)
    ei, ew = g.coo(dev)
    pg = lg.PropGraph(ei, ew, g.num_nodes)
    nu, n = g.n_users, g.num_nodes
    op = pg.forward_op
    x = synth.xavier_table(n, dim, 0, dev)
    y = torch.zeros_like(x)
    if kind == "chunk":
        o = Operator.build(n, op.rowptr, op.entries, nu, n, 32, 256)
    else:
        o = Operator.build(n, op.rowptr, op.entries, nu, n, 32, 256, sweep_cols=(0, nu))
        o._sweep[cfg.get('groups', 4)] = SweepPlan(op.rowptr, op.entries, nu, n, 0, nu, cfg)
    for _ in range(iters):
        o.apply(x, y)
    torch.cuda.synchronize()
    print("done", kind, cfg, iters)


if __name__ == "__main__":
    main()
