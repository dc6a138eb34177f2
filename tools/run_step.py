#!/usr/bin/env python3
"""One part of a full-size hop, N times, for rocprofv3 passes (library: LGCN_LIB_PATH, planner/launch knobs: LGCN_*).
Usage: python3 tools/run_step.py user|userr|item|hop [iters] [dim]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gnn_ecommerce_amd as lg
from gnn_ecommerce_amd import propagate, synth
from tools.exp_r3 import cached_graph


def main():
    what = sys.argv[1]
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    dim = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    dev = torch.device("cuda:0")
    g = cached_graph()
    ei, ew = g.coo(dev)
    pg = lg.PropGraph(ei, ew, g.num_nodes)
    user_op, item_op = pg.halves()
    x = synth.xavier_table(g.num_nodes, dim, 0, dev)
    y = torch.empty_like(x)
    for _ in range(iters):
        if what == "user":
            user_op.apply(x, y)
        elif what == "userr":
            user_op.apply(x, y, a=0.25, r=x, b=0.25)
        elif what == "item":
            item_op.apply(x, y)
        else:
            propagate.propagate_sum(x, pg, (0.25, 0.25, 0.25, 0.25))
    torch.cuda.synchronize()
    print("done", what, iters, dim)


if __name__ == "__main__":
    main()
