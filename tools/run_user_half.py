#!/usr/bin/env python3
"""One configuration of the user half of a full-size hop, N times (for rocprofv3 passes).
Usage: python tools/run_user_half.py old|tiles [order parts tiles_per_wave] [fold M] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gnn_ecommerce_amd as lg
from gnn_ecommerce_amd import synth
from tools.exp_tiles import TileClass, make_order, row_keys
from tools.exp_floor import folded


def main():
    kind = sys.argv[1]
    order, parts, tpw = (sys.argv[2], int(sys.argv[3]), int(sys.argv[4])) if kind == "tiles" else ("natural", 1, 1)
    rest = sys.argv[5:] if kind == "tiles" else sys.argv[2:]
    fold = int(rest[0]) if len(rest) > 0 else 0
    iters = int(rest[1]) if len(rest) > 1 else 5
    dim = 64
    dev = torch.device("cuda:0")
    g = synth.make_bipartite(**synth.CONFIG_COSMETICS, seed=0)
    ei, ew = g.coo(dev)
    pg = lg.PropGraph(ei, ew, g.num_nodes)
    user_op, _ = pg.halves()
    if fold:
        user_op = folded(user_op, g.n_users, g.n_items, fold)
    x = synth.xavier_table(g.num_nodes, dim, 0, dev)
    y = torch.zeros_like(x)
    if kind == "old":
        run = lambda: user_op.apply(x, y)
    else:
        deg, cold = row_keys(user_op.rowptr, user_op.entries, 0, g.n_users)
        sels = [(deg <= 8, 8), ((deg > 8) & (deg <= 16), 16), ((deg > 16) & (deg <= 32), 32)]
        classes = [TileClass(user_op.rowptr, user_op.entries, make_order(deg, cold, 0, sel, order), w) for sel, w in sels]

        def run():
            for c in classes:
                c.apply(x, y, tpw, parts)
    for _ in range(iters):
        run()
    torch.cuda.synchronize()
    print("done", kind, order, parts, tpw, fold, iters)


if __name__ == "__main__":
    main()
