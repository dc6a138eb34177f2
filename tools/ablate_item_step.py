#!/usr/bin/env python3
"""Times the item step (band sweep + combine) of one full-size hop -- full table, and with the gathered columns folded
onto 1024 rows -- with the library named by LGCN_LIB_PATH (A/B and ablation builds of k_sweep)."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnn_ecommerce_amd as lg
from gnn_ecommerce_amd import graph as G, synth
from gnn_ecommerce_amd.graph import Operator


def timed(fn, reps=15, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return statistics.median(ts)


dev = torch.device("cuda:0")
g = synth.make_bipartite(**synth.CONFIG_COSMETICS, seed=0)
ei, ew = g.coo(dev)
pg = lg.PropGraph(ei, ew, g.num_nodes)
_, item_op = pg.halves()
x = synth.xavier_table(g.num_nodes, 64, 0, dev)
y = torch.empty_like(x)
full = timed(lambda: item_op.apply(x, y))
G.USE_SWEEP = "1"
ent = item_op.entries.clone()
ent[:, 0] = ent[:, 0] % 1024
folded = Operator.build(item_op.n_rows, item_op.rowptr, ent, g.n_users, g.num_nodes, 32, 256, sweep_cols=(0, 1024))
print(f"{os.environ.get('LGCN_LIB_PATH', 'product library'):40s} item step full {full:7.1f} us   folded onto 1024 rows "
      f"{timed(lambda: folded.apply(x, y)):7.1f} us", flush=True)
