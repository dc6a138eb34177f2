#!/usr/bin/env python3
"""Times the user step of one full-size hop (full table, and with the gathered columns folded onto 64 rows) with the
library named by LGCN_LIB_PATH -- used for A/B and ablation builds of the tile kernel (a build
whose tile kernel skips its output stores, one whose gathers are replaced by constants; such builds are numerically
meaningless and are not kept in the source)."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnn_ecommerce_amd as lg
from gnn_ecommerce_amd import synth
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
user_op, _ = pg.halves()
x = synth.xavier_table(g.num_nodes, 64, 0, dev)
y = torch.empty_like(x)
full = timed(lambda: user_op.apply(x, y))
ent = user_op.entries.clone()
col = ent[:, 0]
sel = col >= g.n_users
col[sel] = g.n_users + (col[sel] - g.n_users) % 64
p = user_op.plan
folded = Operator.build(user_op.n_rows, user_op.rowptr, ent, p.row_begin, p.row_end, 32, 256)
print(f"{os.environ.get('LGCN_LIB_PATH', 'product library'):50s} user step full {full:7.1f} us   folded onto 64 rows "
      f"{timed(lambda: folded.apply(x, y)):7.1f} us", flush=True)
