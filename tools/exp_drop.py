#!/usr/bin/env python3
"""DIAGNOSTIC: what would a per-CU LDS cache of item rows buy the user step?  Times the user step with the entries such
a cache would serve turned into padding (LGCN_EXP_DROP, numerically meaningless): none / the 600 most gathered columns /
those plus every row's coldest column.  Usage: python tools/exp_drop.py"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnn_ecommerce_amd as lg
from gnn_ecommerce_amd import graph, synth
from tools.exp_r3 import cached_graph, timed

dev = torch.device("cuda:0")
g = cached_graph()
ei, ew = g.coo(dev)
pg = lg.PropGraph(ei, ew, g.num_nodes)
user_op, _ = pg.halves()
x = synth.xavier_table(g.num_nodes, 64, 0, dev)
y = torch.empty_like(x)
ops = {}
for spec in ("", "600", "600+cold", "2000+cold", "16000"):
    os.environ["LGCN_EXP_DROP"] = spec
    if not spec:
        os.environ.pop("LGCN_EXP_DROP")
    p = user_op.plan
    op = graph.Operator.build(user_op.n_rows, user_op.rowptr, user_op.entries, p.row_begin, p.row_end, p.short_max, 256)
    op.tiles
    ops[spec or "none"] = op
os.environ.pop("LGCN_EXP_DROP", None)
for rnd in range(3):
    for name, op in ops.items():
        t = timed(lambda: op.apply(x, y))
        tr = timed(lambda: op.apply(x, y, a=0.25, r=x, b=0.25))
        if rnd == 2:
            print(f"dropped {name:>10}: user step {t:7.1f} us   with epilogue row {tr:7.1f} us", flush=True)
