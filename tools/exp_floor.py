#!/usr/bin/env python3
"""Where can a hop's time go?  Times the item step and the user step of one full-size hop separately, then
again with the gathered columns folded onto a small set of rows (col -> base + (col - base) % M), which
leaves every instruction, row length and store unchanged but makes the gathers hit L2 (or L1): the
difference is what re-fetch traffic costs each half, the remainder is its floor with perfect locality.
Results are numerically meaningless (diagnostic only).  Usage: python tools/exp_floor.py [dim]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gnn_ecommerce_amd as lg
from gnn_ecommerce_amd import synth
from gnn_ecommerce_amd.graph import Operator


def timed(fn, reps=20, warm=3):
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
    return statistics.median(ts), min(ts)


def folded(op: Operator, base: int, span: int, m: int) -> Operator:
    """The same rows with every gathered column of [base, base + span) folded onto m rows (plans rebuilt; the
    chunk + tile path, no sweep: the point is the memory behaviour of the gathers)."""
    ent = op.entries.clone()
    col = ent[:, 0]
    sel = (col >= base) & (col < base + span)
    col[sel] = base + (col[sel] - base) % m
    p = op.plan
    return Operator.build(op.n_rows, op.rowptr, ent, p.row_begin, p.row_end, 32, 256)


def main():
    dim = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = torch.device("cuda:0")
    g = synth.make_bipartite(**synth.CONFIG_COSMETICS, seed=0)
    ei, ew = g.coo(dev)
    pg = lg.PropGraph(ei, ew, g.num_nodes)
    user_op, item_op = pg.halves()
    item_chunked = Operator.build(item_op.n_rows, item_op.rowptr, item_op.entries, g.n_users, g.num_nodes, 32, 256)
    x = synth.xavier_table(g.num_nodes, dim, 0, dev)
    y = torch.empty_like(x)
    nu, ni = g.n_users, g.n_items
    row_b = dim * 4
    print(f"dim {dim}: item half gathers {g.nnz // 2} user rows ({nu * row_b / 1e6:.0f} MB table), "
          f"user half gathers {g.nnz // 2} item rows ({ni * row_b / 1e6:.1f} MB table)", flush=True)
    med, mn = timed(lambda: item_op.apply(x, y))
    print(f"item half  full (product)   : {med:8.1f} us (min {mn:.1f})", flush=True)
    med, mn = timed(lambda: item_chunked.apply(x, y))
    print(f"item half  full (chunked)   : {med:8.1f} us (min {mn:.1f})", flush=True)
    med, mn = timed(lambda: user_op.apply(x, y))
    print(f"user half  full             : {med:8.1f} us (min {mn:.1f})", flush=True)
    med, mn = timed(lambda: user_op.apply(x, y, a=1.0, r=x, b=0.25))
    print(f"user half  full + epilogue  : {med:8.1f} us (min {mn:.1f})", flush=True)
    for m in (64, 2048, 8192, 16384, 32768):
        op = folded(user_op, nu, ni, m)
        med, mn = timed(lambda: op.apply(x, y))
        print(f"user half  items mod {m:7d} ({m * row_b / 1e6:7.2f} MB): {med:8.1f} us (min {mn:.1f})", flush=True)
        del op
    for m in (64, 4096, 12288, 65536, 262144, 786432):
        op = folded(item_chunked, 0, nu, m)
        med, mn = timed(lambda: op.apply(x, y))
        print(f"item half  users mod {m:7d} ({m * row_b / 1e6:7.2f} MB): {med:8.1f} us (min {mn:.1f})", flush=True)
        del op
    # the band sweep itself with its gathers folded: what do its steps (slab, broadcast, LDS read-modify-write, partial
    # write-out, combine) cost when the gathered rows come from L1 / L2?
    from gnn_ecommerce_amd import graph as G
    G.USE_SWEEP = "1"
    for m in (1024, 16384, 131072):
        ent = item_op.entries.clone()
        ent[:, 0] = ent[:, 0] % m
        op = Operator.build(item_op.n_rows, item_op.rowptr, ent, g.n_users, g.num_nodes, 32, 256, sweep_cols=(0, m))
        med, mn = timed(lambda: op.apply(x, y))
        print(f"item half  SWEEP, users mod {m:7d} ({m * row_b / 1e6:7.2f} MB): {med:8.1f} us (min {mn:.1f})  "
              f"rounds {op.sweep_plan(4).dims['rounds']}", flush=True)
        del op


if __name__ == "__main__":
    main()
