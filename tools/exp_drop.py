#!/usr/bin/env python3
"""DIAGNOSTIC (numerically meaningless): what would a per-CU LDS cache of item rows buy the user step?  Times the user
step with the entries such a cache would serve turned into padding (col = -1: no memory access) in the tile slabs -- the
processing order and everything else unchanged: none / the 600 most gathered columns / those plus every row's coldest
column / ...  Result of round 3: profiles/r03f_lds_cache_bound.txt (241 -> 232 / 227 / 220 / 212 us).
Usage: python tools/exp_drop.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnn_ecommerce_amd as lg
from gnn_ecommerce_amd import _native, graph, synth
from tools.exp_r3 import cached_graph, timed


def drop_entries(rowptr, entries, row_begin, row_end, spec):
    """A copy of ``entries`` with the dropped entries' column set to -1.  spec = "<H>" (the H most gathered columns) or
    "<H>+cold" (plus every row's coldest column)."""
    lo, hi = int(rowptr[row_begin]), int(rowptr[row_end])
    out = entries.clone()
    hot_n = int(spec.split("+")[0])
    cols = entries[lo:hi, 0].to(torch.int64)
    pop = torch.bincount(cols)
    order = torch.argsort(pop, descending=True, stable=True)
    rank = torch.empty_like(order)
    rank[order] = torch.arange(order.numel(), device=order.device)
    drop = rank[cols] < hot_n
    if spec.endswith("+cold"):
        rp = rowptr[row_begin:row_end + 1].to(torch.int64)
        rows = torch.repeat_interleave(torch.arange(row_end - row_begin, device=rowptr.device), rp[1:] - rp[:-1])
        key = pop[cols] * (int(cols.max()) + 1) + cols
        rowmin = torch.full((row_end - row_begin,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=rowptr.device)
        rowmin.scatter_reduce_(0, rows, key, "amin")
        drop |= key == rowmin[rows]
    out[lo:hi, 0] = torch.where(drop, torch.full_like(entries[lo:hi, 0], -1), entries[lo:hi, 0])
    return out


def tiles_with_dropped(op, spec):
    """The operator's tile classes (same processing order, same batch lengths) over slabs built from the thinned entries."""
    lib = _native.load()
    p = op.plan
    thin = drop_entries(op.rowptr, op.entries, p.row_begin, p.row_end, spec) if spec else op.entries
    classes = []
    for width, order, meta in graph.plan_tile_classes(op.rowptr, op.entries[:, 0], p.row_begin, p.row_end, p.short_max, graph.TILE_ORDER):
        slab = torch.empty((order.numel() * width, 2), dtype=torch.int32, device=op.rowptr.device)
        _native.check(lib.lgc_build_tiles(_native.ptr(op.rowptr), _native.ptr(thin), _native.ptr(order), order.numel(), width,
                                          _native.ptr(slab), _native.stream_of(op.rowptr.device)), "lgc_build_tiles")
        classes.append(graph.TileClass(width, order, meta, slab))
    return classes


def main():
    dev = torch.device("cuda:0")
    g = cached_graph()
    ei, ew = g.coo(dev)
    pg = lg.PropGraph(ei, ew, g.num_nodes)
    user_op, _ = pg.halves()
    x = synth.xavier_table(g.num_nodes, 64, 0, dev)
    y = torch.empty_like(x)
    ops = {}
    for spec in ("", "600", "600+cold", "2000+cold", "16000"):
        p = user_op.plan
        op = graph.Operator.build(user_op.n_rows, user_op.rowptr, user_op.entries, p.row_begin, p.row_end, p.short_max, 256)
        op._tiles = tiles_with_dropped(op, spec)
        ops[spec or "none"] = op
    for rnd in range(3):
        for name, op in ops.items():
            t = timed(lambda: op.apply(x, y))
            tr = timed(lambda: op.apply(x, y, a=0.25, r=x, b=0.25))
            if rnd == 2:
                print(f"dropped {name:>10}: user step {t:7.1f} us   with epilogue row {tr:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
