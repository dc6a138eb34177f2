#!/usr/bin/env python3
"""A/B of the band sweep (lgc_spmm_sweep) against the chunked long-row path on the item half of a full-size hop.
Usage: python tools/exp_sweep.py [dim]"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gnn_ecommerce_amd as lg
from gnn_ecommerce_amd import synth
from gnn_ecommerce_amd.graph import Operator, SweepPlan


def timed(fn, reps=15, warm=3):
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
    dim = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = torch.device("cuda:0")
    g = synth.make_bipartite(**synth.CONFIG_COSMETICS, seed=0)
    ei, ew = g.coo(dev)
    pg = lg.PropGraph(ei, ew, g.num_nodes)
    nu, n = g.n_users, g.num_nodes
    op = pg.forward_op
    x = synth.xavier_table(n, dim, 0, dev)
    r = synth.xavier_table(n, dim, 1, dev)
    chunked = Operator.build(n, op.rowptr, op.entries, nu, n, 32, 256)
    ref = torch.zeros_like(x)
    chunked.apply(x, ref, a=0.5, r=r, b=0.25)
    print(f"chunked item half: {timed(lambda: chunked.apply(x, ref, a=0.5, r=r, b=0.25)):.1f} us", flush=True)
    ref64 = None
    cfgs = [dict(waves_per_band_round=256, row_cap=78),
            dict(waves_per_band_round=2048, row_cap=78, sequential=1),
            dict(n_bands=4, waves_per_band_round=2048, row_cap=78, sequential=1),
            dict(n_bands=16, waves_per_band_round=2048, row_cap=78, sequential=1),
            dict(n_bands=4, waves_per_band_round=2048, row_cap=78, sequential=1, piece_cap=128),
            dict(n_bands=2, waves_per_band_round=2048, row_cap=78, sequential=1)]
    for cfg in cfgs:
        t0 = time.perf_counter()
        sw = Operator.build(n, op.rowptr, op.entries, nu, n, 32, 256, sweep_cols=(0, nu))
        assert sw.sweep_cols is not None
        sw._sweep[cfg.get('groups', 4)] = SweepPlan(op.rowptr, op.entries, nu, n, 0, nu, cfg)
        torch.cuda.synchronize()
        t_plan = time.perf_counter() - t0
        y = torch.zeros_like(x)
        sw.apply(x, y, a=0.5, r=r, b=0.25)
        torch.cuda.synchronize()
        a, b = y[nu:].double(), ref[nu:].double()
        fro = ((a - b).norm() / b.norm()).item()
        worst = ((a - b).norm(dim=1) / b.norm(dim=1)).max().item()
        t = timed(lambda: sw.apply(x, y, a=0.5, r=r, b=0.25))
        d = sw.sweep_plan(cfg.get('groups', 4)).dims
        print(f"sweep {cfg}: {t:.1f} us   vs chunked: fro {fro:.2e} worst row {worst:.2e}   rounds {d['rounds']} "
              f"piece_cap {d['piece_cap']} waves {d['n_waves']} slabs {d['n_slabs']} slots {d['n_slots']} "
              f"pad {d['n_padding'] / (4 * d['n_steps']):.3f}  plan {t_plan:.2f} s  {sw.sweep_plan(cfg.get('groups', 4)).nbytes() / 1e6:.0f} MB", flush=True)
        del sw, y


if __name__ == "__main__":
    main()
