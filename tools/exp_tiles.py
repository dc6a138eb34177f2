#!/usr/bin/env python3
"""Prototype A/B of the tiled short-row kernel (lgc_spmm_tiles) on the user half of a full-size hop:
processing order x column parts x tiles per wave, against the current lgc_spmm user half.
Usage: python tools/exp_tiles.py [dim]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gnn_ecommerce_amd as lg
from gnn_ecommerce_amd import _native, synth
from gnn_ecommerce_amd.graph import build_row_plan


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


def row_keys(rowptr, entries, lo, hi):
    """per row of [lo, hi): degree, and the popularity (gather count) + id of its least-gathered column"""
    rp = rowptr[lo:hi + 1].long()
    deg = rp[1:] - rp[:-1]
    cols = entries[rp[0]:rp[-1], 0].long()
    pop = torch.bincount(entries[:, 0].long(), minlength=int(entries[:, 0].max()) + 1)
    rows = torch.repeat_interleave(torch.arange(hi - lo, device=rowptr.device), deg)
    key = pop[cols] * (1 << 22) + cols                  # popularity major, column id minor
    best = torch.full((hi - lo,), torch.iinfo(torch.int64).max, device=rowptr.device)
    best.scatter_reduce_(0, rows, key, "amin")
    return deg, best


def make_order(deg, cold, lo, sel, mode, bucket=2048):
    rows = torch.nonzero(sel).flatten()
    if mode == "natural":
        return (rows + lo).int()
    k = cold[rows]
    rows = rows[torch.argsort(k, stable=True)]
    if mode == "cold":
        return (rows + lo).int()
    # cold+deg: inside buckets of `bucket` rows of the cold order, longest rows first (fewer idle gather slots)
    n = rows.numel()
    b = torch.arange(n, device=rows.device) // bucket
    k2 = b * 64 + (63 - deg[rows].clamp(max=63))
    rows = rows[torch.argsort(k2, stable=True)]
    return (rows + lo).int()


class TileClass:
    def __init__(self, rowptr, entries, order, width):
        lib = _native.load()
        self.width = width
        L = 2 if width == 32 else 1
        R = 128 * L // width
        B = R // 4
        n = order.numel()
        pad = (-n) % R
        if pad:
            order = torch.cat([order, torch.full((pad,), -1, dtype=torch.int32, device=order.device)])
        # inside a tile: longest rows first; rank rho -> lane group rho % 4, batch rho // 4 -> slot g * B + bt
        tiles = order.view(-1, R).long()
        rp = rowptr.long()
        deg = torch.where(tiles >= 0, rp[(tiles + 1).clamp(min=0)] - rp[tiles.clamp(min=0)], torch.zeros_like(tiles))
        rank = torch.argsort(deg, dim=1, descending=True, stable=True)
        sorted_rows = torch.gather(tiles, 1, rank)
        sorted_deg = torch.gather(deg, 1, rank)
        rho = torch.arange(R, device=order.device)
        slot = (rho % 4) * B + rho // 4
        placed = torch.empty_like(sorted_rows)
        placed[:, slot] = sorted_rows
        bmax = sorted_deg.view(-1, B, 4).amax(dim=2)                       # rank 4bt .. 4bt+3 = batch bt
        shifts = (8 * torch.arange(B, device=order.device)).view(1, B)
        self.meta = (bmax << shifts).sum(dim=1).int().contiguous()
        order = placed.reshape(-1).int()
        self.order = order.contiguous()
        self.n_tiles = self.order.numel() // R
        self.slab = torch.empty((self.order.numel() * width, 2), dtype=torch.int32, device=order.device)
        _native.check(lib.lgc_build_tiles(rowptr.data_ptr(), entries.data_ptr(), self.order.data_ptr(), self.order.numel(),
                                          width, self.slab.data_ptr(), _native.stream_of(order.device)), "lgc_build_tiles")

    def apply(self, x, y, tpw, parts, r=None, a=1.0, b=0.0):
        lib = _native.load()
        _native.check(lib.lgc_spmm_tiles(self.order.data_ptr(), self.meta.data_ptr(), self.slab.data_ptr(), self.n_tiles, self.width, tpw, parts,
                                         min(x.size(0), y.size(0)), x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0),
                                         None if r is None else r.data_ptr(), 0 if r is None else r.stride(0), a, b,
                                         x.size(1), _native.stream_of(x.device)), "lgc_spmm_tiles")


def main():
    dim = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = torch.device("cuda:0")
    g = synth.make_bipartite(**synth.CONFIG_COSMETICS, seed=0)
    ei, ew = g.coo(dev)
    pg = lg.PropGraph(ei, ew, g.num_nodes)
    user_op, item_op = pg.halves()
    nu = g.n_users
    x = synth.xavier_table(g.num_nodes, dim, 0, dev)
    ref = torch.zeros_like(x)
    user_op.apply(x, ref)
    t_ref = timed(lambda: user_op.apply(x, ref))
    print(f"current user half: {t_ref:.1f} us", flush=True)

    rowptr, entries = user_op.rowptr, user_op.entries
    deg, cold = row_keys(rowptr, entries, 0, nu)
    lib = _native.load()
    plan_long = build_row_plan(rowptr, 0, nu, 32, 256)
    partials = torch.empty((max(plan_long.n_slots, 1), dim), dtype=torch.float32, device=dev)

    def chunks(xx, yy):
        _native.check(lib.lgc_spmm(rowptr.data_ptr(), entries.data_ptr(), 0, 0, 32, plan_long.chunks.data_ptr(),
                                   plan_long.n_chunks, plan_long.multi.data_ptr() if plan_long.n_multi else None,
                                   plan_long.n_multi, partials.data_ptr(), xx.size(0), xx.data_ptr(), xx.stride(0),
                                   yy.data_ptr(), yy.stride(0), None, 0, 1.0, 0.0, dim, _native.stream_of(dev)), "lgc_spmm")

    t_chunks = timed(lambda: chunks(x, ref))
    print(f"chunk part alone (rows > 32): {t_chunks:.1f} us   classes: <=8 {(deg <= 8).sum().item()}, "
          f"9-16 {((deg > 8) & (deg <= 16)).sum().item()}, 17-32 {((deg > 16) & (deg <= 32)).sum().item()}, "
          f">32 {(deg > 32).sum().item()}", flush=True)

    sels = [(deg <= 8, 8), ((deg > 8) & (deg <= 16), 16), ((deg > 16) & (deg <= 32), 32)]
    for mode in ("natural", "cold"):
        classes = [TileClass(rowptr, entries, make_order(deg, cold, 0, sel, mode), w) for sel, w in sels]
        for parts in (1,):
            for tpw in (1, 2, 4):
                y = torch.zeros_like(x)

                def run():
                    chunks(x, y)
                    for c in classes:
                        c.apply(x, y, tpw, parts)
                run()
                torch.cuda.synchronize()
                same = torch.equal(y[:nu], ref[:nu])
                err = ((y[:nu].double() - ref[:nu].double()).norm() / ref[:nu].double().norm()).item()
                t_all = timed(run)
                t_each = [timed(lambda c=c: c.apply(x, y, tpw, parts)) for c in classes]
                print(f"order {mode:9s} parts {parts} tiles/wave {tpw}: total {t_all:7.1f} us  "
                      f"(w8 {t_each[0]:.1f}, w16 {t_each[1]:.1f}, w32 {t_each[2]:.1f}, chunks {t_chunks:.1f})  "
                      f"bit-exact {same}  rel {err:.1e}", flush=True)
        del classes


if __name__ == "__main__":
    main()
