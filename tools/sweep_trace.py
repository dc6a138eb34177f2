#!/usr/bin/env python3
"""Diagnostic: how far apart do the wavefronts of one band drift during a sweep round?  Every wave stamps
s_memrealtime (100 MHz) at its start and after each slab (LGCN_SWEEP_TRACE = address of a [n_waves, 16] u64 buffer; slots 0..14 = start and the first 14 slabs, 15 = end).
Prints, per round, the spread of start times and of the times at which waves finish their k-th slab."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gnn_ecommerce_amd as lg
from gnn_ecommerce_amd import synth

dev = torch.device("cuda:0")
g = synth.make_bipartite(**synth.CONFIG_COSMETICS, seed=0)
ei, ew = g.coo(dev)
pg = lg.PropGraph(ei, ew, g.num_nodes)
_, item_op = pg.halves()
x = synth.xavier_table(g.num_nodes, 64, 0, dev)
y = torch.empty_like(x)
for _ in range(3):
    item_op.apply(x, y)
sw = item_op.sweep
nw = sw.dims["n_waves"]
trace = torch.zeros((nw, 16), dtype=torch.int64, device=dev)
os.environ["LGCN_SWEEP_TRACE"] = str(trace.data_ptr())
item_op.apply(x, y)
torch.cuda.synchronize()
del os.environ["LGCN_SWEEP_TRACE"]
t = trace.cpu().numpy().astype(np.float64) / 100.0          # us
nslab = np.diff(sw.wave_slab_ptr.cpu().numpy())
wpbr, nb, rounds = nw // (sw.dims["rounds"] * 8), 8, sw.dims["rounds"]
w = np.arange(nw)
band = (w // 4) % nb
rnd = (w // 4) // nb // (wpbr // 4)
t0 = t[t > 0].min()
print(f"waves {nw}, rounds {rounds}, slabs per wave {nslab.min()}..{nslab.max()}, kernel span {t.max() - t0:.1f} us")
for r in range(rounds):
    for b in (0, 3, 7):
        sel = (band == b) & (rnd == r) & (nslab > 0)
        tt = t[sel] - t0
        ns = nslab[sel]
        line = f"round {r} band {b}: start {tt[:, 0].min():6.1f}..{tt[:, 0].max():6.1f}"
        for k in (1, 3, 5, 7, 11):
            ok = ns >= k
            if ok.any():
                v = tt[ok, k]
                line += f" | slab{k} {np.percentile(v, 5):6.1f}..{np.percentile(v, 95):6.1f} (sd {v.std():4.1f})"
        end = tt[:, 15]                                      # slot 15 = end of the wave's last slab
        line += f" | end {end.min():6.1f}..{end.max():6.1f}"
        print(line)
