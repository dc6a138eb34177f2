#!/usr/bin/env python3
"""Phase times of k_mask_topk for one request row (debug build with -DLGC_TOPK_TRACE, see the build line below).
Build:  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Iinclude -DLGC_TOPK_TRACE -shared \
            -o tools/_dbg/liblgconv_trace.so gnn-ecommerce_amd/csrc/lgconv_hip.hip
Run:    LGCN_LIB_PATH=tools/_dbg/liblgconv_trace.so python tools/topk_trace.py [n_cols]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_ecommerce_amd import _native
from gnn_ecommerce_amd.propagate import mask_topk

n_cols = int(sys.argv[1]) if len(sys.argv) > 1 else 54571
dev = torch.device("cuda:0")
lib = ctypes.CDLL(_native.LIB_PATH)
lib.lgc_debug_topk_trace.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
names = {1: "keys loaded", 2: "p0 hist / short list", 3: "p0 fold", 4: "p0 scan", 5: "p1 hist", 6: "p1 fold", 7: "p1 scan",
         8: "p2 hist", 9: "p2 fold", 10: "p2 scan", 11: "collect", 12: "sort / rank"}
for mode in ("none", "dense"):
    sc = torch.randn(1, n_cols, device=dev)
    m = None if mode == "none" else (torch.rand(1, n_cols, device=dev) < 0.001).float()
    for _ in range(5):
        mask_topk(sc, m, 20)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    assert lib.lgc_debug_topk_trace(buf) == 0
    t = list(buf)[:13]
    slots = [i for i in range(13) if t[i] >= t[0] and (i == 0 or t[i] > 0)]
    slots = [i for i in slots if i == 0 or t[i] <= t[12]]
    parts, prev = [], 0
    for i in slots[1:]:
        parts.append(f"{names[i]}: +{(t[i] - t[prev]) / 100.0:.1f}us")   # s_memrealtime ticks at 100 MHz
        prev = i
    print(mode, " ".join(parts), f"total {(t[12] - t[0]) / 100.0:.1f}us", flush=True)
