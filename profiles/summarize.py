#!/usr/bin/env python3
"""Condense a profiles/collect.sh output directory into per-kernel averages (JSON on stdout).

HBM bytes follow MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced reads, so the read side is
given both raw and doubled ("fetch_bytes_x2") -- the row gathers here are 16 B/lane.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def kernel_stats(d):
    out = {}
    for path in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            out[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                     "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3,
                                     "pct": float(r["Percentage"])}
    return out


def counters(d, sub):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


HOP_KERNELS = ("k_spmm_hop", "k_spmm_combine", "k_lincomb", "k_rows_tile", "k_sweep", "k_combine_rows", "k_apply_fused")


def main(d, n_hops=0):
    stats = kernel_stats(d)
    pmc = defaultdict(dict)
    for sub in ("pmc_fetch", "pmc_write"):
        for k, cs in counters(d, sub).items():
            pmc[k].update(cs)
    res = {}
    for k, s in stats.items():
        if not k.startswith(HOP_KERNELS) and not k.startswith("k_pair") and s["pct"] < 1.0:
            continue
        e = dict(s)
        c = pmc.get(k, {})
        if "FETCH_SIZE" in c:
            e["fetch_bytes_raw"] = c["FETCH_SIZE"] * 1024
            e["fetch_bytes_x2"] = c["FETCH_SIZE"] * 2048
        if "WRITE_SIZE" in c:
            e["write_bytes"] = c["WRITE_SIZE"] * 1024
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            e["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0)
            e["l2_requests"] = c["TCC_HIT_sum"] + c["TCC_MISS_sum"]
        res[k] = e
    if n_hops:
        # one hop = every launch of the propagation kernels divided by the number of hops profiled
        hop = {"n_hops": n_hops, "kernel_us": 0.0, "hbm_bytes": 0.0}
        for k, e in res.items():
            if k.startswith(HOP_KERNELS):
                hop["kernel_us"] += e["calls"] * e["avg_us"] / n_hops
                hop["hbm_bytes"] += e["calls"] * (e.get("fetch_bytes_x2", 0.0) + e.get("write_bytes", 0.0)) / n_hops
        res["per_hop"] = hop
        # the figure bench.py quotes as roofline.traffic, stamped with the source it was measured on
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        try:
            import bench
            res["traffic_json"] = {"kernel_source_sha256": bench.kernel_source_hash(), "hbm_bytes_per_hop": int(hop["hbm_bytes"])}
        except Exception as exc:  # pragma: no cover
            res["traffic_json"] = {"error": str(exc)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
