#!/usr/bin/env python3
"""DESIGN.md section 5's line-cost model against a rocprofv3 PMC summary (profiles/collect.sh + profiles/summarize.py):
a kernel's time ~ sum over its L2 requests of cost(kind) / 256 CUs, with the per-CU costs of
tools/microbench/line_cost.hip (profiles/r03a_line_cost.txt).

    python tools/line_model.py profiles/r03b_pmc_summary.json [--table hbm|mall]

Per kernel: L2 requests, hits, read misses (misses - written lines), written lines; model us; measured us."""
import json
import sys

HIT, MISS_MALL, MISS_HBM, WRITE = 1.10, 3.8, 4.34, 3.1      # ns per 128-byte line and CU
CUS = 256
# which memory a kernel's read misses come from: the user step gathers a 14-21 MB item table (Infinity Cache),
# everything else streams or gathers from tables far beyond it
MALL_KERNELS = ("k_apply_fused",)


def main():
    d = json.load(open(sys.argv[1]))
    print(f"{'kernel':<24}{'requests':>10}{'hits':>9}{'rd miss':>9}{'written':>9}{'model us':>10}{'measured':>10}")
    tot_m = tot_t = 0.0
    for name, k in d.items():
        if not isinstance(k, dict) or "l2_requests" not in k or not name.startswith(("k_apply_fused", "k_sweep", "k_lincomb", "k_spmm_combine")):
            continue
        req = k["l2_requests"]
        hits = req * k["l2_hit_rate"]
        written = k["write_bytes"] / 128.0
        rd_miss = max(req - hits - written, 0.0)
        miss_cost = MISS_MALL if name.startswith(MALL_KERNELS) else MISS_HBM
        model = (hits * HIT + rd_miss * miss_cost + written * WRITE) / CUS * 1e-3
        print(f"{name:<24}{req / 1e6:>9.2f}M{hits / 1e6:>8.2f}M{rd_miss / 1e6:>8.2f}M{written / 1e6:>8.2f}M{model:>10.1f}{k['avg_us']:>10.1f}")
        calls = k.get("calls", 0)
        tot_m += model * calls
        tot_t += k["avg_us"] * calls
    if tot_t:
        print(f"all launches of these kernels: model {tot_m / 1e3:.2f} ms, measured {tot_t / 1e3:.2f} ms ({100 * tot_m / tot_t:.0f} %)")


if __name__ == "__main__":
    main()
