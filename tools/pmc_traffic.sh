#!/bin/bash
# Fabric traffic of a command's kernels: FETCH_SIZE and WRITE_SIZE + L2 hits/misses in two rocprofv3 passes
# (--kernel-trace + --pmc only), then per kernel: mean duration, fetched bytes (x2: MI355X_MICROARCH.md, gfx950),
# written bytes, L2 hit rate.    bash tools/pmc_traffic.sh <tag> python3 tools/run_step.py user
set -o pipefail
tag=${1:?tag}; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/traffic_$tag; mkdir -p "$out"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/p$i" -- "$@" > "$out/p$i.log" 2>&1 || echo "pass $i failed: $(tail -2 "$out/p$i.log")"
done
python3 - "$out" "$tag" <<'PY' | tee "$out/summary.txt"
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
for path in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if k.startswith("k_"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
for k, cs in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    hit = m.get("TCC_HIT_sum", 0.0); miss = m.get("TCC_MISS_sum", 0.0)
    print("%-12s %-28s %7.1f us  fetch %6.3f GB  write %6.3f GB  L2 hit %5.1f %%  requests %6.2f M" % (
        sys.argv[2], k, sum(dur[k]) / len(dur[k]), m.get("FETCH_SIZE", 0) * 1024 * 2 / 1e9, m.get("WRITE_SIZE", 0) * 1024 / 1e9,
        100 * hit / max(hit + miss, 1), (hit + miss) / 1e6))
PY
