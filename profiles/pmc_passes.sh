#!/bin/bash
# PMC passes around ANY command (each pass its own rocprofv3 run with --kernel-trace + --pmc only, as
# MI355X_MICROARCH.md prescribes; a pass whose counter set the hardware refuses is reported and skipped).
#     bash profiles/pmc_passes.sh <tag> python3 tools/run_step.py user
# Output: gpurun_out/pmc_<tag>/summary.txt (+ deep.json): per kernel, mean of every counter and of the duration.
# The TA / TCP stall counters are split over passes of at most two TA counters: four in one pass exceeded the
# block's capacity (rocprofiler error 38, round 1 gpurun_out/deep_a/p4.log).
set -o pipefail
tag=${1:?tag}; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/pmc_$tag; mkdir -p "$out"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUSY_avr" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  if ! rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/p$i" -- "$@" > "$out/p$i.log" 2>&1; then
    echo "pass $i ($set) failed: $(grep -m1 -i 'error\|exceed' "$out/p$i.log")"
  fi
done
python3 - "$out" <<'PY' | tee "$out/summary.txt"
import csv, glob, sys, json
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for path in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if k.startswith("k_"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and "End_Timestamp" in r:
                dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
for k in res:
    if dur[k]:
        res[k]["_duration_us_under_pmc"] = sum(dur[k]) / len(dur[k])
json.dump(res, open(sys.argv[1] + "/deep.json", "w"), indent=1)
for k, cs in sorted(res.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-40s %.5g" % (c, v))
PY
