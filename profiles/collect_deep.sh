#!/bin/bash
# Deeper PMC passes on the propagation kernels (wave wait states, TLB, L2 read latency, DRAM share).
# Run ON THE GPU BOX: bash profiles/collect_deep.sh <tag>.  Each pass is a separate rocprofv3 run with
# --pmc only (plus the kernel trace), as the guide prescribes.
set -o pipefail
tag=${1:?tag}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/deep_$tag; mkdir -p "$out"
args=(--steps 5 --warmup 2 --no-cpu-baseline)
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/p$i" -- python3 bench.py "${args[@]}" > "$out/p$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$out/p$i.log"; }
done
python3 - "$out" <<'PY'
import csv, glob, sys, json
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if k.startswith("k_spmm") or k.startswith("k_lincomb"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
json.dump(res, open(sys.argv[1] + "/deep.json", "w"), indent=1)
for k, cs in res.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-40s %.4g" % (c, v))
PY
