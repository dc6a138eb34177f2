#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline numbers are judged against.  Run ON THE GPU BOX:
#     bash profiles/collect.sh <tag> [bench.py args...]
# Writes raw output under gpurun_out/prof_<tag>/ and a compact summary to gpurun_out/prof_<tag>/summary.json
# (copy that, plus kernel_stats.csv, into profiles/ to commit).  Counters are collected in their OWN
# passes (never together with --stats or other trace domains), as MI355X_MICROARCH.md prescribes;
# FETCH_SIZE and WRITE_SIZE do not fit one pass (3 + 2 of 4 TCC slots).
set -o pipefail
tag=${1:?tag}; shift
args=("$@")
[ ${#args[@]} -eq 0 ] && args=(--steps 10 --warmup 2 --no-cpu-baseline --no-deep --no-train --no-rank-local)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/prof_$tag
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py "${args[@]}" > "$out/trace.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 bench.py "${args[@]}" > "$out/pmc_fetch.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/pmc_write" -- python3 bench.py "${args[@]}" > "$out/pmc_write.log" 2>&1 || exit 1
# hops profiled per pass = (steps + warmup) * layers of the bench.py arguments above (defaults: 12 * 3)
hops=${HOPS:-36}
python3 profiles/summarize.py "$out" "$hops" > "$out/summary.json" && cat "$out/summary.json"
