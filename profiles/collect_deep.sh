#!/bin/bash
# Deeper PMC passes on the propagation kernels of the bench workload (wave wait states, instruction mix, L2 request
# latency, TA / TCP stall counters).  Run ON THE GPU BOX:  bash profiles/collect_deep.sh <tag>
# Round 1's version asked for four TA/TCP counters in one pass, which exceeds the block's capacity (rocprofiler
# error 38 -> SIGABRT, gpurun_out/deep_a/p4.log); the passes now live in profiles/pmc_passes.sh, at most two TA
# counters each, and a pass the hardware refuses is reported and skipped instead of aborting the run.
exec bash "$(dirname "$0")/pmc_passes.sh" "${1:?tag}" python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline
