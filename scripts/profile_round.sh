#!/bin/bash
# Profiling recipe for one round (run on the GPU box from the repo root, e.g. through gpurun):
#   1. rocprofv3 --kernel-trace --stats of the default bench workload (no CPU baseline, eager single stream so that every
#      launch is a separate dispatch with its own duration);
#   2. three PMC passes (FETCH_SIZE; WRITE_SIZE; MFMA busy cycles + GRBM_GUI_ACTIVE) of the same command, counters only (never combined with other traces).
#   (passes of the path per run: 2 warm-up + 5 timed + 2 attribution + the 4-image calibration pass = 9.25 sixteen-image steps)
# Outputs land under gpurun_out/prof_<tag>/ ; scripts/pmc_summary.py turns them into the json committed under profiles/.
set -e -o pipefail
TAG=${1:-r01_d}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-step --no-pmc --no-parity --streams 1 --no-graph"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o run -- python3 $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o run -- python3 $CMD > $OUT/pmc_write.log 2>&1
# matrix-pipe utilisation and the clock the chip holds: MFMA busy cycles (summed over the 1024 SIMDs) against GRBM_GUI_ACTIVE (summed over the 8 XCDs)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace --output-format csv -d $OUT/pmc_mfma -o run -- python3 $CMD > $OUT/pmc_mfma.log 2>&1
python3 scripts/pmc_summary.py $OUT ${PASSES:-9.25} > $OUT/traffic.json
tail -2 $OUT/stats.log
