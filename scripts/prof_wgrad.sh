#!/bin/bash
# Counters of the weight-gradient kernel inside the training step (separate --pmc passes, counters only).
set -e -o pipefail
OUT=gpurun_out/prof_wgrad
mkdir -p $OUT
export TMPDIR=/tmp
CMD="scripts/bench_train.py --steps 2 --warmup 1"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/a -o run -- python3 $CMD > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/b -o run -- python3 $CMD > $OUT/b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/c -o run -- python3 $CMD > $OUT/c.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/d -o run -- python3 $CMD > $OUT/d.log 2>&1
python3 scripts/pmc_kernel.py conv_wgrad $OUT/a $OUT/b $OUT/c $OUT/d > $OUT/summary.txt
python3 scripts/pmc_kernel.py wgrad_reduce $OUT/a $OUT/c $OUT/d >> $OUT/summary.txt
cat $OUT/summary.txt
