#!/bin/bash
# Instruction mix of the weight-gradient kernel inside the training step (counters only).
set -e -o pipefail
OUT=gpurun_out/prof_wgrad_mix
mkdir -p $OUT
export TMPDIR=/tmp
CMD="scripts/bench_train.py --steps 2 --warmup 1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/a -o run -- python3 $CMD > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/b -o run -- python3 $CMD > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/c -o run -- python3 $CMD > $OUT/c.log 2>&1
python3 scripts/pmc_kernel.py conv_wgrad $OUT/a $OUT/b $OUT/c
