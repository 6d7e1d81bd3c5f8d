#!/bin/bash
# Counter passes over scripts/exp_bottleneck.py (counters only, one group per pass). Output: gpurun_out/pb_<tag>/
set -e -o pipefail
TAG=${1:-a}
OUT=$PWD/gpurun_out/pb_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="$PWD/scripts/exp_bottleneck.py 4"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o run -- python3 $CMD > $OUT/p$i.log 2>&1) || echo "pass $i failed"
done
python3 scripts/pmc_kernel.py bottleneck64 $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 $OUT/p6 > $OUT/summary.txt
cat $OUT/summary.txt
