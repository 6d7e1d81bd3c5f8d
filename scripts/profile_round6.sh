#!/bin/bash
# Round 6: everything profiles/r06_<tag>_* is made from, in one gpurun call (run from the repo root on the GPU box):
#   scripts/profile_round.sh (kernel-trace stats + FETCH / WRITE / MFMA-busy counter passes of the inference bench command),
#   the config-3 train step alone under rocprofv3 --kernel-trace --stats (+ its per-queue timeline), and the per-layer floors table.
TAG=${1:-r06_b}
OUT=gpurun_out/prof_$TAG
ROOT=$(pwd)
timeout -k 10 ${PROFILE_LIMIT:-500} bash scripts/profile_round.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1 || { echo "profile_round failed"; tail -5 gpurun_out/${TAG}_profile.log; exit 1; }
tail -1 gpurun_out/${TAG}_profile.log | cut -c1-300
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/train -o run -- python3 $ROOT/bench.py --train-only --no-config4 --train-steps 5 > $ROOT/$OUT/train.log 2>&1 ) || { echo "train profile failed"; tail -5 $OUT/train.log; exit 1; }
grep '^{' $OUT/train.log | tail -1 | cut -c1-400
python3 scripts/trace_train_step.py $OUT/train/run_kernel_trace.csv > $OUT/train_timeline.txt 2>&1
head -6 $OUT/train_timeline.txt
timeout -k 10 200 python3 scripts/exp_layer_floor.py > $OUT/layer_floors.txt 2>/dev/null || echo "layer floors failed"
tail -1 $OUT/layer_floors.txt
