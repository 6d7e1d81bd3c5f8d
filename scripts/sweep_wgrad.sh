#!/bin/bash
# sweep of the weight-gradient split targets inside the training step (diagnostic build with the env knobs)
OSR_EXTRA_HIPCC_FLAGS="-DOSR_EXPERIMENT" python3 openset-rcnn_amd/build.py > /dev/null 2>&1 || exit 1
for cfg in "192 384" "128 384" "256 384" "192 256" "192 512" "160 320" "256 512" "192 384"; do
  set -- $cfg
  OSR_WGRAD_TARGET_BIG=$1 OSR_WGRAD_TARGET_SMALL=$2 python3 bench.py --train-only --train-steps 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('big $1 small $2:', d['ms_per_iter'], 'ms; bwd', d['backward_ms'])" || exit 1
done
python3 openset-rcnn_amd/build.py > /dev/null 2>&1
