#!/bin/bash
# A/B two compile-time variants of the HIP library on the SAME GPU box (boxes differ by up to ~10 % in clocks, so numbers
# from different gpurun calls are not comparable). usage: ab_build_bench.sh "<flags A>" "<flags B>" [bench args]
set -e -o pipefail
FA="$1"; FB="$2"; shift 2
ARGS=${@:-"--steps 20 --warmup 5 --no-cpu-baseline"}
for round in 1 2; do
  for v in A B; do
    if [ $v = A ]; then F="$FA"; else F="$FB"; fi
    OSR_EXTRA_HIPCC_FLAGS="$F" python3 openset-rcnn_amd/build.py > /dev/null 2>&1
    python3 bench.py $ARGS 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.readline()); print('$v [$F] round $round:', d['value'], 'img/s', d['ms_per_step'], 'ms/step; conv family', d['roofline']['kernel_ms_per_step'], 'ms')"
  done
done
