#!/bin/bash
# Like ab_build_bench.sh for any number of flag sets: ab_multi.sh "<flags1>" "<flags2>" ...   (2 rounds, same box)
set -e -o pipefail
for round in 1 2; do
  for F in "$@"; do
    OSR_EXTRA_HIPCC_FLAGS="$F" python3 openset-rcnn_amd/build.py > /dev/null 2>&1
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.readline()); print('[$F] round $round:', d['value'], 'img/s', d['ms_per_step'], 'ms/step; conv family', d['roofline']['kernel_ms_per_step'], 'ms')"
  done
done
