#!/bin/bash
# A/B compile-time variants of the RoIAlign kernel on ONE box: ab_roi.sh "<flags A>" "<flags B>" ...
set -e -o pipefail
for F in "$@"; do
  OSR_EXTRA_HIPCC_FLAGS="$F" python3 openset-rcnn_amd/build.py > /dev/null 2>&1
  echo "== [$F]"
  python3 scripts/exp_roi5.py 2>&1 | grep -v Warning
done
