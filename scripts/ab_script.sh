#!/bin/bash
# A/B compile-time variants of the HIP library on ONE box with any experiment script:
#   ab_script.sh "<python script + args>" "<flags A>" "<flags B>" ...
set -e -o pipefail
CMD="$1"; shift
for F in "$@"; do
  OSR_EXTRA_HIPCC_FLAGS="$F" python3 openset-rcnn_amd/build.py > /dev/null 2>&1
  echo "== [$F]"
  python3 $CMD 2>&1 | grep -v "Warning\|amdgpu.ids"
done
