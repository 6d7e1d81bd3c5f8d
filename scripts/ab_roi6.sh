#!/bin/bash
# Round 6 A/B of the RoIAlign forward on ONE box: ab_roi6.sh "<flags A>" "<flags B>" ... ; the word BASE stands for the round-5 kernel
# (scripts/_variants/osr_roi_align_r05.hip, a git-ignored copy of the previous source made for the run: `git show <rev>:...`).
set -e -o pipefail
cp openset-rcnn_amd/csrc/osr_roi_align.hip /tmp/osr_roi_align_new.hip
for F in "$@"; do
  if [ "$F" = "BASE" ]; then cp scripts/_variants/osr_roi_align_r05.hip openset-rcnn_amd/csrc/osr_roi_align.hip; FL=""; else cp /tmp/osr_roi_align_new.hip openset-rcnn_amd/csrc/osr_roi_align.hip; FL="$F"; fi
  OSR_EXTRA_HIPCC_FLAGS="$FL" python3 openset-rcnn_amd/build.py > /dev/null 2>&1
  echo "== [$F]"
  python3 scripts/exp_roi5.py 2>&1 | grep -v "Warning\|amdgpu.ids"
done
cp /tmp/osr_roi_align_new.hip openset-rcnn_amd/csrc/osr_roi_align.hip
python3 openset-rcnn_amd/build.py > /dev/null 2>&1
