#!/bin/bash
# Round 6 A/B of the dense RoIAlign backward on ONE box: ab_roibwd6.sh "<flags A>" ... ; BASE = the kernel before the row prefetch
# (scripts/_variants/osr_roi_align_r06a.hip, a git-ignored copy made for the run)
set -e -o pipefail
cp openset-rcnn_amd/csrc/osr_roi_align.hip /tmp/osr_roi_align_new.hip
for F in "$@"; do
  if [ "$F" = "BASE" ]; then cp scripts/_variants/osr_roi_align_r06a.hip openset-rcnn_amd/csrc/osr_roi_align.hip; FL=""; else cp /tmp/osr_roi_align_new.hip openset-rcnn_amd/csrc/osr_roi_align.hip; FL="$F"; fi
  OSR_EXTRA_HIPCC_FLAGS="$FL" python3 openset-rcnn_amd/build.py > /dev/null 2>&1
  echo "== [$F]"
  python3 scripts/exp_roi_bwd.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -2
  python3 scripts/exp_roi_bwd.py clustered 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -2 | sed 's/^/clustered: /'
done
cp /tmp/osr_roi_align_new.hip openset-rcnn_amd/csrc/osr_roi_align.hip
python3 openset-rcnn_amd/build.py > /dev/null 2>&1
