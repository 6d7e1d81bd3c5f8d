#!/bin/bash
# Round 6: the bench's headline (4 lanes) and single-pass figures for RoIAlign build variants on ONE box, two rounds.
# usage: ab_bench6.sh "<flags A>" "<flags B>" ... ; BASE = the round-5 kernel (scripts/_variants/osr_roi_align_r05.hip, git-ignored copy)
set -e -o pipefail
cp openset-rcnn_amd/csrc/osr_roi_align.hip /tmp/osr_roi_align_new.hip
for round in 1 2; do
  for F in "$@"; do
    if [ "$F" = "BASE" ]; then cp scripts/_variants/osr_roi_align_r05.hip openset-rcnn_amd/csrc/osr_roi_align.hip; FL=""; else cp /tmp/osr_roi_align_new.hip openset-rcnn_amd/csrc/osr_roi_align.hip; FL="$F"; fi
    OSR_EXTRA_HIPCC_FLAGS="$FL" python3 openset-rcnn_amd/build.py > /dev/null 2>&1
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-step --no-pmc --no-parity --no-pcie 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.readline()); r = [k for k in d['roofline_hbm']['kernels'] if k['kernel'] == 'roi_align'][0]; print('[$F] round $round:', d['value'], 'img/s', d['ms_per_step'], 'ms/step; single pass', d['single_pass']['ms_per_step'], 'ms; family', d['roofline']['kernel_ms_per_step'], 'ms; roi_align', r['ms'], 'ms')"
  done
done
cp /tmp/osr_roi_align_new.hip openset-rcnn_amd/csrc/osr_roi_align.hip
python3 openset-rcnn_amd/build.py > /dev/null 2>&1
