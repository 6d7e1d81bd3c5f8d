#!/bin/bash
# A/B compile-time variants of the tiled RoIAlign on ONE box: ab_roi_tiled.sh "<flags A>" "<flags B>" ... (kernel stats per variant)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for F in "$@"; do
  OSR_EXTRA_HIPCC_FLAGS="$F" python3 openset-rcnn_amd/build.py > /dev/null 2>&1
  echo "== [$F]"
  rm -rf gpurun_out/prof_ab
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab -o t -- python3 scripts/exp_roi_tiled.py 2>&1 | grep "^rows\|^wave\|^tiled"
  cut -d, -f1-4 gpurun_out/prof_ab/t_kernel_stats.csv | grep -i "rt_\|roi_align" | cut -c1-110
done
