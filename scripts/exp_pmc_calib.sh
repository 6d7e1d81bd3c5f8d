#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of scripts/exp_pmc_calib.hip's kernels (separate passes), summed per kernel name.
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
hipcc -O3 --offload-arch=gfx950 $R/scripts/exp_pmc_calib.hip -o /tmp/calib
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/calib_$c -o calib --output-format csv -- /tmp/calib > /tmp/calib_$c.log 2>&1
  python3 - "$c" <<'PY'
import csv, glob, sys, collections
c = sys.argv[1]
tot = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(f"/tmp/calib_{c}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            k = r["Kernel_Name"].split("(")[0]
            tot[k][0] += float(r["Counter_Value"]); tot[k][1] += 1
for k, (v, n) in sorted(tot.items()):
    print(f"{c} {k:12s} launches {n}  per launch {v / n:.1f} (raw counter units)")
PY
done
tail -1 /tmp/calib_WRITE_SIZE.log
