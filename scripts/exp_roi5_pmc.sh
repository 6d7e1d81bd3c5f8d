#!/bin/bash
# PMC counters of every roi_align_kernel dispatch of scripts/exp_roi5.py, in dispatch order
# (0: inside the engine pass = locality order; 1: list order; 2: locality order; then the timing loops). One rocprofv3 pass per group.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for c in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c -d /tmp/roi_$i -o roi --output-format csv -- python3 $R/scripts/exp_roi5.py > /tmp/roi_$i.log 2>&1
  python3 - "$i" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
rows = collections.OrderedDict()
for f in glob.glob(f"/tmp/roi_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "roi_align_kernel" in r["Kernel_Name"]:
            rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for i, (d, v) in enumerate(sorted(rows.items())[1:3]):
    print("list order    " if i == 0 else "locality order", {k: f"{x:.4g}" for k, x in v.items()})
PY
done
