#!/bin/bash
# SQ counters of the fused stem kernel (scripts/exp_stem_ablate.py as the workload): exp_stem_pmc.sh "<counters>" "<counters>" ...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for c in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c -d /tmp/sp_$i -o sp --output-format csv -- python3 $R/scripts/exp_stem_ablate.py > /tmp/sp_$i.log 2>&1
  python3 - "$i" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(f"/tmp/sp_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "stem_pool_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k:32s} {sum(v) / len(v):.5g}  (per launch, {len(v)} launches)")
PY
done
