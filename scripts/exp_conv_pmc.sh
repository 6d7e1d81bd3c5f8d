#!/bin/bash
# Per-kernel SQ counters of the conv family on the bench workload (eager single stream): exp_conv_pmc.sh "<counters>" ...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for c in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c -d /tmp/cv_$i -o cv --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-step --no-pmc --streams 1 --no-graph > /tmp/cv_$i.log 2>&1
  python3 - "$i" <<'PY'
import csv, glob, sys, collections, re
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"/tmp/cv_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_igemm64" not in k: continue
        m = re.search(r"Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)", k)
        key = "x".join(m.groups()[:2]) + f" w{m.group(3)}x{m.group(4)} epi{m.group(5)} two{m.group(6)}" if m else k[:40]
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(agg.items()):
    print(k, {a: f"{b:.4g}" for a, b in v.items()})
PY
done
