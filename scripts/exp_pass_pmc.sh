#!/bin/bash
# Per-kernel SQ counters over one eager single-stream inference pass of the bench workload: exp_pass_pmc.sh "<counters>" "<counters>" ...
# (separate rocprofv3 --pmc passes; prints, per kernel name, the counters summed over its launches in ONE pass)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for c in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pp_$i -o pp --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-step --no-pmc --no-parity --no-pcie --streams 1 --no-graph > /tmp/pp_$i.log 2>&1
done
python3 - "$i" <<'PY'
import csv, glob, sys, collections, re
n = int(sys.argv[1])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for t in range(1, n + 1):
    seen = collections.Counter()
    for f in glob.glob(f"/tmp/pp_{t}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            m = re.search(r"conv_igemm64_kernelI\w+?Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)", k)
            key = ("igemm64 %sx%s epi%s two%s split%s %s" % (m.group(1), m.group(2), m.group(5), m.group(6), m.group(7), "f32out" if "DF16_f" in k or "DF16bf" in k else "")) if m else re.sub(r"\(.*", "", k)[:44]
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
            seen[(key, r["Counter_Name"])] += 1
    for (key, cn), v in seen.items():
        calls[key] = max(calls[key], v)
names = sorted({c for v in agg.values() for c in v})
print("per kernel, summed over ALL launches of the run (counts are launches x passes); columns:", " ".join(names))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    print(f"{k:46s} n={calls[k]:4d} " + " ".join(f"{v.get(c, 0):11.4g}" for c in names))
PY
