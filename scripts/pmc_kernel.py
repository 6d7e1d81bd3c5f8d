"""Per-kernel averages (per dispatch) of every counter found in the rocprofv3 --pmc output directories given on the command
line, plus the average dispatch duration of the same pass: `python scripts/pmc_kernel.py <name filter> dir1 dir2 ...`."""
import csv, glob, sys
from collections import defaultdict

flt = sys.argv[1]
for d in sys.argv[2:]:
    cnt = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if flt not in k:
                continue
            k = k.split("(")[0][:70]
            cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    dur = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if flt not in k:
                continue
            k = k.split("(")[0][:70]
            dur[k][0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); dur[k][1] += 1
    for k, c in cnt.items():
        n = max(len(disp[k]), 1)
        us = dur[k][0] / max(dur[k][1], 1) / 1e3
        print(f"{d}: {k}  dispatches {n}  avg {us:.1f} us  " + "  ".join(f"{name}={v / n:.4g}" for name, v in sorted(c.items())))
