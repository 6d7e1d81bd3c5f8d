"""Per-launch L2 hit rate of the conv family from a `rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum` pass (diagnostic):
groups the dispatches of the last pass of the path by kernel and grid size."""
import csv, glob, sys
from collections import defaultdict, OrderedDict
sys.path.insert(0, "scripts")
from pmc_summary import short

rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
d = OrderedDict()
for r in rows:
    key = (int(r["Dispatch_Id"]), short(r["Kernel_Name"]), int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
    d.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
agg = defaultdict(lambda: [0, 0.0, 0.0])
for (did, k, g), c in d.items():
    a = agg[(k, g)]
    a[0] += 1; a[1] += c.get("TCC_HIT_sum", 0); a[2] += c.get("TCC_MISS_sum", 0)
for (k, g), (n, h, m) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    if h + m < 1e5: continue
    print(f"{k:90s} wgs={g:6d} launches={n:4d} req/launch={(h + m) / n / 1e6:8.2f}M  hit={h / (h + m):.3f}")
