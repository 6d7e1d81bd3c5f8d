import csv,glob,sys
from collections import defaultdict
agg=defaultdict(dict)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "roi_align_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].setdefault(r["Dispatch_Id"], float(r["Counter_Value"]))
for k,v in sorted(agg.items()):
    vals=list(v.values()); print(f"{k:28s} {vals[0]:.4g}")
