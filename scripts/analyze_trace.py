"""Gap analysis of a rocprofv3 --kernel-trace CSV (not part of the product): per queue, the time between the end of a kernel and
the start of the next one, and the union busy time of the GPU, over the last replayed step(s)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
win = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0  # ms window at the end of the trace
rows = [r for r in rows if int(r["Start_Timestamp"]) > t_end - win * 1e6]
byq = defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6
print(f"window {span:.2f} ms, {len(rows)} kernels, {len(byq)} queues")
for q, ks in byq.items():
    busy = sum(e - s for s, e, _ in ks) / 1e6
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    small = [g for g in gaps if 0 <= g < 200e3]
    print(f"queue {q}: {len(ks)} kernels, busy {busy:.2f} ms, sum of gaps < 200 us: {sum(small) / 1e6:.2f} ms (mean {sum(small) / max(len(small), 1) / 1e3:.1f} us), overlaps {sum(1 for g in gaps if g < 0)}")
# union busy
ev = sorted([(int(r["Start_Timestamp"]), 1) for r in rows] + [(int(r["End_Timestamp"]), -1) for r in rows])
busy, depth, last = 0, 0, ev[0][0]
for t, d in ev:
    if depth > 0:
        busy += t - last
    depth += d
    last = t
print(f"GPU busy (any kernel running) {busy / 1e6:.2f} ms of {span:.2f} ms = {100 * busy / 1e6 / span:.1f} %")
