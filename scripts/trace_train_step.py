"""Timeline of ONE training iteration from a rocprofv3 --kernel-trace CSV of `bench.py --train-only`: per queue (stream) the busy time,
the first / last kernel, and what runs on the device towards the end of the iteration (the tail that nothing overlaps).
python scripts/trace_train_step.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# an iteration ends with its SGD launch(es) (one sgd_multi_kernel, or a burst of sgd_kernel): take the window between the last two
sgd = [int(r["End_Timestamp"]) for r in rows if "sgd_kernel" in r["Kernel_Name"] or "sgd_multi_kernel" in r["Kernel_Name"]]
bursts = []
for t in sgd:
    if not bursts or t - bursts[-1][1] > 2e6:
        bursts.append([t, t])
    else:
        bursts[-1][1] = t
t0, t1 = bursts[-2][1], bursts[-1][1]
it = [r for r in rows if t0 < int(r["Start_Timestamp"]) <= t1]
print(f"iteration window {(t1 - t0) / 1e6:.2f} ms, {len(it)} kernels")
byq = defaultdict(list)
for r in it:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Kernel_Name"].split("(")[0][:60]))
for q, ks in sorted(byq.items(), key=lambda kv: kv[1][0][0]):
    busy = sum(e - s for s, e, _ in ks) / 1e6
    print(f"queue {q}: {len(ks):4d} kernels, first at {ks[0][0] / 1e6:6.2f} ms ({ks[0][2][:30]}), last ends {ks[-1][1] / 1e6:6.2f} ms ({ks[-1][2][:30]}), busy {busy:6.2f} ms")
# occupancy over time: number of queues with a kernel running, in 1 ms bins
nb = int((t1 - t0) / 1e6) + 1
for q, ks in sorted(byq.items(), key=lambda kv: kv[1][0][0]):
    line = []
    for b in range(nb):
        lo, hi = b * 1e6, (b + 1) * 1e6
        cov = sum(max(0, min(e, hi) - max(s, lo)) for s, e, _ in ks) / 1e6
        line.append("#" if cov > 0.66 else ("+" if cov > 0.33 else ("." if cov > 0.02 else " ")))
    print(f"queue {q:>3s} |{''.join(line)}|")
# --list: every kernel of the busiest queue in launch order with the idle gap in front of it (where the critical path waits)
if "--list" in sys.argv:
    q, ks = max(byq.items(), key=lambda kv: sum(e - s for s, e, _ in kv[1]))
    print(f"queue {q} in order: start ms, duration us, gap before us, kernel")
    prev = None
    gaps = 0.0
    for s, e, n in ks:
        gap = (s - prev) / 1e3 if prev is not None else 0.0
        gaps += max(gap, 0.0)
        print(f"{s / 1e6:7.3f} {(e - s) / 1e3:8.1f} {gap:7.1f}  {n}")
        prev = e
    print(f"sum of gaps {gaps / 1e3:.2f} ms")
