#!/bin/bash
# Per-layer time of every conv / FC launch under each forced tile configuration (OSR_CONV_FORCE_TILE, diagnostic knob of
# osr_conv_gemm64.hip), single stream, for the given batch sizes. Output: gpurun_out/tiles_b<batch>_f<cfg>.log
for B in ${BATCHES:-16 8}; do
  for F in ${CFGS:-0 1 2 3 4 5 6 7}; do
    OSR_CONV_FORCE_TILE=$F python3 bench.py --batch $B --steps 3 --warmup 2 --no-cpu-baseline --streams 1 --no-graph --layers > /dev/null 2> gpurun_out/tiles_b${B}_f${F}.log
  done
done
python3 - <<'PY'
import re, glob, os
names = {0: "model", 1: "128x128/1", 2: "128x128/2", 3: "256x256/2", 4: "128x256/1", 5: "256x128/1", 6: "128x64/1", 7: "128x64/2"}
cfgs = [int(x) for x in os.environ.get("CFGS", "0 1 2 3 4 5 6 7").split()]
for B in os.environ.get("BATCHES", "16 8").split():
    tab = {}
    order = []
    for F in cfgs:
        idx = 0
        for line in open(f"gpurun_out/tiles_b{B}_f{F}.log"):
            m = re.match(r"\s+(\S+)\s+([\d.]+) us", line)
            if m:
                key = (idx, m.group(1)); idx += 1
                if F == 0: order.append(key)
                tab.setdefault(key, {})[F] = float(m.group(2))
    print(f"== batch {B}: us per launch; columns " + " ".join(f"{names[F]:>11s}" for F in cfgs))
    tot0 = totb = 0.0
    for key in order:
        row = tab[key]
        best = min(row, key=row.get)
        tot0 += row[0]; totb += row[best]
        print(f"{key[1][-34:]:34s} " + " ".join(f"{row.get(F, float('nan')):11.1f}" for F in cfgs) + f"   best={names[best]} ({row[best] / row[0]:.2f})")
    print(f"total heuristic {tot0:.0f} us, per-layer best {totb:.0f} us")
PY
