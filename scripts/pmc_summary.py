"""Summarise the rocprofv3 outputs of scripts/profile_round.sh: per-kernel launches / average duration (kernel-trace stats)
and HBM bytes per 16-image step from the FETCH_SIZE / WRITE_SIZE passes. FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950
reports half of the bytes of wide coalesced reads); both counters are in KiB-like units of 1024 B... rocprofv3 reports them
in kilobytes (counter definition: bytes / 1024)."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name):
    m = re.match(r"_Z\d+(conv_igemm64_kernel)I(DF16_|DF16b)(DF16_|DF16b|f)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E(?:Li(\d+)E)?", name)
    if m:
        split = ",split-K tail" if m.group(10) == "1" else ""
        return f"conv_igemm64<{m.group(4)}x{m.group(5)},{m.group(6)}x{m.group(7)} waves,epi{m.group(8)},{'8-phase' if m.group(9) == '2' else 'stages' + str(1 + int(m.group(9)))},out={'f32' if m.group(3) == 'f' else 'same'}{split}>"
    m = re.match(r"_Z\d+(bottleneck64_kernel)I(DF16_|DF16b)Li(\d+)ELi(\d)E", name)
    if m:
        return f"bottleneck64<cin {m.group(3)},{'projection' if m.group(4) == '1' else 'identity'}>"
    m = re.match(r"_Z\d+([a-z_0-9]+?)(I|E|P|v)", name)
    return m.group(1) if m else name.split("(")[0][:60]


def counter_sum(path, steps):
    tot, calls = defaultdict(float), defaultdict(int)
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            tot[k] += float(row["Counter_Value"])
            calls[k] += 1
    return {k: (v / steps, calls[k] / steps) for k, v in tot.items()}


def mfma_pass(path):
    """Per kernel: sum of SQ_VALU_MFMA_BUSY_CYCLES (all SIMDs), GRBM_GUI_ACTIVE (sum over the 8 XCDs), MFMA MOPS (x512 = FLOP) and
    the dispatch durations of the same pass (kernel trace), over every dispatch of the kernel."""
    cnt = defaultdict(lambda: defaultdict(float))
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            cnt[short(row["Kernel_Name"])][row["Counter_Name"]] += float(row["Counter_Value"])
    dur = defaultdict(float)
    for f in glob.glob(path + "/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            dur[short(row["Kernel_Name"])] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    out = {}
    for k, c in cnt.items():
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # cycles of the dispatches (the counter is summed over the XCDs)
        if gui <= 0:
            continue
        out[k] = dict(mfma_util=round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024.0), 4),  # fraction of SIMD cycles with the matrix pipe busy
                      clock_GHz=round(gui / dur[k], 3) if dur.get(k) else None,
                      mfma_TFLOP=round(c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) * 512 / 1e12, 3))
    return out


def main():
    out_dir, steps = sys.argv[1], float(sys.argv[2])
    stats = {}
    for f in glob.glob(out_dir + "/stats/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Name"])
            c, t = int(row["Calls"]), float(row["TotalDurationNs"])
            if k in stats:
                c, t = c + stats[k][0], t + stats[k][1]
            stats[k] = (c, t)
    fetch, write = counter_sum(out_dir + "/pmc_fetch", steps), counter_sum(out_dir + "/pmc_write", steps)
    mfma = mfma_pass(out_dir + "/pmc_mfma")
    kernels = {}
    for k, (c, t) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        if t / steps < 20e3 and not k.startswith("splitk_reduce"):
            continue
        kernels[k] = dict(launches_per_step=round(c / steps, 2), ms_per_step=round(t / steps / 1e6, 3), avg_us=round(t / c / 1e3, 1),
                          fetch_GB_per_step_x2corrected=round(2 * fetch.get(k, (0, 0))[0] * 1024 / 1e9, 3),
                          write_GB_per_step=round(write.get(k, (0, 0))[0] * 1024 / 1e9, 3))
        if k in mfma:
            kernels[k].update(mfma_util=mfma[k]["mfma_util"], clock_GHz=mfma[k]["clock_GHz"], mfma_executed_TFLOP_per_step=round(mfma[k]["mfma_TFLOP"] / steps, 3))
    conv = [v for k, v in kernels.items() if k.startswith("conv_igemm") or k.startswith("bottleneck64") or k.startswith("splitk_reduce") or k.startswith("stem_pool")]
    print(json.dumps(dict(
        note="scripts/profile_round.sh: rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE, --pmc WRITE_SIZE and --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 in separate passes of "
             "`bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-step --streams 1 --no-graph` (`steps` passes of the path each: 2 warm-up + 5 timed + the attribution passes + the 4-image calibration pass); FETCH_SIZE doubled per "
             "MI355X_MICROARCH.md; GB per 16-image step; mfma_util = MFMA busy cycles / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), clock_GHz = GRBM_GUI_ACTIVE / 8 / dispatch time of the counter pass",
        conv_family=dict(ms_per_step=round(sum(v["ms_per_step"] for v in conv), 3),
                         fetch_GB_per_step_x2corrected=round(sum(v["fetch_GB_per_step_x2corrected"] for v in conv), 3),
                         write_GB_per_step=round(sum(v["write_GB_per_step"] for v in conv), 3),
                         mfma_util_time_weighted=round(sum(v.get("mfma_util", 0.0) * v["ms_per_step"] for v in conv) / max(sum(v["ms_per_step"] for v in conv), 1e-9), 4),
                         mfma_executed_TFLOP_per_step=round(sum(v.get("mfma_executed_TFLOP_per_step", 0.0) for v in conv), 3)),
        roi_align=next((v for k, v in kernels.items() if k.startswith("roi_align")), {}),
        kernels=kernels), indent=1))


if __name__ == "__main__":
    main()
