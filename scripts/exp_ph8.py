"""Experiment driver (not part of the product; needs a -DOSR_EXPERIMENT build): the 8-phase K loop of the 256 x 256 conv / FC kernel
(OSR_CONV_PH8=1) against the one-barrier-per-slice loop (=0) in ONE process: (i) outputs bit-identical, (ii) a race screen -- repeated launches
at several shapes, alone and beside a second stream that hammers memory, every output compared with the first -- and (iii) interleaved
timing rounds on random operands (median and min, HIP events per launch)."""
import os, sys, math, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); pkg._lib.load()
from openset_rcnn_amd.host import ops
g = torch.Generator().manual_seed(0)
ROUNDS = int(os.environ.get("ROUNDS", 7)); REPS = int(os.environ.get("REPS", 10)); SCREEN = int(os.environ.get("SCREEN", 40))


def set_ph8(v):
    os.environ["OSR_CONV_PH8"] = str(v)


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def conv_case(n, h, w, cin, cout, k, out_f32=False):
    x = (torch.randn(n, h, w, cin, generator=g) * 0.5).half().cuda()
    wt = (torch.randn(cout, k, k, cin, generator=g) / math.sqrt(k * k * cin)).half().cuda()
    b = torch.randn(cout, generator=g).cuda()
    fl = 2.0 * n * h * w * cout * k * k * cin
    return (lambda: ops.conv2d(x, wt, b, 1, k // 2, relu=True, out_dtype=torch.float32 if out_f32 else None)), fl


def fc_case(m, k, nout, seg=None):
    x = (torch.randn(m, k, generator=g) * 0.5).half().cuda()
    wt = (torch.randn(nout, k, generator=g) / math.sqrt(k)).half().cuda()
    b = torch.randn(nout, generator=g).cuda()
    rs = None
    if seg:
        counts = torch.tensor(seg[0], dtype=torch.int32).cuda()
        rs = (counts, seg[1])
    return (lambda: ops.linear(x, wt, b, relu=True, row_seg=rs)), 2.0 * m * k * nout


def head_case(n, h, w):
    x = (torch.randn(n, h, w, 256, generator=g) * 0.5).half().cuda()
    wt = (torch.randn(256, 3, 3, 256, generator=g) / math.sqrt(2304)).half().cuda()
    b = torch.randn(256, generator=g).cuda() * 0.1
    wtail = (torch.randn(5, 256, generator=g) * 0.05).cuda()
    btail = torch.randn(5, generator=g).cuda() * 0.1
    def run():
        r = ops.cfrpn_head_fused(x, wt, b, wtail, btail)
        return torch.cat([r[0].reshape(-1), r[1].reshape(-1)])
    return run, 2.0 * n * h * w * 256 * 2304


small = os.environ.get("SMALL", "0") == "1"
cases = {
    "fpn_output2 3x3 256->256 @16x200x336": conv_case(16, 200, 336, 256, 256, 3),
    "fpn_output3 3x3 256->256 @16x100x168": conv_case(16, 100, 168, 256, 256, 3),
    "fc1 68368x12544->1024": fc_case(68368, 12544, 1024),
    "fc1 ragged lists (16 x 4273 slots)": fc_case(68368, 12544, 1024, seg=([4273, 3000, 4000, 100, 0, 4273, 2500, 3999, 4273, 1, 255, 257, 4100, 3800, 2900, 4273], 4273)),
    "rpn head p2 @16x200x336": head_case(16, 200, 336),
    "rpn head p3 @16x100x168": head_case(16, 100, 168),
    "odd K tiles: 1x1 448->256 @4x100x168": conv_case(4, 100, 168, 448, 256, 1),
    "single K tile: 1x1 64->256 @8x100x168": conv_case(8, 100, 168, 64, 256, 1),
    "1x1 1024->512 f32 out @2x131x67": conv_case(2, 131, 67, 1024, 512, 1, out_f32=True),
}
if small:
    cases = {k: v for k, v in cases.items() if "fc1" not in k}

VARS = [int(v) for v in os.environ.get("VARS", "0 1 2 3").split()]  # OSR_CONV_PH8: 0 = one barrier per slice, 1.. = 8-phase, piece placement 0 / 1 / 2
print(f"{'case':44s} {'identical':>9s} {'screen':>8s}  " + "  ".join(f"v{v} us med/min (TF/s)".rjust(26) for v in VARS), flush=True)
hammer = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
side = torch.cuda.Stream()
os.environ["OSR_CONV_FORCE_TILE"] = "3"  # the 256 x 256 tile wherever it fits (diagnostic knob)
for name, (fn, fl) in cases.items():
    ragged = "ragged" in name
    def run():
        out = fn()
        if ragged:  # rows beyond a list's count are not written: compare the data rows only
            keep = torch.zeros(68368, dtype=torch.bool, device="cuda")
            for i, c in enumerate([4273, 3000, 4000, 100, 0, 4273, 2500, 3999, 4273, 1, 255, 257, 4100, 3800, 2900, 4273]): keep[i * 4273:i * 4273 + c] = True
            out = out[keep]
        return out
    set_ph8(0); ref = run().clone(); torch.cuda.synchronize()
    same, bad = True, 0
    for v in VARS[1:]:
        set_ph8(v); new = run().clone(); torch.cuda.synchronize()
        same &= torch.equal(ref, new)
        for i in range(SCREEN):
            if i % 2:
                with torch.cuda.stream(side):
                    hammer.fill_(i & 255); hammer.add_(1)
            if not torch.equal(run(), ref): bad += 1
    torch.cuda.synchronize()
    ts = {v: [] for v in VARS}
    for _ in range(ROUNDS):
        for v in VARS:
            set_ph8(v); fn(); ts[v].append(timed(fn, REPS))
    print(f"{name:44s} {str(same):>9s} {bad:3d}/{SCREEN * (len(VARS) - 1):<4d}  " +
          "  ".join(f"{statistics.median(ts[v]):9.1f}/{min(ts[v]):8.1f} ({fl / statistics.median(ts[v]) / 1e6:5.0f})" for v in VARS), flush=True)
