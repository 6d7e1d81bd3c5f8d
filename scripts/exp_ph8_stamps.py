"""Diagnostic (not part of the product; needs OSR_EXTRA_HIPCC_FLAGS="-DC64_STAMPS -DOSR_EXPERIMENT"): shader-clock cycles of the sections of one
K tile of the 8-phase loop (wave 0 = leading group, wave 4 = trailing group), median over workgroups."""
import ctypes as C, os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); lib = pkg._lib.load()
from openset_rcnn_amd.host import ops
lib.osr_debug_set_p8_stamps.argtypes = [C.c_void_p]; lib.osr_debug_set_p8_stamps.restype = None
os.environ["OSR_CONV_FORCE_TILE"] = "3"
g = torch.Generator().manual_seed(0)
names = ["reads", "issue", "wait+bar a", "mfma issue", "bar b"]
def run(name, fn, nblocks):
    st = torch.zeros(nblocks * 2 * 24, dtype=torch.int64, device="cuda")
    for _ in range(3): fn()
    torch.cuda.synchronize()
    lib.osr_debug_set_p8_stamps(C.c_void_p(st.data_ptr())); fn(); torch.cuda.synchronize(); lib.osr_debug_set_p8_stamps(None)
    s = st.view(nblocks, 2, 24).cpu().double()
    ok = s[:, 0, 20] > 0
    s = s[ok]
    print(f"{name}: {int(ok.sum())} workgroups stamped; K tile = {float((s[:, 0, 20] - s[:, 0, 0]).median()):.0f} cycles (leading), {float((s[:, 1, 20] - s[:, 1, 0]).median()):.0f} (trailing)")
    for grp in (0, 1):
        for ph in range(4):
            d = [float((s[:, grp, ph * 5 + i + 1] - s[:, grp, ph * 5 + i]).median()) for i in range(5)]
            print(f"   group {grp} phase {ph + 1}: " + "  ".join(f"{n} {v:5.0f}" for n, v in zip(names, d)))
x = (torch.randn(16, 200, 336, 256, generator=g) * 0.5).half().cuda()
wt = (torch.randn(256, 3, 3, 256, generator=g) / 48).half().cuda(); b = torch.randn(256, generator=g).cuda()
run("fpn_output2", lambda: ops.conv2d(x, wt, b, 1, 1, relu=True), 4200)
xf = (torch.randn(68368, 12544, generator=g) * 0.5).half().cuda()
wf = (torch.randn(1024, 12544, generator=g) / 112).half().cuda(); bf = torch.randn(1024, generator=g).cuda()
ops.SPLIT_K_TAIL = False
run("fc1", lambda: ops.linear(xf, wf, bf, relu=True), 1072)
