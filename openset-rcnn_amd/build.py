"""Builds libosr_hip.so (gfx950) in-tree with hipcc. No JIT cache, no torch extension machinery: the library
is a plain C-ABI shared object (include/osr.h) that the host side loads with ctypes."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libosr_hip.so")
ARCH = "gfx950"

# source -> extra flags. The "exact" kernels are built without FMA contraction so that their fp32 arithmetic
# rounds like the CPU oracle's (bit-exact top-k / NMS indices).
SOURCES = {
    "osr_status.hip": [],
    "osr_preproc_pool.hip": ["-ffp-contract=off"],
    "osr_resize.hip": [],
    "osr_conv_gemm.hip": [],
    "osr_conv_gemm64.hip": [],
    "osr_conv_f32.hip": [],
    "osr_bottleneck.hip": [],
    "osr_stem_pool.hip": [],
    "osr_rpn.hip": ["-ffp-contract=off"],
    "osr_roi_align.hip": ["-ffp-contract=off"],
    "osr_det_tail.hip": ["-ffp-contract=off"],
    "osr_train_fwd.hip": ["-ffp-contract=off"],
    "osr_rpn_sparse.hip": ["-ffp-contract=off"],
    "osr_multi_tensor.hip": [],  # (same contraction setting as osr_train_bwd.hip: the multi-tensor SGD must round like osr_sgd_step)
    "osr_conv_bwd.hip": [],
    "osr_train_bwd.hip": [],
}
COMMON = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"] + \
    os.environ.get("OSR_EXTRA_HIPCC_FLAGS", "").split()  # diagnostic builds only (e.g. -DC64_STAMPS)


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (needed to build libosr_hip.so for gfx950)")


def _digest(path: str, flags) -> str:
    h = hashlib.sha1()
    # every header under csrc/ (osr_common.h, osr_pln_dist.h, osr_box_loss.h, ...) and the public header: an edit to any of them rebuilds
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    for f in [path] + headers + [os.path.join(HERE, "..", "include", "osr.h")]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(COMMON + list(flags)).encode())
    return h.hexdigest()


def _compile(src: str, flags, verbose: bool) -> str:
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    stamp = obj + ".sha1"
    dig = _digest(path, flags)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj
    cmd = [_hipcc()] + COMMON + list(flags) + ["-c", path, "-o", obj]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    with open(stamp, "w") as fh:
        fh.write(dig)
    return obj


def build(verbose: bool = False, force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(lambda kv: _compile(kv[0], kv[1], verbose), SOURCES.items()))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        # clang's host pass silently drops a __global__ template whose body it cannot check (seen with a type-dependent
        # argument of an amdgcn builtin): the launch stub then stays undefined and dlopen fails on the GPU box. Catch it here.
        und = subprocess.run(["nm", "-D", "--undefined-only", LIB], capture_output=True, text=True).stdout
        bad = [l.split()[-1] for l in und.splitlines() if "__device_stub__" in l]
        if bad:
            os.remove(LIB)
            raise RuntimeError(f"{len(bad)} kernel launch stubs are undefined in {os.path.basename(LIB)} (host pass dropped them), e.g. {bad[0]}")
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
