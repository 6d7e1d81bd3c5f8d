"""ctypes binding of oracle/_build/libosr_oracle.so (TEST INFRASTRUCTURE ONLY; see osr_oracle_c.c)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libosr_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "osr_oracle_c.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.osr_oracle_nms.restype = ctypes.c_int64
        _lib.osr_oracle_batched_nms.restype = ctypes.c_int64
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def roi_align(feat: torch.Tensor, rois: torch.Tensor, scale: float, out_size: int = 7,
              sampling_ratio: int = 0, aligned: bool = True) -> torch.Tensor:
    f = np.ascontiguousarray(feat.detach().float().numpy())
    r = np.ascontiguousarray(rois.detach().float().numpy())
    n, c, h, w = f.shape
    out = np.zeros((r.shape[0], c, out_size, out_size), dtype=np.float32)
    lib().osr_oracle_roi_align(_p(f), n, c, h, w, _p(r), r.shape[0], ctypes.c_float(scale), out_size,
                               sampling_ratio, int(aligned), _p(out))
    return torch.from_numpy(out)


def argsort_desc(v: np.ndarray) -> np.ndarray:
    v = np.ascontiguousarray(v, dtype=np.float32)
    o = np.zeros(v.shape[0], dtype=np.int64)
    lib().osr_oracle_argsort_desc(_p(v), ctypes.c_int64(v.shape[0]), _p(o))
    return o


def nms(boxes: np.ndarray, scores: np.ndarray, thr: float) -> np.ndarray:
    b = np.ascontiguousarray(boxes, dtype=np.float32)
    s = np.ascontiguousarray(scores, dtype=np.float32)
    keep = np.zeros(max(b.shape[0], 1), dtype=np.int64)
    k = lib().osr_oracle_nms(_p(b), _p(s), ctypes.c_int64(b.shape[0]), ctypes.c_float(thr), _p(keep))
    return keep[:k]


def batched_nms(boxes: np.ndarray, scores: np.ndarray, cls: np.ndarray, thr: float) -> np.ndarray:
    b = np.ascontiguousarray(boxes, dtype=np.float32)
    s = np.ascontiguousarray(scores, dtype=np.float32)
    c = np.ascontiguousarray(cls, dtype=np.int64)
    keep = np.zeros(max(b.shape[0], 1), dtype=np.int64)
    k = lib().osr_oracle_batched_nms(_p(b), _p(s), _p(c), ctypes.c_int64(b.shape[0]), ctypes.c_float(thr), _p(keep))
    return keep[:k]


def set_threads(k: int) -> None:
    """OpenMP thread count of the C half (bench.py's cpu_baseline times a 1-thread and an all-core point)."""
    lib().omp_set_num_threads(int(k))
