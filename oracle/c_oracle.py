"""ctypes wrapper of oracle/awq_oracle.c (test infrastructure; see that file's header)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libawq_oracle.so")


def build() -> str:
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return LIB_PATH


def _load():
    if not os.path.exists(LIB_PATH):
        build()
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.awq_c_dequantize.argtypes = [vp, vp, vp, vp, ci, ci, ci]
    lib.awq_c_dequantize.restype = None
    lib.awq_c_gemm.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci]
    lib.awq_c_gemm.restype = ci
    lib.awq_c_f2h.argtypes = [ctypes.c_float]
    lib.awq_c_f2h.restype = ctypes.c_uint16
    lib.awq_c_h2f.argtypes = [ctypes.c_uint16]
    lib.awq_c_h2f.restype = ctypes.c_float
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _p(a: np.ndarray) -> int:
    return a.ctypes.data


def dequantize(qweight, scales, qzeros) -> np.ndarray:
    qweight = np.ascontiguousarray(qweight, dtype=np.int32)
    scales = np.ascontiguousarray(scales, dtype=np.float16)
    qzeros = np.ascontiguousarray(qzeros, dtype=np.int32)
    K, NC = qweight.shape
    N, G = NC * 8, K // scales.shape[0]
    out = np.empty((K, N), dtype=np.float16)
    lib().awq_c_dequantize(_p(qweight), _p(scales), _p(qzeros), _p(out), K, N, G)
    return out


def gemm(x, qweight, scales, qzeros, split_k: int = 8, faithful: bool = True) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float16)
    qweight = np.ascontiguousarray(qweight, dtype=np.int32)
    scales = np.ascontiguousarray(scales, dtype=np.float16)
    qzeros = np.ascontiguousarray(qzeros, dtype=np.int32)
    M, K = x.shape
    N, G = qweight.shape[1] * 8, K // scales.shape[0]
    out = np.empty((M, N), dtype=np.float16)
    rc = lib().awq_c_gemm(_p(x), _p(qweight), _p(scales), _p(qzeros), _p(out), M, K, N, G, split_k, int(faithful))
    if rc:
        raise MemoryError("awq_c_gemm")
    return out
