"""ctypes binding of libw4a16.so (the C ABI declared in include/w4a16.h).

The library is built in-tree by `__graft_entry__.build()` (hipcc, gfx950).  There
is no fallback of any kind: if the shared object is missing or lacks a symbol,
importing this module raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_int, c_size_t, c_void_p

# torch first: the PyTorch-ROCm wheel carries its own libamdhip64.so.  Loading it before
# libw4a16.so makes the dynamic loader bind both to that ONE HIP runtime (same SONAME); in the
# other order the process ends up with two runtimes and launches fail with hipErrorNoDevice.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# W4A16_LIBRARY points at another build of the same ABI (kernel experiments); default: the in-tree build.
LIB_PATH = os.environ.get("W4A16_LIBRARY") or os.path.join(_HERE, "csrc", "libw4a16.so")

ABI_VERSION = 1

# name -> (restype, argtypes); must list every symbol include/w4a16.h declares.
SYMBOLS = {
    "w4a16_strerror": (c_char_p, [c_int]),
    "w4a16_last_hip_error": (c_int, []),
    "w4a16_abi_version": (c_int, []),
    "w4a16_gemm_f16": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "w4a16_dequantize_f16": (c_int, [c_void_p] * 4 + [c_int] * 3 + [c_void_p]),
    "w4a16_packed_weight_bytes": (c_size_t, [c_int, c_int]),
    "w4a16_packed_sz_bytes": (c_size_t, [c_int, c_int, c_int]),
    "w4a16_repack": (c_int, [c_void_p] * 5 + [c_int] * 3 + [c_void_p]),
    "w4a16_gemm_packed_f16": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p]),
    "w4a16_gemm_packed_f16_cfg": (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_void_p]),
    "w4a16_gemm_packed_f16_tiled_cfg": (c_int, [c_void_p] * 5 + [c_int] * 8 + [c_void_p]),
    "w4a16_gemm_packed_f16_wide_cfg": (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_void_p]),
    "w4a16_gemm_packed_f16_batched": (c_int, [c_void_p, ctypes.c_longlong] + [c_void_p] * 4 +
                                      [ctypes.c_longlong] + [c_int] * 5 + [c_void_p]),
    "w4a16_gemm_packed_f16_batched_cfg": (c_int, [c_void_p, ctypes.c_longlong] + [c_void_p] * 4 +
                                          [ctypes.c_longlong] + [c_int] * 9 + [c_void_p]),
    "w4a16_gemm_packed_f16_ex": (c_int, [c_void_p, ctypes.c_longlong] + [c_void_p] * 4 + [ctypes.c_longlong] +
                                 [c_int] * 10 + [c_void_p]),
    "w4a16_gemm_packed_f16_masked": (c_int, [c_void_p, ctypes.c_longlong] + [c_void_p] * 4 + [ctypes.c_longlong] +
                                     [c_int] * 6 + [c_void_p, c_int, c_void_p]),
    "w4a16_dequantize_packed_f16": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_void_p]),
    "w4a16_gemm_packed_f16_decode_cfg": (c_int, [c_void_p, ctypes.c_longlong] + [c_void_p] * 4 + [ctypes.c_longlong] +
                                         [c_int] * 10 + [c_void_p, c_size_t, c_void_p]),
    "w4a16_decode_workspace_bytes": (c_size_t, [c_int] * 7),
    "w4a16_gemm_packed_f16_wide_split_cfg": (c_int, [c_void_p] * 5 + [c_int] * 9 + [c_void_p, c_size_t, c_void_p]),
    "w4a16_wide_workspace_bytes": (c_size_t, [c_int] * 7),
    "w4a16_workspace_bytes": (c_size_t, [c_int] * 6),
    "w4a16_gemm_packed_f16_ws": (c_int, [c_void_p, ctypes.c_longlong] + [c_void_p] * 4 + [ctypes.c_longlong] +
                                 [c_int] * 6 + [c_void_p, c_size_t, c_void_p]),
    "w4a16_launch_shape_ws": (c_int, [c_int] * 6 + [ctypes.POINTER(c_int)]),
    "w4a16_static_scratch_bytes": (c_size_t, []),
    "w4a16_gemm_packed_f16_decode_static": (c_int, [c_void_p, ctypes.c_longlong] + [c_void_p] * 4 +
                                            [ctypes.c_longlong] + [c_int] * 10 + [c_void_p]),
    "w4a16_launch_shape": (c_int, [c_int] * 5 + [ctypes.POINTER(c_int)]),
    "w4a16_built_shape": (c_int, [c_int, c_int, ctypes.POINTER(c_int)]),
    "w4a16_stream_read": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),
    "w4a16_gemm_packed_bf16": (c_int, [c_void_p] * 5 + [c_int] * 8 + [c_void_p]),
    "w4a16_built_shape_bf16": (c_int, [c_int, ctypes.POINTER(c_int)]),
    "w4a16_moe_route_f16": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p]),
    "w4a16_moe_combine_f16": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_void_p]),
}

# Return codes that mirror the reference's std::invalid_argument (-> ValueError).
_VALUE_ERRORS = {-1, -2, -3, -5}


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no non-HIP fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover - build mismatch
            if os.environ.get("W4A16_LIBRARY"):
                continue  # an experiment build of an older revision: the symbol is simply not callable
            raise ImportError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.w4a16_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI {lib.w4a16_abi_version()} != expected {ABI_VERSION}")
    return lib


lib = _load()


KERNEL_NAMES = {0: "gemm_packed_kernel", 1: "gemm_tiled_kernel", 2: "gemm_wide_kernel", 3: "gemm_wide_shared_kernel",
                4: "gemm_decode_kernel"}


def launch_shape(M: int, K: int, N: int, batch: int = 1, epilogue: int = 0) -> str:
    """The kernel and template arguments the library picks for a problem, as rocprof prints them."""
    buf = (c_int * 5)()
    check(lib.w4a16_launch_shape(M, K, N, batch, epilogue, buf), "launch_shape")
    kind, mt, t, a, b = list(buf)
    args = {0: (mt, t, a), 1: (mt, t, a, b), 2: (mt, t, b), 3: (mt, t, a), 4: (t, a)}[kind]
    return f"{KERNEL_NAMES[kind]}<{','.join(str(v) for v in args)}>"


def built_shapes(kernel: int):
    """Every launch shape of one kernel this build holds, as tuples of the _cfg entry points' arguments."""
    out, buf = [], (c_int * 4)()
    while lib.w4a16_built_shape(kernel, len(out), buf) == 0:
        out.append(tuple(buf)[:4 if kernel == 1 else 3])
    return out


def check(code: int, what: str) -> None:
    """Raise the Python exception that corresponds to a C return code."""
    if code == 0:
        return
    msg = lib.w4a16_strerror(code).decode()
    if code in _VALUE_ERRORS:
        raise ValueError(f"{what}: {msg}")
    if code == -7:
        raise RuntimeError(f"{what}: {msg} (hipError {lib.w4a16_last_hip_error()})")
    raise RuntimeError(f"{what}: {msg}")
