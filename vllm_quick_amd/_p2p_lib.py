"""ctypes binding of libp2p_ar.so (the C ABI declared in include/p2p_allreduce.h).

Built in-tree by `__graft_entry__.build()`.  No fallback inside this module: if the shared object is
missing or lacks a symbol, importing it raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_int, c_int64, c_void_p

import torch  # noqa: F401  (first, so that this library binds to PyTorch's HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("P2P_AR_LIBRARY") or os.path.join(_HERE, "csrc", "libp2p_ar.so")

ABI_VERSION = 2
IPC_HANDLE_BYTES = 64
F32, F16, BF16 = 0, 1, 2
DTYPE_CODES = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}

# name -> (restype, argtypes); must list every symbol include/p2p_allreduce.h declares.
SYMBOLS = {
    "p2p_ar_strerror": (c_char_p, [c_int]),
    "p2p_ar_last_hip_error": (c_int, []),
    "p2p_ar_abi_version": (c_int, []),
    "p2p_ar_should": (c_int, [c_int64, c_int64, c_int, c_int]),
    "p2p_ar_mailbox_bytes": (c_int64, [c_int, c_int64]),
    "p2p_ar_create": (c_int, [c_int, c_int, c_int64, POINTER(c_void_p)]),
    "p2p_ar_ipc_handle": (c_int, [c_void_p, c_void_p]),
    "p2p_ar_connect": (c_int, [c_void_p, c_void_p]),
    "p2p_ar_connect_ptrs": (c_int, [c_void_p, POINTER(c_void_p)]),
    "p2p_ar_mailbox_ptr": (c_void_p, [c_void_p]),
    "p2p_ar_all_reduce": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "p2p_ar_set_timeout_ms": (c_int, [c_void_p, c_int]),
    "p2p_ar_status": (c_int, [c_void_p, POINTER(c_int), c_int]),
    "p2p_ar_poll": (c_int, [c_void_p, POINTER(c_int)]),
    "p2p_ar_destroy": (c_int, [c_void_p]),
}

_VALUE_ERRORS = {-1, -3, -4, -5}
E_TIMEOUT = -8


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover - build mismatch
            raise ImportError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.p2p_ar_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI {lib.p2p_ar_abi_version()} != expected {ABI_VERSION}")
    return lib


lib = _load()


def check(code: int, what: str) -> None:
    """Raise the Python exception that corresponds to a C return code (the reference throws
    std::invalid_argument / std::runtime_error, csrc/custom_all_reduce.cu:17-25,95-97)."""
    if code == 0:
        return
    msg = lib.p2p_ar_strerror(code).decode()
    if code in _VALUE_ERRORS:
        raise ValueError(f"{what}: {msg}")
    if code == -6:
        raise RuntimeError(f"{what}: {msg} (hipError {lib.p2p_ar_last_hip_error()})")
    raise RuntimeError(f"{what}: {msg}")
