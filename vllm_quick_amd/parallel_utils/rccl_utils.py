"""RCCL all-reduce called directly, for use inside HIP graph capture.

Counterpart of the reference's vllm/model_executor/parallel_utils/cupy_utils.py (same module
functions): the reference swaps torch.distributed.all_reduce for CuPy's NCCL binding while it
captures CUDA graphs, "because torch.distributed.all_reduce causes errors when capturing"
(cupy_utils.py:1-9).  The same holds on ROCm: ProcessGroupNCCL's watchdog thread queries the
events of a collective recorded in a capturing stream and the process aborts.  Here the
collective library is called through its C API (`ncclAllReduce` on the caller's stream), with no
helper thread, so the call can be captured and replayed.  No CuPy is involved; the library is the
`librccl.so` PyTorch itself is linked against.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
from typing import Optional

import torch
from torch.distributed import ReduceOp, TCPStore

_OP_MAPPING = {ReduceOp.SUM: 0, ReduceOp.PRODUCT: 1, ReduceOp.MAX: 2, ReduceOp.MIN: 3}  # ncclRedOp_t
_DTYPE_MAPPING = {  # ncclDataType_t
    torch.int8: 0, torch.uint8: 1, torch.int32: 2, torch.int64: 4, torch.float16: 6, torch.float32: 7,
    torch.float64: 8, torch.bfloat16: 9,
}
UNIQUE_ID_BYTES = 128


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_byte * UNIQUE_ID_BYTES)]


_LIB: Optional[ctypes.CDLL] = None
_COMM: Optional[ctypes.c_void_p] = None
_WORLD_SIZE = 0
_RANK = 0
_STREAM: Optional[torch.cuda.Stream] = None


def _library() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        candidates = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so",
                      "librccl.so"]
        err = None
        for path in candidates:
            try:
                lib = ctypes.CDLL(path)
                break
            except OSError as e:
                err = e
        else:
            raise ImportError(f"librccl.so not found (tried {candidates})") from err
        lib.ncclGetErrorString.restype = ctypes.c_char_p
        lib.ncclGetErrorString.argtypes = [ctypes.c_int]
        lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
        lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_void_p]
        lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        _LIB = lib
    return _LIB


def _check(code: int, what: str) -> None:
    if code != 0:
        raise RuntimeError(f"{what}: {_library().ncclGetErrorString(code).decode()} (ncclResult {code})")


def is_initialized() -> bool:
    """Returns whether the communicator exists."""
    return _COMM is not None


@contextlib.contextmanager
def set_cupy_stream(stream: torch.cuda.Stream):
    """Set the stream the all-reduces are enqueued on (name kept from cupy_utils.py:61-67)."""
    global _STREAM
    old = _STREAM
    _STREAM = stream
    try:
        yield
    finally:
        _STREAM = old


set_stream = set_cupy_stream


def init_process_group(world_size: int, rank: int, host: str, port: int) -> None:
    """Creates the communicator of `world_size` ranks on the current device; rank 0's unique id
    travels through a TCP store at host:port (cupy_utils.py:70-96 does the same through CuPy)."""
    global _COMM, _WORLD_SIZE, _RANK
    assert not is_initialized()
    assert world_size > 0, f"{world_size=} should be a positive integer"
    assert 0 <= rank < world_size, f"{rank=} should be a integer between [0, {world_size})"
    lib = _library()
    uid = _UniqueId()
    store = TCPStore(host, port, world_size, is_master=(rank == 0))
    if rank == 0:
        _check(lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        store.set("rccl_unique_id", bytes(bytearray(uid.internal)))
    else:
        raw = store.get("rccl_unique_id")
        ctypes.memmove(ctypes.byref(uid), raw, UNIQUE_ID_BYTES)
    comm = ctypes.c_void_p()
    _check(lib.ncclCommInitRank(ctypes.byref(comm), world_size, uid, rank), "ncclCommInitRank")
    # ncclCommInitRank returns once every rank has joined, i.e. has read the id: the store can go
    del store
    _COMM, _WORLD_SIZE, _RANK = comm, world_size, rank


def all_reduce(input_: torch.Tensor, op=ReduceOp.SUM) -> None:
    """All-reduces the input tensor across the communicator, in place, on the set (or current) stream."""
    assert input_.is_cuda, f"{input_} should be a cuda tensor"
    assert is_initialized(), "rccl_utils.init_process_group() has not run"
    if not input_.is_contiguous():
        raise ValueError("all_reduce needs a contiguous tensor")
    stream = _STREAM if _STREAM is not None else torch.cuda.current_stream(input_.device)
    _check(
        _library().ncclAllReduce(input_.data_ptr(), input_.data_ptr(), input_.numel(), _DTYPE_MAPPING[input_.dtype],
                                 _OP_MAPPING[op], _COMM, stream.cuda_stream), "ncclAllReduce")


def destroy_process_group() -> None:
    """Destroys the communicator."""
    global _COMM, _WORLD_SIZE, _RANK
    if _COMM is not None:
        torch.cuda.synchronize()
        _library().ncclCommDestroy(_COMM)
    _COMM, _WORLD_SIZE, _RANK = None, 0, 0


def get_world_size() -> int:
    """Returns the world size."""
    return _WORLD_SIZE


def get_rank() -> int:
    return _RANK


def get_nccl_backend():
    return _COMM
