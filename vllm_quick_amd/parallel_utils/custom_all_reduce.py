"""One-kernel all-reduce between the GPUs of a node, for the row-parallel linear's partial sums.

Host-side mirror of the reference's vllm/model_executor/parallel_utils/custom_all_reduce.py
(module functions :26-102, class CustomAllreduce :141-227): same names, same decision logic.  The
reference leaves `custom_ar` unbuilt on ROCm (csrc/pybind.cpp:99-113; "For AMD GPUs: custom_ar =
None", custom_all_reduce.py:11-17); here it is backed by libp2p_ar.so.

What differs underneath: the reference's kernels READ the peers' input buffers, so every input has
to be IPC-registered (eager inputs are first copied into a registered buffer, graph inputs are
registered after capture, custom_all_reduce.py:192-200).  The kernels here WRITE into mailboxes the
library owns, so any device tensor can be an input: `register_buffer` and `register_graph_buffers`
are kept for call-site compatibility and do nothing, and `all_reduce_reg` / `all_reduce_unreg` are
the same launch.

A late peer: the reference's kernels spin until every peer has arrived (custom_all_reduce.cuh:144-178).
The kernels here stop waiting after the handle's timeout (5 minutes unless `VLLM_QUICK_P2P_AR_TIMEOUT_MS`
or `set_timeout_ms` says otherwise) so that a GPU is never left with a kernel that cannot end; the result
of such a call is invalid, and the kernel says so in a host-visible flag.  That flag turns into a
RuntimeError at the next all-reduce of the handle (eager or while capturing), at the exit of `capture()`,
in `destroy_custom_ar()` and wherever the caller asks with `get_handle().check_health()` - the call for a
loop that replays captured graphs, after it has synchronised a step.
"""
from __future__ import annotations

import ctypes
import os
from contextlib import contextmanager
from typing import List, Optional

import torch
import torch.distributed as dist

from .. import _p2p_lib
from .parallel_state import (get_tensor_model_parallel_group, get_tensor_model_parallel_rank,
                             get_tensor_model_parallel_world_size)

_CA_HANDLE: Optional["CustomAllreduce"] = None
_IS_CAPTURING = False
_SUPPORTED_WORLD_SIZES = [2, 4, 6, 8]
_TIMEOUT_ENV = "VLLM_QUICK_P2P_AR_TIMEOUT_MS"
_DEFAULT_TIMEOUT_MS = 300000


def init_custom_ar() -> None:
    """Create the group's handle once; leaves it unset (RCCL stays the path) for one GPU, an
    unsupported world size or GPUs without peer access (custom_all_reduce.py:26-49)."""
    global _CA_HANDLE
    if _CA_HANDLE is not None:
        return
    rank = get_tensor_model_parallel_rank()
    world_size = get_tensor_model_parallel_world_size()
    if world_size == 1:
        return
    if world_size not in _SUPPORTED_WORLD_SIZES:
        return
    if not _can_p2p(rank, world_size):
        return
    handle = CustomAllreduce(rank, world_size)
    if _self_check(handle):
        _CA_HANDLE = handle
    else:
        # every rank took the same decision (the verdicts were gathered): the collective library stays the path
        torch.cuda.synchronize()
        dist.barrier(group=get_tensor_model_parallel_group())
        handle.close()


def _self_check(handle: "CustomAllreduce") -> bool:
    """One all-reduce of a known pattern through each of the two algorithms, compared bit for bit with the
    sum every rank can work out by itself; all ranks learn whether ALL of them saw the right answer.
    (No counterpart in the reference; it guards the first use of a path whose memory model - peers
    writing into uncached IPC-mapped memory - depends on the node.)"""
    group = get_tensor_model_parallel_group()
    world, rank = handle.world_size, handle.rank
    ok = True
    try:
        handle.set_timeout_ms(30000)
        dev = torch.device("cuda", torch.cuda.current_device())
        # 8 KiB: one kernel stage for every world size; 2 MiB: two stages from 4 ranks on
        for n in (4096, min(1 << 20, handle.max_size // 2)):
            base = (torch.arange(n, device=dev, dtype=torch.int32) % 7).to(torch.float16)
            got = handle.all_reduce_unreg(base + float(rank + 1))
            want = base * world + float(world * (world + 1) // 2)  # small integers: exact in fp16
            torch.cuda.synchronize()
            ok = ok and bool(torch.equal(got, want))
        ok = ok and not handle.timed_out(clear=True)
        handle.set_timeout_ms(int(os.environ.get(_TIMEOUT_ENV) or _DEFAULT_TIMEOUT_MS))
    except Exception:  # noqa: BLE001 - whatever went wrong, the collective library is the answer
        ok = False
    verdicts: List[Optional[bool]] = [None] * world
    dist.all_gather_object(verdicts, ok, group=group)
    return all(verdicts)


def begin_capture() -> None:
    global _IS_CAPTURING
    _IS_CAPTURING = True


def end_capture() -> None:
    global _IS_CAPTURING
    _IS_CAPTURING = False


def is_capturing() -> bool:
    return _IS_CAPTURING and _CA_HANDLE is not None


def get_handle() -> Optional["CustomAllreduce"]:
    return _CA_HANDLE


def destroy_custom_ar() -> None:
    """Drop the group's handle (no counterpart in the reference, which keeps it for the process's life)."""
    global _CA_HANDLE
    if _CA_HANDLE is not None:
        # a mailbox may go only when no peer can still be writing into it
        torch.cuda.synchronize()
        dist.barrier(group=get_tensor_model_parallel_group())
        handle, _CA_HANDLE = _CA_HANDLE, None
        gave_up = handle.timed_out()
        handle.close()
        if gave_up:
            raise RuntimeError(_GAVE_UP)


@contextmanager
def capture():
    try:
        begin_capture()
        yield
    finally:
        end_capture()
        handle = get_handle()
        if handle is not None:
            handle.register_graph_buffers()
            handle.check_health()


def custom_all_reduce(input: torch.Tensor) -> Optional[torch.Tensor]:
    """The reduced tensor (out of place), or None when this path does not take the input and the
    caller should use the collective library (custom_all_reduce.py:82-102)."""
    ca_handle = get_handle()
    if ca_handle is None:
        return None
    if is_capturing():
        if torch.cuda.is_current_stream_capturing():
            if ca_handle.should_custom_ar(input):
                return ca_handle.all_reduce_reg(input)
        else:
            if ca_handle.should_custom_ar(input):
                # warm-up run before a capture: same allocation pattern, no communication
                return torch.empty_like(input)
    else:
        if ca_handle.should_custom_ar(input):
            return ca_handle.all_reduce_unreg(input)
    return None


def _can_p2p(rank: int, world_size: int) -> bool:
    """Every peer's GPU is this process's own or reachable by peer access (custom_all_reduce.py:132-138;
    the reference takes device index = rank, here the ranks tell each other their devices)."""
    mine = torch.cuda.current_device()
    devices: List[Optional[int]] = [None] * world_size
    dist.all_gather_object(devices, mine, group=get_tensor_model_parallel_group())
    ok = all(d == mine or torch.cuda.can_device_access_peer(mine, d) for i, d in enumerate(devices) if i != rank)
    verdicts: List[Optional[bool]] = [None] * world_size
    dist.all_gather_object(verdicts, ok, group=get_tensor_model_parallel_group())
    return all(verdicts)


def _is_weak_contiguous(t: torch.Tensor) -> bool:
    """The tensor's bytes are one dense run (csrc/custom_all_reduce.cu:52-56)."""
    return t.is_contiguous() or (t.untyped_storage().nbytes() - t.storage_offset() * t.element_size()
                                 == t.numel() * t.element_size())


_GAVE_UP = ("a peer-to-peer all-reduce gave up waiting for a peer (a rank arrived later than the timeout, "
            "or not at all): the results since then are invalid")


class CustomAllreduce:

    # max_size: max supported allreduce size
    def __init__(self, rank: int, world_size: int, max_size: int = 8192 * 1024) -> None:
        self.max_size = max_size
        self.world_size = world_size
        self.rank = rank
        self._ptr = ctypes.c_void_p()
        _p2p_lib.check(_p2p_lib.lib.p2p_ar_create(rank, world_size, max_size, ctypes.byref(self._ptr)),
                       "init_custom_ar")
        # MI355X nodes wire every GPU to every other one, and peer access was checked by the caller
        self.full_nvlink = True
        self.fast_cond = self.full_nvlink or world_size <= 2
        mine = ctypes.create_string_buffer(_p2p_lib.IPC_HANDLE_BYTES)
        _p2p_lib.check(_p2p_lib.lib.p2p_ar_ipc_handle(self._ptr, mine), "ipc handle")
        handles = self._gather_ipc_meta(mine.raw)
        _p2p_lib.check(_p2p_lib.lib.p2p_ar_connect(self._ptr, b"".join(handles)), "open peer mailboxes")
        if os.environ.get(_TIMEOUT_ENV):
            self.set_timeout_ms(int(os.environ[_TIMEOUT_ENV]))

    def _gather_ipc_meta(self, shard_data):
        all_data = [None] * self.world_size
        dist.all_gather_object(all_data, shard_data, group=get_tensor_model_parallel_group())
        return all_data

    def register_buffer(self, inp: torch.Tensor) -> None:
        """Nothing to register: inputs are read by the owning GPU only."""

    def register_graph_buffers(self) -> None:
        """Nothing to register after a capture, for the same reason."""

    def should_custom_ar(self, inp: torch.Tensor) -> bool:
        if inp.dtype not in _p2p_lib.DTYPE_CODES or not inp.is_cuda:
            return False
        if not _is_weak_contiguous(inp) or inp.data_ptr() % 16 != 0:
            return False
        return bool(
            _p2p_lib.lib.p2p_ar_should(inp.numel() * inp.element_size(), self.max_size, self.world_size,
                                       int(self.full_nvlink)))

    def _all_reduce(self, inp: torch.Tensor, out: Optional[torch.Tensor]) -> torch.Tensor:
        if out is None:
            out = torch.empty_like(inp)
        if inp.dtype != out.dtype or inp.numel() != out.numel():
            raise ValueError("input and output must have the same dtype and number of elements")
        if inp.dtype not in _p2p_lib.DTYPE_CODES:
            raise RuntimeError("custom allreduce only supports float32, float16 and bfloat16")
        if not _is_weak_contiguous(out) or not _is_weak_contiguous(inp):
            raise ValueError("custom allreduce needs dense input and output tensors")
        stream = torch.cuda.current_stream(inp.device).cuda_stream
        rc = _p2p_lib.lib.p2p_ar_all_reduce(self._ptr, inp.data_ptr(), out.data_ptr(), inp.numel(),
                                            _p2p_lib.DTYPE_CODES[inp.dtype], stream)
        if rc == _p2p_lib.E_TIMEOUT:  # the host-visible flag of an earlier call: nothing was launched
            raise RuntimeError(_GAVE_UP)
        _p2p_lib.check(rc, "custom_all_reduce")
        return out

    # the two entry points of the reference (registered / unregistered input) are one launch here
    def all_reduce_reg(self, inp: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        return self._all_reduce(inp, out)

    def all_reduce_unreg(self, inp: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        return self._all_reduce(inp, out)

    def set_timeout_ms(self, ms: int) -> None:
        _p2p_lib.check(_p2p_lib.lib.p2p_ar_set_timeout_ms(self._ptr, ms), "set timeout")

    def check_health(self) -> None:
        """Raises RuntimeError if a kernel of this handle has given up waiting for a peer.  Reads host
        memory only (no device call): cheap enough for every step, and legal while a stream is captured.
        A kernel that is still running has not reported yet; synchronise first to be sure of a step."""
        flag = ctypes.c_int(0)
        _p2p_lib.check(_p2p_lib.lib.p2p_ar_poll(self._ptr, ctypes.byref(flag)), "poll")
        if flag.value:
            raise RuntimeError(_GAVE_UP)

    def timed_out(self, clear: bool = False) -> bool:
        """True if a kernel of this handle gave up waiting for a peer (synchronises the device)."""
        torch.cuda.synchronize()
        flag = ctypes.c_int(0)
        _p2p_lib.check(_p2p_lib.lib.p2p_ar_status(self._ptr, ctypes.byref(flag), int(clear)), "status")
        return bool(flag.value)

    def close(self) -> None:
        if self._ptr:
            _p2p_lib.lib.p2p_ar_destroy(self._ptr)
            self._ptr = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass
