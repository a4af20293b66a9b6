"""Peer-to-peer all-reduce on the GPU, through the C ABI, bit-exact against the oracle.

One MI355X is enough for a functional check: the ranks are separate processes that all use
device 0 and map each other's mailboxes through HIP IPC handles, exactly the code path of one
process per GPU (what a single GPU cannot show is xGMI traffic and cross-GPU cache behaviour).
Shapes follow the reference's test (tests/distributed/test_custom_all_reduce.py:15-18,30-53,64-73):
random sizes up to 2 M elements in the three element types, eager and captured in a graph with
the inputs rewritten between replays.
"""
import ctypes
import os
import random
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

TORCH_DTYPES = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def make_input(kind, dtype, n, seed, rank):
    """Rank `rank`'s vector of case `seed`, reproducible on every rank: numpy array in the oracle's
    representation (bf16 as uint16 bit patterns)."""
    from oracle import allreduce_oracle as aro
    rng = np.random.default_rng([seed, rank])
    v = rng.integers(1, 16, n).astype(np.float32) if kind == "int" else (rng.standard_normal(n) * 4).astype(
        np.float32)
    if dtype == "f32":
        return v
    if dtype == "f16":
        return v.astype(np.float16)
    return aro.f32_to_bf16(v)


def to_device(a, dtype, dev):
    t = torch.from_numpy(a.view(np.int16) if dtype == "bf16" else a).to(dev)
    return t.view(torch.bfloat16) if dtype == "bf16" else t


def bits(t):
    t = t.detach().cpu().contiguous()
    return t.view(torch.int32 if t.element_size() == 4 else torch.int16).numpy()


def want_bits(inputs, dtype):
    from oracle import allreduce_oracle as aro
    r = aro.all_reduce(inputs, dtype)
    return r.view(np.int32 if dtype == "f32" else np.int16)


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        from vllm_quick_amd.parallel_utils import communication_op as comm
        from vllm_quick_amd.parallel_utils import custom_all_reduce as car
        from vllm_quick_amd.parallel_utils import parallel_state as ps

        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        ps.initialize_model_parallel(world)
        car.init_custom_ar()
        fa = car.get_handle()
        assert fa is not None, "custom all-reduce did not initialise"
        fa.set_timeout_ms(20000)

        # the reference's eager test: ones, float32 and bfloat16 (test_custom_all_reduce.py:64-73)
        for dt, n in ((torch.float32, 1024), (torch.bfloat16, 4096)):
            inp = torch.ones(n, dtype=dt, device=dev)
            out = fa.all_reduce_unreg(inp)
            assert torch.equal(out, inp * world)

        # sizes of the reference's graph test (:15-18) plus the corners of the two algorithms
        rnd = random.Random(42)
        sizes = [rnd.randint(1024, 2048 * 1024) for _ in range(4)]
        sizes = [s - s % 8 for s in sizes] + [8, 16, 1024, 128 * 1024 - 8, 128 * 1024, 256 * 1024, 2048 * 1024]
        case = 0
        for dtype in ("f32", "f16", "bf16"):
            for n in sizes:
                if n * torch.empty((), dtype=TORCH_DTYPES[dtype]).element_size() % 16:
                    continue
                for kind in ("int", "gauss"):
                    case += 1
                    ins = [make_input(kind, dtype, n, case, r) for r in range(world)]
                    x = to_device(ins[rank], dtype, dev)
                    y = comm.tensor_model_parallel_all_reduce(x)
                    assert y.data_ptr() != x.data_ptr(), "the peer-to-peer path is out of place"
                    assert np.array_equal(bits(y), want_bits(ins, dtype)), (dtype, n, kind)
                    assert np.array_equal(bits(x), ins[rank].view(bits(x).dtype)), "input must be left alone"

        # inputs this path declines go to the collective library (here: none is set up for GPU tensors,
        # so only the decision is checked)
        assert car.custom_all_reduce(torch.ones(9, dtype=torch.float16, device=dev)) is None  # 18 bytes
        assert car.custom_all_reduce(torch.ones(8192 * 1024 // 4 + 4, device=dev)) is None  # > max_size
        assert car.custom_all_reduce(torch.ones(64, dtype=torch.float64, device=dev)) is None

        # in place, many calls back to back with changing sizes (changing grids and mailbox halves)
        seq = [8, 4096, 300 * 1024, 8, 64, 200 * 1024, 1024 * 1024, 16]
        bufs, wants = [], []
        for i in range(64):
            n = seq[i % len(seq)]
            ins = [make_input("gauss", "f16", n, 1000 + i, r) for r in range(world)]
            x = to_device(ins[rank], "f16", dev)
            fa.all_reduce_reg(x, x)
            bufs.append(x)
            wants.append(want_bits(ins, "f16"))
        for x, w in zip(bufs, wants):
            assert np.array_equal(bits(x), w)

        # captured in a graph, replayed with rewritten inputs (test_custom_all_reduce.py:30-53)
        for dtype, n in (("f16", 4096), ("bf16", 640 * 1024), ("f32", 100 * 1024)):
            inp1 = torch.zeros(n, dtype=TORCH_DTYPES[dtype], device=dev)
            inp2 = torch.zeros(n, dtype=TORCH_DTYPES[dtype], device=dev)
            with car.capture():
                # warm-up outside a capture: same allocations, no communication (custom_all_reduce.py:91-95)
                w = comm.tensor_model_parallel_all_reduce(inp1)
                assert w.shape == inp1.shape and w.data_ptr() != inp1.data_ptr()
                torch.cuda.synchronize()
                stream = torch.cuda.Stream(device=dev)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.stream(stream), torch.cuda.graph(graph, stream=stream):
                    out1 = comm.tensor_model_parallel_all_reduce(inp1)
                    inp1.add_(1)  # the input is modified right after: the result must not see it
                    out2 = comm.tensor_model_parallel_all_reduce(inp2)
            for rep in range(3):
                a = [make_input("gauss", dtype, n, 5000 + rep, r) for r in range(world)]
                b = [make_input("int", dtype, n, 6000 + rep, r) for r in range(world)]
                inp1.copy_(to_device(a[rank], dtype, dev))
                inp2.copy_(to_device(b[rank], dtype, dev))
                torch.cuda.synchronize()
                graph.replay()
                torch.cuda.synchronize()
                assert np.array_equal(bits(out1), want_bits(a, dtype)), (dtype, n, rep)
                assert np.array_equal(bits(out2), want_bits(b, dtype)), (dtype, n, rep)
            # eager calls between replays share the call counter with the graph
            ins = [make_input("int", dtype, n, 7000, r) for r in range(world)]
            y = comm.tensor_model_parallel_all_reduce(to_device(ins[rank], dtype, dev))
            assert np.array_equal(bits(y), want_bits(ins, dtype))
            graph.replay()
            torch.cuda.synchronize()
            del graph

        # the caller this exists for: a row-parallel W4A16 linear, partial products summed over the ranks
        # (vllm/model_executor/layers/linear.py:541-542)
        if world in (2, 4):
            from oracle import awq_oracle as orc
            from vllm_quick_amd.layers import linear as L
            from vllm_quick_amd.layers.quantization import AWQConfig
            method = AWQConfig(4, 32, True).get_linear_method()
            M, K, N = 5, 256, 192
            xh, qw, sc, qz = orc.make_inputs(M, K, N, 32, seed=41)
            want = orc.gemm_faithful(xh, qw, sc, qz).astype(np.float32)
            row = L.RowParallelLinear(K, N, bias=False, input_is_parallel=False, params_dtype=torch.float16,
                                      linear_method=method).to(dev)
            for name, arr in (("qweight", qw), ("scales", sc), ("qzeros", qz)):
                prm = getattr(row, name)
                prm.weight_loader(prm, torch.from_numpy(arr))
            calls = []
            inner = fa.all_reduce_unreg
            fa.all_reduce_unreg = lambda t, out=None: (calls.append(t.shape), inner(t, out))[1]
            y, _ = row(torch.from_numpy(xh).to(dev))
            fa.all_reduce_unreg = inner
            assert calls == [(M, N)], "the layer's all-reduce did not take the peer-to-peer path"
            # tolerance of the W4A16 GEMM parity tests: rtol 1e-2, atol 1e-3 (+1e-3 for the extra fp16 rounding
            # of each rank's partial product)
            assert np.allclose(y.float().cpu().numpy(), want, rtol=1e-2, atol=2e-3)

        assert not fa.timed_out(), "a kernel gave up waiting for a peer"
        car.destroy_custom_ar()
        ps.destroy_model_parallel()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
        raise


@pytest.mark.parametrize("world", [2, 4, 6, 8])
def test_multi_process_all_reduce_on_one_gpu(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        results = [q.get(timeout=280) for _ in procs]
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    for rank, status in results:
        assert status == "ok", f"rank {rank}:\n{status}"


def _late_worker(rank, world, port, q):
    """Rank 1 arrives 1.5 s late at the first all-reduce; rank 0 waits 0.3 s at most."""
    try:
        import time
        sys.path.insert(0, ROOT)
        from vllm_quick_amd.parallel_utils import communication_op as comm
        from vllm_quick_amd.parallel_utils import custom_all_reduce as car
        from vllm_quick_amd.parallel_utils import parallel_state as ps

        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        ps.initialize_model_parallel(world)
        car.init_custom_ar()
        fa = car.get_handle()
        assert fa is not None
        fa.set_timeout_ms(300 if rank == 0 else 20000)
        x = torch.full((4096, ), float(rank + 1), dtype=torch.float16, device=dev)
        dist.barrier()
        if rank == 1:
            time.sleep(1.5)
        y = comm.tensor_model_parallel_all_reduce(x)  # returns at once: the kernel is what waits
        torch.cuda.synchronize()
        if rank == 0:
            # the result is invalid, and every way of asking says so
            with pytest.raises(RuntimeError, match="gave up waiting"):
                fa.check_health()
            with pytest.raises(RuntimeError, match="gave up waiting"):
                comm.tensor_model_parallel_all_reduce(x)  # nothing launched: the call sequence stays aligned
            with pytest.raises(RuntimeError, match="gave up waiting"):
                with car.capture():
                    pass
            assert fa.timed_out(clear=True)
            fa.check_health()  # cleared
        else:
            # the late rank found rank 0's contribution waiting in its mailbox
            assert torch.equal(y, torch.full_like(y, 3.0))
            fa.check_health()
        dist.barrier()
        # both ranks made call 1, so call 2 pairs up again
        fa.set_timeout_ms(20000)
        y = comm.tensor_model_parallel_all_reduce(x)
        torch.cuda.synchronize()
        assert torch.equal(y, torch.full_like(y, 3.0))
        # destroy_custom_ar() reports a flag nobody has cleared
        dist.barrier()
        if rank == 1:
            time.sleep(1.0)
        fa.set_timeout_ms(300 if rank == 0 else 20000)
        comm.tensor_model_parallel_all_reduce(x)
        torch.cuda.synchronize()
        if rank == 0:
            with pytest.raises(RuntimeError, match="gave up waiting"):
                car.destroy_custom_ar()
            assert car.get_handle() is None
        else:
            car.destroy_custom_ar()
        ps.destroy_model_parallel()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except BaseException:  # noqa: BLE001 - pytest.raises failures are not Exceptions
        import traceback
        q.put((rank, traceback.format_exc()))
        raise


def test_a_late_peer_raises_on_the_waiting_rank():
    """A rank delayed past the timeout: the rank that gave up gets a RuntimeError at its next all-reduce,
    from check_health(), at the exit of capture() and from destroy_custom_ar(); after the flag is cleared
    the two ranks pair up again."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_late_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        results = [q.get(timeout=120) for _ in procs]
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    for rank, status in results:
        assert status == "ok", f"rank {rank}:\n{status}"


def test_two_contexts_in_one_process():
    """Both ranks in this process, connected by plain device pointers, one stream each."""
    from vllm_quick_amd import _p2p_lib as P
    L = P.lib
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctxs = [ctypes.c_void_p() for _ in range(2)]
    for r in range(2):
        P.check(L.p2p_ar_create(r, 2, 1 << 20, ctypes.byref(ctxs[r])), "create")
    try:
        boxes = (ctypes.c_void_p * 2)(*[L.p2p_ar_mailbox_ptr(c) for c in ctxs])
        for c in ctxs:
            P.check(L.p2p_ar_connect_ptrs(c, boxes), "connect")
            P.check(L.p2p_ar_set_timeout_ms(c, 5000), "timeout")
        assert L.p2p_ar_connect_ptrs(ctxs[0], boxes) == -7  # already connected
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        for dtype, n in (("f16", 8), ("f16", 40 * 1024), ("bf16", 512 * 1024), ("f32", 256 * 1024)):
            ins = [make_input("gauss", dtype, n, 99, r) for r in range(2)]
            xs = [to_device(a, dtype, dev) for a in ins]
            outs = [torch.empty_like(x) for x in xs]
            torch.cuda.synchronize()
            for r in range(2):
                P.check(
                    L.p2p_ar_all_reduce(ctxs[r], xs[r].data_ptr(), outs[r].data_ptr(), n,
                                        P.DTYPE_CODES[TORCH_DTYPES[dtype]], streams[r].cuda_stream), "all_reduce")
            torch.cuda.synchronize()
            flag = ctypes.c_int(0)
            for r in range(2):
                P.check(L.p2p_ar_status(ctxs[r], ctypes.byref(flag), 0), "status")
                assert flag.value == 0, "the two kernels did not run side by side"
                assert np.array_equal(bits(outs[r]), want_bits(ins, dtype)), (dtype, n, r)
        # argument checks that need a live context
        x = torch.zeros(64, dtype=torch.float16, device=dev)
        assert L.p2p_ar_all_reduce(ctxs[0], x.data_ptr(), x.data_ptr(), 12, P.F16, None) == -3  # 24 bytes
        assert L.p2p_ar_all_reduce(ctxs[0], x.data_ptr() + 2, x.data_ptr(), 8, P.F16, None) == -4
        assert L.p2p_ar_all_reduce(ctxs[0], x.data_ptr(), x.data_ptr(), 8, 7, None) == -5
        assert L.p2p_ar_all_reduce(ctxs[0], x.data_ptr(), x.data_ptr(), (1 << 20) // 2 + 8, P.F16, None) == -3
    finally:
        torch.cuda.synchronize()
        for c in ctxs:
            L.p2p_ar_destroy(c)


def test_a_missing_peer_ends_in_a_status_flag_not_a_hang():
    """Rank 0 calls, rank 1 never does: the kernel must give up after the context's timeout."""
    import time

    from vllm_quick_amd import _p2p_lib as P
    L = P.lib
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctxs = [ctypes.c_void_p() for _ in range(2)]
    for r in range(2):
        P.check(L.p2p_ar_create(r, 2, 1 << 20, ctypes.byref(ctxs[r])), "create")
    try:
        boxes = (ctypes.c_void_p * 2)(*[L.p2p_ar_mailbox_ptr(c) for c in ctxs])
        for c in ctxs:
            P.check(L.p2p_ar_connect_ptrs(c, boxes), "connect")
        P.check(L.p2p_ar_set_timeout_ms(ctxs[0], 300), "timeout")
        assert L.p2p_ar_set_timeout_ms(ctxs[0], 0) == -3
        x = torch.ones(4096, dtype=torch.float16, device=dev)
        y = torch.empty_like(x)
        flag = ctypes.c_int(0)
        P.check(L.p2p_ar_status(ctxs[0], ctypes.byref(flag), 0), "status")
        assert flag.value == 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        P.check(L.p2p_ar_all_reduce(ctxs[0], x.data_ptr(), y.data_ptr(), x.numel(), P.F16, None), "all_reduce")
        torch.cuda.synchronize()
        waited = time.perf_counter() - t0
        assert 0.25 < waited < 5.0, f"gave up after {waited:.3f} s, timeout was 0.3 s"
        P.check(L.p2p_ar_poll(ctxs[0], ctypes.byref(flag)), "poll")
        assert flag.value == 1, "the host copy of the flag must be set once the kernel has ended"
        assert L.p2p_ar_all_reduce(ctxs[0], x.data_ptr(), y.data_ptr(), x.numel(), P.F16, None) == P.E_TIMEOUT
        P.check(L.p2p_ar_status(ctxs[0], ctypes.byref(flag), 1), "status")
        assert flag.value == 1
        P.check(L.p2p_ar_status(ctxs[0], ctypes.byref(flag), 0), "status")
        assert flag.value == 0, "clear must reset the sticky flag"
        P.check(L.p2p_ar_poll(ctxs[0], ctypes.byref(flag)), "poll")
        assert flag.value == 0, "and its host copy"
    finally:
        torch.cuda.synchronize()
        for c in ctxs:
            L.p2p_ar_destroy(c)
