"""Latency of the peer-to-peer all-reduce against the collective library, one process per GPU.

    torchrun --nproc-per-node 8 benchmarks/allreduce_bench.py             # RCCL as the comparison
    torchrun --nproc-per-node 4 benchmarks/allreduce_bench.py --same-device --backend gloo
                                                                            # rehearsal: all ranks on GPU 0

For each message size (the row-parallel outputs of the W4A16 layers: [M, 8192] fp16 for M = 1..64, the
[128, 11008] output of the benchmark layer, and the 8 MiB maximum) it
  * checks the peer-to-peer result on small-integer inputs, whose sum is exact in any order, against
    world * (world + 1) / 2 patterns and, with RCCL, against RCCL's result on the same input
    (the reference's test does the same, tests/distributed/test_custom_all_reduce.py:30-53);
  * times both, launched one by one and replayed from a HIP graph (16 all-reduces per replay),
    HIP events on the stream.
Rank 0 prints one JSON object.  Every wait inside the kernels is bounded, so a rank whose peer is
missing reports `timed_out` instead of hanging.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from vllm_quick_amd.parallel_utils import custom_all_reduce as car  # noqa: E402
from vllm_quick_amd.parallel_utils import parallel_state as ps  # noqa: E402

PER_GRAPH = 16


def time_graph(fn, stream, reps=10):
    with torch.cuda.stream(stream):
        fn()
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
            for _ in range(PER_GRAPH):
                fn()
        g.replay()
        stream.synchronize()
        ts = []
        for _ in range(reps):
            dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            g.replay()
            e1.record(stream)
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / PER_GRAPH)
    ts.sort()
    return ts[len(ts) // 2]


def time_eager(fn, stream, n=64):
    with torch.cuda.stream(stream):
        for _ in range(8):
            fn()
        stream.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n):
            fn()
        e1.record(stream)
        e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def time_steps(fa, rank, world, dev, stream, direct, nsets=8):
    """`direct`: the initialised rccl_utils module, or None (no RCCL comparison)."""
    from bench import GROUP, K_IN, M_HEADLINE, N_OUT, make_weights, make_x
    from vllm_quick_amd._C import ops
    K, N, G, M, p = K_IN, N_OUT, GROUP, M_HEADLINE, world
    if N % (8 * p) or (K // p) % G:
        raise RuntimeError(f"{K}x{N} g{G} cannot be sharded {p} ways")
    col, row = [], []
    for i in range(nsets):
        qw, s, qz = make_weights(K, N, G, seed=i, device=dev)
        nc, n, kk, gg = N // 8 // p, N // p, K // p, K // G // p
        col.append(ops.awq_repack(qw[:, rank * nc:(rank + 1) * nc].contiguous(), s[:, rank * n:(rank + 1) * n].contiguous(),
                                  qz[:, rank * nc:(rank + 1) * nc].contiguous()))
        row.append(ops.awq_repack(qw[rank * kk:(rank + 1) * kk].contiguous(), s[rank * gg:(rank + 1) * gg].contiguous(),
                                  qz[rank * gg:(rank + 1) * gg].contiguous()))
    x_full = make_x(M, K, seed=1234, device=dev)
    x_row = x_full[:, rank * (K // p):(rank + 1) * (K // p)].contiguous()
    res = {"weight_sets": nsets, "M": M}

    def run(reduce_fn):
        def steps():
            for i in range(nsets):
                ops.awq_gemm_packed(x_full, col[i])
                reduce_fn(ops.awq_gemm_packed(x_row, row[i]))
        with torch.cuda.stream(stream):
            steps()
            stream.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                steps()
            g.replay()
            stream.synchronize()
            ts = []
            for _ in range(10):
                dist.barrier()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                g.replay()
                e1.record(stream)
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / nsets)
        ts.sort()
        return ts[len(ts) // 2]

    res["gemms_only_us"] = run(lambda y: y)
    res["with_p2p_us"] = run(lambda y: fa.all_reduce_reg(y, y))
    if direct is not None:
        res["with_rccl_us"] = run(lambda y: direct.all_reduce(y))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, also the comparison) or gloo (control plane only)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--timeout-ms", type=int, default=5000)
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if a.same_device else int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    rccl = a.backend == "nccl"
    if rccl:
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(a.backend)
    ps.initialize_model_parallel(world)
    out = {"world": world, "comparison": "rccl" if rccl else None, "same_device": a.same_device, "sizes": {}}
    try:
        car.init_custom_ar()
        fa = car.get_handle()
        if fa is None:
            raise RuntimeError("custom all-reduce not available for this group (world size or peer access)")
        fa.set_timeout_ms(a.timeout_ms)
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)  # inputs are made on the stream that reduces them
        sizes = [("M1x8192", 8192), ("M8x8192", 8 * 8192), ("M16x8192", 16 * 8192), ("M64x8192", 64 * 8192),
                 ("M128x11008", 128 * 11008), ("8MiB", 4 * 1024 * 1024)]
        ok_all = True
        for name, n in sizes:
            x = ((torch.arange(n, device=dev) % 7) + rank + 1).to(torch.float16)  # small integers: exact sums
            want = (torch.arange(n, device=dev) % 7).to(torch.float32) * world + world * (world + 1) / 2
            with torch.cuda.stream(stream):
                y = fa.all_reduce_unreg(x)
                stream.synchronize()
            ok = bool(torch.equal(y.float(), want))
            entry = {"bytes": 2 * n, "exact": ok}
            if rccl:
                z = x.clone()
                dist.all_reduce(z)
                torch.cuda.synchronize()
                entry["equals_rccl"] = bool(torch.equal(z, y))
                ok = ok and entry["equals_rccl"]
            # non-integer data: every rank must hold the same bits
            g = torch.Generator(device=dev)
            g.manual_seed(1234 + rank)
            xr = (torch.randn(n, device=dev, generator=g) * 3).to(torch.float16)
            with torch.cuda.stream(stream):
                yr = fa.all_reduce_unreg(xr)
                stream.synchronize()
            ref = yr.clone()
            if rccl:
                dist.broadcast(ref, src=0)
            else:
                h = ref.cpu()
                dist.broadcast(h, src=0)
                ref = h.to(dev)
            entry["same_bits_on_all_ranks"] = bool(torch.equal(ref, yr))
            ok = ok and entry["same_bits_on_all_ranks"]
            buf = torch.empty_like(x)
            entry["p2p_graph_us"] = time_graph(lambda: fa.all_reduce_reg(x, buf), stream)
            entry["p2p_eager_us"] = time_eager(lambda: fa.all_reduce_reg(x, buf), stream)
            if rccl:
                z = x.clone()
                entry["rccl_eager_us"] = time_eager(lambda: dist.all_reduce(z), stream)
            ok_all = ok_all and ok
            out["sizes"][name] = entry
        out["timed_out"] = fa.timed_out()
        out["ok"] = ok_all and not out["timed_out"]
        oks = [None] * world
        dist.all_gather_object(oks, out["ok"])
        out["ok_all_ranks"] = all(oks)
        if rank == 0:
            print(json.dumps(out), flush=True)  # what is known so far, should anything below end the process
        direct = None
        if rccl:
            # the collective library replayed from a graph too: called directly (torch.distributed's RCCL path
            # aborts the process when captured)
            try:
                import socket

                from vllm_quick_amd.parallel_utils import rccl_utils
                port = [0]
                if rank == 0:
                    with socket.socket() as sock:
                        sock.bind(("127.0.0.1", 0))
                        port[0] = sock.getsockname()[1]
                dist.broadcast_object_list(port, src=0)
                rccl_utils.init_process_group(world, rank, "127.0.0.1", port[0])
                direct = rccl_utils
                for name, n in sizes:
                    z = torch.ones(n, dtype=torch.float16, device=dev)
                    out["sizes"][name]["rccl_graph_us"] = time_graph(lambda: direct.all_reduce(z), stream)
            except Exception as e:  # noqa: BLE001
                out["rccl_graph_error"] = f"{type(e).__name__}: {e}"
            if rank == 0:
                print(json.dumps(out), flush=True)
        # the benchmark layer's tensor-parallel step (column-parallel GEMM, row-parallel GEMM, all-reduce of
        # its [128, 11008] output), replayed from a HIP graph, with either collective inside
        try:
            out["step"] = time_steps(fa, rank, world, dev, stream, direct)
        except Exception as e:  # noqa: BLE001
            out["step_error"] = f"{type(e).__name__}: {e}"
        if direct is not None:
            direct.destroy_process_group()
        car.destroy_custom_ar()
    except Exception as e:  # noqa: BLE001 - reported, not raised: this is a measurement aid
        out["ok"] = False
        out["error"] = f"{type(e).__name__}: {e}"
    if rank == 0:
        print(json.dumps(out), flush=True)
    ps.destroy_model_parallel()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
