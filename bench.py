"""Benchmark of the W4A16 (AWQ) GEMM path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload: the single 4096 x 11008 group-128 W4A16 linear of the kernel-only sweep (the Llama-2-7B
gate/up/down shape), synthetic seeded inputs already resident in HBM.
One STEP = that linear applied twice to a batch of M = 128 tokens, once per tensor-parallel
sharding of the layer over the N ranks:
  * column-parallel: every rank multiplies by its N/p output columns, no collective;
  * row-parallel   : every rank multiplies its K/p input slice, then one RCCL all-reduce
                     of the [M, 11008] partial products over xGMI.
At N = 1 both are the unsharded GEMM.  `value` = 2 * (2 M K N) flops per step / step time,
whole job, i.e. the TFLOP/s of the 4096 x 11008 layer ("strong" scaling: the layer is fixed).
16 distinct weight sets are rotated (375 MB > the 256 MB Infinity Cache) so weights come
from HBM on every launch.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline       dominant kernel of the timed region (M = 128, MFMA-bound),
  roofline_m1    the M = 1 decode launch of the same layer (HBM-bound), N = 1 only,
  sweep          per-M kernel times of the whole M sweep (and of the stock-layout kernel, which serves a call
                 only when no packed copy can be made), N = 1 only,
  sweep_bf16     the same layer with bf16 activations at 1, 16, 128 and 1024 rows (beside the fp16 times), N = 1 only,
  cpu_baseline   torch.matmul(x, dequant(W)) with the oracle's dequantiser on the host cores, N = 1 only,
  decode         Llama-2-7B-AWQ-shaped greedy decode tokens/s (benchmarks/llama_decode.py), N = 1 only,
  experts        Mixtral-8x7B-shaped expert block, batched launches vs per-expert loop, N = 1 only,
  tp             column-/row-parallel halves of the step and the all-reduce timed separately, with the
                 roofline of each half on this rank's shard.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from vllm_quick_amd._C import ops  # noqa: E402  (raises if libw4a16.so is missing)

K_IN, N_OUT, GROUP = 4096, 11008, 128
M_HEADLINE = 128
NBUF = 16
PEAK_HBM_GBPS = 8000.0  # MI355X HBM3E, vendor figure
PEAK_F16_TFLOPS = 2500.0  # MI355X dense fp16 matrix, vendor figure


def algorithmic_bytes(M, K, N, G):
    """qweight + scales + qzeros + x + out, each touched once."""
    return K * N // 2 + (K // G) * N * 2 + (K // G) * (N // 8) * 4 + 2 * M * K + 2 * M * N


def make_weights(K, N, G, seed, device):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    qw = torch.randint(-2**31, 2**31 - 1, (K, N // 8), generator=g, device=device, dtype=torch.int64).to(torch.int32)
    qz = torch.randint(-2**31, 2**31 - 1, (K // G, N // 8), generator=g, device=device,
                       dtype=torch.int64).to(torch.int32)
    s = (torch.rand((K // G, N), generator=g, device=device) * (2e-2 - 1e-3) + 1e-3).half()
    return qw, s, qz


def make_x(M, K, seed, device):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return (torch.randn((M, K), generator=g, device=device) * 0.25).half()


def graph_of(fns, stream):
    """Capture one launch of each callable into a HIP graph on `stream`."""
    with torch.cuda.stream(stream):
        for f in fns:
            f()
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        # thread-local capture mode: a process group's helper threads may call the runtime meanwhile
        mode = "thread_local" if dist.is_initialized() else "global"
        with torch.cuda.graph(g, stream=stream, capture_error_mode=mode):
            for f in fns:
                f()
    return g


def time_launches(fns, stream, launches, warmup):
    """Mean microseconds per launch over `launches` launches (graph replays of len(fns) launches
    each, rotating the weight sets), timed with HIP events recorded on the launch stream."""
    g = graph_of(fns, stream)
    reps = max(1, launches // len(fns))
    with torch.cuda.stream(stream):
        for _ in range(max(1, warmup // len(fns))):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            g.replay()
        e1.record(stream)
        e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(fns))


def cpu_baseline(M, K, N, G):
    """torch.matmul(x, dequant(W)) on the host cores: the oracle's dequantiser (numpy) followed by
    torch's fp32 CPU matmul, rounded to fp16.  Whole 4096 x 11008 layer, repeated for about 12 s of CPU
    work (at least 3 passes), median pass."""
    import numpy as np
    from oracle import awq_oracle as orc
    x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=0)
    t_dq, t_mm = [], []
    begun = time.perf_counter()
    while len(t_dq) < 3 or (time.perf_counter() - begun < 12.0 and len(t_dq) < 200):
        t0 = time.perf_counter()
        w = orc.dequantize(qw, s, qz)
        t1 = time.perf_counter()
        y = torch.matmul(torch.from_numpy(x).float(), torch.from_numpy(w).float()).half()
        t2 = time.perf_counter()
        t_dq.append(t1 - t0)
        t_mm.append(t2 - t1)
    dq, mm = float(np.median(t_dq)), float(np.median(t_mm))
    flops = 2.0 * M * K * N
    return {
        "value": flops / (dq + mm) * 1e-12,
        "unit": "TFLOP/s",
        "cores": torch.get_num_threads(),
        "host_cpus": os.cpu_count(),
        "kind": "port",
        "sample": f"whole {K}x{N} g{G} layer at M={M}, dequant+matmul, median of {len(t_dq)} passes "
                  f"({time.perf_counter() - begun:.1f} s of CPU work)",
        "dequant_s": dq,
        "matmul_s": mm,
        "matmul_only_tflops": flops / mm * 1e-12,
        "checksum": float(y.float().abs().sum()),
    }


def load_traffic():
    """HBM bytes per launch from the committed PMC profile (separate rocprofv3 --pmc passes), if any.
    The file names the kernels it measured; figures of a kernel that is no longer the one launched are
    not reported."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f)
    return {}


def load_mfma():
    """Matrix-core utilisation per kernel from the committed counter profile (its own rocprofv3 --pmc pass:
    SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)), if any; stamped with kernel names like the traffic."""
    path = os.path.join(ROOT, "profiles", "mfma.json")
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f)
    return {}


def mfma_busy_of(mfma, m, kernel):
    """The counter figure measured at `m` rows, if the profile was taken on `kernel`."""
    if mfma.get(f"m{m}_kernel") != kernel:
        return None
    return mfma.get(f"m{m}_mfma_busy")


def traffic_of(traffic, key, kernel):
    """The measured HBM bytes per launch under `key`, if the profile was taken on `kernel`."""
    if traffic.get(f"{key}_kernel") != kernel:
        return None
    return traffic.get(f"{key}_hbm_bytes_per_launch")


def shard_roofline(M, K, n, G, us):
    """MFMA roofline of one launch on a [K, n] shard: algorithmic flops / mean launch duration."""
    from vllm_quick_amd._lib import launch_shape
    fl = 2.0 * M * K * n
    return {"kernel": launch_shape(M, K, n), "bound": "mfma", "achieved": fl / us * 1e-6, "peak": PEAK_F16_TFLOPS,
            "unit": "TFLOP/s", "frac": fl / us * 1e-6 / PEAK_F16_TFLOPS,
            "algorithmic_bytes": algorithmic_bytes(M, K, n, G), "launch_us": us}


def run_children(script, script_args, rank, world, timeout_s=180):
    """Runs `script` once per rank in child processes that form a process group of their own (same
    RANK / LOCAL_RANK / WORLD_SIZE, a fresh rendezvous port), so that nothing they do can disturb the
    figures measured in this process; returns rank 0's JSON line.  A failure is reported in the
    result, not raised."""
    import socket
    import subprocess
    port = [0]
    if rank == 0:
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port[0] = sock.getsockname()[1]
    dist.broadcast_object_list(port, src=0)
    # a rendezvous of the children's own: not the launcher's store
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port[0]))
    cmd = [sys.executable, os.path.join(ROOT, script)] + list(script_args)
    if torch.cuda.device_count() < world:
        cmd.append("--same-device")
    res = {"ok": False}
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if rank == 0:
            res = json.loads(lines[-1]) if lines else {"ok": False, "error": f"exit {r.returncode}: {r.stderr[-400:]}"}
    except Exception as e:  # noqa: BLE001 - a measurement aid must not take the benchmark down
        res = {"ok": False, "error": f"{type(e).__name__}: {e}"}
    dist.barrier()
    return res if rank == 0 else None


def p2p_all_reduce_section(backend, rank, world):
    """The one-kernel peer-to-peer all-reduce against the collective library on this node's GPUs:
    correctness flags and latencies per message size (benchmarks/allreduce_bench.py)."""
    return run_children(os.path.join("benchmarks", "allreduce_bench.py"), ["--backend", backend], rank, world,
                        timeout_s=150)


def decode_tp_section(backend, rank, world, with_p2p):
    """Tensor-parallel half of the headline metric: Llama-2-70B-AWQ-shaped greedy decode over `world`
    ranks (8 of the 80 layers run, extrapolated), with RCCL and, if it just proved exact on these GPUs
    (`with_p2p`), with the peer-to-peer all-reduce.
    (Llama-2-7B's 11008-wide down_proj cannot be row-split 4 or 8 ways at group 128.)"""
    base = ["--model", "70b", "--layers", "8", "--bs", "1,8,64", "--backend", backend]
    res = {}
    if backend == "nccl":
        res["rccl"] = run_children(os.path.join("benchmarks", "llama_decode.py"), base, rank, world, timeout_s=150)
    if with_p2p:
        res["p2p"] = run_children(os.path.join("benchmarks", "llama_decode.py"), base + ["--p2p-all-reduce"], rank,
                                  world, timeout_s=150)
    else:
        res["p2p"] = {"ok": False, "skipped": "the peer-to-peer all-reduce was not run, or not exact, on these GPUs"}
    return res if rank == 0 else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=320)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--sweep-ms", default="1,8,16,32,128,512,1024,4096",
                    help="batch sizes of the sweep (1 must be among them); counter passes use 1,128 so that "
                         "each kernel's launches are of one size")
    ap.add_argument("--no-decode", action="store_true", help="skip the Llama-2-7B-shaped decode harness and the Mixtral-shaped experts block")
    ap.add_argument("--no-p2p", action="store_true", help="N > 1: skip the peer-to-peer all-reduce comparison (child processes)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo for rehearsals")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    local %= max(1, torch.cuda.device_count())  # rehearsals may put several ranks on one GPU
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # RCCL on ROCm
        else:
            dist.init_process_group(args.backend)
    p = world
    K, N, G, M = K_IN, N_OUT, GROUP, M_HEADLINE
    if N % (8 * p) or (K // p) % G:
        raise SystemExit(f"{K}x{N} g{G} cannot be sharded {p} ways")

    # Weight sets: column shards [K, N/p] and row shards [K/p, N] of the same layers.
    col, row = [], []
    for i in range(NBUF):
        qw, s, qz = make_weights(K, N, G, seed=i, device=device)
        nc, n = N // 8 // p, N // p
        col.append(ops.awq_repack(qw[:, rank * nc:(rank + 1) * nc].contiguous(),
                                  s[:, rank * n:(rank + 1) * n].contiguous(),
                                  qz[:, rank * nc:(rank + 1) * nc].contiguous()))
        if p == 1:
            row.append(col[-1])
        else:
            kk, gg = K // p, K // G // p
            row.append(ops.awq_repack(qw[rank * kk:(rank + 1) * kk].contiguous(),
                                      s[rank * gg:(rank + 1) * gg].contiguous(),
                                      qz[rank * gg:(rank + 1) * gg].contiguous()))
        del qw, s, qz
    if rank == 0:
        print(f"[bench] {NBUF} weight sets packed on {world} rank(s)", file=sys.stderr, flush=True)
    x_full = make_x(M, K, seed=1234, device=device)
    x_row = x_full[:, rank * (K // p):(rank + 1) * (K // p)].contiguous()
    torch.cuda.synchronize()

    stream = torch.cuda.Stream(device=device)
    group = dist.group.WORLD if world > 1 else None

    # The collective of the step.  With RCCL it is called directly (rccl_utils), so that the whole step
    # can be captured in a HIP graph like the N=1 step (torch.distributed's RCCL path cannot be captured:
    # its watchdog thread aborts the process); if that set-up fails, torch.distributed launched eagerly.
    collective = "torch.distributed, launched eagerly"
    direct = None
    if world > 1 and args.backend == "nccl":
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
            probe = torch.full((1024, ), float(rank + 1), dtype=torch.float16, device=device)
            rccl_utils.all_reduce(probe)
            torch.cuda.synchronize()
            if not bool((probe == world * (world + 1) / 2).all()):
                raise RuntimeError("direct RCCL all-reduce returned a wrong sum")
            direct = rccl_utils
            collective = "RCCL called directly, step replayed from a HIP graph"
        except Exception as e:  # noqa: BLE001
            print(f"[bench] direct RCCL communicator unavailable ({e}); eager torch.distributed", file=sys.stderr,
                  flush=True)
        flags = [None] * world
        dist.all_gather_object(flags, direct is not None)
        if not all(flags):
            direct, collective = None, "torch.distributed, launched eagerly"

    def all_reduce(y):
        if direct is not None:
            direct.all_reduce(y)  # in place, on the current stream
        elif args.backend == "nccl":
            dist.all_reduce(y, group=group)  # RCCL, in place, ordered after the GEMM on this stream
        else:  # rehearsal backends (gloo): reduce a host copy
            h = y.cpu()
            dist.all_reduce(h, group=group)
            y.copy_(h)

    # Outputs are allocated once; the step calls the C ABI directly (3 us of host time per launch
    # instead of 10 through the operator wrapper), so that a step of three short GPU operations is
    # not bound by the host when it is launched eagerly (N > 1).
    from vllm_quick_amd._lib import check, lib
    y_col = torch.empty((M, N // p), dtype=torch.float16, device=device)
    y_row = torch.empty((M, N), dtype=torch.float16, device=device)

    def gemm_into(x, pk, out):
        check(lib.w4a16_gemm_packed_f16(x.data_ptr(), pk.packed_w.data_ptr(), pk.packed_sz.data_ptr(), None,
                                        out.data_ptr(), x.shape[0], pk.K, pk.N, pk.G,
                                        torch.cuda.current_stream().cuda_stream), "w4a16_gemm_packed_f16")
        return out

    def step(i):
        gemm_into(x_full, col[i % NBUF], y_col)  # column-parallel: no collective
        y = gemm_into(x_row, row[i % NBUF], y_row)  # row-parallel partial product
        if world > 1:
            all_reduce(y)
        return y

    # ---- timed region: exactly `steps` steps, barrier + synchronize on both sides ----------
    # On one GPU the steps are replayed from a HIP graph (NBUF steps per replay, the remainder
    # launched directly) so that host launch cost cannot bound a ~10 us kernel; so are the steps with
    # the directly called RCCL all-reduce in them.  Through torch.distributed they are launched one by one.
    graph = None
    if world == 1:
        graph = graph_of([(lambda i=i: step(i)) for i in range(NBUF)], stream)
    elif direct is not None:
        try:
            graph = graph_of([(lambda i=i: step(i)) for i in range(NBUF)], stream)
        except Exception as e:  # noqa: BLE001 - same collective sequence either way, so ranks may differ here
            collective = f"RCCL called directly, launched eagerly (capture failed: {type(e).__name__})"
            print(f"[bench] rank {rank}: capturing the step failed: {e}", file=sys.stderr, flush=True)

    def run_steps(n):
        done = 0
        if graph is not None:
            while done + NBUF <= n:
                graph.replay()
                done += NBUF
        for i in range(done, n):
            step(i)

    with torch.cuda.stream(stream):
        run_steps(args.warmup)
        stream.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(args.steps)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        wall = time.perf_counter() - t0
    t = torch.tensor([wall], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t.item())
    ms_per_step = wall * 1e3 / args.steps
    flops_step = 2 * (2.0 * M * K * N)
    value = flops_step / (ms_per_step * 1e-3) * 1e-12

    # ---- the two halves of the step, and the kernel alone, via graph replays + HIP events ---
    if rank == 0:
        print(f"[bench] timed region done: {ms_per_step:.4f} ms/step", file=sys.stderr, flush=True)
    launches = max(NBUF, args.steps)
    col_us = time_launches([(lambda i=i: ops.awq_gemm_packed(x_full, col[i])) for i in range(NBUF)], stream,
                           launches, args.warmup)
    row_us = time_launches([(lambda i=i: ops.awq_gemm_packed(x_row, row[i])) for i in range(NBUF)], stream,
                           launches, args.warmup)
    tp = {"column_gemm_us": col_us, "row_gemm_us": row_us, "all_reduce_us": 0.0, "collective": "none (one rank)",
          "column_shard": [K, N // p], "row_shard": [K // p, N],
          "column_roofline": shard_roofline(M, K, N // p, G, col_us),
          "row_roofline": shard_roofline(M, K // p, N, G, row_us)}
    if world > 1:
        with torch.cuda.stream(stream):
            y = ops.awq_gemm_packed(x_row, row[0])
            for _ in range(8):
                all_reduce(y)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(64):
                all_reduce(y)
            e1.record(stream)
            e1.synchronize()
        tp["all_reduce_us"] = e0.elapsed_time(e1) * 1e3 / 64
        tp["all_reduce_bytes"] = 2 * M * N
        tp["collective"] = collective

    p2p = decode_tp = None
    if world > 1 and not args.no_p2p:
        p2p = p2p_all_reduce_section(args.backend, rank, world)
    if world > 1 and not args.no_decode:
        # every rank must take the same branch: rank 0 holds the verdict of the comparison above
        verdict = [bool(p2p and p2p.get("ok") and p2p.get("ok_all_ranks", True) and not p2p.get("timed_out"))]
        dist.broadcast_object_list(verdict, src=0)
        decode_tp = decode_tp_section(args.backend, rank, world, with_p2p=verdict[0])

    out = {
        "metric": "W4A16 GEMM TFLOPS (4096x11008 g128, M=128)",
        "value": value,
        "unit": "TFLOP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f16",
        "data": "synthetic",
        "config": {
            "workload": f"kernel-only W4A16 linear {K}x{N} group {G}, M={M}; step = column-parallel GEMM + "
                        f"row-parallel GEMM (+ all-reduce), {NBUF} rotating weight sets",
            "M": M, "K": K, "N": N, "group_size": G,
            "parallelism": f"tp{world}",
        },
        "tp": tp,
    }
    if p2p is not None:
        out["p2p_all_reduce"] = p2p
    if decode_tp is not None:
        out["decode_tp"] = decode_tp

    # roofline of the dominant kernel of the timed region: the M=128 GEMM on this rank's column shard
    # (at N=1 the whole layer).  Algorithmic flops per launch / mean launch duration.
    traffic = load_traffic()
    mfma = load_mfma()
    n_local = N // p
    flops_launch = 2.0 * M * K * n_local
    from vllm_quick_amd._lib import launch_shape
    kernel = launch_shape(M, K, n_local)
    out["roofline"] = {
        "kernel": f"{kernel} (the M={M} launch w4a16_gemm_packed_f16 makes on "
                  + ("the whole layer)" if p == 1 else f"this rank's [{K}, {n_local}] column shard)"),
        "bound": "mfma",
        "achieved": flops_launch / col_us * 1e-6,
        "peak": PEAK_F16_TFLOPS,
        "unit": "TFLOP/s",
        "frac": flops_launch / col_us * 1e-6 / PEAK_F16_TFLOPS,
        "traffic": traffic_of(traffic, "m128", kernel) if p == 1 else None,
        "traffic_source": traffic.get("source") if p == 1 else None,
        # counter-based: share of the matrix pipes' cycles in use while their CU is busy (see profiles/README.md);
        # `frac` above divides by the 2.4 GHz data-sheet peak, the counters tick with the clock the chip holds
        "mfma_busy": mfma_busy_of(mfma, M, kernel) if p == 1 else None,
        "mfma_busy_source": mfma.get("source") if p == 1 else None,
        "algorithmic_bytes": algorithmic_bytes(M, K, n_local, G),
        "launch_us": col_us,
    }

    if world == 1 and rank == 0:
        if not args.no_sweep:
            sweep = {}
            # the stock-layout kernel (w4a16_gemm_f16): what a call costs when no packed copy can be made
            # (first call on a weight while the stream is being captured)
            stock = [make_weights(K, N, G, seed=i, device=device) for i in range(4)]
            for m in sorted({1} | {int(v) for v in args.sweep_ms.split(",")}):
                xm = make_x(m, K, seed=m, device=device)
                n_l = launches if m <= 512 else NBUF
                us = time_launches([(lambda i=i: ops.awq_gemm_packed(xm, col[i])) for i in range(NBUF)], stream,
                                   n_l, args.warmup if m <= 512 else NBUF)
                by, fl = algorithmic_bytes(m, K, N, G), 2.0 * m * K * N
                sweep[str(m)] = {"kernel": launch_shape(m, K, N), "us": us, "GBps": by / us * 1e-3,
                                 "TFLOPs": fl / us * 1e-6, "hbm_frac": by / us * 1e-3 / PEAK_HBM_GBPS,
                                 "mfma_frac": fl / us * 1e-6 / PEAK_F16_TFLOPS,
                                 "mfma_busy": mfma_busy_of(mfma, m, launch_shape(m, K, N))}
                if m in (1, 16, 128):
                    sweep[str(m)]["stock_layout_kernel_us"] = time_launches(
                        [(lambda w=w: ops.awq_gemm_stock(xm, *w)) for w in stock], stream, 32, 8)
            del stock
            out["sweep"] = sweep
            # bf16 activations on the same packed weights (an extension: the reference is fp16 only)
            bf = {}
            for m in (1, 16, 128, 1024):
                xb = make_x(m, K, seed=m, device=device).to(torch.bfloat16)
                us = time_launches([(lambda i=i: ops.awq_gemm_packed_bf16(xb, col[i])) for i in range(NBUF)], stream,
                                   launches if m <= 512 else NBUF, args.warmup if m <= 512 else NBUF)
                bf[str(m)] = {"us": us, "GBps": algorithmic_bytes(m, K, N, G) / us * 1e-3,
                              "TFLOPs": 2.0 * m * K * N / us * 1e-6, "fp16_us": sweep[str(m)]["us"] if str(m) in sweep else None}
            out["sweep_bf16"] = bf
            m1 = sweep["1"]
            # floor of the decode launch on this machine: one pass of 16-byte loads over buffers of the
            # layer's packed size, rotated like the weight sets, nothing else in the kernel
            nbytes = col[0].packed_w.numel() * col[0].packed_w.element_size() \
                + col[0].packed_sz.numel() * col[0].packed_sz.element_size()
            bufs = [torch.randint(0, 255, (nbytes, ), dtype=torch.uint8, device=device) for _ in range(NBUF)]
            sink = torch.zeros((4096, ), dtype=torch.int32, device=device)

            def read_all(b):
                check(lib.w4a16_stream_read(b.data_ptr(), nbytes, sink.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream), "w4a16_stream_read")
            floor_us = time_launches([(lambda b=b: read_all(b)) for b in bufs], stream, launches, args.warmup)
            del bufs
            out["roofline_m1"] = {
                "kernel": f"{launch_shape(1, K, N)} (the M=1 launch w4a16_gemm_packed_f16 makes)",
                "streaming_floor_us": floor_us,
                "streaming_floor_GBps": nbytes / floor_us * 1e-3,
                "frac_of_streaming_floor": floor_us / m1["us"],
                "bound": "hbm",
                "achieved": m1["GBps"],
                "peak": PEAK_HBM_GBPS,
                "unit": "GB/s",
                "frac": m1["hbm_frac"],
                "traffic": traffic_of(traffic, "m1", launch_shape(1, K, N)),
                "algorithmic_bytes": algorithmic_bytes(1, K, N, G),
                "launch_us": m1["us"],
            }
            # K split over workgroups on a layer with a long K and too few columns to fill the chip (1 row): the
            # plain launch, the launch with a caller workspace, and the entry point that runs on the library's
            # own device array
            Ks, Ns = 14336, 512
            narrow = [ops.awq_repack(*make_weights(Ks, Ns, G, seed=50 + i, device=device)) for i in range(NBUF)]
            x1 = make_x(1, Ks, seed=9, device=device)
            ws = ops.gemm_workspace(device, 1 << 20)
            o1 = torch.empty((1, Ns), dtype=torch.float16, device=device)

            def static_call(pk):
                check(lib.w4a16_gemm_packed_f16_decode_static(
                    x1.data_ptr(), 0, pk.packed_w.data_ptr(), pk.packed_sz.data_ptr(), None, o1.data_ptr(), 0, 1, Ks,
                    Ns, G, 1, 0, 0, 0, 0, 0, torch.cuda.current_stream().cuda_stream), "decode_static")
            shape_ws = (ctypes.c_int * 5)()
            check(lib.w4a16_launch_shape_ws(1, Ks, Ns, G, 1, 0, shape_ws), "launch_shape_ws")
            out["split_k"] = {
                "layer": f"{Ks}x{Ns}, 1 row", "kernel": launch_shape(1, Ks, Ns), "splits_with_workspace": shape_ws[4],
                "plain_us": time_launches([(lambda pk=pk: ops.awq_gemm_packed(x1, pk)) for pk in narrow], stream,
                                          launches, args.warmup),
                "workspace_us": time_launches([(lambda pk=pk: ops.awq_gemm_packed(x1, pk, workspace=ws))
                                               for pk in narrow], stream, launches, args.warmup),
                "static_scratch_us": time_launches([(lambda pk=pk: static_call(pk)) for pk in narrow], stream,
                                                   launches, args.warmup),
            }
            del narrow
            # context: the dense fp16 GEMM torch dispatches to (hipBLASLt) on the same shape
            wd = torch.randn((K, N), device=device).half()
            us = time_launches([(lambda: torch.matmul(x_full, wd))] * 4, stream, 64, 8)
            out["context"] = {"dense_fp16_matmul_m128_us": us, "dense_fp16_matmul_m128_tflops": 2.0 * M * K * N / us * 1e-6}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(M, K, N, G)
        if not args.no_decode:
            # second half of the headline metric: Llama-2-7B-AWQ-shaped greedy decode, tokens/s (TP=1)
            del col, row
            torch.cuda.empty_cache()
            from benchmarks.llama_decode import run as decode_run
            out["decode"] = decode_run("7b", (1, 8, 64), device=device, keep_stock=False)
            torch.cuda.empty_cache()
            # compute half of the TP=8 metric: one rank's step of the Llama-2-70B-shaped model (8 of 80 layers run,
            # extrapolated), every all-reduce a device copy of the same size; a run on 8 GPUs adds the collectives
            out["decode_tp8_rank_compute"] = decode_run("70b", (1, 8, 64), n_layers=8, device=device, keep_stock=False,
                                                        rehearse_tp=8)
            torch.cuda.empty_cache()
            # small-M batched path: Mixtral-8x7B-shaped expert block, batched launches vs per-expert loop
            from benchmarks.mixtral_experts import run as experts_run
            out["experts"] = experts_run((1, 16), device=device)

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        if direct is not None:
            direct.destroy_process_group()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
