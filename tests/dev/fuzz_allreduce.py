"""Protocol stress of the peer-to-peer all-reduce (developer check, uses the oracle): `world` processes on
GPU 0, hundreds of calls of random size and element type back to back without any host synchronisation
in between (so ranks drift a call apart, the mailbox halves alternate and grids change from call to call),
every result compared bit for bit with the oracle afterwards.

    python tests/dev/fuzz_allreduce.py [world=8] [calls=300] [seed=0]
"""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(rank, world, port, calls, seed, q):
    try:
        from test_allreduce_gpu import TORCH_DTYPES, bits, make_input, to_device, want_bits
        from vllm_quick_amd.parallel_utils import custom_all_reduce as car
        from vllm_quick_amd.parallel_utils import parallel_state as ps
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        ps.initialize_model_parallel(world)
        car.init_custom_ar()
        fa = car.get_handle()
        fa.set_timeout_ms(30000)
        rng = np.random.default_rng(seed)  # the same plan on every rank
        plan = []
        for i in range(calls):
            dtype = ("f32", "f16", "bf16")[rng.integers(3)]
            esz = 4 if dtype == "f32" else 2
            kind = rng.integers(4)
            nbytes = int(rng.integers(1, 64)) * 16 if kind == 0 else int(rng.integers(1, 40000)) * 16 if kind < 3 else \
                int(rng.integers(30000, 260000)) * 16
            plan.append((dtype, nbytes // esz, bool(rng.integers(2))))
        bad = 0
        for lo in range(0, calls, 50):  # batches of 50 calls in flight, checked afterwards
            outs = []
            for i, (dtype, n, inplace) in enumerate(plan[lo:lo + 50], lo):
                ins = [make_input("gauss", dtype, n, 10000 * seed + i, r) for r in range(world)]
                x = to_device(ins[rank], dtype, dev)
                y = fa.all_reduce_reg(x, x if inplace else None)
                outs.append((y, ins, dtype, n))
            for y, ins, dtype, n in outs:
                if not np.array_equal(bits(y), want_bits(ins, dtype)):
                    bad += 1
                    print(f"rank {rank}: MISMATCH {dtype} n={n}", flush=True)
        assert not fa.timed_out(), "a kernel gave up waiting"
        car.destroy_custom_ar()
        dist.destroy_process_group()
        q.put((rank, bad))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, calls, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    print(sorted(res, key=lambda t: t[0]))
    ok = all(isinstance(b, int) and b == 0 for _, b in res)
    print(f"world {world}, {calls} calls, seed {seed}:", "ALL OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)
