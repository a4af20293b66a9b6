"""Can a torch.distributed (RCCL) all-reduce be captured in a HIP graph next to the W4A16 launches?
One rank is enough to exercise PyTorch's capture path:  python scripts/exp_rccl_graph.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
y = torch.ones(128, 11008, dtype=torch.float16, device=dev)
dist.all_reduce(y)
torch.cuda.synchronize()
stream = torch.cuda.Stream(device=dev)
for mode in ("thread_local", "global"):
    try:
        with torch.cuda.stream(stream):
            dist.all_reduce(y)
            stream.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream, capture_error_mode=mode):
                for _ in range(16):
                    y.mul_(0.5)
                    dist.all_reduce(y)
            g.replay()
            stream.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                g.replay()
            stream.synchronize()
            print(mode, "ok: %.1f us per (mul + all_reduce)" % ((time.perf_counter() - t0) / 320 * 1e6), flush=True)
    except Exception as e:  # noqa: BLE001
        print(mode, "FAILED:", type(e).__name__, str(e)[:300], flush=True)
dist.destroy_process_group()
