"""M = 128 on the 4096 x 11008 layer: the private-slab 32x32x16 shapes, with and without the K split over workgroups.
Checks every shape against a dense fp32 product on the dequantised weights, then times them (graph replays over 16
rotating weight sets)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_quick_amd._C import ops
from bench import make_weights, make_x
from scripts._timing import time_graph

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 128
K, N, G = 4096, 11008, 128
if len(sys.argv) > 3:
    K, N = int(sys.argv[2]), int(sys.argv[3])
ws = ops.gemm_workspace(dev, 32 << 20)
stock = [make_weights(K, N, G, i, dev) for i in range(16)]
packs = [ops.awq_repack(*w) for w in stock]
x = make_x(M, K, 1, dev)
dense = ops.awq_dequantize(*stock[0], 0, 0, 0).float()
want = x.float() @ dense
shapes = [("wide", 2, 3, 4), ("wide", 2, 3, 8), ("wide", 4, 3, 4, 2), ("wide", 2, 3, 4, 2), None, "ws"]
if len(sys.argv) > 4:
    shapes = [eval(v) for v in sys.argv[4:]]


def run(p, cfg):
    if cfg == "ws":
        return ops.awq_gemm_packed(x, p, workspace=ws)
    return ops.awq_gemm_packed(x, p, cfg=cfg, workspace=ws if (cfg and len(cfg) > 4) else None)


ok = []
for cfg in shapes:
    try:
        y = run(packs[0], cfg)
    except Exception as e:  # noqa: BLE001
        print(cfg, "->", type(e).__name__, e, flush=True)
        continue
    err = (y.float() - want).abs().max().item()
    same = all(torch.equal(run(packs[0], cfg), y) for _ in range(5))
    print(f"{str(cfg):28s} max|err| {err:.3e} (|y| max {want.abs().max().item():.2f}) repeat-identical {same}", flush=True)
    ok.append(cfg)
# timing: the shapes in turn, several rounds, so that clock drift hits all of them alike
times = {str(c): [] for c in ok}
for rnd in range(5):
    for cfg in ok:
        times[str(cfg)].append(time_graph([(lambda p=p, c=cfg: run(p, c)) for p in packs], iters=10))
for cfg in ok:
    t = sorted(times[str(cfg)][1:])
    print(f"{str(cfg):28s} min {t[0]:7.2f}  median {t[len(t) // 2]:7.2f} us   {2.0 * M * K * N / t[len(t) // 2] * 1e-6:7.1f} TFLOP/s", flush=True)
assert int(ws[:65536].view(torch.int32).abs().max().item()) == 0, "tickets not returned to zero"
