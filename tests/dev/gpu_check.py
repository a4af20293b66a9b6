"""Developer check on a GPU box: parity of every kernel against the oracle on a few
shapes, then a timing sweep over launch shapes.  Not part of the product; the
pytest suite and bench.py are the maintained entry points.  It lives under tests/ because it
computes with the oracle.

usage: python tests/dev/gpu_check.py [--quick] [--sweep]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import awq_oracle as orc  # noqa: E402
from vllm_quick_amd._C import ops  # noqa: E402
from scripts._timing import time_graph  # noqa: E402

dev = torch.device("cuda:0")


def to_dev(*arrs):
    return [torch.from_numpy(a).to(dev) for a in arrs]


def check_shape(M, K, N, G, seed=0, cfgs=(None,)):
    x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=seed)
    want = orc.gemm_faithful(x, qw, s, qz)
    exact = orc.gemm_exact(x, qw, s, qz)
    wdq = orc.dequantize(qw, s, qz)
    tx, tqw, ts, tqz = to_dev(x, qw, s, qz)
    ok = True
    dq = ops.awq_dequantize(tqw, ts, tqz, 0, 0, 0).cpu().numpy()
    bit = (dq.view(np.uint16) == wdq.view(np.uint16)).all()
    packed = ops.awq_repack(tqw, ts, tqz)
    dq2 = ops.awq_dequantize_packed(packed).cpu().numpy()
    bit2 = (dq2.view(np.uint16) == wdq.view(np.uint16)).all()
    ok &= bool(bit) and bool(bit2)
    print(f"[{M}x{K}x{N} G{G}] dequant bit-exact: stock={bit} packed={bit2}")
    outs = {"stock": ops.awq_gemm_stock(tx, tqw, ts, tqz)}
    for cfg in cfgs:
        try:
            outs[f"packed{cfg}"] = ops.awq_gemm_packed(tx, packed, cfg=cfg)
        except Exception as e:  # noqa: BLE001
            print(f"   packed{cfg}: {type(e).__name__}: {e}")
    torch.cuda.synchronize()
    for name, o in outs.items():
        o = o.cpu().numpy()
        close = np.isclose(o.astype(np.float64), want.astype(np.float64), rtol=1e-2, atol=1e-3)
        err_f = np.abs(o.astype(np.float64) - want.astype(np.float64)).max()
        err_e = np.abs(o.astype(np.float64) - exact).max()
        ref_e = np.abs(want.astype(np.float64) - exact).max()
        good = bool(close.all())
        ok &= good
        print(f"   {name:24s} parity={'OK ' if good else 'BAD'} mismatch={(~close).mean():.2e} "
              f"max|d| vs faithful={err_f:.3e} vs exact={err_e:.3e} (faithful vs exact {ref_e:.3e})")
    return ok


def sweep(K=4096, N=11008, G=128, nbuf=16, only_m=None, cfg_override=None):
    print(f"timing sweep K={K} N={N} G={G}, {nbuf} rotating weight sets")
    packs = []
    for i in range(nbuf):
        _, qw, s, qz = orc.make_inputs(1, K, N, G, seed=100 + i)
        tqw, ts, tqz = to_dev(qw, s, qz)
        packs.append(ops.awq_repack(tqw, ts, tqz))
    torch.cuda.synchronize()
    wbytes = K * N // 2 + (K // G) * N * 2 + (K // G) * (N // 8) * 4
    cfg_by_m = {
        1: [(1, 3, 16), (1, 3, 8), (1, 2, 16), (1, 4, 16), (1, 6, 16), (1, 1, 16), (1, 1, 8), (1, 2, 8), (1, 3, 4)],
        8: [(1, 3, 16), (1, 3, 8), (1, 2, 16)],
        16: [(1, 3, 16), (1, 3, 8), (1, 2, 16), (1, 4, 16)],
        32: [(2, 3, 16), (2, 3, 8), (2, 2, 16), (2, 4, 16), (2, 6, 8)],
        128: [(8, 3, 8), (8, 2, 8), (8, 1, 8), (8, 3, 4), (8, 2, 4), (4, 6, 8), (4, 4, 8), (4, 3, 8), (4, 3, 4),
              (2, 6, 8), (2, 6, 16), (2, 4, 16)],
        512: [(8, 3, 8), (8, 2, 8), (4, 6, 8)],
    }
    for M, cfgs in cfg_by_m.items():
        if only_m and M not in only_m:
            continue
        if cfg_override:
            cfgs = cfg_override
        x = (torch.randn(M, K, device=dev) * 0.25).half()
        by = wbytes + 2 * M * K + 2 * M * N
        fl = 2.0 * M * K * N
        for cfg in cfgs:
            try:
                fns = [(lambda p=p: ops.awq_gemm_packed(x, p, cfg=cfg)) for p in packs]
                us = time_graph(fns)
                print(f"  M={M:4d} cfg={cfg}: {us:8.2f} us  {by / us * 1e-3:7.1f} GB/s  {fl / us * 1e-6:8.1f} TFLOP/s")
            except Exception as e:  # noqa: BLE001
                print(f"  M={M:4d} cfg={cfg}: {type(e).__name__}: {e}")
        try:
            fns = [(lambda p=p: ops.awq_gemm_packed(x, p)) for p in packs]
            us = time_graph(fns)
            print(f"  M={M:4d} default   : {us:8.2f} us  {by / us * 1e-3:7.1f} GB/s  {fl / us * 1e-6:8.1f} TFLOP/s")
        except Exception as e:  # noqa: BLE001
            print(f"  M={M:4d} default: {type(e).__name__}: {e}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--only-m", type=str, default="")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--cfgs", type=str, default="", help="e.g. 4,6,8;8,3,8")
    a = ap.parse_args()
    print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count, "CUs")
    ok = True
    t0 = time.time()
    cfgo = [tuple(int(v) for v in c.split(",")) for c in a.cfgs.split(";")] if a.cfgs else None
    if a.no_parity:
        sweep(only_m=[int(v) for v in a.only_m.split(",")] if a.only_m else None, cfg_override=cfgo)
        return 0
    ok &= check_shape(1, 256, 128, 128, cfgs=[None, (1, 1, 16), (1, 2, 8), (1, 1, 4)])
    ok &= check_shape(3, 256, 192, 64, cfgs=[None, (1, 3, 16), (1, 4, 4)])
    ok &= check_shape(17, 512, 256, 128, cfgs=[None, (2, 2, 16), (2, 3, 8), (4, 1, 4)])
    ok &= check_shape(5, 96, 40, 32, cfgs=[None, (1, 1, 4), (1, 3, 16)])
    ok &= check_shape(33, 384, 1376, 128, cfgs=[None, (4, 3, 8), (8, 1, 4), (4, 6, 8)])
    if not a.quick:
        ok &= check_shape(1, 4096, 11008, 128, cfgs=[None, (1, 3, 16), (1, 6, 16), (1, 8, 8)])
        ok &= check_shape(16, 4096, 11008, 128, cfgs=[None, (1, 3, 16)])
        ok &= check_shape(128, 4096, 11008, 128, cfgs=[None, (8, 3, 8), (8, 3, 4), (4, 6, 8), (2, 6, 8)])
        ok &= check_shape(200, 1024, 4096, 128, cfgs=[None, (8, 1, 8), (8, 2, 4)])
    print(f"parity {'ALL OK' if ok else 'FAILURES'} in {time.time() - t0:.1f}s", flush=True)
    if a.sweep:
        sweep(only_m=[int(v) for v in a.only_m.split(",")] if a.only_m else None)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
