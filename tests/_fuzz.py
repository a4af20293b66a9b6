"""Randomised parity cases shared by the collected GPU test (tests/test_fuzz_gpu.py: fixed seeds, a fixed number of
cases) and the developer sweep (tests/dev/fuzz_gpu.py: a time budget): random shapes, group sizes, row counts, entry
points and FORCED launch shapes out of what the library reports as built, against the oracle."""
import time

import numpy as np
import torch

from oracle import awq_oracle as orc
from vllm_quick_amd import _lib
from vllm_quick_amd._C import ops

dev = torch.device("cuda:0")
RT, AT = 1e-2, 1e-3
MODES = ("gemm", "stock", "dequant", "batched", "act", "act_batched", "cfg", "cfg", "act_cfg", "decode", "decode",
         "wide_split")


def built(kernel):
    return [tuple(int(v) for v in s) for s in _lib.built_shapes(kernel)]


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def close(got, want, exact, what):
    """The policy of tests/test_gemm_gpu.py (assert_parity): within rtol/atol of the reference-faithful
    result, except for a few outputs where that result's own fp16 slice rounding (it grows with K and
    |y|) exceeds the bound; those must be within the bound of the fp64 result, and overall the kernel
    must be no further from fp64 than the reference arithmetic is."""
    got = got.astype(np.float64)
    bad = np.abs(got - want) > AT + RT * np.abs(want)
    if bad.sum() == 0:
        return True
    ok = bad.sum() <= max(2, 1e-3 * bad.size) and (np.abs(got - exact)[bad] <= AT + RT * np.abs(exact[bad])).all() \
        and np.abs(got - exact).max() <= np.abs(want - exact).max() * 1.05 + 2.0**-11 * max(1.0, np.abs(exact).max())
    if not ok:
        print(f"FAIL {what}: {bad.sum()} of {bad.size} outside tolerance, max|d| {np.abs(got - want).max():.3e}, "
              f"vs fp64 {np.abs(got - exact).max():.3e} (reference arithmetic {np.abs(want - exact).max():.3e})", flush=True)
    return ok




def one_case(rng, modes=MODES, large=False):
    """Draws and runs one case.  Returns (description, ok); ok is None for a draw the library refuses by design."""
    LARGE = large
    STREAM = built(0)
    TILED = built(1)
    WIDE, SHARED = built(2), built(3)
    ACT = [(mt, t, w, 0) for mt, t, w in STREAM if t % 2 == 0 and mt <= 2] + [c for c in TILED if c[1] == 3 and c[2] % 2 == 0]
    G = int(rng.choice([32, 64, 128, 256]))
    K = G * int(rng.integers(1, 24))
    N = 8 * int(rng.integers(1, 120))
    if LARGE:  # model-sized layers (fewer cases: the CPU oracle takes seconds on these)
        G = 128
        K, N = [(4096, 1376), (4096, 2752), (11008, 4096), (4096, 4096), (2048, 11008), (4096, 5504)][rng.integers(6)]
    M = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 24, 31, 32, 33, 48, 63, 64, 65, 100, 128, 129, 200, 300]))
    mode = rng.choice(list(modes))
    if mode == "wide_split":  # whole 128-k blocks
        M = int(rng.integers(1, 200))
        K = 128 * int(rng.integers(1, 24)) if G <= 128 else 256 * int(rng.integers(1, 12))
    if mode == "decode":  # the decode kernel with a forced shape: up to 16 rows (32 with two row tiles), whole 128-k blocks
        M = int(rng.integers(1, 33))
        K = 128 * int(rng.integers(1, 40)) if G <= 128 else G * int(rng.integers(1, 20))
        N = 16 * int(rng.integers(1, 60))
    x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=int(rng.integers(1 << 30)))
    what = f"{mode} M={M} K={K} N={N} G={G}"
    try:
        if mode == "dequant":
            w = orc.dequantize(qw, s, qz)
            a = ops.awq_dequantize(t(qw), t(s), t(qz), 0, 0, 0).cpu().numpy()
            b = ops.awq_dequantize_packed(ops.awq_repack(t(qw), t(s), t(qz))).cpu().numpy()
            ok = (a.view(np.uint16) == w.view(np.uint16)).all() and (b.view(np.uint16) == w.view(np.uint16)).all()
            if not ok:
                print(f"FAIL {what}: dequantise not bit-exact", flush=True)
        elif mode in ("gemm", "stock"):
            want, exact = orc.gemm_faithful(x, qw, s, qz).astype(np.float64), orc.gemm_exact(x, qw, s, qz)
            fn = ops.awq_gemm if mode == "gemm" else (lambda a, b, c, d, _: ops.awq_gemm_stock(a, b, c, d))
            ok = close(fn(t(x), t(qw), t(s), t(qz), 8).float().cpu().numpy(), want, exact, what)
        elif mode == "cfg":
            # a forced launch shape out of the built tables (row tiles must cover M for the streaming kernel)
            want, exact = orc.gemm_faithful(x, qw, s, qz).astype(np.float64), orc.gemm_exact(x, qw, s, qz)
            fam = int(rng.integers(4))
            if fam == 0:
                cfg = STREAM[rng.integers(len(STREAM))]
            elif fam == 1:
                cfg = TILED[rng.integers(len(TILED))]
            elif fam == 2:  # 32x32x16, private slabs (whole 128-k blocks: refused otherwise)
                cfg = ("wide", ) + WIDE[rng.integers(len(WIDE))]
            else:           # 32x32x16, shared slab: ("wide", mt, t, -nw)
                mt_, t_, nw_ = SHARED[rng.integers(len(SHARED))]
                cfg = ("wide", mt_, t_, -nw_)
            what += f" cfg={cfg}"
            pk = ops.awq_repack(t(qw), t(s), t(qz))
            try:
                y = ops.awq_gemm_packed(t(x), pk, cfg=cfg)
            except ValueError:
                return what, None  # shape/launch combination the library refuses (e.g. K blocks not divisible by kw)
            ok = close(y.float().cpu().numpy(), want, exact, what)
        elif mode == "decode":
            want, exact = orc.gemm_faithful(x, qw, s, qz).astype(np.float64), orc.gemm_exact(x, qw, s, qz)
            shapes = _lib.built_shapes(4)
            if M > 16:  # two row tiles: one column tile x 16 waves
                shapes = [c for c in shapes if c[1] == 16 and c[0] == 1]
            tt, ww, _ = shapes[rng.integers(len(shapes))]
            splits = int(rng.integers(1, 5)) if M <= 4 and tt <= 4 else 1
            cfg = ("decode", int(tt), int(ww), splits, int(rng.integers(0, 3)))
            what += f" cfg={cfg}"
            pk = ops.awq_repack(t(qw), t(s), t(qz))
            y = ops.awq_gemm_packed(t(x), pk, cfg=cfg, workspace=ops.gemm_workspace(dev) if splits > 1 else None)
            ok = close(y.float().cpu().numpy(), want, exact, what)
        elif mode == "wide_split":  # the private-slab 32x32x16 kernel with K split over workgroups
            want, exact = orc.gemm_faithful(x, qw, s, qz).astype(np.float64), orc.gemm_exact(x, qw, s, qz)
            four = [c for c in WIDE if c[2] == 4]
            cfg = ("wide", ) + four[rng.integers(len(four))] + (int(rng.integers(2, 5)), )
            what += f" cfg={cfg}"
            pk = ops.awq_repack(t(qw), t(s), t(qz))
            y = ops.awq_gemm_packed(t(x), pk, cfg=cfg)
            ok = close(y.float().cpu().numpy(), want, exact, what)
        elif mode == "batched":
            E = int(rng.integers(2, 5))
            ws = [orc.make_inputs(M, K, N, G, seed=int(rng.integers(1 << 30))) for _ in range(E)]
            shared = bool(rng.integers(2)) or (M * K) % 8 != 0
            if (M * N) % 4:
                return what, None
            pk = ops.awq_repack_batch([t(w[1]) for w in ws], [t(w[2]) for w in ws], [t(w[3]) for w in ws])
            xs = ws[0][0] if shared else np.stack([w[0] for w in ws])
            y = ops.awq_gemm_packed_batched(t(xs), pk).float().cpu().numpy()
            ok = all(close(y[e], orc.gemm_faithful(xs if shared else xs[e], *ws[e][1:]).astype(np.float64),
                           orc.gemm_exact(xs if shared else xs[e], *ws[e][1:]), what + f" e={e}") for e in range(E))
        else:
            N = 32 * max(1, N // 32)
            x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=int(rng.integers(1 << 30)))
            what = f"{mode} M={M} K={K} N={N} G={G}"
            y = orc.gemm_faithful(x, qw, s, qz)
            g, u = torch.from_numpy(y[:, :N // 2]), torch.from_numpy(y[:, N // 2:])
            want = (torch.nn.functional.silu(g.float()).half() * u).numpy().astype(np.float64)
            gd, ud = g.numpy().astype(np.float64), u.numpy().astype(np.float64)
            sg = gd / (1 + np.exp(-gd))
            tol = 1.1 * np.abs(ud) * (AT + RT * np.abs(gd)) + np.abs(sg) * (AT + RT * np.abs(ud)) + 2.0**-10 * np.abs(sg * ud) + 1e-6
            pk = ops.awq_repack(t(qw), t(s), t(qz))
            if mode == "act":
                got = ops.awq_gemm_packed_silu_and_mul(t(x), pk).float().cpu().numpy()
            elif mode == "act_cfg":
                cfg = ACT[rng.integers(len(ACT))]
                what += f" cfg={cfg}"
                try:
                    got = ops.awq_gemm_packed_silu_and_mul(t(x), pk, cfg=cfg).float().cpu().numpy()
                except ValueError:
                    return what, None
            else:
                if (M * (N // 2)) % 4:
                    return what, None
                pkb = ops.awq_repack_batch([t(qw)] * 2, [t(s)] * 2, [t(qz)] * 2)
                got = ops.awq_gemm_packed_silu_and_mul(t(x), pkb).float().cpu().numpy()[1]
            err = np.abs(got - want)
            ok = bool((err > tol).sum() <= max(2, 1e-4 * err.size) and (err <= 4 * tol).all())
            if not ok:
                print(f"FAIL {what}: {(err > tol).sum()} outside tolerance", flush=True)
    except Exception as e:  # noqa: BLE001
        ok = False
        print(f"ERROR {what}: {type(e).__name__}: {e}", flush=True)
    return what, bool(ok)


def sweep(seed, max_cases=None, budget_s=None, modes=MODES, large=False, verbose=False):
    """Runs cases until `max_cases` were drawn or `budget_s` seconds have passed.  Returns (cases run, failures)."""
    rng = np.random.default_rng(seed)
    t0, n, fails = time.time(), 0, []
    while (max_cases is None or n < max_cases) and (budget_s is None or time.time() - t0 < budget_s):
        what, ok = one_case(rng, modes, large)
        if ok is None:
            continue
        n += 1
        if not ok:
            fails.append(what)
        if verbose and n % (5 if large else 50) == 0:
            print(f"[fuzz] {n} cases, {len(fails)} failures, {time.time() - t0:.0f} s", flush=True)
    return n, fails
