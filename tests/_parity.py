"""Parity criterion shared by the GPU tests: the tolerance the task states for this floating-point path
(rtol 1e-2 / atol 1e-3 against the reference-faithful arithmetic), with the reference arithmetic's own
rounding bounded per element instead of by a share of outliers."""
import numpy as np

RTOL, ATOL = 1e-2, 1e-3


def half_ulp16(v):
    """Half a unit in the last place of the fp16 value nearest to v (float64 array)."""
    h = np.abs(np.asarray(v)).astype(np.float16)
    h = np.minimum(h, np.float16(65504.0))
    return np.spacing(h).astype(np.float64) / 2


def reference_arithmetic(x, w16):
    """(faithful, exact, slack) from dequantised weights `w16` (fp16 [K, N]).

    faithful: split-K of 8 over interleaved 32-wide chunks, fp16 slices, fp32 sum, fp16 result
    (gemm_kernels.cu:112-115, 262-270, 445).  exact: float64.  slack: how far the faithful value can be
    from a singly rounded fp32 sum through its own nine roundings (per element)."""
    K = w16.shape[0]
    xf, w32 = x.astype(np.float32), w16.astype(np.float32)
    sl = (np.arange(K) // 32) % 8
    tot = np.zeros((x.shape[0], w16.shape[1]), np.float32)
    slack = np.zeros(tot.shape, np.float64)
    for z in range(8):
        if not (sl == z).any():
            continue
        part = (xf[:, sl == z] @ w32[sl == z]).astype(np.float16)
        slack += half_ulp16(part)
        tot += part.astype(np.float32)
    faithful = tot.astype(np.float16)
    exact = x.astype(np.float64) @ w16.astype(np.float64)
    slack += half_ulp16(faithful) + half_ulp16(exact)
    return faithful, exact, slack


def assert_parity(got, faithful, exact, slack, what):
    """Returns the number of outputs that needed the oracle's own rounding slack."""
    got = got.astype(np.float64)
    f = faithful.astype(np.float64)
    d = np.abs(got - f)
    out = d > ATOL + RTOL * np.abs(f)
    n_out = int(out.sum())
    if n_out:
        # fp32 accumulation order (kernel vs numpy): a few 2^-24 of the magnitudes summed, far below half an ulp
        assert (d[out] <= slack[out] * 1.01 + 1e-6).all(), \
            f"{what}: {int((d[out] > slack[out] * 1.01 + 1e-6).sum())} outputs beyond the oracle's rounding slack"
        assert (np.abs(got - exact)[out] <= ATOL + RTOL * np.abs(exact[out])).all(), what
    # the kernel rounds once: it must be no further from float64 than the reference arithmetic is
    assert np.abs(got - exact).max() <= np.abs(f - exact).max() * 1.05 + 2.0**-11, what
    return n_out
