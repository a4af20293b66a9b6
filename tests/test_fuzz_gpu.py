"""A fixed slice of the randomised parity sweep (tests/_fuzz.py; the developer script tests/dev/fuzz_gpu.py runs the
same cases against a time budget): random shapes, group sizes and row counts through every entry point, and FORCED
launch shapes of every kernel family out of what the library reports as built (K not a multiple of 128, 1-17 rows on
shapes the default choice would never pick for them, every prefetch depth and K split), against the oracle.
Seeds and case counts are fixed, so a failure names a case that fails again.
"""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_fuzz_slice(cuda, seed):
    from _fuzz import sweep
    n, fails = sweep(seed, max_cases=400, budget_s=25.0)
    assert not fails, f"{len(fails)} of {n} cases failed: {fails[:5]}"
    assert n >= 100, f"only {n} cases in the time budget"


@pytest.mark.parametrize("seed", [10, 11])
def test_fuzz_forced_shapes(cuda, seed):
    """Forced launch shapes only (streaming / LDS-staged / both 32x32x16 kernels / decode / K split over workgroups)."""
    from _fuzz import sweep
    n, fails = sweep(seed, max_cases=400, budget_s=25.0, modes=("cfg", "cfg", "act_cfg", "decode", "decode", "wide_split"))
    assert not fails, f"{len(fails)} of {n} cases failed: {fails[:5]}"
    assert n >= 100, f"only {n} cases in the time budget"
