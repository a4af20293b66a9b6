"""Randomised parity sweep on a GPU box: random shapes, group sizes, row counts and entry points (default and
forced launch shapes) against the oracle.  Developer check with a time budget; a fixed slice of the same cases is
collected by pytest (tests/test_fuzz_gpu.py).  Prints a line per failure.

usage: python tests/dev/fuzz_gpu.py [seconds] [seed] [large]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _fuzz import sweep  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
large = len(sys.argv) > 3 and sys.argv[3] == "large"
n, fails = sweep(seed, budget_s=budget, large=large, verbose=True)
print(f"[fuzz] done: {n} cases, {len(fails)} failures")
sys.exit(1 if fails else 0)
