"""The C ABI from a host program with no PyTorch in it (tests/native/c_abi_check.cpp, built by
`__graft_entry__.build()`): plain hipMalloc'd buffers, every stock- and packed-layout entry point, results
against the C oracle (rtol 1e-2 / atol 1e-3 for the GEMMs, bit-exact dequantisation), error codes."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
EXE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "c_abi_check")


def test_c_abi_from_a_plain_hip_program(cuda):
    assert os.path.exists(EXE), "run `python -c 'import __graft_entry__ as g; g.build()'` first"
    # a child process (never an exec from this GPU-initialised one)
    run = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    print(run.stdout)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "ALL OK" in run.stdout and "FAIL" not in run.stdout


@pytest.mark.parametrize("world", [2, 8])
def test_p2p_all_reduce_from_plain_hip_processes(cuda, world):
    """tests/native/p2p_abi_check.cpp: one forked process per rank on GPU 0, IPC handles over pipes, results
    bit for bit against CPU sums in the documented order."""
    exe = os.path.join(os.path.dirname(EXE), "p2p_abi_check")
    assert os.path.exists(exe), "run `python -c 'import __graft_entry__ as g; g.build()'` first"
    run = subprocess.run([exe, str(world)], capture_output=True, text=True, timeout=200)
    print(run.stdout)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "ALL OK" in run.stdout and "MISMATCH" not in run.stdout
