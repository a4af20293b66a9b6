"""The C-ABI library: it loads, exports every symbol include/w4a16.h declares, and its
host-only entry points and argument checks behave (no kernel is launched here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "w4a16.h")


@pytest.fixture(scope="module")
def lib():
    from vllm_quick_amd import _lib
    return _lib


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(w4a16_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported_and_bound(lib):
    names = declared_symbols()
    assert len(names) >= 10
    raw = ctypes.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/w4a16.h but not exported"
        assert n in lib.SYMBOLS, f"{n} has no ctypes signature in vllm_quick_amd/_lib.py"
    assert sorted(lib.SYMBOLS) == names


def test_header_cites_reference_for_replaced_entry_points():
    text = open(HEADER).read()
    assert "gemm_kernels.cu:389-446" in text and "gemm_kernels.cu:332-381" in text


def test_version_and_strerror(lib):
    assert lib.lib.w4a16_abi_version() == lib.ABI_VERSION
    assert lib.lib.w4a16_strerror(0) == b"ok"
    assert b"pack_num = 8" in lib.lib.w4a16_strerror(-1)
    assert b"multiple of 32" in lib.lib.w4a16_strerror(-2)
    assert lib.lib.w4a16_strerror(-99) == b"unknown w4a16 error"


def test_packed_sizes(lib):
    L = lib.lib
    # 4096 x 11008: same bytes as the stock qweight; 4 bytes per (group, column) of scale+zero
    assert L.w4a16_packed_weight_bytes(4096, 11008) == 4096 * 11008 // 2
    assert L.w4a16_packed_sz_bytes(4096, 11008, 128) == 32 * 11008 * 4
    # padding to 16 columns x (a multiple of 4) 128-k blocks
    assert L.w4a16_packed_weight_bytes(96, 40) == 3 * 4 * 1024
    assert L.w4a16_packed_weight_bytes(11008, 4096) == 256 * 88 * 1024
    assert L.w4a16_packed_sz_bytes(96, 40, 32) == 3 * 3 * 16 * 4
    assert L.w4a16_packed_weight_bytes(0, 8) == 0
    assert L.w4a16_packed_sz_bytes(100, 8, 32) == 0


def test_argument_errors_return_codes_without_launching(lib):
    L = lib.lib
    p = ctypes.c_void_p(4096)  # never dereferenced: every case below fails validation first
    assert L.w4a16_gemm_f16(None, p, p, p, p, 1, 128, 64, 128, 8, None) == -4
    assert L.w4a16_gemm_f16(p, p, p, p, p, 1, 128, 60, 128, 8, None) == -1
    assert L.w4a16_gemm_f16(p, p, p, p, p, 1, 128, 64, 48, 8, None) == -2
    assert L.w4a16_gemm_f16(p, p, p, p, p, 1, 160, 64, 64, 8, None) == -3
    assert L.w4a16_gemm_f16(p, p, p, p, p, 0, 128, 64, 128, 8, None) == -5
    assert L.w4a16_dequantize_f16(p, p, p, None, 128, 64, 128, None) == -4
    assert L.w4a16_repack(p, p, p, p, None, 128, 64, 128, None) == -4
    assert L.w4a16_gemm_packed_f16(p, p, p, None, p, 1, 128, 60, 128, None) == -1
    assert L.w4a16_gemm_packed_f16_cfg(p, p, p, None, p, 1, 128, 64, 128, 3, 1, 8, None) == -5  # mt=3 not built
    misaligned = ctypes.c_void_p(4100)
    assert L.w4a16_gemm_packed_f16(misaligned, p, p, None, p, 1, 128, 64, 128, None) == -6
    with pytest.raises(ValueError, match="pack_num"):
        lib.check(-1, "x")
    with pytest.raises(RuntimeError):
        lib.check(-7, "x")


def test_native_binding_module_loads_and_mirrors_the_reference_ops():
    """The compiled pybind11 module imports without a GPU and exposes `ops.awq_gemm` / `ops.awq_dequantize`
    with the reference's positional signatures (csrc/pybind.cpp:52-55, csrc/ops.h:69-82)."""
    from vllm_quick_amd import _C_native, _lib
    assert _C_native.abi_version() == _lib.ABI_VERSION
    assert "arg4" in _C_native.ops.awq_gemm.__doc__ and "arg5" not in _C_native.ops.awq_gemm.__doc__
    assert "arg5" in _C_native.ops.awq_dequantize.__doc__ and "arg6" not in _C_native.ops.awq_dequantize.__doc__
    assert _C_native.stock_kernel_calls() == 0
    import torch
    x = torch.zeros((1, 32), dtype=torch.float16)
    import pytest
    with pytest.raises(RuntimeError, match="GPU"):  # no CPU implementation, as the package's own ops module
        _C_native.ops.awq_gemm(x, torch.zeros((32, 1), dtype=torch.int32), torch.zeros((1, 8), dtype=torch.float16),
                               torch.zeros((1, 1), dtype=torch.int32), 8)
