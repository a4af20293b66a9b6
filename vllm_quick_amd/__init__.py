"""MI355X-native AWQ W4A16 GEMM path: drop-in for `vllm._C.ops.awq_gemm` /
`awq_dequantize` and `AWQLinearMethod` (reference: vLLM 0.3.0 tree).

Importing the package loads libw4a16.so (hand-written gfx950 HIP kernels behind a
C ABI, see include/w4a16.h) and raises if it has not been built.
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is missing)
from ._C import ops  # noqa: F401

__version__ = "0.1.0"
