"""Stand-in for the reference's `vllm._C` extension module: only the AWQ operators."""
from . import ops  # noqa: F401
