"""The direct RCCL binding (counterpart of the reference's CuPy-NCCL shim, cupy_utils.py): a
one-rank communicator on the one GPU there is, eager and captured in a HIP graph.  One rank cannot
show a sum over peers, it shows that the library is found, the C signatures are right and the call
is capturable, which is what the shim exists for."""
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_one_rank_communicator_eager_and_captured():
    from vllm_quick_amd.parallel_utils import rccl_utils as R
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    assert not R.is_initialized() and R.get_world_size() == 0
    R.init_process_group(1, 0, "127.0.0.1", _free_port())
    try:
        assert R.is_initialized() and R.get_world_size() == 1 and R.get_rank() == 0
        for dt in (torch.float16, torch.bfloat16, torch.float32):
            x = torch.arange(4096, device=dev).to(dt)
            want = x.clone()
            R.all_reduce(x)
            R.all_reduce(x, op=torch.distributed.ReduceOp.MAX)
            torch.cuda.synchronize()
            assert torch.equal(x, want)
        with pytest.raises(ValueError):
            R.all_reduce(torch.zeros(8, 8, device=dev).t())
        # captured next to other work, replayed; on a side stream that is also handed over explicitly
        x = torch.ones(128, 11008, dtype=torch.float16, device=dev)
        stream = torch.cuda.Stream(device=dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(stream):
            with torch.cuda.graph(g, stream=stream):
                for _ in range(4):
                    x.mul_(2)
                    R.all_reduce(x)
                with R.set_cupy_stream(stream):
                    R.all_reduce(x)
        torch.cuda.synchronize()
        assert torch.equal(x, torch.ones_like(x)), "capture must not run anything"
        g.replay()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(x, torch.full_like(x, 256.0))
    finally:
        R.destroy_process_group()
    assert not R.is_initialized()
