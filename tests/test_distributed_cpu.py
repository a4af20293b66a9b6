"""Tensor-parallel column/row sharding of the W4A16 linear, and the expert-parallel split of the
mixture-of-experts block, with world_size 2 and 4 on the gloo backend (at 4 ranks with 1376-column
shards, the width an 8-way split of the 11008-wide benchmark layer gives each rank: 86 column tiles,
not a multiple of any launch shape's tile count).

Each rank builds the reference-style layers on top of AWQLinearMethod (native ops replaced by
the oracle-backed fakes: there is no GPU here), loads its shard of one full checkpoint through
the layers' weight loaders, and the results are compared with the unsharded oracle GEMM:
column-parallel + all-gather must reproduce it, row-parallel partial products + all-reduce too.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q, N=192):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, HERE)
        from _fakes import FakeOps
        from oracle import awq_oracle as orc
        from vllm_quick_amd.layers import linear as L
        from vllm_quick_amd.layers.quantization import AWQConfig
        from vllm_quick_amd.layers.quantization import awq as awq_mod
        from vllm_quick_amd.parallel_utils import communication_op as comm
        from vllm_quick_amd.parallel_utils import parallel_state as ps

        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        ps.initialize_model_parallel(world)
        assert ps.get_tensor_model_parallel_world_size() == world
        assert ps.get_tensor_model_parallel_rank() == rank
        awq_mod.ops = FakeOps()
        method = AWQConfig(4, 32, True).get_linear_method()

        # collectives
        t = torch.full((3, 4), float(rank + 1))
        r = comm.tensor_model_parallel_all_reduce(t.clone())
        assert torch.equal(r, torch.full((3, 4), float(sum(range(1, world + 1)))))
        g = comm.tensor_model_parallel_all_gather(torch.full((2, 3), float(rank)), dim=-1)
        assert g.shape == (2, 3 * world)
        assert all(torch.equal(g[:, 3 * i:3 * i + 3], torch.full((2, 3), float(i))) for i in range(world))
        g0 = comm.tensor_model_parallel_all_gather(torch.full((2, 3), float(rank)), dim=0)
        assert g0.shape == (2 * world, 3) and torch.equal(g0[2 * rank], torch.full((3, ), float(rank)))

        # inside with_rccl_for_all_reduce() (graph capture) the all-reduce goes to the direct binding
        # (parallel_state.py:217-245, communication_op.py:35-37); a stand-in records the routing here
        seen = []
        real = comm.rccl_utils.all_reduce
        comm.rccl_utils.all_reduce = lambda t: (seen.append(tuple(t.shape)), dist.all_reduce(t))[1]
        assert not ps.is_rccl_enabled_for_all_reduce()
        with ps.with_cupy_nccl_for_all_reduce():
            assert ps.is_cupy_nccl_enabled_for_all_reduce()
            r = comm.tensor_model_parallel_all_reduce(torch.full((2, 2), float(rank + 1)))
        comm.rccl_utils.all_reduce = real
        assert seen == [(2, 2)] and torch.equal(r, torch.full((2, 2), float(sum(range(1, world + 1)))))
        assert not ps.is_rccl_enabled_for_all_reduce()
        comm.tensor_model_parallel_all_reduce(torch.ones(2))
        assert seen == [(2, 2)]

        M, K = 5, 128
        x, qw, s, qz = orc.make_inputs(M, K, N, 32, seed=41)
        want = orc.gemm_faithful(x, qw, s, qz).astype(np.float32)
        ck = {"qweight": qw, "scales": s, "qzeros": qz}

        col = L.ColumnParallelLinear(K, N, bias=False, gather_output=True, params_dtype=torch.float16,
                                     linear_method=method)
        for n, a in ck.items():
            p = getattr(col, n)
            p.weight_loader(p, torch.from_numpy(a))
        assert col.qweight.shape == (K, N // 8 // world)
        y, _ = col(torch.from_numpy(x))
        assert y.shape == (M, N)
        assert np.allclose(y.numpy().astype(np.float32), want, rtol=1e-2, atol=1e-3)

        row = L.RowParallelLinear(K, N, bias=True, input_is_parallel=False, params_dtype=torch.float16,
                                  linear_method=method)
        for n, a in ck.items():
            p = getattr(row, n)
            p.weight_loader(p, torch.from_numpy(a))
        assert row.qweight.shape == (K // world, N // 8) and row.scales.shape == (K // 32 // world, N)
        bias = torch.linspace(-1, 1, N).half()
        row.bias.data.copy_(bias)
        y, ob = row(torch.from_numpy(x))
        assert ob is None
        assert np.allclose(y.detach().numpy().astype(np.float32), want + bias.float().numpy(), rtol=1e-2, atol=2e-3)

        # mixture-of-experts block: the experts are split over the ranks (mixtral_quant.py:113-118),
        # every rank runs its own on all tokens, one all-reduce sums the partial blocks
        from vllm_quick_amd.layers import experts as X
        from test_experts_cpu import fill, reference_block
        X.ops = awq_mod.ops
        moe = X.MixtralMoE(4, 2, 64, 96, linear_method=method, params_dtype=torch.float16)
        per = 4 // world
        assert moe.expert_indicies == list(range(per * rank, per * rank + per))
        assert [e is not None for e in moe.experts] == [i // per == rank for i in range(4)]
        ck_local, gate = fill(moe, seed=3)
        gate_t = torch.from_numpy(gate)
        dist.broadcast(gate_t, src=0)  # the router is replicated: every rank must hold rank 0's
        with torch.no_grad():
            moe.gate.weight.copy_(gate_t)
        xs = (np.random.default_rng(5).standard_normal((1, 6, 64)) * 0.5).astype(np.float16)
        y = moe(torch.from_numpy(xs))
        # every rank's checkpoint entries, gathered, give the whole block's reference
        gathered = [None] * world
        dist.all_gather_object(gathered, ck_local)
        ck_all = {k: v for part in gathered for k, v in part.items()}
        want_moe = reference_block(xs.reshape(6, 64), gate_t.numpy(), ck_all, [0, 1, 2, 3], 2)
        assert np.allclose(y.reshape(6, 64).numpy().astype(np.float32), want_moe.astype(np.float32), rtol=1e-2,
                           atol=3e-3)

        ps.destroy_model_parallel()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
        raise


@pytest.mark.parametrize("world,N", [(2, 192), (4, 4 * 1376)])
def test_column_and_row_parallel_gloo(world, N):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, N)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status in results:
        assert status == "ok", f"rank {rank}:\n{status}"
