"""`bench.py --gpus N` rehearsed with two ranks on one GPU (gloo carries the all-reduce): the N > 1 code path of the
benchmark - sharding, the timed step with its collective, max-over-ranks timing, the JSON line with the tensor-parallel
fields - runs end to end before an 8-GPU node ever sees it.  RCCL itself refuses two ranks on one device, so the
collective here is torch.distributed's gloo on a host copy; the figures mean nothing, the fields and their
consistency do.  Also: one rank's step of an 8-way tensor-parallel Llama-2-70B-shaped model, run alone
(`decode_tp8_rank_compute` of the N = 1 bench line)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_rehearsal_prints_the_tensor_parallel_fields(cuda):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8",
           "--warmup", "2", "--backend", "gloo", "--no-p2p", "--no-decode"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 8 and out["warmup"] == 2
    assert out["metric"].startswith("W4A16 GEMM TFLOPS") and out["unit"] == "TFLOP/s" and out["scaling"] == "strong"
    assert out["config"]["parallelism"] == "tp2" and out["vs_baseline"] is None and out["dtype"] == "f16"
    # whole-job value: both GEMMs of the step over the max-over-ranks step time
    flops = 2 * (2.0 * out["config"]["M"] * out["config"]["K"] * out["config"]["N"])
    assert out["value"] == pytest.approx(flops / (out["ms_per_step"] * 1e-3) * 1e-12, rel=1e-6)
    tp = out["tp"]
    assert tp["all_reduce_us"] > 0 and tp["all_reduce_bytes"] == 2 * 128 * 11008
    assert "torch.distributed" in tp["collective"]  # (gloo: never the direct RCCL communicator)
    assert tp["column_shard"] == [4096, 5504] and tp["row_shard"] == [2048, 11008]
    for key, shard in (("column_roofline", (4096, 5504)), ("row_roofline", (2048, 11008))):
        rf = tp[key]
        assert rf["bound"] == "mfma" and rf["peak"] == 2500.0 and rf["launch_us"] > 0
        assert rf["achieved"] == pytest.approx(2.0 * 128 * shard[0] * shard[1] / rf["launch_us"] * 1e-6, rel=1e-6)
        assert rf["kernel"].startswith("gemm_")
    rl = out["roofline"]
    assert rl["bound"] == "mfma" and 0 < rl["frac"] < 1 and "column shard" in rl["kernel"]
    assert rl["traffic"] is None  # counter profiles are taken on one GPU's whole-layer launch only


def test_one_rank_of_a_tensor_parallel_model_runs_alone(cuda):
    """Layers built inside parallel_state.rehearse_one_rank(8) shard as rank 0 of 8 (Llama-2-70B: qkv 8192 x 1280,
    o 1024 x 8192, gate_up 8192 x 7168, down 3584 x 8192) and the all-reduce is a device copy."""
    sys.path.insert(0, ROOT)
    from benchmarks.llama_decode import run
    from vllm_quick_amd.parallel_utils import parallel_state as ps
    out = run("70b", (1, 8), n_layers=2, device=cuda, keep_stock=False, rehearse_tp=8)
    assert out["tp"] == 8 and out["ranks_run"] == 1 and "device copy" in out["all_reduce"]
    assert ps.get_tensor_model_parallel_world_size() == 1  # the rehearsal ended with the block
    for bs in ("1", "8"):
        r = out["by_batch"][bs]
        assert r["tokens_per_s"] > 0 and 0 < r["w4a16_share"] < 1
    # a rank holds an eighth of the model's quantised weights (here: of 2 of its 80 layers, extrapolated)
    per_layer = (8192 * 1280 + 1024 * 8192 + 8192 * 7168 + 3584 * 8192) // 2
    assert out["quantised_weight_bytes_resident"] == pytest.approx(80 * per_layer * 1.04, rel=0.03)
    assert out["quantised_weight_bytes_with_stock_tensors"] > 1.9 * out["quantised_weight_bytes_resident"]
    x = torch.ones(4, 8, device=cuda)
    with ps.rehearse_one_rank(8):
        from vllm_quick_amd.parallel_utils.communication_op import tensor_model_parallel_all_reduce
        y = tensor_model_parallel_all_reduce(x)
        assert ps.get_tensor_model_parallel_world_size() == 8 and ps.get_tensor_model_parallel_rank() == 0
    assert torch.equal(y, x) and y.data_ptr() != x.data_ptr()
