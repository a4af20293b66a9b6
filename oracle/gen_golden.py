"""Generates the committed fixtures under tests/golden/.  Run in the BUILD container only
(it imports the reference's Python from /root/reference; nothing under tests/ does).

    python oracle/gen_golden.py

Two families of fixtures:

1. awq_kernel_cases.npz - seeded inputs with the oracle's outputs (dequantised
   weights, reference-faithful GEMM, float64 GEMM) and known-answer vectors.
   PARITY UNPINNED: the reference has no AWQ test or golden vector and its CUDA
   kernel cannot be built here, so these pin the implementation to the oracle's
   restatement of the reference arithmetic, not to reference output.

2. awq_reference_python.npz / .json - behaviour captured from the REFERENCE's own
   Python (`AWQConfig`, `AWQLinearMethod`, the five linear layers' weight
   loaders), imported from /root/reference with the native module `vllm._C`
   replaced by recording fakes that compute with the oracle.  These pin the
   host-side mirror: parameter shapes/dtypes/attributes, error conditions, the
   operator call trace of `apply_weights`, bias/reshape semantics and the
   tensor-parallel shard arithmetic.
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import awq_oracle as orc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"

# (name, M, K, N, G, seed)
KERNEL_CASES = [
    ("k256_n128_g128_m1", 1, 256, 128, 128, 11),
    ("k256_n192_g64_m3", 3, 256, 192, 64, 12),  # N % 128 != 0 (the reference's <64> instantiation)
    ("k512_n256_g128_m17", 17, 512, 256, 128, 13),  # M not a multiple of 16
    ("k96_n40_g32_m5", 5, 96, 40, 32, 14),  # K % 128 != 0, N % 16 != 0
    ("k4096_n256_g128_m1", 1, 4096, 256, 128, 15),
    ("k4096_n256_g128_m16", 16, 4096, 256, 128, 15),
]


def gen_kernel_cases():
    blob = {}
    for name, M, K, N, G, seed in KERNEL_CASES:
        x, qw, s, qz = orc.make_inputs(M, K, N, G, seed=seed)
        blob[f"{name}/x"] = x
        blob[f"{name}/qweight"] = qw
        blob[f"{name}/scales"] = s
        blob[f"{name}/qzeros"] = qz
        blob[f"{name}/faithful"] = orc.gemm_faithful(x, qw, s, qz)
        blob[f"{name}/exact"] = orc.gemm_exact(x, qw, s, qz).astype(np.float32)
        if K <= 512:
            blob[f"{name}/dequant"] = orc.dequantize(qw, s, qz)
    # known-answer vectors
    blob["ka/word"] = np.array([[0x76543210]], dtype=np.uint32).view(np.int32)
    blob["ka/word_columns"] = np.array([[0, 4, 1, 5, 2, 6, 3, 7]], dtype=np.uint8)
    # unit scales, zero zeros, x = e_k picks out row k of the weights
    K, N, G = 64, 64, 32
    rng = np.random.default_rng(99)
    q = rng.integers(0, 16, size=(K, N), dtype=np.uint8)
    blob["ka/unit_q"] = q
    blob["ka/unit_qweight"] = orc.pack_words(q)
    blob["ka/unit_scales"] = np.ones((K // G, N), dtype=np.float16)
    blob["ka/unit_qzeros"] = np.zeros((K // G, N // 8), dtype=np.int32)
    blob["ka/unit_x"] = np.eye(K, dtype=np.float16)[:16]
    blob["ka/unit_out"] = q[:16].astype(np.float16)
    np.savez_compressed(os.path.join(OUT, "awq_kernel_cases.npz"), **blob)
    return len(blob)


# --------------------------------------------------------------------------
# the reference's Python, imported with the native module stubbed
# --------------------------------------------------------------------------
def import_reference():
    pkg = types.ModuleType("vllm")
    pkg.__path__ = [os.path.join(REF, "vllm")]
    sys.modules["vllm"] = pkg
    stub = types.ModuleType("vllm._C")
    stub.ops = types.SimpleNamespace()
    sys.modules["vllm._C"] = stub
    import vllm.model_executor.layers.linear as ref_linear
    import vllm.model_executor.layers.quantization as ref_quant
    import vllm.model_executor.layers.quantization.awq as ref_awq
    return ref_awq, ref_linear, ref_quant, stub.ops


def attrs_of(p):
    return {k: getattr(p, k) for k in ("input_dim", "output_dim", "packed_dim", "pack_factor") if hasattr(p, k)}


def gen_reference_python():
    ref_awq, ref_linear, ref_quant, ops = import_reference()
    meta = {}
    arrays = {}

    # --- AWQConfig --------------------------------------------------------
    cfg = ref_awq.AWQConfig.from_config({"w_bit": 4, "q_group_size": 128, "zero_point": True})
    cfg2 = ref_awq.AWQConfig.from_config({"bits": 4, "group_size": 64, "zero_point": False})
    meta["config"] = {
        "repr": repr(cfg),
        "repr2": repr(cfg2),
        "name": cfg.get_name(),
        "pack_factor": cfg.pack_factor,
        "act_dtypes": [str(d) for d in cfg.get_supported_act_dtypes()],
        "min_capability": cfg.get_min_capability(),
        "filenames": ref_awq.AWQConfig.get_config_filenames(),
        "scaled_act_names": cfg.get_scaled_act_names(),
        "registry_awq_is_AWQConfig": ref_quant.get_quantization_config("awq") is ref_awq.AWQConfig,
    }
    errs = {}
    for label, fn in {
            "bits8": lambda: ref_awq.AWQConfig(8, 128, True),
            "missing_key": lambda: ref_awq.AWQConfig.from_config({"bits": 4, "zero_point": True}),
            "bad_registry": lambda: ref_quant.get_quantization_config("nope"),
    }.items():
        try:
            fn()
            errs[label] = None
        except Exception as e:  # noqa: BLE001
            errs[label] = [type(e).__name__, str(e)]
    meta["config_errors"] = errs

    # --- create_weights ---------------------------------------------------
    method = cfg.get_linear_method()
    w = method.create_weights(4096, 11008, 4096, 11008, torch.float16)
    meta["create_weights_4096x11008"] = {
        k: {
            "shape": list(v.shape),
            "dtype": str(v.dtype),
            "requires_grad": v.requires_grad,
            "attrs": attrs_of(v)
        }
        for k, v in w.items()
    }
    cw_err = {}
    for label, args in {"n_unaligned": (4096, 11004, 4096, 11004), "k_unaligned": (4000, 11008, 4000, 11008)}.items():
        try:
            method.create_weights(*args, torch.float16)
            cw_err[label] = None
        except Exception as e:  # noqa: BLE001
            cw_err[label] = [type(e).__name__, str(e)]
    meta["create_weights_errors"] = cw_err

    # --- apply_weights: call trace and outputs ----------------------------
    K, N, G = 256, 128, 128
    _, qw, s, qz = orc.make_inputs(1, K, N, G, seed=21)
    weights = {"qweight": torch.from_numpy(qw), "scales": torch.from_numpy(s), "qzeros": torch.from_numpy(qz)}
    arrays["apply/qweight"], arrays["apply/scales"], arrays["apply/qzeros"] = qw, s, qz
    trace = []

    def fake_gemm(x, qweight, scales, qzeros, split_k):
        trace.append(["awq_gemm", list(x.shape), int(split_k), bool(x.is_contiguous())])
        return torch.from_numpy(orc.gemm_faithful(x.numpy(), qweight.numpy(), scales.numpy(), qzeros.numpy(), split_k))

    def fake_dequant(qweight, scales, qzeros, split_k, thx, thy):
        trace.append(["awq_dequantize", list(qweight.shape), int(split_k), int(thx), int(thy)])
        return torch.from_numpy(orc.dequantize(qweight.numpy(), scales.numpy(), qzeros.numpy()))

    ref_awq.ops.awq_gemm = fake_gemm
    ref_awq.ops.awq_dequantize = fake_dequant
    method128 = ref_awq.AWQConfig(4, G, True).get_linear_method()
    rng = np.random.default_rng(22)
    bias = (rng.standard_normal(N) * 0.1).astype(np.float16)
    arrays["apply/bias"] = bias
    calls = {}
    for label, shape, use_bias in [("m1", (1, K), False), ("m255", (255, K), False), ("m256", (256, K), False),
                                   ("b2x3_bias", (2, 3, K), True), ("m7_bias", (7, K), True)]:
        x = (rng.standard_normal(shape) * 0.25).astype(np.float16)
        del trace[:]
        y = method128.apply_weights(weights, torch.from_numpy(x), torch.from_numpy(bias) if use_bias else None)
        calls[label] = {"x_shape": list(shape), "bias": use_bias, "trace": [list(t) for t in trace],
                        "out_shape": list(y.shape), "out_dtype": str(y.dtype)}
        arrays[f"apply/{label}/x"] = x
        arrays[f"apply/{label}/out"] = y.numpy()
    meta["apply_weights"] = calls

    # --- tensor-parallel weight loaders -----------------------------------
    # The reference layers ask parallel_state for (world, rank); patch the names
    # they imported to emulate rank r of p without a process group.
    state = {"world": 1, "rank": 0}
    ref_linear.get_tensor_model_parallel_world_size = lambda: state["world"]
    ref_linear.get_tensor_model_parallel_rank = lambda: state["rank"]
    g32 = ref_awq.AWQConfig(4, 32, True).get_linear_method()

    def ckpt(K, N, seed):
        _, qw, s, qz = orc.make_inputs(1, K, N, 32, seed=seed)
        return {"qweight": qw, "scales": s, "qzeros": qz}

    loaders = {}

    def record(label, layer, world, rank, feed):
        """feed: list of (param_name, numpy tensor, shard_id or None)."""
        for name, arr, shard in feed:
            p = getattr(layer, name)
            if shard is None:
                p.weight_loader(p, torch.from_numpy(arr))
            else:
                p.weight_loader(p, torch.from_numpy(arr), shard)
        for name in ("qweight", "scales", "qzeros"):
            arrays[f"loader/{label}/tp{world}r{rank}/{name}"] = getattr(layer, name).data.numpy().copy()

    for world in (1, 2, 4):
        for rank in range(world):
            state["world"], state["rank"] = world, rank
            # column parallel 64 -> 128
            ck = ckpt(64, 128, 31)
            layer = ref_linear.ColumnParallelLinear(64, 128, bias=False, params_dtype=torch.float16, linear_method=g32)
            record("column", layer, world, rank, [(n, ck[n], None) for n in ck])
            # row parallel 128 -> 64
            ck = ckpt(128, 64, 32)
            layer = ref_linear.RowParallelLinear(128, 64, bias=False, params_dtype=torch.float16, linear_method=g32)
            record("row", layer, world, rank, [(n, ck[n], None) for n in ck])
            # merged column 64 -> [64, 128]: per-shard checkpoints and one fused checkpoint
            a, b = ckpt(64, 64, 33), ckpt(64, 128, 34)
            layer = ref_linear.MergedColumnParallelLinear(64, [64, 128], bias=False, params_dtype=torch.float16,
                                                          linear_method=g32)
            record("merged_split", layer, world, rank, [(n, a[n], 0) for n in a] + [(n, b[n], 1) for n in b])
            fused = {n: np.concatenate([a[n], b[n]], axis=1) for n in a}
            layer = ref_linear.MergedColumnParallelLinear(64, [64, 128], bias=False, params_dtype=torch.float16,
                                                          linear_method=g32)
            record("merged_fused", layer, world, rank, [(n, fused[n], None) for n in fused])
            # qkv: hidden 64, head 16, 4 q heads, 2 kv heads (replicated at tp=4)
            q, k, v = ckpt(64, 64, 35), ckpt(64, 32, 36), ckpt(64, 32, 37)
            layer = ref_linear.QKVParallelLinear(64, 16, 4, 2, bias=False, params_dtype=torch.float16,
                                                 linear_method=g32)
            record("qkv_split", layer, world, rank,
                   [(n, q[n], "q") for n in q] + [(n, k[n], "k") for n in k] + [(n, v[n], "v") for n in v])
            fused = {n: np.concatenate([q[n], k[n], v[n]], axis=1) for n in q}
            layer = ref_linear.QKVParallelLinear(64, 16, 4, 2, bias=False, params_dtype=torch.float16,
                                                 linear_method=g32)
            record("qkv_fused", layer, world, rank, [(n, fused[n], None) for n in fused])
            loaders[f"tp{world}r{rank}"] = {
                "qkv_num_heads": layer.num_heads,
                "qkv_num_kv_heads": layer.num_kv_heads,
                "qkv_num_kv_head_replicas": layer.num_kv_head_replicas,
                "qkv_output_size": layer.output_size,
            }
    meta["loaders"] = loaders
    meta["loader_checkpoint_seeds"] = {"column": 31, "row": 32, "merged": [33, 34], "qkv": [35, 36, 37]}

    with open(os.path.join(OUT, "awq_reference_python.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(OUT, "awq_reference_python.npz"), **arrays)
    return len(arrays)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    n1 = gen_kernel_cases()
    n2 = gen_reference_python()
    print(f"wrote {n1} kernel arrays and {n2} reference-python arrays to {OUT}")
