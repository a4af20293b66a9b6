"""The oracle against the committed fixtures, its own bit-level emulation and the C restatement.

The fixtures under tests/golden/awq_kernel_cases.npz were produced by oracle/gen_golden.py;
the reference holds no AWQ vectors of its own (parity unpinned), so what these tests pin is
that the oracle is stable, self-consistent across two independent implementations (numpy and
C) and matches a bit-level emulation of the reference's PTX conversion sequence.
"""
import os

import numpy as np
import pytest

from oracle import awq_oracle as orc
from oracle import c_oracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "awq_kernel_cases.npz")


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


def case_names(golden):
    return sorted({k.split("/")[0] for k in golden.files if not k.startswith("ka/")})


def bits(a):
    return np.ascontiguousarray(a).view(np.uint16)


def test_known_answer_nibble_order(golden):
    # word 0x76543210 -> columns [0,4,1,5,2,6,3,7] (dequantize.cuh:40-56)
    assert (orc.unpack_words(golden["ka/word"]) == golden["ka/word_columns"]).all()
    assert (orc.s4_to_fp16x2_bits(golden["ka/word"]).reshape(1, 8) == golden["ka/word_columns"]).all()


def test_pack_unpack_roundtrip():
    rng = np.random.default_rng(0)
    q = rng.integers(0, 16, size=(33, 64), dtype=np.uint8)
    assert (orc.unpack_words(orc.pack_words(q)) == q).all()
    w = rng.integers(0, 2**32, size=(5, 7), dtype=np.uint64).astype(np.uint32).view(np.int32)
    assert (orc.pack_words(orc.unpack_words(w)) == w).all()
    with pytest.raises(ValueError):
        orc.pack_words(np.zeros((2, 7), dtype=np.uint8))
    with pytest.raises(ValueError):
        orc.pack_words(np.full((2, 8), 16, dtype=np.uint8))


def test_bit_level_converter_covers_every_nibble_value():
    # all 16 values in all 8 positions
    words = np.array([sum(((v + p) % 16) << (4 * p) for p in range(8)) for v in range(16)], dtype=np.uint32)
    got = orc.s4_to_fp16x2_bits(words)
    want = orc.unpack_words(words.view(np.int32).reshape(16, 1)).astype(np.float16)
    assert (got == want).all()


def test_unit_weights_identity_rows(golden):
    # scales 1, zeros 0, x = e_k: the GEMM returns row k of the integer weights exactly
    y = orc.gemm_faithful(golden["ka/unit_x"], golden["ka/unit_qweight"], golden["ka/unit_scales"],
                          golden["ka/unit_qzeros"])
    assert (y == golden["ka/unit_out"]).all()


def test_dequantize_matches_fixture_bit_exact(golden):
    for name in case_names(golden):
        if f"{name}/dequant" not in golden.files:
            continue
        w = orc.dequantize(golden[f"{name}/qweight"], golden[f"{name}/scales"], golden[f"{name}/qzeros"])
        assert (bits(w) == bits(golden[f"{name}/dequant"])).all(), name
        w2 = orc.dequantize_via_bits(golden[f"{name}/qweight"], golden[f"{name}/scales"], golden[f"{name}/qzeros"])
        assert (bits(w2) == bits(w)).all(), name


def test_gemm_matches_fixture(golden):
    for name in case_names(golden):
        args = [golden[f"{name}/{k}"] for k in ("x", "qweight", "scales", "qzeros")]
        y = orc.gemm_faithful(*args).astype(np.float32)
        want = golden[f"{name}/faithful"].astype(np.float32)
        # fp32 summation order inside a slice depends on the BLAS build: allow one fp16 ulp
        assert np.allclose(y, want, rtol=2**-10, atol=2**-14), name
        e = orc.gemm_exact(*args)
        assert np.allclose(e, golden[f"{name}/exact"], rtol=1e-6, atol=1e-6), name
        # and the reference-faithful result sits within the stated parity tolerance of the exact one
        assert np.allclose(want, golden[f"{name}/exact"], rtol=1e-2, atol=1e-3), name


def test_c_restatement_agrees_with_numpy(golden):
    for name in case_names(golden):
        x, qw, s, qz = [golden[f"{name}/{k}"] for k in ("x", "qweight", "scales", "qzeros")]
        if qw.shape[0] > 512:
            continue
        assert (bits(c_oracle.dequantize(qw, s, qz)) == bits(orc.dequantize(qw, s, qz))).all(), name
        yc = c_oracle.gemm(x, qw, s, qz).astype(np.float32)
        yn = orc.gemm_faithful(x, qw, s, qz).astype(np.float32)
        assert np.allclose(yc, yn, rtol=2**-10, atol=2**-14), name
        y1 = c_oracle.gemm(x, qw, s, qz, faithful=False).astype(np.float64)
        assert np.allclose(y1, orc.gemm_exact(x, qw, s, qz), rtol=2e-3, atol=1e-3), name


def test_c_fp16_conversions_match_numpy():
    lib = c_oracle.lib()
    hs = np.arange(0, 65536, 13, dtype=np.uint32).astype(np.uint16)
    for h in hs:
        ref = np.array([h], dtype=np.uint16).view(np.float16)[0]
        got = lib.awq_c_h2f(int(h))
        assert (np.isnan(ref) and np.isnan(got)) or np.float32(ref) == np.float32(got)
    rng = np.random.default_rng(1)
    fs = np.concatenate([rng.standard_normal(3000).astype(np.float32) * s for s in (1e-8, 1e-5, 1e-3, 1.0, 300.0)])
    fs = np.concatenate([fs, np.array([65504, 65519.9, 65520, 6.1e-5, 5.96e-8, 2.98e-8, 0.0, -0.0], np.float32)])
    with np.errstate(over="ignore"):
        want = fs.astype(np.float16).view(np.uint16)
    got = np.array([lib.awq_c_f2h(float(f)) for f in fs], dtype=np.uint16)
    assert (got == want).all()


def test_shape_checks():
    x, qw, s, qz = orc.make_inputs(2, 64, 40, 32, seed=5)
    orc.gemm_faithful(x, qw, s, qz)  # N % 64 != 0 is fine unless the reference's checks are requested
    with pytest.raises(ValueError, match="cta_N"):
        orc.gemm_faithful(x, qw, s, qz, strict_reference=True)
    with pytest.raises(ValueError):
        orc.gemm_faithful(x[:, :32], qw, s, qz)
    with pytest.raises(ValueError):
        orc.dequantize(qw, s[:, :32], qz)
