"""Per-kernel parity of the fp16-operand build of the kernels (engine precision "f16", namespace f5hf in csrc/op16.hpp).

The op tests of test_ops_gpu.py are written against "the operand type": here a fixture switches both the test helpers
(f5test.operand_mode: buffers and reference rounding become torch.float16) and the library's f5_op_* entry points to fp16
and re-runs a representative subset of them -- every MFMA kernel family (small-tile, ring, 8-wave ring, 256x256 GEMMs with
each fused epilogue; the attention kernels incl. ragged masks and the online-softmax rescale path; the conv position
embedding) and the 16-bit producers (LN-modulate, depthwise conv + LN, GRN).
"""
import numpy as np
import pytest
import torch

import test_ops_gpu as T
from f5test import DEV, E, P, bf16r, op_dtype, operand_mode, randn, report, rng, split_bf16, stream

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    return E.load_library()


@pytest.fixture(autouse=True)
def f16_operands():
    with operand_mode("f16"):
        assert op_dtype() == torch.float16
        yield
    assert op_dtype() == torch.bfloat16


@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (937, 100, 256), (1874, 1024, 1024)])
def test_gemm_f32_out_f16(lib, M, N, K):
    T.test_gemm_f32_out(lib, M, N, K)


def test_gemm_asymmetric_identity_f16(lib):
    T.test_gemm_asymmetric_identity(lib)


@pytest.mark.parametrize("epi", [1, 2, 3])
def test_gemm_epilogues_f16(lib, epi):
    T.test_gemm_epilogues(lib, epi)


@pytest.mark.parametrize("N", [1, 63, 65, 499, 937])
def test_attention_shapes_f16(lib, N):
    T._attention_case(lib, 1, 2, N, None, 1, seed=N)


def test_attention_ragged_mask_f16(lib):
    T._attention_case(lib, 3, 2, 200, [200, 130, 1], 1, seed=7)


def test_attention_large_grid_f16(lib):
    """>= 1024 workgroups: the 256-query kernel with two query blocks per wave"""
    T._attention_case(lib, 8, 16, 937, None, 1, seed=11)


def test_attention_softmax_spike_f16(lib):
    T.test_attention_softmax_spike(lib)
    T.test_attention_softmax_spike(lib, hp=0)


@pytest.mark.parametrize("N", [1, 33, 64, 65, 130, 257, 320, 499, 937])
def test_attention_pipelined_v2p_shapes_f16(lib, N):
    T.test_attention_pipelined_v2p_shapes(lib, N)


def test_attention_pipelined_v2p_ragged_and_spikes_f16(lib):
    T.test_attention_pipelined_v2p_ragged_and_spikes(lib)


def test_attention_pipelined_v2p_large_grid_f16(lib):
    T.test_attention_pipelined_v2p_large_grid_matches_v2f(lib)


@pytest.mark.lab
@pytest.mark.parametrize("N", [1, 64, 130, 937])
def test_attention_pipelined_kernel_f16(lib, N):
    T.test_attention_pipelined_kernel_shapes(lib, 5, N)


@pytest.mark.lab
def test_attention_pipelined_kernel_ragged_f16(lib):
    T.test_attention_pipelined_kernel_ragged_and_batched(lib, 5)


@pytest.mark.parametrize("tile", [4, 13, 14])
def test_transposed_qkv_tiles_race_screen_f16(lib, tile):
    T.test_transposed_qkv_tiles_race_screen(lib, tile)


@pytest.mark.parametrize("tps", [0, 1, 2])
@pytest.mark.parametrize("B,N,C", [(2, 130, 256), (1, 937, 1024)])
def test_convpos_f16(lib, B, N, C, tps):
    T.test_convpos(lib, B, N, C, 31, 1, tps)


@pytest.mark.parametrize("rows,dim", [(5, 512), (937, 1024)])
def test_ln_modulate_f16(lib, rows, dim):
    T.test_ln_modulate(lib, rows, dim)


def test_dwconv_ln_f16(lib):
    T.test_dwconv_ln(lib, 2, 100, 512)


@pytest.mark.parametrize("tile", [0, 2, 10])
def test_gemm_resid_gate_fused_ln_f16(lib, tile):
    T.test_gemm_resid_gate_fused_ln_is_bit_identical(lib, tile, 1)


@pytest.mark.parametrize("tile", [0, 4, 8, 10])
def test_gemm_resid_gate_f16(lib, tile):
    T.test_gemm_resid_gate(lib, tile, 1)


@pytest.mark.parametrize("tile", [1, 2, 5])
def test_gemm_small_tile_kernels_f16(lib, tile):
    T.test_gemm_all_small_tile_kernels(lib, tile)


def test_fp16_producers_saturate(lib):
    """|x| > 65504 must come out as +-65504, not inf (csrc/op16.hpp f5_sat): LN-modulate with a huge scale."""
    rows, dim = 4, 256
    r = rng(5)
    x = randn(r, rows, dim)
    sc = torch.full((dim,), 1.0e6)
    sh = torch.zeros(dim)
    hi = torch.zeros((rows, dim), dtype=torch.float16, device=DEV)
    xd, scd, shd = x.to(DEV), sc.to(DEV), sh.to(DEV)
    E.check(lib.f5_op_ln_modulate(P(xd), P(scd), P(shd), P(hi), P(None), rows, dim, stream()))
    torch.cuda.synchronize()
    assert torch.isfinite(hi).all() and float(hi.abs().max()) == 65504.0


@pytest.mark.parametrize("tile", [0, 14])
def test_gemm_rs128_several_rounds_all_epilogues_f16(lib, tile):
    T.test_gemm_rs128_several_rounds_all_epilogues(lib, tile)


@pytest.mark.parametrize("tile", [4, 14])
@pytest.mark.parametrize("stress", range(len(T.FOLD_STRESS)))
def test_ln_modulate_folded_into_the_gemms_around_it_f16(lib, tile, stress):
    T.test_ln_modulate_folded_into_the_gemms_around_it(lib, tile, stress)
