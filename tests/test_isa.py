"""ISA-level guard (CPU: hipcc cross-compiles): the attention tile loops must not contain a vmcnt wait that the compiler added.

Round 3 found `s_waitcnt vmcnt(0)` in front of the first QK^T MFMA of every iteration of the shipped attention kernels (the Q^T
fragments are loaded before the loop, first used inside it, and the hand-counted waits are invisible to the compiler), which
drained the K / V^T prefetch each time: +3.6 % at batch 32, +2.7 % at batch 1 once removed (profiles/r03/attention_q_pin_ab.txt)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.scan_isa_waits import compile_to_asm, scan  # noqa: E402


@pytest.mark.parametrize("f16", [1])
def test_attention_loops_have_no_compiler_vmcnt_wait(f16):
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    asm = compile_to_asm(os.path.join(ROOT, "f5_tts_mlx_amd", "csrc", "attention.hip"), [f"-DF5_F16={f16}"])
    found = {k: v for k, v in scan(asm).items() if "f5_attn" in k}
    assert "v_mfma_f32_32x32x16_f16" in asm and "f5_attn2f_kernel" in asm and "f5_attn2s_kernel" in asm
    assert not found, found
