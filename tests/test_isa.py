"""ISA-level guard (CPU: hipcc cross-compiles): the attention tile loops must not contain a vmcnt wait that the compiler added.

Round 3 found `s_waitcnt vmcnt(0)` in front of the first QK^T MFMA of every iteration of the shipped attention kernels (the Q^T
fragments are loaded before the loop, first used inside it, and the hand-counted waits are invisible to the compiler), which
drained the K / V^T prefetch each time: +3.6 % at batch 32, +2.7 % at batch 1 once removed (profiles/r03/attention_q_pin_ab.txt)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.scan_isa_waits import compile_to_asm, scan, scan_pk_hazard  # noqa: E402


@pytest.mark.parametrize("f16", [0, 1])
def test_attention_loops_have_no_compiler_vmcnt_wait(f16):
    """both operand builds (build.sh ships the bf16 and the fp16 flavour of every kernel source)"""
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    asm = compile_to_asm(os.path.join(ROOT, "f5_tts_mlx_amd", "csrc", "attention.hip"), [f"-DF5_F16={f16}"])
    found = {k: v for k, v in scan(asm).items() if "f5_attn" in k}
    assert ("v_mfma_f32_32x32x16_f16" if f16 else "v_mfma_f32_32x32x16_bf16") in asm and "f5_attn2f_kernel" in asm and "f5_attn2s_kernel" in asm
    assert not found, found
    # the in-wave pipelined kernel (v2p) keeps ~400 registers live with the register classes chosen by hand: a spill anywhere puts
    # scratch round trips (and their vmcnt(0) drains) on the tile path -- round 4 measured 10 us per spilled tail tile
    import re
    m = re.search(r"\.amdhsa_kernel (\S*f5_attn2p_kernel\S*)(.*?)\.end_amdhsa_kernel", asm, re.S)
    assert m, "f5_attn2p_kernel not found"
    body = m.group(2)
    assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", body), "f5_attn2p_kernel uses scratch (spills)"
    start = asm.index(m.group(1) + ":")
    code = asm[start:asm.index(".Lfunc_end", start)]
    assert "scratch_" not in code
    # hazard guard of the asm MFMAs: a VALU write of an MFMA operand must be two wait states away -- every asm MFMA opens with s_nop 1
    lines = [l.strip() for l in code.split("\n") if l.strip() and not l.strip().startswith(";")]
    mf = [i for i, l in enumerate(lines) if l.startswith("v_mfma")]
    assert len(mf) > 100 and all(lines[i - 1].startswith("s_nop 1") for i in mf), "an asm MFMA of f5_attn2p_kernel lost its hazard guard"


@pytest.mark.parametrize("name", ["gemm256", "gemm_rs128", "gemm_f8"])
def test_large_gemm_k_loops_have_no_compiler_vmcnt_wait(name):
    """the same scan over the large-shape GEMM kernels, every instantiation -- the bf16x3 passes are the same instantiation
    (nseg is a run-time loop bound), the MX-fp8 kernel (gemm_f8.hip) its own: the K loops count their outstanding loads by hand, a wait
    the compiler adds inside one of them drains the operand prefetch every step."""
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    for f16 in (0, 1):
        asm = compile_to_asm(os.path.join(ROOT, "f5_tts_mlx_amd", "csrc", name + ".hip"), [f"-DF5_F16={f16}", "-fno-slp-vectorize"])
        found = {k: v for k, v in scan(asm).items() if "gemm" in k}
        assert "v_mfma" in asm and not found, (name, f16, found)
        # ... and no instantiation spills: the FOLD epilogues (LN fold, round 4) sit on top of 128 live accumulators -- a persistent-walk
        # experiment on these kernels spilled 200-600 bytes and ran 0.2-0.6x (profiles/r04/gemm256_persistent_walk_rejected.jsonl)
        import re
        sizes = re.findall(r"\.amdhsa_kernel (\S+).*?\.amdhsa_private_segment_fixed_size (\d+)", asm, re.S)
        assert sizes and all(int(n) == 0 for k, n in sizes if "gemm" in k), [(k, n) for k, n in sizes if int(n)]


def _noslp_list():
    """the files build.sh compiles with -fno-slp-vectorize, parsed from the script (single source of truth)"""
    import re
    sh = open(os.path.join(ROOT, "f5_tts_mlx_amd", "csrc", "build.sh")).read()
    m = re.search(r'noslp\(\) \{ case "\$1" in ([a-z0-9_|]+)\) echo "-fno-slp-vectorize"', sh)
    assert m, "build.sh: noslp() not found"
    return tuple(m.group(1).split("|"))


NOSLP = _noslp_list()


def _asm(job):
    name, f16 = job
    flags = [f"-DF5_F16={f16}"] + (["-fno-slp-vectorize"] if name in NOSLP else [])
    return name, compile_to_asm(os.path.join(ROOT, "f5_tts_mlx_amd", "csrc", name + ".hip"), flags)


def test_no_packed_f32_with_hi_to_lo_select_in_mfma_kernels():
    """Round-2 hazard (a), root-caused in round 3: v_pk_mul_f32 / v_pk_add_f32 whose LO result reads the HI register of src1
    (op_sel:[x,1]) returns 0 for that operand on lanes 48-63 when another wave of the SIMD has MFMAs in flight
    (tools/probes/pk_f32_vs_mfma2.hip: 0 wrong in 4e10 with idle partners, hundreds with MFMA partners; plain selects never).  The
    SLP vectoriser emits that form for the epilogue arithmetic, so the GEMM files are built with -fno-slp-vectorize: no kernel
    that contains MFMAs may carry the form, in the flags the library is built with."""
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    assert {"gemm", "gemm256", "gemm_rs128", "gemm_f8", "rowops"} <= set(NOSLP)
    from concurrent.futures import ThreadPoolExecutor
    jobs = [(n, f16) for n in ("gemm", "gemm256", "gemm_rs128", "gemm_f8", "attention", "convpos", "rowops") for f16 in (0, 1)]
    from tools.scan_isa_waits import scan_small_load_batches, scan_store_waits
    with ThreadPoolExecutor(max_workers=7) as ex:
        for name, asm in ex.map(_asm, jobs):
            assert "v_mfma" in asm, name
            found = scan_pk_hazard(asm)
            assert not found, (name, found)
            # the serialised epilogues found in round 3 stay fixed (profiles/r03/convpos_epilogue_ab.txt, resid_epilogue_loads_ab.txt,
            # ln_modulate_loads_ab.txt): no store-serialising wait clusters in the conv-pos kernels or in the residual epilogue of
            # the 256x256 kernel, no runs of full drains behind single loads in the residual epilogue or in LN-modulate
            if name == "convpos":
                assert not scan_store_waits(asm), scan_store_waits(asm)
            if name == "gemm256":
                resid = "f5_gemm256_kernelILi4E"
                assert not [k for k in scan_store_waits(asm) if resid in k]
                assert not [k for k in scan_small_load_batches(asm) if resid in k]
            if name == "rowops":
                assert not [k for k in scan_small_load_batches(asm) if "ln_modulate_kernel" in k]


def test_isa_scanners_on_synthetic_listings():
    """The scanners themselves, on hand-written listings (no compiler needed)."""
    from tools.scan_isa_waits import scan_small_load_batches, scan_store_waits
    asm = "\n".join([
        "_Zk1:",
        "\tglobal_load_dword v1, v[2:3], off",
        ".LBB0_1:                               ; =>This Inner Loop Header: Depth=1",
        "\ts_waitcnt vmcnt(0)",                                  # compiler wait inside a loop -> flagged
        "\tv_mfma_f32_32x32x16_f16 v[0:15], v[16:19], v[20:23], v[0:15]",
        "\t;;#ASMSTART",
        "\ts_waitcnt vmcnt(2)",                                  # hand-written -> ignored
        "\t;;#ASMEND",
        "\tv_pk_mul_f32 v[4:5], v[6:7], v[8:9] op_sel:[0,1] op_sel_hi:[0,0]",   # hi -> lo select in an MFMA kernel -> flagged
        "\tv_pk_add_f32 v[4:5], v[6:7], v[8:9] op_sel_hi:[1,0]",                # broadcast form -> fine
        "\ts_cbranch_scc1 .LBB0_1",
        "_Zk2:",
        "\tv_pk_mul_f32 v[4:5], v[6:7], v[8:9] op_sel:[0,1]",                   # no MFMA in this kernel -> not flagged
    ] + sum([["\tglobal_load_dword v1, v[2:3], off", "\ts_waitcnt vmcnt(0)", "\tglobal_store_dword v[2:3], v1, off"] for _ in range(6)], []))
    assert list(scan(asm)) == ["_Zk1"] and len(scan(asm)["_Zk1"]) == 1
    assert scan_pk_hazard(asm) == {"_Zk1": 1}
    sw = scan_store_waits(asm)
    assert list(sw) == ["_Zk2"] and sw["_Zk2"][0][2] == 5        # five of the six drains are reached with a store outstanding
    assert scan_small_load_batches(asm) == {"_Zk2": (6, 6)}
