"""Scan gfx950 ISA (hipcc -S output) for what the compiler added behind hand-scheduled memory traffic: `s_waitcnt vmcnt(..)` it put
inside a loop (scan), waits that serialise stores (scan_store_waits), packed-f32 forms that are unsafe next to MFMAs (scan_pk_hazard).

Why: the kernels here count their own outstanding loads (`asm volatile("s_waitcnt vmcnt(N)")`) so that K / V^T or A / W tiles stay
in flight across barriers.  The compiler's waitcnt pass does not read inline asm.  If a value loaded global -> register before a
loop is first used inside it, the compiler adds its own wait at that use -- in the loop body, typically vmcnt(0), which drains the
hand-counted prefetch on every iteration (found in round 3 in the attention kernels: profiles/r03/attention_q_pin_ab.txt).  The
same happens to every ds_read of a kernel that has TWO __shared__ arrays and uses global_load_lds (alias scopes from the LDS
lowering).  Hand-written waits appear between `;;#ASMSTART` / `;;#ASMEND` markers and are skipped.

usage:  python tools/scan_isa_waits.py f5_tts_mlx_amd/csrc/attention.hip [-DF5_F16=1 ...]      (compiles, then scans)
        python tools/scan_isa_waits.py file.s                                                  (scans)
"""
import os
import re
import subprocess
import sys
import tempfile


def scan(asm_text):
    """-> {kernel symbol: [(line number, instruction), ...]} of compiler-inserted vmcnt waits inside loop blocks"""
    out, fn, inloop = {}, None, False
    lines = asm_text.split("\n")
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            fn, inloop = m.group(1), False
        if re.match(r"^\.LBB", l) or re.match(r"^; %bb", l):
            inloop = "in Loop" in l or "Loop Header" in l
        if fn and inloop and "s_waitcnt" in l and "vmcnt" in l and not lines[i - 1].strip().startswith(";;#ASMSTART"):
            out.setdefault(fn, []).append((i + 1, l.strip()))
    return out


_PK = re.compile(r"^\s*(v_pk_(?:mul|add|fma)_f32)\s.*op_sel:\[(\d),(\d)(?:,(\d))?\]")


def scan_pk_hazard(asm_text):
    """-> {kernel symbol: count} of packed-f32 VALU instructions whose LO result reads the HI register of src1 (op_sel:[x,1,..]) in
    kernels that also contain MFMAs.  On gfx950 that operand comes back as 0 on lanes 48-63 now and then while ANOTHER wave of the
    SIMD has MFMAs in flight (tools/probes/pk_f32_vs_mfma2.hip; profiles/r03/pk_f32_next_to_mfma_hazard.txt) -- the root cause of the
    'lanes 48-63' nondeterminism of round 2.  The SLP vectoriser emits the form; the GEMM files are built with -fno-slp-vectorize."""
    out, mfma, fn = {}, set(), None
    for l in asm_text.split("\n"):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            fn = m.group(1)
        if not fn:
            continue
        if "v_mfma" in l:
            mfma.add(fn)
        m = _PK.match(l)
        if m and "1" in (m.group(2), m.group(3), m.group(4) or "0"):     # measured for src1; flagged for every source (none is needed)
            out[fn] = out.get(fn, 0) + 1
    return {k: v for k, v in out.items() if k in mfma}


def scan_store_waits(asm_text, min_cluster=4, gap=150):
    """-> {kernel symbol: [(first line, last line, count), ...]}: clusters of compiler-inserted vmcnt waits that are reached with
    stores outstanding.  A value loaded before a run of `if (row < M) { ... store }` blocks and first USED inside them gets the
    compiler's wait inside every block (the blocks are separate basic blocks, any of them may be the first one executed); with no
    separate store counter on gfx9 that `s_waitcnt vmcnt(0)` also waits for every store issued so far, so the stores of a lane
    leave one memory round trip apart.  Found in the conv-pos, residual and direct GEMM epilogues in round 3
    (profiles/r03/convpos_epilogue_ab.txt); the cure is one `asm volatile("" : "+v"(x))` on the loaded values before the blocks.
    Clusters in code that is never executed (e.g. the fused LN tail, default off) are reported too: read the listing."""
    out, fn, pend, ev = {}, None, 0, []
    lines = asm_text.split("\n")

    def flush():
        if fn is None or not ev:
            return
        cl = []
        for e in ev:
            if cl and e - cl[-1][1] < gap:
                cl[-1][1] = e
                cl[-1][2] += 1
            else:
                cl.append([e, e, 1])
        cl = [tuple(c) for c in cl if c[2] >= min_cluster]
        if cl:
            out[fn] = cl

    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            flush()
            fn, pend, ev = m.group(1), 0, []
        if fn is None:
            continue
        if "global_store" in l or "buffer_store" in l:
            pend += 1
        m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)
        if m and not lines[i - 1].strip().startswith(";;#ASMSTART"):
            if pend > int(m.group(1)):
                ev.append(i + 1)
            pend = min(pend, int(m.group(1)))
    flush()
    return out


def scan_small_load_batches(asm_text, max_loads=2, min_count=4):
    """-> {kernel symbol: (drain points with <= max_loads loads issued since the previous drain, all drain points)} for kernels with
    at least min_count such points.  A "drain point" is a compiler-inserted `s_waitcnt vmcnt(0 | 1)`.  Many of them with one or two
    loads in between = loads fetched one memory round trip after the other.  Found in the residual epilogue of the 256x256 kernel: the
    row-keep byte was converted to float inside the loop that requests the x rows, so every row load was followed by a full drain
    (profiles/r03/resid_epilogue_loads_ab.txt: batch-32 sample() -2.1 % once the byte stays raw until all rows are requested)."""
    out, fn, nl, small, tot = {}, None, 0, 0, 0
    lines = asm_text.split("\n")
    for i, l in enumerate(lines + ["_Zend:"]):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            if fn and small >= min_count:
                out[fn] = (small, tot)
            fn, nl, small, tot = m.group(1), 0, 0, 0
        if fn is None:
            continue
        if re.search(r"\b(global_load|buffer_load)_", l) and "lds" not in l:
            nl += 1
        m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)
        if m and i > 0 and not lines[i - 1].strip().startswith(";;#ASMSTART") and int(m.group(1)) <= 1:
            if nl > 0:
                tot += 1
                small += nl <= max_loads
            nl = 0
    return out


def compile_to_asm(src, flags=()):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as d:
        o = os.path.join(d, "k.s")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DF5_LAB=0", *flags, "-S", "--cuda-device-only", src, "-o", o]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        return open(o).read()


if __name__ == "__main__":
    path, flags = sys.argv[1], sys.argv[2:]
    text = open(path).read() if path.endswith(".s") else compile_to_asm(path, flags)
    res = scan(text)
    for k, v in res.items():
        print(k, len(v), v[:6])
    print(f"{len(res)} kernel(s) with compiler-inserted vmcnt waits inside loops")
    pk = scan_pk_hazard(text)
    for k, v in pk.items():
        print(k, v)
    print(f"{len(pk)} MFMA kernel(s) with packed-f32 instructions whose lo result reads a source's hi register")
    sw = scan_store_waits(text)
    for k, v in sw.items():
        print(k, v)
    print(f"{len(sw)} kernel(s) with clusters of compiler-inserted vmcnt waits reached with stores outstanding")
    sb = scan_small_load_batches(text)
    for k, v in sb.items():
        print(k, v)
    print(f"{len(sb)} kernel(s) with runs of full drains behind one or two loads")
