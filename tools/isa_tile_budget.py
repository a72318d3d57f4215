"""Static issue budget of the large-grid attention kernel's tile loop (VERDICT r3 #4b: "an ISA timeline of one tile iteration").

Compiles attention.hip for gfx950 (-DF5_F16=1), takes the basic blocks of f5_attn2f_kernel<true>'s tile loop that the FAST path runs
(the slow path -- reference point moves -- is the pair of ~160-instruction VALU blocks behind a wave-uniform branch), and tallies the
instructions by issue class with the issue cost of each class on CDNA3/4 (MI355X_MICROARCH.md: MFMA 32x32x16 16-bit = 8 passes x 4
cycles on the matrix pipe; full-rate VALU 4 cycles per wave64 instruction; transcendental v_exp_f32 quarter rate = 16).  Output:
profiles/r04/attention_v2f_tile_loop_isa_budget.txt.

    python tools/isa_tile_budget.py [> profiles/r04/attention_v2f_tile_loop_isa_budget.txt]
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.scan_isa_waits import compile_to_asm  # noqa: E402

COST = dict(mfma=32, exp=16, valu=4, vpk=4, cvt=4, ds_rd=4, ds_wr=4, gld=4, salu=1, wait=0, barrier=0, br=1, nop=1, prio=1)


def cls(t):
    op = t.split()[0]
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_exp") or op.startswith("v_rcp") or op.startswith("v_log"): return "exp"
    if op.startswith("v_pk_"): return "vpk"
    if op.startswith("v_cvt"): return "cvt"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "ds_rd"
    if op.startswith("ds_"): return "ds_wr"
    if op.startswith("global_load") or op.startswith("buffer_load"): return "gld"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "br"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_setprio"): return "prio"
    return "salu"


def main():
    asm = compile_to_asm(os.path.join(ROOT, "f5_tts_mlx_amd", "csrc", "attention.hip"), ["-DF5_F16=1"])
    m = re.search(r"^(_ZN4f5hf16f5_attn2f_kernelILb1EEEvNS_10F5AttnArgsE):", asm, re.M)
    code = asm[m.start():asm.index(".Lfunc_end", m.start())].split("\n")
    blocks, cur, inloop = [], None, False
    for ln in code:
        t = ln.strip()
        if re.match(r"^(\.LBB\S+:|; %bb\.\d+:)", t):
            inloop = "in Loop" in ln or "Loop Header" in ln
            cur = dict(name=t.split()[0] if not t.startswith(";") else t.split()[1], ins=[], loop=inloop)
            blocks.append(cur)
            continue
        if cur is None or not t or t.startswith(";") or t.startswith("."):
            continue
        cur["ins"].append(t)
    loop = [b for b in blocks if b["loop"]]
    print("f5_attn2f_kernel<true>, f16 build: basic blocks of the tile loop (one iteration = one 64-key tile, 64 queries per wave)\n")
    print(f"{'block':12s} {'instr':>5s}  classes")
    fast = []
    for b in loop:
        c = {}
        for t in b["ins"]:
            c[cls(t)] = c.get(cls(t), 0) + 1
        n = len(b["ins"])
        slow = c.get("valu", 0) > 100 and c.get("mfma", 0) == 0 and c.get("exp", 0) <= 4          # the max / rescale blocks
        dup_qk = False
        tag = "slow path (reference point moves)" if slow else ""
        print(f"{b['name']:12s} {n:5d}  {dict(sorted(c.items()))} {tag}")
        if not slow:
            fast.append((b, c))
    # the fast path runs QK^T once (the second 16-MFMA QK^T block belongs to the slow path's recomputation) and one of the two
    # exponential blocks: keep the first QK^T block, the first exp block, the P V block (the one with the conversions)
    qk = [x for x in fast if x[1].get("mfma", 0) == 16 and x[1].get("cvt", 0) == 0]
    ex = [x for x in fast if x[1].get("exp", 0) >= 60]
    pv = [x for x in fast if x[1].get("mfma", 0) == 16 and x[1].get("cvt", 0) >= 16]
    small = [x for x in fast if x[1].get("mfma", 0) == 0 and x[1].get("exp", 0) == 0]
    chosen = qk[:1] + ex[:1] + pv[:1] + small
    tot = {}
    for _, c in chosen:
        for k, v in c.items():
            tot[k] = tot.get(k, 0) + v
    print("\nfast path of one iteration (QK^T block + exponential block + P V block + the loop-control / staging blocks):")
    cyc = {k: v * COST[k] for k, v in tot.items()}
    for k in sorted(tot, key=lambda k: -cyc[k]):
        print(f"  {k:8s} {tot[k]:4d} instructions x {COST[k]:2d} = {cyc[k]:5d} issue cycles")
    matrix = cyc.get("mfma", 0)
    vector = sum(v for k, v in cyc.items() if k in ("exp", "valu", "vpk", "cvt"))
    print(f"\n  matrix pipe {matrix} cycles, vector ALU {vector} cycles ({cyc.get('exp', 0)} of them v_exp_f32), LDS / staging issue "
          f"{sum(v for k, v in cyc.items() if k in ('ds_rd', 'ds_wr', 'gld'))}, scalar {sum(v for k, v in cyc.items() if k in ('salu', 'br', 'nop', 'prio'))}")
    print(f"  additive (no overlap inside a wave, what two co-resident waves of a SIMD get: tools/probes/coissue.hip): {matrix + vector} cycles per wave tile")
    print(f"  overlapped inside one wave (tools/probes/inwave_overlap.hip): >= max = {max(matrix, vector)} cycles -> matrix pipe <= {100.0 * matrix / max(matrix, vector):.0f} % busy")
    print("  measured (profiles/r04/attention_v2q_persistent_rejected.jsonl fits, 2.0 GHz): v2f 0.99 us = ~1 980 cycles per tile and SIMD in steady state,")
    print("  v2p (in-wave pipeline) 0.92 us = ~1 840; whole launch incl. per-workgroup costs: 2 730 cycles per wave tile (336 us at 64 x 16 x 937).")
    print("  waits in the loop: only the hand-counted vmcnt at the top and the lgkmcnt(0) in front of each MFMA group's first use of an LDS fragment")
    for b, _ in chosen:
        w = [t for t in b["ins"] if t.startswith("s_waitcnt")]
        if w:
            print(f"    {b['name']}: {w}")


if __name__ == "__main__":
    main()
