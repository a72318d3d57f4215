"""Model-level parity on the GPU: DiT forward and F5TTS.sample (HIP engine through the C ABI) vs the oracle.

Gate (BASELINE.json north_star): mean |mel_engine - mel_oracle| <= 1e-3 on identical weights / inputs /
injected y0.  It is asserted in the modes that meet it: "f16" (IEEE-half MFMA operands, one pass -- the
mode bench.py times) and "bf16x3" (hi/lo split bf16 operands, three passes, fp32-class).  The plain
bf16 mode is checked against the oracle evaluated with the SAME bf16 operand rounding (kernel-bug
detector) and its drift vs the fp32 oracle is reported, not hidden.
"""
import ctypes as C
import warnings

import numpy as np
import pytest
import torch

from f5test import DEV, E, O, TINY, F5TTS_335M, report, synth_inputs, synthetic_weights
from f5_tts_mlx_amd.cfm import F5TTS
from f5_tts_mlx_amd.dit import DiT

pytestmark = pytest.mark.gpu

MEL_L1_TOL = 1e-3


@pytest.fixture(scope="module")
def tiny_weights():
    return synthetic_weights(TINY, seed=42)


def _model(cfg, weights, precision):
    m = DiT.from_config(cfg, precision=precision, device=DEV)
    m.load_weights(weights)
    return m


@pytest.fixture(scope="module")
def tiny_bf16(tiny_weights):
    return _model(TINY, tiny_weights, "bf16")


@pytest.fixture(scope="module")
def tiny_x3(tiny_weights):
    return _model(TINY, tiny_weights, "bf16x3")


@pytest.fixture(scope="module")
def tiny_f16(tiny_weights):
    return _model(TINY, tiny_weights, "f16")


def _pad_cond(cond, N):
    B, n_ref, mel = cond.shape
    out = torch.zeros((B, N, mel))
    out[:, :n_ref] = cond
    return out


@pytest.mark.parametrize("B,N,ragged", [(1, 70, False), (2, 150, True), (3, 64, False)])
def test_dit_forward_parity(tiny_weights, tiny_bf16, tiny_x3, B, N, ragged):
    cfg = TINY
    cond, text, durations, y0 = synth_inputs(cfg, B, N, nt=24, n_ref=30, seed=B * 100 + N, ragged=ragged)
    step_cond = _pad_cond(cond, N)          # dit-level input: already masked/padded
    mask = O.lens_to_mask(torch.tensor(durations), N) if B > 1 else None
    t = 0.37
    ref = {}
    for name, kw in (("fp32", dict()), ("emu", dict(emulate_bf16=True))):
        orc = O.DiTOracle(cfg, tiny_weights, **kw)
        ref[name] = (orc.forward(y0, step_cond, text, torch.tensor(t), False, False, mask),
                     orc.forward(y0, step_cond, text, torch.tensor(t), True, True, mask))
    for model, rname, tol in ((tiny_x3, "fp32", 2e-4), (tiny_bf16, "emu", 1.5e-2)):
        pred, null = model.engine.dit_forward(y0.to(DEV), text.to(DEV), step_cond.to(DEV).contiguous(), [N] * B, durations, t,
                                              use_mask=B > 1)
        torch.cuda.synchronize()
        for nm, got, want in (("pred", pred, ref[rname][0]), ("null", null, ref[rname][1])):
            assert torch.isfinite(got).all()
            mx, mean, refm = report(f"dit_forward[{model.precision}] {nm} vs oracle[{rname}] B{B} N{N}", got.cpu(), want)
            assert mean <= tol * max(1.0, refm), (nm, model.precision)
    # drift of plain bf16 vs the fp32 oracle (reported; bounded loosely)
    pred, _ = tiny_bf16.engine.dit_forward(y0.to(DEV), text.to(DEV), step_cond.to(DEV).contiguous(), [N] * B, durations, t,
                                           use_mask=B > 1)
    _, mean, refm = report(f"dit_forward[bf16] drift vs oracle[fp32] B{B} N{N}", pred.cpu(), ref["fp32"][0])
    assert mean <= 3e-2 * max(1.0, refm)


def test_dit_call_signature(tiny_bf16, tiny_weights):
    """DiT.__call__ keeps the reference signature (dit.py:374-383)."""
    cfg = TINY
    cond, text, durations, y0 = synth_inputs(cfg, 1, 40, nt=10, n_ref=12, seed=3)
    sc = _pad_cond(cond, 40)
    out = tiny_bf16(x=y0, cond=sc, text=text, time=torch.tensor(0.5), drop_audio_cond=False, drop_text=False, mask=None)
    null = tiny_bf16(x=y0, cond=sc, text=text, time=torch.tensor(0.5), drop_audio_cond=True, drop_text=True, mask=None)
    assert out.shape == (1, 40, cfg.mel_dim) and null.shape == out.shape
    assert float((out - null).abs().mean()) > 1e-4
    # audio conditioning dropped, text kept (training combination) == conditional branch on a zero cond (dit.py:245-247)
    a = tiny_bf16(x=y0, cond=sc, text=text, time=torch.tensor(0.5), drop_audio_cond=True, drop_text=False)
    b = tiny_bf16(x=y0, cond=torch.zeros_like(sc), text=text, time=torch.tensor(0.5), drop_audio_cond=False, drop_text=False)
    assert torch.equal(a, b) and float((a - out).abs().mean()) > 1e-4
    # text dropped, audio kept: accepted since round 4 (test_dit_call_remaining_argument_combinations checks it against the oracle)
    c = tiny_bf16(x=y0, cond=sc, text=text, time=torch.tensor(0.5), drop_audio_cond=False, drop_text=True)
    assert c.shape == out.shape and float((c - out).abs().mean()) > 1e-4 and float((c - null).abs().mean()) > 1e-4


@pytest.mark.parametrize("method,steps", [("euler", 8), ("midpoint", 5), ("rk4", 4)])
@pytest.mark.parametrize("B", [1, 2])
def test_sample_parity(tiny_weights, tiny_bf16, tiny_x3, method, steps, B):
    cfg = TINY
    N = 96
    cond, text, durations, y0 = synth_inputs(cfg, B, N, nt=20, n_ref=33, seed=17 + B, ragged=B > 1)
    kw = dict(steps=steps, method=method, cfg_strength=2.0, sway_sampling_coef=-1.0)
    dur_t = torch.tensor(durations)
    o_fp32 = O.sample(O.DiTOracle(cfg, tiny_weights), cond, text, dur_t, y0=y0, return_aux=True, **kw)
    o_emu = O.sample(O.DiTOracle(cfg, tiny_weights, emulate_bf16=True), cond, text, dur_t, y0=y0, **kw)

    f5 = F5TTS(transformer=tiny_x3)
    out, traj = f5.sample(cond, text, duration=dur_t, y0=y0, **kw)
    torch.cuda.synchronize()
    assert out.shape == (B, N, cfg.mel_dim) and traj.shape == (steps, B, N, cfg.mel_dim)
    _, l1, _ = report(f"sample[bf16x3] {method} B{B} final mel vs oracle[fp32]", out.cpu(), o_fp32[0])
    assert l1 <= MEL_L1_TOL
    _, l1t, _ = report(f"sample[bf16x3] {method} B{B} trajectory vs oracle[fp32]", traj.cpu(), o_fp32[1])
    assert l1t <= MEL_L1_TOL
    assert torch.equal(traj[0].cpu(), y0)                       # trajectory[0] is the injected noise
    aux = o_fp32[2]                                             # conditioning frames are spliced back exactly
    cm = aux["cond_mask"]
    assert torch.equal(out.cpu()[cm], torch.nn.functional.pad(cond, (0, 0, 0, N - cond.shape[1]))[cm])

    f5b = F5TTS(transformer=tiny_bf16)
    outb, _ = f5b.sample(cond, text, duration=dur_t, y0=y0, **kw)
    _, l1e, _ = report(f"sample[bf16] {method} B{B} vs oracle[bf16-emulated]", outb.cpu(), o_emu[0])
    _, l1d, _ = report(f"sample[bf16] {method} B{B} drift vs oracle[fp32]", outb.cpu(), o_fp32[0])
    assert l1e <= 2e-2 and l1d <= 5e-2


def test_sample_graph_equals_eager(tiny_bf16):
    cfg = TINY
    cond, text, durations, y0 = synth_inputs(cfg, 2, 80, nt=16, n_ref=20, seed=5, ragged=True)
    f5 = F5TTS(transformer=tiny_bf16)
    kw = dict(duration=torch.tensor(durations), steps=6, method="euler", y0=y0)
    a, ta = f5.sample(cond, text, use_graph=False, **kw)
    b, tb = f5.sample(cond, text, use_graph=True, **kw)
    c, tc = f5.sample(cond, text, use_graph=True, **kw)          # replay of the cached graph
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(ta, tb) and torch.equal(b, c) and torch.equal(tb, tc)


@pytest.mark.parametrize("method,steps", [("euler", 8), ("midpoint", 5), ("rk4", 4)])
@pytest.mark.parametrize("B", [1, 2])
def test_sample_parity_f16(tiny_weights, tiny_f16, method, steps, B):
    """The one-pass fp16-operand mode meets the 1e-3 gate against the fp32 oracle, and agrees with the oracle run with the
    same fp16 operand rounding (kernel-bug detector: that difference is accumulation order + fast-math activations only)."""
    cfg = TINY
    N = 96
    cond, text, durations, y0 = synth_inputs(cfg, B, N, nt=20, n_ref=33, seed=17 + B, ragged=B > 1)
    kw = dict(steps=steps, method=method, cfg_strength=2.0, sway_sampling_coef=-1.0)
    dur_t = torch.tensor(durations)
    o_fp32 = O.sample(O.DiTOracle(cfg, tiny_weights), cond, text, dur_t, y0=y0, **kw)
    o_emu = O.sample(O.DiTOracle(cfg, tiny_weights, emulate_f16=True), cond, text, dur_t, y0=y0, **kw)
    out, traj = F5TTS(transformer=tiny_f16).sample(cond, text, duration=dur_t, y0=y0, **kw)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    _, l1, _ = report(f"sample[f16] {method} B{B} final mel vs oracle[fp32]", out.cpu(), o_fp32[0])
    _, l1t, _ = report(f"sample[f16] {method} B{B} trajectory vs oracle[fp32]", traj.cpu(), o_fp32[1])
    _, l1e, _ = report(f"sample[f16] {method} B{B} vs oracle[f16-emulated]", out.cpu(), o_emu[0])
    assert l1 <= MEL_L1_TOL and l1t <= MEL_L1_TOL
    assert l1e <= MEL_L1_TOL


@pytest.mark.parametrize("B,N,ragged", [(1, 70, False), (2, 150, True)])
def test_dit_forward_parity_f16(tiny_weights, tiny_f16, B, N, ragged):
    cfg = TINY
    cond, text, durations, y0 = synth_inputs(cfg, B, N, nt=24, n_ref=30, seed=B * 100 + N, ragged=ragged)
    step_cond = _pad_cond(cond, N)
    mask = O.lens_to_mask(torch.tensor(durations), N) if B > 1 else None
    t = 0.37
    pred, null = tiny_f16.engine.dit_forward(y0.to(DEV), text.to(DEV), step_cond.to(DEV).contiguous(), [N] * B, durations, t,
                                             use_mask=B > 1)
    torch.cuda.synchronize()
    for kw, tol in ((dict(), 1e-3), (dict(emulate_f16=True), 3e-4)):
        orc = O.DiTOracle(cfg, tiny_weights, **kw)
        for nm, got, drop in (("pred", pred, False), ("null", null, True)):
            want = orc.forward(y0, step_cond, text, torch.tensor(t), drop, drop, mask)
            _, mean, refm = report(f"dit_forward[f16] {nm} vs oracle[{'f16-emulated' if kw else 'fp32'}] B{B} N{N}", got.cpu(), want)
            assert mean <= tol * max(1.0, refm)


@pytest.mark.parametrize("variant", [dict(conv_layers=0), dict(text_mask_padding=False), dict(conv_layers=0, text_mask_padding=False)])
def test_dit_text_embedding_variants(variant):
    """TextEmbedding constructor variants of the reference DiT (dit.py:182-194,226-227): conv_layers=0 (plain embedding: no
    positional table, no ConvNeXt blocks, no masking) and text_mask_padding=False (blocks without zeroing the filler positions);
    cond and null branch, ragged masked batch, vs the oracle (itself fuzzed against the reference's code with these variants)."""
    import dataclasses
    cfg = dataclasses.replace(TINY, **variant)
    w = synthetic_weights(cfg, seed=11)
    B, N = 2, 90
    cond, text, durations, y0 = synth_inputs(cfg, B, N, nt=20, n_ref=25, seed=4, ragged=True)
    text[1, 12:] = -1                                            # filler tokens inside the text: where the masking differs
    step_cond = _pad_cond(cond, N)
    mask = O.lens_to_mask(torch.tensor(durations), N)
    orc = O.DiTOracle(cfg, w)
    m = _model(cfg, w, "bf16x3")
    for drop in (False, True):
        ref = orc.forward(y0, step_cond, text, torch.tensor(0.35), drop, drop, mask)
        got = m(y0, step_cond, text, 0.35, drop, drop, mask)
        torch.cuda.synchronize()
        mx_, l1, refm = report(f"DiT forward {variant} drop={drop} [bf16x3]", got.cpu(), ref)
        assert l1 <= 2e-4 * max(1.0, refm)


def test_graph_replay_with_changed_cfg(tiny_bf16):
    """cfg_strength is read from workspace memory staged per call, not frozen into the captured graph: the same shapes with
    guidance 2.0 -> 3.0 -> 2.0 replay ONE cached graph and every result equals the eager run bit for bit."""
    cfg = TINY
    cond, text, durations, y0 = synth_inputs(cfg, 2, 72, nt=16, n_ref=20, seed=8, ragged=True)
    f5 = F5TTS(transformer=tiny_bf16)
    kw = dict(duration=torch.tensor(durations), steps=5, method="midpoint", y0=y0)
    eager = {c: f5.sample(cond, text, cfg_strength=c, use_graph=False, **kw)[0].clone() for c in (2.0, 3.0)}
    assert float((eager[2.0] - eager[3.0]).abs().mean()) > 1e-3           # the guidance scale matters
    n0 = tiny_bf16.engine.graph_count()
    got = [f5.sample(cond, text, cfg_strength=c, use_graph=True, **kw)[0].clone() for c in (2.0, 3.0, 2.0)]
    torch.cuda.synchronize()
    assert tiny_bf16.engine.graph_count() == n0 + 1                          # one signature, one graph
    assert torch.equal(got[0], eager[2.0]) and torch.equal(got[1], eager[3.0]) and torch.equal(got[2], eager[2.0])


@pytest.mark.parametrize("prec", ["f16", "bf16x3"])
def test_fused_ln_tail_equals_separate_ln_launches(tiny_weights, prec):
    """LN-modulate fused behind the residual GEMMs (small-M launches) must not change a single bit of sample():
    ragged masked batch and batch 1, eager and graph, tiny configuration; the 335M / N = 937 shapes are covered per op
    (test_gemm_resid_gate_fused_ln_is_bit_identical).  The fusion is off by default (measured slower)."""
    lib = E.load_library()
    cfg = TINY
    m = _model(cfg, tiny_weights, prec)
    f5 = F5TTS(transformer=m)
    for B, N, ragged in ((2, 72, True), (1, 130, False)):
        cond, text, durations, y0 = synth_inputs(cfg, B, N, nt=16, n_ref=20, seed=B + N, ragged=ragged)
        kw = dict(duration=torch.tensor(durations), steps=4, method="midpoint", y0=y0)
        got = {}
        for on in (1, 0):
            m.engine.set_option("ln_fusion", on)                 # this engine only; part of its graph key
            try:
                got[on] = [f5.sample(cond, text, use_graph=g, **kw) for g in (False, True)]
                torch.cuda.synchronize()
                got[on] = [(o.clone(), t.clone()) for o, t in got[on]]
            finally:
                m.engine.set_option("ln_fusion", 0)
        for (o1, t1), (o0, t0) in zip(got[1], got[0]):
            assert torch.isfinite(o1).all()
            assert torch.equal(o1, o0) and torch.equal(t1, t0)
        assert torch.equal(got[1][0][0], got[1][1][0])           # graph == eager


def test_two_engines_keep_their_own_options(tiny_weights):
    """Launch options are per engine (f5_engine_set_option), not process-global: two engines of one process with different
    settings give their own results in any call order, graph or eager, and a process-wide default set later
    (f5_debug_set_q_premul) does not reach engines that already exist."""
    lib = E.load_library()
    cfg = TINY
    a, b = _model(cfg, tiny_weights, "f16"), _model(cfg, tiny_weights, "f16")
    a.engine.set_option("q_premul", 0)                           # plain q, attention scales the scores itself
    assert a.engine.get_option("q_premul") == 0 and b.engine.get_option("q_premul") == 1
    cond, text, durations, y0 = synth_inputs(cfg, 2, 96, nt=16, n_ref=20, seed=9, ragged=True)
    kw = dict(duration=torch.tensor(durations), steps=4, method="euler", y0=y0)
    fa, fb = F5TTS(transformer=a), F5TTS(transformer=b)
    ra = fa.sample(cond, text, use_graph=False, **kw)[0].clone()
    rb = fb.sample(cond, text, use_graph=False, **kw)[0].clone()
    assert not torch.equal(ra, rb) and float((ra - rb).abs().mean()) < 1e-3      # one rounding of q apart, not the same bits
    try:
        E.check(lib.f5_debug_set_q_premul(0))                    # a new DEFAULT: existing engines keep their option
        for g in (True, False, True):
            for f5, ref in ((fa, ra), (fb, rb), (fb, rb), (fa, ra)):
                assert torch.equal(f5.sample(cond, text, use_graph=g, **kw)[0], ref)
        c = _model(cfg, tiny_weights, "f16")                     # created under the new default
        assert c.engine.get_option("q_premul") == 0
        assert torch.equal(F5TTS(transformer=c).sample(cond, text, use_graph=False, **kw)[0], ra)
    finally:
        E.check(lib.f5_debug_set_q_premul(1))
    with pytest.raises(RuntimeError):
        a.engine.set_option("no_such_option", 1)


def test_graph_cache_is_bounded_and_auto_mode(tiny_weights):
    """At most `set_graph_cache(n)` graph executables are kept (LRU); "auto" runs a new signature eagerly and captures it on
    its second sighting; every variant returns the eager bits."""
    cfg = TINY
    m = _model(cfg, tiny_weights, "bf16")
    f5 = F5TTS(transformer=m)
    m.engine.set_graph_cache(2)
    # size the workspace for the largest shape first: a re-allocation would (correctly) drop the graphs captured before it
    m.engine.workspace(1, 70, 12, 4, "euler")
    ref = {}
    for N in (40, 52, 64, 70):
        cond, text, durations, y0 = synth_inputs(cfg, 1, N, nt=12, n_ref=10, seed=N)
        kw = dict(duration=N, steps=4, method="euler", y0=y0)
        ref[N] = (cond, text, kw, f5.sample(cond, text, use_graph=False, **kw)[0].clone())
    assert m.engine.graph_count() == 0
    for N in (40, 52, 64, 40):
        cond, text, kw, want = ref[N]
        assert torch.equal(f5.sample(cond, text, use_graph=True, **kw)[0], want)
        assert m.engine.graph_count() <= 2
    cond, text, kw, want = ref[70]
    n0 = m.engine.graph_count()
    assert torch.equal(f5.sample(cond, text, use_graph="auto", **kw)[0], want) and m.engine.graph_count() == n0   # eager
    assert torch.equal(f5.sample(cond, text, use_graph="auto", **kw)[0], want) and m.engine.graph_count() <= 2    # captured
    assert torch.equal(f5.sample(cond, text, use_graph="auto", **kw)[0], want)                                     # replayed


@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
def test_graph_split_per_ode_step_is_bit_identical(tiny_weights, method):
    """Engine option graph_split (round-6 probe of the hipGraph replay cost, tools/r6_graph_probe.py): the call captured as one exec
    per ODE step and replayed back to back returns the bits of the single exec and of the eager launches, for every solver, with a
    changed guidance scale on the replay (scalars are data, not baked arguments), and counts as ONE entry of the graph cache."""
    cfg = TINY
    m = _model(cfg, tiny_weights, "f16")
    f5 = F5TTS(transformer=m)
    cond, text, durations, y0 = synth_inputs(cfg, 2, 48, nt=12, n_ref=10, seed=3, ragged=True)
    kw = dict(duration=torch.tensor(durations), steps=5, method=method, y0=y0)
    want = {c: f5.sample(cond, text, use_graph=False, cfg_strength=c, **kw)[0].clone() for c in (2.0, 0.7)}
    for split in (0, 1):
        m.engine.set_option("graph_split", split)
        n0 = m.engine.graph_count()
        for c in (2.0, 0.7, 2.0):
            assert torch.equal(f5.sample(cond, text, use_graph=True, cfg_strength=c, **kw)[0], want[c]), (split, c)
        assert m.engine.graph_count() == n0 + 1
    m.engine.set_option("graph_split", 0)


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_split_sample_is_bit_identical(tiny_weights, prec):
    """Round 6: sample() of a large batch runs as two half batches on two HIP streams (a sibling handle on the same weights arena,
    enqueued by a second host thread; engine.py Engine._sample_split).  Utterances do not interact, so the bits are those of the unsplit
    call: ragged batch of 5 (halves of 3 and 2, both padded to the batch's N, both under the batch's key-padding mask rule), graph
    replay and eager launches (two threads inside f5_sample at once: the per-thread status pointer of the range detector), every
    solver output incl. the trajectory, a caller-provided trajectory buffer, a changed guidance scale on the replay."""
    cfg = TINY
    m = _model(cfg, tiny_weights, prec)
    eng = m.engine
    f5 = F5TTS(transformer=m)
    cond, text, durations, y0 = synth_inputs(cfg, 5, 96, nt=20, n_ref=16, seed=11, ragged=True)
    kw = dict(duration=torch.tensor(durations), steps=5, y0=y0)
    for method in ("euler", "rk4"):
        eng.split_batch = 0
        want = {(c, g): [x.clone() for x in f5.sample(cond, text, method=method, cfg_strength=c, use_graph=g, **kw)]
                for c in (2.0, 0.0) for g in (True, False)}
        eng.split_batch = 4
        n0 = eng.split_events
        for rep in range(3):                           # first call of a shape: the halves one after the other (capture), then concurrently
            for (c, g), (wo, wt) in want.items():
                out, traj = f5.sample(cond, text, method=method, cfg_strength=c, use_graph=g, **kw)
                torch.cuda.synchronize()
                assert torch.equal(out, wo) and torch.equal(traj, wt), (method, rep, c, g)
        assert eng.split_events == n0 + 12 and eng._sibling is not None and eng._sibling.arena.data_ptr() == eng.arena.data_ptr()
    # the engine-level entry with caller-owned buffers
    B, N, mel = 5, 96, cfg.mel_dim
    cond_p = _pad_cond(cond, N).to(DEV)
    lens = [16] * B
    t = np.linspace(0, 1, 5, dtype=np.float32)
    outb, trb = torch.empty((B, N, mel), device=DEV), torch.empty((5, B, N, mel), device=DEV)
    args = (text.to(DEV).to(torch.int32).contiguous(), cond_p, lens, durations, y0.to(DEV), t)
    eng.split_batch = 0
    o0, t0 = eng.sample(*args, method="midpoint")
    eng.split_batch = 2
    for _ in range(2):
        o1, t1 = eng.sample(*args, method="midpoint", out=outb, trajectory=trb)
        torch.cuda.synchronize()
        assert o1.data_ptr() == outb.data_ptr() and t1.data_ptr() == trb.data_ptr()
        assert torch.equal(o1, o0) and torch.equal(t1, t0)
    # options set later reach the sibling
    eng.set_option("null_keeps_cond", 1)
    assert eng._sibling.get_option("null_keeps_cond") == 1
    eng.set_option("null_keeps_cond", 0)


def test_split_sample_soak():
    """tools/r6_split_soak.py: random batch sizes / lengths / solvers / graph modes / guidance scales through the split path, more shapes
    than the graph cache holds (graphs are evicted and captured again from two host threads at once), every result bitwise equal to
    the unsplit call."""
    import os
    import subprocess
    import sys
    from f5test import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "r6_split_soak.py"), "150"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count('"mismatches": []') == 2


def test_split_sample_flagged_half_repeats_the_call_unsplit():
    """A half batch whose status word comes back set (here: FF1 / adaLN rows that saturate fp16, as in
    test_f16_saturation_detector_batch1) makes the split path drop its result; the ordinary path then repeats the WHOLE call, warns and
    falls back to bf16x3 as it does without the split: the caller gets the bf16x3 bits."""
    cfg = TINY
    w = synthetic_weights(cfg, seed=42)
    r = np.random.default_rng(5)
    pre = "transformer.transformer_blocks.1."
    for row in r.integers(0, w[pre + "ff.ff.layers.0.layers.0.weight"].shape[0], 4):
        w[pre + "ff.ff.layers.0.layers.0.weight"][row] *= 3.0e6
    cond, text, durations, y0 = synth_inputs(cfg, 4, 64, nt=12, n_ref=10, seed=3, ragged=True)
    kw = dict(duration=torch.tensor(durations), steps=4, method="euler", y0=y0)
    ref, _ = F5TTS(transformer=_model(cfg, w, "bf16x3")).sample(cond, text, **kw)
    torch.cuda.synchronize()
    m = _model(cfg, w, "f16")
    m.engine.split_batch = 2
    f5 = F5TTS(transformer=m)
    with pytest.warns(E.OperandRangeWarning, match="re-running it in bf16x3"):
        out, _ = f5.sample(cond, text, **kw)
    torch.cuda.synchronize()
    assert m.engine.saturation_events == 1 and m.engine.split_events == 0 and m.engine._use_fallback
    assert torch.equal(out.cpu(), ref.cpu())


def test_c_abi_weight_broadcast_over_rccl(tiny_weights):
    """f5_broadcast_weights: the contiguous arena goes through ONE ncclBroadcast on a caller-owned RCCL communicator (here a
    1-rank communicator created with ctypes on librccl, the way a non-Python host would own one); the receiving engine is
    marked loaded.  The N-rank protocol itself is covered on CPU by tests/test_dist.py (gloo)."""
    import ctypes as C
    class UniqueId(C.Structure):                       # ncclUniqueId is passed BY VALUE
        _fields_ = [("internal", C.c_char * 128)]
    rccl = C.CDLL("librccl.so.1")
    torch.cuda.set_device(torch.device(DEV))
    torch.zeros(1, device=DEV)                          # make sure the HIP context of this device exists
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        m = _model(TINY, tiny_weights, "f16")
        before = m.engine.arena.clone()
        lib = E.load_library()
        E.check(lib.f5_broadcast_weights(m.engine._h, comm, 0, 0, E.stream_ptr(torch.device(DEV))), "f5_broadcast_weights")
        torch.cuda.synchronize()
        assert torch.equal(m.engine.arena, before)                      # root keeps its bytes
        # a second engine that never saw f5_load_tensor: copy the arena (what a broadcast delivers), mark through the C call
        m2 = DiT.from_config(TINY, precision="f16", device=DEV)
        m2.engine.arena.copy_(m.engine.arena)
        E.check(lib.f5_broadcast_weights(m2.engine._h, comm, 0, 1, E.stream_ptr(torch.device(DEV))), "f5_broadcast_weights")
        torch.cuda.synchronize()
        m2.engine.finalize()                                            # would raise "never loaded" without the mark
        cond, text, durations, y0 = synth_inputs(TINY, 1, 48, nt=10, n_ref=12, seed=2)
        kw = dict(duration=48, steps=3, method="euler", y0=y0)
        assert torch.equal(F5TTS(transformer=m).sample(cond, text, **kw)[0], F5TTS(transformer=m2).sample(cond, text, **kw)[0])
    finally:
        rccl.ncclCommDestroy(comm)


def test_sample_no_cfg_and_errors(tiny_bf16, tiny_weights):
    cfg = TINY
    cond, text, durations, y0 = synth_inputs(cfg, 1, 64, nt=12, n_ref=16, seed=8)
    f5 = F5TTS(transformer=tiny_bf16)
    out, _ = f5.sample(cond, text, duration=64, steps=4, method="euler", cfg_strength=0.0, y0=y0)
    ref = O.sample(O.DiTOracle(cfg, tiny_weights, emulate_bf16=True), cond, text, 64, y0=y0, steps=4, method="euler",
                   cfg_strength=0.0)
    _, l1, _ = report("sample cfg=0 vs oracle[emu]", out.cpu(), ref[0])
    assert l1 <= 2e-2
    with pytest.raises(ValueError, match="Unknown method: heun"):
        f5.sample(cond, text, duration=64, steps=4, method="heun", y0=y0)
    with pytest.raises(ValueError, match="Duration must be provided"):
        f5.sample(cond, text, duration=None, steps=4, method="euler", y0=y0)
    # duration is clamped to lens + 1 (cfm.py:317): asking for fewer frames than the reference audio
    out2, traj2 = f5.sample(cond, text, duration=3, steps=3, method="euler", seed=1)
    assert out2.shape[1] == 17


def test_seed_determinism(tiny_bf16):
    cfg = TINY
    cond, text, _, _ = synth_inputs(cfg, 1, 64, nt=12, n_ref=16, seed=8)
    f5 = F5TTS(transformer=tiny_bf16)
    a, _ = f5.sample(cond, text, duration=48, steps=3, method="euler", seed=1234)
    b, _ = f5.sample(cond, text, duration=48, steps=3, method="euler", seed=1234)
    c, _ = f5.sample(cond, text, duration=48, steps=3, method="euler", seed=1235)
    assert torch.equal(a, b) and not torch.equal(a, c)


def test_full_size_forward_parity():
    """One full-size (335M, N=937) CFG evaluation: bf16x3 engine vs the fp32 oracle."""
    cfg = F5TTS_335M
    w = synthetic_weights(cfg, seed=42)
    N = 937
    cond, text, durations, y0 = synth_inputs(cfg, 1, N, nt=160, n_ref=281, seed=1)
    sc = _pad_cond(cond, N)
    orc = O.DiTOracle(cfg, w)
    want = orc.forward(y0, sc, text, torch.tensor(0.25), False, False, None)
    for prec, tol in (("bf16x3", 2e-4), ("bf16", 2e-2)):
        m = _model(cfg, w, prec)
        pred, null = m.engine.dit_forward(y0.to(DEV), text.to(DEV), sc.to(DEV), [N], durations, 0.25)
        torch.cuda.synchronize()
        _, mean, refm = report(f"full-size forward [{prec}] vs oracle[fp32]", pred.cpu(), want)
        assert mean <= tol * max(1.0, refm)
        del m
        torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------
# vocoder + full wave path
# ------------------------------------------------------------------------------------------------
def test_istft_op_matches_torch_istft():
    lib = E.load_library()
    r = np.random.default_rng(2)
    N = 37
    y = torch.from_numpy(r.standard_normal((N, 1026)).astype(np.float32))
    y[:, :513] *= 1.5
    y[3, 7] = 9.0                                                           # exercises the exp clip at 1e2
    yd = y.to(DEV)
    win = torch.hann_window(1024, device=DEV)
    frames = torch.empty((N, 1024), device=DEV)
    wave = torch.empty(256 * (N - 1), device=DEV)
    E.check(lib.f5_op_istft(E.ptr(yd), 1026, E.ptr(win), E.ptr(frames), E.ptr(wave), N, 1024, 256, E.stream_ptr(torch.device(DEV))))
    torch.cuda.synchronize()
    mag, ph = y.double().T.chunk(2, dim=0)
    S = torch.clip(torch.exp(mag), max=1e2) * (torch.cos(ph) + 1j * torch.sin(ph))
    ref = torch.istft(S[None], 1024, 256, 1024, torch.hann_window(1024, dtype=torch.float64), center=True)[0]
    mx, mean, refm = report("istft op vs torch.istft", wave.cpu(), ref)
    assert mx <= 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("B,N", [(1, 60), (2, 131)])
def test_vocos_decode_parity(B, N):
    """f5_vocode (one C-ABI call, hipGraph) vs the CPU restatement of the published Vocos architecture: fp32-class in bf16x3,
    and in the one-pass modes against the oracle with the same operand rounding."""
    from oracle import vocos_oracle as VO
    from f5_tts_mlx_amd.vocos import Vocos, synthetic_vocos_weights
    w = synthetic_vocos_weights(seed=7)
    r = np.random.default_rng(N)
    mel = torch.from_numpy((r.standard_normal((B, N, 100)) * 2.0 - 1.0).astype(np.float32))
    ref = VO.decode(w, mel, dtype=torch.float64)
    for prec, want, tol in (("bf16x3", ref, 2e-4), ("bf16", VO.decode(w, mel, dtype=torch.float64, emulate_bf16=True), 2e-3),
                            ("f16", VO.decode(w, mel, dtype=torch.float64, emulate_f16=True), 5e-4)):
        v = Vocos(w, precision=prec, device=DEV)
        wave = v.decode(mel)
        torch.cuda.synchronize()
        wave = wave[None] if wave.ndim == 1 else wave
        assert wave.shape == (B, 256 * (N - 1))
        mx, mean, refm = report(f"vocos decode [{prec}] B{B} N{N}", wave.cpu(), want)
        assert mean <= tol * max(1e-3, refm) + 1e-6 and torch.isfinite(wave).all()
        if prec == "f16":
            _, mean32, _ = report(f"vocos decode [f16] B{B} N{N} vs fp64 oracle", wave.cpu(), ref)
            assert mean32 <= 2e-3 * max(1e-3, refm)


def test_vocoder_c_abi_graph_batch_and_layouts():
    """f5_vocode: a replayed graph equals the eager run bit for bit; a batch equals its utterances decoded one at a time (the
    ISTFT of the whole batch is two launches); MLX-layout conv weights (out, k, in) load to the same model as PyTorch-layout
    ones; unknown / missing tensors are refused."""
    from f5_tts_mlx_amd.vocos import Vocos, synthetic_vocos_weights
    w = synthetic_vocos_weights(seed=7)
    r = np.random.default_rng(1)
    mel = torch.from_numpy((r.standard_normal((3, 77, 100)) * 2.0 - 1.0).astype(np.float32)).to(DEV)
    vg = Vocos(w, precision="f16", device=DEV, use_graph=True)
    ve = Vocos(w, precision="f16", device=DEV, use_graph=False)
    a, b, c = vg.decode(mel), vg.decode(mel), ve.decode(mel)
    torch.cuda.synchronize()
    assert a.shape == (3, 256 * 76) and torch.equal(a, b) and torch.equal(a, c)
    for i in range(3):
        one = ve.decode(mel[i:i + 1])
        assert one.ndim == 1 and torch.equal(one, a[i])
    w_mlx = dict(w)
    w_mlx["backbone.embed.weight"] = np.ascontiguousarray(np.transpose(w["backbone.embed.weight"], (0, 2, 1)))       # (out, k, in)
    for i in range(8):
        k = f"backbone.convnext.{i}.dwconv.weight"
        w_mlx[k] = np.ascontiguousarray(np.transpose(w[k], (0, 2, 1)))                                             # (dim, 7, 1)
    w_mlx["feature_extractor.mel_spec.spectrogram.window"] = np.zeros(1024, np.float32)                               # ignored buffer
    assert torch.equal(Vocos(w_mlx, precision="f16", device=DEV).decode(mel), a)
    bad = dict(w)
    bad.pop("head.out.bias")
    with pytest.raises(ValueError, match="missing vocoder parameter"):
        Vocos(bad, device=DEV)
    bad = dict(w)
    bad["backbone.convnext.0.extra"] = np.zeros(4, np.float32)
    with pytest.raises(ValueError, match="unexpected vocoder parameter"):
        Vocos(bad, device=DEV)
    bad = dict(w)
    bad["head.out.weight"] = np.zeros((1026, 511), np.float32)
    with pytest.raises(RuntimeError, match="shape"):
        Vocos(bad, device=DEV)


def test_mel_front_end_batched_launch_equals_per_utterance():
    """f5_mel_spectrogram_batch: one launch for (B, L) equals B single launches bit for bit."""
    from f5_tts_mlx_amd.audio import log_mel_spectrogram
    r = np.random.default_rng(9)
    waves = torch.from_numpy((r.standard_normal((4, 256 * 33 + 17)) * 0.1).astype(np.float32)).to(DEV)
    batch = log_mel_spectrogram(waves)
    torch.cuda.synchronize()
    assert batch.shape == (4, 33, 100)
    for i in range(4):
        assert torch.equal(log_mel_spectrogram(waves[i])[0], batch[i])
    y = torch.from_numpy(r.standard_normal((3 * 21, 1026)).astype(np.float32)).to(DEV)
    lib = E.load_library()
    win = torch.hann_window(1024, device=DEV)
    frames = torch.empty((3 * 21, 1024), device=DEV)
    wave = torch.empty((3, 256 * 20), device=DEV)
    st = E.stream_ptr(torch.device(DEV))
    E.check(lib.f5_op_istft_batch(E.ptr(y), 1026, E.ptr(win), E.ptr(frames), E.ptr(wave), 3, 21, 1024, 256, st))
    one = torch.empty(256 * 20, device=DEV)
    fr1 = torch.empty((21, 1024), device=DEV)
    for i in range(3):
        E.check(lib.f5_op_istft(E.ptr(y[i * 21:(i + 1) * 21]), 1026, E.ptr(win), E.ptr(fr1), E.ptr(one), 21, 1024, 256, st))
        torch.cuda.synchronize()
        assert torch.equal(one, wave[i])


def test_sample_with_vocoder_and_raw_wave(tiny_x3, tiny_weights):
    """F5TTS.sample with a raw-wave `cond` (mel front-end on the GPU) and an injected vocoder (cfm.py:283-286, 399-400)."""
    from f5_tts_mlx_amd.vocos import Vocos, synthetic_vocos_weights
    cfg = TINY
    r = np.random.default_rng(4)
    wave = torch.from_numpy((r.standard_normal(256 * 20 + 100) * 0.1).astype(np.float32))
    voc = Vocos(synthetic_vocos_weights(seed=7), precision="bf16x3", device=DEV)
    f5 = F5TTS(transformer=tiny_x3, vocoder=voc.decode)
    text = torch.from_numpy(r.integers(0, cfg.text_num_embeds, (1, 12)).astype(np.int32))
    out, traj = f5.sample(wave[None], text, duration=50, steps=3, method="euler", seed=3)
    torch.cuda.synchronize()
    assert out.ndim == 1 and out.shape[0] == 256 * 49 and traj.shape == (3, 1, 50, 100) and torch.isfinite(out).all()
    # same call through the oracle pieces: mel -> sample -> vocoder
    from oracle import vocos_oracle as VO
    from f5_tts_mlx_amd.rng import mlx_like_normal
    y0 = torch.from_numpy(np.ascontiguousarray(mlx_like_normal(3, (100, 50)).T))[None]
    mel_ref, traj_ref = O.sample(O.DiTOracle(cfg, tiny_weights), wave[None], text, 50, y0=y0, steps=3, method="euler")
    _, l1, _ = report("raw-wave sample: final mel (from trajectory) vs oracle", traj[-1].cpu(), traj_ref[-1])
    assert l1 <= MEL_L1_TOL
    wref = VO.decode(synthetic_vocos_weights(seed=7), mel_ref, dtype=torch.float64)[0]
    _, l1w, refm = report("raw-wave sample: waveform vs oracle", out.cpu(), wref)
    assert l1w <= 5e-3 * max(1e-3, refm) + 1e-5


def test_device_noise_matches_the_host_generator():
    """f5_noise_normal (threefry2x32 + float32 uniform mapping + float64 erfinv on the GPU, cfm.py:369-375) against
    rng.mlx_like_normal: ragged batch, channel-major draw order, zero padding, odd element counts, 64-bit seeds; the same seed
    for every element gives equal-duration elements identical noise (reference quirk)."""
    from f5_tts_mlx_amd.engine import noise_normal
    from f5_tts_mlx_amd.rng import mlx_like_normal
    mel = 100
    for seeds, durs, N in (([3, 3, 3], [50, 937, 50], 937), ([0], [1], 4), ([2 ** 40 + 7, 123456789], [333, 2], 400),
                           ([2 ** 63 - 1], [937], 937)):
        got = noise_normal(seeds, durs, N, mel, DEV)
        torch.cuda.synchronize()
        got = got.cpu().numpy()
        for b, (sd, d) in enumerate(zip(seeds, durs)):
            ref = mlx_like_normal(sd, (mel, d)).T                             # (dur, mel)
            g = got[b, :d]
            assert np.all(got[b, d:] == 0.0)
            diff = np.abs(g.astype(np.float64) - ref.astype(np.float64))
            ulp = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
            exact = float(np.mean(g == ref))
            print(f"[noise] seed {sd} dur {d}: bit-equal fraction {exact:.6f}, max diff {diff.max():.3e}")
            assert np.all(diff <= ulp) and exact >= 0.999                     # at most a last-place rounding tie apart
            assert np.isfinite(g).all() and abs(float(g.mean())) < 0.05 + 3.0 / np.sqrt(g.size) and (d * mel < 1000 or abs(float(g.std()) - 1) < 0.05)
        if len(durs) == 3:
            assert np.array_equal(got[0, :50], got[2, :50])
    # odd element counts (the generator pairs counter i with i + ceil(n / 2): an odd n leaves the last pair half used) and more
    # than 256 batch elements (the C entry point takes 256 per call; the wrapper chunks)
    for mel_o, seeds, durs, N in ((7, [5, 6, 7], [5, 1, 9], 9), (3, list(range(300)), [1 + (i % 11) for i in range(300)], 11)):
        got = noise_normal(seeds, durs, N, mel_o, DEV).cpu().numpy()
        for b, (sd, d) in enumerate(zip(seeds, durs)):
            ref = mlx_like_normal(sd, (mel_o, d)).T
            assert np.all(got[b, d:] == 0.0)
            diff = np.abs(got[b, :d].astype(np.float64) - ref.astype(np.float64))
            assert np.all(diff <= np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)), (mel_o, b, d)


def test_generate_end_to_end(tiny_weights, tmp_path):
    """generate() control flow with the packaged reference voice: wav -> mel -> sample -> vocoder -> trimmed wave -> file."""
    from f5_tts_mlx_amd import generate as G
    from f5_tts_mlx_amd.vocos import Vocos, synthetic_vocos_weights
    import dataclasses
    vocab = {v: i for i, v in enumerate(open(str(E.library_path().parent.parent / "assets" / "vocab.txt")).read().split("\n"))}
    cfg = dataclasses.replace(TINY, text_num_embeds=len(vocab) - 1)
    model = DiT.from_config(cfg, precision="bf16", device=DEV)
    model.load_weights(synthetic_weights(cfg, seed=1))
    voc = Vocos(synthetic_vocos_weights(seed=7), device=DEV)
    f5 = F5TTS(transformer=model, vocab_char_map=vocab, vocoder=voc.decode)
    out_path = tmp_path / "gen.wav"
    wave = G.generate("Hello world.", duration=7.0, steps=3, method="euler", seed=0, output_path=str(out_path), f5tts=f5)
    total_frames = int(7.0 * 93.75)
    assert wave.ndim == 1 and wave.shape[0] == 256 * (total_frames - 1) - 127987      # reference trimmed by samples
    data, sr = G.read_wav(str(out_path))
    assert sr == 24000 and data.shape[0] == wave.shape[0] and np.isfinite(data).all()
    # two sentences -> per-sentence generation, concatenated (generate.py:199-233); needs a duration heuristic
    w2 = G.generate("One. Two.", estimate_duration=True, steps=2, method="euler", seed=0, f5tts=f5)
    assert w2.ndim == 1 and w2.shape[0] > 0


@pytest.mark.parametrize("B,N,nt", [(1, 90, 30), (2, 64, 80)])
def test_duration_predictor_parity(B, N, nt):
    """DurationPredictor (duration.py) on the HIP ops vs the oracle; also wired through F5TTS.predict_duration."""
    from oracle import duration_oracle as DO
    from f5_tts_mlx_amd.duration import DurationPredictor, DurationTransformer, synthetic_duration_weights
    kw = dict(dim=512, depth=3, text_num_embeds=70, text_dim=512, conv_layers=2, ff_mult=2)
    w = synthetic_duration_weights(seed=5, **kw)
    r = np.random.default_rng(B + N)
    mel = torch.from_numpy((r.standard_normal((B, N, 100)) * 1.5 - 1.0).astype(np.float32))
    text = torch.from_numpy(r.integers(0, 70, (B, nt)).astype(np.int32))
    text[-1, nt - 5:] = -1
    lens = torch.tensor([N] * B) if B == 1 else torch.tensor([N, N - 11])
    want = DO.predict(w, mel, text, lens=lens, depth=3, dtype=torch.float64)
    want_emu = DO.predict(w, mel, text, lens=lens, depth=3, dtype=torch.float64, emulate_bf16=True)
    for prec, ref, tol in (("bf16x3", want, 2e-4), ("bf16", want_emu, 5e-3)):
        tr = DurationTransformer(dim=512, depth=3, heads=8, text_dim=512, ff_mult=2, conv_layers=2, text_num_embeds=70,
                                 precision=prec, device=DEV)
        dp = DurationPredictor(tr)
        dp.load_weights(w)
        got = dp(mel, text, lens=lens)
        torch.cuda.synchronize()
        assert got.shape == (B,)
        mx, _, refm = report(f"duration predictor [{prec}] B{B} N{N} nt{nt}", got.cpu(), ref)
        assert mx <= tol * max(1.0, refm) and (got > 0).all()
    # F5TTS.sample(duration=None) asks the predictor (cfm.py:307-308) and turns seconds into frames at 93 frames/s (:260)
    cfg = TINY
    model = DiT.from_config(cfg, precision="bf16", device=DEV)
    model.load_weights(synthetic_weights(cfg, seed=2))
    f5 = F5TTS(transformer=model, duration_predictor=dp)
    text_small = torch.from_numpy(r.integers(0, cfg.text_num_embeds, (1, 12)).astype(np.int32))
    d = f5.predict_duration(mel[:1], text_small, speed=1.0)
    assert d.dtype == torch.int32 and d.shape == (1,)
    out, _ = f5.sample(mel[:1], text_small, duration=None, steps=2, method="euler", seed=0)
    assert out.shape[1] == max(int(d[0]), N + 1)


def test_sample_max_duration_4096(tiny_weights, tiny_x3):
    """longest legal sequence (max_duration = 4096, cfm.py:277,318): 64 KV tiles per query block, text table clamp at 4095"""
    cfg = TINY
    N = 4096
    cond, text, durations, y0 = synth_inputs(cfg, 1, N, nt=50, n_ref=100, seed=40)
    ref, _ = O.sample(O.DiTOracle(cfg, tiny_weights), cond, text, 9999, y0=y0, steps=3, method="euler")   # clipped to 4096
    out, traj = F5TTS(transformer=tiny_x3).sample(cond, text, duration=9999, y0=y0, steps=3, method="euler")
    torch.cuda.synchronize()
    assert out.shape == (1, 4096, cfg.mel_dim)
    _, l1, _ = report("sample N=4096 [bf16x3] vs oracle[fp32]", out.cpu(), ref)
    assert l1 <= MEL_L1_TOL


def test_sample_ragged_batch5(tiny_weights, tiny_x3):
    cfg = TINY
    r = np.random.default_rng(77)
    B, N = 5, 300
    durations = [300, 41, 299, 128, 65]
    n_ref = 30
    cond = torch.from_numpy(r.standard_normal((B, n_ref, cfg.mel_dim)).astype(np.float32))
    text = torch.full((B, 40), -1, dtype=torch.int32)
    for i, L in enumerate((40, 7, 33, 1, 20)):
        text[i, :L] = torch.from_numpy(r.integers(0, cfg.text_num_embeds, L).astype(np.int32))
    y0 = np.zeros((B, N, cfg.mel_dim), np.float32)
    for i, d in enumerate(durations):
        y0[i, :d] = r.standard_normal((cfg.mel_dim, d)).astype(np.float32).T
    y0 = torch.from_numpy(y0)
    dur_t = torch.tensor(durations)
    ref, _, aux = O.sample(O.DiTOracle(cfg, tiny_weights), cond, text, dur_t, y0=y0, steps=4, method="midpoint", return_aux=True)
    out, _ = F5TTS(transformer=tiny_x3).sample(cond, text, duration=dur_t, y0=y0, steps=4, method="midpoint")
    torch.cuda.synchronize()
    assert aux["duration"].tolist() == [300, 41, 299, 128, 65] and aux["lens"].tolist() == [40, 30, 33, 30, 30]
    _, l1, _ = report("sample ragged B=5 [bf16x3] vs oracle[fp32]", out.cpu(), ref)
    assert l1 <= MEL_L1_TOL


def test_full_size_sample_parity_short_solve():
    """the real 335M configuration end to end (B=1, N=937, CFG, 5-point Euler = 8 DiT forwards): bf16x3 meets the 1e-3 gate,
    bf16 drift is reported."""
    cfg = F5TTS_335M
    w = synthetic_weights(cfg, seed=42)
    cond, text, durations, y0 = synth_inputs(cfg, 1, 937, nt=160, n_ref=281, seed=1)
    ref, _ = O.sample(O.DiTOracle(cfg, w), cond, text, 937, y0=y0, steps=5, method="euler")
    for prec in ("bf16x3", "bf16"):
        m = _model(cfg, w, prec)
        out, _ = F5TTS(transformer=m).sample(cond, text, duration=937, y0=y0, steps=5, method="euler")
        torch.cuda.synchronize()
        _, l1, _ = report(f"full-size 5-point Euler sample [{prec}] vs oracle[fp32]", out.cpu(), ref)
        assert l1 <= (MEL_L1_TOL if prec == "bf16x3" else 3e-2)
        del m
        torch.cuda.empty_cache()


def test_full_size_full_length_parity_golden():
    """THE benchmark workload at full depth (bench.py utterance 0: 335M, N = 937, 32-point Euler + sway + CFG = 62 DiT
    forwards) against the fp32 oracle's answer committed as tests/golden/full_b1_euler32.npz (made by
    tests/golden/make_fullsize_golden.py, ~6 min of CPU).  The modes that claim parity must meet the 1e-3 gate on the final mel
    AND at every stored trajectory depth; plain bf16's accumulated drift is measured and reported."""
    import os
    from f5test import ROOT
    sys_path_golden = os.path.join(ROOT, "tests", "golden")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_fullsize_golden", os.path.join(sys_path_golden, "make_fullsize_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    g = np.load(os.path.join(sys_path_golden, "full_b1_euler32.npz"))
    wave, text, y0 = mg.inputs(0)
    cfg = F5TTS_335M
    w = synthetic_weights(cfg, seed=42)
    from f5_tts_mlx_amd.audio import log_mel_spectrogram
    cond = log_mel_spectrogram(torch.from_numpy(wave).to(DEV))                       # (1, 281, 100): the HIP mel front-end
    _, l1c, _ = report("full-length golden: mel front-end (HIP) vs oracle", cond[0].cpu(), torch.from_numpy(g["cond281"]))
    assert l1c <= 1e-5
    res = {}
    for prec in ("f16", "bf16x3", "bf16"):
        m = _model(cfg, w, prec)
        out, traj = F5TTS(transformer=m).sample(cond, torch.from_numpy(text)[None], duration=mg.N_FRAMES,
                                                y0=torch.from_numpy(y0)[None], steps=mg.ODE_POINTS, method="euler",
                                                cfg_strength=2.0, sway_sampling_coef=-1.0)
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        res[prec] = [report(f"full-length 32-point Euler [{prec}] {k} vs oracle[fp32] golden", v.cpu(), torch.from_numpy(g[k]))[1]
                     for k, v in (("traj_8", traj[8, 0]), ("traj_16", traj[16, 0]), ("traj_24", traj[24, 0]), ("out", out[0]))]
        del m
        torch.cuda.empty_cache()
    assert max(res["f16"]) <= MEL_L1_TOL, res
    assert max(res["bf16x3"]) <= MEL_L1_TOL, res
    assert max(res["bf16"]) <= 3e-2, res                                             # reported; NOT a parity mode


def test_batch32_every_utterance_parity():
    """BASELINE configs[2], the configuration the MFMA roofline is quoted on (335M, batch 32 x N = 937, 32-point Euler + sway +
    CFG, f16 operands): EVERY utterance of the batched call -- the 256 x 256 GEMM tiles, the large-grid attention kernel, all row
    ranges of the M = 59 968 launches -- against the same utterance solved alone at batch 1 in `bf16x3` (fp32-class arithmetic on
    the small-shape kernels, itself pinned to the fp32 oracle's golden for utterance 0), mel L1 <= 1e-3 each (cfm.py:333-336: the
    key-padding mask exists only when batch > 1; all durations are equal here, so it keeps every key)."""
    import os
    import sys
    from f5test import ROOT
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    cfg = F5TTS_335M
    w = synthetic_weights(cfg, seed=42)
    B = 32
    cond, text, y0, _ = bench.synth_batch(B, 0, DEV)
    kw = dict(duration=bench.N_FRAMES, steps=bench.ODE_POINTS, method="euler", cfg_strength=2.0, sway_sampling_coef=-1.0)
    m16 = _model(cfg, w, "f16")
    out32, _ = F5TTS(transformer=m16).sample(cond, text, y0=y0, **kw)
    torch.cuda.synchronize()
    out32 = out32.cpu()
    assert torch.isfinite(out32).all()
    del m16
    torch.cuda.empty_cache()
    mx3 = _model(cfg, w, "bf16x3")
    f5x3 = F5TTS(transformer=mx3)
    worst = 0.0
    for i in range(B):
        ref, _ = f5x3.sample(cond[i:i + 1].contiguous(), text[i:i + 1].contiguous(), y0=y0[i:i + 1].contiguous(), **kw)
        l1 = float((out32[i] - ref[0].cpu()).abs().mean())
        worst = max(worst, l1)
        assert l1 <= MEL_L1_TOL, (i, l1)
        if i == 0:
            g = np.load(os.path.join(ROOT, "tests", "golden", "full_b1_euler32.npz"))
            l1g = float((ref[0].cpu() - torch.from_numpy(g["out"])).abs().mean())
            print(f"[b32 parity] batch-1 bf16x3 reference of utterance 0 vs the fp32 oracle golden: {l1g:.3e}")
            assert l1g <= 1e-4
    print(f"[b32 parity] worst utterance of 32: mel L1 {worst:.3e} (gate {MEL_L1_TOL})")


def _golden_module(name):
    import importlib.util
    import os
    from f5test import ROOT
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tests", "golden", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def full_f16():
    return _model(F5TTS_335M, synthetic_weights(F5TTS_335M, seed=42), "f16")


@pytest.mark.parametrize("B", [2, 4, 8, 16])
def test_fullsize_mid_batches_every_utterance_vs_oracle(full_f16, B):
    """batch 2 ... 16 at the full 335M size run the role-split 128 x 256 GEMM IN SEVERAL ROUNDS with the RESID_GATE /
    GELU / QKV epilogues (the `mid` rule of gemm.hip launch_epi, commit 943a2ee) -- the shipped default path of those batch sizes.
    Here: `f16`, automatic dispatch, N = 937, 5-point Euler + sway + CFG (8 DiT forwards), EVERY utterance against the fp32 CPU
    oracle's answer for the same utterance (tests/golden/full_b16_euler5.npz, made by make_batch_golden.py: bench.py's utterances
    0..15; at equal durations the oracle's batch elements do not interact, so its first B rows are the batch-B answer)."""
    import os
    from f5test import ROOT
    import bench
    mg = _golden_module("make_batch_golden")
    g = np.load(os.path.join(ROOT, "tests", "golden", "full_b16_euler5.npz"))
    cond, text, y0, _ = bench.synth_batch(B, 0, DEV)
    out, _ = F5TTS(transformer=full_f16).sample(cond, text, duration=mg.N_FRAMES, y0=y0, use_graph=False, **mg.KW)
    torch.cuda.synchronize()
    out = out.cpu()
    assert torch.isfinite(out).all()
    l1 = [float((out[i] - torch.from_numpy(g["out"][i])).abs().mean()) for i in range(B)]
    print(f"[mid batch] B={B} f16 vs fp32 oracle, mel L1 per utterance: worst {max(l1):.3e} mean {np.mean(l1):.3e}")
    assert max(l1) <= MEL_L1_TOL, l1
    # the captured graph replays the same launches: bit-identical
    out2, _ = F5TTS(transformer=full_f16).sample(cond, text, duration=mg.N_FRAMES, y0=y0, use_graph=True, **mg.KW)
    torch.cuda.synchronize()
    assert torch.equal(out2.cpu(), out)


def test_batch32_every_utterance_vs_oracle_golden(full_f16):
    """VERDICT r5 weak #2 / "next" #3(a): BASELINE configs[2] -- the batch the roofline is quoted on -- anchored to the ORACLE for every
    utterance, not to the engine's own batch-1 result: the default batch-32 call (LN fold on: 59 968 rows, 256 x 256 kernels with the
    folded epilogues), N = 937, 5-point Euler + sway + CFG, each of the 32 bench utterances against the fp32 CPU oracle's answer
    (tests/golden/full_b32_euler5.npz, make_batch_golden.py --which b32).  Then the same call replayed from its hipGraph, bitwise."""
    import os
    from f5test import ROOT
    import bench
    mg = _golden_module("make_batch_golden")
    g = np.load(os.path.join(ROOT, "tests", "golden", "full_b32_euler5.npz"))["out"]
    assert g.shape == (32, mg.N_FRAMES, 100)
    cond, text, y0, _ = bench.synth_batch(32, 0, DEV)
    eng = full_f16.engine
    assert eng.get_option("ln_fold") == -1
    f5 = F5TTS(transformer=full_f16)
    out, _ = f5.sample(cond, text, duration=mg.N_FRAMES, y0=y0, use_graph=False, **mg.KW)
    torch.cuda.synchronize()
    assert eng.range_events == 0 and eng.saturation_events == 0
    out = out.cpu()
    assert torch.isfinite(out).all()
    l1 = [float((out[i] - torch.from_numpy(g[i])).abs().mean()) for i in range(32)]
    print(f"[batch 32 vs oracle] f16, LN fold on, mel L1 per utterance vs the fp32 oracle: worst {max(l1):.3e} (utterance {int(np.argmax(l1))}) "
          f"mean {np.mean(l1):.3e} best {min(l1):.3e}")
    assert max(l1) <= MEL_L1_TOL, l1
    out2, _ = f5.sample(cond, text, duration=mg.N_FRAMES, y0=y0, use_graph=True, **mg.KW)
    torch.cuda.synchronize()
    assert torch.equal(out2.cpu(), out)
    # ... and as two half batches of 16 on two streams (Engine.split_batch, opt-in): the first call captures the halves one after the
    # other, the second replays them from two host threads at once -- the same bits (both halves are above the LN-fold threshold)
    eng.split_batch = 32
    try:
        for _ in range(2):
            out3, _ = f5.sample(cond, text, duration=mg.N_FRAMES, y0=y0, use_graph=True, **mg.KW)
            torch.cuda.synchronize()
            assert torch.equal(out3.cpu(), out)
        assert eng.split_events == 2
    finally:
        eng.split_batch = 0


@pytest.mark.parametrize("B", [4, 8, 16])
def test_fullsize_ln_fold_every_utterance_vs_oracle(full_f16, B):
    """LN fold (engine option "ln_fold", csrc/gemm.hpp fold_*): 44 of the 46 LN-modulate launches of a forward folded into the epilogues
    of the GEMMs around them.  B = 4: all four block GEMMs on the role-split 128 x 256 kernel; B = 8 / 16: QKV (and FF1) on the 256 x 256
    kernel, the residual GEMMs role-split.  Same gate and same oracle answers as the test above; the result must differ from the unfolded
    path (it is a different rounding sequence -- identical bits would mean the option did nothing) and replay bit-identically from a graph."""
    import os
    from f5test import ROOT
    import bench
    mg = _golden_module("make_batch_golden")
    g = np.load(os.path.join(ROOT, "tests", "golden", "full_b16_euler5.npz"))
    cond, text, y0, _ = bench.synth_batch(B, 0, DEV)
    f5 = F5TTS(transformer=full_f16)
    try:
        full_f16.engine.set_option("ln_fold", 0)
        base, _ = f5.sample(cond, text, duration=mg.N_FRAMES, y0=y0, use_graph=False, **mg.KW)
        full_f16.engine.set_option("ln_fold", 1)
        out, _ = f5.sample(cond, text, duration=mg.N_FRAMES, y0=y0, use_graph=False, **mg.KW)
        out2, _ = f5.sample(cond, text, duration=mg.N_FRAMES, y0=y0, use_graph=True, **mg.KW)
        full_f16.engine.set_option("ln_fold", -1)
        auto, _ = f5.sample(cond, text, duration=mg.N_FRAMES, y0=y0, use_graph=False, **mg.KW)
        torch.cuda.synchronize()
    finally:
        full_f16.engine.set_option("ln_fold", -1)
    # the default (-1) folds from 22 000 rows on (2 branches x B x 937 frames: batch >= 12)
    assert torch.equal(auto.cpu(), (out if 2 * B * mg.N_FRAMES >= 22000 else base).cpu())
    out, base = out.cpu(), base.cpu()
    assert torch.isfinite(out).all() and torch.equal(out2.cpu(), out)
    l1 = [float((out[i] - torch.from_numpy(g["out"][i])).abs().mean()) for i in range(B)]
    l1b = [float((base[i] - torch.from_numpy(g["out"][i])).abs().mean()) for i in range(B)]
    print(f"[ln_fold] B={B} f16 vs fp32 oracle, mel L1 per utterance: worst {max(l1):.3e} mean {np.mean(l1):.3e} "
          f"(unfolded: worst {max(l1b):.3e} mean {np.mean(l1b):.3e}); folded vs unfolded {float((out - base).abs().mean()):.3e}")
    assert max(l1) <= MEL_L1_TOL, l1
    assert not torch.equal(out, base)


def test_ln_fold_batch1_route_vs_oracle_golden(full_f16):
    """Round 6 (VERDICT r5 "next" #6): the LN fold at BATCH 1 -- every block GEMM is one round of workgroups; the residual GEMMs (64 x 128
    split-K ring kernel) write the folded operand + slice statistics, FF1 (8-wave ring kernel) and QKV (one round of role-split tiles)
    merge the statistics themselves, no row-factor launch: 44 of the 46 LN-modulate launches of a forward disappear.  Opt-in (ln_fold = 1;
    measured -0.5 ... -1.5 %, below the bar for a default).  The benchmark call itself: 335M, N = 937, 32-point Euler, against the fp32
    oracle's golden like the unfolded path (gate 1e-3, trajectory points too), graph == eager bitwise, and the default (-1) still unfolded."""
    import os
    from f5test import ROOT
    import bench
    g = np.load(os.path.join(ROOT, "tests", "golden", "full_b1_euler32.npz"))
    eng = full_f16.engine
    f5 = F5TTS(transformer=full_f16)
    cond, text, y0, _ = bench.synth_batch(1, 0, DEV)
    kw = dict(duration=937, steps=32, method="euler", cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0)
    base, _ = f5.sample(cond, text, use_graph=False, **kw)
    try:
        eng.set_option("ln_fold", 1)
        out, traj = f5.sample(cond, text, use_graph=False, **kw)
        out2, _ = f5.sample(cond, text, use_graph=True, **kw)
        torch.cuda.synchronize()
        # no-CFG call: one branch = 937 rows, the single-round kernels do not apply -> ln_fold = 1 must fail loudly, not fall back silently
        with pytest.raises(RuntimeError, match="ln_fold = 1"):
            f5.sample(cond, text, use_graph=False, **dict(kw, cfg_strength=0.0))
    finally:
        eng.set_option("ln_fold", -1)
    auto, _ = f5.sample(cond, text, use_graph=False, **kw)
    torch.cuda.synchronize()
    assert torch.equal(auto, base), "the automatic mode must not pick the batch-1 route"
    assert torch.equal(out2, out) and not torch.equal(out, base)
    l1 = float((out.cpu()[0].double() - torch.from_numpy(g["out"]).double()).abs().mean())
    l1b = float((base.cpu()[0].double() - torch.from_numpy(g["out"]).double()).abs().mean())
    print(f"[ln_fold batch 1] f16 vs fp32 oracle over 62 forwards: folded {l1:.3e}, unfolded {l1b:.3e}; folded vs unfolded {float((out - base).abs().mean()):.3e}")
    assert l1 <= MEL_L1_TOL and eng.range_events == 0 and eng.saturation_events == 0


def test_ln_fold_single_forward_and_single_branch(full_f16):
    """ln_fold through the other two entry shapes: f5_dit_forward (one evaluation, its constants computed for nfe = 1) at batch 12, where
    the default (-1) folds, and sample() WITHOUT classifier-free guidance (one branch: 24 x 937 = 22 488 rows) -- each against the same
    call with the fold off: close (a different rounding sequence), not identical."""
    import bench
    mg = _golden_module("make_batch_golden")
    eng = full_f16.engine
    cond, text, y0, _ = bench.synth_batch(12, 0, DEV)
    lens = [int(cond.shape[1])] * 12
    durs = [mg.N_FRAMES] * 12
    condp = torch.nn.functional.pad(cond, (0, 0, 0, mg.N_FRAMES - cond.shape[1]))
    res = {}
    try:
        for opt in (0, -1):
            eng.set_option("ln_fold", opt)
            pred, null = eng.dit_forward(y0, text, condp, lens, durs, 0.37, cfg_strength=2.0)
            torch.cuda.synchronize()
            res[opt] = (pred.cpu(), null.cpu())
        for k in (0, 1):
            a, b = res[0][k], res[-1][k]
            rel = float((a - b).abs().mean() / a.abs().mean())
            print(f"[ln_fold dit_forward] branch {k}: mean |d| / mean |v| = {rel:.3e}")
            assert torch.isfinite(b).all() and rel <= 2e-3 and not torch.equal(a, b)
        cond24, text24, y24, _ = bench.synth_batch(24, 0, DEV)
        kw = dict(mg.KW)
        kw["cfg_strength"] = 0.0
        outs = {}
        for opt in (0, -1):
            eng.set_option("ln_fold", opt)
            out, _ = F5TTS(transformer=full_f16).sample(cond24, text24, duration=mg.N_FRAMES, y0=y24, use_graph=False, **kw)
            torch.cuda.synchronize()
            outs[opt] = out.cpu()
        d = float((outs[0] - outs[-1]).abs().mean())
        print(f"[ln_fold no-CFG batch 24] mel L1 folded vs unfolded = {d:.3e}")
        assert torch.isfinite(outs[-1]).all() and 0.0 < d <= MEL_L1_TOL
    finally:
        eng.set_option("ln_fold", -1)


def test_ln_fold_ragged_batch_and_where_it_cannot_run(full_f16):
    """ln_fold on the RAGGED full-size batch (masked residual rows keep x: their x16 / row sums must still be written), and the option's
    three values: 1 fails loudly where no fold-capable kernels run (batch 2 of 937 frames: neither the staged kernels nor the single-round
    route of batch 1), -1 (the default) keeps the LN kernels there and at batch 1, 0 = never."""
    import os
    from f5test import ROOT
    import bench
    from f5_tts_mlx_amd.audio import log_mel_spectrogram
    mg = _golden_module("make_batch_golden")
    g = np.load(os.path.join(ROOT, "tests", "golden", "full_b8_ragged_euler5.npz"))
    waves, text, y0, dur = mg.ragged_inputs()
    cond = log_mel_spectrogram(torch.from_numpy(waves).to(DEV))
    f5 = F5TTS(transformer=full_f16)
    eng = full_f16.engine
    assert eng.get_option("ln_fold") == -1
    eng.set_option("ln_fold", 1)
    try:
        out, _ = f5.sample(cond, torch.from_numpy(text), duration=torch.from_numpy(dur), y0=torch.from_numpy(y0), use_graph=False, **mg.KW)
        torch.cuda.synchronize()
        out, ref = out.cpu(), torch.from_numpy(g["out"])
        l1v = [float((out[i, :d] - ref[i, :d]).abs().mean()) for i, d in enumerate(dur.tolist())]
        print(f"[ln_fold ragged] mel L1 on the valid frames: worst {max(l1v):.3e}")
        assert torch.isfinite(out).all() and max(l1v) <= MEL_L1_TOL, l1v
        c2, t2, y2, _ = bench.synth_batch(2, 0, DEV)
        with pytest.raises(RuntimeError, match="ln_fold"):
            f5.sample(c2, t2, duration=mg.N_FRAMES, y0=y2, use_graph=False, **mg.KW)
        c1, t1, y1, _ = bench.synth_batch(1, 0, DEV)
        eng.set_option("ln_fold", -1)
        assert eng.get_option("ln_fold") == -1
        a, _ = f5.sample(c1, t1, duration=mg.N_FRAMES, y0=y1, use_graph=False, **mg.KW)
        eng.set_option("ln_fold", 0)
        b, _ = f5.sample(c1, t1, duration=mg.N_FRAMES, y0=y1, use_graph=False, **mg.KW)
        torch.cuda.synchronize()
        assert torch.equal(a, b)
    finally:
        eng.set_option("ln_fold", -1)


def test_fullsize_ragged_batch_vs_oracle(full_f16):
    """a RAGGED full-size batch (B = 8, durations 937 ... 500, text padded by a different amount per utterance):
    the key mask with real kv_len in the attention kernels, attention-output rows zeroed at padded positions through `rowkeep` in the
    residual epilogue, GRN / conv-pos over the padded length (cfm.py:317-336, dit.py:160-173) -- against the fp32 oracle's BATCHED
    answer (full_b8_ragged_euler5.npz).  The gate is asserted on each utterance's valid frames; the padded frames (which the
    reference lets evolve through the MLP path and returns untrimmed) are compared too."""
    import os
    from f5test import ROOT
    from f5_tts_mlx_amd.audio import log_mel_spectrogram
    mg = _golden_module("make_batch_golden")
    g = np.load(os.path.join(ROOT, "tests", "golden", "full_b8_ragged_euler5.npz"))
    waves, text, y0, dur = mg.ragged_inputs()
    cond = log_mel_spectrogram(torch.from_numpy(waves).to(DEV))
    f5 = F5TTS(transformer=full_f16)
    for use_graph in (False, True):
        out, _ = f5.sample(cond, torch.from_numpy(text), duration=torch.from_numpy(dur), y0=torch.from_numpy(y0), use_graph=use_graph, **mg.KW)
        torch.cuda.synchronize()
        out = out.cpu()
        assert torch.isfinite(out).all() and tuple(out.shape) == tuple(g["out"].shape)
        ref = torch.from_numpy(g["out"])
        l1v = [float((out[i, :d] - ref[i, :d]).abs().mean()) for i, d in enumerate(dur.tolist())]
        l1p = [float((out[i, d:] - ref[i, d:]).abs().mean()) for i, d in enumerate(dur.tolist()) if d < out.shape[1]]
        print(f"[ragged full size] graph={use_graph} f16 vs fp32 oracle: valid frames worst {max(l1v):.3e}, padded frames worst {max(l1p):.3e}")
        assert max(l1v) <= MEL_L1_TOL, l1v
        assert max(l1p) <= 5e-3, l1p
    # the same batch as two shards padded to the global maximum (pad_to): rows equal the unsharded call
    o_a, _ = f5.sample(cond[:1], torch.from_numpy(text[:1]), duration=torch.from_numpy(dur[:1]), y0=torch.from_numpy(y0[:1]), use_graph=False, **mg.KW)
    o_b, _ = f5.sample(cond[5:], torch.from_numpy(text[5:]), duration=torch.from_numpy(dur[5:]), y0=torch.from_numpy(y0[5:]).contiguous(),
                       use_graph=False, pad_to=mg.N_FRAMES, **mg.KW)
    torch.cuda.synchronize()
    d0 = float((o_a.cpu()[0] - out[0]).abs().max())
    d5 = max(float((o_b.cpu()[k, :dur[5 + k]] - out[5 + k, :dur[5 + k]]).abs().max()) for k in range(3))
    print(f"[ragged full size] shard of 1 (longest, no pad_to needed) vs batch row: {d0:.3e}; shard of 3 with pad_to vs batch rows: {d5:.3e}")
    assert d5 <= 2e-2 and d0 <= 2e-2      # other tile shapes at M = 2 x 937 / 6 x 937: f16 rounding-level, not mask-level, differences


def test_f16_range_stress_outlier_weights():
    """The hazards of IEEE-half operands, tested instead of argued: trained DiTs carry activation outliers the seeded-random weights do
    not, so a few adaLN scale rows, FF1 rows and q / k rows of the 335M weights are scaled by 1e2 ... 1e3 (LN-modulated activations and
    FF hidden values of 1e3 ... 1e5, attention logits hundreds of times larger).  Batch 1: no LN fold, so until round 5 nothing looked.
    Three engines per scale: `bf16x3` (the fp32-class reference), `f16` RAW (range_check off: what the kernels do when nobody looks) and
    `f16` as a checkpoint is loaded (range_check "sync" + one cross-checked call, the from_pretrained default).
    What round 6 found: the raw path stays finite and inside the gate up to 100x; at 300x it is 3.4e-2 off and at 1000x 0.5 -- and the
    RANGE detector stays silent at both (saturation events 0): no operand reaches +-65 504, it is fp16's 11-bit SIGNIFICAND on logits
    that large (round 3-5 read this as saturation).  The cross-check catches it: the call warns and returns the bf16x3 result, the
    engine stays on bf16x3.  A genuinely saturating checkpoint is test_f16_saturation_detector_batch1 below."""
    cfg = F5TTS_335M
    base = synthetic_weights(cfg, seed=42)
    cond, text, durations, y0 = synth_inputs(cfg, 1, 400, nt=64, n_ref=120, seed=5)
    r = np.random.default_rng(11)
    results = {}
    kw = dict(duration=400, y0=y0, steps=6, method="euler", cfg_strength=2.0)
    for scale in (1.0, 1e2, 3e2, 1e3):
        w = _outlier_weights(base, scale, r)
        outs = {}
        stats = {}
        for prec, check in (("bf16x3", "off"), ("f16", "off"), ("f16", "sync")):
            m = _model(cfg, w, prec)
            m.engine.range_check = check
            if check == "sync":
                m.engine.verify_calls = 1
            with warnings.catch_warnings(record=True) as rec:
                warnings.simplefilter("always")
                out, _ = F5TTS(transformer=m).sample(cond, text, **kw)
                torch.cuda.synchronize()
                if check == "sync":
                    out2, _ = F5TTS(transformer=m).sample(cond, text, **kw)       # the call after the cross-check
                    torch.cuda.synchronize()
                    outs["second"] = out2.cpu()
            outs[(prec, check)] = out.cpu()
            if check == "sync":
                e = m.engine
                stats = dict(sat=e.saturation_events, ver=e.verify_events, l1=e.last_verify_l1, on_fallback=e._use_fallback,
                             warned=any(issubclass(x.category, E.OperandRangeWarning) for x in rec))
                assert stats["warned"] == (stats["sat"] + stats["ver"] > 0), stats
            del m
            torch.cuda.empty_cache()
        ref = outs[("bf16x3", "off")]
        finite = bool(all(torch.isfinite(o).all() for o in outs.values()))
        l1 = float((outs[("f16", "off")] - ref).abs().mean())                 # the raw fp16 path
        l1_checked = float((outs[("f16", "sync")] - ref).abs().mean())        # detector + cross-check
        l1_second = float((outs["second"] - ref).abs().mean())
        results[scale] = (finite, l1, l1_checked, l1_second, stats)
        print(f"[f16 range] outlier scale {scale:g}: finite {finite}, raw f16 vs bf16x3 mel L1 {l1:.3e}; as loaded (detector + one cross-checked "
              f"call) {l1_checked:.3e}, next call {l1_second:.3e}; saturation events {stats['sat']}, failed cross-checks {stats['ver']} "
              f"(measured {stats['l1']:.3e}), on bf16x3 afterwards: {stats['on_fallback']}")
    assert all(v[0] for v in results.values()), results                      # never inf / nan
    assert results[1.0][1] <= MEL_L1_TOL and results[1e2][1] <= MEL_L1_TOL, results
    assert results[1.0][4]["sat"] == 0 and results[1.0][4]["ver"] == 0 and not results[1.0][4]["on_fallback"], "clean weights must stay on fp16"
    assert results[1e3][1] > MEL_L1_TOL, "the stress no longer breaks raw fp16: the test lost its subject"
    for sc, (_, l1, l1c, l1s, st) in results.items():
        assert l1c <= MEL_L1_TOL and l1s <= MEL_L1_TOL, (sc, l1c, l1s)       # what the caller gets is inside the gate at EVERY scale
        if l1 > MEL_L1_TOL:                                                   # ... because the broken scales were caught and moved to bf16x3
            assert st["warned"] and st["on_fallback"] and l1c <= 1e-6 and l1s <= 1e-6, (sc, st)
    print("[f16 range] breaking scale of the RAW path (first scale whose unchecked f16 result leaves the gate): "
          f"{next((sc for sc in sorted(results) if results[sc][1] > MEL_L1_TOL), None)}")


def test_f16_saturation_detector_batch1():
    """VERDICT r5 "next" #2: a checkpoint that really leaves the fp16 RANGE, at BATCH 1 (no LN fold: round 5 had no check of any kind
    there).  Four FF1 rows of three blocks scaled 3e4 (FF hidden values ~1e6) and four adaLN scale rows scaled 1e5 (LN-modulated
    activations ~1e6): the packers clamp at +-65 504, the result is finite and wrong.  range_check "off": silently so.  Default
    ("sync"): status bit 2 comes back, the engine warns, builds the bf16x3 engine from its host weights and re-runs THAT call there --
    the caller gets the bf16x3 result bit for bit, also from f5_dit_forward, and later calls go straight to bf16x3.  "async": reported
    at synchronize(), right from the next call on."""
    cfg = F5TTS_335M
    w = synthetic_weights(cfg, seed=42)
    r = np.random.default_rng(5)
    for blk in (0, 9, 21):
        pre = f"transformer.transformer_blocks.{blk}."
        ff1 = w[pre + "ff.ff.layers.0.layers.0.weight"]
        for row in r.integers(0, ff1.shape[0], 4):
            ff1[row] *= 3.0e4
        ada = w[pre + "attn_norm.linear.weight"]
        for row in r.integers(1024, 2048, 4):
            ada[row] *= 1.0e5
    cond, text, durations, y0 = synth_inputs(cfg, 1, 400, nt=64, n_ref=120, seed=6)
    kw = dict(duration=400, y0=y0, steps=4, method="euler", cfg_strength=2.0)
    ref_m = _model(cfg, w, "bf16x3")
    ref, _ = F5TTS(transformer=ref_m).sample(cond, text, **kw)
    torch.cuda.synchronize()
    ref = ref.cpu()
    del ref_m
    torch.cuda.empty_cache()
    assert torch.isfinite(ref).all()
    outs = {}
    for check in ("off", "sync", "async"):
        m = _model(cfg, w, "f16")
        eng = m.engine
        eng.range_check = check
        f5 = F5TTS(transformer=m)
        if check == "off":
            with warnings.catch_warnings():
                warnings.simplefilter("error", E.OperandRangeWarning)
                out, _ = f5.sample(cond, text, **kw)
                eng.synchronize()
            assert eng.saturation_events == 0
        elif check == "sync":
            with pytest.warns(E.OperandRangeWarning, match="re-running it in bf16x3"):
                out, _ = f5.sample(cond, text, **kw)
            torch.cuda.synchronize()
            assert eng.saturation_events == 1 and eng._use_fallback
            with warnings.catch_warnings():
                warnings.simplefilter("error", E.OperandRangeWarning)      # from now on: bf16x3 directly, nothing to report
                out2, _ = f5.sample(cond, text, **kw)
                torch.cuda.synchronize()
            assert torch.equal(out2.cpu(), out.cpu())
        else:
            with warnings.catch_warnings():
                warnings.simplefilter("error", E.OperandRangeWarning)
                out, _ = f5.sample(cond, text, **kw)                           # returns without looking
            with pytest.warns(E.OperandRangeWarning, match="saturated"):
                eng.synchronize()
            assert eng.saturation_events == 1 and eng._use_fallback
            out, _ = f5.sample(cond, text, **kw)                               # the next call is right
            torch.cuda.synchronize()
        outs[check] = out.cpu()
        del m, f5, eng
        torch.cuda.empty_cache()
    raw = float((outs["off"] - ref).abs().mean())
    print(f"[f16 saturation, batch 1] raw fp16 (clamped, nobody looks) vs bf16x3: mel L1 {raw:.3e}; default engine: "
          f"{float((outs['sync'] - ref).abs().mean()):.3e}; async, call after the report: {float((outs['async'] - ref).abs().mean()):.3e}")
    assert torch.isfinite(outs["off"]).all() and raw > MEL_L1_TOL, "the weights no longer saturate fp16: the test lost its subject"
    assert torch.equal(outs["sync"], ref) and torch.equal(outs["async"], ref)


def _outlier_weights(base, scale, r):
    """a few adaLN scale rows, FF1 rows and q / k rows of blocks 0 / 7 / 21 scaled up (test_f16_range_stress_outlier_weights)"""
    w = {k: v.copy() for k, v in base.items()}
    if scale != 1.0:
        for blk in (0, 7, 21):
            pre = f"transformer.transformer_blocks.{blk}."
            ada = w[pre + "attn_norm.linear.weight"]
            for row in r.integers(1024, 2048, 4):
                ada[row] *= scale
            for row in r.integers(4096, 5120, 4):
                ada[row] *= scale
            ff1 = w[pre + "ff.ff.layers.0.layers.0.weight"]
            for row in r.integers(0, ff1.shape[0], 4):
                ff1[row] *= scale
            for name in ("attn.to_q.weight", "attn.to_k.weight"):
                m = w[pre + name]
                for row in r.integers(0, m.shape[0], 2):
                    m[row] *= np.sqrt(scale)
    return w


def test_f16_range_stress_with_the_ln_fold_active():
    """VERDICT r4 "next" #2(a), model level: the outlier-weight stress of the test above at batch 12 x 937 frames, where the LN fold is
    ACTIVE by default (22 488 rows), against `bf16x3` (fp32-class, never folds), with the fold off next to it: the folded path must stay
    finite, inside the gate wherever the unfolded f16 path is, and no further from `bf16x3` than 1.5x the unfolded path + 1e-4.  Also
    carries a residual-stream offset: the input-projection bias is raised by 30, which puts the row means of the residual stream at
    ~11 sigma in every block (checked with the oracle: 3.1 sigma for +8, proportional) -- what the round-4 formulation lost a factor
    ~10 of operand precision on (profiles/r05/ln_fold_numerics_study.jsonl)."""
    import bench
    cfg = F5TTS_335M
    base = synthetic_weights(cfg, seed=42)
    base["transformer.input_embed.proj.bias"] = base["transformer.input_embed.proj.bias"] + 30.0     # row means of ~11 sigma in every block
    cond, text, y0, _ = bench.synth_batch(12, 0, DEV)
    r = np.random.default_rng(11)
    N = 937
    for scale in (1.0, 1e2):
        w = _outlier_weights(base, scale, r)
        outs = {}
        for prec, fold in (("bf16x3", -1), ("f16", 0), ("f16", -1)):
            m = _model(cfg, w, prec)
            m.engine.set_option("ln_fold", fold)
            with warnings.catch_warnings():
                warnings.simplefilter("error", E.OperandRangeWarning)           # these weights must not leave the fp16 range
                out, _ = F5TTS(transformer=m).sample(cond, text, duration=N, y0=y0, steps=6, method="euler", cfg_strength=2.0, use_graph=False)
                m.engine.synchronize()
            outs[(prec, fold)] = out.cpu()
            if prec == "f16" and fold == -1:
                assert m.engine.range_events == 0
            del m
            torch.cuda.empty_cache()
        ref = outs[("bf16x3", -1)]
        l1_unf = float((outs[("f16", 0)] - ref).abs().mean())
        l1_fold = float((outs[("f16", -1)] - ref).abs().mean())
        print(f"[f16 range, fold active] outlier scale {scale:g}: f16 vs bf16x3 mel L1 unfolded {l1_unf:.3e}, folded {l1_fold:.3e}")
        assert torch.isfinite(outs[("f16", -1)]).all()
        assert not torch.equal(outs[("f16", -1)], outs[("f16", 0)]), "the fold did not run"
        assert l1_fold <= 1.5 * l1_unf + 1e-4, (l1_fold, l1_unf)
        if l1_unf <= MEL_L1_TOL * 0.6:
            assert l1_fold <= MEL_L1_TOL, (l1_fold, l1_unf)


@pytest.mark.parametrize("mode", ["sync", "async", "auto"])
def test_ln_fold_operand_overflow_falls_back_instead_of_raising(mode):
    """VERDICT r4 "next" #2(c, d) / ADVICE: when the folded operand leaves the fp16 range the engine must not raise (round 4 did: a call
    that worked at batch 8 threw at 12) and must not block the host on every call.  Weights whose residual stream reaches ~1e6 after
    block 0 (FF2 scaled): `sync` / `auto` return the UNFOLDED result for that very call (bit-identical to ln_fold = 0) with an
    OperandRangeWarning and leave the engine at ln_fold = 0; `async` returns at once, reports at the next call (or synchronize()) and
    is right from then on.  Ordinary weights: no warning, `auto` stops synchronising after its probation."""
    import bench
    cfg = F5TTS_335M
    w = synthetic_weights(cfg, seed=42)
    w["transformer.transformer_blocks.0.ff.ff.layers.2.weight"] = w["transformer.transformer_blocks.0.ff.ff.layers.2.weight"] * 3.0e6
    cond, text, y0, _ = bench.synth_batch(4, 0, DEV)
    kw = dict(duration=937, y0=y0, steps=3, method="euler", cfg_strength=2.0, use_graph=False)
    ref_m = DiT.from_config(cfg, precision="f16", device=DEV)
    ref_m.load_weights(w)
    ref_m.engine.set_option("ln_fold", 0)
    ref, _ = F5TTS(transformer=ref_m).sample(cond, text, **kw)
    torch.cuda.synchronize()
    ref = ref.cpu()
    del ref_m
    m = DiT.from_config(cfg, precision="f16", device=DEV)
    m.load_weights(w)
    eng = m.engine
    eng.range_check = mode
    eng.set_option("ln_fold", 1)
    f5 = F5TTS(transformer=m)
    if mode in ("sync", "auto"):
        with pytest.warns(E.OperandRangeWarning, match="re-running it unfolded"):
            out, _ = f5.sample(cond, text, **kw)
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), ref) and eng.get_option("ln_fold") == 0 and eng.range_events == 1
    else:
        with warnings.catch_warnings():
            warnings.simplefilter("error", E.OperandRangeWarning)
            out, _ = f5.sample(cond, text, **kw)                               # returns without looking
        with pytest.warns(E.OperandRangeWarning, match="saturated"):
            eng.synchronize()
        assert eng.get_option("ln_fold") == 0 and eng.range_events == 1
        assert torch.isfinite(out).all()                                      # saturated, not inf
    with warnings.catch_warnings():
        warnings.simplefilter("error", E.OperandRangeWarning)
        out2, _ = f5.sample(cond, text, **kw)
        eng.synchronize()
    assert torch.equal(out2.cpu(), ref)
    del m
    torch.cuda.empty_cache()


def test_auto_range_check_stops_synchronising(full_f16):
    """range_check = "auto" (opt-in since round 6; the default is "sync"): synchronous (with fall-back) until three calls of the SAME
    shape in a row came back clean, a pinned 4-byte copy behind the call from then on; a new shape starts a new probation (ADVICE r5:
    whether an operand overflows depends on the input, not only on the weights).  Status slots come from a free list: a synchronous
    read can never be handed the pinned slot of a pending asynchronous one."""
    import bench
    eng = full_f16.engine
    assert eng.range_check == "sync"                                           # the default looks at every call
    saved = eng.range_check
    eng.range_check = "auto"
    f5 = F5TTS(transformer=full_f16)
    eng._clean_shape, eng._clean_calls = None, 0
    c1, t1, y1, _ = bench.synth_batch(1, 0, DEV)
    cond, text, y0, _ = bench.synth_batch(4, 0, DEV)
    try:
        f5.sample(c1, t1, duration=937, y0=y1, steps=3, use_graph=False)
        assert eng._clean_calls == 1 and not eng._status_pending              # batch 1 is checked too (status bit 2: every 16-bit packer)
        for i in range(5):
            f5.sample(cond, text, duration=937, y0=y0, steps=3, use_graph=False)
            if i < 3:
                assert eng._clean_calls == i + 1 and not eng._status_pending    # probation of THIS shape: checked in the call
        assert eng._status_pending                                            # afterwards: pending, resolved later
        free_before = sorted(eng._status_free + [p[1] for p in eng._status_pending])
        f5.sample(c1, t1, duration=937, y0=y1, steps=3, use_graph=False)      # another shape: synchronous again, on a slot nobody is waiting on
        assert eng._clean_shape[0] == 1 and eng._clean_calls == 1
        eng.synchronize()
        assert not eng._status_pending and sorted(eng._status_free) == free_before == list(range(16))
        assert eng.range_events == 0 and eng.saturation_events == 0
    finally:
        eng.range_check = saved


def test_bf16_with_the_ln_fold(full_f16):
    """no model-level test covered bf16 with the fold.  Batch 12 x 937 (fold active by default), 5-point solve: bf16 folded
    vs bf16 unfolded must differ (the option ran) by no more than bf16's own drift from the f16 result on the same inputs."""
    import bench
    cfg = F5TTS_335M
    cond, text, y0, _ = bench.synth_batch(12, 0, DEV)
    kw = dict(duration=937, y0=y0, steps=5, method="euler", cfg_strength=2.0, sway_sampling_coef=-1.0, use_graph=False)
    m = _model(cfg, synthetic_weights(cfg, seed=42), "bf16")
    outs = {}
    for opt in (0, -1):
        m.engine.set_option("ln_fold", opt)
        out, _ = F5TTS(transformer=m).sample(cond, text, **kw)
        torch.cuda.synchronize()
        outs[opt] = out.cpu()
    f16, _ = F5TTS(transformer=full_f16).sample(cond, text, **kw)
    torch.cuda.synchronize()
    d_fold = float((outs[-1] - outs[0]).abs().mean())
    d_prec = float((outs[0] - f16.cpu()).abs().mean())
    print(f"[bf16 + ln_fold] folded vs unfolded {d_fold:.3e}; bf16 vs f16 (both unfolded / default) {d_prec:.3e}")
    assert torch.isfinite(outs[-1]).all() and 0.0 < d_fold <= 1.5 * d_prec
    del m
    torch.cuda.empty_cache()


def test_back_to_back_calls_without_host_sync_full_size():
    """The bench pattern: sample() called back to back with no host synchronisation in between (outputs of earlier calls being
    freed and their blocks re-used by torch's allocator meanwhile), graph and eager.  Every call must return the bits of a
    synchronised call.  (Round 2 found garbage here on some boxes while the sampling path still used hipMemsetAsync /
    hipMemcpyAsync: a captured memset of V^T could land after the first QKV epilogues; the path is kernels only since.)"""
    import os
    from f5test import ROOT
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_fullsize_golden", os.path.join(ROOT, "tests", "golden", "make_fullsize_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    wave, text, y0 = mg.inputs(0)
    from f5_tts_mlx_amd.audio import log_mel_spectrogram
    cond = log_mel_spectrogram(torch.from_numpy(wave).to(DEV))
    m = _model(F5TTS_335M, synthetic_weights(F5TTS_335M, seed=42), "f16")
    f5 = F5TTS(transformer=m)
    kw = dict(duration=mg.N_FRAMES, y0=torch.from_numpy(y0)[None].to(DEV), steps=12, method="euler", cfg_strength=2.0, sway_sampling_coef=-1.0)
    textd = torch.from_numpy(text)[None].to(DEV)
    ref, ref_traj = f5.sample(cond, textd, use_graph=False, **kw)
    torch.cuda.synchronize()
    ref, ref_traj = ref.clone(), ref_traj.clone()
    assert torch.isfinite(ref).all() and float(ref.abs().max()) < 50
    for mode in (True, False, "auto"):
        outs = []
        out = None
        for i in range(6):
            out, traj = f5.sample(cond, textd, use_graph=mode, **kw)        # previous `out` / `traj` are released here
            if i % 2:
                outs.append((out.clone(), traj[-1].clone()))
        torch.cuda.synchronize()
        assert torch.equal(out, ref), mode
        for o, t in outs:
            assert torch.equal(o, ref) and torch.equal(t, ref_traj[-1]), mode
    del m
    torch.cuda.empty_cache()


@pytest.mark.parametrize("drops", [(0.9, 0.9), (0.1, 0.9), (0.9, 0.1)])     # (keep, keep) / audio dropped / both dropped
def test_cfm_loss_forward_parity(tiny_weights, tiny_x3, drops):
    """F5TTS.__call__ (cfm.py:169-251), forward only, every random draw injected identically into engine and oracle;
    the span mask (int32 index math) must be bit-exact, the loss within fp32-class tolerance."""
    cfg, B, N = TINY, 2, 96
    g = torch.Generator().manual_seed(11)
    mel = torch.randn((B, N, cfg.mel_dim), generator=g) * 2.0 - 1.0
    text = torch.randint(0, cfg.text_num_embeds, (B, 20), generator=g, dtype=torch.int32)
    text[1, 15:] = -1
    lens = torch.tensor([N, 71], dtype=torch.int32)
    rand = dict(frac_lengths=torch.tensor([0.73, 0.91]), span_rand=torch.tensor([0.42, 0.08]),
                x0=torch.randn((B, N, cfg.mel_dim), generator=g), time=torch.tensor([0.31, 0.77]),
                rand_audio_drop=drops[0], rand_cond_drop=drops[1])
    orc = O.DiTOracle(cfg, tiny_weights)
    ref, aux = O.cfm_loss(orc, mel, text, lens=lens, return_aux=True, **rand)
    tts = F5TTS(tiny_x3)
    from f5_tts_mlx_amd.utils import mask_from_frac_lengths
    span = mask_from_frac_lengths(lens, rand["frac_lengths"], max_length=N, rand=rand["span_rand"]) & O.lens_to_mask(lens, N)
    assert torch.equal(span, aux["rand_span_mask"])
    got = float(tts(mel, text, lens=lens, rand=rand))
    print(f"cfm loss drops={drops}: engine {got:.6f} oracle {float(ref):.6f} (drop_audio={aux['drop_audio_cond']}, "
          f"drop_text={aux['drop_text']})")
    assert abs(got - float(ref)) <= 2e-4 * max(1.0, abs(float(ref)))
    # unseeded call: draws come from torch's generator, the value is finite and reproducible under the same generator
    a = float(tts(mel, text, lens=lens, generator=torch.Generator().manual_seed(5)))
    b = float(tts(mel, text, lens=lens, generator=torch.Generator().manual_seed(5)))
    assert np.isfinite(a) and a == b


def test_from_pretrained_local_checkpoints(tmp_path):
    """F5TTS.from_pretrained (cfm.py:404-520) on a local directory holding a synthetic 335M checkpoint: the full-precision
    file, an MLX-style 8-bit group-quantised file (expanded on load), vocab, duration predictor and a local Vocos.  The
    loaded models must reproduce a directly constructed model bit for bit."""
    from safetensors.numpy import save_file
    from f5_tts_mlx_amd.duration import synthetic_duration_weights
    from f5_tts_mlx_amd.vocos import synthetic_vocos_weights
    from f5_tts_mlx_amd.weights import dequantize_mlx_checkpoint, quantize_mlx_affine
    vocab_text = open(str(E.library_path().parent.parent / "assets" / "vocab.txt")).read()
    vocab = {v: i for i, v in enumerate(vocab_text.split("\n"))}
    import dataclasses
    cfg = dataclasses.replace(F5TTS_335M, text_num_embeds=len(vocab) - 1)
    w = {k: v.astype(np.float16) for k, v in synthetic_weights(cfg, seed=5).items()}      # fp16 on disk, like a release
    mdir, vdir = tmp_path / "model", tmp_path / "vocos"
    mdir.mkdir(); vdir.mkdir()
    (mdir / "vocab.txt").write_text(vocab_text)
    save_file(w, str(mdir / "model_v1.safetensors"))
    q = {}
    for k, v in w.items():                               # nn.quantize: every Linear whose input width is a multiple of 64
        if k.endswith(".weight") and v.ndim == 2 and v.shape[1] % 64 == 0 and "text_embed.text_embed" not in k:
            pk, sc, bi = quantize_mlx_affine(v.astype(np.float32), 8)
            q[k], q[k[:-7] + ".scales"], q[k[:-7] + ".biases"] = pk, sc.astype(np.float16), bi.astype(np.float16)
        else:
            q[k] = v
    assert "transformer.input_embed.proj.scales" not in q and "transformer.transformer_blocks.0.attn.to_q.scales" in q
    save_file(q, str(mdir / "model_v1_8b.safetensors"))
    save_file(synthetic_duration_weights(seed=11, text_num_embeds=len(vocab) - 1), str(mdir / "duration_v2.safetensors"))
    save_file(synthetic_vocos_weights(seed=7), str(vdir / "model.safetensors"))

    g = torch.Generator().manual_seed(3)
    wave = (torch.randn((1, 24000), generator=g) * 0.1)
    text = ["hello there"]

    def run(model):
        out, _ = model.sample(wave, text=text, duration=150, steps=2, method="euler", seed=1)
        torch.cuda.synchronize()
        return out.cpu()

    f5 = F5TTS.from_pretrained(str(mdir), convert_weights=False, vocoder_name_or_path=str(vdir), device=str(DEV))
    assert f5._duration_predictor is not None and f5._vocoder is not None
    got = run(f5)
    assert got.ndim == 1 and got.shape[0] == 256 * 149 and torch.isfinite(got).all()
    assert f5.transformer.precision == "f16"                                      # the package default is the parity-valid mode
    direct = DiT.from_config(cfg, precision="f16", device=DEV)
    direct.load_weights({k: v.astype(np.float32) for k, v in w.items()})
    ref = run(F5TTS(transformer=direct, vocab_char_map=vocab, vocoder=f5._vocoder))
    assert torch.equal(got, ref)
    assert int(f5.sample(wave, text=text, duration=None, steps=2, method="euler", seed=1)[0].shape[0]) > 0     # duration predictor path
    del f5, direct
    torch.cuda.empty_cache()

    f8 = F5TTS.from_pretrained(str(mdir), quantization_bits=8, vocoder_name_or_path=str(vdir), device=str(DEV))
    got8 = run(f8)
    deq = DiT.from_config(cfg, precision="f16", device=DEV)
    deq.load_weights({k: np.asarray(v, np.float32) for k, v in dequantize_mlx_checkpoint(q, 8).items()})
    ref8 = run(F5TTS(transformer=deq, vocab_char_map=vocab, vocoder=f8._vocoder))
    assert torch.equal(got8, ref8)
    rel = float((got8 - got).abs().mean() / got.abs().mean())
    print(f"8-bit checkpoint vs full precision: relative wave L1 {rel:.3e}")
    assert rel < 0.5
    # a vocoder that cannot be found raises like the reference's Vocos.from_pretrained (no network here) ...
    with pytest.raises(RuntimeError, match="vocoder"):
        F5TTS.from_pretrained(str(mdir), convert_weights=False, vocoder_name_or_path="no-such-org/no-such-vocos", device=str(DEV))
    # ... unless the caller opts out: the model loads and sample returns mel frames (documented difference)
    fm = F5TTS.from_pretrained(str(mdir), convert_weights=False, vocoder_name_or_path=None, device=str(DEV))
    mel, _ = fm.sample(wave, text=text, duration=150, steps=2, method="euler", seed=1)
    assert tuple(mel.shape) == (1, 150, 100)


def test_golden_cfm_loss_on_the_engine(tiny_weights, tiny_x3):
    """the committed golden loss values (tests/golden/tiny_cfm_loss.npz, fp64 oracle) reproduced by F5TTS.__call__ on the GPU"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_cfm_loss.npz"))
    assert int(g["weights_seed"]) == 42
    tts = F5TTS(tiny_x3)
    for name, (ra, rc) in dict(keep=(0.9, 0.9), drop_audio=(0.1, 0.9), drop_both=(0.9, 0.1)).items():
        rand = dict(x0=torch.from_numpy(g["x0"]), time=torch.from_numpy(g["time"]), frac_lengths=torch.from_numpy(g["frac_lengths"]),
                    span_rand=torch.from_numpy(g["span_rand"]), rand_audio_drop=ra, rand_cond_drop=rc)
        got = float(tts(torch.from_numpy(g["mel"]), torch.from_numpy(g["text"]), lens=torch.from_numpy(g["lens"]), rand=rand))
        want = float(g["loss_" + name])
        print(f"golden cfm loss [{name}]: engine {got:.6f} golden {want:.6f}")
        assert abs(got - want) <= 2e-4 * want


def test_real_checkpoint_tooling_on_a_synthetic_directory(tmp_path):
    """the real-weights entry points, runnable without a real checkpoint.  A synthetic 335M checkpoint directory
    with UPSTREAM key names (ema_model. prefix, PyTorch conv layout: what cfm.py:477-508 converts) goes through
    (1) tools/real_checkpoint_parity.py: oracle vs engine in f16 / bf16x3 on the fixture WAV + the per-producer operand maxima,
    (2) `bench.py --weights DIR` (= $F5_WEIGHTS): the bench line on that checkpoint."""
    import json
    import os
    import subprocess
    import sys
    from safetensors.numpy import save_file
    from f5test import ROOT
    vocab_text = open(str(E.library_path().parent.parent / "assets" / "vocab.txt")).read()
    import dataclasses
    cfg = dataclasses.replace(F5TTS_335M, text_num_embeds=len(vocab_text.split("\n")) - 1)
    w = synthetic_weights(cfg, seed=9)
    up = {}
    for k, v in w.items():                                  # MLX-layout names -> upstream names (the inverse of cfm.py:479-504)
        n = k.replace(".to_out.layers.", ".to_out.").replace(".text_blocks.layers.", ".text_blocks.").replace(".ff.ff.layers.0.layers.0", ".ff.ff.0.0")
        n = n.replace(".ff.ff.layers.2", ".ff.ff.2").replace(".time_mlp.layers.", ".time_mlp.").replace(".conv1d.layers.", ".conv1d.")
        if v.ndim == 3 and ("conv1d" in k or "dwconv" in k):
            v = np.ascontiguousarray(np.swapaxes(v, 1, 2))
        up["ema_model." + n] = v.astype(np.float32)
    mdir = tmp_path / "ckpt"
    mdir.mkdir()
    (mdir / "vocab.txt").write_text(vocab_text)
    save_file(up, str(mdir / "model_v1.safetensors"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import real_checkpoint_parity as RP
    wl, _ = RP.load_checkpoint(str(mdir))
    assert set(wl) >= {k for k in w if "inv_freq" not in k} and all(np.array_equal(wl[k], w[k]) for k in w if k in wl)
    res = RP.run(str(mdir), steps=3, seconds=1.0, precisions=("f16", "bf16x3"), batch=10)      # 2 x 10 x ~375 rows: the staged GEMM kernels
    print(json.dumps({k: v for k, v in res.items() if k != "operand_peaks"}), "\n", json.dumps(res["operand_peaks"], indent=0)[:1500])
    assert res["mel_l1"]["bf16x3"] <= 1e-4 and res["mel_l1"]["f16"] <= MEL_L1_TOL and all(res["finite"].values())
    # round 5: what the LN fold would meet on this checkpoint, and the fold forced on next to the fold off at batch 4
    st = res["layer_norm_statistics"]
    assert st["row_mean_over_sigma_max"] > 0 and st["mean_drift_over_sigma_max"] >= 0 and res["ln_fold_operand_peak_estimate"] < 65504.0 / 8
    lf = res["ln_fold"]["f16"]
    assert lf["fold_on"]["finite"] and not lf["fold_on"]["fell_back"] and lf["operand_range_fallbacks"] == 0 and lf["fold_on"]["ln_fold_after"] == 1
    assert lf["fold_on"]["mel_l1_vs_oracle"] <= MEL_L1_TOL and lf["fold_off"]["mel_l1_vs_oracle"] <= MEL_L1_TOL
    assert res["largest_operand"] < 65504.0 / 8 and any("attention q" in k for k in res["operand_peaks"]) and any("ff.ff.layers.2" in k for k in res["operand_peaks"])
    env = dict(os.environ, F5_WEIGHTS=str(mdir))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--ode-points", "3", "--no-sub", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["data"].startswith("REAL checkpoint") and "parity_l1" not in line and line["value"] > 0
    assert line["roofline"]["peak_measured_tflops"] > 500 and 0 < line["roofline"]["frac_of_measured"] < 1


def test_dit_call_remaining_argument_combinations(tiny_weights, tiny_x3):
    """VERDICT r3 missing #8: `DiT.__call__` (dit.py:374-401) accepts any (drop_audio_cond, drop_text) and per-row `time` together with
    a mask.  (False, True) -- text dropped, audio kept -- runs as the engine's second branch with `null_keeps_cond`; per-row times with a
    mask run row by row at the same padded length.  Against the oracle's forward on identical inputs (bf16x3: fp32-class)."""
    cfg = TINY
    B, N = 3, 90
    cond, text, durations, y0 = synth_inputs(cfg, B, N, nt=24, n_ref=30, seed=321, ragged=True)
    step_cond = _pad_cond(cond, N)
    mask = O.lens_to_mask(torch.tensor(durations), N)
    orc = O.DiTOracle(cfg, tiny_weights)
    x = y0
    for drop_a, drop_t in ((False, True), (True, False), (True, True), (False, False)):
        want = orc.forward(x, step_cond, text, torch.tensor(0.4), drop_a, drop_t, mask)
        got = tiny_x3(x=x, cond=step_cond, text=text, time=torch.tensor(0.4), drop_audio_cond=drop_a, drop_text=drop_t, mask=mask)
        _, mean, refm = report(f"dit[bf16x3] drop_audio_cond={drop_a} drop_text={drop_t} masked vs oracle", got.cpu(), want)
        assert mean <= 2e-4 * max(1.0, refm), (drop_a, drop_t)
    assert tiny_x3.engine.get_option("null_keeps_cond") == 0            # restored
    times = torch.tensor([0.1, 0.55, 0.9])
    for m in (None, mask):
        want = orc.forward(x, step_cond, text, times, False, False, m)
        got = tiny_x3(x=x, cond=step_cond, text=text, time=times, drop_audio_cond=False, drop_text=False, mask=m)
        _, mean, refm = report(f"dit[bf16x3] per-row times, mask={'yes' if m is not None else 'no'} vs oracle", got.cpu(), want)
        assert mean <= 2e-4 * max(1.0, refm)
    want = orc.forward(x, step_cond, text, times, False, True, mask)
    got = tiny_x3(x=x, cond=step_cond, text=text, time=times, drop_audio_cond=False, drop_text=True, mask=mask)
    _, mean, refm = report("dit[bf16x3] per-row times + mask + (False, True) vs oracle", got.cpu(), want)
    assert mean <= 2e-4 * max(1.0, refm)


def test_generate_batch_sentences(tiny_weights, tiny_x3):
    """VERDICT r3 missing #5 / SURVEY section 8(f)2: `generate(..., batch_sentences=True)` runs the sentences of a text as ONE ragged
    sample() batch.  It must equal the ORACLE's batched `sample()` on the same inputs (the reference's own batch semantics: key mask,
    GRN / conv-pos over the padding) sentence by sentence, and it is documented NOT to equal the per-sentence loop bit for bit --
    both facts are checked.  Fake frame-synchronous vocoder (256 mel-derived samples per frame), character vocabulary."""
    import os as _os
    from f5_tts_mlx_amd import generate as G
    from f5_tts_mlx_amd.utils import convert_char_to_pinyin, list_str_to_idx
    cfg = TINY
    vocab = {c: i for i, c in enumerate(" abcdefghijklmnopqrstuvwxyz.,!?'ABCDEFGHIJKLMNOPQRSTUVWXYZ")}
    assert len(vocab) <= cfg.text_num_embeds
    idx = torch.arange(256, device=DEV) % 100
    voc = lambda mel: mel[:, :, idx].reshape(mel.shape[0], -1) if mel.shape[0] > 1 else mel[0][:, idx].reshape(-1)   # noqa: E731
    f5 = F5TTS(transformer=tiny_x3, vocab_char_map=vocab, vocoder=voc)
    wav = _os.path.join(_os.path.dirname(G.__file__), "assets", "test_en_1_ref_short.wav")
    ref_text = "some call me nature."
    text = "Hello there. This one is a longer sentence, is it not? Short."
    kw = dict(ref_audio_path=wav, ref_audio_text=ref_text, steps=4, method="euler", seed=5, estimate_duration=True, f5tts=f5)
    got = G.generate(text, batch_sentences=True, **kw)
    loop = G.generate(text, batch_sentences=False, **kw)
    torch.cuda.synchronize()
    # the loop keeps the reference's carried-over `duration` (generate.py:203-208: from the second sentence on the FRAME count of the
    # previous one is multiplied by 93.75 again and clipped to 4096); the batch gives every sentence the estimate itself, so only the
    # first sentence has the same length in both
    # the oracle's batched sample() on the same batch
    audio, _ = G.read_wav(wav)
    audio = torch.from_numpy(np.asarray(audio)).to(torch.float32)
    sentences = G.split_sentences(text)
    toks = convert_char_to_pinyin([ref_text + " " + t for t in sentences])
    cond = torch.from_numpy(np.asarray(O.log_mel_spectrogram(audio.numpy()), np.float32).reshape(1, -1, 100)).repeat(len(sentences), 1, 1)
    dur = int(G.estimated_duration(audio, ref_text, text, 1.0) * G.FRAMES_PER_SEC)
    from f5_tts_mlx_amd.rng import mlx_like_normal
    ids = list_str_to_idx(toks, vocab)
    lens = torch.maximum((ids != -1).sum(-1), torch.full((len(sentences),), cond.shape[1]))
    durs = torch.clip(torch.maximum(lens + 1, torch.full((len(sentences),), dur)), 0, 4096)
    N = int(durs.max())
    y0 = torch.zeros((len(sentences), N, 100))
    for i, d in enumerate(durs.tolist()):
        y0[i, :d] = torch.from_numpy(mlx_like_normal(5, (100, d)).T.copy())
    ref, _ = O.sample(O.DiTOracle(cfg, tiny_weights), cond, ids, durs, y0=y0, steps=4, method="euler", vocab_char_map=vocab)
    ns = audio.shape[0]
    want = torch.cat([ref[i][:, idx.cpu()].reshape(-1)[ns:int(durs[i]) * 256] for i in range(len(sentences))])
    assert want.shape == got.shape, (want.shape, got.shape)
    l1 = float((got.cpu() - want).abs().mean())
    n1 = int(durs[0]) * 256 - ns
    l1_loop = float((got.cpu()[:n1] - loop.cpu()[:n1]).abs().mean())
    print(f"[batch_sentences] vs the oracle's batched sample(): {l1:.3e}; first sentence vs the per-sentence loop's: {l1_loop:.3e}")
    assert l1 <= MEL_L1_TOL and l1_loop <= 5e-3          # batch semantics (mask, padding) vs a batch-1 call: close, not identical
