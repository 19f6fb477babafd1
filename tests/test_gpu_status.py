"""ultra_hip_get_status (ABI 10): the context's own fall-back paths are VISIBLE to the caller, and results on them stay
bit-identical.

The reference's convention is "failure = false / empty, never silent" ("Errors").  Before ABI 10 three product
decisions were silent: launch_demod dropping to the per-symbol chain when the per-(frame, symbol) workspace could not be had,
ultra_hip_create keeping the message-passing decoder when its LDS probe failed, and the decoder's screen staying shut.  Each now
sets a sticky bit (ULTRA_HIP_ST_*) or reports its decision; bench.py prints the word and refuses to call a line "default path"
with a bit set."""
import os

import numpy as np
import pytest

from _util import beq, context_for, make_config, noisy_codewords

pytestmark = pytest.mark.gpu


def test_default_path_reports_nothing(oracle):
    cfg = make_config(1024, "QAM16", "R3_4")
    audio, _ = oracle.make_batch(cfg, 96, seed=0x51A7, channel="watterson", snr_db=30.0)
    ctx = context_for(cfg)
    r = ctx.demod_decode(audio, want_llr=True)
    st = ctx.status()
    assert st["flags"] == 0 and st["paths"] == [], st
    assert st["screen_launches"] == 0                                # 96 codewords: below the screen's launch-size floor
    want = oracle.demod_decode_batch(cfg, audio, n_threads=8)
    assert beq(r["llr"].cpu().numpy(), want["llr"])


@pytest.mark.parametrize("mod,rate,fft", [("QAM16", "R3_4", 1024), ("DQPSK", "R1_2", 512)])
def test_workspace_cap_forces_the_per_symbol_chain_and_is_reported(oracle, mod, rate, fft):
    """The deferred chain (coherent + pilots) and the all-symbols launch (no pilots, CFO 0) both need n_symbols rows of workspace
    per frame.  With the cap below that the batch must still decode — on the per-symbol chain, bit-identical — and the status
    word must say so; with the cap lifted and the status cleared the next batch is back on the default path."""
    cfg = make_config(fft, mod, rate)
    n = 257
    audio, _ = oracle.make_batch(cfg, n, seed=0xCA9, channel="watterson" if fft == 1024 else "awgn", snr_db=18.0)
    want = oracle.demod_decode_batch(cfg, audio, n_threads=8)
    ctx = context_for(cfg)
    ctx.set_workspace_limit(64 * 1024)                               # a few dozen rows: far below 257 frames x 4 (or 11) symbols
    r = ctx.demod_decode(audio, want_llr=True)
    st = ctx.status()
    assert "demod_workspace_fallback" in st["paths"] and st["flags"] & 0x01, st
    assert beq(r["llr"].cpu().numpy(), want["llr"])
    for k in ("bytes", "iters", "ok"):
        assert np.array_equal(r[k].cpu().numpy(), want[k]), k
    ctx.set_workspace_limit(0)
    ctx.clear_status()
    r = ctx.demod_decode(audio, want_llr=True)
    assert ctx.status()["paths"] == []
    assert beq(r["llr"].cpu().numpy(), want["llr"])


def test_cap_applies_to_buffers_already_held(oracle):
    """A context that already holds the large workspace gives it back when the cap arrives."""
    cfg = make_config(1024, "QAM16", "R3_4")
    audio, _ = oracle.make_batch(cfg, 300, seed=0xCAA, channel="watterson", snr_db=25.0)
    want = oracle.demod_decode_batch(cfg, audio, n_threads=8)
    ctx = context_for(cfg)
    ctx.reserve(300)
    assert np.array_equal(ctx.demod_decode(audio)["bytes"].cpu().numpy(), want["bytes"]) and ctx.status()["paths"] == []
    ctx.set_workspace_limit(32 * 1024)
    assert np.array_equal(ctx.demod_decode(audio)["bytes"].cpu().numpy(), want["bytes"])
    assert ctx.status()["paths"] == ["demod_workspace_fallback"]


def test_environment_overrides_are_reported(oracle):
    for env, name in (("ULTRA_HIP_FALLBACK_CHAIN", "forced_fallback_chain"), ("ULTRA_HIP_LDPC_MESSAGES", "forced_message_kernel")):
        old = os.environ.get(env)
        os.environ[env] = "1"
        try:
            cfg = make_config(1024, "QAM16", "R3_4")
            ctx = context_for(cfg)
            assert name in ctx.status()["paths"]
            llr, _ = noisy_codewords(oracle, 4, 64, [0.5, 0.7], seed=5)
            import torch
            ctx.ldpc_decode(torch.from_numpy(llr).cuda())
            st = ctx.status()
            assert ("ldpc_message_kernel" in st["paths"]) == (env == "ULTRA_HIP_LDPC_MESSAGES"), st
            ctx.clear_status()
            assert name in ctx.status()["paths"], "what the environment forced stays true of every launch: clear keeps it"
        finally:
            if old is None:
                del os.environ[env]
            else:
                os.environ[env] = old
    old = os.environ.get("ULTRA_HIP_LDPC_SCREEN")
    os.environ["ULTRA_HIP_LDPC_SCREEN"] = "0"
    try:
        assert "screen_overridden" in context_for(make_config(512, "QPSK", "R1_2")).status()["paths"]
    finally:
        if old is None:
            del os.environ["ULTRA_HIP_LDPC_SCREEN"]
        else:
            os.environ["ULTRA_HIP_LDPC_SCREEN"] = old


@pytest.mark.parametrize("sigma,expect_open", [(0.28, True), (1.4, False)])
def test_screen_decision_is_reported(oracle, sigma, expect_open):
    """A launch large enough for the screen: the status carries the sample (how many of <= 2,048 sampled codewords were clean
    as received), the gate, whether the full pass ran, and how many codewords it left to the iterating kernel — clean channel:
    open, most codewords finished by the pass; noisy channel: shut, no flag (a shut gate is a decision, not a degradation)."""
    import torch
    rate, n_base, reps = 2, 1024, 40                                  # 40,960 codewords >= the screen's floor of 32,768
    llr, _ = noisy_codewords(oracle, rate, n_base, [sigma], seed=91)
    big = np.tile(llr, (reps, 1))
    ctx = context_for(make_config(512, "QPSK", "R1_2"))
    r = ctx.ldpc_decode(torch.from_numpy(big).cuda())
    st = ctx.status()
    assert st["flags"] == 0, st
    assert st["screen_launches"] == 1 and st["screen_sample_n"] == 2048 and st["screen_gate"] == 683, st
    assert bool(st["screen_gate_open"]) == expect_open, st
    ob, oi, ook = oracle.ldpc_decode_batch(rate, llr)[:3]
    assert np.array_equal(r["iters"].cpu().numpy(), np.tile(oi, reps)) and np.array_equal(r["ok"].cpu().numpy(), np.tile(ook, reps))
    assert np.array_equal(r["bytes"].cpu().numpy(), np.tile(ob, (reps, 1)))
    # clean as the screen defines it: the hard decisions of the channel values satisfy every row of the parity-check matrix
    rp, ci = ctx.tanner_graph()
    hard = (llr < 0).astype(np.uint8)
    syndrome = np.stack([hard[:, ci[rp[i]:rp[i + 1]]].sum(axis=1) & 1 for i in range(len(rp) - 1)], axis=1)
    clean = int((syndrome.sum(axis=1) == 0).sum()) * reps
    assert clean <= int((oi == 0).sum()) * reps                      # (a clean codeword is one the reference finishes at iteration 0)
    if expect_open:
        assert st["screen_dirty"] == big.shape[0] - clean, (st, clean)
        assert st["screen_sample_clean"] >= st["screen_gate"]
    else:
        assert st["screen_dirty"] == 0 and st["screen_sample_clean"] < st["screen_gate"]
