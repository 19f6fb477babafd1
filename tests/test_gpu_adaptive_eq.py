"""GPU parity of the adaptive equaliser (ModemConfig::adaptive_eq_enabled, include/ultra/types.hpp:170-174; Impl::equalize's
use_adaptive branch with lmsUpdate / rlsUpdate, src/ofdm/channel_equalizer.cpp:569-581,705-722,773-805): off in every preset
the reference ships, built so that a caller who switches it on finds it.  Through the C-ABI, bitwise: frames of the compiled
reference (tests/golden/adaptive.npz, made by tests/golden/make_golden.py adaptive) and seeded batches against the oracle."""
import numpy as np
import pytest

from _util import beq, cfg_from_array, context_for, geometry, make_config
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

NAMES = ["lms_qpsk", "rls_qam16", "lms_qam64", "rls_bpsk", "lms_qam32_nodd", "rls_qam256"]


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("entry", [0, 1])
def test_golden_reference_frames(name, entry):
    g = np.load(GOLDEN / "adaptive.npz")
    key = f"{name}_e{entry}"
    cfg = cfg_from_array(g[f"{key}__cfg"])
    assert cfg.adaptive_eq_enabled == 1 and cfg.entry == entry
    ctx = context_for(cfg)
    par = g[f"{key}__cfo_phase"]
    llr, state = ctx.demod(g[f"{key}__audio"], cfo_hz=par[:, 0], cfo_phase=par[:, 1] if entry else None, want_state=True)
    ctx.synchronize()
    assert beq(llr.cpu().numpy(), g[f"{key}__llr"]), key
    st, sc = state.cpu().numpy(), g[f"{key}__scal"]
    for col in range(5):                                # cfo, noise variance, snr, timing, cfo phase after the last symbol
        assert beq(st[:, col], sc[:, col]), (key, col)


CASES = [("QPSK", "R1_2", 512, dict(n_data_symbols=22), "lms", {}), ("QAM16", "R3_4", 1024, dict(n_data_symbols=10), "rls", dict(rls_lambda=0.97)),
         ("QAM16", "R3_4", 1024, dict(n_data_symbols=10), "lms", dict(lms_mu=0.1)), ("QAM64", "R3_4", 512, dict(n_data_symbols=9), "rls", {}),
         ("QAM32", "R3_4", 1024, dict(n_data_symbols=8), "lms", {}), ("BPSK", "R1_2", 512, dict(n_data_symbols=12), "rls", {}),
         ("QAM256", "R5_6", 512, dict(n_data_symbols=8), "lms", {}),
         ("QAM16", "R1_2", 1024, dict(n_data_symbols=8, pilot_spacing=2), "rls", {}),            # 30 pilots
         ("QPSK", "R1_2", 512, dict(n_data_symbols=10, use_pilots=0), "rls", {}),                # coherent, no pilots
         ("QAM16", "R1_2", 1024, dict(n_data_symbols=8), "lms", dict(decision_directed=False))]


@pytest.mark.parametrize("mod,rate,fft,kw,kind,akw", CASES)
@pytest.mark.parametrize("entry", [0, 1])
def test_synthetic_batch_vs_oracle(oracle, mod, rate, fft, kw, kind, akw, entry):
    """192 frames per case, Watterson and AWGN, with and without initial offsets: soft bits, tracker scalars and the decode of
    the first codeword against the oracle — and the switch changes the soft bits (it is not silently ignored)."""
    cfg = make_config(fft, mod, rate, entry=entry, adaptive_eq=kind, **kw, **akw)
    plain = make_config(fft, mod, rate, entry=entry, **kw)
    n = 192
    rng = np.random.default_rng(11)
    for chan, snr, with_cfo in (("watterson", 22.0, True), ("awgn", 9.0, False)):
        audio, _ = oracle.make_batch(cfg, n, seed=0xADA + int(snr), channel=chan, snr_db=snr)
        cfo = rng.normal(0, 4.0, n).astype(np.float32) if with_cfo else None
        ph = rng.uniform(-3, 3, n).astype(np.float32) if (with_cfo and entry) else None
        want = oracle.demod_decode_batch(cfg, audio, cfo_hz=cfo, cfo_phase=ph, n_threads=16)
        ctx = context_for(cfg)
        llr, state = ctx.demod(audio, cfo_hz=cfo, cfo_phase=ph, want_state=True)
        ctx.synchronize()
        assert beq(llr.cpu().numpy(), want["llr"]), (mod, kind, chan)
        if ctx.geometry.llrs_per_frame >= 648:              # the fused entry decodes the frame's first codeword
            r = ctx.demod_decode(audio, cfo_hz=cfo, cfo_phase=ph, want_llr=True)
            ctx.synchronize()
            assert beq(r["llr"].cpu().numpy(), want["llr"]), (mod, kind, chan, "fused")
            assert np.array_equal(r["bytes"].cpu().numpy(), want["bytes"]) and np.array_equal(r["iters"].cpu().numpy(), want["iters"])
        st = state.cpu().numpy()
        for idx in (0, 1, 2, 3, 4, 5):
            assert beq(st[:, idx], want["state"][:, idx]), ("state", idx)
        if mod.startswith("QAM"):                           # (BPSK / QPSK at these SNRs: soft bits at the +-10 clip either way)
            off = context_for(plain).demod(audio, cfo_hz=cfo, cfo_phase=ph).cpu().numpy()
            assert not beq(off, want["llr"])


@pytest.mark.parametrize("kind", ["lms", "rls"])
@pytest.mark.parametrize("entry", [0, 1])
def test_streamed_symbols_carry_the_weights(oracle, kind, entry):
    """ultra_hip_demod_stream_batch with the adaptive equaliser: the weights (and the RLS gains) are part of the tracker
    record a call leaves behind — any split of a frame's symbols over calls gives the soft bits of one batch call."""
    cfg = make_config(1024, "QAM16", "R3_4", n_data_symbols=9, entry=entry, adaptive_eq=kind)
    g = geometry(cfg)
    n = 64
    audio, _ = oracle.make_batch(cfg, n, seed=78, channel="watterson", snr_db=24.0)
    cfo = np.random.default_rng(6).normal(0, 5.0, n).astype(np.float32)
    want = oracle.demod_decode_batch(cfg, audio, cfo_hz=cfo, n_threads=16)
    n_train = int(cfg.training_symbols)
    total = n_train + 9
    first = max(n_train, 1)
    for split in ([total], [first] + [1] * (total - first), [first, 4, total - first - 4]):
        ctx = context_for(cfg)
        parts, s0 = [], 0
        for k in split:
            a = audio[:, s0 * g.symbol_samples:(s0 + k) * g.symbol_samples]
            parts.append(ctx.demod_stream(np.ascontiguousarray(a), s0, k, cfo_hz=cfo if s0 == 0 else None).cpu().numpy())
            s0 += k
        got = np.concatenate([p for p in parts if p.shape[1]], axis=1)
        assert beq(got, want["llr"]), (kind, entry, split)


def test_differential_modulations_ignore_the_switch(oracle):
    """equalize returns before the adaptive branch for DBPSK / DQPSK / D8PSK (channel_equalizer.cpp:740-769)."""
    for mod in ("DQPSK", "D8PSK"):
        cfg = make_config(512, mod, "R1_2", adaptive_eq="rls")
        plain = make_config(512, mod, "R1_2")
        audio, _ = oracle.make_batch(cfg, 33, seed=3, channel="awgn", snr_db=10.0)
        a = context_for(cfg).demod(audio).cpu().numpy()
        b = context_for(plain).demod(audio).cpu().numpy()
        assert beq(a, b) and beq(a, oracle.demod_decode_batch(cfg, audio, n_threads=4)["llr"])


def test_flags_are_validated():
    import ctypes as C
    from projectultra_amd import _lib
    lib = _lib.lib()
    g = _lib.ultra_hip_geometry()
    c = make_config(1024, "QAM16", "R3_4", adaptive_eq="lms")
    pc = _lib.ultra_hip_config()
    C.memmove(C.byref(pc), C.byref(c), C.sizeof(pc))
    assert lib.ultra_hip_geometry_for(C.byref(pc), C.byref(g)) == 0
    for field in ("adaptive_eq_enabled", "adaptive_eq_use_rls", "decision_directed"):
        setattr(pc, field, 2)
        assert lib.ultra_hip_geometry_for(C.byref(pc), C.byref(g)) == -1
        setattr(pc, field, 1)
