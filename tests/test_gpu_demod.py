"""GPU parity: HIP demodulator (through the C-ABI) vs the committed reference fixtures and
vs the oracle on seeded synthetic frames.  LLRs are compared BITWISE (stricter than the
1e-5 tolerance of the north star); decoded bytes / iterations / success exactly."""
import numpy as np
import pytest

from _util import INFO_BITS, beq, cfg_from_array, context_for, geometry, long_acquisition_streams, make_config
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
LLR_TOL = 1e-5     # north_star tolerance for soft LLRs; bitwise equality is asserted first

MODES = ["cfg3_qam16_r34", "cfg2_dqpsk_r12", "qpsk_r12_512", "qam32_r34", "d8psk_r34", "dbpsk_r14", "bpsk_r12",
         "qam64_r34", "qam256_r56", "dqpsk_pilots_r14", "qam16_r23_long"]


def _check_llr(got, want, what):
    if beq(got, want):
        return
    bad = np.flatnonzero(got.view(np.uint32).ravel() != want.view(np.uint32).ravel())
    err = np.nanmax(np.abs(got.ravel()[bad] - want.ravel()[bad]))
    raise AssertionError(f"{what}: {bad.size} LLRs differ bitwise (max abs err {err:g}, tol {LLR_TOL}); first {bad[:6]}")


@pytest.mark.parametrize("name", MODES)
def test_golden_reference_frames(name):
    """Audio in, the compiled reference's LLRs / tracker state / decode results out."""
    g = np.load(GOLDEN / "demod.npz")
    cfg = cfg_from_array(g[f"{name}__cfg"])
    ctx = context_for(cfg)
    llr, state = ctx.demod(g[f"{name}__audio"], cfo_hz=g[f"{name}__cfo"], want_state=True)
    ctx.synchronize()
    llr, state = llr.cpu().numpy(), state.cpu().numpy()
    _check_llr(llr, g[f"{name}__llr"], name)
    scal = g[f"{name}__scal"][:, -1, :]             # tracker scalars after the last symbol
    for col, idx in ((0, 0), (1, 1), (2, 2), (3, 3), (4, 4)):   # cfo, noise var, snr, timing, cfo phase
        assert beq(state[:, idx], scal[:, col]), (name, "state", idx, state[:, idx], scal[:, col])
    assert np.array_equal(state[:, 6], scal[:, 7])
    if ctx.geometry.llrs_per_frame >= 648:
        r = ctx.demod_decode(g[f"{name}__audio"], cfo_hz=g[f"{name}__cfo"], want_llr=True)
        ctx.synchronize()
        assert np.array_equal(r["bytes"].cpu().numpy(), g[f"{name}__bytes"])
        assert np.array_equal(r["ok"].cpu().numpy(), g[f"{name}__meta"][:, 0])
        assert np.array_equal(r["iters"].cpu().numpy(), g[f"{name}__meta"][:, 1])
        _check_llr(r["llr"].cpu().numpy(), g[f"{name}__llr"], name + " fused")


@pytest.mark.parametrize("name", ["cfg3_qam16_r34", "cfg2_dqpsk_r12"])
def test_fused_path_with_channel_deinterleaver(oracle, name):
    """RxPipeline order of operations (rx_pipeline.cpp:221-230): soft bits -> per-codeword
    ChannelInterleaver::deinterleave -> decodeSoft.  The fused GPU call with ultra_hip_set_deinterleave must
    equal: reference LLRs (fixture) -> deinterleave on the host -> oracle decode; the LLRs it hands back stay
    in channel order."""
    from projectultra_amd import ChannelInterleaver
    g = np.load(GOLDEN / "demod.npz")
    cfg = cfg_from_array(g[f"{name}__cfg"])
    ctx = context_for(cfg)
    bps = ctx.geometry.llrs_per_symbol
    il = ChannelInterleaver(bps)
    ctx.set_deinterleave(bps)
    try:
        r = ctx.demod_decode(g[f"{name}__audio"], cfo_hz=g[f"{name}__cfo"], want_llr=True)
        ctx.synchronize()
    finally:
        ctx.set_deinterleave(0)
    _check_llr(r["llr"].cpu().numpy(), g[f"{name}__llr"], name + " fused, channel order")
    deint = np.stack([il.deinterleave(row[:648]) for row in g[f"{name}__llr"]])
    ob, oi, ook = oracle.ldpc_decode_batch(int(cfg.code_rate), deint)
    assert np.array_equal(r["bytes"].cpu().numpy(), ob)
    assert np.array_equal(r["iters"].cpu().numpy(), oi) and np.array_equal(r["ok"].cpu().numpy(), ook)


@pytest.mark.parametrize("name", ["ps_dqpsk", "ps_qam16", "ps_d8psk", "ps_qpsk"])
def test_golden_presynced_frames(name):
    g = np.load(GOLDEN / "presynced.npz")
    cfg = cfg_from_array(g[f"{name}__cfg"])
    ctx = context_for(cfg)
    par = g[f"{name}__cfo_phase"]
    llr, state = ctx.demod(g[f"{name}__audio"], cfo_hz=par[:, 0], cfo_phase=par[:, 1], want_state=True)
    ctx.synchronize()
    _check_llr(llr.cpu().numpy(), g[f"{name}__llr"], name)
    st, sc = state.cpu().numpy(), g[f"{name}__scal"]
    for col, idx in ((0, 0), (1, 1), (2, 2), (3, 3), (4, 4)):
        assert beq(st[:, idx], sc[:, col]), (name, idx)


@pytest.mark.parametrize("name", ["cfg3_qam16_r34", "cfg2_dqpsk_r12"])
def test_full_sync_equivalence(name):
    """Whole frames the reference received through OFDMDemodulator::process (Schmidl-Cox search,
    960-sample chunks): entering our SYNCED entry at the data start with the coarse CFO the search
    produced reproduces those LLRs bit for bit."""
    g = np.load(GOLDEN / "fullsync.npz")
    cfg = cfg_from_array(g[f"{name}__cfg"])
    ctx = context_for(cfg)
    fs = ctx.geometry.frame_samples
    meta = g[f"{name}__meta"]
    assert (meta[:, 2] >= 0).all()
    audio = np.stack([a[s:s + fs] for a, s in zip(g[f"{name}__audio"], meta[:, 2])])
    llr = ctx.demod(audio, cfo_hz=g[f"{name}__cfo"])
    ctx.synchronize()
    _check_llr(llr.cpu().numpy(), g[f"{name}__llr"], name)


SYNTH = [("QAM16", "R3_4", 1024, {}, "watterson"), ("DQPSK", "R1_2", 512, {}, "awgn"),
         ("QAM32", "R3_4", 1024, {}, "watterson"), ("D8PSK", "R3_4", 1024, dict(pilot_spacing=2), "awgn"),
         ("QPSK", "R2_3", 512, {}, "watterson"), ("DBPSK", "R1_4", 512, {}, "awgn"),
         ("QAM64", "R5_6", 512, {}, "awgn"), ("QAM16", "R1_2", 1024, dict(n_data_symbols=10), "watterson"),
         # 30 pilots: the pilot kernel's two-frames-per-wavefront instance
         ("QAM16", "R1_2", 1024, dict(pilot_spacing=2), "watterson"),
         ("DQPSK", "R1_2", 1024, dict(pilot_spacing=2, use_pilots=1), "awgn")]


@pytest.mark.parametrize("mod,rate,fft,kw,chan", SYNTH)
@pytest.mark.parametrize("snr", [30.0, 12.0, 4.0])
def test_synthetic_batch_vs_oracle(oracle, mod, rate, fft, kw, chan, snr):
    cfg = make_config(fft, mod, rate, **kw)
    n = 192
    audio, payload = oracle.make_batch(cfg, n, seed=0xABC + int(snr), channel=chan, snr_db=snr)
    rng = np.random.default_rng(5)
    cfo = np.where(np.arange(n) % 3 == 0, 0.0, rng.normal(0, 4.0, n)).astype(np.float32)
    want = oracle.demod_decode_batch(cfg, audio, cfo_hz=cfo, n_threads=8)
    ctx = context_for(cfg)
    r = ctx.demod_decode(audio, cfo_hz=cfo, want_llr=True)
    llr2, state = ctx.demod(audio, cfo_hz=cfo, want_state=True)
    ctx.synchronize()
    _check_llr(r["llr"].cpu().numpy(), want["llr"], f"{mod} {rate} {snr}")
    assert beq(llr2.cpu().numpy(), want["llr"])
    assert np.array_equal(r["bytes"].cpu().numpy(), want["bytes"])
    assert np.array_equal(r["iters"].cpu().numpy(), want["iters"])
    assert np.array_equal(r["ok"].cpu().numpy(), want["ok"])
    st = state.cpu().numpy()
    for idx in (0, 1, 2, 3, 4, 5):
        assert beq(st[:, idx], want["state"][:, idx]), ("state", idx)


@pytest.mark.parametrize("mod,rate,fft,kw", [("QAM16", "R3_4", 1024, {}), ("DQPSK", "R1_2", 512, {}),
                                             ("D8PSK", "R3_4", 1024, dict(pilot_spacing=2)), ("QPSK", "R1_2", 512, {}),
                                             ("QAM16", "R1_2", 1024, dict(pilot_spacing=2))])
def test_synthetic_presynced_vs_oracle(oracle, mod, rate, fft, kw):
    cfg = make_config(fft, mod, rate, entry=1, **kw)
    n = 96
    audio, payload = oracle.make_batch(cfg, n, seed=77, channel="awgn", snr_db=18.0)
    rng = np.random.default_rng(9)
    cfo = rng.normal(0, 6.0, n).astype(np.float32)
    ph = rng.uniform(-3.1, 3.1, n).astype(np.float32)
    want = oracle.demod_decode_batch(cfg, audio, cfo_hz=cfo, cfo_phase=ph, n_threads=8)
    ctx = context_for(cfg)
    r = ctx.demod_decode(audio, cfo_hz=cfo, cfo_phase=ph, want_llr=True)
    ctx.synchronize()
    _check_llr(r["llr"].cpu().numpy(), want["llr"], f"presynced {mod}")
    assert np.array_equal(r["bytes"].cpu().numpy(), want["bytes"])
    assert np.array_equal(r["iters"].cpu().numpy(), want["iters"])


@pytest.mark.parametrize("mod,rate", [("DBPSK", "R1_4"), ("DQPSK", "R1_2"), ("D8PSK", "R2_3")])
def test_pair_tracker_odd_batches(oracle, mod, rate):
    """Differential 512-point layouts without pilots run two frames per wavefront (track_diff_pair_kernel): odd batch
    sizes (the last wavefront's upper half has no frame), the all-symbols-at-once path (no initial CFO) and the
    per-symbol path (with one), soft bits and tracker state against the oracle."""
    cfg = make_config(512, mod, rate)
    audio, _ = oracle.make_batch(cfg, 7, seed=0x51, channel="awgn", snr_db=9.0)
    ctx = context_for(cfg)
    rng = np.random.default_rng(3)
    for n in (1, 3, 7):
        for cfo in (None, rng.normal(0, 3.0, n).astype(np.float32)):
            want = oracle.demod_decode_batch(cfg, audio[:n], cfo_hz=cfo, n_threads=2)
            llr, state = ctx.demod(audio[:n], cfo_hz=cfo, want_state=True)
            ctx.synchronize()
            assert beq(llr.cpu().numpy(), want["llr"]), (mod, n, cfo is None)
            st = state.cpu().numpy()
            for idx in (0, 1, 2, 3, 4, 5):
                assert beq(st[:, idx], want["state"][:, idx]), ("state", idx, mod, n)


@pytest.mark.parametrize("fft,mod,rate,kw", [
    (1024, "QAM16", "R3_4", dict(carriers=64, pilot_spacing=2)),      # 128-position rows, 32 pilots (two frames per wavefront in the pilot half)
    (1024, "QPSK", "R1_2", dict(carriers=63, pilot_spacing=3)),
    (512, "QAM64", "R2_3", dict(carriers=64, pilot_spacing=4)),
    (1024, "DQPSK", "R1_2", dict(carriers=24)),                       # 1024 points, two frames per wavefront in the tracker
    (512, "D8PSK", "R3_4", dict(carriers=12)),
    (512, "DQPSK", "R1_2", dict(carriers=33)),                        # one carrier too many for the pair tracker
    (1024, "DBPSK", "R1_4", dict(carriers=64)),
])
def test_carrier_counts(oracle, fft, mod, rate, kw):
    """Layouts other than the presets' 30 / 59 carriers: rows of 128 bins (carriers beyond +-31), the pilots-first row order
    with 16 to 32 pilots, the two-frames-per-wavefront tracker at and around its 32-carrier limit — soft bits, decoded bytes,
    iteration counts and tracker state against the oracle, with and without an initial CFO."""
    cfg = make_config(fft, mod, rate, **kw)
    n = 45
    audio, _ = oracle.make_batch(cfg, n, seed=0xC0 + fft // 512, channel="awgn", snr_db=14.0)
    rng = np.random.default_rng(17)
    ctx = context_for(cfg)
    for cfo in (None, rng.normal(0, 3.0, n).astype(np.float32)):
        want = oracle.demod_decode_batch(cfg, audio, cfo_hz=cfo, n_threads=8)
        r = ctx.demod_decode(audio, cfo_hz=cfo, want_llr=True)
        llr2, state = ctx.demod(audio, cfo_hz=cfo, want_state=True)
        ctx.synchronize()
        assert beq(r["llr"].cpu().numpy(), want["llr"]), (fft, mod, kw, cfo is None)
        assert beq(llr2.cpu().numpy(), want["llr"])
        assert np.array_equal(r["bytes"].cpu().numpy(), want["bytes"]) and np.array_equal(r["iters"].cpu().numpy(), want["iters"])
        st = state.cpu().numpy()
        for idx in (0, 1, 2, 3, 4, 5):
            assert beq(st[:, idx], want["state"][:, idx]), ("state", idx, fft, mod, kw)


def test_ragged_and_strided_inputs(oracle):
    """Rows longer than a frame (stride > frame_samples), one-frame batches, empty batches."""
    import torch
    cfg = make_config(1024, "QAM16", "R3_4")
    audio, _ = oracle.make_batch(cfg, 5, seed=1, channel="awgn", snr_db=20.0)
    want = oracle.demod_decode_batch(cfg, audio, n_threads=2)
    ctx = context_for(cfg)
    padded = np.concatenate([audio, np.full((5, 37), 9.0, np.float32)], axis=1)   # garbage past the frame
    r = ctx.demod_decode(torch.from_numpy(padded).cuda(), want_llr=True)
    one = ctx.demod(audio[2:3])
    ctx.synchronize()
    assert beq(r["llr"].cpu().numpy(), want["llr"])
    assert beq(one.cpu().numpy(), want["llr"][2:3])
    empty = ctx.demod_decode(torch.zeros((0, ctx.geometry.frame_samples), device="cuda"))
    assert empty["bytes"].shape[0] == 0
    from projectultra_amd import UltraHipError
    with pytest.raises(UltraHipError):
        ctx.demod(audio[:, :100])


def test_device_math_matches_host_libm():
    """pinned_math.h evaluated ON THE GPU equals the libm the reference calls, bit for bit."""
    import ctypes, ctypes.util
    libm = ctypes.CDLL(ctypes.util.find_library("m"))
    for f in ("sinf", "cosf", "atanf"):
        getattr(libm, f).restype = ctypes.c_float; getattr(libm, f).argtypes = [ctypes.c_float]
    for f in ("atan2f", "hypotf"):
        getattr(libm, f).restype = ctypes.c_float; getattr(libm, f).argtypes = [ctypes.c_float, ctypes.c_float]
    cfg = make_config(1024, "QAM16", "R3_4")
    ctx = context_for(cfg)
    rng = np.random.default_rng(17)
    n = 200_000
    a = np.concatenate([rng.uniform(-7, 7, n // 2), rng.normal(0, 100, n // 4), rng.normal(0, 1e-3, n // 4)]).astype(np.float32)
    a[:8] = [0.0, -0.0, np.pi, -np.pi, 1.0, 120.0, 1e-30, 3e4]
    b = rng.normal(0, 3, n).astype(np.float32); b[::97] = 0.0; b[1::97] = 1.0
    for fn, name in enumerate(["sinf", "cosf", "atanf", "atan2f", "hypotf"]):
        got = ctx.selftest_math(fn, a, b if fn >= 3 else None)
        ctx.synchronize()
        got = got.cpu().numpy()
        f = getattr(libm, name)
        want = np.array([f(x) if fn < 3 else f(x, y) for x, y in zip(a.tolist(), b.tolist())], np.float32)
        assert beq(got, want), (name, np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))[:5])


@pytest.mark.parametrize("mod,rate,fft,kw", [("DQPSK", "R1_2", 512, {}), ("QAM16", "R3_4", 1024, {}), ("D8PSK", "R2_3", 1024, dict(pilot_spacing=2))])
def test_presynced_training_cfo_estimate(oracle, mod, rate, fft, kw):
    """ULTRA_ENTRY_PRESYNCED with a NaN initial CFO = processPresynced on a demodulator whose frequency offset was never
    set: the CFO comes from the two training symbols (estimateCFOFromTraining, ofdm_sync.cpp:278-380).  Frequency-shifted
    frames at several SNRs, a noise-only buffer (correlation gate), mixed in one batch with frames that DO carry a preset
    CFO: LLRs bitwise, the tracker's final CFO bitwise, against the oracle (pinned to the reference for this branch by
    tests/test_oracle_vs_ref.py::test_presynced_without_a_preset_cfo); and the host mirror OFDMDemodulator without
    setFrequencyOffset."""
    from scipy.signal import hilbert
    from projectultra_amd import OFDMDemodulator
    from _util import modem_config_from_c
    cfg = make_config(fft, mod, rate, entry=1, **kw)
    g = geometry(cfg)
    rng = np.random.default_rng(15)
    frames, cfo_in = [], []
    for trial in range(12):
        enc = oracle.ldpc_encode(cfg.code_rate, bytes(rng.integers(0, 256, INFO_BITS[cfg.code_rate] // 8, dtype=np.uint8)))
        a = oracle.modulate_presynced(cfg, enc)
        x = a * np.float32(0.5 / np.abs(a).max())
        shift = [0.0, 4.0, -7.5, 12.0, -15.0, 30.0][trial % 6]
        if shift:
            x = np.real(hilbert(x.astype(np.float64)) * np.exp(2j * np.pi * shift * np.arange(x.size) / 48000.0)).astype(np.float32)
        snr = [40.0, 30.0, 20.0, 12.0, 6.0, 25.0][trial % 6]
        x = (x + rng.normal(0, np.sqrt(np.mean(x.astype(np.float64) ** 2) / 10 ** (snr / 10)), x.size)).astype(np.float32)
        if trial == 11:
            x = rng.normal(0, 0.1, x.size).astype(np.float32)
        frames.append(x[:g.frame_samples])
        cfo_in.append(np.nan if trial % 3 != 2 else float(shift))      # every third frame has a preset (trusted) CFO
    audio, cfo_in = np.stack(frames), np.array(cfo_in, np.float32)
    ctx = context_for(cfg)
    llr, state = ctx.demod(audio, cfo_hz=cfo_in, want_state=True)
    ctx.synchronize()
    llr, state = llr.cpu().numpy(), state.cpu().numpy()
    for i in range(len(frames)):
        want, _, scal = oracle.demod_presynced(cfg, frames[i], None if np.isnan(cfo_in[i]) else float(cfo_in[i]), 0.0)
        assert beq(llr[i, :want.size], want), (mod, i)
        assert beq(state[i, 0], scal[0]), (mod, i, state[i, 0], scal[0])
    mc, kwc = modem_config_from_c(cfg)
    d = OFDMDemodulator(mc)
    d.processPresynced(frames[1], training_symbols=2)                   # no setFrequencyOffset* call before
    want, _, _ = oracle.demod_presynced(cfg, frames[1], None)
    got = np.concatenate([d.getSoftBits() for _ in range(8)])
    assert beq(got[:want.size], want[:got.size]) and got.size > 0


def test_device_logf_sqrtf_match_host_libm():
    """The two libm functions of the stimulus generators' Box-Muller transform, evaluated ON THE GPU: logf
    (pinned_math.h, glibc 2.35's FMA build) and the correctly rounded sqrtf — over (0, 1] on the 24-bit grid the
    generator draws from, and over arbitrary floats incl. zeros, subnormals, negatives, infinities and NaN."""
    import ctypes, ctypes.util
    libm = ctypes.CDLL(ctypes.util.find_library("m"))
    for f in ("logf", "sqrtf"):
        getattr(libm, f).restype = ctypes.c_float; getattr(libm, f).argtypes = [ctypes.c_float]
    ctx = context_for(make_config(512, "DQPSK", "R1_2"))
    rng = np.random.default_rng(23)
    grid = ((rng.integers(0, 1 << 24, 150_000).astype(np.float32) + 1.0) * np.float32(2.0 ** -24)).astype(np.float32)
    anyf = rng.integers(0, 1 << 32, 150_000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    special = np.array([0.0, -0.0, 1.0, 2.0 ** -24, 1e-45, 1.1754942e-38, 1.17549435e-38, -1.0, np.inf, -np.inf, np.nan,
                        3.4028235e38, 0.99999994, 1.0000001], np.float32)
    a = np.concatenate([special, grid, anyf])
    for fn, name in ((5, "logf"), (6, "sqrtf")):
        got = ctx.selftest_math(fn, a)
        ctx.synchronize()
        got = got.cpu().numpy()
        f = getattr(libm, name)
        want = np.array([f(x) for x in a.tolist()], np.float32)
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (name, a[~same][:5], got[~same][:5], want[~same][:5])


@pytest.mark.parametrize("fft,mod,rate", [(1024, "QAM16", "R3_4"), (512, "DQPSK", "R1_2")])
def test_acquisition_matches_oracle(oracle, fft, mod, rate):
    """Scope row f1: the chunk-fed Schmidl-Cox search + coarse CFO + LTS refinement on the GPU equals the
    oracle (which tests/test_oracle_vs_ref.py pins to the compiled reference, stage by stage and whole):
    found / data start / Schmidl-Cox offset / samples fed exactly, coarse CFO bitwise.  Streams: whole
    frames at several SNRs (the search fails at low SNR), shifted starts and levels, pure noise, silence;
    two chunkings.  Then the acquired (data_start, cfo) feeds the SYNCED entry: LLRs equal the oracle's."""
    cfg = make_config(fft, mod, rate)
    g = geometry(cfg)
    rng = np.random.default_rng(11)
    n_samples = g.frame_samples + 7 * (fft + g.cp_len) + 3200
    streams = []
    for t in range(10):
        payload = bytes(rng.integers(0, 256, INFO_BITS[cfg.code_rate] // 8, dtype=np.uint8))
        a, _ = oracle.modulate_frame(cfg, oracle.ldpc_encode(int(cfg.code_rate), payload))   # preamble + data
        a = a * np.float32(0.5 / np.abs(a).max())
        snr_db = [30.0, 26.0, 20.0, 12.0][t % 4]
        sigma = np.sqrt(np.mean(a.astype(np.float64) ** 2) / 10 ** (snr_db / 10))
        a = (a + rng.normal(0, sigma, a.size)).astype(np.float32)
        lead = rng.normal(0, 2e-4, int(rng.integers(0, 2800))).astype(np.float32)
        x = np.concatenate([lead, a * np.float32(0.4 + 0.15 * (t % 5))])
        x = np.concatenate([x, rng.normal(0, 2e-4, max(0, n_samples - x.size)).astype(np.float32)])[:n_samples]
        streams.append(x)
    streams.append(rng.normal(0, 0.05, n_samples).astype(np.float32))     # noise only
    streams.append(np.zeros(n_samples, np.float32))                       # silence
    audio = np.stack(streams)
    ctx = context_for(cfg)
    hits = 0
    for chunk in (960, 3000):
        r = ctx.acquire(audio, chunk)
        ctx.synchronize()
        r = {k: v.cpu().numpy() for k, v in r.items()}
        for i, x in enumerate(streams):
            o = oracle.acquire(cfg, x, chunk)
            assert r["found"][i] == o["found"], (i, chunk, o)
            if o["found"]:
                hits += 1
                assert r["data_start"][i] == o["data_start"] and r["sync_offset"][i] == o["sync_offset"], (i, chunk, o)
                assert r["fed_at_sync"][i] == o["fed_at_sync"], (i, chunk, o)
                assert np.float32(r["cfo_hz"][i]).tobytes() == np.float32(o["coarse_cfo"]).tobytes(), (i, chunk, o)
    assert hits >= 6
    # acquired entry -> SYNCED demodulation, against the oracle on the same (data_start, cfo)
    r = ctx.acquire(audio, 960)
    ctx.synchronize()
    found = r["found"].cpu().numpy().astype(bool)
    ds, cfo = r["data_start"].cpu().numpy(), r["cfo_hz"].cpu().numpy()
    ok = found & (ds + g.frame_samples <= n_samples)
    frames = np.stack([audio[i, ds[i]: ds[i] + g.frame_samples] for i in np.flatnonzero(ok)])
    llr = ctx.demod(frames, cfo_hz=cfo[ok])
    ctx.synchronize()
    want = np.stack([oracle.demod_synced(cfg, f, c)[0] for f, c in zip(frames, cfo[ok])])
    _check_llr(llr.cpu().numpy(), want, "acquired entry")


@pytest.mark.parametrize("name", ["cfg3_qam16_r34", "cfg2_dqpsk_r12"])
def test_receive_from_raw_audio_equals_reference(oracle, name):
    """End to end (ultra_hip_receive_batch): the whole frames of tests/golden/fullsync.npz, which the compiled
    reference received through OFDMDemodulator::process fed in 960-sample chunks.  The GPU finds the same
    data start and coarse CFO and produces the reference's LLRs bit for bit; decode results equal the
    oracle's decode of those LLRs.  A truncated and a silent stream yield no frame."""
    g = np.load(GOLDEN / "fullsync.npz")
    cfg = cfg_from_array(g[f"{name}__cfg"])
    ctx = context_for(cfg)
    audio = g[f"{name}__audio"]
    meta = g[f"{name}__meta"]
    n_samples = audio.shape[1]
    # delayed by 100 samples: sync is found but the frame no longer fits into the stream; and silence
    extra = np.stack([np.concatenate([np.zeros(100, np.float32), audio[0][:-100]]), np.zeros(n_samples, np.float32)])
    r = ctx.receive(np.concatenate([audio, extra]), 960, want_llr=True)
    ctx.synchronize()
    r = {k: v.cpu().numpy() for k, v in r.items()}
    n = audio.shape[0]
    assert np.array_equal(r["entry"][:n], meta[:, 2])
    assert beq(r["cfo_hz"][:n], g[f"{name}__cfo"])
    _check_llr(r["llr"][:n], g[f"{name}__llr"], name + " end to end")
    ob, oi, ook = oracle.ldpc_decode_batch(int(cfg.code_rate), g[f"{name}__llr"][:, :648])
    assert np.array_equal(r["bytes"][:n], ob) and np.array_equal(r["iters"][:n], oi) and np.array_equal(r["ok"][:n], ook)
    assert (r["entry"][n:] == -1).all() and not r["ok"][n:].any() and not r["bytes"][n:].any() and not r["iters"][n:].any()


@pytest.mark.parametrize("name", ["cfg3_qam16_r34", "cfg2_dqpsk_r12"])
def test_demodulator_mirror_process_chunk_fed(name):
    """projectultra_amd.OFDMDemodulator.process() fed 960 samples per call, as the harnesses feed the
    reference (tools/test_nvis_mode.cpp:88-99): same sync offset, coarse CFO and soft bits as the compiled
    reference produced for the whole frames of tests/golden/fullsync.npz."""
    from projectultra_amd import OFDMDemodulator
    from _util import modem_config_from_c
    g = np.load(GOLDEN / "fullsync.npz")
    cfg = cfg_from_array(g[f"{name}__cfg"])
    mc, kw = modem_config_from_c(cfg)
    for a, meta, cfo, want in zip(g[f"{name}__audio"], g[f"{name}__meta"], g[f"{name}__cfo"], g[f"{name}__llr"]):
        d = OFDMDemodulator(mc, n_data_symbols=kw.get("n_data_symbols"))
        ready = False
        for i in range(0, a.size, 960):
            ready = d.process(a[i:i + 960])
        assert ready and d.isSynced()
        assert d.getLastSyncOffset() == meta[1]
        assert np.float32(d._cfo_hz).tobytes() == np.float32(cfo).tobytes()
        soft = np.concatenate([d.getSoftBits(), d.getSoftBits()])
        assert beq(soft[: want.size], want[: soft.size]) and soft.size >= 648


@pytest.mark.parametrize("fft,mod,rate", [(1024, "QAM16", "R3_4"), (512, "DQPSK", "R1_2")])
def test_acquisition_long_streams_with_trims(oracle, fft, mod, rate):
    """Streams past 40000 samples (buffer trims, failed LTS confirmations on a tone) on the GPU == oracle
    (== compiled reference: tests/test_oracle_vs_ref.py::test_acquisition_long_streams_with_trims)."""
    cfg = make_config(fft, mod, rate)
    streams = long_acquisition_streams(oracle, cfg, np.random.default_rng(21))
    n = min(x.size for x in streams)
    audio = np.stack([x[:n] for x in streams])
    ctx = context_for(cfg)
    r = ctx.acquire(audio, 960)
    ctx.synchronize()
    r = {k: v.cpu().numpy() for k, v in r.items()}
    hits = 0
    for i in range(audio.shape[0]):
        o = oracle.acquire(cfg, audio[i], 960)
        assert r["found"][i] == o["found"], (i, o)
        if o["found"]:
            hits += 1
            assert r["data_start"][i] == o["data_start"] and r["sync_offset"][i] == o["sync_offset"], (i, o)
            assert r["fed_at_sync"][i] == o["fed_at_sync"]
            assert np.float32(r["cfo_hz"][i]).tobytes() == np.float32(o["coarse_cfo"]).tobytes()
    assert hits >= 1


def test_chirp_sync_matches_oracle(oracle):
    """Scope row f4: OFDMChirpWaveform::detectSync (dual-chirp detection, ofdm_chirp_waveform.cpp:129-172,
    chirp_sync.hpp:349-505) for a batch of buffers — detected flag, training start, chirp positions exact;
    CFO and correlation bitwise."""
    import torch
    from _util import chirp_streams
    cfg = make_config(512, "DQPSK", "R1_2", entry=1)
    streams = chirp_streams(oracle, cfg, np.random.default_rng(8))
    streams.append(np.zeros(60000, np.float32))                      # silence: zero denominators
    streams.append(streams[0][:40000])                               # shorter than two chirps and a gap
    hits = 0
    eng = context_for(cfg)
    for thr in (0.15, 0.6):
        for x in streams:
            o = oracle.chirp_detect(x, threshold=thr)
            g = {k: v.cpu().numpy()[0] for k, v in eng.chirp_sync(torch.from_numpy(x[None, :]).cuda(), threshold=thr).items()}
            assert int(g["detected"]) == o["success"], (o, g)
            assert int(g["start_sample"]) == o["start_sample"], (o, g)
            assert int(g["up_chirp_start"]) == o["up_chirp_start"] and int(g["down_chirp_start"]) == o["down_chirp_start"], (o, g)
            assert np.float32(g["cfo_hz"]).tobytes() == np.float32(o["cfo_hz"]).tobytes(), (o, g)
            want = max(np.float32(o["up_correlation"]), np.float32(o["down_correlation"]))
            assert np.float32(g["correlation"]).tobytes() == np.float32(want).tobytes(), (o, g)
            hits += o["success"]
    assert hits >= 4
    # batched: equal-length buffers in one launch give the same per-stream answers
    n = min(len(x) for x in streams[:5])
    batch = np.stack([x[:n] for x in streams[:5]] * 40)
    g = {k: v.cpu().numpy() for k, v in eng.chirp_sync(torch.from_numpy(batch).cuda()).items()}
    for i in range(5):
        o = oracle.chirp_detect(batch[i])
        for r in range(i, len(batch), 5):
            assert int(g["detected"][r]) == o["success"] and int(g["start_sample"][r]) == o["start_sample"]
            assert np.float32(g["cfo_hz"][r]).tobytes() == np.float32(o["cfo_hz"]).tobytes()


def test_chirp_receive_from_raw_audio(oracle):
    """Row f4 end to end: dual-chirp detection -> PRESYNCED demodulation from the training start with the chirp
    CFO and its accumulated phase -> LDPC decode, against the oracle run stage by stage on each stream."""
    import torch
    from _util import chirp_initial_phase, chirp_streams
    cfg = make_config(512, "DQPSK", "R1_2", entry=1)
    g = geometry(cfg)
    rng = np.random.default_rng(8)
    streams = chirp_streams(oracle, cfg, rng, n=8)
    n = max(len(x) for x in streams)
    audio = np.stack([np.concatenate([x, rng.normal(0, 1e-3, n - len(x)).astype(np.float32)]) for x in streams])
    audio[6, -4000:] = 0.0
    audio[6] = np.roll(audio[6], 6000)                    # frame runs past the end of the buffer: detected but unusable
    ctx = context_for(cfg)
    r = {k: v.cpu().numpy() for k, v in ctx.chirp_receive(audio, want_llr=True).items()}
    usable = 0
    for i, x in enumerate(audio):
        o = oracle.chirp_detect(x)
        ok = o["success"] and o["start_sample"] + g.frame_samples <= n
        if not ok:
            assert r["entry"][i] == -1 and r["ok"][i] == 0 and r["iters"][i] == 0 and not r["bytes"][i].any(), (i, o)
            continue
        usable += 1
        s = o["start_sample"]
        assert r["entry"][i] == s and np.float32(r["cfo_hz"][i]).tobytes() == np.float32(o["cfo_hz"]).tobytes()
        ph = chirp_initial_phase(o["cfo_hz"], s, cfg.sample_rate)
        want = oracle.demod_decode_batch(cfg, x[s:s + g.frame_samples][None, :], cfo_hz=[o["cfo_hz"]], cfo_phase=[ph])
        _check_llr(r["llr"][i], want["llr"][0], f"chirp receive stream {i}")
        assert np.array_equal(r["bytes"][i], want["bytes"][0]) and r["iters"][i] == want["iters"][0] and r["ok"][i] == want["ok"][0]
    assert usable >= 6 and r["ok"].sum() >= 5                # every SNR but the 3 dB stream decodes, CFO or not


def test_waveform_mirror_detect_sync_and_process(oracle):
    """HipOfdmWaveform used the way the harness uses IWaveform (tools/test_nvis_mode.cpp): detectSync ->
    setFrequencyOffset(result.cfo_hz) -> process(samples from start_sample) -> getSoftBits; against
    oracle.chirp_detect + oracle.demod_presynced on the same samples."""
    from _util import chirp_initial_phase, chirp_streams, modem_config_from_c
    from projectultra_amd.waveform import HipOfdmWaveform, SyncResult
    cfg = make_config(512, "DQPSK", "R1_2", entry=1)
    mc, _ = modem_config_from_c(cfg)
    w = HipOfdmWaveform(mc)
    for x in chirp_streams(oracle, cfg, np.random.default_rng(21), n=3):
        o = oracle.chirp_detect(x, threshold=0.3)
        res = SyncResult()
        assert w.detectSync(x, res) == bool(o["success"])                  # IWaveform's default threshold 0.3
        assert np.float32(res.cfo_hz).tobytes() == np.float32(o["cfo_hz"]).tobytes() and res.has_training
        if not o["success"]:
            continue
        assert res.start_sample == o["start_sample"]
        w.setFrequencyOffset(res.cfo_hz)
        assert w.process(x[res.start_sample:])
        want, _, _ = oracle.demod_presynced(cfg, x[res.start_sample:], o["cfo_hz"],
                                            chirp_initial_phase(o["cfo_hz"], o["start_sample"], cfg.sample_rate))
        _check_llr(w.getSoftBits(), want, "waveform mirror")
        w.reset()


def test_sync_golden_reference():
    """Acquisition (row f1) and chirp synchronisation (row f4) on the GPU against the compiled reference's
    fixtures in tests/golden/sync.npz."""
    import torch
    g = np.load(GOLDEN / "sync.npz")
    for name in ("cfg3", "cfg2"):
        cfg = cfg_from_array(g[f"acq_{name}__cfg"])
        ctx = context_for(cfg)
        r = {k: v.cpu().numpy() for k, v in ctx.acquire(g[f"acq_{name}__audio"], 960).items()}
        ints, cfo = g[f"acq_{name}__ints"], g[f"acq_{name}__cfo"]
        assert np.array_equal(r["found"], ints[:, 0])
        ok = ints[:, 0] == 1
        assert np.array_equal(r["fed_at_sync"][ok], ints[ok, 1]) and np.array_equal(r["sync_offset"][ok], ints[ok, 2])
        assert np.array_equal(r["data_start"][ok], ints[ok, 4]) and beq(r["cfo_hz"][ok], cfo[ok])
    ctx = context_for(make_config(512, "DQPSK", "R1_2", entry=1))
    for i in range(4):
        x = g[f"chirp{i}__audio"]
        q = {k: v.cpu().numpy()[0] for k, v in ctx.chirp_sync(torch.from_numpy(x[None, :]).cuda()).items()}
        ints, fl = g[f"chirp{i}__ints"], g[f"chirp{i}__floats"]
        assert [int(q["detected"]), int(q["up_chirp_start"]), int(q["down_chirp_start"]), int(q["start_sample"])] == list(ints)
        assert np.float32(q["cfo_hz"]).tobytes() == fl[0].tobytes()
        assert np.float32(q["correlation"]).tobytes() == max(fl[1], fl[2]).tobytes()


@pytest.mark.parametrize("fft,mod,rate,kw", [(1024, "QAM16", "R3_4", {}), (512, "DQPSK", "R1_2", {}), (1024, "DQPSK", "R1_4", dict(pilot_spacing=2, use_pilots=1)),
                                             (512, "QPSK", "R2_3", {}), (1024, "QAM32", "R1_2", dict(entry=1))])
def test_streamed_symbols_equal_the_batch(oracle, fft, mod, rate, kw):
    """ultra_hip_demod_stream_batch: the symbols of a frame demodulated as they arrive — in any split — give the LLRs and the
    final tracker of one batch call (and of the oracle): deferred and per-symbol carrier halves, zero-CFO and rotating
    layouts, the presynced entry with its training symbols, with and without initial offsets."""
    cfg = make_config(fft, mod, rate, n_data_symbols=9, **kw)
    g = geometry(cfg)
    n = 96
    audio, _ = oracle.make_batch(cfg, n, seed=77, channel="watterson", snr_db=24.0)
    rng = np.random.default_rng(5)
    n_train = int(cfg.training_symbols)
    total = n_train + 9
    for with_cfo in (False, True):
        cfo = rng.normal(0, 6.0, n).astype(np.float32) if with_cfo else None
        ph = rng.uniform(-3, 3, n).astype(np.float32) if (with_cfo and n_train) else None
        want = oracle.demod_decode_batch(cfg, audio, cfo_hz=cfo, cfo_phase=ph, n_threads=16)
        ctx = context_for(cfg)
        whole, st_whole = ctx.demod(audio, cfo_hz=cfo, cfo_phase=ph, want_state=True)
        assert beq(whole.cpu().numpy(), want["llr"])
        first = max(n_train, 1)                                       # the training symbols of the presynced entry go in one call
        for split in ([total], [first] + [1] * (total - first), [first, 3, total - first - 3], [total - 1, 1]):
            ctx2 = context_for(cfg)
            parts, s0 = [], 0
            for k in split:
                a = audio[:, s0 * g.symbol_samples:(s0 + k) * g.symbol_samples]
                llr, st = ctx2.demod_stream(np.ascontiguousarray(a), s0, k, cfo_hz=cfo if s0 == 0 else None,
                                            cfo_phase=ph if s0 == 0 else None, want_state=True)
                parts.append(llr.cpu().numpy()); s0 += k
            got = np.concatenate([p for p in parts if p.shape[1]], axis=1)
            assert beq(got, want["llr"]), (mod, with_cfo, split)
            assert beq(st.cpu().numpy(), st_whole.cpu().numpy()), (mod, with_cfo, split)


@pytest.mark.parametrize("fft,mod,rate,kw", [(1024, "QAM16", "R3_4", {}), (512, "DQPSK", "R1_2", {}), (1024, "DQPSK", "R1_4", dict(pilot_spacing=2, use_pilots=1)),
                                             (1024, "QAM32", "R1_2", dict(entry=1))])
def test_streamed_equalized_symbols(oracle, fft, mod, rate, kw):
    """ultra_hip_demod_stream_batch_eq (ABI 9): the equalized data carriers of every data symbol — what demodulateSymbol
    appends to constellation_symbols (demodulator.cpp:199-208).  The values themselves are held to the compiled reference's
    getConstellationSymbols() by the scripted harness (tests/test_gpu_pimpl.py: the ring's digest in every log line); here:
    the call's LLRs and tracker are those of the plain stream call (= the oracle's), the rows do not depend on how the frame
    is split into calls, and padding beyond the data carriers is not written."""
    cfg = make_config(fft, mod, rate, n_data_symbols=7, **kw)
    g = geometry(cfg)
    n = 40
    audio, _ = oracle.make_batch(cfg, n, seed=91, channel="watterson", snr_db=22.0)
    cfo = np.random.default_rng(8).normal(0, 5.0, n).astype(np.float32)
    n_train = int(cfg.training_symbols)
    total = n_train + 7
    want = oracle.demod_decode_batch(cfg, audio, cfo_hz=cfo, n_threads=16)
    first = max(n_train, 1)
    rows = {}
    for split in ([total], [first] + [1] * (total - first), [first, 2, total - first - 2]):
        ctx = context_for(cfg)
        parts, eqs, s0 = [], [], 0
        for k in split:
            a = np.ascontiguousarray(audio[:, s0 * g.symbol_samples:(s0 + k) * g.symbol_samples])
            llr, st, eq = ctx.demod_stream(a, s0, k, cfo_hz=cfo if s0 == 0 else None, want_state=True, want_equalized=True)
            parts.append(llr.cpu().numpy()); eqs.append(eq.cpu().numpy()); s0 += k
        got = np.concatenate([p for p in parts if p.shape[1]], axis=1)
        assert beq(got, want["llr"]), (mod, split)
        rows[tuple(split)] = np.concatenate([e for e in eqs if e.shape[1]], axis=1)
        assert rows[tuple(split)].shape == (n, 7, g.n_data_carriers)
    ref = rows[(total,)]
    for k, v in rows.items():
        assert beq(v, ref), (mod, k)
    assert np.isfinite(ref.view(np.float32)).all() and np.abs(ref).mean() > 0.1      # symbols, not zeros


def test_stream_and_block_entries_refuse_bad_arguments():
    """The round-3 entries validate before they launch: a resume without a previous call, symbol ranges outside the frame,
    rows too short for the symbols, block runs that overlap or leave the LLR array."""
    import torch
    from projectultra_amd import UltraHipError
    cfg = make_config(1024, "QAM16", "R3_4")
    g = geometry(cfg)
    ctx = context_for(cfg)
    a = torch.zeros((4, g.frame_samples), dtype=torch.float32, device="cuda")
    with pytest.raises(UltraHipError):
        ctx.demod_stream(a[:, :g.symbol_samples], 2, 1)                   # nothing to continue from
    with pytest.raises(UltraHipError):
        ctx.demod_stream(a, 3, 2)                                         # symbols 3, 4 of a 4-symbol frame
    with pytest.raises(UltraHipError):
        ctx.demod_stream(a[:, :g.symbol_samples], 0, 2)                   # rows hold one symbol
    llr = torch.zeros((64, 768), dtype=torch.float32, device="cuda")
    with pytest.raises(UltraHipError):
        ctx.ldpc_decode_blocks(llr, 16, 8, 4)                             # runs overlap
    with pytest.raises(UltraHipError):
        ctx.ldpc_decode_blocks(llr, 16, 32, 3)                            # the third run leaves the array
    with pytest.raises(UltraHipError):
        ctx.demod_into(a, llr[:4, :600])                                  # rows shorter than llrs_per_frame
    r = ctx.ldpc_decode_blocks(llr, 16, 32, 2)
    assert r["ok"].shape[0] == 32
    # ultra_hip_stream_adopt: records travel between the two entries of ONE layout only, and only records that exist
    ps = context_for(make_config(1024, "QAM16", "R3_4", entry=1))
    other = context_for(make_config(512, "DQPSK", "R1_2"))
    with pytest.raises(UltraHipError):
        ctx.adopt_tracker(other)                                          # another carrier layout
    with pytest.raises(UltraHipError):
        ctx.adopt_tracker(ctx)                                            # itself
    with pytest.raises(UltraHipError):
        ctx.adopt_tracker(ps, 4)                                          # the presynced context has demodulated nothing yet
    a_ps = torch.zeros((4, geometry(make_config(1024, "QAM16", "R3_4", entry=1)).frame_samples), dtype=torch.float32, device="cuda")
    ps.demod(a_ps)
    ctx.adopt_tracker(ps, 4)
    # the mid-frame check: a window that does not reach n_samples, samples fed before the window's origin; a buffer shorter
    # than six preamble symbols is not an error — nothing is found
    resume = torch.tensor([[100, 9000, 0, 0]] * 4, dtype=torch.int32, device="cuda")
    with pytest.raises(UltraHipError):
        ctx.acquire_stream(a[:, :1000], 100, 9000, resume, midframe=True)   # 8900 samples claimed, rows hold 1000
    with pytest.raises(UltraHipError):
        ctx.acquire_stream(a, 500, 400, resume, midframe=True)              # n_samples < origin
    short = ctx.acquire_stream(a[:, :3000], 100, 3100, resume, midframe=True)
    ctx.synchronize()
    assert not short["found"].cpu().numpy().any()


@pytest.mark.parametrize("name", ["cfg3_qam16_r34", "cfg2_dqpsk_r12"])
def test_midframe_preamble_check_equals_oracle(oracle, name):
    """ultra_hip_resync_stream_batch — the preamble check of the SYNCED state (demodulator.cpp:605-657) — for a batch of
    streams whose buffers start at different absolute positions, against uo_midframe_search (pinned to process() itself in
    tests/test_oracle_vs_ref.py::test_midframe_search and by tests/golden/stream.npz): found, data start, Schmidl-Cox
    offset exactly, coarse CFO bitwise; records are not written."""
    import torch
    from _util import midframe_buffers, modem_config_from_c
    from projectultra_amd import ReceiveContext
    g = np.load(GOLDEN / "fullsync.npz")
    cfg = cfg_from_array(g[f"{name}__cfg"])
    mc, _ = modem_config_from_c(cfg)
    geo = geometry(cfg)
    meta = g[f"{name}__meta"][0]
    bufs = midframe_buffers(g[f"{name}__audio"], int(meta[0]), geo.symbol_samples, int(meta[1]), seed=21, n=18)
    ctx = ReceiveContext(mc)
    origin = 5000                                              # the window starts here; every stream's buffer a bit later
    base = [origin + 37 * i for i in range(len(bufs))]
    n_samples = max(b + x.size for b, x in zip(base, bufs))
    audio = np.zeros((len(bufs), n_samples - origin), np.float32)
    for i, (b, x) in enumerate(zip(base, bufs)):
        audio[i, :b - origin] = 0.25                           # before rx_buffer: must not be looked at
        x = x[:n_samples - b] if i % 5 else x                  # all buffers end at n_samples (rows are padded by repeating)
        audio[i, b - origin: b - origin + x.size] = x
        bufs[i] = audio[i, b - origin:].copy()                 # what the stream's rx_buffer really holds
    resume = torch.tensor([[b, n_samples, 0x3f000000, 0] for b in base], dtype=torch.int32, device=ctx.device)
    keep = resume.clone()
    r = ctx.acquire_stream(audio, origin, n_samples, resume, midframe=True)
    ctx.synchronize()
    assert torch.equal(resume, keep)
    r = {k: v.cpu().numpy() for k, v in r.items()}
    hits = 0
    for i, (b, x) in enumerate(zip(base, bufs)):
        o = oracle.midframe_search(cfg, x)
        assert r["found"][i] == o["found"], (i, o)
        if o["found"]:
            hits += 1
            assert r["data_start"][i] == b + o["consume"] and r["sync_offset"][i] == o["sts_start"], (i, o, r["data_start"][i])
            assert np.float32(r["cfo_hz"][i]).tobytes() == np.float32(o["coarse_cfo"]).tobytes(), (i, o)
    assert 6 <= hits < len(bufs)


@pytest.mark.parametrize("name", ["cfg3_qam16_r34", "cfg2_dqpsk_r12"])
def test_demodulator_mirror_live_stream(name):
    """projectultra_amd.OFDMDemodulator.process() as a live stream (the Python twin of HipOfdmCoxWaveform) against the compiled
    reference call by call (tests/golden/stream.npz): frame-complete exit and re-acquisition, idle exit, 250-symbol timeout."""
    from projectultra_amd import OFDMDemodulator
    from _util import STREAM_SCENARIOS, build_stream, modem_config_from_c
    g = np.load(GOLDEN / "fullsync.npz")
    want = np.load(GOLDEN / "stream.npz")
    cfg = cfg_from_array(g[f"{name}__cfg"])
    mc, _ = modem_config_from_c(cfg)
    geo = geometry(cfg)
    pre = int(g[f"{name}__meta"][0][0])
    for sc, recipe in STREAM_SCENARIOS.items():
        audio, chunks = build_stream(g[f"{name}__audio"], recipe(geo.symbol_samples, pre))
        d = OFDMDemodulator(mc)
        pos, ready, synced, drained, soft = 0, [], [], [], []
        for c in chunks:
            r = d.process(audio[pos:pos + int(c)]); pos += int(c)
            sb = d.getSoftBits() if r else np.zeros(0, np.float32)      # OFDMNvisWaveform::process
            ready.append(int(r)); synced.append(int(d.isSynced())); drained.append(sb.size); soft.append(sb)
        assert ready == want[f"{name}__{sc}__ready"].tolist(), (name, sc)
        assert synced == want[f"{name}__{sc}__synced"].tolist(), (name, sc)
        assert drained == want[f"{name}__{sc}__drained"].tolist(), (name, sc)
        assert beq(np.concatenate(soft).astype(np.float32), want[f"{name}__{sc}__soft"]), (name, sc)
        # getConstellationSymbols() (demodulator.cpp:827-830): the newest <= 500 equalized data carriers of the symbols demodulated so
        # far (the values themselves are held to the compiled reference by the C++ harness, tests/test_gpu_pimpl.py)
        ring = d.getConstellationSymbols()
        assert ring.dtype == np.complex64 and ring.size <= 500 and (ring.size > 0 or sum(drained) == 0), (name, sc, ring.size)
        assert np.isfinite(ring.view(np.float32)).all()


def test_headline_batch_whole(oracle):
    """north_star's 2^20-frame cfg3 batch (Watterson good channel, 30 dB), checked where one launch over a million frames
    can go wrong and a 16,384-frame prefix cannot show it: the LAST 2,048 frames (tail workgroups of every kernel), 2,048
    frames straddling byte offset 2^32 of the audio buffer (frame 239,674: 32-bit offset arithmetic fails there), 2,048 drawn
    over all of it — soft bits bitwise, bytes, iterations, status against the oracle — and the batch's eight counters equal to
    the sum over four 2^18-frame batches generated separately with first_frame offsets (generators keyed on the global frame
    index; what a rank's shard of the strong-scaling run computes)."""
    import torch
    from projectultra_amd import CodeRate, Modulation, ReceiveContext, presets
    cfg = make_config(1024, "QAM16", "R3_4")
    mc = presets.nvis_mode().with_mode(Modulation.QAM16, CodeRate.R3_4)
    mc.pilot_spacing = 4
    ctx = ReceiveContext(mc)
    n = 1 << 20
    chan = dict(seed=0x5EED, channel="watterson", snr_db=30.0, delay_ms=0.5, doppler_hz=0.1)
    audio, payload = ctx.make_batch(n, first_frame=0, **chan)
    assert audio.numel() * 4 > (1 << 32)
    r = ctx.demod_decode(audio, want_llr=True)
    counters = ctx.count_errors(r, payload).cpu().numpy()
    ctx.synchronize()
    assert counters[0] == n and 0 < counters[1] < n

    edge = (1 << 32) // (audio.shape[1] * 4)
    rng = np.random.default_rng(20)
    idx = np.unique(np.concatenate([np.arange(n - 2048, n), np.arange(edge - 1024, edge + 1024), rng.choice(n, 2048, replace=False)]))
    t_idx = torch.from_numpy(idx).cuda()
    a = audio[t_idx].cpu().numpy()
    want = oracle.demod_decode_batch(cfg, a, n_threads=16)
    _check_llr(r["llr"][t_idx].cpu().numpy(), want["llr"], "2^20 batch, stratified frames")
    for k in ("bytes", "iters", "ok"):
        assert np.array_equal(r[k][t_idx].cpu().numpy(), want[k]), k
    # the device generator's frames at those global indices are the frames a separate call with first_frame makes
    a2, p2 = ctx.make_batch(2048, first_frame=n - 2048, **chan)
    assert torch.equal(a2, audio[n - 2048:]) and torch.equal(p2, payload[n - 2048:])
    del a2, p2

    total = np.zeros(8, np.int64)
    q = n // 4
    for part in range(4):
        pa, pp = ctx.make_batch(q, first_frame=part * q, **chan)
        assert torch.equal(pa[-64:], audio[(part + 1) * q - 64:(part + 1) * q])
        pr = ctx.demod_decode(pa)
        total += ctx.count_errors(pr, pp).cpu().numpy()
        for k in ("bytes", "iters", "ok"):
            assert torch.equal(pr[k], r[k][part * q:(part + 1) * q]), (part, k)
        del pa, pp, pr
    assert np.array_equal(total, counters), (total, counters)


def test_set_cfo_between_stream_calls_golden():
    """ultra_hip_demod_stream_set_cfo = OFDMDemodulator::setFrequencyOffset between two process() calls
    (/root/reference/src/ofdm/demodulator.cpp:805-814): the compiled reference's LLRs (tests/golden/setcfo.npz) for frames that
    started WITHOUT an offset — where the batch runs its zero-offset short cuts until the call — and with one, on a layout
    without pilots (cfg2) and with pilots (cfg3 layout, 8 symbols), the new offset arriving before symbol 1, 2 or 5."""
    g = np.load(GOLDEN / "setcfo.npz")
    for key in g["cases"]:
        key = str(key)
        name = key.split("__")[0]
        cfg = cfg_from_array(g[f"{name}__cfg"])
        geo = geometry(cfg)
        set_at = int(key.split("__at")[1].split("_")[0])
        has0 = key.endswith("has1")
        audio, cfo0, cfon, want = g[f"{key}__audio"], g[f"{key}__cfo0"], g[f"{key}__cfo_new"], g[f"{key}__llr"]
        nsym = geo.frame_samples // geo.symbol_samples
        ctx = context_for(cfg)
        n = audio.shape[0]
        first = ctx.demod_stream(np.ascontiguousarray(audio[:, :set_at * geo.symbol_samples]), 0, set_at, cfo_hz=cfo0 if has0 else None)
        for f in range(n):
            ctx.demod_stream_set_cfo(f, float(cfon[f]))
        rest = ctx.demod_stream(np.ascontiguousarray(audio[:, set_at * geo.symbol_samples:]), set_at, nsym - set_at)
        got = np.concatenate([first.cpu().numpy(), rest.cpu().numpy()], axis=1)
        _check_llr(got, want, key)


@pytest.mark.parametrize("mod,rate,fft", [("QAM16", "R3_4", 1024), ("QPSK", "R1_2", 512), ("QAM64", "R3_4", 512)])
def test_degenerate_audio_vs_oracle(oracle, mod, rate, fft):
    """Audio far outside the normal range — silence, amplitudes of 1e-19 and 1e+17 (the channel powers leave the range in which
    the deep-fade erasure's unordered screen sum is trusted: the serial sum of equalize_demap decides), a NaN and an infinity
    inside a frame: soft bits against the oracle, NaNs in the same places and everything else bit for bit."""
    cfg = make_config(fft, mod, rate)
    n = 48
    audio, _ = oracle.make_batch(cfg, n, seed=0xDE6, channel="watterson", snr_db=20.0)
    audio[0:4] = 0.0
    audio[4:12] *= np.float32(1e-19)
    audio[12:20] *= np.float32(1e17)
    audio[20:24, 1000] = np.nan
    audio[24:28, 3000] = np.inf
    audio[28:32, ::2] = 0.0
    cfo = np.where(np.arange(n) % 2 == 0, 0.0, 2.5).astype(np.float32)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = oracle.demod_decode_batch(cfg, audio, cfo_hz=cfo, n_threads=8, decode=False)["llr"]
    ctx = context_for(cfg)
    got = ctx.demod(audio, cfo_hz=cfo).cpu().numpy()
    nan_w, nan_g = np.isnan(want), np.isnan(got)
    assert np.array_equal(nan_w, nan_g), (mod, np.flatnonzero((nan_w != nan_g).any(axis=1)))
    same = (got.view(np.uint32) == want.view(np.uint32)) | nan_w
    assert same.all(), (mod, "frames", np.flatnonzero(~same.all(axis=1)))
