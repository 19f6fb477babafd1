"""GPU stimulus generator (scope row f2, ultra_hip_make_batch) against the oracle's generator
(oracle/ultra_oracle.c uo_make_batch: pinned modulator / encoder restatements, same counter-based
payload stream).  Channel "none": audio and payloads BITWISE.  Channels: statistics."""
import numpy as np
import pytest

from _util import INFO_BITS, beq, context_for, geometry, make_config

pytestmark = pytest.mark.gpu

MODES = [(1024, "QAM16", "R3_4", {}), (512, "DQPSK", "R1_2", {}), (512, "QPSK", "R1_2", {}), (1024, "QAM32", "R3_4", {}),
         (1024, "D8PSK", "R3_4", dict(pilot_spacing=2)), (512, "DBPSK", "R1_4", {}), (512, "BPSK", "R1_2", {}),
         (512, "QAM64", "R3_4", {}), (512, "QAM256", "R5_6", {}), (1024, "QAM16", "R2_3", dict(n_data_symbols=12))]


@pytest.mark.parametrize("fft,mod,rate,kw", MODES)
def test_clean_frames_are_bit_identical_to_the_oracle(oracle, fft, mod, rate, kw):
    cfg = make_config(fft, mod, rate, **kw)
    ctx = context_for(cfg)
    n, seed, f0 = 40, 0xABCDEF, 1000
    audio, payload = ctx.make_batch(n, seed=seed, first_frame=f0, channel="none")
    ctx.synchronize()
    want_a, want_p = oracle.make_batch(cfg, n, seed=seed, f0=f0, channel="none")
    assert np.array_equal(payload.cpu().numpy(), want_p)
    got = audio.cpu().numpy()
    if not beq(got, want_a):
        bad = np.argwhere(got.view(np.uint32) != want_a.view(np.uint32))
        raise AssertionError(f"{len(bad)} samples differ, first {bad[:4].tolist()}, max abs err {np.abs(got - want_a).max():g}")


def test_awgn_level_and_receive_statistics(oracle):
    """AWGN: the added noise has the power the harness formula asks for; frames decode like the oracle's."""
    cfg = make_config(1024, "QAM16", "R3_4")
    ctx = context_for(cfg)
    n = 2048
    clean, payload = ctx.make_batch(n, seed=7, channel="none")
    # the harness sets the noise from the mean power of the WHOLE signal (preamble with its leading silence
    # included, tools/test_nvis_mode.cpp:78-86); the frame part alone is a little stronger
    g = geometry(cfg)
    off = []
    for t in range(8):
        pl = bytes(np.random.default_rng(t).integers(0, 256, 2 * (INFO_BITS[cfg.code_rate] // 8), dtype=np.uint8))
        whole, pre = oracle.modulate_frame(cfg, oracle.ldpc_encode(int(cfg.code_rate), pl))
        off.append(np.mean(whole[pre: pre + g.frame_samples].astype(np.float64) ** 2) / np.mean(whole.astype(np.float64) ** 2))
    offset_db = 10 * np.log10(np.mean(off))
    for snr_db in (30.0, 14.0):
        noisy, _ = ctx.make_batch(n, seed=7, channel="awgn", snr_db=snr_db)
        ctx.synchronize()
        d = (noisy - clean).double()
        # whole-signal power incl. preamble: estimate it from the oracle's generator on a few frames
        ratio = (clean.double() ** 2).mean().item() / (d ** 2).mean().item()
        assert abs(10 * np.log10(ratio) - (snr_db + offset_db)) < 0.25, (snr_db, offset_db, 10 * np.log10(ratio))
        assert abs(d.mean().item()) < 4 * d.std().item() / np.sqrt(d.numel())
        z = d / d.std(dim=1, keepdim=True)                    # the level is per frame (each frame's own power)
        kurt = (z ** 4).mean().item() / (z ** 2).mean().item() ** 2
        assert abs(kurt - 3.0) < 0.05, kurt
    r = ctx.demod_decode(noisy)
    c = ctx.count_errors(r, payload).cpu().numpy()
    oa, op = oracle.make_batch(cfg, 512, seed=7, channel="awgn", snr_db=14.0)
    ro = ctx.demod_decode(oa)
    co = ctx.count_errors(ro, op).cpu().numpy()
    fer_gpu, fer_cpu = c[1] / c[0], co[1] / co[0]
    assert abs(fer_gpu - fer_cpu) < 0.08, (fer_gpu, fer_cpu)


def test_watterson_statistics_match_the_oracle_generator(oracle):
    """Two-tap fading channel: output power, and the error rates of the receive path, agree with frames
    drawn by the oracle's serial generator within sampling tolerance."""
    cfg = make_config(1024, "QAM16", "R3_4")
    ctx = context_for(cfg)
    n = 4096
    audio, payload = ctx.make_batch(n, seed=99, channel="watterson", snr_db=30.0)
    ctx.synchronize()
    oa, op = oracle.make_batch(cfg, 1024, seed=99, channel="watterson", snr_db=30.0)
    p_gpu = (audio.double() ** 2).mean(dim=1).cpu().numpy()
    p_cpu = (oa.astype(np.float64) ** 2).mean(axis=1)
    assert abs(p_gpu.mean() / p_cpu.mean() - 1.0) < 0.06, (p_gpu.mean(), p_cpu.mean())
    assert abs(p_gpu.std() / p_cpu.std() - 1.0) < 0.25, (p_gpu.std(), p_cpu.std())
    r = ctx.demod_decode(audio)
    c = ctx.count_errors(r, payload).cpu().numpy()
    ro = ctx.demod_decode(oa)
    co = ctx.count_errors(ro, op).cpu().numpy()
    fer_gpu, fer_cpu = c[1] / c[0], co[1] / co[0]
    assert abs(fer_gpu - fer_cpu) < 0.04, (fer_gpu, fer_cpu)
    it_gpu, it_cpu = c[5] / c[0], co[5] / co[0]
    assert abs(it_gpu - it_cpu) < 2.0, (it_gpu, it_cpu)
    # single faded path (no delay): WattersonChannel drops the tap gains (hf_channel.hpp:131-151)
    a0, p0 = ctx.make_batch(n, seed=5, channel="watterson", snr_db=30.0, delay_ms=0.0)
    oa0, op0 = oracle.make_batch(cfg, 1024, seed=5, channel="watterson", snr_db=30.0, delay_ms=0.0)
    pg, pc = (a0.double() ** 2).mean().item(), (oa0.astype(np.float64) ** 2).mean()
    assert abs(pg / pc - 1.0) < 0.06, (pg, pc)
    c0 = ctx.count_errors(ctx.demod_decode(a0), p0).cpu().numpy()
    co0 = ctx.count_errors(ctx.demod_decode(oa0), op0).cpu().numpy()
    assert abs(c0[1] / c0[0] - co0[1] / co0[0]) < 0.04


@pytest.mark.parametrize("name,delay_ms,doppler_hz,snr_db", [("moderate", 1.0, 0.5, 20.0), ("poor", 2.0, 1.0, 20.0), ("flutter", 0.5, 10.0, 20.0)])
def test_watterson_presets_match_the_oracle_generator(oracle, name, delay_ms, doppler_hz, snr_db):
    """The reference's other Watterson presets (itu_r_f1487::moderate / poor / flutter, src/sim/hf_channel.hpp:421-470:
    1 ms / 0.5 Hz, 2 ms / 1 Hz, 0.5 ms / 10 Hz) on the device generator: at 10 Hz Doppler the fading moves through 2.6
    periods inside one frame, which is what the chunk-parallel evaluation of the fading filters has to get right.  8192
    device frames against 2048 frames of the oracle's serial generator: mean and spread of the per-frame power, the
    receive path's FER and mean BP iterations, and the fading's rate of change (lag-1-symbol correlation of the symbol
    powers) within sampling tolerance."""
    cfg = make_config(1024, "QAM16", "R1_2", pilot_spacing=4)
    ctx = context_for(cfg)
    n, n_cpu = 8192, 2048
    audio, payload = ctx.make_batch(n, seed=321, channel="watterson", snr_db=snr_db, delay_ms=delay_ms, doppler_hz=doppler_hz)
    ctx.synchronize()
    oa, op = oracle.make_batch(cfg, n_cpu, seed=321, channel="watterson", snr_db=snr_db, delay_ms=delay_ms, doppler_hz=doppler_hz)
    a = audio.cpu().numpy().astype(np.float64)
    p_gpu, p_cpu = (a ** 2).mean(axis=1), (oa.astype(np.float64) ** 2).mean(axis=1)
    assert abs(p_gpu.mean() / p_cpu.mean() - 1.0) < 0.05, (name, p_gpu.mean(), p_cpu.mean())
    assert abs(p_gpu.std() / p_cpu.std() - 1.0) < 0.15, (name, p_gpu.std(), p_cpu.std())
    # how fast the envelope moves inside a frame: correlation of consecutive symbols' powers
    S = ctx.geometry.symbol_samples

    def sym_corr(x):
        ps = (x[:, : (x.shape[1] // S) * S].reshape(x.shape[0], -1, S) ** 2).mean(axis=2)
        ps = ps / ps.mean(axis=1, keepdims=True)
        return float(np.mean((ps[:, 1:] - 1.0) * (ps[:, :-1] - 1.0)))
    assert abs(sym_corr(a) - sym_corr(oa.astype(np.float64))) < 0.02 + 0.15 * abs(sym_corr(oa.astype(np.float64))), (name, sym_corr(a), sym_corr(oa.astype(np.float64)))
    c = ctx.count_errors(ctx.demod_decode(audio), payload).cpu().numpy()
    co = ctx.count_errors(ctx.demod_decode(oa), op).cpu().numpy()
    assert abs(c[1] / c[0] - co[1] / co[0]) < 0.035, (name, c[1] / c[0], co[1] / co[0])
    assert abs(c[5] / c[0] - co[5] / co[0]) < 1.5, (name, c[5] / c[0], co[5] / co[0])


CFG5_MODS = [(512, "DBPSK"), (512, "DQPSK"), (1024, "D8PSK"), (1024, "QAM16"), (1024, "QAM32")]
CFG5_RATES = ["R1_4", "R1_2", "R2_3", "R3_4", "R5_6"]


@pytest.mark.parametrize("fft,mod", CFG5_MODS)
def test_cfg5_sweep_device_stimulus_vs_oracle(oracle, fft, mod):
    """BASELINE cfg5 shape: {DBPSK, DQPSK, D8PSK, 16QAM, 32QAM} x {R1/4 .. R5/6}, 2^14 distinct frames per
    combination generated on the device (AWGN), demodulated + decoded + counted on the device.  A random
    subset is copied to the host and pushed through the oracle: LLRs, decoded bytes, iterations and success
    must be bit-identical; the device counters must equal a host recount of the device's own outputs."""
    rng = np.random.default_rng(5)
    for rate in CFG5_RATES:
        cfg = make_config(fft, mod, rate)
        ctx = context_for(cfg)
        g = ctx.geometry
        n = 1 << 14
        audio, payload = ctx.make_batch(n, seed=0xC0FFEE, first_frame=12345, channel="awgn", snr_db=[8.0, 14.0, 20.0][rng.integers(3)])
        r = ctx.demod_decode(audio, want_llr=True)
        c = ctx.count_errors(r, payload).cpu().numpy()
        ctx.synchronize()
        by, ok, it = r["bytes"].cpu().numpy(), r["ok"].cpu().numpy().astype(bool), r["iters"].cpu().numpy()
        pl = payload.cpu().numpy()
        good = ok & (by[:, : pl.shape[1]] == pl).all(axis=1)
        assert c[0] == n and c[1] == n - good.sum() and c[4] == (~ok).sum() and c[5] == it.sum(), (mod, rate, c)
        idx = np.sort(rng.choice(n, 48, replace=False))
        sub = audio[torch_index(idx)].cpu().numpy()
        want = oracle.demod_decode_batch(cfg, sub, n_threads=8, want_llr=True, want_state=False)
        got_llr = r["llr"][torch_index(idx)].cpu().numpy()
        assert beq(got_llr, want["llr"]), (mod, rate)
        assert np.array_equal(by[idx], want["bytes"]) and np.array_equal(it[idx], want["iters"])
        assert np.array_equal(ok[idx], want["ok"].astype(bool))


def torch_index(idx):
    import torch
    return torch.from_numpy(np.sort(idx)).cuda()


@pytest.mark.parametrize("cfo_hz", [20.0, -33.25, 0.0005])
def test_channel_cfo_is_bit_identical_to_the_reference_model(oracle, cfo_hz):
    """the channel's carrier frequency offset on the device.  WattersonChannel::applyCFO
    (hf_channel.hpp:161-232: mix down from 1500 Hz, 48-tap running mean, rotate, mix up; every frame by a fresh channel)
    through ultra_hip_channel_cfo_batch equals the oracle's restatement bit for bit — which tests/test_oracle_vs_ref.py::
    test_channel_cfo pins to the compiled reference; offsets within +-0.001 Hz and rows under 256 samples pass unchanged.
    Then the receive path on the shifted frames (entered without a coarse estimate, so the tracker is far off): GPU == oracle,
    LLRs and tracker state bitwise."""
    import torch
    cfg = make_config(1024, "QAM16", "R3_4")
    ctx = context_for(cfg)
    audio, payload = ctx.make_batch(48, seed=91, channel="awgn", snr_db=24.0)
    got = ctx.channel_cfo(audio, cfo_hz).cpu().numpy()
    a = audio.cpu().numpy()
    want = np.stack([oracle.channel_apply_cfo(row, cfo_hz) if abs(cfo_hz) > 0.001 else row for row in a])
    assert beq(got, want)
    assert bool(beq(got, a)) == (abs(cfo_hz) <= 0.001)
    short = ctx.channel_cfo(audio[:, :200].contiguous(), 20.0).cpu().numpy()         # under 256 samples: unchanged
    assert beq(short, a[:, :200])
    strided = torch.zeros((5, 5000), dtype=torch.float32, device=audio.device)      # row stride != row length
    strided[:, :4480] = audio[:5]
    got_s = ctx.channel_cfo(strided[:, :4480], cfo_hz).cpu().numpy()
    assert beq(got_s, want[:5])
    if abs(cfo_hz) > 0.001:
        shifted, _ = ctx.make_batch(48, seed=91, channel="awgn", snr_db=24.0, cfo_hz=cfo_hz)   # the generator's own argument
        assert beq(shifted.cpu().numpy(), want)
        llr, state = ctx.demod(shifted, want_state=True)
        ctx.synchronize()
        w = oracle.demod_decode_batch(cfg, want, n_threads=8, want_llr=True, want_state=True, decode=False)
        assert beq(llr.cpu().numpy(), w["llr"]) and beq(state.cpu().numpy(), w["state"])


def test_raw_watterson_streams_carry_the_batch_generators_frames():
    """ultra_hip_make_raw_batch_channel(kind 2): the transmission of stream f goes through the SAME channel realisation as
    frame f of ultra_hip_make_batch (same keys, same filters), so the frame part of the raw stream equals that batch's frame
    bit for bit; lead and tail hold noise of the channel's level only; and the whole receive (acquisition -> demodulation ->
    decode) of those streams equals the oracle's on the same samples."""
    import torch
    from _util import context_for, make_config
    from oracle.bindings import oracle as get_oracle
    cfg = make_config(1024, "QAM16", "R3_4")
    ctx = context_for(cfg)
    g = ctx.geometry
    n, lead, tail = 96, 1120, 960
    raw, pay_raw = ctx.make_raw_batch(n, seed=0xFADE, first_frame=40, channel="watterson", snr_db=24.0, lead=lead, tail=tail)
    frames, pay = ctx.make_batch(n, seed=0xFADE, first_frame=40, channel="watterson", snr_db=24.0, delay_ms=0.5, doppler_hz=0.1)
    pre = raw.shape[1] - lead - tail - g.frame_samples
    assert torch.equal(pay_raw, pay)
    assert torch.equal(raw[:, lead + pre:lead + pre + g.frame_samples], frames)
    silence = torch.cat([raw[:, :lead], raw[:, lead + pre + g.frame_samples:]], dim=1)
    tx = raw[:, lead:lead + pre + g.frame_samples]
    ratio = (tx.pow(2).mean() / silence.pow(2).mean()).item()
    assert 10 ** 2.0 < ratio < 10 ** 2.8, ratio                  # 24 dB below the transmission's power, give or take the fading
    r = ctx.receive(raw, chunk=960, want_llr=True)
    o = get_oracle()
    a = raw.cpu().numpy()
    entry = r["entry"].cpu().numpy()
    assert (entry >= 0).mean() > 0.9
    for f in range(0, n, 7):
        acq = o.acquire(cfg, a[f], chunk=960)
        assert (entry[f] >= 0) == bool(acq["found"])
        if acq["found"]:
            assert entry[f] == acq["data_start"]
