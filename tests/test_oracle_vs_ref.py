"""Oracle vs the compiled reference, live (needs oracle/_ref, i.e. the build container or a
box that received the prebuilt .so).  Everything is compared BITWISE, stage by stage."""
import numpy as np
import pytest

from _util import beq, chirp_streams, long_acquisition_streams, nonfinite_cases, valid_special_codewords
from oracle.bindings import INFO_BITS, geometry, make_config


def test_fec(oracle, ref):
    rng = np.random.default_rng(1)
    for rate in range(6):
        pl = bytes(rng.integers(0, 256, 150, dtype=np.uint8))
        assert oracle.ldpc_encode(rate, pl) == ref.ldpc_encode(rate, pl)
        k = INFO_BITS[rate]
        enc = oracle.ldpc_encode(rate, bytes(rng.integers(0, 256, k // 8, dtype=np.uint8)))
        bits = np.unpackbits(np.frombuffer(enc, np.uint8))[:648].astype(np.float32)
        for sigma in (0.4, 0.8, 1.2):
            llr = ((2 * (1 - 2 * bits) + rng.normal(0, 2 * sigma, 648)) / sigma).astype(np.float32)
            assert oracle.ldpc_decode_soft(rate, llr) == ref.ldpc_decode_soft(rate, llr)
        for n in (100, 649, 2000):
            llr = rng.normal(0, 3, n).astype(np.float32)
            assert oracle.ldpc_decode_soft(rate, llr, 6) == ref.ldpc_decode_soft(rate, llr, 6)


def test_fec_nonfinite_inputs(oracle, ref):
    rng = np.random.default_rng(77)
    for rate in range(6):
        for llr in nonfinite_cases(rng, oracle, rate):
            assert oracle.ldpc_decode_soft(rate, llr) == ref.ldpc_decode_soft(rate, llr)


def test_fec_valid_codewords_with_special_magnitudes(oracle, ref):
    """What the HIP kernels' iteration-0 shortcut relies on, on the compiled reference itself: a word whose channel hard
    decisions satisfy every row is decoded to exactly those bits with lastIterations() == 0, whatever the magnitudes."""
    for rate in range(6):
        llr = valid_special_codewords(np.random.default_rng(900 + rate), oracle, rate, n=24)
        for i, row in enumerate(llr):
            want = ref.ldpc_decode_soft(rate, row)
            assert oracle.ldpc_decode_soft(rate, row) == want, (rate, i)
            if i % 4 < 2:
                out, ok, iters = want[:3]
                hard = np.packbits((row[:INFO_BITS[rate]] < 0).astype(np.uint8)).tobytes()
                assert ok and iters == 0 and bytes(out)[:len(hard)] == hard, (rate, i)


def test_channel_cfo(oracle, ref):
    """WattersonChannel::applyCFO (hf_channel.hpp:161-232): mix down from 1500 Hz, 48-tap running mean, rotate, mix up —
    float libm calls and running sums in the reference's order, bitwise; buffers under 256 samples pass unchanged."""
    rng = np.random.default_rng(8)
    for n, cfo in ((255, 20.0), (256, 20.0), (4480, 35.5), (4480, -12.25), (12320, 50.0), (9000, 0.0025)):
        t = np.arange(n) / 48000.0
        x = (0.3 * np.cos(2 * np.pi * 1300.0 * t) + 0.2 * np.sin(2 * np.pi * 1950.0 * t) + rng.normal(0, 0.05, n)).astype(np.float32)
        a, b = oracle.channel_apply_cfo(x, cfo), ref.channel_apply_cfo(x, cfo)
        assert beq(a, b), (n, cfo)
        assert (n < 256) == bool(beq(a, x))


def test_interleavers(oracle, ref):
    x = np.random.default_rng(2).normal(size=648).astype(np.float32)
    assert beq(oracle.interleaver_deinterleave(6, 108, x), ref.interleaver_deinterleave(6, 108, x))
    for bps in (60, 116, 176):
        p, inv = oracle.channel_interleaver_perm(bps)
        mine = np.zeros(648, np.float32); mine[inv] = x          # out[inverse_permutation[i]] = in[i]
        assert beq(mine, ref.channel_interleaver(bps, x, inverse=True))
        mine = np.zeros(648, np.float32); mine[p] = x
        assert beq(mine, ref.channel_interleaver(bps, x, inverse=False))


def test_fft_nco_cdiv(oracle, ref):
    rng = np.random.default_rng(3)
    for n in (64, 512, 1024):
        x = (rng.normal(size=n) + 1j * rng.normal(size=n)).astype(np.complex64)
        assert beq(oracle.fft_forward(x), ref.fft_forward(x)) and beq(oracle.fft_inverse(x), ref.fft_inverse(x))
    assert beq(oracle.nco(1500, 48000, 30000), ref.nco(1500, 48000, 30000))


CASES = [("QAM16", "R3_4", 1024, {}), ("DQPSK", "R1_2", 512, {}), ("QPSK", "R1_2", 512, {}), ("QAM32", "R3_4", 1024, {}),
         ("D8PSK", "R3_4", 1024, dict(pilot_spacing=2)), ("DBPSK", "R1_4", 512, {}), ("BPSK", "R1_2", 512, {}),
         ("QAM64", "R3_4", 512, {}), ("QAM256", "R5_6", 512, {}), ("DQPSK", "R1_4", 512, dict(use_pilots=1)),
         ("QAM16", "R2_3", 1024, dict(n_data_symbols=12)),
         ("QAM16", "R1_2", 1024, dict(pilot_spacing=2)), ("DQPSK", "R1_2", 1024, dict(pilot_spacing=2, use_pilots=1))]   # 30 pilots


@pytest.mark.parametrize("mod,rate,fft,kw", CASES)
def test_tables_modulator_demodulator(oracle, ref, mod, rate, fft, kw):
    cfg = make_config(fft, mod, rate, **kw)
    g = geometry(cfg)
    ta, tb = oracle.demod_tables(cfg), ref.demod_tables(cfg)
    assert all(beq(ta[k], tb[k]) for k in ta)
    rng = np.random.default_rng(hash((mod, rate)) & 0xFFFF)
    for trial in range(6):
        nbytes = (g.llrs_per_frame // 648 + 1) * (INFO_BITS[cfg.code_rate] // 8)
        payload = bytes(rng.integers(0, 256, nbytes, dtype=np.uint8))
        enc = oracle.ldpc_encode(cfg.code_rate, payload)
        (a, pa), (b, pb) = oracle.modulate_frame(cfg, enc), ref.modulate_frame(cfg, enc)
        assert pa == pb and beq(a, b)
        audio, pre = ref.harness_awgn(cfg, payload, [30, 15, 6][trial % 3], 10 + trial)
        if trial >= 3:
            audio = ref.watterson(audio, 20.0, 0.5 if trial < 5 else 2.0, 0.1 if trial < 5 else 1.0, 70 + trial)
        shift = [0, -5, 0, -11, 0, 3][trial]
        x = audio[pre + shift: pre + shift + g.frame_samples]
        cfo = [0.0, 1.7, -3.2, 0.005, 12.5, -0.4][trial]
        la, sa = oracle.demod_synced(cfg, x, cfo, stages=True)
        lb, sb = ref.demod_synced(cfg, x, cfo, stages=True)
        assert beq(lb, ref.demod_synced_public(cfg, x, cfo))       # the stage driver IS process()
        for k in sa:
            assert beq(sa[k], sb[k]), (mod, trial, k)
        assert beq(la, lb)


@pytest.mark.parametrize("mod,rate,fft,kw", [("DQPSK", "R1_2", 512, {}), ("QAM16", "R3_4", 1024, {}),
                                             ("D8PSK", "R3_4", 1024, dict(pilot_spacing=2)), ("QPSK", "R1_2", 512, {}),
                                             ("DBPSK", "R1_4", 512, {}), ("QAM16", "R1_2", 1024, dict(pilot_spacing=2))])
def test_presynced(oracle, ref, mod, rate, fft, kw):
    cfg = make_config(fft, mod, rate, entry=1, **kw)
    g = geometry(cfg)
    rng = np.random.default_rng(5)
    for trial in range(6):
        enc = oracle.ldpc_encode(cfg.code_rate, bytes(rng.integers(0, 256, INFO_BITS[cfg.code_rate] // 8, dtype=np.uint8)))
        a, b = oracle.modulate_presynced(cfg, enc), ref.modulate_presynced(cfg, enc)
        assert beq(a, b)
        x = a * np.float32(0.5 / np.abs(a).max())
        x = ref.watterson(x, [30, 18, 10][trial % 3], 0.5, 0.1, 5 + trial, fading=trial % 2, multipath=trial % 2)
        x = x[:g.frame_samples]
        cfo, ph = [(0, 0), (3.0, 0.5), (-11.0, -2.0), (0.004, 0.1), (45.0, 3.0), (-2.0, -3.1)][trial]
        la, Ha, sa = oracle.demod_presynced(cfg, x, cfo, ph)
        lb, Hb, sb = ref.demod_presynced(cfg, x, cfo, ph)
        assert beq(la, lb) and beq(Ha, Hb) and beq(sa, sb), (mod, trial)


@pytest.mark.parametrize("mod,rate,fft,kw", [("DQPSK", "R1_2", 512, {}), ("QAM16", "R3_4", 1024, {}), ("D8PSK", "R2_3", 1024, dict(pilot_spacing=2))])
def test_presynced_without_a_preset_cfo(oracle, ref, mod, rate, fft, kw):
    """processPresynced on a demodulator whose frequency offset was never set (demodulator.cpp:920-925): the CFO comes from
    estimateCFOFromTraining (ofdm_sync.cpp:278-380) — two training symbols mixed down by a fresh NCO, correlated, arg(P)
    scaled, gated at |corr| >= 0.3 and clamped.  Frames with a real frequency shift (so the estimate is not 0), clean and
    noisy, and a noise-only buffer (correlation below the gate): LLRs, H and the tracker scalars equal the reference's."""
    from scipy.signal import hilbert
    cfg = make_config(fft, mod, rate, entry=1, **kw)
    g = geometry(cfg)
    rng = np.random.default_rng(15)
    est = []
    for trial in range(7):
        enc = oracle.ldpc_encode(cfg.code_rate, bytes(rng.integers(0, 256, INFO_BITS[cfg.code_rate] // 8, dtype=np.uint8)))
        a = oracle.modulate_presynced(cfg, enc)
        x = a * np.float32(0.5 / np.abs(a).max())
        shift = [0.0, 4.0, -7.5, 12.0, -15.0, 30.0, 0.0][trial]       # 30 Hz: beyond the unambiguous range of the 1024-FFT symbol
        if shift:
            x = np.real(hilbert(x.astype(np.float64)) * np.exp(2j * np.pi * shift * np.arange(x.size) / 48000.0)).astype(np.float32)
        snr = [40.0, 30.0, 20.0, 12.0, 6.0, 25.0, 0.0][trial]
        x = (x + rng.normal(0, np.sqrt(np.mean(x.astype(np.float64) ** 2) / 10 ** (snr / 10)), x.size)).astype(np.float32)
        if trial == 6:
            x = rng.normal(0, 0.1, x.size).astype(np.float32)            # no signal: the correlation gate returns 0
        x = x[:g.frame_samples]
        la, Ha, sa = oracle.demod_presynced(cfg, x, None)
        lb, Hb, sb = ref.demod_presynced(cfg, x, None)
        assert beq(la, lb) and beq(Ha, Hb) and beq(sa, sb), (mod, trial, sa, sb)
        est.append(float(sb[0]))
    assert any(e != 0.0 for e in est[:6]) and len(set(est[:6])) >= 4          # the estimator did run and produced distinct values


def test_reference_batch_baseline_equals_oracle_batch(oracle, ref):
    cfg = make_config(1024, "QAM16", "R3_4")
    audio, _ = oracle.make_batch(cfg, 48, seed=9, channel="watterson", snr_db=30.0)
    a = oracle.demod_decode_batch(cfg, audio, n_threads=4)
    b = ref.demod_decode_batch(cfg, audio)
    assert np.array_equal(a["bytes"], b["bytes"]) and np.array_equal(a["iters"], b["iters"]) and np.array_equal(a["ok"], b["ok"])


def test_complex_division_is_the_double_formula():
    """libgcc __divsc3 of this image == the double-precision formula the oracle and kernels use."""
    import subprocess, tempfile, textwrap, os
    src = textwrap.dedent(r'''
        #include <complex>
        #include <cstdio>
        #include <cstring>
        #include <random>
        typedef std::complex<float> C;
        int main(){ std::mt19937 r(1); std::uniform_real_distribution<float> u(-20,20); long bad=0;
          for(long i=0;i<2000000;i++){ C x(u(r),u(r)), y(u(r),u(r)); volatile float yr=y.real(); C q=x/C(yr,y.imag());
            double a=x.real(),b=x.imag(),c=y.real(),d=y.imag(),den=c*c+d*d; C w((float)((a*c+b*d)/den),(float)((b*c-a*d)/den));
            if(memcmp(&q,&w,8)) bad++; } printf("%ld\n",bad); }''')
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "d.cpp"); open(p, "w").write(src)
        subprocess.check_call(["g++", "-O2", "-std=c++17", p, "-o", os.path.join(td, "d")])
        assert subprocess.check_output([os.path.join(td, "d")]).strip() == b"0"


ACQ_MODES = [(1024, "QAM16", "R3_4", {}), (512, "DQPSK", "R1_2", {})]


def _streams(ref, cfg, rng, n=6):
    """Whole frames as the harnesses build them (silence + preamble + data, AWGN), plus variations of
    the leading silence and of the level."""
    out = []
    for t in range(n):
        payload = bytes(rng.integers(0, 256, INFO_BITS[cfg.code_rate] // 8, dtype=np.uint8))
        a, pre = ref.harness_awgn(cfg, payload, [30.0, 22.0, 12.0][t % 3], 777 + t)   # the search fails at 12 dB
        if t % 2:
            lead = rng.normal(0, 1e-3, int(rng.integers(200, 3000))).astype(np.float32)
            a = np.concatenate([lead, a * np.float32(0.3 + 0.2 * t)])
            pre += lead.size
        out.append((a.astype(np.float32), pre))
    return out


@pytest.mark.parametrize("fft,mod,rate,kw", ACQ_MODES)
def test_acquisition_stages(oracle, ref, fft, mod, rate, kw):
    """Scope row f1, stage by stage: LTS templates, Schmidl-Cox metric + energy gate at many offsets."""
    cfg = make_config(fft, mod, rate, **kw)
    for x, y in zip(oracle.lts_templates(cfg), ref.lts_templates(cfg)):
        assert beq(x, y)
    rng = np.random.default_rng(5)
    for a, pre in _streams(ref, cfg, rng, 3):
        nf = 0.0
        for off in list(range(0, 2400, 152)) + [pre - 7 * (fft + 96) + d for d in (0, 8, 40, 300)]:
            if off < 0:
                continue
            mo, ho = oracle.sc_metric(cfg, a, off, nf)
            mr, hr = ref.sc_metric(cfg, a, off, nf)
            assert beq(mo, mr) and ho == hr, (off, mo, mr)
            nf = float(mr[4])


@pytest.mark.parametrize("fft,mod,rate,kw", ACQ_MODES)
def test_acquisition_whole_streams(oracle, ref, fft, mod, rate, kw):
    """The chunk-fed search as a whole: sync declared at the same call, same Schmidl-Cox offset, same
    coarse CFO (bitwise), same refined LTS start / data start, same noise floor; different chunkings."""
    cfg = make_config(fft, mod, rate, **kw)
    rng = np.random.default_rng(6)
    hits = 0
    for a, pre in _streams(ref, cfg, rng, 6):
        for chunk in (960, 4800, 441):
            o = oracle.acquire(cfg, a, chunk)
            r = ref.acquire(cfg, a, chunk)
            for k in ("found", "fed_at_sync", "sync_offset", "refined_lts", "data_start"):
                assert o[k] == r[k], (k, chunk, o, r)
            assert np.float32(o["coarse_cfo"]).tobytes() == np.float32(r["coarse_cfo"]).tobytes(), (o, r)
            assert np.float32(o["noise_floor"]).tobytes() == np.float32(r["noise_floor"]).tobytes(), (o, r)
            hits += o["found"]
    assert hits >= 6


@pytest.mark.parametrize("fft,mod,rate,kw", ACQ_MODES)
def test_acquisition_long_streams_with_trims(oracle, ref, fft, mod, rate, kw):
    """Streams past 40000 samples: false Schmidl-Cox triggers on a tone with failing LTS confirmation, buffer
    trims, then the real preamble.  Oracle == compiled reference on every reported quantity."""
    cfg = make_config(fft, mod, rate, **kw)
    found = 0
    for x in long_acquisition_streams(oracle, cfg, np.random.default_rng(21)):
        o = oracle.acquire(cfg, x, 960)
        r = ref.acquire(cfg, x, 960)
        for k in ("found", "fed_at_sync", "sync_offset", "refined_lts", "data_start"):
            assert o[k] == r[k], (k, o, r)
        assert np.float32(o["coarse_cfo"]).tobytes() == np.float32(r["coarse_cfo"]).tobytes(), (o, r)
        assert np.float32(o["noise_floor"]).tobytes() == np.float32(r["noise_floor"]).tobytes(), (o, r)
        found += o["found"]
    assert found >= 1


def test_chirp_sync(oracle, ref):
    """Scope row f4: ChirpSync templates and TX chirps bitwise; detectDualChirp + OFDMChirpWaveform::detectSync
    on whole transmissions (several SNRs, TX CFOs, a noise-only stream): every reported quantity equal."""
    for a, b in zip(oracle.chirp_templates()[0], ref.chirp_templates()[0]):
        assert beq(a, b)
    assert beq(oracle.chirp_templates()[1], ref.chirp_templates()[1])
    for cfo in (0.0, 17.5, -40.0):
        assert beq(oracle.chirp_generate(cfo), ref.chirp_generate(cfo))
    cfg = make_config(512, "DQPSK", "R1_2", entry=1)
    hits = 0
    for x in chirp_streams(oracle, cfg, np.random.default_rng(8)):
        o, r = oracle.chirp_detect(x), ref.chirp_detect(x)
        for k in ("success", "up_chirp_start", "down_chirp_start", "start_sample"):
            assert o[k] == r[k], (k, o, r)
        for k in ("cfo_hz", "up_correlation", "down_correlation"):
            assert np.float32(o[k]).tobytes() == np.float32(r[k]).tobytes(), (k, o, r)
        hits += o["success"]
    assert hits >= 3


@pytest.mark.parametrize("rate,bps", [(0, 0), (2, 60), (4, 176), (5, 0), (1, 0)])
def test_v2_wire_format(oracle, ref, rate, bps):
    """Scope row f4, second half: CRC-16, v2::parseHeader, the frame builder (DataFrame/ControlFrame::serialize +
    encodeFrameWithLDPC) and RxPipeline::processFrame from the soft bits on (detectPing, deinterleaveCodewords,
    decodeFrame: CW0 -> header -> remaining codewords -> reassemble) — every field of the result equal."""
    from _util import v2_frame_cases
    rng = np.random.default_rng(100 + rate)
    for n in (0, 1, 15, 18, 200):
        d = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        assert oracle.crc16(d) == ref.crc16(d)
    assert oracle.crc16(b"123456789") == 0x29B1                       # CRC-16/CCITT-FALSE check value
    for payload, kw in ((b"", {}), (b"hello world" * 9, dict(seq=513)), (b"\x09\x08", dict(type=0x21, flags=0x41)),
                        (bytes(300), dict(total_cw=7, src_hash=1, dst_hash=0xFFFFFF))):
        a, b = oracle.v2_build_frame(rate, payload, **kw), ref.v2_build_frame(rate, payload, **kw)
        assert beq(a, b), (payload[:4], kw)
        dec = oracle.ldpc_decode_batch(rate, (1.0 - 2.0 * np.unpackbits(a[:1], axis=1)).astype(np.float32) * 4)[0][0]
        assert oracle.v2_parse_header(bytes(dec)) == ref.v2_parse_header(bytes(dec))
    seen = set()
    for name, soft in v2_frame_cases(oracle, rate, rng, bps):
        o, r = oracle.v2_decode_frame(rate, soft, bps), ref.v2_decode_frame(rate, soft, bps)
        assert o == r, (name, o, r)
        seen.add(o["status"])
    assert seen >= {0, 1, 2, 4, 5} and (rate == 1 or 3 in seen)      # (R1/3's garbage codeword happens to decode)


def test_v2_default_pipeline_deinterleaves(oracle, ref):
    """A default-constructed RxPipeline already owns ChannelInterleaver(60, 648) with interleaving enabled
    (rx_pipeline.cpp:13-18, rx_pipeline.hpp:177,182): without any setter call it deinterleaves with 60 bits per
    symbol, and setInterleaverConfig(60) changes nothing.  The host mirrors (RxFrameDecoder, HipRxFrameDecoder)
    follow this; here the reference itself is the witness."""
    from _util import v2_frame_cases
    rng = np.random.default_rng(4242)
    AS_CONSTRUCTED = 0xFFFFFFFF
    n_ok = 0
    for name, soft in v2_frame_cases(oracle, 0, rng, 60):
        as_built = ref.v2_decode_frame(0, soft, AS_CONSTRUCTED)
        assert as_built == ref.v2_decode_frame(0, soft, 60) == oracle.v2_decode_frame(0, soft, 60), name
        n_ok += as_built["success"]
    assert n_ok >= 6
    # and frames sent WITHOUT interleaving fail CW0 on a default pipeline (they would pass with interleaving off)
    cws = oracle.v2_build_frame(0, b"plain", seq=1)
    soft = (4.0 * (1.0 - 2.0 * np.unpackbits(cws, axis=1))).astype(np.float32).reshape(-1)
    assert ref.v2_decode_frame(0, soft, 0)["success"] == 1
    assert ref.v2_decode_frame(0, soft, AS_CONSTRUCTED)["success"] == 0



def test_stream_fixture_is_the_reference(ref):
    """tests/golden/stream.npz (the live-stream traces the GPU adapter is held to) regenerated from the compiled reference:
    the committed fixture is what OFDMDemodulator::process + getSoftBits do call by call."""
    from _util import STREAM_SCENARIOS, build_stream, cfg_from_array
    from conftest import GOLDEN
    g = np.load(GOLDEN / "fullsync.npz")
    want = np.load(GOLDEN / "stream.npz")
    name = "cfg2_dqpsk_r12"
    cfg = cfg_from_array(g[f"{name}__cfg"])
    geo = geometry(cfg)
    pre = int(g[f"{name}__meta"][0][0])
    for sc in ("two_frames", "idle_reset", "midframe"):
        audio, chunks = build_stream(g[f"{name}__audio"], STREAM_SCENARIOS[sc](geo.symbol_samples, pre))
        ready, synced, drained, soft = ref.demod_stream(cfg, audio, chunks)
        assert np.array_equal(ready, want[f"{name}__{sc}__ready"]) and np.array_equal(synced, want[f"{name}__{sc}__synced"])
        assert np.array_equal(drained, want[f"{name}__{sc}__drained"]) and beq(soft, want[f"{name}__{sc}__soft"])
        assert synced.any() and not synced[-1]              # it did sync, and it did leave SYNCED again


@pytest.mark.parametrize("name", ["cfg3_qam16_r34", "cfg2_dqpsk_r12"])
def test_midframe_search(oracle, ref, name):
    """The preamble check of the SYNCED state (demodulator.cpp:605-657): the restatement against process() itself (armed and
    called once, oracle/ref_shim.cpp) on buffers with the new preamble inside, outside and across the limits of the scan."""
    from _util import cfg_from_array, midframe_buffers
    from conftest import GOLDEN
    g = np.load(GOLDEN / "fullsync.npz")
    cfg = cfg_from_array(g[f"{name}__cfg"])
    geo = geometry(cfg)
    meta = g[f"{name}__meta"][0]
    hits = 0
    for seed in (7, 8):
        for buf in midframe_buffers(g[f"{name}__audio"], int(meta[0]), geo.symbol_samples, int(meta[1]), seed=seed):
            a, b = ref.midframe_search(cfg, buf), oracle.midframe_search(cfg, buf)
            assert {k: v for k, v in a.items() if k != "coarse_cfo"} == {k: v for k, v in b.items() if k != "coarse_cfo"}, (a, b)
            assert np.float32(a["coarse_cfo"]).tobytes() == np.float32(b["coarse_cfo"]).tobytes(), (a, b)
            hits += a["found"]
    assert 8 <= hits < 24


def test_headline_workload_is_the_references_channel(ref):
    """The bench's cfg3 stimulus vs the reference's own WattersonChannel (itu_r_f1487 good, 30 dB; /root/reference/src/sim/
    hf_channel.hpp:106-168,406-418): 2,048 frames each through the same receive path — FER, undetected errors, BP iteration
    distribution within 4 sigma of the sampling error, the tracker's noise / SNR estimates within 10 %.  (The statistics that
    decide 42 % of the headline's step time; the 4,096-frame table is profiles/r04_workload_check.txt.)"""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    from workload_check import tolerances, workload_statistics
    n = 2048
    r, o = workload_statistics(n)
    tol = tolerances(n)
    assert 0.85 < r["fer"] < 0.93 and 0.5 < r["at_limit"] < 0.68          # the regime the bench line reports (FER 0.894)
    for k, t in tol.items():
        assert abs(o[k] - r[k]) <= t, (k, r[k], o[k], t)
    for k in ("noise_var_median", "noise_var_mean", "snr_linear_median"):
        assert abs(o[k] - r[k]) <= 0.10 * r[k], (k, r[k], o[k])


ADAPTIVE = [("QPSK", "R1_2", 512, dict(n_data_symbols=14), "lms", {}), ("QPSK", "R1_2", 512, dict(n_data_symbols=14), "rls", {}),
            ("QAM16", "R3_4", 1024, dict(n_data_symbols=10), "lms", dict(lms_mu=0.1)), ("QAM16", "R3_4", 1024, dict(n_data_symbols=10), "rls", dict(rls_lambda=0.97)),
            ("BPSK", "R1_2", 512, dict(n_data_symbols=12), "rls", {}), ("QAM32", "R3_4", 1024, dict(n_data_symbols=8), "lms", {}),
            ("QAM64", "R3_4", 512, dict(n_data_symbols=9), "rls", {}), ("QAM256", "R5_6", 512, dict(n_data_symbols=8), "lms", {}),
            ("QAM16", "R1_2", 1024, dict(n_data_symbols=8), "lms", dict(decision_directed=False)),
            ("QPSK", "R1_2", 512, dict(n_data_symbols=10, use_pilots=0), "rls", {}),
            ("DQPSK", "R1_2", 512, dict(n_data_symbols=8), "lms", {})]       # differential: the branch is never reached


@pytest.mark.parametrize("mod,rate,fft,kw,kind,akw", ADAPTIVE)
def test_adaptive_equaliser(oracle, ref, mod, rate, fft, kw, kind, akw):
    """ModemConfig::adaptive_eq_enabled (types.hpp:170-174): Impl::equalize's LMS / RLS branch (channel_equalizer.cpp:569-581,
    705-722,773-805), off in every preset the reference ships — the restatement against the compiled reference, both entries,
    frames long enough for the weights to run free (they are re-seeded from the pilots' estimate during the first three symbols)."""
    for entry in (0, 1):
        cfg = make_config(fft, mod, rate, entry=entry, adaptive_eq=kind, **kw, **akw)
        plain = make_config(fft, mod, rate, entry=entry, **kw)
        g = geometry(cfg)
        rng = np.random.default_rng(77 + entry)
        differs = False
        for trial in range(4):
            nbytes = (g.llrs_per_frame // 648 + 1) * (INFO_BITS[cfg.code_rate] // 8)
            payload = bytes(rng.integers(0, 256, nbytes, dtype=np.uint8))
            enc = oracle.ldpc_encode(cfg.code_rate, payload)
            if entry == 0:
                audio, pre = ref.harness_awgn(cfg, payload, [30, 12, 20, 6][trial], 10 + trial)
                if trial >= 2:
                    audio = ref.watterson(audio, 20.0, 0.5, 1.0, 70 + trial)
                x = audio[pre: pre + g.frame_samples]
                la, sa = oracle.demod_synced(cfg, x, [0.0, 1.7, -3.2, 0.4][trial], stages=True)
                lb, sb = ref.demod_synced(cfg, x, [0.0, 1.7, -3.2, 0.4][trial], stages=True)
                for k in sa:
                    assert beq(sa[k], sb[k]), (mod, kind, trial, k)
                assert beq(la, lb), (mod, kind, trial)
                differs |= not beq(lb, ref.demod_synced(plain, x, [0.0, 1.7, -3.2, 0.4][trial])[0])
            else:
                a = ref.modulate_presynced(cfg, enc)
                x = a * np.float32(0.5 / np.abs(a).max())
                x = ref.watterson(x, [30, 18, 10, 24][trial], 0.5, 0.1, 5 + trial, fading=trial % 2, multipath=trial % 2)[:g.frame_samples]
                la, Ha, sa = oracle.demod_presynced(cfg, x, 3.0 * trial, 0.5 * trial)
                lb, Hb, sb = ref.demod_presynced(cfg, x, 3.0 * trial, 0.5 * trial)
                assert beq(la, lb) and beq(Ha, Hb) and beq(sa, sb), (mod, kind, trial)
                differs |= not beq(lb, ref.demod_presynced(plain, x, 3.0 * trial, 0.5 * trial)[0])
        # the switch does something (coherent modulations) or nothing at all (differential ones)
        assert differs == (mod not in ("DQPSK", "D8PSK", "DBPSK")), (mod, kind, entry)
