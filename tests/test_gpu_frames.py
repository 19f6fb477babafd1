"""GPU parity for the v2 wire format (scope row f4, second half): ultra_hip_decode_frames_batch — what
RxPipeline::processFrame does with the soft bits of a frame — against the oracle, which
tests/test_oracle_vs_ref.py::test_v2_wire_format pins to the compiled reference."""
import numpy as np
import pytest

from _util import INFO_BITS, context_for, make_config, v2_frame_cases

pytestmark = pytest.mark.gpu
KEYS = ("success", "is_ping", "frame_type", "codewords_ok", "codewords_failed", "expected_codewords")


def _check(res, data, want, what):
    got = dict(zip(KEYS, (int(v) for v in res[:6])))
    for k in KEYS:
        assert got[k] == want[k], (what, k, got, want)
    assert int(res[7]) == want["status"], (what, res, want)
    assert bytes(data[:res[6]]) == want["frame_data"], (what, res, want)


@pytest.mark.parametrize("rate,bps", [("R1_4", 0), ("R1_2", 60), ("R3_4", 176), ("R5_6", 0), ("R2_3", 120), ("R1_3", 0)])
def test_frames_match_oracle(oracle, rate, bps):
    """Every case of _util.v2_frame_cases (complete data / control frames, waiting, failed codewords, bad headers,
    pings, noise, short buffers), each at its own length, with and without the channel deinterleaver."""
    cfg = make_config(512, "DQPSK", rate)
    ctx = context_for(cfg)
    ctx.set_deinterleave(bps)
    statuses = set()
    for name, soft in v2_frame_cases(oracle, cfg.code_rate, np.random.default_rng(100 + int(cfg.code_rate)), bps):
        want = oracle.v2_decode_frame(int(cfg.code_rate), soft, bps)
        out = ctx.decode_frames(soft[None, :])
        _check(out["results"].cpu().numpy()[0], out["frame_data"].cpu().numpy()[0], want, name)
        statuses.add(want["status"])
    assert statuses >= {0, 1, 2, 4, 5}


def test_frames_batched_and_strided(oracle):
    """Many frames in one call, zero padded to a common number of codewords, through both LLR layouts (codewords
    back to back = one LDPC launch; padded frame stride = one launch per codeword position)."""
    import torch
    cfg = make_config(512, "DQPSK", "R1_2")
    rate = int(cfg.code_rate)
    rng = np.random.default_rng(5)
    cases = [c for c in v2_frame_cases(oracle, cfg.code_rate, rng) if c[1].size >= 648][:14]
    n_cw = max(s.size // 648 for _, s in cases)
    soft = np.zeros((len(cases) * 6, n_cw * 648), np.float32)
    for i in range(soft.shape[0]):
        s = cases[i % len(cases)][1]
        soft[i, :s.size - s.size % 648] = s[:s.size - s.size % 648]
    want = [oracle.v2_decode_frame(rate, s) for s in soft[:len(cases)]]
    ctx = context_for(cfg)
    for pad in (0, 40):
        dev = torch.zeros((soft.shape[0], soft.shape[1] + pad), device="cuda")
        dev[:, :soft.shape[1]] = torch.from_numpy(soft).cuda()
        out = ctx.decode_frames(dev[:, :soft.shape[1]])
        res, data = out["results"].cpu().numpy(), out["frame_data"].cpu().numpy()
        for i in range(soft.shape[0]):
            _check(res[i], data[i], want[i % len(cases)], (pad, cases[i % len(cases)][0]))
    assert sum(w["success"] for w in want) >= 6


def test_frame_decoder_mirror(oracle):
    """RxFrameDecoder (mirror of RxPipeline's decode half): R1/4 before the connection whatever the data rate,
    the negotiated rate after it; interleaver only once configured; accumulation state."""
    from projectultra_amd import CodeRate, FrameType, RxFrameDecoder
    rng = np.random.default_rng(9)
    dec = RxFrameDecoder()
    dec.setDataMode(CodeRate.R3_4, False)                      # not connected: R1/4
    dec.setInterleavingEnabled(False)
    cws = oracle.v2_build_frame(0, b"CQ CQ", type=0x30, seq=3)
    soft = (4.0 * (1.0 - 2.0 * np.unpackbits(cws, axis=1))).astype(np.float32).reshape(-1)
    r = dec.decode_soft_bits(soft)
    want = oracle.v2_decode_frame(0, soft)
    assert r.success and r.frame_type == FrameType.DATA and r.frame_data == want["frame_data"] and r.frame_data[17:22] == b"CQ CQ"
    assert r.codewords_ok == want["codewords_ok"] == len(cws)
    r = dec.decode_soft_bits(soft[:648])                       # only CW0: waits for the rest
    assert not r.success and dec.isAccumulating() and dec.getExpectedCodewords() == len(cws)
    dec.setDataMode(CodeRate.R3_4, True)
    dec.setInterleavingEnabled(True)
    dec.setInterleaverConfig(176)
    perm = oracle.channel_interleaver_perm(176)[0]
    payload = bytes(rng.integers(0, 256, 333, dtype=np.uint8))
    cws = oracle.v2_build_frame(4, payload, seq=9)
    llr = (4.0 * (1.0 - 2.0 * np.unpackbits(cws, axis=1))).astype(np.float32)
    tx = np.empty_like(llr); tx[:, perm] = llr
    r = dec.decode_soft_bits(tx.reshape(-1))
    assert r.success and r.frame_data[17:17 + 333] == payload and not dec.isAccumulating()
    assert r.frame_data == oracle.v2_decode_frame(4, tx.reshape(-1), 176)["frame_data"]


def test_frame_decoder_default_deinterleaves(oracle):
    """A default-constructed RxFrameDecoder behaves like a default-constructed gui::RxPipeline: interleaving on,
    ChannelInterleaver(60, 648) (rx_pipeline.cpp:13-18) — frames of a default transmitter decode without any setter
    call, setInterleaverConfig(60) changes nothing, plain (non-interleaved) frames fail CW0.  The reference's own
    answer for the same soft bits is checked where the compiled reference is present."""
    from oracle.bindings import have_ref, Ref
    from projectultra_amd import RxFrameDecoder
    from _util import v2_frame_cases
    rng = np.random.default_rng(4242)
    ref = Ref() if have_ref() else None
    dec, dec60 = RxFrameDecoder(), RxFrameDecoder()
    dec60.setInterleaverConfig(60)
    n_ok = 0
    for name, soft in v2_frame_cases(oracle, 0, rng, 60):
        want = oracle.v2_decode_frame(0, soft, 60)
        if ref is not None:
            assert want == ref.v2_decode_frame(0, soft, 0xFFFFFFFF), name
        for d in (dec, dec60):
            r = d.decode_soft_bits(soft)
            assert (int(r.success), int(r.is_ping), r.codewords_ok, r.codewords_failed, r.frame_data) == \
                (want["success"], want["is_ping"], want["codewords_ok"], want["codewords_failed"], want["frame_data"]), name
        n_ok += want["success"]
    assert n_ok >= 6
    cws = oracle.v2_build_frame(0, b"plain", seq=1)
    soft = (4.0 * (1.0 - 2.0 * np.unpackbits(cws, axis=1))).astype(np.float32).reshape(-1)
    assert not dec.decode_soft_bits(soft).success
    dec.setInterleavingEnabled(False)
    assert dec.decode_soft_bits(soft).success


def test_chirp_frame_end_to_end(oracle):
    """A multi-codeword v2 data frame the way the production link carries it: DataFrame -> LDPC codewords ->
    channel interleaver per codeword -> chirp + training + DQPSK symbols -> frequency offset + AWGN; received by
    ultra_hip_chirp_receive_batch (detection, PRESYNCED demodulation) and ultra_hip_decode_frames_batch on its
    soft bits.  The payload comes back, and every stage output equals the oracle's."""
    import torch
    from scipy.signal import hilbert
    from _util import chirp_initial_phase
    rng = np.random.default_rng(77)
    rate, bps = 2, 60                                              # R1/2, 30 DQPSK carriers
    payload = bytes(rng.integers(0, 256, 90, dtype=np.uint8))
    cws = oracle.v2_build_frame(rate, payload, seq=42)             # 4 codewords
    perm = oracle.channel_interleaver_perm(bps)[0]
    bits = np.unpackbits(cws, axis=1)
    tx_bits = np.empty_like(bits); tx_bits[:, perm] = bits          # ChannelInterleaver::interleave per codeword
    n_sym = -(-cws.shape[0] * 648 // bps)
    cfg = make_config(512, "DQPSK", "R1_2", entry=1, n_data_symbols=n_sym)
    body = oracle.modulate_presynced(cfg, np.packbits(tx_bits.reshape(-1)).tobytes())
    sig = np.concatenate([oracle.chirp_generate(), body * np.float32(0.5 / np.abs(body).max())])
    streams = []
    for cfo, snr_db, lead in ((0.0, 25.0, 3000), (-22.0, 18.0, 700), (35.0, 14.0, 5000)):
        s = sig if not cfo else np.real(hilbert(sig.astype(np.float64)) * np.exp(2j * np.pi * cfo * np.arange(sig.size) / 48000.0))
        sigma = np.sqrt(np.mean(sig.astype(np.float64) ** 2) / 10 ** (snr_db / 10))
        x = np.concatenate([np.zeros(lead), s, np.zeros(6000 - lead + 2000)])
        streams.append((x + rng.normal(0, sigma, x.size)).astype(np.float32))
    audio = np.stack(streams)
    ctx = context_for(cfg)
    ctx.set_deinterleave(bps)
    rx = ctx.chirp_receive(audio, want_llr=True)
    n_soft = cws.shape[0] * 648
    out = ctx.decode_frames(rx["llr"][:, :n_soft])
    res, data = out["results"].cpu().numpy(), out["frame_data"].cpu().numpy()
    llr = rx["llr"].cpu().numpy()
    g = ctx.geometry
    for i, x in enumerate(audio):
        o = oracle.chirp_detect(x)
        assert o["success"] and int(rx["entry"][i]) == o["start_sample"]
        s = o["start_sample"]
        want = oracle.demod_decode_batch(cfg, x[s:s + g.frame_samples][None, :], cfo_hz=[o["cfo_hz"]],
                                         cfo_phase=[chirp_initial_phase(o["cfo_hz"], s)], decode=False)
        assert np.array_equal(llr[i].view(np.uint32), want["llr"][0].view(np.uint32)), i
        w = oracle.v2_decode_frame(rate, llr[i, :n_soft], bps)
        _check(res[i], data[i], w, f"stream {i}")
        assert w["success"] and w["frame_data"][17:17 + len(payload)] == payload, (i, w)


def test_frames_golden_reference():
    """The HIP path against the compiled reference's RxPipeline results (tests/golden/frames.npz)."""
    from conftest import GOLDEN
    g = np.load(GOLDEN / "frames.npz")
    keys = sorted(k[:-6] for k in g.files if k.endswith("__soft"))
    ctxs = {}
    for key in keys:
        rate, bps = int(key[1]), int(key.split("__")[0].split("_b")[1])
        if (rate, bps) not in ctxs:
            ctxs[(rate, bps)] = context_for(make_config(512, "DQPSK", rate))
            ctxs[(rate, bps)].set_deinterleave(bps)
        out = ctxs[(rate, bps)].decode_frames(g[key + "__soft"][None, :])
        res, data = out["results"].cpu().numpy()[0], out["frame_data"].cpu().numpy()[0]
        want = g[key + "__res"]
        assert list(res[:6]) == list(want[:6]) and res[7] == want[6], (key, res, want)
        assert bytes(data[:res[6]]) == g[key + "__data"].tobytes(), key
