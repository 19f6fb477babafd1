#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the COMPILED REFERENCE.

Run in the build container (where /root/reference exists):
    python tests/golden/make_golden.py
It drives oracle/_ref/libultra_ref.so (secup/ProjectUltra's own sources compiled by
oracle/Makefile + the extern "C" shim oracle/ref_shim.cpp) and stores only DATA:
inputs and the reference's outputs.  Nothing from the reference's source text is stored.

Fixtures
  ldpc.npz     encoder known answers, decoder cases (LLRs in; bytes / success / iterations out)
  tables.npz   carrier maps, pilot signs, interpolation tables, Zadoff-Chu for the benchmark configs
  demod.npz    per mode: SYNCED-entry audio (data symbols only), initial CFO, the reference's LLRs,
               per-symbol tracker scalars, decoded bytes / success / iterations
  presynced.npz  processPresynced cases (training + data symbols, CFO, initial phase)
  sync.npz     acquisition (SEARCHING state fed in 960-sample chunks: found / fed / sync offset / coarse CFO /
               refined LTS / data start) and chirp synchronisation (detectDualChirp + detectSync's start sample)
               of whole transmissions
  frames.npz   v2 wire format: soft bits of frames in, RxPipeline::processFrame's result out
  stream.npz   live streams call by call (OFDMDemodulator::process + getSoftBits): the three exits of the SYNCED state
               (frame complete, idle calls, timeout) and re-acquisition; audio rebuilt from fullsync.npz's frames
  setcfo.npz   SYNCED-entry frames whose offset is replaced by setFrequencyOffset between two process() calls
  fullsync.npz full Schmidl-Cox receive of whole frames (OFDMDemodulator::process fed in 960-sample
               chunks): sync offset, coarse CFO, LLRs, and the data-start offset at which the
               SYNCED-entry loop reproduces those LLRs bit for bit
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle.bindings import INFO_BITS, Ref, geometry, make_config  # noqa: E402

OUT = Path(__file__).resolve().parent
r = Ref()
rng = np.random.default_rng(20260128)


def ldpc():
    d = {}
    payload = bytes((i * 7 + 0x42) & 0xFF for i in range(20))   # SURVEY.md §8c known answer
    d["kat_payload"] = np.frombuffer(payload, np.uint8)
    for rate in (0, 1, 2, 3, 4, 5):
        enc = r.ldpc_encode(rate, payload)
        d[f"kat_encoded_r{rate}"] = np.frombuffer(enc, np.uint8)
        k = INFO_BITS[rate]
        llrs, outs, oks, its = [], [], [], []
        for case in range(24):
            pl = bytes(rng.integers(0, 256, k // 8, dtype=np.uint8))
            bits = np.unpackbits(np.frombuffer(r.ldpc_encode(rate, pl), np.uint8))[:648].astype(np.float32)
            sigma = [0.35, 0.6, 0.8, 1.0, 1.3, 2.5][case % 6]
            llr = ((2.0 * (1 - 2 * bits) + rng.normal(0, 2 * sigma, 648)) / sigma).astype(np.float32)
            if case == 20: llr = np.zeros(648, np.float32)                    # erasures
            if case == 21: llr = -llr                                          # inverted
            if case == 22: llr = np.where(bits > 0, -6.0, 6.0).astype(np.float32)  # hard-decision input
            if case == 23: llr = np.clip(llr * 100, -1e4, 1e4).astype(np.float32)  # saturating
            out, ok, it = r.ldpc_decode_soft(rate, llr)
            llrs.append(llr); outs.append(np.frombuffer(out, np.uint8)); oks.append(ok); its.append(it)
        d[f"dec_llr_r{rate}"] = np.stack(llrs)
        d[f"dec_bytes_r{rate}"] = np.stack(outs)
        d[f"dec_ok_r{rate}"] = np.array(oks, np.uint8)
        d[f"dec_iters_r{rate}"] = np.array(its, np.int32)
        # the +-2.0 / mt19937(7) known answer of SURVEY.md §8c
        import ctypes as C
        bits = np.unpackbits(np.frombuffer(enc, np.uint8))[:648]
        l = np.where(bits > 0, -2.0, 2.0).astype(np.float32)
        mt = np.random.MT19937(); mt._legacy_seeding(7)
        g = np.random.Generator(mt)  # raw 32-bit draws == std::mt19937(7)
        draws = mt.random_raw(40)
        for v in draws: l[int(v) % 648] *= -0.5
        out, ok, it = r.ldpc_decode_soft(rate, l)
        d[f"kat_flip_llr_r{rate}"] = l
        d[f"kat_flip_iters_r{rate}"] = np.array([it, int(ok)], np.int32)
        # multi-block (bit-level concatenation) and short input
        for n in (300, 1296, 1500):
            llr = rng.normal(0, 4, n).astype(np.float32)
            out, ok, it = r.ldpc_decode_soft(rate, llr, 10)
            d[f"mb_llr_r{rate}_{n}"] = llr
            d[f"mb_out_r{rate}_{n}"] = np.frombuffer(out, np.uint8)
            d[f"mb_meta_r{rate}_{n}"] = np.array([int(ok), it], np.int32)
    np.savez_compressed(OUT / "ldpc.npz", **d)


MODES = [  # name, fft, mod, rate, extra kwargs
    ("cfg3_qam16_r34", 1024, "QAM16", "R3_4", {}),
    ("cfg2_dqpsk_r12", 512, "DQPSK", "R1_2", {}),
    ("qpsk_r12_512", 512, "QPSK", "R1_2", {}),
    ("qam32_r34", 1024, "QAM32", "R3_4", {}),
    ("d8psk_r34", 1024, "D8PSK", "R3_4", dict(pilot_spacing=2)),
    ("dbpsk_r14", 512, "DBPSK", "R1_4", {}),
    ("bpsk_r12", 512, "BPSK", "R1_2", {}),
    ("qam64_r34", 512, "QAM64", "R3_4", {}),
    ("qam256_r56", 512, "QAM256", "R5_6", {}),
    ("dqpsk_pilots_r14", 512, "DQPSK", "R1_4", dict(use_pilots=1)),
    ("qam16_r23_long", 1024, "QAM16", "R2_3", dict(n_data_symbols=12)),
]


def cfg_array(cfg):
    return np.frombuffer(bytes(cfg), np.uint32).copy()      # the struct's words (the two float fields as their bits)


def tables():
    d = {}
    for name, fft, mod, rate, kw in MODES:
        cfg = make_config(fft, mod, rate, **kw)
        t = r.demod_tables(cfg)
        d[f"{name}__cfg"] = cfg_array(cfg)
        for k, v in t.items():
            d[f"{name}__{k}"] = v
    np.savez_compressed(OUT / "tables.npz", **d)


def demod():
    d = {}
    for name, fft, mod, rate, kw in MODES:
        cfg = make_config(fft, mod, rate, **kw)
        g = geometry(cfg)
        audio, cfos, llrs, scals, dec, meta = [], [], [], [], [], []
        nfr = 4 if g.frame_samples > 6000 else 6
        for t in range(nfr):
            snr = [30, 18, 9, 30, 14, 6][t]
            nbytes = (g.llrs_per_frame // 648 + 1) * (INFO_BITS[cfg.code_rate] // 8)
            payload = bytes(rng.integers(0, 256, nbytes, dtype=np.uint8))
            a, pre = r.harness_awgn(cfg, payload, snr, 4000 + t)
            if t >= 3:   # Watterson good / moderate on top (src/sim/hf_channel.hpp presets)
                a = r.watterson(a, snr, 0.5 if t < 5 else 1.0, 0.1 if t < 5 else 0.5, 900 + t)
            shift = [0, 0, -4, 0, 0, -9][t]          # inside the CP: exercises timing tracking
            x = a[pre + shift: pre + shift + g.frame_samples]
            cfo = [0.0, 2.5, -0.8, 0.0, 0.004, -14.0][t]
            l, st = r.demod_synced(cfg, x, cfo, stages=True)
            out, ok, it = r.ldpc_decode_soft(cfg.code_rate, l[:648])
            audio.append(x); cfos.append(cfo); llrs.append(l); scals.append(st["scal"])
            dec.append(np.frombuffer(out, np.uint8)); meta.append([int(ok), it])
        d[f"{name}__cfg"] = cfg_array(cfg)
        d[f"{name}__audio"] = np.stack(audio).astype(np.float32)
        d[f"{name}__cfo"] = np.array(cfos, np.float32)
        d[f"{name}__llr"] = np.stack(llrs)
        d[f"{name}__scal"] = np.stack(scals)
        d[f"{name}__bytes"] = np.stack(dec)
        d[f"{name}__meta"] = np.array(meta, np.int32)
    np.savez_compressed(OUT / "demod.npz", **d)


def presynced():
    d = {}
    from oracle.bindings import oracle
    for name, fft, mod, rate, kw in [("ps_dqpsk", 512, "DQPSK", "R1_2", {}), ("ps_qam16", 1024, "QAM16", "R3_4", {}),
                                     ("ps_d8psk", 1024, "D8PSK", "R3_4", dict(pilot_spacing=2)),
                                     ("ps_qpsk", 512, "QPSK", "R1_2", {})]:
        cfg = make_config(fft, mod, rate, entry=1, **kw)
        g = geometry(cfg)
        audio, par, llrs, scals = [], [], [], []
        for t in range(4):
            payload = bytes(rng.integers(0, 256, INFO_BITS[cfg.code_rate] // 8, dtype=np.uint8))
            enc = r.ldpc_encode(cfg.code_rate, payload)
            x = r.modulate_presynced(cfg, enc)
            x = x * np.float32(0.5 / np.abs(x).max())
            x = r.watterson(x, [30, 16, 9, 24][t], 0.5, 0.1, 300 + t, fading=t % 2, multipath=t % 2)[:g.frame_samples]
            cfo, ph = [(0.0, 0.0), (3.0, 0.5), (-11.0, -2.0), (0.004, 0.1)][t]
            l, H, sc = r.demod_presynced(cfg, x, cfo, ph)
            audio.append(x); par.append([cfo, ph]); llrs.append(l); scals.append(sc)
        d[f"{name}__cfg"] = cfg_array(cfg)
        d[f"{name}__audio"] = np.stack(audio).astype(np.float32)
        d[f"{name}__cfo_phase"] = np.array(par, np.float32)
        d[f"{name}__llr"] = np.stack(llrs)
        d[f"{name}__scal"] = np.stack(scals)
    np.savez_compressed(OUT / "presynced.npz", **d)


def fullsync():
    """Whole frames through OFDMDemodulator::process (Schmidl-Cox search, 960-sample chunks)."""
    d = {}
    for name, fft, mod, rate, kw in MODES[:2]:
        cfg = make_config(fft, mod, rate, **kw)
        g = geometry(cfg)
        rows = []
        for t in range(3):
            payload = bytes(rng.integers(0, 256, INFO_BITS[cfg.code_rate] // 8, dtype=np.uint8))
            a, pre = r.harness_awgn(cfg, payload, 30.0, 12345 + t)
            l_full, sync_off, cfo, final_cfo, fed = r.demod_process_coarse(cfg, a, 960)
            found = -1
            if l_full.size >= g.llrs_per_frame:
                for off in range(pre - 64, pre + 65):
                    x = a[off: off + g.frame_samples]
                    if x.size < g.frame_samples: break
                    l = r.demod_synced_public(cfg, x, cfo)
                    if np.array_equal(l.view(np.uint32), l_full[:g.llrs_per_frame].view(np.uint32)):
                        found = off; break
            rows.append((a, pre, sync_off, cfo, found, l_full[:g.llrs_per_frame]))
            print(name, "frame", t, "nominal data start", pre, "sync_offset", sync_off, "coarse cfo", cfo, "final", final_cfo,
                  "SYNCED-entry reproduces process() at data start", found, "n_llr", l_full.size)
        d[f"{name}__cfg"] = cfg_array(cfg)
        d[f"{name}__audio"] = np.stack([x[0] for x in rows]).astype(np.float32)
        d[f"{name}__meta"] = np.array([[x[1], x[2], x[4]] for x in rows], np.int32)
        d[f"{name}__cfo"] = np.array([x[3] for x in rows], np.float32)
        d[f"{name}__llr"] = np.stack([x[5] for x in rows])
    np.savez_compressed(OUT / "fullsync.npz", **d)


def stream():
    """Live streams through OFDMDemodulator::process + getSoftBits, call by call (ref_demod_stream): the exits of the
    SYNCED state and re-acquisition.  The audio is rebuilt from fullsync.npz's frames by tests/_util.build_stream; only
    the per-call outputs are stored."""
    sys.path.insert(0, str(ROOT / "tests"))
    from _util import STREAM_SCENARIOS, build_stream, cfg_from_array, midframe_buffers
    g = np.load(OUT / "fullsync.npz")
    d = {}
    for name in ("cfg3_qam16_r34", "cfg2_dqpsk_r12"):
        cfg = cfg_from_array(g[f"{name}__cfg"])
        geo = geometry(cfg)
        frames = g[f"{name}__audio"]
        pre = int(g[f"{name}__meta"][0][0])
        # the preamble check of the SYNCED state on its own (ref_midframe_search: process() armed and called once)
        res = [r.midframe_search(cfg, b) for b in midframe_buffers(frames, pre, geo.symbol_samples, int(g[f"{name}__meta"][0][1]))]
        d[f"{name}__midframe_ints"] = np.array([[q["found"], q["sts_start"], q["refined_lts"], q["consume"]] for q in res], np.int64)
        d[f"{name}__midframe_cfo"] = np.array([q["coarse_cfo"] for q in res], np.float32)
        print(name, "midframe probes", [(q["found"], q["consume"]) for q in res])
        for sc, recipe in STREAM_SCENARIOS.items():
            audio, chunks = build_stream(frames, recipe(geo.symbol_samples, pre))
            ready, synced, drained, soft = r.demod_stream(cfg, audio, chunks)
            d[f"{name}__{sc}__ready"] = ready; d[f"{name}__{sc}__synced"] = synced
            d[f"{name}__{sc}__drained"] = drained; d[f"{name}__{sc}__soft"] = soft
            print(name, sc, "calls", chunks.size, "samples", audio.size, "ready calls", int(ready.sum()), "soft bits", soft.size,
                  "synced transitions", int(np.abs(np.diff(synced.astype(int))).sum()))
    np.savez_compressed(OUT / "stream.npz", **d)


def sync():
    sys.path.insert(0, str(ROOT / "tests"))
    from _util import chirp_streams, long_acquisition_streams
    d = {}
    lrng = np.random.default_rng(4242)
    # Schmidl-Cox acquisition: whole frames at several SNRs / leads, one noise-only stream
    for name, (fft, mod, rate) in dict(cfg3=(1024, "QAM16", "R3_4"), cfg2=(512, "DQPSK", "R1_2")).items():
        cfg = make_config(fft, mod, rate)
        g = geometry(cfg)
        n = g.frame_samples + 7 * (fft + g.cp_len) + 3200
        rows = []
        for t in range(5):
            payload = bytes(lrng.integers(0, 256, INFO_BITS[cfg.code_rate] // 8, dtype=np.uint8))
            a, _ = r.modulate_frame(cfg, r.ldpc_encode(int(cfg.code_rate), payload))
            a = a * np.float32(0.5 / np.abs(a).max())
            sigma = np.sqrt(np.mean(a.astype(np.float64) ** 2) / 10 ** ([30.0, 22.0, 12.0, 26.0, 18.0][t] / 10))
            a = (a + lrng.normal(0, sigma, a.size)).astype(np.float32)
            lead = lrng.normal(0, 2e-4, int(lrng.integers(0, 2800))).astype(np.float32)
            x = np.concatenate([lead, a * np.float32(0.4 + 0.15 * t)])
            rows.append(np.concatenate([x, lrng.normal(0, 2e-4, max(0, n - x.size)).astype(np.float32)])[:n])
        rows.append(lrng.normal(0, 0.05, n).astype(np.float32))
        audio = np.stack(rows)
        res = [r.acquire(cfg, x, 960) for x in audio]
        d[f"acq_{name}__cfg"] = cfg_array(cfg)
        d[f"acq_{name}__audio"] = audio
        d[f"acq_{name}__ints"] = np.array([[q["found"], q["fed_at_sync"], q["sync_offset"], q["refined_lts"], q["data_start"]] for q in res], np.int64)
        d[f"acq_{name}__cfo"] = np.array([q["coarse_cfo"] for q in res], np.float32)
        print("acquire", name, [(q["found"], q["data_start"]) for q in res])
    # chirp synchronisation: transmissions with 0 / 12.5 / -30 Hz offset and a noise-only buffer
    cfg = make_config(512, "DQPSK", "R1_2", entry=1)
    streams = chirp_streams(r, cfg, lrng, n=3)
    streams = [streams[0], streams[1], streams[2], streams[-1][:66000]]
    for i, x in enumerate(streams):
        q = r.chirp_detect(x)
        d[f"chirp{i}__audio"] = x
        d[f"chirp{i}__ints"] = np.array([q["success"], q["up_chirp_start"], q["down_chirp_start"], q["start_sample"]], np.int64)
        d[f"chirp{i}__floats"] = np.array([q["cfo_hz"], q["up_correlation"], q["down_correlation"]], np.float32)
        print("chirp", i, q)
    np.savez_compressed(OUT / "sync.npz", **d)


def frames():
    sys.path.insert(0, str(ROOT / "tests"))
    from _util import v2_frame_cases
    d = {}
    frng = np.random.default_rng(777)
    for rate, bps in ((0, 0), (2, 60), (4, 176)):
        for name, soft in v2_frame_cases(r, rate, frng, bps):
            q = r.v2_decode_frame(rate, soft, bps)
            key = f"r{rate}_b{bps}__{name}"
            d[key + "__soft"] = soft.astype(np.float32)
            d[key + "__res"] = np.array([q["success"], q["is_ping"], q["frame_type"], q["codewords_ok"], q["codewords_failed"],
                                         q["expected_codewords"], q["status"]], np.int32)
            d[key + "__data"] = np.frombuffer(q["frame_data"], np.uint8)
    np.savez_compressed(OUT / "frames.npz", **d)


def setcfo():
    """OFDMDemodulator::setFrequencyOffset between two process() calls of a SYNCED frame (demodulator.cpp:805-814): frames
    that started WITHOUT an offset (the case the advisor found unhandled) and with one, on a layout without pilots and on the
    headline's layout with pilots; the new offset arrives before symbol 1, 2 or 5."""
    d = {}
    rs = np.random.default_rng(4242)
    cases = []
    for name, fft, mod, rate, kw in (("cfg2_dqpsk_r12", 512, "DQPSK", "R1_2", {}), ("cfg3_qam16_r34", 1024, "QAM16", "R3_4", dict(n_data_symbols=8))):
        cfg = make_config(fft, mod, rate, **kw)
        g = geometry(cfg)
        d[f"{name}__cfg"] = cfg_array(cfg)
        nsym = g.frame_samples // g.symbol_samples
        for set_at in (1, 2, 5):
            for has0 in (0, 1):
                audio, cfo0, cfon, llrs = [], [], [], []
                for t in range(3):
                    nbytes = (g.llrs_per_frame // 648 + 1) * (INFO_BITS[cfg.code_rate] // 8)
                    payload = bytes(rs.integers(0, 256, nbytes, dtype=np.uint8))
                    a, pre = r.harness_awgn(cfg, payload, 24, 7000 + t)
                    shift = float(rs.normal(0, 3.0))
                    a = r.channel_apply_cfo(a, shift)
                    x = a[pre: pre + g.frame_samples]
                    c0 = float(np.float32(shift + rs.normal(0, 1.0)))
                    cn = float(np.float32(shift + rs.normal(0, 0.3)))
                    l = r.demod_synced_setcfo(cfg, x, set_at, cn, c0 if has0 else None)
                    audio.append(x); cfo0.append(c0); cfon.append(cn); llrs.append(l)
                key = f"{name}__at{set_at}_has{has0}"
                cases.append(key)
                d[f"{key}__audio"] = np.stack(audio).astype(np.float32)
                d[f"{key}__cfo0"] = np.array(cfo0, np.float32)
                d[f"{key}__cfo_new"] = np.array(cfon, np.float32)
                d[f"{key}__llr"] = np.stack(llrs)
    d["cases"] = np.array(cases)
    np.savez_compressed(OUT / "setcfo.npz", **d)

def adaptive():
    """ModemConfig::adaptive_eq_enabled (LMS / RLS, channel_equalizer.cpp:569-581,705-722,773-805): SYNCED-entry and
    processPresynced frames long enough for the weights to leave their seed, AWGN and Watterson."""
    d = {}
    cases = [("lms_qpsk", 512, "QPSK", "R1_2", dict(n_data_symbols=14), dict(adaptive_eq="lms")),
             ("rls_qam16", 1024, "QAM16", "R3_4", dict(n_data_symbols=10), dict(adaptive_eq="rls", rls_lambda=0.97)),
             ("lms_qam64", 512, "QAM64", "R3_4", dict(n_data_symbols=9), dict(adaptive_eq="lms", lms_mu=0.1)),
             ("rls_bpsk", 512, "BPSK", "R1_2", dict(n_data_symbols=12), dict(adaptive_eq="rls")),
             ("lms_qam32_nodd", 1024, "QAM32", "R3_4", dict(n_data_symbols=8), dict(adaptive_eq="lms", decision_directed=False)),
             ("rls_qam256", 512, "QAM256", "R5_6", dict(n_data_symbols=8), dict(adaptive_eq="rls"))]
    for name, fft, mod, rate, kw, akw in cases:
        for entry in (0, 1):
            cfg = make_config(fft, mod, rate, entry=entry, **kw, **akw)
            g = geometry(cfg)
            audio, par, llrs, scals = [], [], [], []
            for t in range(4):
                nbytes = (g.llrs_per_frame // 648 + 1) * (INFO_BITS[cfg.code_rate] // 8)
                payload = bytes(rng.integers(0, 256, nbytes, dtype=np.uint8))
                cfo, ph = [(0.0, 0.0), (3.0, 0.5), (-6.0, -2.0), (0.4, 0.1)][t]
                if entry == 0:
                    a, pre = r.harness_awgn(cfg, payload, [30, 14, 20, 8][t], 7000 + t)
                    if t >= 2:
                        a = r.watterson(a, 20.0, 0.5, 1.0, 950 + t)
                    x = a[pre: pre + g.frame_samples]
                    l, st = r.demod_synced(cfg, x, cfo, stages=True)
                    sc = st["scal"][-1]
                    ph = 0.0
                else:
                    x = r.modulate_presynced(cfg, r.ldpc_encode(cfg.code_rate, payload))
                    x = x * np.float32(0.5 / np.abs(x).max())
                    x = r.watterson(x, [30, 16, 9, 24][t], 0.5, 0.1, 330 + t, fading=t % 2, multipath=t % 2)[:g.frame_samples]
                    l, H, sc = r.demod_presynced(cfg, x, cfo, ph)
                audio.append(x); par.append([cfo, ph]); llrs.append(l); scals.append(sc)
            key = f"{name}_e{entry}"
            d[f"{key}__cfg"] = cfg_array(cfg)
            d[f"{key}__audio"] = np.stack(audio).astype(np.float32)
            d[f"{key}__cfo_phase"] = np.array(par, np.float32)
            d[f"{key}__llr"] = np.stack(llrs)
            d[f"{key}__scal"] = np.stack(scals)
    # whole frames through OFDMDemodulator::process (Schmidl-Cox search, 960-sample chunks) with the equaliser switched on:
    # what a live caller of the waveform sees (fullsync.npz's shape)
    # (frames of many symbols: the weights leave their seed behind the third symbol, and a waveform hands out a frame's
    # first 648 soft bits only — OFDMNvisWaveform::process, ofdm_cox_waveform.cpp:129-131)
    for name, fft, mod, rate, akw, snrs in [("live_lms_qam256", 512, "QAM256", "R5_6", dict(adaptive_eq="lms", lms_mu=0.1), [20.0, 22.0, 18.0]),
                                            ("live_rls_qam64", 512, "QAM64", "R3_4", dict(adaptive_eq="rls", rls_lambda=0.97), [15.0, 16.0, 17.0])]:
        cfg = make_config(fft, mod, rate, **akw)
        plain = make_config(fft, mod, rate)
        g = geometry(cfg)
        rows = []
        for t in range(3):
            payload = bytes(rng.integers(0, 256, INFO_BITS[cfg.code_rate] // 8, dtype=np.uint8))
            a, pre = r.harness_awgn(cfg, payload, snrs[t], 22345 + t)
            l_full, sync_off, cfo, final_cfo, fed = r.demod_process_coarse(cfg, a, 960)
            l_plain = r.demod_process_coarse(plain, a, 960)[0]
            assert l_full.size >= g.llrs_per_frame
            rows.append((a, pre, sync_off, cfo, l_full[:g.llrs_per_frame], not np.array_equal(l_full[:648], l_plain[:648])))
        assert any(x[5] for x in rows), name        # the switch shows in the soft bits a waveform hands out
        d[f"{name}__cfg"] = cfg_array(cfg)
        d[f"{name}__audio"] = np.stack([x[0] for x in rows]).astype(np.float32)
        d[f"{name}__meta"] = np.array([[x[1], x[2]] for x in rows], np.int32)
        d[f"{name}__cfo"] = np.array([x[3] for x in rows], np.float32)
        d[f"{name}__llr"] = np.stack([x[4] for x in rows])
    np.savez_compressed(OUT / "adaptive.npz", **d)


if __name__ == "__main__":
    which = sys.argv[1:] or ["ldpc", "tables", "demod", "presynced", "fullsync", "sync", "frames", "stream", "setcfo", "adaptive"]
    for name in which:
        globals()[name]()
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size // 1024, "KiB")
