"""Shared helpers for the parity tests."""
import ctypes as C
import numpy as np

from oracle.bindings import Config, INFO_BITS, make_config, geometry  # noqa: F401


def cfg_from_array(arr) -> Config:
    c = Config()
    for (name, _), v in zip(Config._fields_, arr.tolist()):
        setattr(c, name, int(v))
    return c


def modem_config_from_c(c):
    """ultra_hip_config -> projectultra_amd.ModemConfig + ReceiveContext kwargs."""
    from projectultra_amd import CodeRate, CyclicPrefixMode, Entry, ModemConfig, Modulation
    mc = ModemConfig(sample_rate=c.sample_rate, center_freq=c.center_freq, fft_size=c.fft_size,
                     num_carriers=c.num_carriers, cp_mode=CyclicPrefixMode(c.cp_mode), symbol_guard=c.symbol_guard,
                     pilot_spacing=c.pilot_spacing, use_pilots=bool(c.use_pilots),
                     modulation=Modulation(c.modulation), code_rate=CodeRate(c.code_rate))
    kw = dict(entry=Entry(c.entry), n_data_symbols=c.n_data_symbols, training_symbols=c.training_symbols or 2,
              max_iterations=c.max_iterations)
    return mc, kw


def context_for(c):
    from projectultra_amd import ReceiveContext
    mc, kw = modem_config_from_c(c)
    return ReceiveContext(mc, **kw)


def beq(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def noisy_codewords(oracle, rate, n, sigmas, seed):
    """BPSK-over-AWGN LLRs 2y/sigma^2 for random codewords of `rate` (SURVEY.md §8d cfg4 shape)."""
    rng = np.random.default_rng(seed)
    k = INFO_BITS[rate]
    llr = np.zeros((n, 648), np.float32)
    payloads = np.zeros((n, k // 8), np.uint8)
    for i in range(n):
        pl = rng.integers(0, 256, k // 8, dtype=np.uint8)
        payloads[i] = pl
        bits = np.unpackbits(np.frombuffer(oracle.ldpc_encode(rate, pl.tobytes()), np.uint8))[:648].astype(np.float32)
        s = sigmas[i % len(sigmas)]
        y = (1 - 2 * bits) + rng.normal(0, s, 648)
        llr[i] = (2 * y / (s * s)).astype(np.float32)
    return llr, payloads


def nonfinite_cases(rng, oracle, rate, n=12):
    """Noisy codewords with NaN / +-inf / huge / -0.0 values sprinkled in (the decoder's `<` tests and
    std::min/std::max clamps decide what those do: ldpc_decoder.cpp:181-224)."""
    k = INFO_BITS[rate]
    out = []
    for i in range(n):
        enc = oracle.ldpc_encode(rate, bytes(rng.integers(0, 256, k // 8, dtype=np.uint8)))
        bits = np.unpackbits(np.frombuffer(enc, np.uint8))[:648].astype(np.float32)
        llr = ((2 * (1 - 2 * bits) + rng.normal(0, 1.6, 648)) / 0.8).astype(np.float32)
        idx = rng.choice(648, 24, replace=False)
        vals = [np.nan, np.inf, -np.inf, 3e38, -3e38, -0.0, 75.0, -75.0]
        for j, ix in enumerate(idx[: 3 * (i % 8) + 1]):
            llr[ix] = vals[(i + j) % len(vals)]
        out.append(llr)
    return np.stack(out)


def long_acquisition_streams(oracle, cfg, rng):
    """Streams longer than 2 * OVERLAP_SAMPLES = 40000 samples, so that the buffer trims of the SEARCHING
    state run (demodulator.cpp:482-487,547-553,592-597): a 1.5 kHz tone (half-symbol periodic: Schmidl-Cox
    fires, the LTS confirmation fails) or low noise, then a whole frame at 30 dB, then a tail."""
    out = []
    for kind in ("tone", "noise", "tone"):
        payload = bytes(rng.integers(0, 256, INFO_BITS[cfg.code_rate] // 8, dtype=np.uint8))
        a, _ = oracle.modulate_frame(cfg, oracle.ldpc_encode(int(cfg.code_rate), payload))
        a = a * np.float32(0.5 / np.abs(a).max())
        a = (a + rng.normal(0, np.sqrt(np.mean(a.astype(np.float64) ** 2) / 1000), a.size)).astype(np.float32)
        n_lead = int(rng.integers(41000, 47000))
        if kind == "tone":
            lead = (0.2 * np.sin(2 * np.pi * 1500.0 * np.arange(n_lead) / 48000.0) + rng.normal(0, 1e-3, n_lead)).astype(np.float32)
            lead[-3000:] = rng.normal(0, 1e-3, 3000).astype(np.float32)
        else:
            lead = rng.normal(0, 2e-3, n_lead).astype(np.float32)
        out.append(np.concatenate([lead, a, rng.normal(0, 1e-3, 2500).astype(np.float32)]))
    return out


def chirp_streams(oracle, cfg, rng, n=5):
    """[noise lead][up chirp, gap, down chirp, gap (ChirpSync::generate)][2 training symbols + data
    (generateTrainingSymbols + modulate)][tail], frequency offsets of 0 / 12.5 / -30 / 55 Hz, AWGN — what OFDMChirpWaveform transmits
    (ofdm_chirp_waveform.cpp:104-127).  cfg must use the PRESYNCED entry."""
    out = []
    for t in range(n):
        payload = bytes(rng.integers(0, 256, 2 * (INFO_BITS[cfg.code_rate] // 8), dtype=np.uint8))
        body = oracle.modulate_presynced(cfg, oracle.ldpc_encode(int(cfg.code_rate), payload))
        sig = np.concatenate([oracle.chirp_generate(), body * np.float32(0.5 / np.abs(body).max())])
        cfo = [0.0, 12.5, -30.0, 55.0, 0.0][t % 5]
        if cfo:                                   # radio frequency error: the whole transmission shifts (harness: Hilbert + rotate)
            from scipy.signal import hilbert
            sig = np.real(hilbert(sig.astype(np.float64)) * np.exp(2j * np.pi * cfo * np.arange(sig.size) / 48000.0)).astype(np.float32)
        snr_db = [30.0, 15.0, 8.0, 20.0, 3.0][t % 5]
        sigma = np.sqrt(np.mean(sig.astype(np.float64) ** 2) / 10 ** (snr_db / 10))
        lead = int(rng.integers(500, 9000))
        x = np.concatenate([np.zeros(lead, np.float32), sig, np.zeros(3000, np.float32)])
        out.append((x + rng.normal(0, sigma, x.size)).astype(np.float32))
    out.append(rng.normal(0, 0.1, 70000).astype(np.float32))            # noise only
    return out


def chirp_initial_phase(cfo_hz, start_sample, sample_rate=48000):
    """OFDMChirpWaveform::process (ofdm_chirp_waveform.cpp:185-189): -2 pi cfo start / fs in double, rounded to
    float, wrapped to [-pi, pi] one float-rounded double step at a time."""
    import math
    ph = np.float32((((-2.0 * math.pi) * float(np.float32(cfo_hz))) * float(start_sample)) / float(sample_rate))
    while float(ph) > math.pi:
        ph = np.float32(float(ph) - 2.0 * math.pi)
    while float(ph) < -math.pi:
        ph = np.float32(float(ph) + 2.0 * math.pi)
    return ph
