"""The reference's own caller over the adapter, RUN — not argued: the compiled reference's unmodified ultra::gui::RxPipeline
(/root/reference/src/gui/modem/rx_pipeline.hpp:56-73; feedAudio -> tryProcessBuffer -> detectSync / setFrequencyOffset / reset /
process / getSoftBits -> decodeFrame -> frame callback: rx_pipeline.cpp:55-78,104-269) holds, through its raw IWaveform*,
  (a) the reference's waveform  (OFDMChirpWaveform / OFDMNvisWaveform: what WaveformFactory::create returns for OFDM_CHIRP /
      OFDM_COX, src/waveform/waveform_factory.cpp:16-17,52-53), then
  (b) this repository's adapter (ultra_hip::HipOfdmWaveform / HipOfdmCoxWaveform over the C-ABI on the GPU),
and is fed the same audio in 960-sample chunks.  oracle/_ref/rx_pipeline_harness (oracle/rx_pipeline_harness.cpp, built by
oracle/Makefile from the reference's sources where they lie; test infrastructure) records every call the pipeline makes on the
waveform, every answer (floats as bit patterns, all soft bits), the frame / ping callbacks, the queue and the buffer size.
The two logs must be IDENTICAL — vtable, ownership rule, call order and every number."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

from _util import make_config

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
HARNESS = ROOT / "oracle" / "_ref" / "rx_pipeline_harness"


def _shift(sig, cfo_hz):
    """radio frequency error: the whole transmission shifts (the harnesses' Hilbert + rotate)"""
    if not cfo_hz:
        return sig
    from scipy.signal import hilbert
    return np.real(hilbert(sig.astype(np.float64)) * np.exp(2j * np.pi * cfo_hz * np.arange(sig.size) / 48000.0)).astype(np.float32)


def _noisy(sig, snr_db, rng):
    sigma = np.sqrt(np.mean(sig.astype(np.float64) ** 2) / 10 ** (snr_db / 10))
    return (sig + rng.normal(0, sigma, sig.size)).astype(np.float32)


def _interleave(oracle, cws, bps):
    """ChannelInterleaver(bps, 648)::interleave per codeword, as the transmitter does (out[perm[i]] = in[i])"""
    if not bps:
        return cws
    perm = oracle.channel_interleaver_perm(bps)[0]
    bits = np.unpackbits(cws, axis=1)
    out = np.empty_like(bits)
    out[:, perm] = bits
    return np.packbits(out, axis=1)


FRAMES = (  # payload, v2 frame type, CFO of the transmission, SNR
    (b"\x01\x02\x03", 0x20, 0.0, 30.0),                 # control frame: one codeword
    (b"hello world", 0x30, 12.5, 18.0),                 # data frame in one codeword
    (bytes(range(100)), 0x30, -30.0, 25.0),             # several codewords: the pipeline sees CW0 and waits
    (b"x" * 7, 0x30, 3.0, 6.0),                         # low SNR
)


def chirp_stream(oracle, cfg, rate, bps, rng):
    """[noise][dual chirp + 2 training symbols + data symbols of a v2 frame]... — what OFDMChirpWaveform transmits
    (ofdm_chirp_waveform.cpp:104-127), one transmission every ~3 s"""
    parts = [rng.normal(0, 0.01, 30000).astype(np.float32)]
    for payload, typ, cfo, snr in FRAMES:
        cws = _interleave(oracle, oracle.v2_build_frame(rate, payload, type=typ, seq=5), bps)
        body = oracle.modulate_presynced(cfg, cws.tobytes())
        sig = _shift(np.concatenate([oracle.chirp_generate(), body * np.float32(0.5 / np.abs(body).max())]), cfo)
        parts += [_noisy(sig, snr, rng), rng.normal(0, 0.01, int(rng.integers(60000, 90000))).astype(np.float32)]
    return np.concatenate(parts)


def cox_stream(oracle, cfg, rate, rng):
    """[noise][Schmidl-Cox preamble + data symbols of a v2 frame]... (OFDMModulator::modulate with its preamble)"""
    parts = [rng.normal(0, 0.01, 20000).astype(np.float32)]
    for payload, typ, cfo, snr in FRAMES:
        sig, _ = oracle.modulate_frame(cfg, oracle.v2_build_frame(rate, payload, type=typ, seq=5).tobytes())
        sig = _shift(sig * np.float32(0.5 / np.abs(sig).max()), cfo)
        parts += [_noisy(sig, snr, rng), rng.normal(0, 0.01, int(rng.integers(40000, 120000))).astype(np.float32)]
    return np.concatenate(parts)


def run_harness(tmp_path, kind, impl, cfg, audio, bps, connected=1, chunk=960):
    f = tmp_path / f"{kind}.f32"
    audio.astype(np.float32).tofile(f)
    log = tmp_path / f"{kind}_{impl}.log"
    args = [str(HARNESS), kind, impl, str(f), str(log)] + [str(int(x)) for x in (
        cfg.fft_size, cfg.num_carriers, cfg.cp_mode, cfg.symbol_guard, cfg.pilot_spacing, cfg.use_pilots, cfg.modulation,
        cfg.code_rate, connected, bps, chunk)]
    r = subprocess.run(args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (kind, impl, r.returncode, r.stderr[-1500:])
    return log.read_text().splitlines()


def compare(ref, hip, what):
    for i, (a, b) in enumerate(zip(ref, hip)):
        assert a == b, f"{what}: line {i} differs\n  reference: {a[:300]}\n  adapter:   {b[:300]}\n  before: {ref[max(0, i - 3):i]}"
    assert len(ref) == len(hip), (what, len(ref), len(hip), ref[len(hip):][:3], hip[len(ref):][:3])


@pytest.fixture(scope="module")
def harness():
    from _refprogs import require
    require(HARNESS)                                                 # missing: fails when oracle/_ref/MANIFEST exists, else skips


@pytest.mark.parametrize("mod,rate,bps", [("DQPSK", "R1_2", 60), ("D8PSK", "R2_3", 0), ("DBPSK", "R1_4", 30)])
def test_rx_pipeline_over_the_chirp_waveform(tmp_path, oracle, harness, mod, rate, bps):
    """OFDM_CHIRP: OFDMChirpWaveform vs HipOfdmWaveform under RxPipeline — dual-chirp detection on the growing buffer (with
    the pipeline's search window and trims), the presynced entry with the accumulated CFO phase, v2 frames delivered."""
    cfg = make_config(512, mod, rate, entry=1)
    rng = np.random.default_rng(11)
    audio = chirp_stream(oracle, cfg, int(cfg.code_rate), bps, rng)
    ref = run_harness(tmp_path, "chirp", "ref", cfg, audio, bps)
    assert sum(l.startswith("FRAME_CALLBACK") for l in ref) >= 1, "the reference must deliver frames for the test to mean anything"
    assert any(l.startswith("QUEUE success=0") for l in ref)           # and the multi-codeword frame is left waiting
    hip = run_harness(tmp_path, "chirp", "hip", cfg, audio, bps)
    compare(ref, hip, f"chirp {mod} {rate}")


@pytest.mark.parametrize("fft,mod,rate,kw", [(1024, "QAM16", "R3_4", {}), (512, "DQPSK", "R1_2", {})])
def test_rx_pipeline_over_the_cox_waveform(tmp_path, oracle, harness, fft, mod, rate, kw):
    """OFDM_COX: OFDMNvisWaveform vs HipOfdmCoxWaveform under RxPipeline.  The pipeline hands detectSync() up to two seconds of
    buffer at once — the whole OFDMDemodulator::process state machine runs inside that call (search, sync, symbols, the exits
    of SYNCED, false alarms on noise) — then resets and restarts from the sync offset; every answer along that path."""
    cfg = make_config(fft, mod, rate, **kw)
    rng = np.random.default_rng(12)
    audio = cox_stream(oracle, cfg, int(cfg.code_rate), rng)
    ref = run_harness(tmp_path, "cox", "ref", cfg, audio, 0)
    assert sum("detectSync" in l and "-> 1" in l for l in ref) >= 2
    hip = run_harness(tmp_path, "cox", "hip", cfg, audio, 0)
    compare(ref, hip, f"cox {mod} {rate}")
