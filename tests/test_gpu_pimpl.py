"""The reference's own programs, UNMODIFIED, on the HIP path — north_star's "tools ... call it unchanged", run and compared.

ultra::OFDMDemodulator (include/ultra/ofdm.hpp:58-127) and ultra::LDPCDecoder (include/ultra/fec.hpp:48-77) are pimpl classes
that every Monte-Carlo harness of the reference constructs directly (tools/test_nvis_mode.cpp:44-46,
tools/test_mode_snr.cpp:34-41).  projectultra_amd/host/hip_ofdm_demodulator.cpp and hip_ldpc_decoder.cpp DEFINE those two
classes over the C-ABI; oracle/Makefile compiles each of the reference's programs below twice from the source where it lies:

    oracle/_ref/tools/<name>.ref   linked against the compiled reference (its demodulator.cpp, channel_equalizer.cpp,
                                   ofdm_sync.cpp, ldpc_decoder.cpp)
    oracle/_ref/tools/<name>.hip   linked against the two drop-ins + libultra_hip.so instead of those four files

Same arguments, same seeds: stdout must be IDENTICAL, byte for byte — success counts, the demodulator's SNR estimates as the
programs print them, sync offsets, decoded bytes, and the programs' own PASS / FAIL verdicts (where the reference's test
fails on the reference, it must fail the same way on the GPU)."""
import re
import subprocess
from pathlib import Path

import pytest

from _refprogs import require

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
TOOLS = ROOT / "oracle" / "_ref" / "tools"

CASES = [
    # the headline harness (tools/test_nvis_mode.cpp:35-114): 8 modes x trials, fresh demodulator + decoder per trial, 960-sample chunks
    ("test_nvis_mode", ["--snr", "30", "--trials", "20"]),
    ("test_nvis_mode", ["--snr", "14", "--trials", "12"]),          # a failing SNR: partial success rates must agree too
    ("test_nvis_mode", ["--snr", "22", "--trials", "10"]),
    ("test_nvis_mode", ["--snr", "200", "--trials", "3"]),          # "infinite" SNR branch (:76)
    # tools/test_mode_snr.cpp: 512-FFT D8PSK / DQPSK over seven SNRs each, 20 trials per point
    ("test_mode_snr", []),
    # tests/test_sync_detection.cpp: the reference's own sync-offset tests (getLastSyncOffset, decode after sync)
    ("test_sync_detection", []),
    # tests/test_basic_ofdm.cpp --loopback: BASELINE configs[0]
    ("test_basic_ofdm", ["--loopback"]),
    # tools/test_ofdm_chirp_pilots.cpp: chirp detection on the host (header-only ChirpSync), then processPresynced with PILOTS
    # (coherent QPSK, Watterson channel) — the presynced entry through the pimpl class
    ("test_ofdm_chirp_pilots", ["--trials", "6"]),
    ("test_ofdm_chirp_pilots", ["--channel", "awgn", "--snr", "20", "--trials", "4", "--rate", "r14"]),
    ("test_ofdm_chirp_pilots", ["--channel", "good", "--snr", "25", "--trials", "4", "--pilots", "2"]),
]


def _run(exe, args):
    r = subprocess.run([str(exe)] + args, capture_output=True, text=True, timeout=900)
    return r.returncode, r.stdout, r.stderr


@pytest.mark.parametrize("name,args", CASES, ids=[f"{n}{'_'.join([''] + a)}" for n, a in CASES])
def test_reference_tool_runs_unmodified_on_the_hip_path(name, args):
    ref, hip = TOOLS / f"{name}.ref", TOOLS / f"{name}.hip"
    require(ref, hip)                                                # missing: fails when oracle/_ref/MANIFEST exists, else skips
    rc_ref, out_ref, _ = _run(ref, args)
    rc_hip, out_hip, err_hip = _run(hip, args)
    assert out_ref.strip(), "the reference build must print something for the comparison to mean anything"
    a, b = out_ref.splitlines(), out_hip.splitlines()
    for i, (x, y) in enumerate(zip(a, b)):
        assert x == y, f"{name} {args}: stdout line {i} differs\n  reference: {x}\n  hip:       {y}\n  stderr tail: {err_hip[-800:]}"
    assert len(a) == len(b), (name, len(a), len(b), err_hip[-800:])
    assert rc_ref == rc_hip, (name, rc_ref, rc_hip, err_hip[-800:])


def test_the_hip_builds_do_not_link_the_reference_receive_path():
    """The .hip binaries must get OFDMDemodulator / LDPCDecoder from the drop-ins: neither libultra_ref.so nor
    libultra_ref_rx.so among their dependencies (they hold the reference's demodulator and decoder), libultra_hip.so and the
    drop-in library present.  (tests/test_gpu_ref_programs.py checks this for every program of the manifest.)"""
    exe = TOOLS / "test_nvis_mode.hip"
    require(exe)
    deps = subprocess.run(["ldd", str(exe)], capture_output=True, text=True).stdout
    assert "libultra_hip.so" in deps and "libultra_hip_rx.so" in deps and "libultra_ref_core.so" in deps
    assert "libultra_ref.so " not in deps and "libultra_ref_rx.so" not in deps


# ---------------------------------------------------------------------------------------------------------------------
# tools/test_hf_modem.cpp: the reference's whole-boundary tool.  TX and RX waveforms come from WaveformFactory::create
# (:408,565), frames are LDPC-encoded, interleaved, modulated through IWaveform (generatePreamble + modulate), laid into 30+ s of
# audio, sent through CFO / Watterson / AWGN, and ONE gui::RxPipeline receives the stream in 960-sample chunks.  Three builds of
# the unmodified source (oracle/Makefile): .ref (the reference), .pimpl (the reference's factory and waveform classes over the
# two pimpl drop-ins — no source change at all), .hip (projectultra_amd/host/hip_waveform_factory.cpp: the HIP adapters, chirp
# detection on the GPU too).  With -v the reference's RxPipeline logs every decision it takes on the waveform's answers
# ("Sync detected at N, CFO=x Hz, corr=y", "process() returned false", trims, decode results: LOG_MODEM lines of the
# unmodified rx_pipeline.cpp, present in all three builds): stdout AND that log must be identical.
HF_CASES = [
    ["-w", "ofdm", "--snr", "25", "--frames", "3"],
    ["-w", "ofdm", "--snr", "30", "--frames", "3", "-m", "16qam", "-r", "3/4", "--cfo", "12"],
    ["-w", "ofdm", "--snr", "20", "--frames", "2", "-c", "good", "--seed", "5"],
    ["-w", "chirp", "--snr", "20", "--frames", "3"],
    ["-w", "chirp", "--snr", "18", "--frames", "2", "--cfo", "20", "-c", "moderate", "--seed", "9", "-m", "d8psk", "-r", "2/3"],
    ["-w", "chirp", "--snr", "15", "--frames", "2", "-r", "1/4", "--no-interleave"],
    ["-w", "dpsk", "--snr", "10", "--frames", "2"],                 # MC-DPSK: the reference's waveform, RxPipeline's decoder on the GPU
]
_STAMP = re.compile(r"^\[\s*\d+\.\d+\]")


def _pipeline_log(stderr):
    return [_STAMP.sub("", l) for l in stderr.splitlines() if "RxPipeline" in l]


def _stdout(out):
    """The reference's header-only ChirpSync printf()s its intermediate peaks to stdout ("[CHIRP-RX] ...",
    src/sync/chirp_sync.hpp); the .hip build detects the chirps on the GPU and has no such debug print.  What the detection
    RETURNS is in the pipeline log ("Sync detected at N, CFO=x Hz, corr=y") and is compared there."""
    return [l for l in out.splitlines() if not l.startswith("[CHIRP-RX]")]


@pytest.mark.parametrize("args", HF_CASES, ids=["_".join(a).replace("/", "") for a in HF_CASES])
def test_hf_modem_through_the_factory(args):
    exes = {k: TOOLS / f"test_hf_modem.{k}" for k in ("ref", "pimpl", "hip")}
    require(*exes.values())
    rc_ref, out_ref, err_ref = _run(exes["ref"], args + ["-v"])
    log_ref = _pipeline_log(err_ref)
    assert any("Sync detected" in l for l in log_ref), "the reference's pipeline must at least synchronise for the comparison to mean anything"
    for kind in ("pimpl", "hip"):
        rc, out, err = _run(exes[kind], args + ["-v"])
        assert _stdout(out) == _stdout(out_ref), (kind, args, out[-600:], out_ref[-600:], err[-600:])
        log = _pipeline_log(err)
        for i, (x, y) in enumerate(zip(log_ref, log)):
            assert x == y, f"{kind} {args}: RxPipeline log line {i} differs\n  reference: {x}\n  {kind}: {y}\n  before: {log_ref[max(0, i - 3):i]}"
        assert len(log) == len(log_ref), (kind, len(log), len(log_ref))
        assert rc == rc_ref


# ---------------------------------------------------------------------------------------------------------------------
# oracle/demod_pimpl_harness.cpp: scripted use of the two pimpl classes' PUBLIC interface where the reference's tools do not go
# (several frames on one object without reset(), setTimingOffset, setFrequencyOffset[WithPhase] at every point of a frame,
# processPresynced in all its branches with the frame's tail through process(), mid-frame preambles, the three exits of SYNCED,
# getData / getChannelQuality, the decoder's multi-block and limit semantics, both interleavers) — one source, linked against
# the compiled reference and against the drop-ins; every answer with floats as bit patterns; outputs identical.
MOD = dict(DBPSK=0, BPSK=1, DQPSK=2, QPSK=3, D8PSK=4, QAM16=6, QAM32=7, QAM64=8)
RATE = dict(R1_4=0, R1_3=1, R1_2=2, R2_3=3, R3_4=4, R5_6=5)
HARNESS_CASES = [
    ("carry", 1024, "QAM16", "R3_4", 1), ("carry", 512, "DQPSK", "R1_2", 2), ("carry", 512, "QPSK", "R1_2", 3), ("carry", 1024, "D8PSK", "R2_3", 4),
    ("timing", 1024, "QAM16", "R3_4", 5), ("timing", 512, "DQPSK", "R1_2", 6),
    ("setcfo", 1024, "QAM16", "R3_4", 7), ("setcfo", 512, "DQPSK", "R1_2", 8), ("setcfo", 512, "QAM64", "R5_6", 9),
    ("presynced", 512, "DQPSK", "R1_2", 10), ("presynced", 512, "QPSK", "R1_2", 11), ("presynced", 1024, "QAM16", "R3_4", 12), ("presynced", 512, "DBPSK", "R1_4", 13),
    ("midframe", 1024, "QAM16", "R3_4", 14), ("midframe", 512, "DQPSK", "R1_2", 15), ("midframe", 512, "QPSK", "R2_3", 16),
    ("exits", 512, "DQPSK", "R1_2", 17), ("exits", 1024, "QAM32", "R3_4", 18),
    ("getdata", 512, "DQPSK", "R1_2", 19), ("getdata", 1024, "QAM16", "R3_4", 20),
    ("decoder", 512, "QPSK", "R1_4", 21), ("decoder", 512, "QPSK", "R1_2", 22), ("decoder", 512, "QPSK", "R2_3", 23), ("decoder", 512, "QPSK", "R3_4", 24),
    ("decoder", 512, "QPSK", "R5_6", 25), ("decoder", 512, "QPSK", "R1_3", 26),
    ("interleave", 512, "QPSK", "R1_2", 27),
    # processPresynced and process() frames on one object without reset(): the Schmidl-Cox frame carries the presynced frame's tracker
    ("mixed", 1024, "QAM16", "R3_4", 31), ("mixed", 512, "DQPSK", "R1_2", 32), ("mixed", 512, "QPSK", "R1_2", 33), ("mixed", 1024, "D8PSK", "R2_3", 34),
]


@pytest.mark.parametrize("sc,fft,mod,rate,seed", HARNESS_CASES, ids=[f"{c[0]}_{c[1]}_{c[2]}_{c[3]}" for c in HARNESS_CASES])
def test_pimpl_classes_scripted(sc, fft, mod, rate, seed):
    ref, hip = TOOLS / "demod_pimpl_harness.ref", TOOLS / "demod_pimpl_harness.hip"
    require(ref, hip)
    args = [sc, str(fft), str(MOD[mod]), str(RATE[rate]), str(seed)]
    rc_ref, out_ref, err_ref = _run(ref, args)
    assert rc_ref == 0, err_ref[-800:]
    rc_hip, out_hip, err_hip = _run(hip, args)
    assert rc_hip == 0, err_hip[-1500:]
    a, b = out_ref.splitlines(), out_hip.splitlines()
    assert len(a) > 5
    if sc not in ("decoder", "interleave", "getdata"):
        assert any("soft 648" in l for l in a), "the reference must deliver codewords for the comparison to mean anything"
    for i, (x, y) in enumerate(zip(a, b)):
        assert x == y, f"{args}: line {i} differs\n  reference: {x[:260]}\n  hip:       {y[:260]}\n  before: {[l[:120] for l in a[max(0, i - 4):i]]}"
    assert len(a) == len(b), (args, len(a), len(b))
