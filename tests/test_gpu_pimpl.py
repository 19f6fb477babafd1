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
import subprocess
from pathlib import Path

import pytest

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
    if not ref.exists() or not hip.exists():
        pytest.skip(f"oracle/_ref/tools/{name}.* not built (needs /root/reference: `make -C oracle tools`)")
    rc_ref, out_ref, _ = _run(ref, args)
    rc_hip, out_hip, err_hip = _run(hip, args)
    assert out_ref.strip(), "the reference build must print something for the comparison to mean anything"
    a, b = out_ref.splitlines(), out_hip.splitlines()
    for i, (x, y) in enumerate(zip(a, b)):
        assert x == y, f"{name} {args}: stdout line {i} differs\n  reference: {x}\n  hip:       {y}\n  stderr tail: {err_hip[-800:]}"
    assert len(a) == len(b), (name, len(a), len(b), err_hip[-800:])
    assert rc_ref == rc_hip, (name, rc_ref, rc_hip, err_hip[-800:])


def test_the_hip_builds_do_not_link_the_reference_receive_path():
    """The .hip binaries must get OFDMDemodulator / LDPCDecoder from the drop-ins: no libultra_ref.so among their
    dependencies (it holds the reference's demodulator and decoder), libultra_hip.so present."""
    exe = TOOLS / "test_nvis_mode.hip"
    if not exe.exists():
        pytest.skip("oracle/_ref/tools not built")
    deps = subprocess.run(["ldd", str(exe)], capture_output=True, text=True).stdout
    assert "libultra_hip.so" in deps and "libultra_ref_tx.so" in deps
    assert "libultra_ref.so" not in deps
