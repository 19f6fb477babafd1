"""The product's host code under a hardened build, on the GPU box.

oracle/Makefile compiles the three drop-in translation units (projectultra_amd/host/*.cpp — all of include/ultra_hip_waveform.hpp
is compiled into them: buffers, rings, the process() / processPresynced() state machine, the slot pool) with
-fsanitize=undefined -fno-sanitize-recover -D_GLIBCXX_ASSERTIONS -D_FORTIFY_SOURCE=3 and links the manifest's `san` programs
against them (`.san`; libultra_hip.so and the reference's sources are the ordinary builds).  A signed overflow, a misaligned or
null access, an out-of-range vector index, an erase past the end or a fortified copy that overruns aborts the program.

Each `.san` run must print what the `.ref` run prints, return the same code and leave no sanitizer or assertion report on stderr.
(tests/test_adapter_sanitizers.py, CPU, runs the same host code against a test-only C-ABI stub under ASan and TSan.)"""
import re

import pytest

from _refprogs import exe, hardened, require, run
from test_gpu_pimpl import HARNESS_CASES, MOD, RATE
from test_gpu_ref_programs import _iwaveform_norm, _no_chirp_debug, _pipeline_log

pytestmark = pytest.mark.gpu
REPORT = re.compile(r"runtime error:|Assertion .* failed|__glibcxx_assert|buffer overflow detected|AddressSanitizer|terminate called")


def _pair(name, args, tmp_path, norm=_no_chirp_debug, attempts=1):
    """attempts > 1: programs whose result depends on how the reference's own engine threads interleave
    (tests/test_gpu_ref_programs.py::_run_until_the_builds_agree) — a sanitizer report fails at once, whatever the attempt."""
    ref, san = exe(name, "ref"), exe(name, "san")
    require(ref, san)
    for k in range(attempts):
        rc_ref, out_ref, err_ref = run(ref, args, cwd=tmp_path)
        rc, out, err = run(san, args, cwd=tmp_path)
        assert not REPORT.search(err), f"{name}.san {args}: sanitizer / assertion report\n{[l for l in err.splitlines() if REPORT.search(l)][:5]}"
        a, b = norm(out_ref), norm(out)
        if a and a == b and rc == rc_ref:
            return err_ref, err
        print(f"{name}.san {args}: attempt {k + 1} of {attempts} differs")
    assert a and a == b, (name, args, [(x, y) for x, y in zip(a, b) if x != y][:2], len(a), len(b), err[-600:])
    assert rc == rc_ref, (name, rc_ref, rc, err[-600:])
    return err_ref, err


def test_the_manifest_names_the_hardened_programs():
    assert {"demod_pimpl_harness", "engine_thread_harness", "test_hf_modem", "test_iwaveform"} <= set(hardened())


@pytest.mark.parametrize("sc,fft,mod,rate,seed", HARNESS_CASES, ids=[f"{c[0]}_{c[1]}_{c[2]}_{c[3]}" for c in HARNESS_CASES])
def test_scripted_pimpl_cases_hardened(sc, fft, mod, rate, seed, tmp_path):
    """Every scripted case of oracle/demod_pimpl_harness.cpp (carry, timing, set-cfo, presynced, mid-frame preambles, the three
    exits of SYNCED, getData, the decoder's limits, both interleavers, mixed entries)."""
    _pair("demod_pimpl_harness", [sc, str(fft), str(MOD[mod]), str(RATE[rate]), str(seed)], tmp_path, norm=lambda o: o.splitlines())


@pytest.mark.parametrize("scenario,seed", [("chirp", 5), ("dpsk", 7)])
def test_engine_threads_hardened(scenario, seed, tmp_path):
    """Two ModemEngines, feeder + GUI-poll + mode-change threads (oracle/engine_thread_harness.cpp) over the hardened drop-ins
    and the hardened factory."""
    err_ref, err = _pair("engine_thread_harness", [scenario, str(seed), "30"], tmp_path)
    assert _pipeline_log(err) == _pipeline_log(err_ref)


def test_legacy_modem_hardened(tmp_path):
    _pair("legacy_modem_harness", ["25", "3"], tmp_path)


@pytest.mark.parametrize("args", [["-w", "ofdm", "--snr", "30", "--frames", "3", "-m", "16qam", "-r", "3/4", "--cfo", "12"],
                                  ["-w", "chirp", "--snr", "18", "--frames", "2", "--cfo", "20", "-c", "moderate", "--seed", "9", "-m", "d8psk", "-r", "2/3"]],
                         ids=["ofdm_16qam", "chirp_d8psk"])
def test_hf_modem_hardened(args, tmp_path):
    _pair("test_hf_modem", args, tmp_path, norm=lambda o: [l for l in o.splitlines() if not l.startswith("[CHIRP-RX]")])


import os


@pytest.mark.parametrize("args", [["--snr", "17", "--cfo", "30", "--channel", "awgn", "-w", "ofdm_chirp", "--frames", "5"],
                                  pytest.param(["--snr", "5", "--cfo", "30", "--channel", "awgn", "-w", "mc_dpsk", "--frames", "3"],
                                               marks=pytest.mark.skipif(os.environ.get("ULTRA_LONG_TESTS") != "1", reason="once per round (ULTRA_LONG_TESTS=1): the reference's acquisition race makes an attempt cost up to 36 s; MC-DPSK through the hardened drop-ins runs in test_engine_threads_hardened[dpsk-7]"))],
                         ids=["ofdm_chirp", "mc_dpsk"])
def test_iwaveform_hardened(args, tmp_path):
    _pair("test_iwaveform", args, tmp_path, norm=_iwaveform_norm, attempts=4 if "mc_dpsk" in args else 1)


def test_headline_harness_and_ctest_pin_hardened(tmp_path):
    _pair("test_nvis_mode", ["--snr", "22", "--trials", "6"], tmp_path)
    _pair("test_multiblock_ldpc", [], tmp_path)
