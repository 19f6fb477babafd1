"""Every program of the reference that constructs the replaced classes — its ctest suite, its command-line tools, its
ModemEngine programs — built UNMODIFIED from one manifest (oracle/ref_programs.txt, oracle/Makefile) and run side by side:

    .ref    the reference throughout (CPU)
    .hip    the product's link-time drop-ins + the product's waveform factory over libultra_hip.so (MI355X)
    .pimpl  (ModemEngine programs) the drop-ins under the REFERENCE's factory and waveform classes

Same arguments, same seeds.  Programs that run on one thread must print IDENTICAL stdout and return the same exit code —
the reference's own PASS / FAIL verdicts included (where a reference test fails on the reference, it must fail the same way on
the GPU).  ModemEngine programs run an acquisition thread and a decode thread beside the caller's (modem_rx.cpp:18-36,153-256),
so WHEN things are printed varies; for those the comparison is over what is deterministic — which frames were decoded, payload
bytes, the pipeline's logged decisions, the verdict and the exit code — as each case's normaliser states.

(tests/test_gpu_pimpl.py holds the five programs and the scripted harness of the earlier rounds; this file holds the rest of the
manifest.  tests/test_manifest.py, CPU, checks that the manifest covers the reference's tests/ and tools/.)"""
import os
import re

import pytest

from _refprogs import VARIANTS, exe, kind_of, require, run_all

pytestmark = pytest.mark.gpu


def _no_chirp_debug(out):
    """The reference's header-only ChirpSync printf()s its intermediate peaks to stdout ("[CHIRP-RX] ...",
    src/sync/chirp_sync.hpp); the product's factory detects the chirps on the GPU and has no such debug print.  What the
    detection RETURNS (start sample, CFO) is printed by the programs themselves and compared."""
    return [l for l in out.splitlines() if not l.startswith("[CHIRP-RX] Dual chirp") and not l.startswith("[CHIRP-RX] Position")
            and not re.match(r"\[CHIRP-RX\] CFO estimate: .* \(cfo_to_samples", l)]


def _compare(name, args, outs, normalise=_no_chirp_debug, same_rc=True):
    rc_ref, out_ref, err_ref = outs["ref"]
    a = normalise(out_ref)
    assert a, f"{name}: the reference build must print something for the comparison to mean anything\n{err_ref[-600:]}"
    for v, (rc, out, err) in outs.items():
        if v == "ref":
            continue
        b = normalise(out)
        for i, (x, y) in enumerate(zip(a, b)):
            assert x == y, (f"{name}.{v} {args}: line {i} differs\n  reference: {x[:300]}\n  {v}: {y[:300]}\n"
                            f"  before: {[l[:100] for l in a[max(0, i - 3):i]]}\n  stderr tail: {err[-800:]}")
        assert len(a) == len(b), (name, v, len(a), len(b), b[-3:], err[-800:])
        if same_rc:
            assert rc == rc_ref, (name, v, rc_ref, rc, err[-800:])


def _run(name, args, tmp_path, **kw):
    variants = VARIANTS[kind_of(name)]
    require(*[exe(name, v) for v in variants])
    return run_all(name, args, variants, cwd=tmp_path, **kw)


def _run_until_the_builds_agree(name, args, tmp_path, normalise, attempts=4, weak=None, **kw):
    """For programs whose RESULT — not only its timing — depends on how the reference's own threads interleave.

    ModemEngine's acquisition loop takes a snapshot of the sample buffer whenever it wakes; a chirp whose data has not arrived yet
    is taken for a PING and CONSUMED (/root/reference/src/gui/modem/modem_rx.cpp:84-141), so a feeder that hands over a frame in a
    burst of 960-sample calls races the acquisition thread inside the reference itself: the same binary can decode 3/3 in one run
    and miss a frame in the next (seen once in five runs of the .pimpl build of tools/test_iwaveform.cpp -w mc_dpsk, the reference's
    MC-DPSK demodulator throughout).  Such a program is run up to `attempts` times; the builds must agree completely in at least
    one attempt.  A systematic difference between the builds fails every attempt.  The builds run one after the other: side by
    side, three processes of a dozen threads each made the race common (two of three attempts in one run of the suite)."""
    last = None
    for k in range(attempts):
        outs = _run(name, args, tmp_path, one_at_a_time=True, **kw)     # three builds at once compete for the cores the race is run on
        try:
            _compare(name, args, outs, normalise=normalise)
            return outs
        except AssertionError as e:
            last = e
            keep = os.environ.get("ULTRA_KEEP_RACE_LOGS")             # a directory: the attempt's stdout / stderr of every build, for a post-mortem
            if keep:
                os.makedirs(keep, exist_ok=True)
                for v, (rc, out, err) in outs.items():
                    with open(os.path.join(keep, f"{name}_{abs(hash(tuple(args))) % 10000}_{k}_{v}.txt"), "w") as f:
                        f.write(f"rc {rc}\n--- stdout\n{out}\n--- stderr\n{err}")
            print(f"{name} {args}: attempt {k + 1} of {attempts}: the builds' threads interleaved differently ({str(e).splitlines()[0][:160]})")
    if weak is not None:
        # never agreed: what does NOT depend on the race must still hold for the last attempt — then the outcome is reported as an
        # expected failure of the REFERENCE's determinism (profiles/r06_variants/r06_reference_acquisition_race.txt), not as a pass
        weak(outs)
        pytest.xfail(f"{name} {args}: the reference's acquisition race gave the builds different frame sets in all {attempts} attempts")
    raise last


# ---------------------------------------------------------------------------------------------------------------------
# One thread, fixed seeds: stdout byte for byte.
EXACT = [
    # --- the reference's ctest suite (/root/reference/tests/CMakeLists.txt:8-108), as ctest runs it
    ("test_multiblock_ldpc", []),            # exact byte round trips, all rates, multi-codeword, interleaved (:104-488)
    ("test_comprehensive_modem", []),        # decoder behaviours, DQPSK LLR sign table, full chain (:59-768)
    ("test_layers", []),                     # layer-by-layer OFDM verification (:668-912 construct the demodulator)
    ("test_ofdm", []),
    ("test_modem_loopback", []),
    ("test_wav_loopback", ["--loopback"]),
    ("test_interleaver", []),                # the drop-in DEFINES Interleaver / ChannelInterleaver (:20-147)
    ("test_frame_v2_modem", []),             # fails 1 of 3 on the reference: the same one must fail here
    ("test_protocol_modem", []),             # 10/11 on the reference
    ("test_adaptive_link", []),
    # --- command-line tools (CMakeLists.txt:145-305)
    ("test_throughput", []),                 # north_star names it: TX-only (tools/test_throughput.cpp:78-134), Interleaver(6,108) from the drop-in
    ("test_otfs_vs_ofdm", ["--snr", "20", "--trials", "5"]),        # SURVEY 3.1's Watterson harness; src/otfs/otfs.cpp compiled where it lies
    ("test_otfs_vs_ofdm", ["--snr", "12", "--trials", "4", "--awgn"]),
    ("test_coherent_quick", []),
    ("test_hf_reality", ["--frames", "4", "--duration", "20", "-o", "hf.f32"]),
    ("test_hf_reality", ["--frames", "3", "--duration", "15", "--mode", "16qam", "--snr", "28", "-o", "hf.f32"]),
    ("test_single_frame", []),
    ("test_simple_noise", []),
    ("test_nvis_data", []),
    ("test_ofdm_chirp_waveform", ["200"]),   # its noise is seeded from std::random_device (:29-30): at 200 dB the draw no longer reaches the printed digits
    ("test_ofdm_chirp_cfo", ["200", "0"]),   # likewise (:81-82)
    ("test_ofdm_chirp_cfo", ["200", "25"]),
    ("test_dpsk_snr", []),                   # the decoder alone under the reference's single-carrier DPSK
    ("test_mc_dpsk", []),
    ("test_mc_dpsk_frame", []),              # MC-DPSK frames, LDPCDecoder from the drop-in
    # --- the legacy facade ultra::Modem (src/modem/modem.cpp:79-112,133-194) through oracle/legacy_modem_harness.cpp
    ("legacy_modem_harness", ["25", "3"]),
    ("legacy_modem_harness", ["14", "5"]),
    ("legacy_modem_harness", ["20", "9"]),
]


@pytest.mark.parametrize("name,args", EXACT, ids=[f"{n}{'_'.join([''] + [a.replace('/', '') for a in args])}" for n, args in EXACT])
def test_reference_program_stdout_identical(name, args, tmp_path):
    _compare(name, args, _run(name, args, tmp_path))


@pytest.mark.parametrize("name", ["test_ofdm_chirp_cfo", "test_ofdm_chirp_waveform"])
def test_ofdm_chirp_tools_default_noise(name, tmp_path):
    """tools/test_ofdm_chirp_cfo.cpp / test_ofdm_chirp_waveform.cpp at their default 15 dB: the noise comes from
    std::random_device, so two runs of the REFERENCE differ in the estimated SNR and the correlation; everything else — sync
    position, CFO estimate, soft-bit count, decode and verification verdicts — must agree."""
    def norm(out):
        return [l for l in _no_chirp_debug(out) if not l.startswith("Estimated SNR") and "Correlation:" not in l]
    _compare(name, [], _run(name, [], tmp_path), normalise=norm)


# ---------------------------------------------------------------------------------------------------------------------
# tools/test_iwaveform.cpp — what /root/reference/tests/regression_matrix.sh:16-23 calls its PRIMARY tool.  The quick matrix
# (:138-185): the five OFDM_CHIRP rows (a TX ModemEngine with its threads alive while ONE OFDMChirpWaveform + LDPCDecoder
# receive the stream) and two MC-DPSK rows (RX through ModemEngine::feedAudio: acquisition thread -> frame queue -> decode
# thread -> LDPC).  Deterministic: everything the program prints, with the decode thread's "[RX] Decoded" lines — printed when
# the thread gets there — compared as a set.  (OFDM_COX through the engine: test_engine_thread_harness[cox]; the tool's own
# `-w ofdm_cox` row waits its full 30 s for frames the reference's disconnected-mode engine never decodes — identical on the
# three builds, run once per round with the other real-time programs: test_iwaveform_ofdm_cox_row.)
IWAVEFORM = [
    ["--snr", "17", "--cfo", "0", "--channel", "awgn", "-w", "ofdm_chirp", "--frames", "5"],
    ["--snr", "17", "--cfo", "30", "--channel", "awgn", "-w", "ofdm_chirp", "--frames", "5"],
    ["--snr", "17", "--cfo", "50", "--channel", "awgn", "-w", "ofdm_chirp", "--frames", "5"],
    ["--snr", "15", "--cfo", "0", "--channel", "moderate", "-w", "ofdm_chirp", "--rate", "r1_4", "--frames", "5"],
    ["--snr", "15", "--cfo", "30", "--channel", "moderate", "-w", "ofdm_chirp", "--rate", "r1_4", "--frames", "5"],
    ["--snr", "5", "--cfo", "30", "--channel", "awgn", "-w", "mc_dpsk", "--frames", "3"],
]
IWAVEFORM_LONG = [                                                   # once per round with the real-time programs (each attempt the race spoils waits 30 s)
    ["--snr", "5", "--cfo", "0", "--channel", "awgn", "-w", "mc_dpsk", "--frames", "3"],
    ["--snr", "20", "--cfo", "0", "--channel", "awgn", "-w", "ofdm_cox", "--frames", "1"],
]


def _iwaveform_norm(out):
    lines = _no_chirp_debug(out)
    threaded = sorted(l for l in lines if l.startswith("  [RX] Decoded"))
    return [l for l in lines if not l.startswith("  [RX] Decoded")] + threaded


def _iwaveform_weak(outs):
    """What holds whatever the engine's threads do: the transmit side and the channel are identical across the builds; every frame a
    build reports as decoded is one that was sent, with its own source callsign (nothing is ever decoded WRONG); the verdict lines and
    the exit code follow from the build's own decoded set."""
    def split(out):
        lines = _no_chirp_debug(out)
        racy = [l for l in lines if l.startswith("  [RX] Decoded") or re.match(r"  Frame +\d+ \(seq=\d+\): (OK|MISSED)", l) or l.startswith("Decoded: ")]
        return [l for l in lines if l not in racy], racy
    fixed_ref, _ = split(outs["ref"][1])
    for v, (rc, out, err) in outs.items():
        fixed, racy = split(out)
        assert fixed == fixed_ref, (v, [(a, b) for a, b in zip(fixed_ref, fixed) if a != b][:2])
        decoded = set()
        for l in racy:
            m = re.match(r"  \[RX\] Decoded seq=(\d+) src=TEST(\d+)$", l)
            if m:
                assert int(m.group(2)) == int(m.group(1)) - 1, (v, l)     # frame k carries source TEST(k-1): decoded content is right
                decoded.add(int(m.group(1)))
        ok = {int(m.group(1)) for l in racy for m in [re.match(r"  Frame +\d+ \(seq=(\d+)\): OK", l)] if m}
        assert ok == decoded, (v, ok, decoded)
        n = len([l for l in racy if re.match(r"  Frame +\d+ \(seq=", l)])
        assert (rc == 0) == (len(decoded) == n), (v, rc, decoded, n)


@pytest.mark.parametrize("args", IWAVEFORM, ids=["_".join(a).replace("--", "") for a in IWAVEFORM])
def test_iwaveform_regression_matrix(args, tmp_path):
    if "mc_dpsk" in args:                                            # RX through the engine's threads: see _run_until_the_builds_agree
        outs = _run_until_the_builds_agree("test_iwaveform", args, tmp_path, _iwaveform_norm, weak=_iwaveform_weak)
    else:                                                            # OFDM_CHIRP rows receive on the calling thread: deterministic
        outs = _run("test_iwaveform", args, tmp_path)
        _compare("test_iwaveform", args, outs, normalise=_iwaveform_norm)
    if "ofdm_chirp" in args or "mc_dpsk" in args:
        assert "Decoded: 0/" not in outs["ref"][1], "the reference must decode something for the row to mean anything"


# ---------------------------------------------------------------------------------------------------------------------
# oracle/engine_thread_harness.cpp: two ModemEngines alive, audio on a feeder thread, the GUI's getters polled from another,
# mode changes from a third, MC-DPSK through the acquisition and decode threads.  stdout (delivered frames in hex, counts) and
# the RX pipeline's logged decisions must be identical across the three builds.
_STAMP = re.compile(r"^\[\s*\d+\.\d+\]")


def _pipeline_log(stderr):
    # "[RX] RxPipeline: Frame decoded/failed" is the decode thread's 10 ms poll (modem_rx.cpp:197-222): whether the last one is
    # printed before the engine is destroyed is a matter of timing; the pipeline's own lines come from the feeding thread
    return [_STAMP.sub("", l) for l in stderr.splitlines() if "] RxPipeline: " in l and "RxPipeline: Frame decoded," not in l
            and "RxPipeline: Frame failed," not in l]


@pytest.mark.parametrize("scenario,seed,snr", [("cox", 3, "28"), ("chirp", 5, "26"), ("dpsk", 7, "28"), ("all", 11, "30")])
def test_engine_thread_harness(scenario, seed, snr, tmp_path):
    outs = _run("engine_thread_harness", [scenario, str(seed), snr], tmp_path)
    _compare("engine_thread_harness", [scenario, seed], outs)
    log_ref = _pipeline_log(outs["ref"][2])
    if scenario != "dpsk":
        assert any("Sync detected" in l for l in log_ref), "the reference's pipeline must at least synchronise"
    if scenario in ("chirp", "all"):
        assert any("Frame decode SUCCESS" in l for l in log_ref), log_ref[-5:]
    if scenario in ("dpsk", "all"):
        assert "delivered 2" in outs["ref"][1]
    for v in ("pimpl", "hip"):
        log = _pipeline_log(outs[v][2])
        for i, (x, y) in enumerate(zip(log_ref, log)):
            assert x == y, f"{v} {scenario}: RxPipeline log line {i} differs\n  reference: {x}\n  {v}: {y}\n  before: {log_ref[max(0, i - 3):i]}"
        assert len(log) == len(log_ref), (v, len(log), len(log_ref))
        assert "[harness]" in outs[v][2]                              # the GUI polls ran beside the feed


# ---------------------------------------------------------------------------------------------------------------------
# The reference's other ModemEngine programs.  They sleep in real time and print as their threads get there; compared: the
# lines that state results, and the exit code.
def _verdicts(patterns):
    rx = re.compile("|".join(patterns))
    return lambda out: [l.rstrip() for l in _no_chirp_debug(out) if rx.search(l)]


def test_modem_engine_loopback(tmp_path):
    """tools/test_modem_engine_loopback.cpp: ten RX engines built and destroyed in turn beside one TX engine.  (On the reference it
    receives 0/10 — RxPipeline only searches every 48,000 fed samples, rx_pipeline.cpp:70-76, and the program feeds 30,476 — so
    what this compares is ten constructions and destructions of engines with live threads, and the same verdict.)"""
    outs = _run("test_modem_engine_loopback", [], tmp_path)
    _compare("test_modem_engine_loopback", [], outs)


def test_profile_acquisition(tmp_path):
    """tools/profile_acquisition.cpp:1-35 — per-trial OK / FAIL of OFDM, DPSK and PING acquisition through ModemEngine; the
    timings it prints are dropped."""
    def norm(out):
        keep = []
        for l in _no_chirp_debug(out):
            if re.search(r"feed=|Feed time|Wall time|Real-time factor", l):
                m = re.match(r"\s*\[\s*(\d+)\].*\b(OK|FAIL)\b", l)
                if m:
                    keep.append(f"[{m.group(1)}] {m.group(2)}")
                continue
            keep.append(l)
        return keep
    outs = _run_until_the_builds_agree("profile_acquisition", ["--trials", "2"], tmp_path, norm)
    assert "Decoded: 2/2" in outs["ref"][1]


def test_the_references_own_command_line_modem(tmp_path):
    """src/main.cpp — `ultra`, the reference's PRODUCT executable (CMakeLists.txt:97-110), built unmodified as main.{ref,pimpl,hip}:
    `ptx connect -o file` writes a CONNECT frame's audio (identical bytes from the three builds), `prx -w dpsk file` receives it
    through ModemEngine — acquisition thread, decode thread, LDPC on the GPU in the two product builds — and reports the frame, its
    callsigns and the counts; `ptx ping` / `prx` is the chirp-only probe; `info` is a fixed text."""
    variants = VARIANTS[kind_of("main")]
    require(*[exe("main", v) for v in variants])
    from _refprogs import run
    audio = {}
    for v in variants:
        rc, out, err = run(exe("main", v), ["ptx", "connect", "-s", "ALPHA", "-d", "BRAVO", "-w", "dpsk", "-o", f"connect_{v}.f32"], cwd=tmp_path)
        assert rc == 0, err[-500:]
        audio[v] = (tmp_path / f"connect_{v}.f32").read_bytes()
        rc, out, err = run(exe("main", v), ["ptx", "ping", "-w", "dpsk", "-o", f"ping_{v}.f32"], cwd=tmp_path)
        assert rc == 0 and (tmp_path / f"ping_{v}.f32").read_bytes() == (tmp_path / "ping_ref.f32").read_bytes()
        assert run(exe("main", v), ["info"], cwd=tmp_path)[1] == run(exe("main", "ref"), ["info"], cwd=tmp_path)[1]
    assert len(audio["ref"]) == 108960 * 4 and all(a == audio["ref"] for a in audio.values())

    def report(err):                                                 # what the receiver says about frames; its own log lines and the SNR figure left out
        return [re.sub(r"\(SNR=[^)]*\)", "(SNR)", l) for l in err.splitlines()
                if l.startswith("  [") and not l.startswith("  [ ") or l.startswith("    ") and "->" in l or l.startswith("  Frames:") or l.startswith("  PINGs:")]

    for what, want in (("connect_ref.f32", ["  [CONNECT] codewords=3", "    ALPHA -> BRAVO", "  Frames: 1", "  PINGs: 0"]), ("ping_ref.f32", None)):
        for attempt in range(4):                                     # the engine's acquisition race (see _run_until_the_builds_agree)
            got = {v: report(run(exe("main", v), ["prx", "-w", "dpsk", what], cwd=tmp_path)[2]) for v in variants}
            if all(g == got["ref"] for g in got.values()) and (want is None or got["ref"] == want):
                break
        assert all(g == got["ref"] for g in got.values()), (what, got)
        if want is not None:
            assert got["ref"] == want, got["ref"]
        # (the chirp-only probe is NOT detected by the reference's own build — its buffer never reaches the acquisition loop's
        # minimum — so what is compared for it is that the three builds say the same)


LONG = pytest.mark.skipif(os.environ.get("ULTRA_LONG_TESTS") != "1",
                          reason="real-time programs, 30-130 s per build: ULTRA_LONG_TESTS=1 (run once per round: profiles/r06_long_programs.txt)")


@LONG
@pytest.mark.parametrize("args", IWAVEFORM_LONG, ids=["_".join(a).replace("--", "") for a in IWAVEFORM_LONG])
def test_iwaveform_ofdm_cox_row(args, tmp_path):
    """The matrix's second MC-DPSK row and the tool's OFDM_COX row (which waits its full 30 s for frames the reference's
    disconnected-mode engine never decodes)."""
    _run_until_the_builds_agree("test_iwaveform", args, tmp_path, _iwaveform_norm, weak=_iwaveform_weak if "mc_dpsk" in args else None)


@LONG
def test_cli_simulator(tmp_path):
    """tools/cli_simulator.cpp: two stations (ModemEngine + ProtocolEngine each) over a simulated channel."""
    norm = _verdicts([r"PHASE", r"✓", r"✗", r"connected", r"Connected", r"received", r"Received", r"timeout", r"PASS", r"FAIL"])
    _run_until_the_builds_agree("cli_simulator", ["--snr", "20"], tmp_path, norm, attempts=2, timeout=600)


@LONG
def test_threaded_simulator(tmp_path):
    norm = _verdicts([r"TEST \d", r"SUCCESS", r"FAILED", r"TIMEOUT", r"PASS"])
    _run_until_the_builds_agree("threaded_simulator", [], tmp_path, norm, attempts=2, timeout=900)


@LONG
def test_sync_robustness(tmp_path):
    norm = _verdicts([r"PASS", r"FAIL", r"passed", r"failed", r"Result", r"RESULT", r"Summary", r"SUMMARY", r"\d+/\d+"])
    outs = _run("test_sync_robustness", ["--quick"], tmp_path, timeout=900)      # one of its eight SNR / rate suites: 160 engine runs
    _compare("test_sync_robustness", ["--quick"], outs, normalise=norm)


def test_hip_builds_link_no_file_of_the_reference_receive_path():
    """Every .hip / .pimpl binary gets OFDMDemodulator / LDPCDecoder from the drop-ins: libultra_hip_rx.so and libultra_hip.so
    among its dependencies, libultra_ref_rx.so (the reference's four receive-path files) and libultra_ref.so not; .hip builds
    take the product's factory, .pimpl builds the reference's."""
    import subprocess
    from _refprogs import programs, name_of
    checked = 0
    for src, kind in programs():
        for v in VARIANTS[kind]:
            if v == "ref":
                continue
            path = exe(name_of(src), v)
            require(path)
            deps = subprocess.run(["ldd", str(path)], capture_output=True, text=True).stdout
            assert "libultra_hip_rx.so" in deps and "libultra_hip.so" in deps, (path.name, deps)
            assert "libultra_ref_rx.so" not in deps and "libultra_ref.so " not in deps, (path.name, deps)
            assert ("libultra_hip_factory.so" in deps) == (v == "hip"), (path.name, deps)
            assert ("libultra_ref_factory.so" in deps) == (v == "pimpl"), (path.name, deps)
            checked += 1
    assert checked >= 45
