"""The CMake form of the link-time drop-in, executed (integration/CMakeLists.txt).

INTEGRATION.md 0 tells a maintainer to take four sources out of the reference's `ultra_core` target and put two in.  Here that is
done to the reference's OWN, unmodified CMakeLists.txt — added as a subdirectory of a wrapper project, its target edited from
outside — so the claim "every target that links ultra_core then runs its receive path on the MI355X" is a build that configures,
compiles and links, not a snippet:

  * CPU (build container: cmake + /root/reference): configure and build `test_nvis_mode` in a temporary directory; the archive holds
    the drop-ins' classes and none of the reference's receive-path functions; the program depends on libultra_hip.so.
  * GPU: the three programs integration/Makefile keeps under integration/_build/bin — the headline harness, the SURVEY 8c ctest pin,
    and ModemEngine's PRIMARY regression tool — print what the reference's own build prints (oracle/_ref/tools/*.ref, plain g++).

(This is the integration demonstration; the checkers under oracle/_ref are NOT built this way.)"""
import shutil
import subprocess
from pathlib import Path

import pytest

from _refprogs import TOOLS, require, run

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
BUILT = ROOT / "integration" / "_build" / "BUILT"
BIN = ROOT / "integration" / "_build" / "bin"


@pytest.mark.skipif(not REF.is_dir() or shutil.which("cmake") is None, reason="needs cmake and /root/reference (the build container)")
def test_reference_cmake_configures_and_builds_over_the_drop_ins(tmp_path, hiplib):
    b = tmp_path / "b"
    subprocess.check_call(["cmake", "-S", str(ROOT / "integration"), "-B", str(b), f"-DULTRA_REFERENCE={REF}"], stdout=subprocess.DEVNULL)
    r = subprocess.run(["cmake", "--build", str(b), "-j8", "-t", "test_nvis_mode"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    log = r.stdout
    assert "hip_ofdm_demodulator.cpp" in log and "hip_ldpc_decoder.cpp" in log
    for gone in ("ofdm/demodulator.cpp", "ofdm/channel_equalizer.cpp", "ofdm/ofdm_sync.cpp", "fec/ldpc_decoder.cpp"):
        assert gone not in log, f"{gone} was compiled: the target still holds the reference's receive path"
    archive = next(b.rglob("libultra_core.a"))
    syms = subprocess.run(["nm", "-C", str(archive)], capture_output=True, text=True).stdout
    assert "ultra_hip::HipOfdmDemodulator" in syms and "ultra_hip::HipLDPCDecoder" in syms
    assert "ultra::OFDMDemodulator::Impl::updateChannelEstimate" not in syms and "ultra::LDPCDecoder::Impl::decodeBP" not in syms
    exe = next(p for p in b.rglob("test_nvis_mode") if p.is_file())
    assert "libultra_hip.so" in subprocess.run(["ldd", str(exe)], capture_output=True, text=True).stdout


def test_built_list_is_complete():
    if not BUILT.exists():
        pytest.skip("integration/_build absent (make -C integration, where cmake and /root/reference exist)")
    files = [l.split(":", 1)[1].strip() for l in BUILT.read_text().splitlines() if l.startswith("file:")]
    assert sorted(files) == ["bin/test_iwaveform", "bin/test_multiblock_ldpc", "bin/test_nvis_mode"]
    assert all((BUILT.parent / f).exists() for f in files), "integration/_build/BUILT lists files that are not there"


CASES = [("test_nvis_mode", ["--snr", "22", "--trials", "8"]), ("test_nvis_mode", ["--snr", "30", "--trials", "6"]),
         ("test_multiblock_ldpc", []),
         ("test_iwaveform", ["--snr", "17", "--cfo", "30", "--channel", "awgn", "-w", "ofdm_chirp", "--frames", "5"])]


@pytest.mark.gpu
@pytest.mark.parametrize("name,args", CASES, ids=[f"{n}{'_'.join([''] + a)}" for n, a in CASES])
def test_cmake_built_program_equals_the_reference_build(name, args, tmp_path):
    exe, ref = BIN / name, TOOLS / f"{name}.ref"
    if not BUILT.exists():
        pytest.skip("integration/_build absent (make -C integration, where cmake and /root/reference exist)")
    assert exe.exists(), f"integration/_build/BUILT exists but {exe} does not"
    require(ref)
    rc_ref, out_ref, _ = run(ref, args, cwd=tmp_path)
    rc, out, err = run(exe, args, cwd=tmp_path)
    keep = lambda o: [l for l in o.splitlines() if not l.startswith("[CHIRP-RX] Dual chirp") and not l.startswith("[CHIRP-RX] Position")]   # noqa: E731
    a, b = keep(out_ref), keep(out)
    assert a and a == b, ([(x, y) for x, y in zip(a, b) if x != y][:2], len(a), len(b), err[-600:])
    assert rc == rc_ref
