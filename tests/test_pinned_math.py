"""pinned_math.h (the device restatement of glibc 2.35 sinf/cosf/atan2f/hypotf) compiled for the
host and compared with this machine's libm — the functions the reference binary calls.
The exhaustive run (all 2^32 floats, 0 mismatches) is recorded in profiles/r01_pinned_math_exhaustive.txt."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_quick_subset_matches_libm(tmp_path):
    exe = tmp_path / "pmc"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-pthread",
                           str(ROOT / "tools" / "pinned_math_check.cpp"), "-o", str(exe), "-lm"])
    out = subprocess.run([str(exe), "quick"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = dict(l.split()[0:1] + [l] for l in out.stdout.splitlines())
    for fn in ("sinf", "cosf", "sincosf", "atanf", "atan2f", "hypotf"):
        assert "mismatches=0" in lines[fn], lines[fn]


def test_exhaustive_record_is_clean():
    txt = (ROOT / "profiles" / "r01_pinned_math_exhaustive.txt").read_text()
    assert txt.count("mismatches=0") == 6 and "checked=4294967296" in txt
