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
    for fn in ("sinf", "cosf", "sincosf", "sincosf_bounded", "atanf", "atan2f", "hypotf", "right_angle_test", "logf"):
        assert "mismatches=0" in lines[fn], lines[fn]


def test_exhaustive_record_is_clean():
    txt = (ROOT / "profiles" / "r01_pinned_math_exhaustive.txt").read_text()
    assert txt.count("mismatches=0") == 8 and "checked=4294967296" in txt and "mismatches=0" in txt.splitlines()[-1]
    assert "right_angle_test checked=2147483904 mismatches=0" in txt


def test_exhaustive_record_round2_includes_logf():
    """logf (the Box-Muller of the LLR stimulus generator) over all 2^32 floats, next to a re-run of the others."""
    txt = (ROOT / "profiles" / "r02_pinned_math_exhaustive.txt").read_text()
    assert "logf checked=4294967296 mismatches=0" in txt and txt.count("mismatches=0") == 9


def test_phase_table_equals_serial_recurrence(tmp_path):
    """phase_table.h (closed-form jumps of the float CFO phase recurrence of toBaseband,
    channel_equalizer.cpp:43-50) against the serial recurrence, position by position: random starts and
    increments, ties, +-pi wraps, zero crossings, denormals, small table capacities."""
    exe = tmp_path / "ptc"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-pthread",
                           str(ROOT / "tools" / "phase_table_check.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe), "400000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches=0" in out.stdout.splitlines()[0], out.stdout


def test_ldpc_plan_is_conflict_free(tmp_path):
    """The host-side LDPC execution plan (csrc/host_tables.h): every information edge has its own LDS
    word and every half-wave access of both decoder steps touches 32 distinct banks, for all six rates."""
    exe = tmp_path / "lpc"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + str(ROOT / "projectultra_amd" / "csrc"),
                           str(ROOT / "tools" / "ldpc_plan_check.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("conflicts 0") == 6, out.stdout


def test_ldpc_kernel_instances_cover_the_six_profiles(tmp_path):
    """The LDPC kernel is instantiated per degree profile (csrc/ultra_hip.hip, UH_LDPC_LAUNCH); a plan
    whose profile matches no instance is refused at run time.  Checked here without a GPU: every rate's
    profile as the plan builder computes it names one instance of the dispatcher, with the slots sorted
    by degree and row_id a permutation (the checker's own exit code)."""
    import re
    exe = tmp_path / "lpc"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + str(ROOT / "projectultra_amd" / "csrc"),
                           str(ROOT / "tools" / "ldpc_plan_check.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    profiles = re.findall(r"profile: rmax (0x[0-9a-f]+ull) rmin (0x[0-9a-f]+ull) vmax (0x[0-9a-f]+ull) "
                          r"vmin (0x[0-9a-f]+ull) row_identity (\d)", out.stdout)
    assert len(profiles) == 6, out.stdout
    src = (ROOT / "projectultra_amd" / "csrc" / "ultra_hip.hip").read_text()
    linear = re.findall(r"^\s+linear (\d)", out.stdout, re.M)
    assert len(linear) == 6, out.stdout
    launches = re.findall(r"UH_LDPC_LAUNCH\(\d+, \d+, (0x[0-9a-f]+ull), (0x[0-9a-f]+ull), (0x[0-9a-f]+ull), "
                          r"(0x[0-9a-f]+ull), (true|false), (true|false), \d\);", src)
    have = {(a, b, c, d, rid == "true", lin == "true") for a, b, c, d, rid, lin in launches}
    for (rmax, rmin, vmax, vmin, ident), lin in zip(profiles, linear):
        assert (rmax, rmin, vmax, vmin, ident == "0", lin == "1") in have, (rmax, rmin, vmax, vmin, ident, lin)


def test_ldpc_totals_plan_and_emulation(tmp_path):
    """The totals LDPC kernels' plans (embedded placements, csrc/ldpc_placement.h and — round 4, the codes with irregular
    rows — csrc/ldpc_placement_low.h -> build_ldpc_tplan): valid for all six codes, and a lane-by-lane CPU emulation of the
    kernel over the plan (per-round degree profiles, plane bases, pad words) decodes 200 noisy codewords per rate exactly as the
    reference's decodeBP (iterations, success, every bit)."""
    import re
    exe = tmp_path / "tpc"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I" + str(ROOT / "projectultra_amd" / "csrc"),
                           str(ROOT / "tools" / "ldpc_tplan_check.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("plan valid") == 6 and out.stdout.count(": 0 mismatches of 200") == 6, out.stdout
    extra = {int(r): int(c) for r, c in re.findall(r"rate (\d): plan valid.*cost (\d+) cycles", out.stdout)}
    assert extra[4] <= 4 and extra[5] <= 4, extra          # R3/4 and R5/6: (nearly) conflict-free gathers
    assert extra[1] <= 4 and extra[2] <= 4 and extra[0] <= 48, extra      # R1/3, R1/2 likewise; R1/4 (13 edges per variable) keeps 44

