"""oracle/ref_programs.txt against the reference tree and against what was built (CPU).

1. Coverage: every file under /root/reference/{tests,tools} that names a class the drop-ins replace (OFDMDemodulator,
   LDPCDecoder) or a caller north_star names (ModemEngine, WaveformFactory, RxPipeline, the interleavers the drop-in defines)
   is in the manifest or in its exclusion table with a reason.  A new program in the reference turns this red.
2. The Makefile builds exactly what the manifest lists; oracle/_ref/MANIFEST, when present, lists exactly that too and every
   listed file exists — so a deleted binary is a failure here (and in the GPU tests that need it), not a silent skip.
3. Every manifest program is exercised by a test case."""
import re
from pathlib import Path

import pytest

import _refprogs as rp

REF = Path("/root/reference")
CLASSES = re.compile(r"\b(OFDMDemodulator|LDPCDecoder|ModemEngine|WaveformFactory|RxPipeline|ChannelInterleaver|Interleaver)\b")


def test_manifest_rows_are_well_formed():
    rows = rp.programs()
    assert len(rows) >= 35
    names = [rp.name_of(s) for s, _ in rows]
    assert len(set(names)) == len(names), "binary names must be unique"
    for src, kind in rows:
        root = rp.ORACLE if kind.startswith("own") else REF
        if root.exists():
            assert (root / src).is_file(), f"{src}: no such source under {root}"


@pytest.mark.skipif(not REF.is_dir(), reason="/root/reference absent (GPU box): the coverage check runs in the build container")
def test_manifest_covers_every_reference_program_that_names_a_replaced_class():
    listed = {s for s, k in rp.programs() if not k.startswith("own")}
    excluded = rp.exclusions()
    uncovered = []
    for d in ("tests", "tools"):
        for f in sorted((REF / d).glob("*.cpp")):
            rel = f"{d}/{f.name}"
            if CLASSES.search(f.read_text(errors="replace")) and rel not in listed and rel not in excluded:
                uncovered.append(rel)
    assert not uncovered, (f"{uncovered}: construct or call a replaced class but are neither in oracle/ref_programs.txt nor in its "
                           f"`!exclude` table (add a row — or an exclusion with its reason)")
    for rel, reason in excluded.items():
        assert (REF / rel).is_file(), f"!exclude {rel}: no such file"
        assert len(reason) > 20, f"!exclude {rel}: give the reason"
        assert rel not in listed


def test_makefile_builds_what_the_manifest_lists():
    """`make -n tools` names every binary of every row (and only manifest rows)."""
    import subprocess
    if not REF.is_dir():
        pytest.skip("/root/reference absent: nothing to build here")
    out = subprocess.run(["make", "-C", str(rp.ORACLE), "-n", "-B", "tools"], capture_output=True, text=True).stdout
    linked = set(re.findall(r"-o (_ref/tools/\S+)", out))
    want = {"_ref/" + f for f in rp.expected_files() if f.startswith("tools/")}
    assert linked == want, (sorted(want - linked)[:5], sorted(linked - want)[:5])


def test_built_manifest_matches_and_nothing_is_missing():
    if not rp.BUILT.exists():
        if rp.REFDIR.exists() and REF.is_dir():
            pytest.fail("oracle/_ref exists but has no MANIFEST: run `make -C oracle tools` (or __graft_entry__.build())")
        pytest.skip("oracle/_ref/MANIFEST absent (no compiled reference on this box)")
    built = rp.built_files()
    assert sorted(built) == sorted(rp.expected_files()), (sorted(set(rp.expected_files()) - set(built))[:5], sorted(set(built) - set(rp.expected_files()))[:5])
    missing = [f for f in built if not (rp.REFDIR / f).exists()]
    assert not missing, f"oracle/_ref/MANIFEST lists files that are not there: {missing[:6]}"
    text = rp.BUILT.read_text()
    assert re.search(r"^reference_id: [0-9a-f]{40}$", text, re.M), "the manifest must say which reference tree it was built from"


def test_every_manifest_program_has_a_test_case():
    here = Path(__file__).resolve().parent
    text = "".join((here / f).read_text() for f in ("test_gpu_ref_programs.py", "test_gpu_pimpl.py", "test_gpu_live_latency.py")
                   if (here / f).exists())
    text += (rp.ROOT / "tools" / "collect_round.sh").read_text() if (rp.ROOT / "tools" / "collect_round.sh").exists() else ""
    unused = [rp.name_of(s) for s, _ in rp.programs() if f'"{rp.name_of(s)}"' not in text and f"{rp.name_of(s)}." not in text]
    assert not unused, f"{unused}: in the manifest but no test runs them"


def test_missing_binary_is_a_failure_when_the_manifest_exists(tmp_path, monkeypatch):
    """require(): strict with a manifest (or ULTRA_REQUIRE_REF=1) — the outcome is `failed`, not `skipped`."""
    ghost = rp.TOOLS / "no_such_program.hip"
    monkeypatch.setattr(rp, "BUILT", tmp_path / "MANIFEST")
    monkeypatch.delenv("ULTRA_REQUIRE_REF", raising=False)
    with pytest.raises(pytest.skip.Exception):
        rp.require(ghost)
    rp.SKIPPED.clear()                                                # (this test's own probe is not a real skip)
    (tmp_path / "MANIFEST").write_text("file: tools/no_such_program.hip\n")
    with pytest.raises(pytest.fail.Exception):
        rp.require(ghost)
    monkeypatch.setattr(rp, "BUILT", tmp_path / "absent")
    monkeypatch.setenv("ULTRA_REQUIRE_REF", "1")
    with pytest.raises(pytest.fail.Exception):
        rp.require(ghost)
