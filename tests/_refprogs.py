"""The manifest of the reference's own programs (oracle/ref_programs.txt) and the rule for checker binaries that are missing.

oracle/Makefile builds every manifest row from the unmodified source where it lies under /root/reference and writes
oracle/_ref/MANIFEST (what it built, from which reference tree).  The binaries are git-ignored and reach the GPU box with the
working tree.  Rule ("make missing checker binaries a failure, not a skip"):

  * oracle/_ref/MANIFEST exists, or ULTRA_REQUIRE_REF=1  ->  a test that needs a binary the box lacks FAILS;
  * neither                                                 ->  it skips, and the session prints ONE summary line with the count.
"""
import os
import subprocess
import threading
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
ORACLE = ROOT / "oracle"
REFDIR = ORACLE / "_ref"
TOOLS = REFDIR / "tools"
MANIFEST_TXT = ORACLE / "ref_programs.txt"
BUILT = REFDIR / "MANIFEST"

VARIANTS = {"core": ("ref", "hip"), "own": ("ref", "hip"), "engine": ("ref", "pimpl", "hip"), "ownengine": ("ref", "pimpl", "hip")}
SKIPPED = []                                   # (test id, file) of every skip this session: conftest prints the count


def _rows():
    for line in MANIFEST_TXT.read_text().splitlines():
        line = line.strip()
        if not line or line[0] in "#!":
            continue
        cols = line.split()
        assert cols[1] in VARIANTS and cols[2:] in ([], ["san"]), cols
        yield cols[0], cols[1], cols[2:] == ["san"]


def programs():
    """[(source relative to the reference root — or to oracle/ for own kinds —, kind)] in manifest order."""
    return [(src, kind) for src, kind, _ in _rows()]


def hardened():
    """Names of the programs that also have a .san build (drop-ins under UBSan + libstdc++ assertions + _FORTIFY_SOURCE=3)."""
    return [name_of(src) for src, _, san in _rows() if san]


def exclusions():
    """{source: reason} from the `!exclude <source> <reason>` rows."""
    out = {}
    for line in MANIFEST_TXT.read_text().splitlines():
        if line.startswith("!exclude"):
            _, src, reason = line.split(None, 2)
            out[src] = reason.strip()
    return out


def name_of(src):
    return Path(src).stem


def kind_of(name):
    for src, kind in programs():
        if name_of(src) == name:
            return kind
    raise KeyError(name)


def expected_files():
    """Every file oracle/Makefile's `tools` target leaves under oracle/_ref (relative to it)."""
    out = ["libultra_ref.so", "rx_pipeline_harness"]
    for src, kind in programs():
        out += [f"tools/{name_of(src)}.{v}" for v in VARIANTS[kind]]
    out += [f"tools/{n}.san" for n in hardened()]
    return out


def built_files():
    """Relative paths oracle/_ref/MANIFEST lists (empty when there is no manifest)."""
    if not BUILT.exists():
        return []
    return [l.split(":", 1)[1].strip() for l in BUILT.read_text().splitlines() if l.startswith("file:")]


def strict():
    return BUILT.exists() or os.environ.get("ULTRA_REQUIRE_REF") == "1"


def require(*paths):
    """The given checker binaries, or: fail (strict) / skip with a record (otherwise)."""
    missing = [p for p in paths if not Path(p).exists()]
    if not missing:
        return
    what = ", ".join(str(Path(p).relative_to(ROOT)) for p in missing)
    if strict():
        pytest.fail(f"checker binary missing: {what} — oracle/_ref/MANIFEST says the reference's programs were built for this "
                    f"tree (or ULTRA_REQUIRE_REF=1); rebuild with `make -C oracle tools` where /root/reference exists", pytrace=False)
    SKIPPED.append(what)
    pytest.skip(f"{what} not built (needs /root/reference: `make -C oracle tools`)")


def exe(name, variant):
    return TOOLS / f"{name}.{variant}"


def run(path, args, timeout=900, cwd=None, env=None):
    """(rc, stdout, stderr) of one run.  The streams go to FILES, not pipes: the reference's engine logs thousands of DEBUG lines
    from its acquisition and decode threads, a pipe holds 64 KB, and a thread blocked in fprintf while this process drains the pipe
    shifts exactly the timing its acquisition loop's snapshot race depends on (with pipes the reference's own build lost frames in
    a third of its runs under pytest; with files, as from a shell, in none of fourteen)."""
    import tempfile
    with tempfile.TemporaryFile(mode="w+b") as out, tempfile.TemporaryFile(mode="w+b") as err:
        r = subprocess.run([str(path)] + list(args), stdout=out, stderr=err, timeout=timeout, cwd=cwd, env=env)
        out.seek(0); err.seek(0)
        return r.returncode, out.read().decode(errors="replace"), err.read().decode(errors="replace")


def run_all(name, args, variants, timeout=900, cwd=None, env=None, one_at_a_time=False):
    """The program's builds side by side (one process each): {variant: (rc, stdout, stderr)}.
    one_at_a_time: programs whose outcome depends on how their own threads get scheduled run alone on the box."""
    if one_at_a_time:
        return {v: run(exe(name, v), args, timeout=timeout, cwd=cwd, env=env) for v in variants}
    out, errs = {}, []

    def one(v):
        try:
            out[v] = run(exe(name, v), args, timeout=timeout, cwd=cwd, env=env)
        except Exception as e:                                        # noqa: BLE001 - reported below, in the test's thread
            errs.append((v, e))

    threads = [threading.Thread(target=one, args=(v,)) for v in variants]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errs:
        raise AssertionError(f"{name} {args}: {errs}")
    return out
