"""The host code above the C-ABI (include/ultra_hip_waveform.hpp) under AddressSanitizer + UBSan and under ThreadSanitizer — in the
build container, without a GPU (the GPU box cannot run sanitized device code and the product library needs
a GPU, so the host code runs here against tests/stub/ultra_hip_teststub.cpp).

The stub is TEST-ONLY: "device" memory is heap memory (an adapter that hands a kernel a window one sample short is an ASan
report), a fake modem whose soft bits are a function of the samples each call received (tests/stub/adapter_driver.cpp checks
that the right samples reached the right call through every chunking, trim, reset, 2^29-sample rebase), strict argument
validation, and fault injection (every C-ABI call of an exchange fails once: no leak, no crash, no exception through the
IWaveform boundary).  It is compiled into pytest's temporary directory and linked into the driver by file name; nothing in
projectultra_amd/ or include/ can resolve it (checked below and in tests/test_abi.py).

Each sanitizer is first shown a canary it MUST report (a heap overflow, a data race): a toolchain whose sanitizer is inert
skips instead of passing vacuously."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
STUB = ROOT / "tests" / "stub"
FLAGS = {
    "asan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-D_GLIBCXX_ASSERTIONS", "-O1", "-g"],
    "tsan": ["-fsanitize=thread", "-O1", "-g"],
}
CANARY = {
    "asan": "#include <cstdlib>\nint main(int c, char**) { int* p = (int*)std::malloc(16); int v = p[4 + c]; std::free(p); return v & 1; }\n",
    "tsan": "#include <thread>\nint x; int main() { std::thread a([]{ for (int i = 0; i < 100000; ++i) x = x + 1; }), b([]{ for (int i = 0; i < 100000; ++i) x = x + 1; });"
            " a.join(); b.join(); return 0; }\n",
}
REPORT = ("AddressSanitizer", "ThreadSanitizer", "runtime error:", "LeakSanitizer", "Assertion")


@pytest.fixture(scope="module")
def drivers(tmp_path_factory):
    out = tmp_path_factory.mktemp("adapter_san")
    built = {}
    for kind, flags in FLAGS.items():
        canary = out / f"canary_{kind}.cpp"
        canary.write_text(CANARY[kind])
        exe = out / f"canary_{kind}"
        r = subprocess.run(["g++", "-std=c++20", *flags, str(canary), "-o", str(exe), "-lpthread"], capture_output=True, text=True)
        if r.returncode != 0:
            built[kind] = (None, f"g++ {flags[0]} does not link here: {r.stderr[-300:]}")
            continue
        c = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
        if not any(w in c.stderr for w in REPORT):
            built[kind] = (None, f"{flags[0]} is inert in this container (the canary's defect was not reported)")
            continue
        drv = out / f"adapter_driver_{kind}"
        subprocess.check_call(["g++", "-std=c++20", *flags, f"-I{ROOT / 'include'}", str(STUB / "ultra_hip_teststub.cpp"),
                               str(STUB / "adapter_driver.cpp"), "-o", str(drv), "-lpthread"])
        built[kind] = (drv, None)
    return built


def _run(drivers, kind, scenario, seed):
    drv, why = drivers[kind]
    if drv is None:
        pytest.skip(why)
    r = subprocess.run([str(drv), scenario, str(seed)], capture_output=True, text=True, timeout=600)
    reports = [l for l in r.stderr.splitlines() if any(w in l for w in REPORT) or l.startswith("FAIL")]
    assert not reports and r.returncode == 0, (scenario, seed, r.returncode, reports[:6], r.stderr[-1500:])
    assert f"{scenario} seed {seed}: 0 failures" in r.stdout


@pytest.mark.parametrize("scenario,seed", [("stream", 3), ("stream", 11), ("presynced", 5), ("decoder", 7), ("faults", 9), ("rebase", 1)])
def test_adapter_state_machine_under_asan_ubsan(drivers, scenario, seed):
    """One thread: buffers and indices (ASan on host vectors AND on the stub's "device" allocations, libstdc++ assertions on every
    vector index / erase), arithmetic (UBSan), leaks at exit (LeakSanitizer) — with the stub's soft bits proving that every call
    saw exactly the samples it should have."""
    _run(drivers, "asan", scenario, seed)


@pytest.mark.parametrize("seed", [2, 13])
def test_adapter_threads_under_tsan(drivers, seed):
    """ModemEngine's threading (modem_rx.cpp:18-36,153-256; modem_engine.cpp:812-827; modem_mode.cpp:133-240): a feeder on one
    demodulator, the GUI's getters on the same object from another thread, two threads building / using / destroying
    demodulators and decoders — all through the shared slot pool."""
    _run(drivers, "tsan", "threads", seed)


def test_adapter_threads_under_asan(drivers):
    _run(drivers, "asan", "threads", 4)


def test_the_stub_is_unreachable_from_the_product():
    """Nothing the product ships names the stub; the stub does not call itself libultra_hip; _lib.py loads one path only."""
    for p in list((ROOT / "projectultra_amd").rglob("*")) + list((ROOT / "include").rglob("*")) + [ROOT / "bench.py", ROOT / "__graft_entry__.py"]:
        if p.is_file() and p.suffix in (".py", ".h", ".hip", ".hpp", ".cpp", ""):
            assert "teststub" not in p.read_text(errors="ignore"), p
    assert not list(ROOT.glob("**/libultra_hip_teststub*")) and not list((ROOT / "projectultra_amd").glob("*teststub*"))
    lib = (ROOT / "projectultra_amd" / "_lib.py").read_text()
    assert 'PKG_DIR / "libultra_hip.so"' in lib
