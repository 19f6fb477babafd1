import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_terminal_summary(terminalreporter):
    """ONE line for every test skipped because a checker binary of oracle/_ref was not built (tests/_refprogs.py: with
    oracle/_ref/MANIFEST present or ULTRA_REQUIRE_REF=1 those are failures instead)."""
    try:
        from _refprogs import SKIPPED
    except Exception:
        return
    if SKIPPED:
        terminalreporter.write_line(f"checker binaries missing: {len(SKIPPED)} tests SKIPPED for lack of oracle/_ref programs "
                                    f"(no oracle/_ref/MANIFEST, ULTRA_REQUIRE_REF unset) — first: {SKIPPED[0]}", yellow=True)


@pytest.fixture(scope="session")
def oracle():
    from oracle.bindings import oracle as get
    return get()


@pytest.fixture(scope="session")
def ref():
    """The compiled reference (oracle/_ref) — present in the build container, optional elsewhere."""
    from oracle import bindings
    if not bindings.have_ref():
        pytest.skip("oracle/_ref/libultra_ref.so not available (needs /root/reference to build)")
    return bindings.Ref()


@pytest.fixture(scope="session")
def hiplib():
    """libultra_hip.so — the product; built in-tree by __graft_entry__.build()."""
    from projectultra_amd import _lib
    if not _lib.LIB_PATH.exists():
        _lib.build()
    return _lib.lib()


GOLDEN = Path(__file__).resolve().parent / "golden"


def bits_equal(a, b):
    import numpy as np
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))
