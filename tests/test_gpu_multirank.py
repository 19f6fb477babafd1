"""The N > 1 path, executed: `python bench.py --gpus N` starts its own N ranks (a child torch.distributed.run), each rank
takes its contiguous shard (independent trials, /root/reference/tools/test_otfs_vs_ofdm.cpp:112-125) and the
eight counters are all-reduced.  One card is all the box has, so the ranks SHARE it and the 64 bytes travel through host
memory (--backend gloo); everything else — sharding, generators keyed on the global trial index, the kernels, the counting
launch, the collective's place in the step — is the code an 8-GPU RCCL run executes.

What must hold: the collective really spans N ranks, every trial is counted exactly once, and the eight counters of N = 2,
3 (ragged shards) and 4 equal N = 1's EXACTLY.  (No more than four ranks: the box allows six processes on its card and this
test process is one of them.)"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(*argv, expect_rc=0, env=None):
    cmd = [sys.executable, str(ROOT / "bench.py"), *argv, "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=e)
    assert p.returncode == expect_rc, (p.returncode, p.stderr[-3000:])
    if expect_rc:
        return p
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one JSON line, got {len(lines)}: {p.stdout[-2000:]}"
    return json.loads(lines[0])


def test_cfg3_strong_scaling_counters_equal_one_rank():
    """One 2^17-frame cfg3 batch (Watterson, 30 dB) sharded over 1, 2, 3 and 4 ranks."""
    total = 1 << 17
    one = run_bench("--gpus", "1", "--total-frames", str(total))
    assert one["n_gpus"] == 1 and one["trials_counted"] == total and one["counters"][0] == total
    assert one["counters"][1] > 0 and one["counters"][5] > 0            # frame errors and iterations: the counters carry information
    for n in (2, 3, 4):
        got = run_bench("--gpus", str(n), "--backend", "gloo", "--total-frames", str(total))
        assert got["n_gpus"] == n and got["scaling"] == "strong"
        assert got["collective"]["world_size"] == n and got["collective"]["backend"] == "gloo"
        assert got["trials_counted"] == total
        assert got["counters"] == one["counters"], (n, got["counters"], one["counters"])
        assert got["config"]["trials_per_step_all_gpus"] == total


def test_cfg4_points_equal_one_rank():
    """The R1/4 Es/N0 sweep (configs[3]): 42 points x 8,192 codewords, one all-reduce per point; every point's counters."""
    one = run_bench("--config", "cfg4", "--gpus", "1", "--frames", "8192")
    assert len(one["counters_per_point"]) == 42
    for n in (2, 4):
        got = run_bench("--config", "cfg4", "--gpus", str(n), "--backend", "gloo", "--frames", str(8192 // n))
        assert got["collective"]["world_size"] == n and got["collective"]["per_step"] == 42
        assert got["trials_counted"] == one["trials_counted"] == 42 * 8192
        assert got["counters_per_point"] == one["counters_per_point"], n


def test_cfg5_grid_equals_one_rank():
    """The adaptive-mode grid (configs[4]): 30 cells x 11 points x 192 frames, one all-reduce of the whole block."""
    one = run_bench("--config", "cfg5", "--gpus", "1", "--frames", "192")
    got = run_bench("--config", "cfg5", "--gpus", "2", "--backend", "gloo", "--frames", "96")
    assert got["collective"]["world_size"] == 2
    assert got["trials_counted"] == one["trials_counted"] == 330 * 192
    assert got["counters_per_point"] == one["counters_per_point"]


def test_wrong_world_size_is_refused():
    """A launcher that started another number of ranks than --gpus names gets no line (never a silent one-GPU result)."""
    p = run_bench("--gpus", "2", "--total-frames", "4096", expect_rc=1,
                  env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29511"})
    assert "refusing" in p.stderr and not p.stdout.strip()
