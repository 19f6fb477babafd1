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
        # every rank's own clock between the two barriers, and the fall-back paths its contexts took: a straggler or a rank on a
        # slower chain shows in the line (four ranks is what the box's process guard allows beside this process)
        ranks = got["ms_per_step_ranks"]
        assert len(ranks["all"]) == n and ranks["min"] <= ranks["max"] <= got["ms_per_step"] * 1.0001 + 1e-9
        assert got["roofline"]["path_status"]["default_path"] is True and got["roofline"]["path_status"]["per_rank_flags"] == [0] * n


def test_cfg3_headline_batch_eight_ranks_rehearsed_in_one_process():
    """north_star's N = 8 case at FULL size — the 2^20-frame batch, a rank's share 2^17 — without eight processes (the box
    allows six on its card): bench.py's own workload object built for (rank r, world 8), r = 0..7, one step each, the eight
    counter blocks summed the way the all-reduce sums them; must equal the N = 1 batch's counters exactly.  Everything an
    8-GPU run executes except the transport: shard_range(2^20, r, 8), the generators keyed on the global trial index, the
    kernels at a 2^17 launch, the counting launch."""
    import argparse
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    total = 1 << 20
    args = argparse.Namespace(config="cfg3", gpus=8, steps=1, warmup=0, frames=0, total_frames=total, snr_db=None, raw_channel="awgn",
                              no_cpu_baseline=True, cpu_sample=0, backend="gloo", sync_collective=False, no_build=True)

    def one_step(rank, world):
        wl = bench.ModemWorkload("cfg3", args, rank, world, torch)
        wl.step(lambda t: None)
        wl.ring.drain()
        torch.cuda.synchronize()
        c = wl.counters.reshape(-1, 8).sum(dim=0).cpu()
        n, st = wl.n, wl.ctx.status()
        del wl
        torch.cuda.empty_cache()
        return c, n, st

    whole, n1, st1 = one_step(0, 1)
    assert n1 == total and int(whole[0]) == total and int(whole[1]) > 0 and st1["paths"] == []
    acc = torch.zeros(8, dtype=whole.dtype)
    for r in range(8):
        c, n, st = one_step(r, 8)
        assert n == total // 8 and int(c[0]) == n and st["paths"] == [], (r, n, st)
        acc += c
    assert acc.tolist() == whole.tolist(), (acc.tolist(), whole.tolist())


def test_a_rank_without_a_device_is_refused_not_moved_to_device_0():
    """The device-selection code with an index other than 0 — all one card can execute of it: under RCCL a LOCAL_RANK beyond the
    visible devices gets no line (never a second rank silently sharing device 0), and the drop-ins' ULTRA_HIP_DEVICE
    (hip_ofdm_demodulator.cpp) naming a device the box does not have fails loudly instead of running on device 0."""
    import torch
    n_dev = torch.cuda.device_count()                                 # 1 on the GPU box; the first index that does NOT exist on any box
    p = run_bench("--gpus", str(n_dev + 1), "--total-frames", "4096", expect_rc=1,
                  env={"RANK": str(n_dev), "LOCAL_RANK": str(n_dev), "WORLD_SIZE": str(n_dev + 1), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29513"})
    assert "has no GPU of its own" in p.stderr and not p.stdout.strip()
    # the same through HIP_VISIBLE_DEVICES: an empty list leaves the rank no device at all
    p = run_bench("--gpus", "1", "--total-frames", "4096", expect_rc=1, env={"HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""})
    assert not p.stdout.strip()
    from _refprogs import exe, require
    tool = exe("test_nvis_mode", "hip")
    require(tool)
    r = subprocess.run([str(tool), "--snr", "30", "--trials", "1"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ULTRA_HIP_DEVICE=str(n_dev)))
    assert "ultra_hip" in r.stderr and ("FAILED" in r.stderr or "no device" in r.stderr.lower() or r.returncode != 0), (r.returncode, r.stderr[-600:])
    assert "100.0%" not in r.stdout and "100%" not in r.stdout, "a device that does not exist must not decode anything"


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
