"""Host-side mirror of the reference interface + multi-rank sharding (CPU only)."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_modem_config_mirror():
    """include/ultra/types.hpp:139-234,262-367 — defaults, helpers and presets."""
    from projectultra_amd import CodeRate, ModemConfig, Modulation, presets, getBitsPerSymbol
    c = ModemConfig()
    assert (c.fft_size, c.num_carriers, c.getCyclicPrefix(), c.getSymbolDuration(), c.getDataCarriers()) == (512, 30, 48, 564, 15)
    n = presets.nvis_mode()
    assert (n.fft_size, n.num_carriers, n.getCyclicPrefix(), n.getSymbolDuration(), n.use_pilots) == (1024, 59, 96, 1120, False)
    q = n.with_mode(Modulation.QAM16, CodeRate.R3_4); q.pilot_spacing = 4
    assert q.use_pilots and q.getDataCarriers() == 44                 # 59 - ceil(59/4)
    assert presets.high_throughput().pilot_spacing == 4 and presets.turbo().getCyclicPrefix() == 32
    assert [getBitsPerSymbol(m) for m in (Modulation.DBPSK, Modulation.DQPSK, Modulation.D8PSK, Modulation.QAM16,
                                          Modulation.QAM32, Modulation.QAM64, Modulation.QAM256)] == [1, 2, 3, 4, 5, 6, 8]
    assert abs(q.getTheoreticalThroughput(Modulation.QAM16, CodeRate.R3_4) - 44 * 4 * 0.75 * 48000 / 1120) < 1e-6


def test_c_config_matches_harness_config():
    """make_c_config(ModemConfig) == the POD the oracle tests use for the same mode."""
    import ctypes as C
    from oracle.bindings import make_config
    from projectultra_amd import CodeRate, Entry, ModemConfig, Modulation, presets
    from projectultra_amd.engine import make_c_config
    mc = presets.nvis_mode().with_mode(Modulation.QAM16, CodeRate.R3_4); mc.pilot_spacing = 4
    a, b = make_c_config(mc), make_config(1024, "QAM16", "R3_4")
    assert bytes(a) == bytes(b)
    mc = ModemConfig().with_mode(Modulation.DQPSK, CodeRate.R1_2)
    assert bytes(make_c_config(mc)) == bytes(make_config(512, "DQPSK", "R1_2"))
    assert bytes(make_c_config(mc, entry=Entry.PRESYNCED)) == bytes(make_config(512, "DQPSK", "R1_2", entry=1))


def test_waveform_geometry_getters():
    """OFDMChirpWaveform geometry (src/waveform/ofdm_chirp_waveform.cpp:20-31,298-331; the values the compiled reference
    reports under RxPipeline: tests/test_gpu_rx_pipeline.py)."""
    from projectultra_amd import CodeRate, ModemConfig, Modulation
    from projectultra_amd.waveform import HipOfdmWaveform
    w = HipOfdmWaveform.__new__(HipOfdmWaveform)           # geometry needs no device
    w._config = HipOfdmWaveform._chirp_config(ModemConfig(use_pilots=True))      # QPSK with pilots asked for ...
    assert w._config.modulation == Modulation.DQPSK and not w._config.use_pilots  # ... the chirp mode is DQPSK without
    assert w.getSamplesPerSymbol() == 564 and w.getPreambleSamples() == 57600 + 1128
    assert w.getMinSamplesForFrame() == 2 * 564 + 11 * 564   # DQPSK, 30 data carriers -> 11 symbols
    assert w.getCarrierCount() == 30


def test_shard_range_partitions_exactly():
    from projectultra_amd.montecarlo import shard_range
    for n in (0, 1, 7, 1 << 20, (1 << 22) + 3):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from projectultra_amd.montecarlo import run_sharded, counters_dict
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
N = 1001
def run_shard(lo, hi):      # stand-in for the per-rank GPU pass: counters derived from frame ids
    ids = torch.arange(lo, hi, dtype=torch.int64)
    return torch.stack([torch.tensor(hi - lo), (ids % 7 == 0).sum(), (ids % 5).sum(), torch.tensor((hi - lo) * 480),
                        (ids % 11 == 0).sum(), (ids % 3).sum(), torch.tensor(0), torch.tensor(0)]).to(torch.int64)
c = run_sharded(N, rank, world, run_shard)
ids = torch.arange(N)
want = [N, int((ids % 7 == 0).sum()), int((ids % 5).sum()), N * 480, int((ids % 11 == 0).sum()), int((ids % 3).sum()), 0, 0]
assert c.tolist() == want, (c.tolist(), want)
d = counters_dict(c)
assert d["frames"] == N and abs(d["fer"] - want[1] / N) < 1e-12
dist.destroy_process_group()
print("rank", rank, "ok")
'''


RING_WORKER = r'''
import os, sys, importlib.util, torch, torch.distributed as dist
spec = importlib.util.spec_from_file_location("bench", os.path.join(sys.argv[1], "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
ring = bench.CounterRing(torch, (3, 8), device="cpu")
seen = []
for step in range(7):                       # bench.py's step: take the next block, fill it, issue the all-reduce, move on
    c = ring.next()
    if step >= 2:                           # the block of step - 2 comes back reduced: next() waited for its collectives
        assert c.tolist() == seen[step - 2], (step, c.tolist(), seen[step - 2])
    c.zero_()
    for point in range(3):
        c[point] += torch.arange(8) * (step + 1) + point + rank
        ring.note(dist.all_reduce(c[point], op=dist.ReduceOp.SUM, async_op=True))
    seen.append([[world * (k * (step + 1) + point) + sum(range(world)) for k in range(8)] for point in range(3)])
ring.drain()
assert ring.bufs[ring.k].tolist() == seen[-1] and all(not w for w in ring.works)
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_counter_ring_overlaps_the_allreduce_gloo(tmp_path):
    """bench.py's CounterRing (two counter blocks used by alternate steps, the all-reduce of a step issued asynchronously and
    awaited when its block comes round again): world size 2 on gloo, three collectives per step as in the cfg4 sweep."""
    script = tmp_path / "ring_worker.py"
    script.write_text(RING_WORKER)
    port = 29500 + os.getpid() % 2000 + 7
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), str(ROOT)]
    r = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("ok") == 2


@pytest.mark.parametrize("world", [2, 3, 8])
def test_counter_allreduce_across_ranks_gloo(tmp_path, world):
    """N>1 path: contiguous shards, one all-reduce of the 8 counters (gloo on CPU; RCCL on the GPUs) — at north_star's 2, at a
    ragged 3, and at its 8 (the collective and the shard arithmetic at the rank count no single card can host)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = 29500 + os.getpid() % 2000 + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), str(ROOT)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("ok") == world


def test_channel_interleaver_mirror_matches_oracle(oracle):
    """projectultra_amd.ChannelInterleaver (host mirror of src/fec/ldpc_decoder.cpp:547-617) against the
    oracle's restatement, which tests/test_oracle_vs_ref.py pins to the compiled reference."""
    from projectultra_amd.fec import ChannelInterleaver
    x = np.random.default_rng(3).normal(size=648).astype(np.float32)
    for bps in (30, 60, 116, 176, 236, 300, 472):
        il = ChannelInterleaver(bps)
        perm, inv = oracle.channel_interleaver_perm(bps)
        assert np.array_equal(il.permutation, perm) and np.array_equal(il.inverse_permutation, inv)
        assert np.array_equal(il.deinterleave(il.interleave(x)), x)
        short = il.deinterleave(x[:100])
        want = np.zeros(648, np.float32); want[inv[:100]] = x[:100]
        assert np.array_equal(short, want)


def test_row_column_interleaver_mirror_matches_oracle(oracle):
    """projectultra_amd.Interleaver (host mirror of ultra::Interleaver, ldpc_decoder.cpp:454-540) against the oracle's
    restatement (pinned to the compiled reference by tests/test_oracle_vs_ref.py): deinterleave(soft) for the 6 x 108
    layout of tools/test_throughput.cpp and two others, and interleave as its inverse."""
    from projectultra_amd import Interleaver
    x = np.random.default_rng(4).normal(size=648).astype(np.float32)
    for rows, cols in ((6, 108), (108, 6), (24, 27), (8, 81)):
        il = Interleaver(rows, cols)
        assert np.array_equal(il.deinterleave(x), oracle.interleaver_deinterleave(rows, cols, x))
        assert np.array_equal(il.deinterleave(il.interleave(x)), x)
        assert sorted(il.permutation.tolist()) == list(range(648))



def test_evidence_is_tied_to_the_kernel_sources(tmp_path):
    """profiles/traffic.json is quoted by bench.py only for the kernels it was collected on: _lib.source_hash() covers every
    file the library is built from and changes with any of them."""
    import shutil
    from projectultra_amd import _lib
    h = _lib.source_hash()
    assert len(h) == 16 and h == _lib.source_hash()
    pkg = tmp_path / "projectultra_amd"
    shutil.copytree(_lib.CSRC_DIR, pkg / "csrc")
    (tmp_path / "include").mkdir()
    shutil.copy(_lib.PKG_DIR.parent / "include" / "ultra_hip.h", tmp_path / "include" / "ultra_hip.h")
    old = (_lib.CSRC_DIR, _lib.PKG_DIR)
    try:
        _lib.CSRC_DIR, _lib.PKG_DIR = pkg / "csrc", pkg
        assert _lib.source_hash() == h                       # same sources elsewhere: same hash
        with open(pkg / "csrc" / "demod_kernel.h", "a") as f:
            f.write("\n// touched\n")
        assert _lib.source_hash() != h
    finally:
        _lib.CSRC_DIR, _lib.PKG_DIR = old


def test_bench_headline_is_one_sharded_batch():
    """bench.py's default for the headline: ONE 2^20-frame batch per step, rank r takes shard_range(2^20, r, N) (strong
    scaling, north_star); --frames switches to the same batch size on every GPU (weak)."""
    import argparse
    sys.path.insert(0, str(ROOT))
    import bench
    from projectultra_amd.montecarlo import shard_range
    argv = sys.argv
    try:
        sys.argv = ["bench.py"]
        a = bench.parse()
        assert a.config == "cfg3" and a.frames == 0 and a.total_frames == 0 and a.gpus == 1
        sys.argv = ["bench.py", "--frames", "262144", "--gpus", "8"]
        b = bench.parse()
        assert b.frames == 262144
    finally:
        sys.argv = argv
    total = 1 << 20
    for world in (1, 2, 4, 8):
        spans = [shard_range(total, r, world) for r in range(world)]
        assert sum(hi - lo for lo, hi in spans) == total and all(hi - lo == total // world for lo, hi in spans)


def test_stream_scenarios_rebuild_deterministically():
    """tests/golden/stream.npz stores only the reference's outputs: the audio is rebuilt from fullsync.npz's frames."""
    from _util import STREAM_SCENARIOS, build_stream
    from conftest import GOLDEN
    g = np.load(GOLDEN / "fullsync.npz")
    want = np.load(GOLDEN / "stream.npz")
    frames = g["cfg2_dqpsk_r12__audio"]
    for sc, recipe in STREAM_SCENARIOS.items():
        a1, c1 = build_stream(frames, recipe(564, int(g["cfg2_dqpsk_r12__meta"][0][0])))
        a2, c2 = build_stream(frames, recipe(564, int(g["cfg2_dqpsk_r12__meta"][0][0])))
        assert np.array_equal(a1, a2) and np.array_equal(c1, c2) and int(c1.sum()) == a1.size
        assert c1.size == want[f"cfg2_dqpsk_r12__{sc}__ready"].size


def test_bench_starts_its_own_ranks_and_a_failing_rank_fails_the_run():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (a child torch.distributed.run, before anything
    touches the GPU); here no GPU exists, so every rank refuses ("no CPU fallback") — and the run must report that as a
    failure with nothing on stdout, never as a result for fewer GPUs.  A launcher whose WORLD_SIZE differs from --gpus is
    refused outright."""
    import os, subprocess, sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""            # also on a GPU box this test sees no device
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-build"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0
    assert "starting 2 ranks" in p.stderr and "needs a GPU" in p.stderr, p.stderr[-2000:]
    assert not p.stdout.strip()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--no-build"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0 and "refusing" in p.stderr and not p.stdout.strip()


def test_device_build_flags():
    """The flags the parity and the measured rates depend on (csrc/Makefile): no contraction, no fast-math, correctly rounded
    division, and no SLP vectoriser (its packed pairs cost 4.6 % of the headline step: profiles/r04_ab_no_slp_vectorize.txt);
    the variant builds and the ISA tools use the same set."""
    from conftest import ROOT
    mk = (ROOT / "projectultra_amd" / "csrc" / "Makefile").read_text()
    flags = ("-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-fhip-fp32-correctly-rounded-divide-sqrt", "--offload-arch=$(ARCH)")
    for f in flags:
        assert f in mk, f
    for tool in ("build_variants.sh", "kernel_isa.sh", "kernel_resources.sh", "mix_fft_stalls.py", "ldpc_stalls.py", "issue_model.py"):
        text = (ROOT / "tools" / tool).read_text()
        assert "-fno-slp-vectorize" in text and "-ffp-contract=off" in text.replace('", "', " ").replace('"', ""), tool


def test_issue_model_instances_map_to_profile_classes():
    """tools/issue_model.py: a kernel INSTANCE as rocprofv3 names it -> the class of ultra_hip_profile_read_items and the mangled
    name of its ISA (the rotating transform is a class of its own; template arguments select the function)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("issue_model", ROOT / "tools" / "issue_model.py")
    im = importlib.util.module_from_spec(spec); spec.loader.exec_module(im)
    assert im.class_of("mix_fft2_kernel<10, true>") == ("mix_fft_rot_kernel", "frame-symbol")
    assert im.class_of("mix_fft2_kernel<10, false>") == ("mix_fft_kernel", "frame-symbol")
    assert im.class_of("ldpc_totals_kernel<3, 6, 1638ull, 3355443ull, false, 5>")[0] == "ldpc_decode_kernel"
    assert im.class_of("track_all_kernel<6>")[0] == "track_kernel" and im.class_of("track_pilot_kernel<16, true>")[0] == "track_pilot_kernel"
    assert im.class_of("stimulus_kernel<10>") == (None, None)
    m = "_ZN9ultra_hip3dev18ldpc_totals_kernelILi3ELi6ELy1638ELy3355443ELb0ELi5EEEvPKNS_9LdpcTPlanE"
    assert im._inst_match("ldpc_totals_kernel<3, 6, 1638ull, 3355443ull, false, 5>", m)
    assert not im._inst_match("ldpc_totals_kernel<3, 6, 1638ull, 3355443ull, true, 5>", m)
    assert not im._inst_match("ldpc_totals_kernel<8, 3, 591746662ull, 3277ull, false, 3>", m)


def test_bench_quotes_the_issue_model_only_for_the_tree_it_was_collected_on(tmp_path, monkeypatch):
    """bench.load_issue_model: profiles/issue.json is quoted only when its csrc_sha is the tree's and it holds the config."""
    import json
    import bench
    from projectultra_amd._lib import source_hash
    prof = tmp_path / "profiles"; prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    m, why = bench.load_issue_model("cfg3")
    assert m is None and "absent" in why
    (prof / "issue.json").write_text(json.dumps({"csrc_sha": "0000", "commit": "x", "configs": {"cfg3": {"classes": {}}}}))
    m, why = bench.load_issue_model("cfg3")
    assert m is None and "another kernel" in why
    (prof / "issue.json").write_text(json.dumps({"csrc_sha": source_hash(), "commit": "x", "configs": {"cfg4": {"classes": {}}}}))
    m, why = bench.load_issue_model("cfg3")
    assert m is None and "cfg4" in why
    m, why = bench.load_issue_model("cfg4")
    assert m is not None and why is None
