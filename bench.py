#!/usr/bin/env python3
"""bench.py — benchmarks of the batched OFDM-demodulate + LDPC-decode receive path, one line per BASELINE config.

    python bench.py [--config cfg3] --gpus N --steps K --warmup W
    N > 1 with no launcher around it: bench.py starts its own N ranks (a child `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N ... bench.py --gpus N ...`, before anything touches the GPU).  Under a launcher (RANK / WORLD_SIZE in the
    environment) it is one rank of it and refuses a WORLD_SIZE other than --gpus.  --backend gloo reduces the counters through
    host memory, so ranks may share a card (rehearsal of the N > 1 path on a one-GPU box).

--config (default cfg3 = BASELINE.json's headline metric and configs[2]):
  cfg3  OFDM-1024 16QAM R3/4, 59 carriers (15 pilots), Watterson "good" 30 dB, post-sync entry, ONE 2^20-frame batch per step
  cfg2  OFDM-512 DQPSK R1/2, 30 carriers, AWGN, 65,536 frames per GPU                       (configs[1])
  cfg4  LDPC R1/4 Es/N0 sweep -11..+30 dB, BPSK/AWGN LLRs, 2^17 codewords per point per GPU  (configs[3]: 2^20 on 8 GPUs)
  cfg5  {DBPSK,DQPSK,D8PSK,16QAM,32QAM} x {R1/4..R5/6} x 11 SNR points, 1,920 frames per point per GPU
                                                                                         (configs[4]: 2^22 on 8 GPUs)
  raw   cfg3 from raw audio: Schmidl-Cox acquisition + demodulation + decode (ultra_hip_receive_batch), 2^16 streams

A "step" = one pass of the configuration's hot path over this rank's batch of synthetic input, inputs already resident
in HBM (generated on the device before the timed region), ending in the configuration's collective: ONE all-reduce of
the eight uint64 counters per step (cfg2/cfg3/raw) or per sweep point (cfg4/cfg5).  Trials shard embarrassingly; per-GPU
work is fixed as N grows (weak scaling) for every config but cfg3 (below).

cfg3 (the headline) is STRONG scaling by default: ONE batch of 2^20 frames per step (north_star), rank r takes the
contiguous shard shard_range(2^20, r, N); --frames n switches to weak scaling with n frames per GPU (configs[2] is
--frames 262144 on one GPU).  The timed step delivers everything SURVEY.md 8(d) prices: decoded bytes, iteration counts,
status AND the frame's soft bits in a caller-owned buffer.

Prints ONE JSON line on rank 0 (see the driver contract) with the extra objects
  roofline      dominant kernel: algorithmic bytes per launch / mean launch duration (HIP events), HBM and LDS/VALU view
  cpu_baseline  the compiled reference (oracle/_ref, kind "reference") timed on the host's physical cores on a bounded
                sample of the same workload, with the oracle port beside it
  collective    backend, world size and mean latency of the counter all-reduce
"""
from __future__ import annotations

import argparse
import math
import fcntl
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
CLOCK_HZ = 2.4e9                # MI355X peak engine clock (MI355X_MICROARCH.md); the chip holds ~2.3 GHz under these kernels
# LDS-pipeline cycles per wave-instruction (MI355X_MICROARCH.md, LDS table; re-measured in profiles/r02_issue_table.txt)
LDS_CYC = dict(read_b32=2, write_b32=4, write_addtid_b32=2)
# Compute-side roofline: profiles/issue.json (tools/issue_model.py --json: dynamic VALU / SALU / LDS wave-instructions per
# launch of every kernel INSTANCE from rocprofv3 --pmc, the work items those launches covered, the measured cost per
# instruction of the instance's own opcode mix, and the hash of the kernel sources they were collected on).  Quoted only when
# that hash is the tree's; scaled by the work items this run's launches covered (ultra_hip_profile_read_items).
def load_issue_model(config):
    """(model, None) or (None, why) — the per-class issue cycles per work item of profiles/issue.json, if it describes THESE kernels."""
    f = ROOT / "profiles" / "issue.json"
    if not f.exists():
        return None, "profiles/issue.json absent (tools/issue_model.py --json after a PMC sweep of this build)"
    try:
        from projectultra_amd._lib import source_hash
        m = json.loads(f.read_text())
        if m.get("csrc_sha") != source_hash():
            return None, (f"profiles/issue.json was collected on kernel sources {m.get('csrc_sha')} (commit {m.get('commit')}), the tree holds "
                          f"{source_hash()}: instruction counts of another kernel are not quoted")
        if config not in m.get("configs", {}):
            return None, f"profiles/issue.json holds {sorted(m.get('configs', {}))}, this line is {config}"
        return m, None
    except Exception as e:
        return None, f"profiles/issue.json unreadable: {e}"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=("cfg2", "cfg3", "cfg4", "cfg5", "raw"), default="cfg3")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=0, help="trials per GPU per step (per sweep point for cfg4/cfg5): WEAK scaling; 0 = the config's size")
    ap.add_argument("--total-frames", type=int, default=0,
                    help="cfg2/cfg3: trials per step over ALL GPUs, sharded contiguously (STRONG scaling). cfg3 defaults to "
                         "2^20 — north_star's batch at 1, 2, 4 and 8 GPUs — unless --frames is given")
    ap.add_argument("--snr-db", type=float, default=None, help="cfg2/cfg3/raw: channel SNR (default 30 dB cfg3/raw, 3 dB cfg2)")
    ap.add_argument("--raw-channel", choices=("awgn", "watterson"), default="awgn",
                    help="raw: channel of the streams — AWGN (the decoder has nothing to do at 30 dB) or the headline's Watterson "
                         "good channel (0.5 ms / 0.1 Hz) over the transmission")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="trials for the CPU baseline (0 = auto)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="process-group backend of the counter all-reduce: nccl (= RCCL over xGMI, one GPU per rank) or gloo "
                         "(counters reduced through host memory; ranks may then SHARE a card, rank r on device r %% device_count: "
                         "the many-ranks-one-card rehearsal of tests/test_gpu_multirank.py)")
    ap.add_argument("--sync-collective", action="store_true",
                    help="issue the per-step all-reduce synchronously on the launch stream (default: asynchronously, one step deep)")
    ap.add_argument("--no-build", action="store_true",
                    help="do not (re)build the checker libraries; REQUIRED under rocprofv3, whose preloaded library "
                         "initialises the GPU before main() — the compiler must never be started from such a process")
    return ap.parse_args()


def ensure_built(no_build: bool):
    """Build the CHECKER libraries (oracle/libultra_oracle.so, and oracle/_ref when /root/reference exists) BEFORE
    anything touches the GPU, under a file lock so that the N ranks of a multi-GPU run do not race.  The product
    library is never built here: it must exist (there is no CPU fallback).  After this, oracle.bindings never
    starts the compiler again in this process (ULTRA_ORACLE_NO_BUILD)."""
    lib = ROOT / "projectultra_amd" / "libultra_hip.so"
    if not lib.exists():
        raise SystemExit(f"{lib} is missing — run `python -c 'import __graft_entry__ as g; g.build()'` first")
    src, so = ROOT / "oracle" / "ultra_oracle.c", ROOT / "oracle" / "libultra_oracle.so"
    stale = (not so.exists()) or so.stat().st_mtime < src.stat().st_mtime
    if no_build:
        if stale:
            raise SystemExit("oracle/libultra_oracle.so is missing or stale and --no-build was given — run "
                             "`python -c 'import __graft_entry__ as g; g.build()'` (or bench.py once without --no-build) first")
    else:
        with open(ROOT / "oracle" / ".build.lock", "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            subprocess.check_call(["make", "-s", "-C", str(ROOT / "oracle"), "libultra_oracle.so"], stdout=subprocess.DEVNULL)
            if Path("/root/reference/src").is_dir():
                subprocess.check_call(["make", "-s", "-C", str(ROOT / "oracle"), "_ref/libultra_ref.so"], stdout=subprocess.DEVNULL)
    os.environ["ULTRA_ORACLE_NO_BUILD"] = "1"


def host_cpu():
    """(model string, physical cores, logical CPUs) from /proc/cpuinfo."""
    model, cores, logical = "unknown", set(), 0
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "processor":
                logical += 1
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core)); phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    logical = logical or (os.cpu_count() or 1)
    n_phys = len(cores) or logical
    usable = n_phys
    try:
        usable = max(1, min(usable, len(os.sched_getaffinity(0))))      # never more workers than this process may use
    except AttributeError:
        pass
    # a container's CPU share (cgroup quota): more runnable threads than that only thrash (profiles/r02_cpu_scaling.txt:
    # on the GPU box, 16 CPUs' worth of quota on a 128-core host, both CPU legs peak at 16-32 threads and fall beyond)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                       # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())               # cgroup v1
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        usable = max(1, min(usable, int(quota + 0.5)))
    return model, usable, n_phys, logical, quota


def pick_threads(cores, run):
    """Worker threads for the CPU legs: run(n_threads) -> trials/s on a small sample, tried at the usable core count and
    at fractions of it.  A container may own far fewer CPUs than the host shows without exposing its quota (the GPU
    box: 128 cores visible, about 16 CPUs' worth of share — profiles/r02_cpu_scaling.txt), and more runnable threads
    than that only thrash; the fastest count is used for every leg and reported as `cores`."""
    cands = sorted({c for c in (cores, cores // 2, cores // 4, cores // 8, 32, 16) if 1 <= c <= cores}, reverse=True)
    rates = {c: run(c) for c in cands}
    return max(rates, key=rates.get), rates


def stratified_frames(n, frame_bytes, block=4096, seed=0x5712A7):
    """Indices of the frames the CPU legs check beside the timed prefix: first / middle / last `block`, `block` drawn
    uniformly over the batch, and block/2 around the frame whose audio crosses byte offset 2^32 of the buffer (32-bit
    offset arithmetic in a kernel would fail exactly there).  Sorted, unique."""
    block = min(block, n)
    rng = np.random.default_rng(seed)
    parts = [np.arange(0, block), np.arange((n - block) // 2, (n - block) // 2 + block), np.arange(n - block, n),
             rng.choice(n, size=block, replace=False)]
    strata = f"first, middle and last {block}, {block} drawn uniformly"
    edge = (1 << 32) // frame_bytes
    if edge + block // 4 < n:
        parts.append(np.arange(edge - block // 4, edge + block // 4))
        strata += f", {block // 2} around frame {edge} (byte offset 2^32 of the audio)"
    return np.unique(np.concatenate(parts)).astype(np.int64), strata


def best_of(fn, runs=3):
    best, out = None, None
    for _ in range(runs):
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best, out


# ------------------------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------------------------
class CounterRing:
    """Two counter blocks used by alternate steps, so that the all-reduce of step k (asynchronous on RCCL's own stream) runs
    under the kernels of step k + 1 instead of holding the launch stream until every rank has arrived.  A block is reused two
    steps later; next() first makes the launch stream wait for the collectives that were issued on it (long finished by then)."""

    def __init__(self, torch, shape, depth=2, device="cuda"):
        self.bufs = [torch.zeros(shape, dtype=torch.int64, device=device) for _ in range(depth)]
        self.works = [[] for _ in range(depth)]
        self.k = 0

    def next(self):
        self.k = (self.k + 1) % len(self.bufs)
        self.wait(self.k)
        return self.bufs[self.k]

    def note(self, work):
        if work is not None:
            self.works[self.k].append(work)

    def wait(self, k):
        for w in self.works[k]:
            w.wait()
        self.works[k] = []

    def drain(self):
        for k in range(len(self.bufs)):
            self.wait(k)


class ModemWorkload:
    """cfg2 / cfg3: post-sync demodulate + decode + count of n frames (ultra_hip_demod_decode_batch)."""

    def __init__(self, name, args, rank, world, torch):
        from projectultra_amd import CodeRate, ModemConfig, Modulation, ReceiveContext, presets
        from projectultra_amd.montecarlo import shard_range
        self.name, self.torch = name, torch
        if name == "cfg3":
            mc = presets.nvis_mode().with_mode(Modulation.QAM16, CodeRate.R3_4)
            mc.pilot_spacing = 4                           # tools/test_nvis_mode.cpp:208-212
            total = 0 if args.frames else (args.total_frames or (1 << 20))
            self.channel, self.snr_db = "watterson", 30.0 if args.snr_db is None else args.snr_db
            self.metric = "OFDM-1024 16QAM R3/4 frames decoded/sec"
            self.what = "OFDM 1024-FFT 16QAM R3/4, 59 carriers (15 pilots), Watterson good channel (0.5 ms / 0.1 Hz)"
            self.oracle_cfg = (1024, "QAM16", "R3_4")
        else:
            mc = ModemConfig().with_mode(Modulation.DQPSK, CodeRate.R1_2)      # tools/test_mode_snr.cpp:21-31
            total = 0 if args.frames else args.total_frames
            self.channel, self.snr_db = "awgn", 3.0 if args.snr_db is None else args.snr_db
            self.metric = "OFDM-512 DQPSK R1/2 frames decoded/sec"
            self.what = "OFDM 512-FFT DQPSK R1/2, 30 carriers (no pilots), AWGN"
            self.oracle_cfg = (512, "DQPSK", "R1_2")
        self.unit = "frames/s"
        self.ctx = ReceiveContext(mc)
        g = self.geo = self.ctx.geometry
        if total:                                                  # strong scaling: one batch, contiguous shards
            lo, hi = shard_range(total, rank, world)
            self.n, self.scaling, self.total_units = hi - lo, "strong", total
        else:                                                      # weak scaling: the same batch size on every GPU
            self.n = args.frames or (1 << 16)
            lo, _ = shard_range(self.n * world, rank, world)       # global frame ids of this rank
            self.scaling, self.total_units = "weak", self.n * world
        t0 = time.time()
        self.d_audio, self.d_payload = self.ctx.make_batch(self.n, seed=0x5EED, first_frame=lo, channel=self.channel,
                                                           snr_db=self.snr_db, delay_ms=0.5, doppler_hz=0.1)
        torch.cuda.synchronize()
        self.t_gen = time.time() - t0
        # everything SURVEY.md 8(d) counts as output is handed to the caller, the soft bits included
        self.out = dict(bytes=torch.empty((self.n, g.decoded_bytes), dtype=torch.uint8, device="cuda"),
                        iters=torch.empty(self.n, dtype=torch.int32, device="cuda"),
                        ok=torch.empty(self.n, dtype=torch.uint8, device="cuda"),
                        llr=torch.empty((self.n, g.llrs_per_frame), dtype=torch.float32, device="cuda"))
        self.ring = CounterRing(torch, 8)
        self.counters = self.ring.bufs[0]
        self.units_per_step = self.n
        self.points_per_step = 1
        # audio in + cfo in; 648 LLRs + decoded bytes + iters/status out
        self.bytes_audio = g.frame_samples * 4
        self.bytes_per_unit = self.bytes_audio + 4 + 648 * 4 + g.decoded_bytes + 4
        n_sym = self.ctx.cfg.n_data_symbols
        self.per_launch = {"cfo_walk_kernel": 0, "mix_fft_kernel": g.symbol_samples * 4, "track_pilot_kernel": 0,
                           "track_kernel": -(-648 * 4 // n_sym), "ldpc_decode_kernel": 648 * 4 + g.decoded_bytes + 4 + 1}
        self.per_step = {"mix_fft_kernel": self.n * n_sym * g.symbol_samples * 4, "track_kernel": self.n * 648 * 4}
        self.launch_units = self.n
        self.data = (f"synthetic ({self.n} distinct {self.channel} realisations per GPU generated on the device in HBM; "
                     f"random-payload codewords, {self.snr_db:g} dB)")
        if self.scaling == "strong":
            self.workload = (f"{self.what}, post-sync entry, LDPC min-sum <= 50 iterations, ONE batch of {total} frames per step "
                             f"sharded over {world} GPU(s) ({self.n} on rank {rank}); soft bits delivered to the caller")
        else:
            self.workload = (f"{self.what}, post-sync entry, LDPC min-sum <= 50 iterations, {self.n} frames per GPU per step; "
                             f"soft bits delivered to the caller")
        self.parallelism = f"frames sharded over {world} GPU(s), one counter all-reduce per step"

    def step(self, allreduce):
        self.counters = self.ring.next()
        self.counters.zero_()
        r = self.ctx.demod_decode(self.d_audio, out=self.out)
        self.ctx.count_errors(r, self.d_payload, self.counters)
        self.ring.note(allreduce(self.counters))

    def contexts(self):
        return [self.ctx]

    def cpu_baseline(self, cores, sample):
        from oracle.bindings import have_ref, make_config, oracle, Ref
        ccfg = make_config(*self.oracle_cfg)
        o = oracle()
        probe = self.d_audio[:min(self.n, 2048)].cpu().numpy()

        def rate(t):
            t0 = time.perf_counter()
            o.demod_decode_batch(ccfg, probe, n_threads=t, want_llr=False, want_state=False)
            return probe.shape[0] / (time.perf_counter() - t0)
        cores, self.thread_probe = pick_threads(cores, rate)
        sample = sample or min(self.n, 1024 * cores)        # about a second of the reference on every worker
        audio = self.d_audio[:sample].cpu().numpy()
        got = {k: v[:sample].cpu().numpy() for k, v in self.out.items()}
        res = {}
        t_port, want = best_of(lambda: o.demod_decode_batch(ccfg, audio, n_threads=cores, want_llr=False, want_state=False))
        res["port"] = dict(value=sample / t_port, unit="frames/s", cores=cores, kind="port", seconds=t_port,
                           gpu_matches_bitwise=bool(all(np.array_equal(got[k], want[k]) for k in ("bytes", "iters", "ok"))))
        ref = Ref() if have_ref() else None
        if ref is not None:
            t_ref, rr = best_of(lambda: ref.demod_decode_batch_mt(ccfg, audio, cores))
            res["reference"] = dict(value=sample / t_ref, unit="frames/s", cores=cores, kind="reference", seconds=t_ref,
                                    gpu_matches_bitwise=bool(all(np.array_equal(got[k], rr[k]) for k in ("bytes", "iters", "ok"))))
        # The WHOLE batch is what the step decoded, so the comparison is stratified over the whole batch (untimed): the first,
        # middle and last 4,096 frames, the 2,048 frames around the 4 GiB byte offset of the audio buffer where the batch
        # reaches it, and 4,096 frames drawn over all of it — soft bits bitwise against the oracle port, bytes / iterations /
        # status against the compiled reference as well.
        idx, strata = stratified_frames(self.n, self.bytes_audio)
        t_idx = self.torch.from_numpy(idx).cuda()
        a = self.d_audio[t_idx].cpu().numpy()
        g_s = {k: v[t_idx].cpu().numpy() for k, v in self.out.items()}
        w = o.demod_decode_batch(ccfg, a, n_threads=cores, want_llr=True, want_state=False)
        n_llr = min(g_s["llr"].shape[1], w["llr"].shape[1])
        ver = dict(frames=int(idx.size), strata=strata,
                   llr_bitwise_vs_port=bool(np.array_equal(g_s["llr"][:, :n_llr].view(np.uint32), w["llr"][:, :n_llr].view(np.uint32))),
                   decode_vs_port=bool(all(np.array_equal(g_s[k], w[k]) for k in ("bytes", "iters", "ok"))))
        if ref is not None:
            rr = ref.demod_decode_batch_mt(ccfg, a, cores)
            ver["decode_vs_reference"] = bool(all(np.array_equal(g_s[k], rr[k]) for k in ("bytes", "iters", "ok")))
        ver["all_equal"] = all(v for k, v in ver.items() if k not in ("frames", "strata"))
        for leg in res.values():
            leg["gpu_matches_bitwise"] = bool(leg["gpu_matches_bitwise"] and ver["all_equal"])
        self.verified = ver
        self.cores_used = cores
        return res, (f"timed: first {sample} frames of the same batch (copied from the device), post-sync demodulate + decode; "
                     f"compared: those and {idx.size} frames stratified over the whole batch ({strata})")


class RawWorkload(ModemWorkload):
    """raw: cfg3 end to end from raw audio — Schmidl-Cox acquisition (chunk-fed search, energy gate, coarse CFO, LTS
    refinement), SYNCED demodulation from each stream's own data start, decode (ultra_hip_receive_batch)."""

    def __init__(self, args, rank, world, torch):
        from projectultra_amd import CodeRate, Modulation, ReceiveContext, presets
        from projectultra_amd.montecarlo import shard_range
        self.name, self.torch = "raw", torch
        mc = presets.nvis_mode().with_mode(Modulation.QAM16, CodeRate.R3_4)
        mc.pilot_spacing = 4
        self.n = args.frames or (1 << 16)
        self.channel, self.snr_db = args.raw_channel, 30.0 if args.snr_db is None else args.snr_db
        self.metric = "OFDM-1024 16QAM R3/4 raw-audio streams received/sec (acquisition + demodulation + decode)"
        self.unit = "streams/s"
        self.oracle_cfg = (1024, "QAM16", "R3_4")
        self.ctx = ReceiveContext(mc)
        g = self.geo = self.ctx.geometry
        lo, _ = shard_range(self.n * world, rank, world)
        t0 = time.time()
        self.lead, self.tail = 1120, 960
        self.d_audio, self.d_payload = self.ctx.make_raw_batch(self.n, seed=0x5EED, first_frame=lo, channel=self.channel,
                                                               snr_db=self.snr_db, lead=self.lead, tail=self.tail)
        torch.cuda.synchronize()
        self.t_gen = time.time() - t0
        self.n_samples = self.d_audio.shape[1]
        self.out = None
        self.ring = CounterRing(torch, 8)
        self.counters = self.ring.bufs[0]
        self.units_per_step = self.n
        self.points_per_step = 1
        self.bytes_audio = self.n_samples * 4
        self.bytes_per_unit = self.bytes_audio + g.decoded_bytes + 4 + 4 + 4
        n_sym = self.ctx.cfg.n_data_symbols
        self.per_launch = {"acquire_kernel": self.n_samples * 4, "cfo_walk_kernel": 0, "mix_fft_kernel": g.symbol_samples * 4,
                           "track_pilot_kernel": 0, "track_kernel": -(-648 * 4 // n_sym),
                           "ldpc_decode_kernel": 648 * 4 + g.decoded_bytes + 4 + 1}
        self.launch_units = self.n
        self.data = (f"synthetic ({self.n} distinct raw streams per GPU generated on the device in HBM: {self.lead} samples of "
                     f"noise, preamble, 4 data symbols, {self.tail} samples of noise; "
                     f"{'AWGN' if self.channel == 'awgn' else 'Watterson good channel (0.5 ms / 0.1 Hz) over the transmission,'} {self.snr_db:g} dB)")
        self.workload = (f"OFDM 1024-FFT 16QAM R3/4 from raw audio ({self.n_samples} samples per stream, fed in 960-sample "
                         f"chunks): Schmidl-Cox acquisition + demodulation + LDPC decode, {self.n} streams per GPU per step")
        self.parallelism = f"streams sharded over {world} GPU(s), one counter all-reduce per step"

    def step(self, allreduce):
        self.counters = self.ring.next()
        self.counters.zero_()
        self.out = self.ctx.receive(self.d_audio, chunk=960)
        self.ctx.count_errors(self.out, self.d_payload, self.counters)
        self.ring.note(allreduce(self.counters))

    def cpu_baseline(self, cores, sample):
        """The whole receive of a raw stream on the host: the compiled reference's OFDMDemodulator::process fed 960 samples per
        call (search, sync, SYNCED demodulation) + LDPCDecoder::decodeSoft of the first 648 soft bits, one stream per worker
        thread at a time (kind "reference"); beside it the oracle port's acquisition alone (98 % of the reference's time on
        this path).  Best of three; every stream of the sample compared with the device's result."""
        import concurrent.futures as cf
        from oracle.bindings import have_ref, make_config, oracle, Ref
        ccfg = make_config(*self.oracle_cfg)
        cores = min(cores, 32)                   # see pick_threads: more runnable threads than the container's share thrash
        self.cores_used = cores
        sample = sample or min(self.n, 64 * cores)
        audio = self.d_audio[:sample].cpu().numpy()
        o = oracle()
        got = {k: v[:sample].cpu().numpy() for k, v in self.out.items()}

        def run_port():
            with cf.ThreadPoolExecutor(cores) as ex:
                return list(ex.map(lambda a: o.acquire(ccfg, a, chunk=960), audio))
        t, acq = best_of(run_port, runs=3)
        want = np.array([a["data_start"] if a["found"] else -1 for a in acq], np.int64)
        res = {"port": dict(value=sample / t, unit="streams/s", cores=cores, kind="port", seconds=t,
                            note="acquisition only (98 % of the reference's CPU time on this path); the post-sync part is the cfg3 line",
                            gpu_matches_bitwise=bool(np.array_equal(np.where(got["entry"] < 0, -1, got["entry"]), want)))}
        if have_ref():
            ref = Ref()
            t, rr = best_of(lambda: ref.receive_batch_mt(ccfg, audio, cores, chunk=960), runs=3)
            lost = got["entry"] < 0
            same = bool(np.array_equal(lost, rr["found"] == 0) and all(np.array_equal(got[k][~lost], rr[k][~lost]) for k in ("bytes", "iters", "ok")))
            res["reference"] = dict(value=sample / t, unit="streams/s", cores=cores, kind="reference", seconds=t,
                                    note="OFDMDemodulator::process in 960-sample chunks + LDPCDecoder::decodeSoft per stream",
                                    gpu_matches_bitwise=bool(same))
        return res, (f"first {sample} raw streams of the same batch (copied from the device): the compiled reference's whole receive, "
                     f"best of 3; the oracle port's acquisition beside it")


class LdpcSweepWorkload:
    """cfg4: the whole Es/N0 sweep of one LDPC code per step — 42 points x n codewords per GPU, LLRs of every point
    resident in HBM, one all-reduce per point."""

    def __init__(self, args, rank, world, torch):
        from projectultra_amd import CodeRate
        from projectultra_amd.montecarlo import shard_range
        from projectultra_amd.sweep import CFG4_SNR_POINTS, HipLdpcShard, point_seed
        self.name, self.torch = "cfg4", torch
        self.n = args.frames or (1 << 17)
        self.snrs = list(CFG4_SNR_POINTS)
        self.shard = HipLdpcShard(CodeRate.R1_4, batch=self.n)
        self.ctx = self.shard.ctx
        g = self.geo = self.ctx.geometry
        lo, _ = shard_range(self.n * world, rank, world)
        t0 = time.time()
        P = len(self.snrs)
        self.d_llr = torch.empty((P, self.n, 648), dtype=torch.float32, device="cuda")
        self.d_payload = torch.empty((P, self.n, g.ldpc_k // 8), dtype=torch.uint8, device="cuda")
        for i, snr in enumerate(self.snrs):
            self.ctx.make_llr_batch(self.n, snr, seed=point_seed(0x5EED, i), first_cw=lo, out=(self.d_llr[i], self.d_payload[i]))
        torch.cuda.synchronize()
        self.t_gen = time.time() - t0
        self.out = dict(bytes=torch.empty((P, self.n, g.decoded_bytes), dtype=torch.uint8, device="cuda"),
                        iters=torch.empty((P, self.n), dtype=torch.int32, device="cuda"),
                        ok=torch.empty((P, self.n), dtype=torch.uint8, device="cuda"))
        self.ring = CounterRing(torch, (P, 8))
        self.counters = self.ring.bufs[0]
        self.units_per_step = self.n * P
        self.points_per_step = P
        self.metric = "LDPC R1/4 codewords decoded/sec over the Es/N0 sweep -11..+30 dB"
        self.unit = "codewords/s"
        self.bytes_per_unit = 648 * 4 + g.decoded_bytes + 4                   # 2,617 B per codeword
        self.per_launch = {"ldpc_decode_kernel": 648 * 4 + g.decoded_bytes + 4 + 1}
        self.launch_units = self.n
        self.data = (f"synthetic ({P} Es/N0 points x {self.n} distinct random R1/4 codewords per GPU, BPSK over AWGN, LLR = 2y/sigma^2, "
                     f"generated on the device in HBM: {self.d_llr.numel() * 4 / 1e9:.1f} GB resident)")
        self.workload = (f"LDPC R1/4 (k=162, m=486) scaled min-sum <= 50 iterations, Es/N0 sweep -11..+30 dB in 1 dB steps, "
                         f"{self.n} codewords per point per GPU per step")
        self.parallelism = f"codewords of every point sharded over {world} GPU(s), one counter all-reduce per point"

    def step(self, allreduce):
        self.counters = self.ring.next()
        self.counters.zero_()
        for i in range(len(self.snrs)):
            r = self.ctx.ldpc_decode(self.d_llr[i], out={k: v[i] for k, v in self.out.items()})
            self.ctx.count_errors(r, self.d_payload[i], self.counters[i])
            self.ring.note(allreduce(self.counters[i]))

    def contexts(self):
        return [self.ctx]

    def curves(self):
        from projectultra_amd.montecarlo import counters_dict
        return {"LDPC R1_4": [dict(snr_db=s, **counters_dict(c)) for s, c in zip(self.snrs, self.counters.cpu())]}

    def cpu_baseline(self, cores, sample):
        from oracle.bindings import have_ref, oracle, Ref
        o = oracle()
        probe = self.d_llr[8:12, :256].reshape(-1, 648).cpu().numpy()         # around the waterfall

        def rate(t):
            t0 = time.perf_counter()
            o.ldpc_decode_batch_mt(0, probe, t)
            return probe.shape[0] / (time.perf_counter() - t0)
        cores, self.thread_probe = pick_threads(cores, rate)
        per = sample or 48 * cores                                          # codewords per point
        per = min(per, self.n)
        llr = self.d_llr[:, :per].reshape(-1, 648).cpu().numpy()
        got = {k: v[:, :per].reshape((-1,) + tuple(v.shape[2:])).cpu().numpy() for k, v in self.out.items()}
        res = {}
        t, w = best_of(lambda: o.ldpc_decode_batch_mt(0, llr, cores))
        res["port"] = dict(value=llr.shape[0] / t, unit="codewords/s", cores=cores, kind="port", seconds=t,
                           gpu_matches_bitwise=bool(np.array_equal(got["bytes"], w[0]) and np.array_equal(got["iters"], w[1])
                                                    and np.array_equal(got["ok"], w[2])))
        if have_ref():
            ref = Ref()
            t, w = best_of(lambda: ref.ldpc_decode_batch_mt(0, llr, cores))
            res["reference"] = dict(value=llr.shape[0] / t, unit="codewords/s", cores=cores, kind="reference", seconds=t,
                                    gpu_matches_bitwise=bool(np.array_equal(got["bytes"], w[0]) and np.array_equal(got["iters"], w[1])
                                                             and np.array_equal(got["ok"], w[2])))
        self.cores_used = cores
        return res, f"first {per} codewords of each of the {len(self.snrs)} points (same LLRs, copied from the device), LDPCDecoder::decodeSoft"


class ModeSweepWorkload:
    """cfg5: the 5 x 6 mode/rate grid x 11 SNR points per step, n frames per point per GPU, audio of every point resident
    in HBM.  The launches are shared ACROSS cells (projectultra_amd.sweep.HipModeGrid): one demodulation per modulation
    over the frames of its six rates and eleven points, one LDPC launch per code rate over the soft bits of all five
    modulations, one counting launch per rate; the [30 x 11][8] counter block is all-reduced once per step."""

    def __init__(self, args, rank, world, torch):
        from projectultra_amd.montecarlo import shard_range
        from projectultra_amd.sweep import CFG5_MODULATIONS, CFG5_RATES, CFG5_SNR_POINTS, HipModeGrid
        self.name, self.torch = "cfg5", torch
        self.n = args.frames or 1920
        self.snrs = list(CFG5_SNR_POINTS)
        self.cells = [(m, r) for m in CFG5_MODULATIONS for r in CFG5_RATES]
        S, n = len(self.snrs), self.n
        lo, _ = shard_range(n * world, rank, world)
        t0 = time.time()
        self.grid = HipModeGrid(CFG5_MODULATIONS, CFG5_RATES, self.snrs, frames_per_point=n)
        self.grid.generate(lo, seed=0x5EED)
        torch.cuda.synchronize()
        self.t_gen = time.time() - t0
        resident = self.grid.audio_bytes
        P = len(self.cells) * S
        self.ring = CounterRing(torch, (P, 8))
        self.counters = self.ring.bufs[0]
        self.units_per_step = n * P
        self.points_per_step = P
        self.collectives_per_step = 1
        self.collective_op = f"all_reduce(SUM) of {P} x 8 x int64 (every point of the grid)"
        self.metric = "adaptive-mode Monte-Carlo frames decoded/sec (5 modulations x 6 code rates x 11 SNR points)"
        self.unit = "frames/s"
        self.bytes_per_unit = resident / (n * P) + 4 + 648 * 4 + 60 + 4      # mean over the grid
        # per launch and frame: one symbol of audio (1120 samples) into mix_fft; the LLRs out of track_kernel and the
        # decoder's in/out differ from cell to cell (mean over the grid used)
        self.per_launch = {"mix_fft_kernel": 1120 * 4, "ldpc_decode_kernel": 648 * 4 + 50 + 4 + 1}
        self.per_step = {"mix_fft_kernel": sum(S * n * self.grid.ctx[c].cfg.n_data_symbols * 1120 * 4 for c in self.cells)}
        self.launch_units = n * S * len(CFG5_RATES)
        self.geo = self.grid.demod_ctx[0].geometry
        self.data = (f"synthetic ({P} points x {n} distinct frames per GPU generated on the device in HBM, AWGN; "
                     f"{resident / 1e9:.1f} GB of audio resident)")
        self.workload = (f"{{DBPSK,DQPSK,D8PSK,16QAM,32QAM}} x {{R1/4,R1/3,R1/2,R2/3,R3/4,R5/6}} on OFDM 1024-FFT / 59 carriers, SNR "
                         f"{self.snrs[0]:g}..{self.snrs[-1]:g} dB in 3 dB steps, post-sync entry, {n} frames per point per GPU per step; "
                         f"one demodulation per modulation, one LDPC launch per code rate")
        self.parallelism = (f"frames of every point sharded over {world} GPU(s), one all-reduce of the grid's counter block per step")

    def step(self, allreduce):
        c = self.grid.receive()                                   # [modulation][rate][point][8]
        self.counters = self.ring.next()
        self.counters.copy_(c.reshape(-1, 8))
        self.ring.note(allreduce(self.counters))

    def contexts(self):
        return self.grid.contexts()

    def curves(self):
        from projectultra_amd.montecarlo import counters_dict
        c = self.counters.cpu()
        S = len(self.snrs)
        return {f"{m.name} {r.name}": [dict(snr_db=s, **counters_dict(c[ci * S + si])) for si, s in enumerate(self.snrs)]
                for ci, (m, r) in enumerate(self.cells)}

    def cpu_baseline(self, cores, sample):
        from oracle.bindings import have_ref, make_config, oracle, Ref
        from projectultra_amd.sweep import CFG5_MODULATIONS, CFG5_RATES
        o = oracle()
        ref = Ref() if have_ref() else None
        ccfg0 = make_config(1024, "QAM16", "R3_4")
        S, n, R = len(self.snrs), self.n, len(CFG5_RATES)
        mi16, ri34 = [m.name for m in CFG5_MODULATIONS].index("QAM16"), [r.name for r in CFG5_RATES].index("R3_4")
        cell_audio = lambda mi, ri: self.grid.audio[mi][ri * S * n:(ri + 1) * S * n]          # [S * n][frame_samples]
        probe = cell_audio(mi16, ri34)[8 * n:8 * n + min(512, n)].cpu().numpy()             # the 16QAM R3/4 cell, 15 dB

        def rate(t):
            t0 = time.perf_counter()
            o.demod_decode_batch(ccfg0, probe, n_threads=t, want_llr=False, want_state=False)
            return probe.shape[0] / (time.perf_counter() - t0)
        cores, self.thread_probe = pick_threads(cores, rate)
        per = min(sample or max(8, 4 * cores), self.n)                      # frames per point
        t_port = t_ref = 0.0
        ok_port = ok_ref = True
        total = 0
        for ci, (m, r) in enumerate(self.cells):
            mi, ri = ci // R, ci % R
            ccfg = make_config(1024, m.name, r.name)
            a_cell = cell_audio(mi, ri)
            audio = np.concatenate([a_cell[si * n:si * n + per].cpu().numpy() for si in range(S)])
            got = self.grid.ctx[(m, r)].demod_decode(self.torch.from_numpy(audio).cuda())
            got = {k: v.cpu().numpy() for k, v in got.items()}
            t0 = time.perf_counter()
            w = o.demod_decode_batch(ccfg, audio, n_threads=cores, want_llr=False, want_state=False)
            t_port += time.perf_counter() - t0
            ok_port &= all(np.array_equal(got[k], w[k]) for k in ("bytes", "iters", "ok"))
            if ref is not None:
                t0 = time.perf_counter()
                w = ref.demod_decode_batch_mt(ccfg, audio, cores)
                t_ref += time.perf_counter() - t0
                ok_ref &= all(np.array_equal(got[k], w[k]) for k in ("bytes", "iters", "ok"))
            total += audio.shape[0]
        res = {"port": dict(value=total / t_port, unit="frames/s", cores=cores, kind="port", seconds=t_port, gpu_matches_bitwise=bool(ok_port))}
        if ref is not None:
            res["reference"] = dict(value=total / t_ref, unit="frames/s", cores=cores, kind="reference", seconds=t_ref,
                                    gpu_matches_bitwise=bool(ok_ref))
        self.cores_used = cores
        return res, f"first {per} frames of each of the {len(self.cells) * len(self.snrs)} points (same audio, copied from the device)"


def self_launch(args) -> int:
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves, as a CHILD
    `python -m torch.distributed.run --nproc-per-node N bench.py ...` (never an exec, and before anything in this process has
    touched the GPU).  The child's rank 0 prints the JSON line on the stdout it inherits; the exit code is the child's, so a
    rank that fails fails the run."""
    import socket
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    rc = 1
    for attempt in range(2):                        # the port is free when we look, not necessarily when the child binds it: one retry
        with socket.socket() as s:                  # a free rendezvous port on the loopback interface
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:] + ["--no-build"]
        print(f"bench.py: --gpus {args.gpus} and no RANK in the environment: starting {args.gpus} ranks ({' '.join(cmd[1:8])} ...)",
              file=sys.stderr, flush=True)
        t0 = time.time()
        rc = subprocess.call(cmd, env=env)
        if rc == 0 or time.time() - t0 > 20.0:      # a rendezvous that cannot bind fails within seconds; anything later is the run's own failure
            break
    return rc


# ------------------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    ensure_built(args.no_build)                     # before anything initialises the GPU
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    # stdout carries exactly ONE line (the JSON): whatever libraries print there meanwhile (RCCL's version banner at
    # communicator creation) goes to stderr instead
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:                          # never a silent fall-back to another number of GPUs
        raise SystemExit(f"bench.py: the launcher started WORLD_SIZE={world} ranks but --gpus {args.gpus} was asked for: refusing "
                         f"to report a line for another number of GPUs than the one named")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the receive path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and local_rank >= n_dev:
        raise SystemExit(f"bench.py: rank {rank} has no GPU of its own ({n_dev} visible, {world} ranks): RCCL needs one device per "
                         f"rank (--backend gloo lets ranks share a card for a rehearsal)")
    device_index = local_rank % n_dev
    torch.cuda.set_device(device_index)
    distributed = "RANK" in os.environ and "WORLD_SIZE" in os.environ      # launched by torch.distributed.run (also for N = 1)
    backend = None
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world)   # "nccl" IS RCCL on ROCm
        backend = dist.get_backend()
    via_host = distributed and args.backend == "gloo"    # gloo: the 64 bytes go through host memory

    from projectultra_amd.montecarlo import counters_dict

    if args.config in ("cfg2", "cfg3"):
        wl = ModemWorkload(args.config, args, rank, world, torch)
    elif args.config == "raw":
        wl = RawWorkload(args, rank, world, torch)
    elif args.config == "cfg4":
        wl = LdpcSweepWorkload(args, rank, world, torch)
    else:
        wl = ModeSweepWorkload(args, rank, world, torch)

    # the single collective of the path; timed with HIP events on the launch stream when profiling
    ar_events, ar_host = [], []

    def reduce_max(x: float) -> float:
        """max over the ranks of one host scalar (through the device for RCCL, through host memory for gloo)"""
        if not distributed:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if via_host else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_ranks(values):
        """every rank's list of host scalars, on every rank: [[rank 0's], [rank 1's], ...]"""
        if not distributed:
            return [list(values)]
        t = torch.tensor(list(values), dtype=torch.float64, device="cpu" if via_host else "cuda")
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [[float(v) for v in o.tolist()] for o in out]

    def allreduce(t):
        """SUM over the ranks, in place.  Returns the collective's handle where it was issued asynchronously (RCCL, outside the
        profiled pass): the caller's CounterRing waits for it before the block is used again, barrier() before the clock stops."""
        if not distributed:
            return None
        if via_host:
            t0 = time.perf_counter()
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)
            if allreduce.timed:
                ar_host.append(time.perf_counter() - t0)
            return None
        if ar_events is not None and allreduce.timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); dist.all_reduce(t, op=dist.ReduceOp.SUM); e1.record()
            ar_events.append((e0, e1))
            return None
        if args.sync_collective:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
    allreduce.timed = False

    def barrier():
        wl.ring.drain()             # every collective issued so far is ordered in front of what follows on the launch stream
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # Steady state before anything is counted: the first ten or so steps after set-up run up to 30 % slower than the rest
    # (clocks ramp up, workspaces are allocated and touched for the first time; tools/step_times.py shows the decay over
    # ~35 ms), which W warm-up steps do not cover when a step is 3 ms — a rank's share of the strong-scaling batch at N = 8.
    # ~0.15 s of untimed steps first, the same number on every rank (the step holds the collective).
    barrier()
    t_p = time.perf_counter()
    for _ in range(2):
        wl.step(allreduce)
    barrier()
    t_step = reduce_max((time.perf_counter() - t_p) / 2)
    n_prime = int(min(64, max(0, math.ceil(0.15 / max(t_step, 1e-4)) - 2)))
    for _ in range(n_prime):
        wl.step(allreduce)
    for _ in range(args.warmup):
        wl.step(allreduce)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step(allreduce)
    barrier()
    elapsed = time.perf_counter() - t0
    # The same K steps once more (all ranks: the step holds the collective) with, on rank 0, every kernel launch
    # bracketed by HIP events on the launch stream (ultra_hip_profile_enable) and every all-reduce by a pair of stream
    # events: per-kernel durations for the roofline object.  Kept out of the timed region: the event records cost ~2 %.
    prof, prof_items = {}, {}
    if rank == 0:
        for c in wl.contexts():
            c.profile_read(); c.profile_enable(True)
    allreduce.timed = True
    for _ in range(args.steps):
        wl.step(allreduce)
    barrier()
    allreduce.timed = False
    if rank == 0:
        for c in wl.contexts():
            c.profile_enable(False)
            for k, (ms, cnt, items) in c.profile_read_items().items():
                a = prof_items.setdefault(k, [0.0, 0, 0]); a[0] += ms; a[1] += cnt; a[2] += items
                # the HBM view keeps the nine classes: the rotating transform is part of mix_fft_kernel there
                a = prof.setdefault("mix_fft_kernel" if k == "mix_fft_rot_kernel" else k, [0.0, 0]); a[0] += ms; a[1] += cnt
    ar_us = float(np.mean([a.elapsed_time(b) for a, b in ar_events]) * 1e3) if ar_events else None
    if ar_host:
        ar_us = float(np.mean(ar_host) * 1e6)
    # Every rank's own clock over the timed region and the fall-back paths its contexts took (ultra_hip_get_status): a straggler,
    # or one rank on a slower chain, is visible in the line instead of hidden in the maximum.
    my_status = [c.status() for c in wl.contexts()]
    my_flags = 0
    for st in my_status:
        my_flags |= st["flags"]
    per_rank = gather_ranks([elapsed / args.steps * 1e3, float(my_flags)])
    elapsed = reduce_max(elapsed)
    wl.ring.drain()
    last = wl.counters.reshape(-1, 8).sum(dim=0).cpu()
    stats = counters_dict(last)
    expect = getattr(wl, "total_units", wl.units_per_step * world)
    assert stats["frames"] == expect, f"counted {stats['frames']} trials per step, expected {expect}"

    # ---- roofline of the dominant kernel (largest share of the step), rank 0 --------------------------------------
    roofline = None
    if rank == 0:
        kernels = {}
        for name, (ms_total, launches) in prof.items():
            if launches == 0:
                continue
            avg = ms_total / launches
            alg = wl.launch_units * wl.per_launch.get(name, 0)
            # kernels whose launch may cover one symbol or all symbols of a frame (zero-CFO layouts): bytes per step / launches
            if name in getattr(wl, "per_step", {}):
                alg = wl.per_step[name] / (launches / args.steps)
            kernels[name] = {"avg_launch_ms": avg, "launches_per_step": launches / args.steps, "ms_per_step": ms_total / args.steps,
                             "algorithmic_bytes_per_launch": alg, "GBps": alg / (avg * 1e-3) / 1e9,
                             "frac": alg / (avg * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
        props = torch.cuda.get_device_properties(device_index)
        roofline = {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": kernels[dom]["frac"], "traffic": None, "launch_ms": kernels[dom]["avg_launch_ms"],
                    "algorithmic_bytes_per_launch": kernels[dom]["algorithmic_bytes_per_launch"],
                    "path_GBps": wl.units_per_step * wl.bytes_per_unit * args.steps / elapsed / 1e9,
                    "kernels": kernels}
        # PMC-derived HBM bytes per launch, if collected for THIS build and batch size (tools/collect_profiles.sh
        # writes profiles/traffic.json with the commit and the batch size it was measured at)
        tf = ROOT / "profiles" / "traffic.json"
        if tf.exists():
            try:
                from projectultra_amd._lib import source_hash
                t = json.loads(tf.read_text())
                why = None
                if t.get("config", "cfg3") != args.config:
                    why = f"collected for {t.get('config')}, this line is {args.config}"
                elif t.get("n_frames") != wl.launch_units:
                    why = f"collected at {t.get('n_frames')} units per launch, this run has {wl.launch_units}"
                elif t.get("csrc_sha") != source_hash():
                    why = (f"collected on kernel sources {t.get('csrc_sha')} (commit {t.get('commit')}), the tree holds "
                           f"{source_hash()}: counters of another kernel are not quoted")
                if why is None:
                    roofline["traffic"] = t.get("kernels", {}).get(dom, {}).get("hbm_bytes_per_launch")
                    roofline["traffic_source"] = {"file": "profiles/traffic.json", "commit": t.get("commit"),
                                                  "csrc_sha": t.get("csrc_sha"), "collected": t.get("collected")}
                else:
                    roofline["traffic_dropped"] = why
            except Exception as e:
                roofline["traffic_dropped"] = f"profiles/traffic.json unreadable: {e}"
        # Compute-side view: which unit of the dominant kernel is busiest, and how busy.  Issue cycles per work item come from
        # profiles/issue.json (per kernel CLASS of ultra_hip_profile_read_items — the rotating transform apart from the one
        # without rotation), guarded by the hash of the kernel sources like the traffic; they are scaled by the work items
        # THIS run's launches covered and set against the cycles the measured launches had.
        roofline["work_items"] = {cls: {"launches_per_step": l / args.steps, "items_per_step": it / args.steps, "ms_per_step": ms / args.steps}
                                  for cls, (ms, l, it) in prof_items.items() if l}
        model, why = load_issue_model(args.config)
        if model is None:
            roofline["compute"] = None
            roofline["compute_dropped"] = why
        else:
            cu = props.multi_processor_count
            clock = float(model.get("clock_hz", CLOCK_HZ))
            per_item = model["configs"][args.config]["classes"]
            views = {}
            for cls, (ms_total, launches, items) in prof_items.items():
                if launches == 0 or items == 0 or cls not in per_item:
                    continue
                m = per_item[cls]
                t = ms_total * 1e-3 * clock                          # cycles the class's launches had, all steps
                u = dict(valu=m["valu_cycles_per_item"] * items / (4 * cu * t), salu=m["salu_cycles_per_item"] * items / (4 * cu * t),
                         lds=m["lds_cycles_per_item"] * items / (cu * t))
                views[cls] = u
                host = "mix_fft_kernel" if cls == "mix_fft_rot_kernel" else cls
                kernels[host].setdefault("issue", {})[cls] = dict(
                    item=m["item"], items_per_step=items / args.steps, launches_per_step=launches / args.steps, ms_per_step=ms_total / args.steps,
                    cycles_per_item={k: m[k + "_cycles_per_item"] for k in ("valu", "salu", "lds")}, busy_frac=u, clock_hz=clock,
                    instances=m.get("instances"), items_per_launch_at_collection=m.get("items_per_launch"))
            # the dominant CLASS of the HBM view may be two instances here (mix_fft_kernel): the busier one speaks for it
            cands = [c for c in views if c == dom or (dom == "mix_fft_kernel" and c == "mix_fft_rot_kernel")]
            if cands:
                best = max(cands, key=lambda c: prof_items[c][0])
                res = max(views[best], key=views[best].get)
                roofline.update({"bound": res, "compute": {"resource": {"valu": "vector issue (4 SIMDs per CU)", "salu": "scalar issue",
                                                                        "lds": "LDS pipeline"}[res], "class": best,
                                                           "frac": views[best][res], "all": views[best]},
                                 "hbm_frac": kernels[dom]["frac"]})
            roofline["compute_source"] = {"file": "profiles/issue.json", "commit": model.get("commit"), "csrc_sha": model.get("csrc_sha"),
                                          "collected": model.get("collected")}
        if dom == "ldpc_decode_kernel" and wl.name == "cfg4":
            # R1/4 on the totals kernel with its degree profile (round 4; ldpc_totals_kernel.h): per codeword-iteration the row
            # phase issues one gather and one lane-linear store per R plane (37: row profile 6 6 6 5 5 4 3 2), the variable
            # phase one gather per edge slot of its profile (13 + 12 + 12) and one lane-linear store per round (3); the
            # placement's residual gather collisions add 44 cycles (csrc/ldpc_placement_low.h).  Cycles per wave-instruction:
            # MI355X_MICROARCH.md's LDS table, re-measured in profiles/r02_issue_table.txt.
            planes, var_slots, var_rounds, extra = 37, 37, 3, 44
            cyc = (planes + var_slots) * LDS_CYC["read_b32"] + (planes + var_rounds) * LDS_CYC["write_addtid_b32"] + extra
            executed = stats["iters_sum"] / world + (stats["frames"] - stats["ldpc_fail"]) / world
            avail = kernels[dom]["ms_per_step"] * 1e-3 * CLOCK_HZ * props.multi_processor_count
            frac = cyc * executed / avail
            roofline.update({"bound": "lds", "lds": {"cycles_per_codeword_iteration": cyc, "codeword_iterations_per_step": executed,
                                                     "compute_units": props.multi_processor_count, "clock_hz": CLOCK_HZ,
                                                     "achieved_lds_cycles_per_s": cyc * executed / (kernels[dom]["ms_per_step"] * 1e-3),
                                                     "peak_lds_cycles_per_s": CLOCK_HZ * props.multi_processor_count,
                                                     "frac_of_lds_peak": frac},
                             "hbm_frac": kernels[dom]["frac"]})
        roofline["note"] = ("achieved/peak/frac = HBM view: algorithmic bytes per launch / mean launch duration (HIP events around every "
                            "launch of a repeat of the timed steps) against 8 TB/s. bound = the busiest unit of the dominant kernel where an "
                            "issue model exists (compute.frac: issue cycles per work item from PMC instruction counts x measured per-opcode costs, "
                            "profiles/issue.json — quoted only when its source hash is the tree's, else compute = null —, times the work items "
                            "this run's launches covered, over the cycles those launches had; kernels.*.issue for the others); 'lds' with the "
                            "instruction count of the totals decoder's profile for R1/4 (cfg4); 'hbm' otherwise")

    # ---- CPU baseline: the compiled reference on the host's physical cores, bounded sample (rank 0, N=1 only) ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        model, cores, physical, logical, quota = host_cpu()
        res, sample = wl.cpu_baseline(cores, args.cpu_sample)
        head = res.get("reference", res["port"])
        cores = getattr(wl, "cores_used", cores)
        cpu = dict(value=head["value"], unit=head["unit"], cores=cores, kind=head["kind"], sample=sample, cpu_model=model,
                   thread_probe_trials_per_s=getattr(wl, "thread_probe", None),
                   physical_cores=physical, logical_cpus=logical, cgroup_cpu_quota=quota,
                   threads="one worker thread per usable core: the physical cores, capped by the affinity mask and the container's "
                           "CPU quota where visible, then the fastest of {all, 1/2, 1/4, 1/8, 32, 16} on a probe sample "
                           "(thread_probe_trials_per_s); disjoint trial ranges, best of 3",
                   gpu_matches_cpu_bitwise=head["gpu_matches_bitwise"], verified=getattr(wl, "verified", None), legs=res)

    if rank == 0:
        from projectultra_amd._lib import STATUS_FLAGS
        rank_ms = [r[0] for r in per_rank]
        rank_flags = [int(r[1]) for r in per_rank]
        all_flags = 0
        for f in rank_flags:
            all_flags |= f
        path_status = {"default_path": all_flags == 0, "flags": all_flags,
                       "paths": [name for bit, name in sorted(STATUS_FLAGS.items()) if all_flags & bit],
                       "per_rank_flags": rank_flags,
                       "screen_rank0": [{k: st[k] for k in ("screen_launches", "screen_sample_n", "screen_sample_clean", "screen_gate",
                                                            "screen_gate_open", "screen_dirty")} for st in my_status],
                       "note": "ultra_hip_get_status of every context after the timed and the profiled pass: fall-back paths taken (sticky bits, "
                               "OR over contexts and ranks) and the decoder screen's last decision; a line with default_path false was "
                               "NOT measured on the kernels DESIGN.md describes as the default"}
        if roofline is not None:
            roofline["path_status"] = path_status
        if all_flags:
            wl.workload += " [NOT THE DEFAULT PATH: " + ", ".join(path_status["paths"]) + "]"
        total = getattr(wl, "total_units", wl.units_per_step * world) * args.steps
        value = total / elapsed
        line = {
            "metric": wl.metric, "value": value, "unit": wl.unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "untimed_priming_steps": n_prime + 2,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_ranks": {"min": min(rank_ms), "max": max(rank_ms), "all": rank_ms},     # each rank's own clock between the two barriers
            "higher_is_better": True, "scaling": getattr(wl, "scaling", "weak"), "vs_baseline": None,
            "dtype": "f32", "data": wl.data,
            "config": {"workload": wl.workload, "name": args.config, "trials_per_gpu_per_step": wl.units_per_step,
                       "trials_per_step_all_gpus": getattr(wl, "total_units", wl.units_per_step * world),
                       "launch_units": wl.launch_units,
                       "bytes_per_trial": wl.bytes_per_unit, "parallelism": wl.parallelism},
            "achieved_hbm_GBps": value * wl.bytes_per_unit / 1e9,
            "hbm_frac_of_peak": value * wl.bytes_per_unit / 1e9 / (HBM_PEAK_GBPS * world),
            "fer": stats["fer"], "ber": stats["ber"], "mean_bp_iterations": stats["mean_iters"], "trials_counted": stats["frames"],
            "counters": [int(v) for v in last.tolist()],           # the eight counters of the last step, summed over ranks (and points)
            "counters_per_point": wl.counters.reshape(-1, 8).cpu().tolist() if wl.points_per_step > 1 else None,
            "stimulus_seconds": wl.t_gen,
            "collective": {"backend": backend or "none (single process, no process group)", "world_size": dist.get_world_size() if distributed else 1,
                           "ranks_per_device": -(-world // n_dev) if distributed else 1,
                           "op": getattr(wl, "collective_op", "all_reduce(SUM) of 8 x int64"),
                           "per_step": getattr(wl, "collectives_per_step", wl.points_per_step), "allreduce_us": ar_us,
                           "issued": ("no process group" if not distributed else "through host memory, synchronous" if via_host else
                                      "synchronous on the launch stream" if args.sync_collective else
                                      "asynchronous (RCCL's stream), awaited when its counter block is reused two steps later and at the closing barrier")},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if hasattr(wl, "curves"):
            line["curves"] = wl.curves()
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
