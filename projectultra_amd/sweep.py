"""SNR / mode sweeps over the batched receive path, sharded over the GPUs of one node.

The reference's harnesses are an SNR loop around a trial loop (tools/test_mode_snr.cpp:18-109,126-160), with a
mode table around that (tools/test_nvis_mode.cpp:169-260; tests/regression_matrix.sh runs such a matrix and reports
success rates).  BASELINE.json configs[3] and [4] are those loops at Monte-Carlo scale:

  configs[3]  LDPC R1/4, 50-iteration min-sum, Es/N0 sweep -11 .. +30 dB, 2^20 codewords per point
  configs[4]  {DBPSK, DQPSK, D8PSK, 16QAM, 32QAM} x {R1/4, R1/3, R1/2, R2/3, R3/4, R5/6} BER/FER curves, 2^22 frames

Shape here: the trials of ONE point are a contiguous index range [0, n); rank r of W owns
shard_range(n, r, W), generates the stimulus of exactly those indices on its own GPU (counter-based generators:
the union over ranks is the same set of trials whatever W is), pushes them through the HIP path in batches,
accumulates the eight uint64 counters on the device, and the ranks meet in ONE all-reduce per point (64 bytes;
RCCL over xGMI with the "nccl" backend, gloo in the CPU tests) — per CURVE for the mode sweep, whose points share
their launches (HipModemShard.run_points) and reduce as one [points][8] block.  Nothing else crosses GPUs.

The per-shard work is behind a small interface (`run(lo, hi, snr_db, seed) -> int64[8]`): HipLdpcShard and
HipModemShard are the product; the CPU tests drive the same loops with a stub that counts on the host.
"""
from __future__ import annotations

import json
import time
from dataclasses import dataclass, field
from typing import Callable, Iterable, List, Optional, Sequence

from ._lib import COUNTER_NAMES
from .montecarlo import allreduce_counters, counters_dict, shard_range
from .types import CodeRate, Modulation, ModemConfig, is_differential, presets

# BASELINE.json configs[3]: one point per dB
CFG4_SNR_POINTS = tuple(float(s) for s in range(-11, 31))
# BASELINE.json configs[4]: the mode x rate grid ("DPSK" = the reference's DBPSK, types.hpp:27-39)
CFG5_MODULATIONS = (Modulation.DBPSK, Modulation.DQPSK, Modulation.D8PSK, Modulation.QAM16, Modulation.QAM32)
CFG5_RATES = (CodeRate.R1_4, CodeRate.R1_3, CodeRate.R1_2, CodeRate.R2_3, CodeRate.R3_4, CodeRate.R5_6)
# AWGN SNR over the whole audio band as the reference harness defines it (tools/test_nvis_mode.cpp:78-86): the 59
# carriers occupy ~6 % of it, so the waterfalls of the 30 cells lie between about -10 and +20 dB
CFG5_SNR_POINTS = tuple(float(s) for s in range(-9, 22, 3))


def point_seed(seed: int, point_index: int) -> int:
    """Every point of a sweep draws its own payloads and noise."""
    return (int(seed) ^ ((int(point_index) + 1) * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF


def nvis_cell_config(mod: Modulation, rate: CodeRate) -> ModemConfig:
    """ModemConfig of one mode-table cell exactly as tools/test_nvis_mode.cpp:195-212 builds it: presets::nvis_mode(),
    pilots (every 4th carrier) iff the modulation is coherent."""
    mc = presets.nvis_mode().with_mode(Modulation(mod), CodeRate(rate))
    mc.pilot_spacing = 4 if mc.use_pilots else 2
    return mc


# --------------------------------------------------------------------------------------------------------------
# shards on the GPU
# --------------------------------------------------------------------------------------------------------------
class HipLdpcShard:
    """LDPC-only trials (configs[3]): ultra_hip_make_llr_batch -> ultra_hip_ldpc_decode_batch ->
    ultra_hip_count_errors, `batch` codewords at a time, buffers allocated once."""

    def __init__(self, rate: CodeRate, max_iterations: int = 50, batch: int = 1 << 20, device: Optional[int] = None):
        import torch
        from .engine import ReceiveContext
        self.rate = CodeRate(rate)
        self.ctx = ReceiveContext(ModemConfig(code_rate=self.rate), max_iterations=max_iterations, device=device)
        self.batch = int(batch)
        self._torch = torch
        self._bufs = None

    def _buffers(self, n):
        torch, g, dev = self._torch, self.ctx.geometry, self.ctx.device
        if self._bufs is None or self._bufs["llr"].shape[0] < n:
            self._bufs = dict(llr=torch.empty((n, 648), dtype=torch.float32, device=dev),
                              payload=torch.empty((n, g.ldpc_k // 8), dtype=torch.uint8, device=dev),
                              bytes=torch.empty((n, g.decoded_bytes), dtype=torch.uint8, device=dev),
                              iters=torch.empty(n, dtype=torch.int32, device=dev),
                              ok=torch.empty(n, dtype=torch.uint8, device=dev))
        return self._bufs

    def run(self, lo: int, hi: int, snr_db: float, seed: int, keep=None):
        """Counters (device int64[8]) of codewords [lo, hi) at Es/N0 = snr_db.  keep: optional callback
        (first_index, llr, payload, result) per batch, for the tests' host recount."""
        torch = self._torch
        counters = torch.zeros(8, dtype=torch.int64, device=self.ctx.device)
        b = self._buffers(min(self.batch, max(hi - lo, 1)))
        for c0 in range(lo, hi, self.batch):
            n = min(self.batch, hi - c0)
            llr, payload = b["llr"][:n], b["payload"][:n]
            self.ctx.make_llr_batch(n, snr_db, seed=seed, first_cw=c0, out=(llr, payload))
            r = self.ctx.ldpc_decode(llr, out=dict(bytes=b["bytes"][:n], iters=b["iters"][:n], ok=b["ok"][:n]))
            self.ctx.count_errors(r, payload, counters)
            if keep is not None:
                keep(c0, llr, payload, r)
        return counters


class HipModemShard:
    """Whole-path trials (configs[4], configs[1], configs[2]): ultra_hip_make_batch (payload -> encode -> preamble +
    modulate -> 0.5 peak -> channel) -> ultra_hip_demod_decode_batch -> ultra_hip_count_errors."""

    def __init__(self, config: ModemConfig, channel: str = "awgn", delay_ms: float = 0.5, doppler_hz: float = 0.1,
                 max_iterations: int = 50, batch: int = 1 << 16, device: Optional[int] = None):
        import torch
        from .engine import ReceiveContext
        self.config, self.channel, self.delay_ms, self.doppler_hz = config, channel, float(delay_ms), float(doppler_hz)
        self.ctx = ReceiveContext(config, max_iterations=max_iterations, device=device)
        self.batch = int(batch)
        self._torch = torch
        self._out = None
        self._pts = None

    def run(self, lo: int, hi: int, snr_db: float, seed: int, keep=None):
        torch = self._torch
        g = self.ctx.geometry
        counters = torch.zeros(8, dtype=torch.int64, device=self.ctx.device)
        nb = min(self.batch, max(hi - lo, 1))
        if self._out is None or self._out["iters"].shape[0] < nb:
            dev = self.ctx.device
            self._out = dict(bytes=torch.empty((nb, g.decoded_bytes), dtype=torch.uint8, device=dev),
                             iters=torch.empty(nb, dtype=torch.int32, device=dev),
                             ok=torch.empty(nb, dtype=torch.uint8, device=dev))
        for f0 in range(lo, hi, self.batch):
            n = min(self.batch, hi - f0)
            audio, payload = self.ctx.make_batch(n, seed=seed, first_frame=f0, channel=self.channel, snr_db=snr_db,
                                                 delay_ms=self.delay_ms, doppler_hz=self.doppler_hz)
            out = {k: v[:n] for k, v in self._out.items()}
            r = self.ctx.demod_decode(audio, out=out, want_llr=keep is not None)
            self.ctx.count_errors(r, payload, counters)
            if keep is not None:
                keep(f0, audio, payload, r)
        return counters

    def run_points(self, lo: int, hi: int, snr_points: Sequence[float], seeds: Sequence[int]):
        """Frames [lo, hi) of SEVERAL points of this cell as one batch per demodulate + decode (the receive path does
        not depend on the SNR — only the stimulus does — and frames are independent, so the points of a curve share
        launches: with a few thousand frames per point the path is otherwise bound by launch latency, ~30 launches
        per point).  Per-point stimulus and per-point counters; returns device int64 [len(snr_points)][8], row i equal
        to run(lo, hi, snr_points[i], seeds[i])."""
        torch = self._torch
        g, dev = self.ctx.geometry, self.ctx.device
        n, P = hi - lo, len(snr_points)
        counters = torch.zeros((P, 8), dtype=torch.int64, device=dev)
        if n <= 0 or P == 0:
            return counters
        if n > self.batch:                                         # a point larger than a batch: the per-point loop
            for i in range(P):
                counters[i] = self.run(lo, hi, float(snr_points[i]), int(seeds[i]))
            return counters
        per = max(1, self.batch // n)                              # points per batch
        rows = min(per, P) * n
        if self._pts is None or self._pts["audio"].shape[0] < rows:
            self._pts = dict(audio=torch.empty((rows, g.frame_samples), dtype=torch.float32, device=dev),
                             payload=torch.empty((rows, g.ldpc_k // 8), dtype=torch.uint8, device=dev),
                             bytes=torch.empty((rows, g.decoded_bytes), dtype=torch.uint8, device=dev),
                             iters=torch.empty(rows, dtype=torch.int32, device=dev),
                             ok=torch.empty(rows, dtype=torch.uint8, device=dev))
        b = self._pts
        for p0 in range(0, P, per):
            k = min(per, P - p0)
            for i in range(k):
                self.ctx.make_batch(n, seed=int(seeds[p0 + i]), first_frame=lo, channel=self.channel, snr_db=float(snr_points[p0 + i]),
                                    delay_ms=self.delay_ms, doppler_hz=self.doppler_hz,
                                    out=(b["audio"][i * n:(i + 1) * n], b["payload"][i * n:(i + 1) * n]))
            r = self.ctx.demod_decode(b["audio"][:k * n], out={q: b[q][:k * n] for q in ("bytes", "iters", "ok")})
            self.ctx.count_errors_points(r, b["payload"][:k * n], counters[p0:p0 + k])
        return counters


class HipModeGrid:
    """The whole mode x rate grid of configs[4] with SHARED launches across cells.  Demodulation does not depend on the
    code rate and decoding does not depend on the modulation (every cell: 648-bit codewords; the data symbols of a frame
    depend on the modulation only), so a pass over the grid is
        one demodulation per MODULATION over the frames of all its rates and SNR points   (5 launch chains, not 30)
        one LDPC launch per CODE RATE over the soft bits of all modulations                (6 launches, not 30)
        one counting launch per code rate, counters per (modulation, SNR point)
    through ONE soft-bit array [modulation][rate][point * frame][768] (ultra_hip_demod_batch_strided writes its rows,
    ultra_hip_ldpc_decode_blocks reads one rate's runs out of it).  Same counters as HipModemShard.run_points cell by
    cell (tests/test_gpu_sweep.py); tools/test_mode_snr.cpp:126-160 is the loop being batched."""
    LLR_STRIDE = 768

    def __init__(self, mods=CFG5_MODULATIONS, rates=CFG5_RATES, snr_points=CFG5_SNR_POINTS, frames_per_point: int = 1920,
                 channel: str = "awgn", delay_ms: float = 0.5, doppler_hz: float = 0.1, max_iterations: int = 50,
                 device: Optional[int] = None):
        import torch
        from .engine import ReceiveContext
        self._torch = torch
        self.mods, self.rates, self.snrs = [Modulation(m) for m in mods], [CodeRate(r) for r in rates], [float(x) for x in snr_points]
        self.n, self.channel, self.delay_ms, self.doppler_hz = int(frames_per_point), channel, float(delay_ms), float(doppler_hz)
        M, R, S, n = len(self.mods), len(self.rates), len(self.snrs), self.n
        # one context per cell for the stimulus (payload -> encoder of the rate -> modulator of the modulation); the
        # receive side uses one of them per modulation (demodulation) and one per rate (decoding, counting)
        self.ctx = {(m, r): ReceiveContext(nvis_cell_config(m, r), max_iterations=max_iterations, device=device)
                    for m in self.mods for r in self.rates}
        dev = next(iter(self.ctx.values())).device
        self.demod_ctx = [self.ctx[(m, self.rates[0])] for m in self.mods]
        self.ldpc_ctx = [self.ctx[(self.mods[0], r)] for r in self.rates]
        if any(c.geometry.llrs_per_frame > self.LLR_STRIDE or c.geometry.llrs_per_frame < 648 for c in self.demod_ctx):
            raise ValueError("HipModeGrid: a frame must carry between 648 and 768 soft bits")
        rows = S * n
        self.audio = [torch.empty((R * rows, c.geometry.frame_samples), dtype=torch.float32, device=dev) for c in self.demod_ctx]
        self.llr = torch.empty((M * R * rows, self.LLR_STRIDE), dtype=torch.float32, device=dev)
        self.payload = [torch.empty((M * rows, c.geometry.ldpc_k // 8), dtype=torch.uint8, device=dev) for c in self.ldpc_ctx]
        self.out = [dict(bytes=torch.empty((M * rows, c.geometry.decoded_bytes), dtype=torch.uint8, device=dev),
                         iters=torch.empty(M * rows, dtype=torch.int32, device=dev),
                         ok=torch.empty(M * rows, dtype=torch.uint8, device=dev)) for c in self.ldpc_ctx]
        self.counters = torch.zeros((R, M * S, 8), dtype=torch.int64, device=dev)
        self.audio_bytes = sum(a.numel() * 4 for a in self.audio)

    def generate(self, lo: int, seed: int = 0x5EED):
        """Stimulus of frames [lo, lo + n) of every point of every cell; the point index of (cell ci, SNR si) is
        ci * len(snrs) + si with ci = modulation-major, as in mode_sweep."""
        M, R, S, n = len(self.mods), len(self.rates), len(self.snrs), self.n
        for mi, m in enumerate(self.mods):
            for ri, r in enumerate(self.rates):
                c = self.ctx[(m, r)]
                for si, snr in enumerate(self.snrs):
                    a0, p0 = (ri * S + si) * n, (mi * S + si) * n
                    c.make_batch(n, seed=point_seed(seed, (mi * R + ri) * S + si), first_frame=lo, channel=self.channel, snr_db=snr,
                                 delay_ms=self.delay_ms, doppler_hz=self.doppler_hz,
                                 out=(self.audio[mi][a0:a0 + n], self.payload[ri][p0:p0 + n]))

    def receive(self):
        """One pass over the grid -> device int64 counters [modulation][rate][point][8]."""
        M, R, S, n = len(self.mods), len(self.rates), len(self.snrs), self.n
        rows = S * n
        self.counters.zero_()
        for mi, c in enumerate(self.demod_ctx):
            c.demod_into(self.audio[mi], self.llr[mi * R * rows:(mi + 1) * R * rows])
        for ri, c in enumerate(self.ldpc_ctx):
            r = c.ldpc_decode_blocks(self.llr[ri * rows:], rows, R * rows, M, out=self.out[ri])
            c.count_errors_points(r, self.payload[ri], self.counters[ri])
        return self.counters.reshape(R, M, S, 8).permute(1, 0, 2, 3)

    def contexts(self):
        return list(self.ctx.values())


# --------------------------------------------------------------------------------------------------------------
# the loops
# --------------------------------------------------------------------------------------------------------------
@dataclass
class SweepPoint:
    label: str
    snr_db: float
    trials: int
    seed: int
    counters: dict = field(default_factory=dict)
    seconds: float = 0.0

    def as_dict(self) -> dict:
        d = dict(label=self.label, snr_db=self.snr_db, trials=self.trials, seed=self.seed, seconds=self.seconds)
        d.update(self.counters)
        return d


def run_point(shard, n_trials: int, snr_db: float, seed: int, rank: int = 0, world: int = 1, group=None, keep=None):
    """One point: this rank's shard of the n_trials, then the single all-reduce.  Returns the global counters as a
    CPU int64[8] tensor (identical on every rank)."""
    lo, hi = shard_range(n_trials, rank, world)
    counters = shard.run(lo, hi, float(snr_db), int(seed), keep) if keep is not None else shard.run(lo, hi, float(snr_db), int(seed))
    counters = allreduce_counters(counters, group=group)
    return counters.cpu()


def sweep(label: str, shard, snr_points: Sequence[float], n_trials: int, seed: int = 0x5EED, rank: int = 0,
          world: int = 1, group=None, first_point_index: int = 0, on_point: Optional[Callable] = None) -> List[SweepPoint]:
    """The SNR loop around the trial loop (tools/test_mode_snr.cpp:126-160) for one code / mode."""
    out = []
    seeds = [point_seed(seed, first_point_index + i) for i in range(len(snr_points))]
    if hasattr(shard, "run_points"):
        # the curve's points share launches (HipModemShard.run_points); ONE all-reduce of the [points][8] block
        lo, hi = shard_range(n_trials, rank, world)
        t0 = time.perf_counter()
        block = allreduce_counters(shard.run_points(lo, hi, [float(x) for x in snr_points], seeds), group=group).cpu()
        dt = (time.perf_counter() - t0) / max(len(snr_points), 1)
        per_point = [(block[i], dt) for i in range(len(snr_points))]
    else:
        per_point = None
    for i, snr in enumerate(snr_points):
        ps = seeds[i]
        if per_point is not None:
            c, dt = per_point[i]
        else:
            t0 = time.perf_counter()
            c = run_point(shard, n_trials, snr, ps, rank, world, group)
            dt = time.perf_counter() - t0
        p = SweepPoint(label=label, snr_db=float(snr), trials=int(n_trials), seed=ps, counters=counters_dict(c), seconds=dt)
        if p.counters["frames"] != n_trials:
            raise RuntimeError(f"sweep point {label} @ {snr} dB counted {p.counters['frames']} of {n_trials} trials")
        out.append(p)
        if on_point is not None:
            on_point(p)
    return out


def ldpc_snr_sweep(rate: CodeRate = CodeRate.R1_4, snr_points: Iterable[float] = CFG4_SNR_POINTS, n_codewords: int = 1 << 20,
                   seed: int = 0x5EED, rank: int = 0, world: int = 1, group=None, max_iterations: int = 50,
                   batch: int = 1 << 20, shard=None, on_point=None) -> List[SweepPoint]:
    """BASELINE configs[3]: one LDPC code, BPSK over AWGN, BER/FER/iterations per Es/N0 point."""
    shard = shard if shard is not None else HipLdpcShard(rate, max_iterations=max_iterations, batch=batch)
    return sweep(f"LDPC {CodeRate(rate).name}", shard, list(snr_points), n_codewords, seed, rank, world, group, 0, on_point)


def mode_sweep(cells=None, snr_points: Iterable[float] = CFG5_SNR_POINTS, frames_per_point: int = 1 << 14,
               channel: str = "awgn", delay_ms: float = 0.5, doppler_hz: float = 0.1, seed: int = 0x5EED, rank: int = 0,
               world: int = 1, group=None, batch: int = 1 << 16, shard_factory=None, on_point=None) -> List[SweepPoint]:
    """BASELINE configs[4]: the mode table (tools/test_nvis_mode.cpp:169-260) x an SNR axis.  One receive context per
    cell; cells run one after another, each point sharded over all ranks."""
    cells = list(cells) if cells is not None else [(m, r) for m in CFG5_MODULATIONS for r in CFG5_RATES]
    snr_points = list(snr_points)
    factory = shard_factory or (lambda mc: HipModemShard(mc, channel=channel, delay_ms=delay_ms, doppler_hz=doppler_hz, batch=batch))
    out = []
    for ci, (mod, rate) in enumerate(cells):
        shard = factory(nvis_cell_config(mod, rate))
        out += sweep(f"{Modulation(mod).name} {CodeRate(rate).name}", shard, snr_points, frames_per_point, seed, rank, world,
                     group, ci * len(snr_points), on_point)
        del shard
    return out


def mode_sweep_grid(snr_points: Iterable[float] = CFG5_SNR_POINTS, frames_per_point: int = 1 << 14, channel: str = "awgn",
                    delay_ms: float = 0.5, doppler_hz: float = 0.1, seed: int = 0x5EED, rank: int = 0, world: int = 1, group=None,
                    grid_frames: int = 1920, on_point=None) -> List[SweepPoint]:
    """BASELINE configs[4] with launches shared across cells (HipModeGrid): this rank's shard of every point goes
    through the grid grid_frames frames at a time, the counters of all 30 x len(snr_points) points accumulate on the
    device and meet in ONE all-reduce.  Same counters as mode_sweep (same point seeds, same trials)."""
    import torch
    snr_points = [float(x) for x in snr_points]
    lo, hi = shard_range(frames_per_point, rank, world)
    t0 = time.perf_counter()
    total = None
    grid = None
    f0 = lo
    while f0 < hi or total is None:
        n = min(grid_frames, max(hi - f0, 0))
        if n == 0:                                                  # an empty shard still takes part in the all-reduce
            M, R, S = len(CFG5_MODULATIONS), len(CFG5_RATES), len(snr_points)
            total = torch.zeros((M, R, S, 8), dtype=torch.int64, device="cuda")
            break
        if grid is None or grid.n != n:
            grid = None                                             # the ragged tail: free the 30 contexts and their buffers first
            grid = HipModeGrid(CFG5_MODULATIONS, CFG5_RATES, snr_points, frames_per_point=n, channel=channel, delay_ms=delay_ms,
                               doppler_hz=doppler_hz)
        grid.generate(f0, seed=seed)
        c = grid.receive().contiguous()
        total = c.clone() if total is None else total + c
        f0 += n
    block = allreduce_counters(total.reshape(-1, 8), group=group).cpu()
    dt = (time.perf_counter() - t0) / max(block.shape[0], 1)
    out = []
    S = len(snr_points)
    for ci, (mod, rate) in enumerate((m, r) for m in CFG5_MODULATIONS for r in CFG5_RATES):
        for si, snr in enumerate(snr_points):
            p = SweepPoint(label=f"{Modulation(mod).name} {CodeRate(rate).name}", snr_db=snr, trials=int(frames_per_point),
                           seed=point_seed(seed, ci * S + si), counters=counters_dict(block[ci * S + si]), seconds=dt)
            if p.counters["frames"] != frames_per_point:
                raise RuntimeError(f"grid point {p.label} @ {snr} dB counted {p.counters['frames']} of {frames_per_point} trials")
            out.append(p)
            if on_point is not None:
                on_point(p)
    return out


def curves_document(kind: str, points: List[SweepPoint], **meta) -> dict:
    """JSON-serialisable curves: per label, the points in SNR order with BER, FER, undetected-error rate and mean
    BP iterations next to the raw counters."""
    curves = {}
    for p in points:
        curves.setdefault(p.label, []).append(p.as_dict())
    return dict(kind=kind, counters=list(COUNTER_NAMES), total_trials=sum(p.trials for p in points),
                total_seconds=sum(p.seconds for p in points), **meta, curves=curves)


def write_curves(path, kind: str, points: List[SweepPoint], **meta) -> dict:
    doc = curves_document(kind, points, **meta)
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
    return doc
