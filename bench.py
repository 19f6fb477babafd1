#!/usr/bin/env python3
"""bench.py — headline benchmark of the batched OFDM-demodulate + LDPC-decode receive path.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Metric (BASELINE.json): OFDM-1024 16QAM R3/4 frames decoded/sec; achieved HBM GB/s.
A "step" = one pass of the fused hot path (demodulate 4 data symbols -> first 648 LLRs ->
LDPC scaled-min-sum <= 50 iterations -> error counters -> ONE all-reduce of the 8 counters)
over this rank's batch of frames, inputs already resident in HBM.  Frames shard
embarrassingly: every rank owns `--frames` frames (weak scaling), nothing crosses GPUs on
the data path; the only collective is the 64-byte counter all-reduce (RCCL over xGMI).

Prints ONE JSON line on rank 0 (see the driver contract) with two extra objects:
  roofline     dominant kernel: algorithmic bytes per launch / measured HIP-event time
  cpu_baseline the oracle ("port") timed on the host cores on a bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# algorithmic bytes per frame for the fused path, cfg3 geometry
#   4*1120*4 audio + 4 cfo in; 648*4 LLR + 61 decoded bytes + 4 iters/status out
BYTES_AUDIO = 4 * 1120 * 4
BYTES_PER_FRAME_FUSED = BYTES_AUDIO + 4 + 648 * 4 + 61 + 4          # 20,581
BYTES_PER_FRAME_DEMOD = BYTES_AUDIO + 4 + 704 * 4                   # demod kernel alone: 704 LLRs written
BYTES_PER_CW_LDPC = 648 * 4 + 61 + 4 + 1                            # LDPC kernel alone


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=1 << 18, help="frames per GPU per step")
    ap.add_argument("--unique", type=int, default=8192, help="distinct synthetic frames generated on the host")
    ap.add_argument("--snr-db", type=float, default=30.0)
    ap.add_argument("--stimulus", choices=("host", "device"), default="device",
                    help="device (default): every frame distinct, generated on the GPU (ultra_hip_make_batch, scope row f2: "
                         "payload / encoder / modulator bit-identical to the oracle's generator, channel statistically "
                         "equivalent); host: --unique frames from the oracle's TX chain, tiled in HBM")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="frames for the CPU baseline (0 = auto)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}", file=sys.stderr)
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the receive path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)   # "nccl" IS RCCL on ROCm

    # the checker library is (re)built by one rank only: concurrent `make`s of N ranks would race on the .so
    from oracle.bindings import build_oracle
    if local_rank == 0:
        build_oracle()
    if world > 1:
        dist.barrier()
    from oracle.bindings import geometry, have_ref, make_config, oracle, Ref
    from projectultra_amd import CodeRate, Modulation, ReceiveContext, presets
    from projectultra_amd.montecarlo import allreduce_counters, counters_dict, shard_range

    # ---- workload: BASELINE.json configs[2] geometry -----------------------------------------
    mc = presets.nvis_mode().with_mode(Modulation.QAM16, CodeRate.R3_4)
    mc.pilot_spacing = 4                           # tools/test_nvis_mode.cpp:208-212
    ccfg = make_config(1024, "QAM16", "R3_4")      # same config as a POD for the oracle
    geo = geometry(ccfg)
    assert geo.frame_samples * 4 == BYTES_AUDIO and geo.llrs_per_frame == 704 and geo.decoded_bytes == 61
    n_frames = args.frames
    unique = min(args.unique, n_frames)
    reps = -(-n_frames // unique)
    glob_lo, _ = shard_range(n_frames * world, rank, world)    # global frame ids of this rank

    # ---- synthetic stimulus: TX chain + Watterson "good" (0.5 ms / 0.1 Hz) ----------------------
    o = oracle()
    ctx = ReceiveContext(mc)
    t0 = time.time()
    if args.stimulus == "device":
        # every frame of the rank distinct, generated in HBM (payload/encoder/modulator bit-identical to the
        # oracle's generator, channel statistically equivalent: tests/test_gpu_stimulus.py)
        d_audio, d_payload = ctx.make_batch(n_frames, seed=0x5EED, first_frame=glob_lo, channel="watterson",
                                            snr_db=args.snr_db, delay_ms=0.5, doppler_hz=0.1)
        torch.cuda.synchronize()
        unique = n_frames
        audio_u = None
    else:
        audio_u, payload_u = o.make_batch(ccfg, unique, seed=0x5EED, f0=glob_lo, channel="watterson",
                                          snr_db=args.snr_db, delay_ms=0.5, doppler_hz=0.1)
        d_audio_u = torch.from_numpy(audio_u).cuda()
        d_audio = d_audio_u.repeat(reps, 1)[:n_frames].contiguous()        # [n_frames][4480] f32, resident in HBM
        d_payload = torch.from_numpy(payload_u).cuda().repeat(reps, 1)[:n_frames].contiguous()
        del d_audio_u
    t_gen = time.time() - t0
    out = dict(bytes=torch.empty((n_frames, geo.decoded_bytes), dtype=torch.uint8, device="cuda"),
               iters=torch.empty(n_frames, dtype=torch.int32, device="cuda"),
               ok=torch.empty(n_frames, dtype=torch.uint8, device="cuda"))
    counters = torch.zeros(8, dtype=torch.int64, device="cuda")

    def step():
        counters.zero_()
        r = ctx.demod_decode(d_audio, out=out)
        ctx.count_errors(r, d_payload, counters)
        allreduce_counters(counters)               # the single collective of the path
        return r

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    # The same K steps once more (all ranks: the step holds the collective) with, on rank 0, every
    # kernel launch bracketed by HIP events on the launch stream (ultra_hip_profile_enable): per-kernel
    # durations for the roofline object.  Kept out of the timed region because the ~22 event records
    # per step cost ~1.7 % of a step.
    prof = None
    if rank == 0:
        ctx.profile_read()
        ctx.profile_enable(True)
    for _ in range(args.steps):
        step()
    barrier()
    if rank == 0:
        ctx.profile_enable(False)
        prof = ctx.profile_read()
    t_max = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    elapsed = float(t_max.item())
    stats = counters_dict(counters.cpu())

    # ---- roofline of the dominant kernel (largest share of the step), rank 0 ------------------
    # Algorithmic bytes per launch = SURVEY.md 8(d)'s per-frame figure (20,581 B = 4 x 4480 audio + 4
    # + 4 x 648 LLR + 61 + 4) apportioned to the launch that moves them, x the frames of one launch:
    #   mix_fft_kernel  one OFDM symbol of audio in            4480 B/frame/launch
    #   cfo_walk_kernel, track_pilot_kernel  work on the per-frame records between the kernels only   0 B/frame/launch
    #   track_kernel    that symbol's share of the 648 LLRs     648 B/frame/launch
    #   ldpc_decode     648 LLRs in, 61 bytes + iters + ok out 2658 B/frame/launch
    roofline = None
    if rank == 0:
        per_launch = {"cfo_walk_kernel": 0, "mix_fft_kernel": geo.symbol_samples * 4, "track_pilot_kernel": 0, "track_kernel": 648,
                      "ldpc_decode_kernel": BYTES_PER_CW_LDPC}
        traffic_all = {}
        tf = ROOT / "profiles" / "traffic.json"              # PMC-derived HBM bytes per launch, if collected
        if tf.exists():
            try:
                t = json.loads(tf.read_text())
                if t.get("n_frames") == n_frames:
                    traffic_all = {k: v["hbm_bytes_per_launch"] for k, v in t.get("kernels", {}).items()}
            except Exception:
                traffic_all = {}
        kernels = {}
        for name, (ms_total, launches) in prof.items():
            if launches == 0 or name not in per_launch:
                continue
            avg = ms_total / launches
            alg = n_frames * per_launch[name]
            kernels[name] = {"avg_launch_ms": avg, "launches_per_step": launches / args.steps,
                             "ms_per_step": ms_total / args.steps, "algorithmic_bytes_per_launch": alg,
                             "GBps": alg / (avg * 1e-3) / 1e9, "frac": alg / (avg * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                             "traffic": traffic_all.get(name)}
        # The decoder's limiters are its VALU issue and the LDS pipeline, not HBM: report the LDS utilisation next to
        # the HBM figure.  One codeword-iteration of the R3/4 instance issues 36 ds_read_b32 (2 cycles each for 64
        # lanes), 18 ds_write_b32 (4 cycles: address + data, the check step's scattered stores) and 18
        # ds_write_addtid_b32 (2 cycles: the variable step's lane-linear stores) = 180 LDS cycles on the CU's single
        # LDS pipeline; a codeword that converges at iteration `it` executes it + 1 iterations, a
        # failing one max_iterations.
        if "ldpc_decode_kernel" in kernels and world == 1:
            props = torch.cuda.get_device_properties(0)
            clock_hz = 2.4e9                                   # MI355X peak engine clock (MI355X_MICROARCH.md)
            executed = stats["iters_sum"] + (stats["frames"] - stats["ldpc_fail"])
            lds_cycles = 180.0 * executed
            avail = kernels["ldpc_decode_kernel"]["avg_launch_ms"] * 1e-3 * clock_hz * props.multi_processor_count
            kernels["ldpc_decode_kernel"]["lds"] = {"lds_cycles_per_codeword_iteration": 180,
                                                    "codeword_iterations_per_launch": executed,
                                                    "compute_units": props.multi_processor_count, "clock_hz": clock_hz,
                                                    "frac_of_lds_peak": lds_cycles / avail}
        dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
        roofline = {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["GBps"], "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": kernels[dom]["frac"], "traffic": kernels[dom]["traffic"],
                    "launch_ms": kernels[dom]["avg_launch_ms"],
                    "algorithmic_bytes_per_launch": kernels[dom]["algorithmic_bytes_per_launch"],
                    "kernels": kernels,
                    "path_GBps": n_frames * BYTES_PER_FRAME_FUSED * args.steps / elapsed / 1e9,
                    "note": "achieved = algorithmic bytes per launch / mean launch duration (HIP events around every "
                            "launch of a repeat of the timed steps). The path is not HBM-bound at this operating point: mix_fft is "
                            "latency/VALU-bound (double-precision sincos of the CFO rotation, 1024-point FFT through "
                            "LDS), and the reference decodes only ~11 % of these frames (R3/4 leaves 161 info bits "
                            "unchecked, the two-tap channel nulls 1 kHz), so most codewords run all 50 BP iterations "
                            "and ldpc_decode_kernel is bound by VALU issue (~200 instructions per codeword-iteration) and "
                            "the LDS pipeline (180 cycles per codeword-iteration, conflict-free). path_GBps = frames/s x 20,581 B (SURVEY 8d) for one GPU"}

    # ---- CPU baseline: the oracle on the host cores, bounded sample (rank 0, N=1 only) ----------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cores = os.cpu_count() or 1
        threads = min(cores, 64)
        # bounded sample: ~10-20 core-seconds of CPU work (the first frames of this rank's batch)
        sample = args.cpu_sample or min(n_frames, 1024 * threads)
        if audio_u is None:
            audio_cpu = d_audio[:sample].cpu().numpy()          # device stimulus: copy the sample to the host
        else:
            audio_cpu = np.tile(audio_u, (-(-sample // unique), 1))[:sample]
        t0 = time.perf_counter()
        want = o.demod_decode_batch(ccfg, audio_cpu, n_threads=threads, want_llr=False, want_state=False)
        t_cpu = time.perf_counter() - t0
        audio_ref = np.ascontiguousarray(audio_cpu[:512])           # sample of the compiled reference, below
        del audio_cpu
        got = {k: v[:sample].cpu().numpy() for k, v in out.items()}
        parity = bool(np.array_equal(got["bytes"], want["bytes"]) and np.array_equal(got["iters"], want["iters"])
                      and np.array_equal(got["ok"], want["ok"]))
        cpu = {"value": sample / t_cpu, "unit": "frames/s", "cores": threads, "kind": "port",
               "sample": f"first {sample} frames of the same batch, oracle/ultra_oracle.c, one worker thread per core "
                         f"({threads} of {cores} logical CPUs)",
               "seconds": t_cpu, "gpu_matches_cpu_bitwise": parity}
        if have_ref():
            nref = min(512, sample)
            t0 = time.perf_counter()
            rr = Ref().demod_decode_batch(ccfg, audio_ref[:nref])
            t_ref = time.perf_counter() - t0
            cpu["reference_1core"] = {"value": nref / t_ref, "unit": "frames/s", "cores": 1, "kind": "reference",
                                      "sample": f"first {nref} frames, compiled reference (oracle/_ref), 1 thread",
                                      "gpu_matches_reference_bitwise": bool(
                                          np.array_equal(got["bytes"][:nref], rr["bytes"])
                                          and np.array_equal(got["iters"][:nref], rr["iters"]))}

    if rank == 0:
        total_frames = n_frames * world * args.steps
        value = total_frames / elapsed
        line = {
            "metric": "OFDM-1024 16QAM R3/4 frames decoded/sec",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": (f"synthetic ({n_frames} distinct Watterson realisations per GPU generated on the device in HBM; "
                     if args.stimulus == "device" else
                     f"synthetic ({unique} distinct Watterson realisations per GPU generated on the host, tiled to "
                     f"{n_frames} frames resident in HBM; ") + "random-payload R3/4 codewords, 30 dB, 0.5 ms / 0.1 Hz)",
            "config": {"workload": "OFDM 1024-FFT 16QAM R3/4, 59 carriers (15 pilots), Watterson good channel, "
                                   "post-sync entry, LDPC min-sum <= 50 iterations, "
                                   f"{n_frames} frames per GPU per step",
                       "frames_per_gpu": n_frames, "bytes_per_frame": BYTES_PER_FRAME_FUSED,
                       "parallelism": f"frames sharded over {world} GPU(s), one counter all-reduce per step"},
            "achieved_hbm_GBps": value * BYTES_PER_FRAME_FUSED / 1e9,
            "hbm_frac_of_peak": value * BYTES_PER_FRAME_FUSED / 1e9 / (HBM_PEAK_GBPS * world),
            "fer": stats["fer"], "ber": stats["ber"], "mean_bp_iterations": stats["mean_iters"],
            "frames_counted": stats["frames"], "stimulus_seconds": t_gen,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
