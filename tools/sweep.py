#!/usr/bin/env python3
"""Sharded Monte-Carlo sweeps of BASELINE.json configs[3] and configs[4] (projectultra_amd/sweep.py).

    python tools/sweep.py --config cfg4 [--trials 1048576] [--out profiles/r02_sweep_cfg4.json]
    python tools/sweep.py --config cfg5 [--trials 12800]   [--channel awgn|watterson] [--out ...]
    N GPUs of one node:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                              --master-port P tools/sweep.py --config cfg4 ...

cfg4: LDPC R1/4 (or --rate), BPSK over AWGN, one point per dB from -11 to +30 dB Es/N0, --trials codewords per point
      (2^20 in BASELINE.json) sharded over the ranks, ONE all-reduce of the eight counters per point (cfg5: of the
      [points][8] block per curve — the points of a curve share their launches).
cfg5: {DBPSK, DQPSK, D8PSK, 16QAM, 32QAM} x {R1/4, R1/3, R1/2, R2/3, R3/4, R5/6} on the NVIS geometry (1024-FFT, 59 carriers,
      tools/test_nvis_mode.cpp:195-212), --trials frames per point; the default 30 cells x 11 SNR points x 12,800 frames
      = 4,224,000 frames, the 2^22 of BASELINE.json.
Rank 0 prints one summary line per curve and writes the curves (BER, FER, undetected-error rate, mean BP iterations,
raw counters per point) as JSON."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=("cfg4", "cfg5"), required=True)
    ap.add_argument("--trials", type=int, default=0, help="trials per point over all ranks (0 = the BASELINE size)")
    ap.add_argument("--rate", default="R1_4", help="cfg4: code rate")
    ap.add_argument("--channel", choices=("awgn", "watterson"), default="awgn", help="cfg5: channel")
    ap.add_argument("--snr", type=float, nargs="*", default=None, help="SNR points in dB (default: the config's axis)")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--per-cell", action="store_true", help="cfg5: one receive chain per cell (mode_sweep) instead of the shared grid")
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from projectultra_amd import CodeRate
    from projectultra_amd import sweep as sw

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("tools/sweep.py needs a GPU: the receive path has no CPU fallback")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)     # "nccl" IS RCCL on ROCm

    def show(p):
        if rank == 0:
            c = p.counters
            print(f"  {p.label:12s} {p.snr_db:6.1f} dB  FER {c['fer']:.5f}  BER {c['ber']:.3e}  undetected {c['undetected_rate']:.2e}  "
                  f"iters {c['mean_iters']:5.2f}  {p.trials / p.seconds / 1e6:7.2f} M trials/s", flush=True)

    t0 = time.perf_counter()
    if args.config == "cfg4":
        rate = CodeRate[args.rate]
        trials = args.trials or (1 << 20)
        snrs = args.snr if args.snr else sw.CFG4_SNR_POINTS
        pts = sw.ldpc_snr_sweep(rate, snrs, trials, seed=args.seed, rank=rank, world=world, batch=args.batch or (1 << 20), on_point=show)
        meta = dict(config="BASELINE.json configs[3]", rate=rate.name, stimulus="BPSK over AWGN, LLR = 2y/sigma^2, sigma^2 = 1/(2 Es/N0)",
                    snr_axis="Es/N0 dB", trials_per_point=trials)
    else:
        trials = args.trials or 12800
        snrs = args.snr if args.snr else sw.CFG5_SNR_POINTS
        if args.per_cell:
            pts = sw.mode_sweep(None, snrs, frames_per_point=trials, channel=args.channel, seed=args.seed, rank=rank, world=world,
                                batch=args.batch or (1 << 16), on_point=show)
        else:                                        # launches shared across cells (HipModeGrid): same counters
            pts = sw.mode_sweep_grid(snrs, frames_per_point=trials, channel=args.channel, seed=args.seed, rank=rank, world=world,
                                     grid_frames=args.batch or 1920, on_point=show)
        meta = dict(config="BASELINE.json configs[4]", geometry="1024-FFT, 59 carriers, CP 96 (presets::nvis_mode)", channel=args.channel,
                    snr_axis="SNR dB over the audio band (tools/test_nvis_mode.cpp:78-86)", trials_per_point=trials)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if rank == 0:
        doc = sw.curves_document("ldpc_snr_sweep" if args.config == "cfg4" else "mode_sweep", pts, n_gpus=world, seed=args.seed,
                                 wall_seconds=wall, device=torch.cuda.get_device_name(0), **meta)
        print(f"{args.config}: {len(pts)} points, {doc['total_trials']} trials on {world} GPU(s) in {wall:.2f} s "
              f"(incl. stimulus generation and context set-up) = {doc['total_trials'] / wall / 1e6:.2f} M trials/s")
        if args.out:
            Path(args.out).parent.mkdir(parents=True, exist_ok=True)
            Path(args.out).write_text(json.dumps(doc, indent=1))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
