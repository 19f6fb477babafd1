#!/usr/bin/env python3
"""Where acquire_kernel spends a stream's time: shader-clock stamps per phase of the search (diagnostic build
-DUH_ACQ_STAMPS, acquire_kernel.h), for the raw-audio bench's streams (bench.py --config raw: 65,536 streams of 14,400 samples,
a frame behind ~2,200 samples of noise, fed in 960-sample chunks).

    bash tools/build_variants.sh acqstamps="-DUH_ACQ_STAMPS"
    python3 tools/acquire_stalls.py [--streams 16384] > profiles/r05_acquire_stalls.txt
"""
import argparse, ctypes as C, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
NAMES = ("whole search", "energy-gate groups (64 candidates each)", "DC groups (64 windows each)", "window metrics: analytic FFT pair + half-symbol sums",
         "CFO metric at the chosen offset", "LTS matched filter (refineLTSTiming)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=1 << 14)
    ap.add_argument("--lib", default=str(ROOT / "build" / "v_acqstamps.so"))
    ap.add_argument("--fft", type=int, default=1024)
    a = ap.parse_args()
    os.environ["ULTRA_HIP_LIB"] = a.lib
    import numpy as np, torch
    from projectultra_amd import CodeRate, Modulation, ReceiveContext, presets, ModemConfig
    if a.fft == 1024:
        mc = presets.nvis_mode().with_mode(Modulation.QAM16, CodeRate.R3_4); mc.pilot_spacing = 4
    else:
        mc = ModemConfig().with_mode(Modulation.DQPSK, CodeRate.R1_2)
    ctx = ReceiveContext(mc)
    n = a.streams
    audio, _ = ctx.make_raw_batch(n, seed=0x5EED, channel="awgn", snr_db=30.0)
    for _ in range(2):
        ctx.acquire(audio, 960)
    torch.cuda.synchronize()
    W = 12
    buf = torch.zeros(n * W, dtype=torch.int64, device="cuda")
    fn = ctx.lib.ultra_hip_debug_set_acq_stamps
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
    assert fn(ctx._ctx, buf.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = ctx.acquire(audio, 960); e1.record(); torch.cuda.synchronize()
    assert fn(ctx._ctx, None) == 0
    rec = buf.cpu().numpy().reshape(n, W).astype(np.float64)
    tot = rec[:, 0].mean()
    print(f"# acquire_kernel<{10 if a.fft == 1024 else 9}, false>, {n} raw streams x {audio.shape[1]} samples (AWGN 30 dB), chunk 960: {e0.elapsed_time(e1):.2f} ms with stamps; "
          f"found {r['found'].float().mean().item():.3f}")
    print(f"# per stream: {rec[:, 10].mean():.1f} process() calls, {rec[:, 9].mean():.0f} candidates visited, {rec[:, 11].mean():.0f} metric-cache hits, "
          f"{rec[:, 6].mean():.1f} metrics evaluated, {rec[:, 7].mean():.1f} gate groups, {rec[:, 8].mean():.1f} DC groups")
    acc = 0.0
    for k in range(1, 6):
        c = rec[:, k].mean(); acc += c
        per = {1: rec[:, 7], 2: rec[:, 8], 3: rec[:, 6]}.get(k)
        each = f"  ({c / max(per.mean(), 1e-9):8.0f} cycles each)" if per is not None else ""
        print(f"   {NAMES[k]:58s} {c:12.0f} cycles  {100 * c / tot:5.1f} %{each}")
    print(f"   {'the walk itself (state machine, cache lookups, verdicts)':58s} {tot - acc:12.0f} cycles  {100 * (tot - acc) / tot:5.1f} %  ({(tot - acc) / rec[:, 9].mean():6.0f} cycles per candidate visited)")
    print(f"   {'whole search':58s} {tot:12.0f} cycles")
    return 0


if __name__ == "__main__":
    sys.exit(main())
