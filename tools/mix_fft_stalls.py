#!/usr/bin/env python3
"""Where mix_fft_kernel / mix_fft2_kernel spend their wave-cycles: per-phase shader-clock stamps of EVERY work item of one
launch (diagnostic build -DUH_MIXFFT_STAMPS, demod_kernel.h) against the issue cycles of the same phases' static ISA.

    hipcc ... -DUH_MIXFFT_STAMPS -o build/stamps.so ultra_hip.hip          (done by tools/build_variants.sh)
    python3 tools/mix_fft_stalls.py [--frames 131072] [--one-wave] [--lib build/stamps.so] > profiles/r03_mix_fft_stalls_<variant>.txt

Per phase it prints the elapsed cycles per wavefront (mean / median / p90 over all wavefronts of the LAST data symbol's
launch: tracker CFO on, rotation table in use), the phase's static instruction mix priced with the measured issue table
(profiles/r02_issue_table.txt via tools/issue_model.py) and the ratio elapsed / own issue cycles.  With W wavefronts
resident per SIMD that all issue VALU work, a phase that only waits for the SIMD's issue slot shows a ratio of about W; a
larger ratio is time the wavefront spent waiting for something else (LDS, L2, HBM, the partner wavefront).  The HW_ID of
every record gives the measured residency (wavefronts per SIMD actually working at a time)."""
import argparse
import collections
import ctypes as C
import os
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))

PHASES = ["table header, wait for the staged audio, read samples", "(stamp pair)", "oscillator request, segment lookup, phases",
          "sincos block + park rotation factors", "mixing (oscillator values arrive)", "FFT group A + transpose out",
          "FFT group B (transpose in/out)", "FFT group C (transpose in, L2 twiddles)", "exchange + workgroup barrier",
          "last stage + bin store", "phase store + next prefetch issue"]


def static_phases(one_wave, norot=False):
    import issue_model as im
    costs, _ = im.parse_issue_table(ROOT / "profiles" / "r02_issue_table.txt")
    src = ROOT / "projectultra_amd" / "csrc" / "ultra_hip.hip"
    asm = subprocess.check_output(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
                                   "-fhip-fp32-correctly-rounded-divide-sqrt", "-DUH_MIXFFT_STAMPS"] + os.environ.get("STAMP_FLAGS", "").split() + ["-S", "--cuda-device-only", "-o", "-", str(src)],
                                  stderr=subprocess.DEVNULL, cwd=src.parent).decode()
    fns = im.functions(asm)
    key = "mix_fft_kernelILi10E" if one_wave else ("mix_fft2_kernelILi10ELb0E" if norot else "mix_fft2_kernelILi10ELb1E")
    body = next(v for k, v in fns.items() if key in k)
    cuts = [i for i, l in enumerate(body) if "UHSTAMP" in l]
    ids = [int(re.search(r"UHSTAMP (\d+)", body[i]).group(1)) for i in cuts]
    # dynamic phase k = stamps k -> k + 1.  Markers in program order: 0, 1, 2 (outer), [11, 12, 13: sub-stamps of the
    # lookup, two-wave kernel], 2 (inside `if (cfo_on)`), 3 .. 10; the code behind marker m belongs to phase m, the lookup
    # phase (1) also owns the code behind the outer 2 and the sub-stamps
    phases = [dict(valu=0, valu_cycles=0.0, salu=0, lds=0, lds_cycles=0.0) for _ in range(10)]
    seen2 = 0
    for a, b, m in zip(cuts, cuts[1:] + [len(body)], ids):
        if m == 2:
            seen2 += 1
        k = 0 if m == 0 else 1 if (m in (1, 11, 12, 13) or (m == 2 and seen2 == 1)) else m
        if k > 9:
            continue
        x = im.mix(body[a:b], costs)
        for f in phases[k]:
            phases[k][f] += x[f]
    return phases, costs, len(cuts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1 << 17)
    ap.add_argument("--one-wave", action="store_true", help=argparse.SUPPRESS)   # round 2's kernel: removed from the product in round 4
    ap.add_argument("--norot", action="store_true", help="the instance WITHOUT the rotation: stamps of symbol 1's launch (CFO 0 in every frame)")
    ap.add_argument("--lib", default=str(ROOT / "build" / "stamps.so"))
    ap.add_argument("--no-static", action="store_true")
    args = ap.parse_args()
    os.environ["ULTRA_HIP_LIB"] = args.lib
    if args.one_wave:
        raise SystemExit("the one-wavefront transform left the product in round 4 (its stall table: profiles/r03_mix_fft_stalls_one_wave.txt)")
    stat = None
    if not args.no_static:
        stat = static_phases(args.one_wave, args.norot)  # before anything touches the GPU (starts the compiler)
    import numpy as np
    import torch
    from projectultra_amd import CodeRate, Modulation, ReceiveContext, presets, _lib
    mc = presets.nvis_mode().with_mode(Modulation.QAM16, CodeRate.R3_4)
    mc.pilot_spacing = 4
    ctx = ReceiveContext(mc)
    n = args.frames
    audio, _ = ctx.make_batch(n, seed=0x5EED, channel="watterson", snr_db=30.0)
    for _ in range(2):
        ctx.demod(audio)
    torch.cuda.synchronize()
    waves = 1 if args.one_wave else 2
    WORDS = 16
    buf = torch.zeros(n * waves * WORDS, dtype=torch.int64, device="cuda")
    fn = ctx.lib.ultra_hip_debug_set_stamps
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
    assert fn(ctx._ctx, buf.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if args.norot:                                   # symbols 0 and 1 only: both launches are the instance without the rotation
        g = ctx.geometry
        two = audio[:, :2 * g.symbol_samples].contiguous()
        e0.record(); ctx.demod_stream(two, 0, 2); e1.record()
    else:
        e0.record(); ctx.demod(audio); e1.record()
    torch.cuda.synchronize()
    assert fn(ctx._ctx, None) == 0
    r = buf.cpu().numpy().reshape(n * waves, WORDS).astype(np.uint64)
    t = r[:, :11].astype(np.int64)
    hw = (r[:, 11] & np.uint64(0xffffffff)).astype(np.int64); xcc = (r[:, 11] >> np.uint64(32)).astype(np.int64) & 0xf
    simd = (xcc << 20) | (((hw >> 13) & 7) << 16) | (((hw >> 12) & 1) << 12) | (((hw >> 8) & 0xf) << 4) | ((hw >> 4) & 3)
    d = np.diff(t, axis=1)                       # [records][10]
    ex = r[:, 12:16].astype(np.int64)            # sub-stamps 11, 12, 13 inside the lookup phase (two-wave kernel)
    total = t[:, 10] - t[:, 0]
    if os.environ.get("UH_STAMP_FLAGS_HIST"):
        vals, cnts = np.unique(r[:, 15], return_counts=True)
        print("# flag word histogram:", dict(zip(vals.tolist(), cnts.tolist())), file=sys.stderr)
        for fl in (16, 32):
            m = (r[:, 15] & np.uint64(fl)) != 0
            if m.any():
                print(f"# flag {fl}: {m.mean():.3f} of items, sincos phase mean {d[m, 3].mean():.0f} cycles, item mean {total[m].mean():.0f}", file=sys.stderr)
    print(f"# {'mix_fft_kernel<10> (one wavefront per frame)' if args.one_wave else ('mix_fft2_kernel<10, false> (no rotation; symbols 0 and 1 only)' if args.norot else 'mix_fft2_kernel<10> (two wavefronts per frame)')}, "
          f"{n} frames, stamps of the last data symbol's launch; demodulation of the batch with stamps on: {e0.elapsed_time(e1):.3f} ms")
    # residency: wavefronts working at a time per SIMD
    res, spans = [], []
    for k in np.unique(simd):
        m = simd == k
        span = t[m, 10].max() - t[m, 0].min()
        res.append(total[m].sum() / span); spans.append(span)
    print(f"# SIMDs seen {len(res)}, wavefront-items per SIMD {len(simd) / len(res):.1f}; working wavefronts per SIMD (sum of item lifetimes / span): "
          f"mean {np.mean(res):.2f}, min {np.min(res):.2f}, max {np.max(res):.2f}; span per SIMD mean {np.mean(spans) / 1e3:.0f} k cycles, max {np.max(spans) / 1e3:.0f} k")
    print(f"# cycles per wavefront-item: mean {total.mean():.0f}, median {np.median(total):.0f}, p90 {np.percentile(total, 90):.0f}")
    W = float(np.mean(res))
    hdr = f"{'phase':58s} {'mean':>7s} {'median':>7s} {'p90':>7s} {'share':>6s}"
    if stat:
        hdr += f" {'VALU':>5s} {'issue':>6s} {'SALU':>5s} {'LDS':>4s} {'LDScyc':>6s} {'elapsed/issue':>13s}"
    print(hdr)
    parts = None
    if stat:
        parts, costs, ncuts = stat
    tot_issue = 0.0
    for k in range(10):
        name = PHASES[k if k < 1 else k + 1] if k >= 1 else PHASES[0]
        col = d[:, k]
        line = f"{name:58s} {col.mean():7.0f} {np.median(col):7.0f} {np.percentile(col, 90):7.0f} {100 * col.mean() / total.mean():5.1f}%"
        if parts:
            ps = [parts[k]]
            valu = sum(p["valu"] for p in ps); vc = sum(p["valu_cycles"] for p in ps); salu = sum(p["salu"] for p in ps)
            lds = sum(p["lds"] for p in ps); lc = sum(p["lds_cycles"] for p in ps)
            tot_issue += vc
            line += f" {valu:5d} {vc:6.0f} {salu:5d} {lds:4d} {lc:6.0f} {col.mean() / max(vc, 1):13.1f}"
        print(line)
    if parts:
        print(f"# static VALU issue cycles per wavefront-item {tot_issue:.0f} (loops counted once: the segment count loop runs n_segments times); "
              f"elapsed / issue overall {total.mean() / tot_issue:.2f} at {W:.2f} working wavefronts per SIMD -> VALU issue busy about {100 * W * tot_issue / total.mean():.0f} %")
    if not args.one_wave and ex[:, 0].any():
        cfo = ex[:, 0] > 0
        a = (ex[cfo, 0] - t[cfo, 1]); b = (ex[cfo, 1] - ex[cfo, 0]); c = (ex[cfo, 2] - ex[cfo, 1]); dd = (t[cfo, 2] - ex[cfo, 2])
        print(f"# inside the lookup phase ({cfo.mean() * 100:.0f} % of the items rotate): header readlanes + table entries into LDS {a.mean():.0f}, "
              f"segment mask + entries back {b.mean():.0f}, phase evaluation (+ search when needed) {c.mean():.0f}, "
              f"bin store + request of the next item {dd.mean():.0f} cycles (mean); no lane had to walk on "
              f"through the table (at most one boundary per run) in {100.0 * (ex[cfo, 3] == 1).mean():.1f} % of the rotating items")
    return 0


if __name__ == "__main__":
    sys.exit(main())
