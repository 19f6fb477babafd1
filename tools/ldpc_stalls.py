#!/usr/bin/env python3
"""Where ldpc_totals_kernel spends a codeword's time: shader-clock time per phase of the decoding loop, summed per codeword by
the diagnostic build (-DUH_LDPC_STAMPS, ldpc_totals_kernel.h), against the issue cycles of the same phases' static ISA priced
with the measured per-opcode costs (profiles/r02_issue_table.txt via tools/issue_model.py).

    bash tools/build_variants.sh ldstamps="-DUH_LDPC_STAMPS"
    python3 tools/ldpc_stalls.py [--frames 131072] [--lib build/v_ldstamps.so] > profiles/r04_ldpc_stalls.txt

Workload: the headline's soft bits (cfg3: OFDM-1024 16QAM R3/4 over the Watterson good channel at 30 dB, demodulated on the
device), decoded by ultra_hip_ldpc_decode_batch.  Per phase: cycles per executed iteration of a codeword that runs all 50
(59 % of the workload and 96 % of its iterations), the phase's static instruction mix and issue cost, and elapsed / issue.
With W wavefronts per SIMD that all want the vector unit, a phase that only waits for its issue slot shows about W."""
import argparse
import ctypes as C
import os
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))

PHASES = {2: "LLRs landed, staged, first totals stored (per codeword)", 3: "row phase: gather totals, verdict, check step, c2v stores",
          4: "drain behind the row phase", 5: "variable phase: gather c2v, sums, totals stored", 6: "drain behind the variable phase",
          7: "outputs: bytes, iterations, status (per codeword)"}


def static_phases(flags):
    import issue_model as im
    costs, _ = im.parse_issue_table(ROOT / "profiles" / "r02_issue_table.txt")
    src = ROOT / "projectultra_amd" / "csrc" / "ultra_hip.hip"
    asm = subprocess.check_output(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
                                   "-fhip-fp32-correctly-rounded-divide-sqrt", "-DUH_LDPC_STAMPS"] + flags + ["-S", "--cuda-device-only", "-o", "-", str(src)],
                                  stderr=subprocess.DEVNULL, cwd=src.parent).decode()
    fns = im.functions(asm)
    body = next(v for k, v in fns.items() if "ldpc_totals_kernelILi3ELi6ELy1638ELy3355443ELb0E" in k)
    cuts = [(i, int(re.search(r"UHLDSTAMP (\d+)", l).group(1))) for i, l in enumerate(body) if "UHLDSTAMP" in l]
    # the code IN FRONT of marker k belongs to phase k (the accumulation closes the phase); marker order in the loop: 3, 4, 5, 6
    out = {}
    prev = 0
    for i, k in cuts:
        out.setdefault(k, dict(valu=0, valu_cycles=0.0, salu=0, lds=0, lds_cycles=0.0))
        x = im.mix(body[prev:i], costs)
        for f in out[k]:
            out[k][f] += x[f]
        prev = i
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1 << 17)
    ap.add_argument("--lib", default=str(ROOT / "build" / "v_ldstamps.so"))
    ap.add_argument("--no-static", action="store_true")
    args = ap.parse_args()
    os.environ["ULTRA_HIP_LIB"] = args.lib
    stat = None if args.no_static else static_phases([])        # before anything touches the GPU (starts the compiler)
    import numpy as np
    import torch
    from projectultra_amd import CodeRate, Modulation, ReceiveContext, presets
    mc = presets.nvis_mode().with_mode(Modulation.QAM16, CodeRate.R3_4)
    mc.pilot_spacing = 4
    ctx = ReceiveContext(mc)
    n = args.frames
    audio, _ = ctx.make_batch(n, seed=0x5EED, channel="watterson", snr_db=30.0)
    llr = ctx.demod(audio)[:, :648].contiguous()
    for _ in range(2):
        ctx.ldpc_decode(llr)
    torch.cuda.synchronize()
    WORDS = 10
    buf = torch.zeros(n * WORDS, dtype=torch.int64, device="cuda")
    fn = ctx.lib.ultra_hip_debug_set_ldpc_stamps
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
    assert fn(ctx._ctx, buf.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = ctx.ldpc_decode(llr); e1.record()
    torch.cuda.synchronize()
    assert fn(ctx._ctx, None) == 0
    e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e2.record(); ctx.ldpc_decode(llr); e3.record()
    torch.cuda.synchronize()
    rec = buf.cpu().numpy().reshape(n, WORDS).astype(np.uint64)
    iters = r["iters"].cpu().numpy()
    life = (rec[:, 1] - rec[:, 0]).astype(np.int64)
    ph = rec[:, 2:8].astype(np.int64)
    executed = rec[:, 8].astype(np.int64)
    hw = (rec[:, 9] & np.uint64(0xffffffff)).astype(np.int64); xcc = (rec[:, 9] >> np.uint64(32)).astype(np.int64) & 0xf
    simd = (xcc << 20) | (((hw >> 13) & 7) << 16) | (((hw >> 12) & 1) << 12) | (((hw >> 8) & 0xf) << 4) | ((hw >> 4) & 3)
    res = []
    for k in np.unique(simd):
        m = simd == k
        res.append(life[m].sum() / (rec[m, 1].max() - rec[m, 0].min()))
    W = float(np.mean(res))
    full = iters == 50
    print(f"# ldpc_totals_kernel<3, 6, 0x666, 0x333333, false, 5> (R3/4), {n} codewords of the headline workload; decode with stamps on {e0.elapsed_time(e1):.3f} ms, "
          f"the same library with the stamp buffer off {e2.elapsed_time(e3):.3f} ms")
    print(f"# codewords that run all 50 iterations: {full.mean() * 100:.1f} % ({executed[full].sum() / executed.sum() * 100:.1f} % of the executed iterations); "
          f"mean iterations {iters.mean():.2f}; SIMDs seen {len(res)}; wavefronts decoding at a time per SIMD (sum of lifetimes / span): mean {W:.2f}")
    print(f"# cycles per codeword: all 50 iterations {life[full].mean():.0f}, the others {life[~full].mean():.0f} (mean {executed[~full].mean():.1f} row phases)")
    hdr = f"{'phase (codewords that run 50 iterations)':62s} {'cycles':>8s} {'per it.':>8s} {'share':>6s}"
    if stat:
        hdr += f" {'VALU':>5s} {'issue':>6s} {'SALU':>5s} {'LDS':>4s} {'LDScyc':>6s} {'elapsed/issue':>13s}"
    print(hdr)
    tot = life[full].mean()
    for j, k in enumerate(range(2, 8)):
        c = ph[full, j].mean()
        per_it = c / 50.0 if k in (3, 4, 5, 6) else float("nan")
        line = f"{PHASES[k]:62s} {c:8.0f} {per_it:8.1f} {100 * c / tot:5.1f}%"
        if stat and k in stat and k in (3, 4, 5, 6):
            p = stat[k]
            line += f" {p['valu']:5d} {p['valu_cycles']:6.0f} {p['salu']:5d} {p['lds']:4d} {p['lds_cycles']:6.0f} {per_it / max(p['valu_cycles'], 1.0):13.2f}"
        print(line)
    it_cycles = sum(ph[full, j].mean() for j in (1, 2, 3, 4)) / 50.0
    if stat:
        vi = sum(stat[k]["valu_cycles"] for k in (3, 4, 5, 6) if k in stat)
        li = sum(stat[k]["lds_cycles"] for k in (3, 4, 5, 6) if k in stat)
        print(f"# one iteration: {it_cycles:.0f} cycles elapsed for {vi:.0f} cycles of vector issue (static; the verdict's rounds 1 and 2 counted although a failing "
              f"codeword skips them) and {li:.0f} cycles of the CU's LDS pipeline; at {W:.2f} wavefronts per SIMD the vector unit is busy about "
              f"{100 * W * vi / it_cycles:.0f} %, the LDS pipeline (shared by 4 SIMDs) about {100 * 4 * W * li / it_cycles:.0f} %")
    return 0


if __name__ == "__main__":
    sys.exit(main())
