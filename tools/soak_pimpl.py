#!/usr/bin/env python3
"""Randomised soak of the link-time drop-ins against the compiled reference, call by call (GPU box).

oracle/demod_pimpl_harness.cpp scripts the two pimpl classes' public interface (frames on one object without reset(), timing and
frequency offsets at every point of a frame, processPresynced in all its branches, mid-frame preambles, the exits of SYNCED, mixed
entries, the decoder's limits) from a SEED: stream contents, chunk sizes, noise, offsets.  tests/test_gpu_pimpl.py runs 31 fixed
(scenario, layout, seed) cases; this runs the same scenarios over many more seeds and layouts, `.ref` (the reference, CPU) against
`.hip` (the drop-ins, MI355X), and requires identical output — every answer with floats as bit patterns.

    python3 tools/soak_pimpl.py [seeds per case, default 12] [first seed, default 1000]
"""
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
T = ROOT / "oracle" / "_ref" / "tools"
MOD = dict(DBPSK=0, BPSK=1, DQPSK=2, QPSK=3, D8PSK=4, QAM16=6, QAM32=7, QAM64=8)
RATE = dict(R1_4=0, R1_3=1, R1_2=2, R2_3=3, R3_4=4, R5_6=5)
LAYOUTS = [(1024, "QAM16", "R3_4"), (512, "DQPSK", "R1_2"), (512, "QPSK", "R1_2"), (1024, "D8PSK", "R2_3"), (512, "QAM64", "R5_6"), (1024, "QAM32", "R3_4"),
           (512, "DBPSK", "R1_4")]
SCENARIOS = ["carry", "timing", "setcfo", "presynced", "midframe", "exits", "getdata", "mixed"]

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
t0 = time.time()
runs = mismatches = calls = 0
for sc in SCENARIOS:
    for fft, mod, rate in LAYOUTS:
        for seed in range(first, first + n_seeds):
            args = [sc, str(fft), str(MOD[mod]), str(RATE[rate]), str(seed)]
            ref = subprocess.run([str(T / "demod_pimpl_harness.ref")] + args, capture_output=True, text=True, timeout=600)
            hip = subprocess.run([str(T / "demod_pimpl_harness.hip")] + args, capture_output=True, text=True, timeout=600)
            runs += 1
            calls += ref.stdout.count("\n")
            if ref.returncode != hip.returncode or ref.stdout != hip.stdout:
                mismatches += 1
                a, b = ref.stdout.splitlines(), hip.stdout.splitlines()
                i = next((k for k, (x, y) in enumerate(zip(a, b)) if x != y), min(len(a), len(b)))
                print(f"MISMATCH {' '.join(args)}: rc {ref.returncode}/{hip.returncode}, line {i}\n  ref: {a[i][:200] if i < len(a) else '<end>'}\n  hip: {b[i][:200] if i < len(b) else '<end>'}", flush=True)
    print(f"... {sc}: {runs} runs, {calls} logged answers, {mismatches} mismatching runs, {time.time() - t0:.0f} s", flush=True)
print(f"soak_pimpl: {runs} runs ({len(SCENARIOS)} scenarios x {len(LAYOUTS)} layouts x {n_seeds} seeds), {calls} logged answers, {mismatches} mismatching runs, {time.time() - t0:.0f} s")
sys.exit(1 if mismatches else 0)
