#!/usr/bin/env python3
"""Per-step wall times of the cfg3 bench workload right after its set-up (is the first timed step slower?):
python3 tools/step_times.py [n_frames]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import types

def main():
    import torch
    import bench
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    args = types.SimpleNamespace(frames=n, total_frames=0, snr_db=None)
    wl = bench.ModemWorkload("cfg3", args, 0, 1, torch)
    ar = lambda t: None
    ts = []
    for i in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        wl.step(ar)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("single steps, ms:", " ".join(f"{t:.3f}" for t in ts))
    for k in (5, 5, 20):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k):
            wl.step(ar)
        torch.cuda.synchronize(); print(f"{k} steps back to back: {(time.perf_counter() - t0) / k * 1e3:.3f} ms per step")

if __name__ == "__main__":
    main()
