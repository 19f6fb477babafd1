#!/usr/bin/env python3
"""Soft bits of the first frames of the cfg3 bench workload (same seed, channel and SNR as bench.py) as a .npy file, for offline
studies of the decoder's iteration behaviour:  python3 tools/dump_llrs.py [n_frames] [out.npy]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    from projectultra_amd import CodeRate, Modulation, ReceiveContext, presets
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    out = sys.argv[2] if len(sys.argv) > 2 else str(ROOT / "gpurun_out" / "cfg3_llrs.npy")
    mc = presets.nvis_mode().with_mode(Modulation.QAM16, CodeRate.R3_4)
    mc.pilot_spacing = 4
    ctx = ReceiveContext(mc)
    audio, payload = ctx.make_batch(n, seed=0x5EED, first_frame=0, channel="watterson", snr_db=30.0, delay_ms=0.5, doppler_hz=0.1)
    r = ctx.demod_decode(audio, want_llr=True)
    ctx.synchronize()
    Path(out).parent.mkdir(parents=True, exist_ok=True)
    np.save(out, r["llr"].cpu().numpy()[:, :648])
    np.save(out.replace(".npy", "_iters.npy"), r["iters"].cpu().numpy())
    print("frames", n, "mean iterations", float(r["iters"].float().mean()), "ok", float(r["ok"].float().mean()), "->", out)


if __name__ == "__main__":
    main()
