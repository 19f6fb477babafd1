#!/usr/bin/env python3
"""Is the headline's synthetic workload the reference's channel?  (CPU only; needs the compiled reference, oracle/_ref.)

The bench generates cfg3 frames on the device with a counter-based Watterson generator that is checked statistically against the
oracle's (tests/test_gpu_stimulus.py), and the oracle's generator is a structural restatement with its own random numbers.  FER
and the BP iteration distribution of this workload decide 42 % of the step time, so this tool closes the chain on the CPU: the
same 16QAM R3/4 frames (OFDM-1024, 59 carriers, pilot spacing 4) go through
  (a) the compiled reference's own WattersonChannel — itu_r_f1487 "good": two paths 0.5 ms apart, 0.1 Hz Doppler spread, gains
      0.707 / 0.707, 30 dB, fading restarted per frame (/root/reference/src/sim/hf_channel.hpp:106-168,406-418) — and
  (b) the oracle's generator (uo_make_batch, the twin of the device generator),
and both sets are received by the same post-sync demodulate + decode.  Compared: FER, undetected errors, BP iterations (mean,
share at the limit, share at <= 2), the tracker's noise-variance and SNR estimates.

    python tools/workload_check.py [n_frames] > profiles/r04_workload_check.txt
"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def workload_statistics(n=4096, snr_db=30.0, threads=8):
    from oracle.bindings import INFO_BITS, Ref, geometry, make_config, oracle
    o, r = oracle(), Ref()
    cfg = make_config(1024, "QAM16", "R3_4")
    g = geometry(cfg)
    k8 = INFO_BITS[4] // 8
    a_o, p_o = o.make_batch(cfg, n, seed=0x5EED, channel="watterson", snr_db=snr_db, delay_ms=0.5, doppler_hz=0.1)
    rng = np.random.default_rng(2026)
    a_r, p_r = np.zeros((n, g.frame_samples), np.float32), np.zeros((n, k8), np.uint8)
    for f in range(n):
        p_r[f] = rng.integers(0, 256, k8, dtype=np.uint8)
        sig, pre = o.modulate_frame(cfg, o.ldpc_encode(4, p_r[f].tobytes()))          # == the reference's modulator, bitwise
        sig = sig * np.float32(0.5 / np.abs(sig).max())                               # the harnesses' 0.5 peak
        a_r[f] = r.watterson(sig, snr_db, 0.5, 0.1, 1000 + f)[pre:pre + g.frame_samples]   # WattersonChannel(good), fresh per frame

    def stats(a, p):
        w = o.demod_decode_batch(cfg, a, n_threads=threads, want_llr=False, want_state=True)    # == the compiled reference, bitwise
        wrong = (w["bytes"][:, :k8] != p).any(axis=1)
        it, ok = w["iters"], w["ok"] == 1
        return dict(frames=n, fer=float(wrong.mean()), ldpc_fail=float((~ok).mean()), undetected=float((wrong & ok).mean()),
                    mean_iters=float(it.mean()), at_limit=float((it == 50).mean()), at_most_2=float((it <= 2).mean()),
                    noise_var_median=float(np.median(w["state"][:, 1])), noise_var_mean=float(w["state"][:, 1].mean()),
                    snr_linear_median=float(np.median(w["state"][:, 2])))
    return stats(a_r, p_r), stats(a_o, p_o)


def tolerances(n):
    """4 sigma of the difference of two independent samples of n frames (shares: binomial at the observed level; iterations:
    the distribution is bimodal {~1, 50}, sigma ~ 24)."""
    s = lambda p: 4.0 * np.sqrt(2.0 * p * (1.0 - p) / n)
    return dict(fer=s(0.9), ldpc_fail=s(0.59), undetected=s(0.3), at_limit=s(0.59), at_most_2=s(0.4), mean_iters=4.0 * 24.0 * np.sqrt(2.0 / n))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    ref, orc = workload_statistics(n)
    tol = tolerances(n)
    print(f"# cfg3 workload, {n} frames each, 30 dB: the compiled reference's WattersonChannel(good) vs the oracle's generator (twin of the")
    print("# device generator); same receive path (oracle port == compiled reference, bitwise).  tolerance = 4 sigma of the difference.")
    print(f"{'statistic':<22}{'reference channel':>20}{'oracle generator':>20}{'difference':>14}{'tolerance':>12}")
    for k in ref:
        if k == "frames":
            continue
        d = orc[k] - ref[k]
        t = tol.get(k)
        rel = f"{d / ref[k] * 100:+.1f} %" if t is None else f"{d:+.4f}"
        print(f"{k:<22}{ref[k]:>20.4f}{orc[k]:>20.4f}{rel:>14}{('%.4f' % t) if t is not None else '10 %':>12}")
