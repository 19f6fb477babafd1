"""Randomised parity soak (run on the GPU box): many mode / rate / channel / SNR / CFO combinations, a few
thousand frames each, HIP path vs the oracle on the host cores — LLRs, decoded bytes, iteration counts and the
tracker state compared BITWISE; presynced cases with the CFO "never set" for a tenth of the frames (training-symbol
estimate), half of the cases with the channel's own CFO shift on their first rows (device shift vs oracle shift);
the layouts without pilots once more without initial offsets (the zero-CFO path of launch_demod).    python3 tools/soak_parity.py [frames_per_case] [seed]"""
import itertools, sys, time
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle.bindings import Oracle, make_config
from _util import context_for

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
o = Oracle()
rng = np.random.default_rng(seed)
modes = [(1024, "QAM16", "R3_4", {}), (1024, "QAM16", "R1_2", {}), (1024, "QAM32", "R3_4", {}), (1024, "QAM32", "R1_2", {}),
         (1024, "DQPSK", "R1_2", {}), (1024, "D8PSK", "R3_4", {}), (512, "DQPSK", "R1_2", {}), (512, "QPSK", "R2_3", {}),
         (512, "DBPSK", "R1_4", {}), (512, "BPSK", "R1_2", {}), (512, "QAM64", "R5_6", {}), (512, "QAM256", "R5_6", {}),
         (1024, "QAM16", "R2_3", dict(pilot_spacing=2)), (1024, "DQPSK", "R1_4", dict(pilot_spacing=2, use_pilots=1)),
         (512, "DQPSK", "R1_4", dict(use_pilots=1)), (1024, "QPSK", "R1_2", dict(pilot_spacing=3))]
total = bad = 0
t0 = time.time()
t_last = t0
for (fft, mod, rate, kw), (chan, snr) in itertools.product(modes, (("watterson", 30.0), ("watterson", 14.0), ("awgn", 22.0), ("awgn", 6.0))):
    for entry in (0, 1):
        cfg = make_config(fft, mod, rate, entry=entry, **kw)
        audio, payload = o.make_batch(cfg, n, seed=int(rng.integers(1 << 30)), channel=chan, snr_db=snr, n_threads=64)
        cfo = rng.normal(0, 8.0, n).astype(np.float32)
        cfo[rng.random(n) < 0.1] = 0.0
        cfo[rng.random(n) < 0.02] *= 10.0                    # tracker saturation at +-90 Hz
        ph = rng.uniform(-3.1, 3.1, n).astype(np.float32) if entry == 1 else None
        if entry == 1:
            cfo[rng.random(n) < 0.1] = np.nan               # "never set": estimateCFOFromTraining (ofdm_sync.cpp:278-380)
        ctx = context_for(cfg)
        chan_ok = True
        if rng.random() < 0.5:
            # the channel's own carrier offset (WattersonChannel::applyCFO) on the first rows: device shift == oracle shift,
            # bitwise, and the receive path then runs on the shifted audio with a coarse estimate near it
            chan_cfo = float(rng.uniform(-30.0, 30.0))
            m = min(n, 512)
            shifted = np.stack([o.channel_apply_cfo(row, chan_cfo) for row in audio[:m]])
            dev = ctx.channel_cfo(torch.from_numpy(audio[:m]).cuda(), chan_cfo).cpu().numpy()
            chan_ok = np.array_equal(dev.view(np.uint32), shifted.view(np.uint32))
            audio[:m] = shifted
            keep_nan = np.isnan(cfo[:m])
            cfo[:m] = np.where(keep_nan, np.nan, chan_cfo + rng.normal(0, 1.0, m)).astype(np.float32)
        want = o.demod_decode_batch(cfg, audio, cfo_hz=cfo, cfo_phase=ph, n_threads=64)
        r = ctx.demod_decode(audio, cfo_hz=cfo, cfo_phase=ph, want_llr=True)
        ctx.synchronize()
        ok = chan_ok and (np.array_equal(r["llr"].cpu().numpy().view(np.uint32), want["llr"].view(np.uint32))
              and np.array_equal(r["bytes"].cpu().numpy(), want["bytes"]) and np.array_equal(r["iters"].cpu().numpy(), want["iters"])
              and np.array_equal(r["ok"].cpu().numpy(), want["ok"]))
        if entry == 0 and not kw.get("use_pilots") and mod in ("DQPSK", "D8PSK", "DBPSK"):
            # no pilots, no initial offsets: the zero-CFO path (all symbols in one transform launch and one tracking launch)
            want0 = o.demod_decode_batch(cfg, audio, n_threads=64)
            r0 = ctx.demod_decode(audio, want_llr=True)
            ctx.synchronize()
            ok = ok and (np.array_equal(r0["llr"].cpu().numpy().view(np.uint32), want0["llr"].view(np.uint32))
                         and np.array_equal(r0["bytes"].cpu().numpy(), want0["bytes"]) and np.array_equal(r0["iters"].cpu().numpy(), want0["iters"])
                         and np.array_equal(r0["ok"].cpu().numpy(), want0["ok"]))
            total += n
        total += n
        if mod not in ("DQPSK", "D8PSK", "DBPSK"):
            # the adaptive equaliser (ModemConfig::adaptive_eq_enabled, off in every preset): LMS or RLS with random step
            # sizes, frames three symbols longer so that the weights run free behind their re-seeding, a quarter of the frames
            from _util import geometry
            kind = "lms" if rng.random() < 0.5 else "rls"
            nsym = int(geometry(cfg).llrs_per_frame // geometry(cfg).llrs_per_symbol) + 3
            acfg = make_config(fft, mod, rate, entry=entry, adaptive_eq=kind, lms_mu=float(rng.uniform(0.01, 0.2)),
                               rls_lambda=float(rng.uniform(0.95, 0.999)), decision_directed=bool(rng.random() < 0.9),
                               n_data_symbols=nsym, **kw)
            na = max(n // 4, 64)
            a_audio, _ = o.make_batch(acfg, na, seed=int(rng.integers(1 << 30)), channel=chan, snr_db=snr, n_threads=64)
            a_cfo = rng.normal(0, 6.0, na).astype(np.float32)
            a_ph = rng.uniform(-3.1, 3.1, na).astype(np.float32) if entry == 1 else None
            wa = o.demod_decode_batch(acfg, a_audio, cfo_hz=a_cfo, cfo_phase=a_ph, n_threads=64)
            ra = context_for(acfg).demod_decode(a_audio, cfo_hz=a_cfo, cfo_phase=a_ph, want_llr=True)
            torch.cuda.synchronize()
            ok = ok and (np.array_equal(ra["llr"].cpu().numpy().view(np.uint32), wa["llr"].view(np.uint32))
                         and np.array_equal(ra["bytes"].cpu().numpy(), wa["bytes"]) and np.array_equal(ra["iters"].cpu().numpy(), wa["iters"]))
            total += na
        if not ok:
            bad += 1
            print("MISMATCH", fft, mod, rate, kw, chan, snr, "entry", entry)
        if time.time() - t_last > 60.0:                      # a line a minute: a silent run looks hung to the GPU box's watchdog
            t_last = time.time()
            print(f"... {total} frames, {bad} mismatching cases, {time.time() - t0:.0f} s", flush=True)
print(f"soak: {total} frames in {len(modes) * 8} cases, {bad} mismatching cases, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
