"""Randomised parity soak of the synchronisers (run on the GPU box): Schmidl-Cox acquisition (row f1) and chirp
synchronisation (row f4) on random streams — SNR, lead, level, noise-only — against the oracle, every reported
quantity compared exactly.    python3 tools/soak_sync.py [n_acquire_streams] [n_chirp_streams] [seed]"""
import sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle.bindings import INFO_BITS, Oracle, geometry, make_config
from _util import chirp_streams, context_for

n_acq = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n_chirp = int(sys.argv[2]) if len(sys.argv) > 2 else 96
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 3
o = Oracle()
rng = np.random.default_rng(seed)
pool = ThreadPoolExecutor(48)
bad = 0
t0 = time.time()
for fft, mod, rate in ((1024, "QAM16", "R3_4"), (512, "DQPSK", "R1_2")):
    cfg = make_config(fft, mod, rate)
    g = geometry(cfg)
    n = g.frame_samples + 7 * (fft + g.cp_len) + 3200
    frames = []
    for t in range(64):
        payload = bytes(rng.integers(0, 256, INFO_BITS[cfg.code_rate] // 8, dtype=np.uint8))
        a, _ = o.modulate_frame(cfg, o.ldpc_encode(int(cfg.code_rate), payload))
        frames.append(a * np.float32(0.5 / np.abs(a).max()))
    streams = np.zeros((n_acq, n), np.float32)
    for s in range(n_acq):
        a = frames[s % 64]
        if s % 37 == 0:
            streams[s] = rng.normal(0, 0.05, n)
            continue
        snr = rng.uniform(8.0, 32.0)
        sigma = np.sqrt(np.mean(a.astype(np.float64) ** 2) / 10 ** (snr / 10))
        lead = int(rng.integers(0, 2800))
        x = np.concatenate([rng.normal(0, 2e-4, lead), (a + rng.normal(0, sigma, a.size)) * rng.uniform(0.3, 1.2)])
        streams[s, :min(n, x.size)] = x[:n]
        if x.size < n: streams[s, x.size:] = rng.normal(0, 2e-4, n - x.size)
    want = list(pool.map(lambda x: o.acquire(cfg, x, 960), streams))
    ctx = context_for(cfg)
    r = {k: v.cpu().numpy() for k, v in ctx.acquire(streams, 960).items()}
    found = 0
    for s, w in enumerate(want):
        ok = int(r["found"][s]) == w["found"]
        if w["found"]:
            found += 1
            ok = ok and int(r["data_start"][s]) == w["data_start"] and int(r["sync_offset"][s]) == w["sync_offset"] \
                and int(r["fed_at_sync"][s]) == w["fed_at_sync"] and np.float32(r["cfo_hz"][s]).tobytes() == np.float32(w["coarse_cfo"]).tobytes()
        if not ok:
            bad += 1
            if bad < 5: print("ACQUIRE MISMATCH", fft, s, w, {k: v[s] for k, v in r.items()})
    print(f"acquire fft {fft}: {n_acq} streams, {found} found, mismatches so far {bad}")
cfg = make_config(512, "DQPSK", "R1_2", entry=1)
streams = []
while len(streams) < n_chirp:
    streams += chirp_streams(o, cfg, rng, n=5)
streams = streams[:n_chirp]
want = list(pool.map(lambda x: o.chirp_detect(x), streams))
ctx = context_for(cfg)
det = 0
for x, w in zip(streams, want):
    q = {k: v.cpu().numpy()[0] for k, v in ctx.chirp_sync(torch.from_numpy(x[None, :]).cuda()).items()}
    ok = int(q["detected"]) == w["success"] and int(q["start_sample"]) == w["start_sample"] \
        and int(q["up_chirp_start"]) == w["up_chirp_start"] and int(q["down_chirp_start"]) == w["down_chirp_start"] \
        and np.float32(q["cfo_hz"]).tobytes() == np.float32(w["cfo_hz"]).tobytes() \
        and np.float32(q["correlation"]).tobytes() == max(np.float32(w["up_correlation"]), np.float32(w["down_correlation"])).tobytes()
    det += w["success"]
    if not ok:
        bad += 1
        if bad < 5: print("CHIRP MISMATCH", w, q)
print(f"chirp: {n_chirp} buffers, {det} detected")
print(f"soak_sync: {2 * n_acq} acquisition streams + {n_chirp} chirp buffers, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
