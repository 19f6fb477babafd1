"""Acquisition / end-to-end receive throughput (scope row f1).  Run on the GPU box:
   python3 tools/acquire_bench.py [n_streams]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle.bindings import Oracle, INFO_BITS, make_config, geometry
from _util import context_for

n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
oracle = Oracle()
for fft, mod, rate in ((1024, "QAM16", "R3_4"), (512, "DQPSK", "R1_2")):
    cfg = make_config(fft, mod, rate)
    g = geometry(cfg)
    rng = np.random.default_rng(1)
    uniq = []
    for t in range(64):
        payload = bytes(rng.integers(0, 256, INFO_BITS[cfg.code_rate] // 8, dtype=np.uint8))
        a, pre = oracle.modulate_frame(cfg, oracle.ldpc_encode(int(cfg.code_rate), payload))
        a = a * np.float32(0.5 / np.abs(a).max())
        sigma = np.sqrt(np.mean(a.astype(np.float64) ** 2) / 10 ** 3.0)
        uniq.append((a + rng.normal(0, sigma, a.size)).astype(np.float32))
    uniq = np.stack(uniq)
    t0 = time.perf_counter()
    res = [oracle.acquire(cfg, x, 960) for x in uniq[:8]]
    t_cpu = (time.perf_counter() - t0) / 8
    ctx = context_for(cfg)
    d = torch.from_numpy(uniq).cuda().repeat(n_streams // 64, 1)
    for name, fn in (("acquire", lambda: ctx.acquire(d, 960)), ("receive", lambda: ctx.receive(d, 960))):
        r = fn(); ctx.synchronize()
        times = []
        for _ in range(5):                                   # best of five: single runs scatter by a few percent
            ctx.timer_begin(); r = fn(); times.append(ctx.timer_end())
        ms = min(times)
        extra = f"found {r['found'].float().mean().item():.2f}" if name == "acquire" else f"ok {r['ok'].float().mean().item():.2f}"
        print(f"fft {fft} {mod} {rate}: {name} {d.shape[0]} streams x {d.shape[1]} samples: {ms:.1f} ms, "
              f"{d.shape[0] / ms * 1e3:.0f} streams/s ({extra}); oracle CPU 1 thread {t_cpu * 1e3:.1f} ms/stream "
              f"= {1 / t_cpu:.0f} streams/s")
