"""Chirp synchronisation / chirp-synchronised receive throughput (scope row f4).  Run on the GPU box:
   python3 tools/chirp_bench.py [n_streams]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle.bindings import Oracle, make_config, geometry
from _util import chirp_streams, context_for

n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
oracle = Oracle()
cfg = make_config(512, "DQPSK", "R1_2", entry=1)
rng = np.random.default_rng(3)
uniq = chirp_streams(oracle, cfg, rng, n=16)[:16]
n = max(len(x) for x in uniq)
uniq = np.stack([np.concatenate([x, rng.normal(0, 1e-3, n - len(x)).astype(np.float32)]) for x in uniq])
t0 = time.perf_counter()
res = [oracle.chirp_detect(x) for x in uniq[:4]]
t_cpu = (time.perf_counter() - t0) / 4
ctx = context_for(cfg)
d = torch.from_numpy(uniq).cuda().repeat(n_streams // 16, 1)
for name, fn in (("chirp_sync", lambda: ctx.chirp_sync(d)), ("chirp_receive", lambda: ctx.chirp_receive(d))):
    r = fn(); ctx.synchronize()
    ctx.timer_begin(); r = fn(); ms = ctx.timer_end()
    extra = f"detected {r['detected'].float().mean().item():.2f}" if name == "chirp_sync" else f"ok {r['ok'].float().mean().item():.2f}"
    print(f"{name} {d.shape[0]} streams x {d.shape[1]} samples: {ms:.1f} ms, {d.shape[0] / ms * 1e3:.0f} streams/s "
          f"({extra}); oracle detection, CPU 1 thread {t_cpu * 1e3:.1f} ms/stream = {1 / t_cpu:.1f} streams/s")
