"""Thread scaling of the CPU legs of bench.py's cpu_baseline on this host (cfg3 frames, post-sync demodulate + decode):
the compiled reference (one OFDMDemodulator / LDPCDecoder per thread) and the oracle port.  python3 tools/ref_scaling.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from oracle.bindings import Oracle, Ref, have_ref, make_config
import bench
model, cores, physical, logical, quota = bench.host_cpu()
print(f"# {model}: {physical} physical cores, {logical} logical CPUs, cgroup CPU quota {quota}, usable {cores}")
o = Oracle()
cfg = make_config(1024, "QAM16", "R3_4")
a, _ = o.make_batch(cfg, 8192, channel="watterson", snr_db=30.0, n_threads=min(cores, 64))
ref = Ref() if have_ref() else None
for nt in [t for t in (1, 2, 4, 8, 16, 32, 64, 128, 256) if t <= logical]:
    n = min(8192, 256 * nt)
    t0 = time.perf_counter(); o.demod_decode_batch(cfg, a[:n], n_threads=nt, want_llr=False, want_state=False); tp = time.perf_counter() - t0
    line = f"threads {nt:4d}: port {n / tp:10.0f} frames/s"
    if ref is not None:
        t0 = time.perf_counter(); ref.demod_decode_batch_mt(cfg, a[:n], nt); tr = time.perf_counter() - t0
        line += f"   reference {n / tr:10.0f} frames/s"
    print(line, flush=True)
