#!/usr/bin/env python3
"""Phase stamps of ldpc_totals_kernel for ANY code rate on BPSK / AWGN soft bits (the cfg4 workload), by Es/N0 — where a
codeword's time goes when it converges at once (most points of the sweep) and when it runs all 50 iterations.

    bash tools/build_variants.sh ldstamps="-DUH_LDPC_STAMPS"
    python3 tools/ldpc_stalls_rate.py [--rate 0] [--esn0 -11 10] [--cw 131072] > profiles/r05_ldpc_stalls_r14.txt
"""
import argparse, ctypes as C, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
PH = ("LLRs landed + staged", "row phases", "drain after rows", "variable phases", "drain after variables", "outputs")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rate", type=int, default=0)
    ap.add_argument("--esn0", type=float, nargs="*", default=[-11.0, 10.0])
    ap.add_argument("--cw", type=int, default=1 << 17)
    ap.add_argument("--lib", default=str(ROOT / "build" / "v_ldstamps.so"))
    a = ap.parse_args()
    os.environ["ULTRA_HIP_LIB"] = a.lib
    import numpy as np, torch
    from projectultra_amd import CodeRate, LDPCDecoder
    d = LDPCDecoder(CodeRate(a.rate)); ctx = d.context
    fn = ctx.lib.ultra_hip_debug_set_ldpc_stamps
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
    n, W = a.cw, 10
    for es in a.esn0:
        llr, _ = ctx.make_llr_batch(n, es, seed=7)
        for _ in range(2): ctx.ldpc_decode(llr)
        torch.cuda.synchronize()
        buf = torch.zeros(n * W, dtype=torch.int64, device="cuda")
        assert fn(ctx._ctx, buf.data_ptr()) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = ctx.ldpc_decode(llr); e1.record(); torch.cuda.synchronize()
        assert fn(ctx._ctx, None) == 0
        e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e2.record(); ctx.ldpc_decode(llr); e3.record(); torch.cuda.synchronize()
        rec = buf.cpu().numpy().reshape(n, W).astype(np.uint64)
        life = (rec[:, 1] - rec[:, 0]).astype(np.int64); ph = rec[:, 2:8].astype(np.int64); ex = rec[:, 8].astype(np.int64)
        it = r["iters"].cpu().numpy()
        print(f"# rate {a.rate}, Es/N0 {es:+.1f} dB, {n} codewords: {e0.elapsed_time(e1):.3f} ms with stamps, {e2.elapsed_time(e3):.3f} ms without; mean iterations {it.mean():.2f}, "
              f"row phases executed per codeword {ex.mean():.2f}; cycles per codeword {life.mean():.0f} (shader clock, s_memtime)")
        tot = life.mean()
        for j, name in enumerate(PH):
            print(f"   {name:26s} {ph[:, j].mean():10.0f} cycles  {100 * ph[:, j].mean() / tot:5.1f} %")
        print(f"   {'unaccounted (queue, tail)':26s} {tot - ph.sum(axis=1).mean():10.0f} cycles")
    return 0


if __name__ == "__main__":
    sys.exit(main())
