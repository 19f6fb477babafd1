"""LDPC-only throughput per code rate (SURVEY.md 8d cfg4 shape): BPSK/AWGN LLRs, 2^17 codewords.
Run on the GPU box: python3 tools/ldpc_bench.py"""
import sys, time
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from _util import noisy_codewords
from oracle.bindings import Oracle
from projectultra_amd import CodeRate, LDPCDecoder

oracle = Oracle()
SIG = {0: [1.6, 2.0, 2.4], 1: [1.0, 1.3, 1.6], 2: [0.8, 1.0, 1.2], 3: [0.6, 0.75, 0.9], 4: [0.5, 0.6, 0.7], 5: [0.4, 0.5, 0.6]}
for rate in range(6):
    llr, _ = noisy_codewords(oracle, rate, 2048, SIG[rate], seed=5)
    d = LDPCDecoder(CodeRate(rate)); ctx = d.context
    big = torch.from_numpy(llr).cuda().repeat(64, 1)
    for _ in range(2): r = ctx.ldpc_decode(big)
    ctx.synchronize(); ctx.timer_begin()
    for _ in range(5): r = ctx.ldpc_decode(big)
    ms = ctx.timer_end() / 5
    it = r["iters"].float().mean().item(); ok = r["ok"].float().mean().item()
    print(f"rate {rate}: {big.shape[0]} cw, {ms:.3f} ms, {big.shape[0]/ms/1e3:.2f} Mcw/s, mean iters {it:.1f}, ok {ok:.3f}")
