"""The product's FALL-BACK kernels, held to the same parity as the kernels that normally run.

libultra_hip.so reads two environment switches at ultra_hip_create (ultra_hip.hip, ultra_hip_ctx::old_chain):
  ULTRA_HIP_FALLBACK_CHAIN=1   per-symbol track_pilot_kernel + track_kernel for every layout — what launch_demod drops to
                               when the n_sym-fold workspace of the deferred carrier half cannot be had — and no pair tracker
  ULTRA_HIP_LDPC_MESSAGES=1    the message-passing decoder for every rate — what the totals decoders drop to when their LDS
                               placement is refused (no code runs it otherwise since round 4)
Each is a second product path; nothing else in tests/ reaches them for the layouts the fast kernels cover."""
import os

import numpy as np
import pytest

from _util import beq, cfg_from_array, context_for, make_config, noisy_codewords, nonfinite_cases
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture
def switch(request):
    name = request.param
    old = os.environ.get(name)
    os.environ[name] = "1"          # read by ultra_hip_create: every context made inside the test takes the fall-back
    yield name
    if old is None:
        del os.environ[name]
    else:
        os.environ[name] = old


@pytest.mark.parametrize("switch", ["ULTRA_HIP_FALLBACK_CHAIN"], indirect=True)
def test_fallback_chain_golden_and_synthetic(oracle, switch):
    """Every golden mode (the compiled reference's LLRs) and three synthetic batches (oracle) through the per-symbol chain:
    LLRs bitwise, tracker scalars, decode results."""
    g = np.load(GOLDEN / "demod.npz")
    for name in ["cfg3_qam16_r34", "cfg2_dqpsk_r12", "qam32_r34", "d8psk_r34", "dbpsk_r14", "bpsk_r12", "qam64_r34",
                 "dqpsk_pilots_r14", "qam16_r23_long"]:
        cfg = cfg_from_array(g[f"{name}__cfg"])
        ctx = context_for(cfg)
        llr, state = ctx.demod(g[f"{name}__audio"], cfo_hz=g[f"{name}__cfo"], want_state=True)
        ctx.synchronize()
        assert beq(llr.cpu().numpy(), g[f"{name}__llr"]), name
        scal = g[f"{name}__scal"][:, -1, :]
        for col in range(5):
            assert beq(state.cpu().numpy()[:, col], scal[:, col]), (name, col)
    for mod, rate, fft, chan, nodd in (("QAM16", "R3_4", 1024, "watterson", 193), ("DQPSK", "R1_2", 512, "awgn", 191),
                                       ("QAM32", "R3_4", 1024, "watterson", 64)):
        cfg = make_config(fft, mod, rate)
        audio, _ = oracle.make_batch(cfg, nodd, seed=0xFA11, channel=chan, snr_db=14.0)
        cfo = np.where(np.arange(nodd) % 3 == 0, 0.0, np.random.default_rng(3).normal(0, 4.0, nodd)).astype(np.float32)
        for c in (None, cfo):
            want = oracle.demod_decode_batch(cfg, audio, cfo_hz=c, n_threads=8)
            r = context_for(cfg).demod_decode(audio, cfo_hz=c, want_llr=True)
            assert beq(r["llr"].cpu().numpy(), want["llr"]), (mod, c is None)
            for k in ("bytes", "iters", "ok"):
                assert np.array_equal(r[k].cpu().numpy(), want[k]), (mod, k)


@pytest.mark.parametrize("switch", ["ULTRA_HIP_LDPC_MESSAGES"], indirect=True)
@pytest.mark.parametrize("rate", [0, 1, 2, 3, 4, 5])
def test_message_decoder_for_every_rate(oracle, switch, rate):
    """All six codes on the message-passing kernel (since round 4 every code normally runs a totals kernel, so this switch is
    the only way to it): waterfall mix, non-finite inputs, the golden reference cases."""
    from projectultra_amd import CodeRate, LDPCDecoder
    sig = {0: [0.9, 1.3, 1.7, 2.2], 1: [0.6, 0.8, 1.0, 1.3], 2: [0.6, 0.8, 1.0, 1.3], 3: [0.45, 0.6, 0.75, 0.9], 4: [0.4, 0.5, 0.6, 0.75],
           5: [0.3, 0.4, 0.5, 0.6]}[rate]
    llr, _ = noisy_codewords(oracle, rate, 768, sig, seed=300 + rate)
    d = LDPCDecoder(CodeRate(rate))
    r = d.decode_batch(llr, want_total=True)
    ob, oi, ook, ototal = oracle.ldpc_decode_batch(rate, llr, want_total=True)
    assert np.array_equal(r["iters"], oi) and np.array_equal(r["ok"], ook) and np.array_equal(r["bytes"], ob)
    assert beq(r["llr_total"], ototal)
    cases = nonfinite_cases(np.random.default_rng(177 + rate), oracle, rate, n=32)
    r = d.decode_batch(cases)
    ob, oi, ook = oracle.ldpc_decode_batch(rate, cases)[:3]
    assert np.array_equal(r["iters"], oi) and np.array_equal(r["ok"], ook) and np.array_equal(r["bytes"], ob)
    g = np.load(GOLDEN / "ldpc.npz")
    r = d.decode_batch(g[f"dec_llr_r{rate}"])
    assert np.array_equal(r["bytes"], g[f"dec_bytes_r{rate}"]) and np.array_equal(r["ok"], g[f"dec_ok_r{rate}"])
    assert np.array_equal(r["iters"], g[f"dec_iters_r{rate}"])


def test_unaligned_rows_and_odd_strides(oracle):
    """The C-ABI takes any float pointer and any row stride >= the frame: the same frames at row offsets of 0, 1, 2, 3 floats
    and odd row strides give the same bits (nothing in the transform's asynchronous copies may assume 16-byte alignment)."""
    import torch
    from _util import geometry
    for fft, mod, rate in ((1024, "QAM16", "R3_4"), (512, "DQPSK", "R1_2")):
        cfg = make_config(fft, mod, rate)
        g = geometry(cfg)
        n = 67
        audio, _ = oracle.make_batch(cfg, n, seed=0xA11C, channel="watterson", snr_db=20.0)
        want = oracle.demod_decode_batch(cfg, audio, n_threads=8)
        ctx = context_for(cfg)
        for shift, pad in ((0, 0), (1, 0), (2, 3), (3, 1), (0, 2), (4, 4)):
            stride = g.frame_samples + pad
            big = torch.zeros(n * stride + 8, dtype=torch.float32, device="cuda")
            rows = big[shift:shift + n * stride].view(n, stride)      # contiguous rows of `stride` floats that start `shift` floats in
            rows[:, :g.frame_samples].copy_(torch.from_numpy(audio))
            assert rows.is_contiguous() and rows.data_ptr() == big.data_ptr() + 4 * shift
            r = ctx.demod_decode(rows, want_llr=True)
            assert beq(r["llr"].cpu().numpy(), want["llr"]), (mod, shift, pad)
            for k in ("bytes", "iters", "ok"):
                assert np.array_equal(r[k].cpu().numpy(), want[k]), (mod, shift, pad, k)
