"""GPU parity: HIP LDPC decode (through the C-ABI) vs the oracle, bit-exact.

Oracle = oracle/ultra_oracle.c, pinned against the compiled reference; the committed
fixtures in tests/golden/ldpc.npz come straight from the compiled reference."""
import numpy as np
import pytest

from _util import INFO_BITS, beq, noisy_codewords, nonfinite_cases, valid_special_codewords
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
RATES = [0, 1, 2, 3, 4, 5]   # R1_4, R1_3 (default 324/324 code, seed+1), R1_2, R2_3, R3_4, R5_6


def _decoder(rate, max_iter=50):
    from projectultra_amd import CodeRate, LDPCDecoder
    d = LDPCDecoder(CodeRate(rate))
    if max_iter != 50:
        d.setMaxIterations(max_iter)
    return d


@pytest.mark.parametrize("rate", RATES)
def test_tanner_graph_matches_oracle(oracle, rate):
    d = _decoder(rate)
    rp, ci = d.context.tanner_graph()
    orp, oci, k, m = oracle.ldpc_graph(rate)
    assert np.array_equal(rp, orp) and np.array_equal(ci, oci)


@pytest.mark.parametrize("rate", RATES)
def test_golden_reference_cases(rate):
    """Inputs/outputs captured from the compiled reference (LDPCDecoder::decodeSoft)."""
    g = np.load(GOLDEN / "ldpc.npz")
    d = _decoder(rate)
    r = d.decode_batch(g[f"dec_llr_r{rate}"])
    assert np.array_equal(r["bytes"], g[f"dec_bytes_r{rate}"])
    assert np.array_equal(r["ok"], g[f"dec_ok_r{rate}"])
    assert np.array_equal(r["iters"], g[f"dec_iters_r{rate}"])
    # +-2.0 LLRs with 40 mt19937(7) sign/scale flips (SURVEY §8c known answer: iterations 1/2/4/4/4)
    out = d.decodeSoft(g[f"kat_flip_llr_r{rate}"])
    assert [d.lastIterations(), int(d.lastDecodeSuccess())] == g[f"kat_flip_iters_r{rate}"].tolist()
    if d.lastDecodeSuccess():
        assert out[:20] == g["kat_payload"].tobytes()


@pytest.mark.parametrize("rate", RATES)
def test_multiblock_and_short_inputs_match_reference(rate):
    g = np.load(GOLDEN / "ldpc.npz")
    d = _decoder(rate, max_iter=10)
    for n in (300, 1296, 1500):
        out = d.decodeSoft(g[f"mb_llr_r{rate}_{n}"])
        assert out == g[f"mb_out_r{rate}_{n}"].tobytes(), (rate, n)
        assert [int(d.lastDecodeSuccess()), d.lastIterations()] == g[f"mb_meta_r{rate}_{n}"].tolist()
    assert d.decodeSoft(np.zeros(0, np.float32)) == b"" and not d.lastDecodeSuccess()


@pytest.mark.parametrize("rate", RATES)
def test_random_noisy_codewords_bit_exact(oracle, rate):
    """Waterfall region: iteration counts spread over 0..50, failures included."""
    sig = {0: [0.9, 1.3, 1.7, 2.2], 1: [0.6, 0.8, 1.0, 1.3], 2: [0.6, 0.8, 1.0, 1.3], 3: [0.45, 0.6, 0.75, 0.9],
           4: [0.4, 0.5, 0.6, 0.75], 5: [0.3, 0.4, 0.5, 0.6]}[rate]
    llr, _ = noisy_codewords(oracle, rate, 768, sig, seed=100 + rate)
    d = _decoder(rate)
    r = d.decode_batch(llr, want_total=True)
    ob, oi, ook, ototal = oracle.ldpc_decode_batch(rate, llr, want_total=True)
    assert np.array_equal(r["iters"], oi), np.flatnonzero(r["iters"] != oi)[:8]
    assert np.array_equal(r["ok"], ook)
    assert np.array_equal(r["bytes"], ob)
    assert beq(r["llr_total"], ototal)          # final a-posteriori LLRs, bitwise
    assert 0 < oi.min() + 1 and oi.max() == 50 and (ook == 1).any() and (ook == 0).any(), "case mix too narrow"


@pytest.mark.parametrize("max_iter", [0, 1, 3, 200])
@pytest.mark.parametrize("rate", [0, 2, 4])
def test_iteration_limits(oracle, rate, max_iter):
    """setMaxIterations 0 / 1 / 3 / 200 on the three kernel families' codes (R1/4: profile kernel with 13-edge variables, R1/2:
    profile kernel, R3/4: regular totals kernel), through BOTH instances of each: with the a-posteriori LLRs (no iteration-0
    short cut) and without (the production instance)."""
    sig = {0: [1.3, 1.9], 2: [0.8, 1.1], 4: [0.55, 0.7]}[rate]
    llr, _ = noisy_codewords(oracle, rate, 96, sig, seed=7)
    valid = np.where(np.unpackbits(np.frombuffer(oracle.ldpc_encode(rate, bytes(INFO_BITS[rate] // 8)), np.uint8))[:648] > 0, -3.0, 3.0)
    llr[0] = valid.astype(np.float32)                                      # converges at iteration 0
    d = _decoder(rate, max_iter=max_iter)
    r = d.decode_batch(llr, want_total=True)
    ob, oi, ook, ototal = oracle.ldpc_decode_batch(rate, llr, max_iters=max_iter, want_total=True)
    assert np.array_equal(r["iters"], oi) and np.array_equal(r["ok"], ook) and np.array_equal(r["bytes"], ob)
    assert beq(r["llr_total"], ototal)
    p = d.decode_batch(llr)
    assert np.array_equal(p["iters"], oi) and np.array_equal(p["ok"], ook) and np.array_equal(p["bytes"], ob)


def test_edge_llrs(oracle):
    """Zero (erasure), negative-zero, huge, denormal and single-codeword batches."""
    rate = 2
    rng = np.random.default_rng(3)
    cases = np.stack([
        np.zeros(648, np.float32), np.full(648, -0.0, np.float32), np.full(648, 1e30, np.float32),
        np.full(648, -1e30, np.float32), rng.normal(0, 1e-40, 648).astype(np.float32),
        rng.normal(0, 60, 648).astype(np.float32), np.where(rng.random(648) < .5, -50.0, 50.0).astype(np.float32)])
    d = _decoder(rate)
    r = d.decode_batch(cases, want_total=True)
    ob, oi, ook, ototal = oracle.ldpc_decode_batch(rate, cases, want_total=True)
    assert np.array_equal(r["iters"], oi) and np.array_equal(r["ok"], ook) and np.array_equal(r["bytes"], ob)
    assert beq(r["llr_total"], ototal)
    one = d.decode_batch(cases[5:6])
    assert np.array_equal(one["bytes"], ob[5:6])


@pytest.mark.parametrize("rate", range(6))
def test_nonfinite_llrs(oracle, rate):
    """NaN, +-inf, +-3e38, -0.0 and |LLR| > 50 sprinkled into noisy codewords: the kernel defers the
    +-50 clamp to the reader and takes minima on bit patterns, which must not change any decision
    (the oracle is checked against the compiled reference on the same generator in
    test_oracle_vs_ref.py::test_fec_nonfinite_inputs)."""
    cases = nonfinite_cases(np.random.default_rng(77 + rate), oracle, rate, n=32)
    d = _decoder(rate)
    r = d.decode_batch(cases, want_total=True)
    ob, oi, ook, ototal = oracle.ldpc_decode_batch(rate, cases, want_total=True)
    assert np.array_equal(r["iters"], oi) and np.array_equal(r["ok"], ook) and np.array_equal(r["bytes"], ob)
    got = np.asarray(r["llr_total"])
    both_nan = np.isnan(got) & np.isnan(ototal)          # a fresh NaN's sign bit differs between x86 and gfx950
    assert np.array_equal(np.isnan(got), np.isnan(ototal))
    assert beq(np.where(both_nan, np.float32(0), got), np.where(both_nan, np.float32(0), ototal))
    p = d.decode_batch(cases)                       # the production instance (no a-posteriori LLRs: iteration-0 short cut, other registers)
    assert np.array_equal(p["iters"], oi) and np.array_equal(p["ok"], ook) and np.array_equal(p["bytes"], ob)


@pytest.mark.parametrize("bps", [60, 116, 176])
@pytest.mark.parametrize("rate", [2, 4])
def test_fused_channel_deinterleaver(oracle, rate, bps):
    """ChannelInterleaver(bits_per_symbol, 648)::deinterleave per codeword
    (rx_pipeline.cpp:475-491) fused into the decoder's LLR load == oracle decode of the deinterleaved LLRs.
    The transmit side interleaves, so these codewords decode; plus pure-noise rows."""
    from projectultra_amd import ChannelInterleaver
    llr, _ = noisy_codewords(oracle, rate, 96, [0.45, 0.6] if rate == 2 else [0.3, 0.4], seed=40 + bps)
    il = ChannelInterleaver(bps)
    perm, inv = oracle.channel_interleaver_perm(bps)
    assert np.array_equal(il.permutation, perm) and np.array_equal(il.inverse_permutation, inv)
    tx = np.stack([il.interleave(r) for r in llr])               # what the channel carries
    tx[::7] = np.random.default_rng(bps).normal(0, 3, (tx[::7].shape[0], 648)).astype(np.float32)
    want_in = np.stack([il.deinterleave(r) for r in tx])
    assert beq(want_in[1], llr[1])
    d = _decoder(rate)
    d.setDeinterleave(bps)
    r = d.decode_batch(tx, want_total=True)
    ob, oi, ook, ototal = oracle.ldpc_decode_batch(rate, want_in, want_total=True)
    assert np.array_equal(r["iters"], oi) and np.array_equal(r["ok"], ook) and np.array_equal(r["bytes"], ob)
    assert beq(r["llr_total"], ototal)
    assert ook.mean() > 0.5
    d.setDeinterleave(0)                                           # off again: channel order decodes differently
    r0 = d.decode_batch(tx)
    ob0, oi0, ook0 = oracle.ldpc_decode_batch(rate, tx)
    assert np.array_equal(r0["bytes"], ob0) and np.array_equal(r0["iters"], oi0)


@pytest.mark.parametrize("rate,rows,cols", [(2, 6, 108), (4, 6, 108), (0, 24, 27), (5, 108, 6)])
def test_fused_deinterleave_table(oracle, rate, rows, cols):
    """ultra_hip_set_deinterleave_table: ANY 648-entry permutation fused into the decoder's LLR load — here the reference's
    row-column Interleaver (ldpc_decoder.cpp:454-540; Interleaver(6, 108) is what tools/test_throughput.cpp builds).
    Interleaved noisy codewords decode exactly as the oracle decodes the oracle-deinterleaved LLRs; the table takes
    precedence over the ChannelInterleaver step and switches off again; bad tables are refused."""
    from projectultra_amd import CodeRate, Interleaver, LDPCDecoder
    from projectultra_amd._lib import UltraHipError
    il = Interleaver(rows, cols)
    llr, _ = noisy_codewords(oracle, rate, 300, [0.6, 0.9, 1.3], seed=21)
    tx = np.stack([il.interleave(c) for c in llr])                       # what the channel delivers
    want_in = np.stack([oracle.interleaver_deinterleave(rows, cols, c) for c in tx])
    assert beq(want_in, llr)
    ob, oi, ook = oracle.ldpc_decode_batch(rate, want_in)
    d = LDPCDecoder(CodeRate(rate))
    d.setDeinterleave(60)                                                # the table below wins over the step
    d.setDeinterleaveTable(il.permutation)
    r = d.decode_batch(tx)
    assert np.array_equal(r["bytes"], ob) and np.array_equal(r["iters"], oi) and np.array_equal(r["ok"], ook)
    d.setDeinterleaveTable(None); d.setDeinterleave(0)
    r0 = d.decode_batch(tx)
    ob0, oi0, _ = oracle.ldpc_decode_batch(rate, tx)
    assert np.array_equal(r0["bytes"], ob0) and np.array_equal(r0["iters"], oi0)
    for bad in (np.arange(647), np.full(648, 648), np.arange(648) + 1):
        with pytest.raises(UltraHipError):
            d.context.set_deinterleave_table(bad)


def test_full_size_round_trip_property(oracle):
    """BASELINE cfg4 shape at 2^18 codewords: encode -> BPSK/AWGN -> decode; every frame the decoder
    declares OK must equal its payload or be counted as undetected; high-SNR frames all decode."""
    import torch
    from projectultra_amd import CodeRate, LDPCDecoder
    rate, n_unique, reps = 0, 2048, 128          # 2^18 codewords of R1/4
    llr, payloads = noisy_codewords(oracle, rate, n_unique, [0.8, 1.2, 1.6, 2.0], seed=11)
    d = LDPCDecoder(CodeRate(rate))
    ctx = d.context
    big = torch.from_numpy(llr).cuda().repeat(reps, 1)
    r = ctx.ldpc_decode(big)
    pay = torch.from_numpy(np.concatenate([payloads, np.zeros((n_unique, 1), np.uint8)], 1)[:, :20]).cuda().repeat(reps, 1)
    counters = ctx.count_errors(r, pay)
    ctx.synchronize()
    c = counters.cpu().numpy()
    assert c[0] == n_unique * reps
    ob, oi, ook = oracle.ldpc_decode_batch(rate, llr)
    # replicas are identical inputs -> identical outputs; compare one replica with the oracle, bitwise
    assert np.array_equal(r["bytes"][:n_unique].cpu().numpy(), ob)
    assert np.array_equal(r["bytes"][-n_unique:].cpu().numpy(), ob)
    assert c[4] == int((ook == 0).sum()) * reps and c[5] == int(oi.sum()) * reps
    clean = np.arange(n_unique) % 4 == 0         # sigma 0.8 -> Es/N0 ~ 1.9 dB: R1/4 decodes
    assert ook[clean].mean() > 0.99


@pytest.mark.parametrize("rate,esn0_db", [(0, -11.0), (0, -3.0), (2, 0.5), (3, 2.5), (4, 3.5), (5, 30.0), (1, -1.0)])
def test_llr_stimulus_bitwise_vs_oracle(oracle, rate, esn0_db):
    """ultra_hip_make_llr_batch (BASELINE configs[3] stimulus: random payload -> encode -> BPSK -> AWGN -> 2y/sigma^2,
    on the device) against the oracle's twin, which calls libm: payload bytes and every LLR bit for bit, at a
    ragged batch size and a codeword offset; then the decode of those LLRs, exactly."""
    from projectultra_amd import CodeRate, LDPCDecoder
    ctx = LDPCDecoder(CodeRate(rate)).context
    n, c0, seed = 1500 + 37, (1 << 33) + 12345, 0xABCDEF12345
    llr, payload = ctx.make_llr_batch(n, esn0_db, seed=seed, first_cw=c0)
    r = ctx.ldpc_decode(llr)
    ctx.synchronize()
    want_llr, want_payload = oracle.make_llr_batch(rate, n, esn0_db, seed=seed, c0=c0)
    assert np.array_equal(payload.cpu().numpy(), want_payload)
    got = llr.cpu().numpy()
    assert beq(got, want_llr), np.argwhere(got.view(np.uint32) != want_llr.view(np.uint32))[:5]
    ob, oi, ook = oracle.ldpc_decode_batch(rate, want_llr)
    assert np.array_equal(r["bytes"].cpu().numpy(), ob) and np.array_equal(r["iters"].cpu().numpy(), oi)
    assert np.array_equal(r["ok"].cpu().numpy(), ook)
    # the same codewords in two halves (a sharded sweep generates disjoint index ranges): identical bits
    half = n // 2
    a, _ = ctx.make_llr_batch(half, esn0_db, seed=seed, first_cw=c0)
    b, _ = ctx.make_llr_batch(n - half, esn0_db, seed=seed, first_cw=c0 + half)
    assert beq(np.concatenate([a.cpu().numpy(), b.cpu().numpy()]), want_llr)


def test_llr_stimulus_noise_statistics(oracle):
    """The generator's noise is N(0, sigma^2) with sigma^2 = 1/(2 Es/N0): mean, variance, kurtosis and the tail mass of
    y - x over 2^14 codewords, and independence from the payload (different seeds give different noise)."""
    from projectultra_amd import CodeRate, LDPCDecoder
    ctx = LDPCDecoder(CodeRate.R1_2).context
    n, snr = 1 << 14, 1.0
    llr, payload = ctx.make_llr_batch(n, snr, seed=99)
    ctx.synchronize()
    s2 = 1.0 / (2.0 * 10.0 ** (snr / 10.0))
    y = llr.cpu().numpy().astype(np.float64) * s2 / 2.0
    enc = np.stack([np.unpackbits(np.frombuffer(oracle.ldpc_encode(2, bytes(p)), np.uint8))[:648] for p in payload.cpu().numpy()[:256]])
    z = (y[:256] - (1.0 - 2.0 * enc)) / np.sqrt(s2)
    assert abs(z.mean()) < 0.01 and abs(z.var() - 1.0) < 0.01 and abs(((z - z.mean()) ** 4).mean() / z.var() ** 2 - 3.0) < 0.05
    assert abs((np.abs(z) > 3.0).mean() - 0.0027) < 0.0006
    # whole batch: |y| is +-1 plus noise, so E[y^2] = 1 + sigma^2
    assert abs((y ** 2).mean() - (1.0 + s2)) < 0.003
    other, _ = ctx.make_llr_batch(64, snr, seed=100)
    assert not np.array_equal(other.cpu().numpy(), llr[:64].cpu().numpy())


def test_counters_allreduce_over_rccl():
    """ultra_hip_counters_allreduce on a one-rank RCCL communicator created by the host (ncclCommInitRank through
    ctypes): the plumbing a C++ Monte-Carlo harness uses for the single collective of the path.
    One rank: the sum is the identity; the multi-rank reduction itself is covered on CPU with gloo."""
    import ctypes as C
    import torch
    from projectultra_amd import CodeRate, LDPCDecoder
    try:
        rccl = C.CDLL("librccl.so.1")
    except OSError:
        pytest.skip("librccl.so.1 not loadable")
    uid = (C.c_char * 128)()
    assert rccl.ncclGetUniqueId(uid) == 0

    class Uid(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    u = Uid(); C.memmove(C.byref(u), uid, 128)
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(C.byref(comm), 1, u, 0) == 0
    ctx = LDPCDecoder(CodeRate.R1_2).context
    counters = torch.arange(1, 9, dtype=torch.int64, device="cuda") * 1000003
    want = counters.clone()
    from projectultra_amd._lib import check
    check(ctx.lib.ultra_hip_counters_allreduce(ctx._ctx, comm, counters.data_ptr()), "ultra_hip_counters_allreduce")
    ctx.synchronize()
    assert torch.equal(counters, want)
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)


RCCL_WORKER = r'''
import ctypes as C, os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dist.init_process_group("gloo", rank=rank, world_size=world)          # only to hand the RCCL unique id around
rccl = C.CDLL("librccl.so.1")
class Uid(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]
u = Uid()
if rank == 0:
    assert rccl.ncclGetUniqueId(C.byref(u)) == 0
t = torch.frombuffer(bytearray(bytes(u)), dtype=torch.uint8).clone()
dist.broadcast(t, 0)
C.memmove(C.byref(u), bytes(t.tolist()), 128)
comm = C.c_void_p()
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
assert rccl.ncclCommInitRank(C.byref(comm), world, u, rank) == 0
from projectultra_amd import CodeRate, LDPCDecoder
from projectultra_amd._lib import check
ctx = LDPCDecoder(CodeRate.R1_2).context
counters = (torch.arange(1, 9, dtype=torch.int64, device="cuda") * 1000003) * (rank + 1)
check(ctx.lib.ultra_hip_counters_allreduce(ctx._ctx, comm, counters.data_ptr()), "ultra_hip_counters_allreduce")
ctx.synchronize()
want = torch.arange(1, 9, dtype=torch.int64) * 1000003 * sum(r + 1 for r in range(world))
assert torch.equal(counters.cpu(), want), (counters, want)
rccl.ncclCommDestroy.argtypes = [C.c_void_p]; rccl.ncclCommDestroy(comm)
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_counters_allreduce_over_rccl_two_ranks(tmp_path):
    """ultra_hip_counters_allreduce over a TWO-rank RCCL communicator, one process per GPU — runs where the box shows at
    least two devices (the round's GPU box shows one: skipped there; the 8-GPU scaling run is the driver's)."""
    import os, subprocess, sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    script = tmp_path / "w.py"
    script.write_text(RCCL_WORKER)
    root = str(__import__("pathlib").Path(__file__).resolve().parent.parent)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(30500 + os.getpid() % 1000), str(script), root]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert r.stdout.count("ok") == 2



@pytest.mark.parametrize("rate", RATES)
def test_valid_codewords_with_special_magnitudes_end_at_iteration_0(oracle, rate):
    """Codewords whose channel hard decisions already satisfy every row: the kernels end them before the first
    iteration (bits = the channel decisions, iterations = 0), which must be what the reference's iteration 0 produces
    for ANY magnitudes — zeros, negative zeros and NaNs on bit-0 positions (`x < 0` is false for all three), denormals,
    infinities, 3e38, one flipped-but-still-valid all-zero word — and, one bit away from valid, must not trigger."""
    llr = valid_special_codewords(np.random.default_rng(900 + rate), oracle, rate)
    d = _decoder(rate)
    ob, oi, ook, ototal = oracle.ldpc_decode_batch(rate, llr, want_total=True)
    assert (oi[0::4] == 0).all() and (ook[0::4] == 1).all() and (oi[1::4] == 0).all()      # the reference ends them at 0
    assert (oi[2::4] > 0).any()
    r = d.decode_batch(llr)                                     # the shortcut's path
    assert np.array_equal(r["iters"], oi) and np.array_equal(r["ok"], ook) and np.array_equal(r["bytes"], ob)
    r = d.decode_batch(llr, want_total=True)                    # the full iteration (totals wanted)
    assert np.array_equal(r["iters"], oi) and np.array_equal(r["ok"], ook) and np.array_equal(r["bytes"], ob)
    got = np.asarray(r["llr_total"])
    assert np.array_equal(np.isnan(got), np.isnan(ototal))
    both_nan = np.isnan(got)
    assert beq(np.where(both_nan, np.float32(0), got), np.where(both_nan, np.float32(0), ototal))
