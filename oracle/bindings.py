"""ctypes bindings for the two CHECKERS — test infrastructure, never product code.

  * ``Oracle``  → oracle/libultra_oracle.so  (C restatement, oracle/ultra_oracle.c)
  * ``Ref``     → oracle/_ref/libultra_ref.so (the compiled reference + ref_shim.cpp)

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may
import this module.  The product package (projectultra_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ORACLE_SO = HERE / "libultra_oracle.so"
REF_SO = HERE / "_ref" / "libultra_ref.so"
REFERENCE_ROOT = Path("/root/reference")


class Config(C.Structure):
    """ultra_hip_config (include/ultra_hip.h) — POD shared with the product ABI."""
    _fields_ = [(n, C.c_uint32) for n in (
        "sample_rate", "center_freq", "fft_size", "num_carriers", "cp_mode", "symbol_guard",
        "pilot_spacing", "use_pilots", "modulation", "code_rate", "max_iterations",
        "n_data_symbols", "entry", "training_symbols",
        "adaptive_eq_enabled", "adaptive_eq_use_rls", "decision_directed")] + [("lms_mu", C.c_float), ("rls_lambda", C.c_float), ("sync_threshold", C.c_float)]

    def copy(self, **kw):
        c = Config()
        C.memmove(C.byref(c), C.byref(self), C.sizeof(Config))
        for k, v in kw.items():
            setattr(c, k, v)
        return c


class Geometry(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "cp_len", "symbol_samples", "frame_samples", "n_data_carriers", "n_pilot_carriers",
        "bits_per_carrier", "llrs_per_symbol", "llrs_per_frame", "ldpc_n", "ldpc_k", "ldpc_m",
        "ldpc_edges", "decoded_bytes")]


MOD = dict(DBPSK=0, BPSK=1, DQPSK=2, QPSK=3, D8PSK=4, QAM8=5, QAM16=6, QAM32=7, QAM64=8, QAM256=10)
RATE = dict(R1_4=0, R1_3=1, R1_2=2, R2_3=3, R3_4=4, R5_6=5)
DIFFERENTIAL = (MOD["DBPSK"], MOD["DQPSK"], MOD["D8PSK"])
BITS = {0: 1, 1: 1, 2: 2, 3: 2, 4: 3, 5: 3, 6: 4, 7: 5, 8: 6, 10: 8}
INFO_BITS = {0: 162, 1: 324, 2: 324, 3: 432, 4: 486, 5: 540}


def make_config(fft=1024, mod="QAM16", rate="R3_4", *, carriers=None, pilot_spacing=None,
                use_pilots=None, guard=None, entry=0, training=2, n_data_symbols=None,
                max_iterations=50, cp_mode=1, adaptive_eq=None, decision_directed=True, lms_mu=0.05, rls_lambda=0.99, sync_threshold=0.80):
    """ModemConfig as the reference harnesses build it (tools/test_nvis_mode.cpp:35-41,
    198-212): 512-FFT = ModemConfig defaults (30 carriers, guard 4, spacing 2);
    1024-FFT = presets::nvis_mode() with pilot_spacing 4; use_pilots = !differential."""
    m = MOD[mod] if isinstance(mod, str) else mod
    r = RATE[rate] if isinstance(rate, str) else rate
    c = Config()
    c.sample_rate, c.center_freq, c.fft_size = 48000, 1500, fft
    c.num_carriers = carriers if carriers is not None else (30 if fft == 512 else 59)
    c.cp_mode = cp_mode
    c.symbol_guard = guard if guard is not None else (4 if fft == 512 else 0)
    c.pilot_spacing = pilot_spacing if pilot_spacing is not None else (2 if fft == 512 else 4)
    c.use_pilots = int(m not in DIFFERENTIAL) if use_pilots is None else int(use_pilots)
    c.modulation, c.code_rate, c.max_iterations = m, r, max_iterations
    c.entry, c.training_symbols = entry, (training if entry == 1 else 0)
    if n_data_symbols is None:
        n_pil = -(-c.num_carriers // c.pilot_spacing) if c.use_pilots else 0
        bps = (c.num_carriers - n_pil) * BITS[m]
        n_data_symbols = -(-648 // bps)
    c.n_data_symbols = n_data_symbols
    # ModemConfig::adaptive_eq_* (include/ultra/types.hpp:170-174): adaptive_eq = None (off), "lms" or "rls"
    c.adaptive_eq_enabled, c.adaptive_eq_use_rls = int(adaptive_eq is not None), int(adaptive_eq == "rls")
    c.decision_directed, c.lms_mu, c.rls_lambda = int(bool(decision_directed)), lms_mu, rls_lambda
    c.sync_threshold = sync_threshold               # ModemConfig::sync_threshold (types.hpp:188); 0 would mean the same default
    return c


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t))


def build_oracle(force=False):
    """Compile oracle/libultra_oracle.so (and oracle/_ref when /root/reference exists).  ULTRA_ORACLE_NO_BUILD=1
    (set by bench.py after its own, pre-GPU build step) turns it into a no-op: a process that has initialised the GPU
    must not start the compiler."""
    if os.environ.get("ULTRA_ORACLE_NO_BUILD") == "1" and not force:
        return
    if force or not ORACLE_SO.exists() or ORACLE_SO.stat().st_mtime < (HERE / "ultra_oracle.c").stat().st_mtime:
        subprocess.check_call(["make", "-C", str(HERE), "libultra_oracle.so"], stdout=subprocess.DEVNULL)
    if REFERENCE_ROOT.is_dir() and (force or not REF_SO.exists()
                                    or REF_SO.stat().st_mtime < (HERE / "ref_shim.cpp").stat().st_mtime):
        subprocess.check_call(["make", "-C", str(HERE), "_ref/libultra_ref.so"], stdout=subprocess.DEVNULL)


class _Base:
    prefix = ""

    def __init__(self, path):
        self.lib = C.CDLL(str(path))

    def _fn(self, name):
        return getattr(self.lib, self.prefix + name)

    # -------------------------------------------------------- chirp sync
    def chirp_detect(self, audio, threshold=0.15, sample_rate=48000.0):
        """detectDualChirp + OFDMChirpWaveform::detectSync -> dict(success, up_chirp_start, down_chirp_start,
        start_sample, cfo_hz, up_correlation, down_correlation)."""
        audio = _f32(audio)
        out = (C.c_int32 * 6)(); fout = (C.c_float * 3)()
        rc = self._fn("chirp_detect")(C.c_float(sample_rate), _ptr(audio), C.c_uint32(audio.size), C.c_float(threshold), out, fout)
        assert rc == 0
        return dict(success=out[0], up_chirp_start=out[1], down_chirp_start=out[2], start_sample=out[3],
                    cfo_hz=fout[0], up_correlation=fout[1], down_correlation=fout[2])

    def chirp_generate(self, tx_cfo_hz=0.0, sample_rate=48000.0):
        out = np.zeros(1 << 17, np.float32)
        n = self._fn("chirp_generate")(C.c_float(sample_rate), C.c_float(tx_cfo_hz), _ptr(out), C.c_uint32(out.size))
        assert n > 0
        return out[:n].copy()

    # ------------------------------------------------------- v2 wire format
    FRAME_STATUS = ("cw0_failed", "bad_header", "waiting", "codewords_failed", "complete", "ping")

    def crc16(self, data: bytes) -> int:
        fn = self._fn("crc16"); fn.restype = C.c_uint16
        return int(fn(bytes(data), C.c_uint32(len(data))))

    def v2_parse_header(self, data: bytes):
        out = (C.c_int32 * 4)()
        valid = self._fn("v2_parse_header")(bytes(data), C.c_uint32(len(data)), out)
        return dict(valid=bool(valid), type=out[0], total_cw=out[1], payload_len=out[2], is_control=bool(out[3]))

    def v2_decode_frame(self, rate, soft, deint_bps=0, max_iters=50):
        """RxPipeline::processFrame from the soft bits on -> dict(success, is_ping, frame_type, codewords_ok,
        codewords_failed, expected_codewords, status, frame_data)."""
        soft = _f32(soft)
        res = (C.c_int32 * 8)()
        buf = np.zeros(8192, np.uint8)
        rc = self._fn("v2_decode_frame")(C.c_uint32(int(rate)), C.c_uint32(deint_bps), C.c_int(max_iters), _ptr(soft),
                                         C.c_uint32(soft.size), res, _ptr(buf, C.c_uint8), C.c_uint32(buf.size))
        assert rc == 0, rc
        return dict(success=res[0], is_ping=res[1], frame_type=res[2], codewords_ok=res[3], codewords_failed=res[4],
                    expected_codewords=res[5], status=res[7], frame_data=bytes(buf[:res[6]]))

    def v2_build_frame(self, rate, payload=b"", type=0x30, flags=0x01, seq=0, src_hash=0x123456, dst_hash=0xABCDEF,
                       total_cw=-1):
        """DataFrame/ControlFrame::serialize + encodeFrameWithLDPC -> [n_cw][81] encoded codewords."""
        out = np.zeros((256, 81), np.uint8)
        n = self._fn("v2_build_frame")(C.c_uint32(int(rate)), C.c_uint8(type), C.c_uint8(flags), C.c_uint16(seq),
                                       C.c_uint32(src_hash), C.c_uint32(dst_hash), bytes(payload), C.c_uint32(len(payload)),
                                       C.c_int(total_cw), _ptr(out, C.c_uint8), C.c_uint32(256))
        assert n > 0, n
        return out[:n].copy()

    def chirp_templates(self, sample_rate=48000.0):
        cap = 1 << 16
        t = [np.zeros(cap, np.float32) for _ in range(4)]
        e = np.zeros(2, np.float32)
        m = self._fn("chirp_templates")(C.c_float(sample_rate), _ptr(t[0]), _ptr(t[1]), _ptr(t[2]), _ptr(t[3]), _ptr(e), C.c_uint32(cap))
        assert m > 0
        return [x[:m].copy() for x in t], e

    # ------------------------------------------------------- acquisition
    def acquire(self, cfg, audio, chunk=960):
        """SEARCHING state fed in chunk-sample calls -> dict(found, fed_at_sync, sync_offset, coarse_cfo,
        refined_lts, data_start, noise_floor)."""
        audio = _f32(audio)
        u = [C.c_uint32(0) for _ in range(5)]
        cfo, nf = C.c_float(0), C.c_float(0)
        name = "acquire" if self.prefix == "uo_" else "demod_acquire"
        rc = self._fn(name)(C.byref(cfg), _ptr(audio), C.c_uint32(audio.size), C.c_uint32(chunk), C.byref(u[0]),
                            C.byref(u[1]), C.byref(u[2]), C.byref(cfo), C.byref(u[3]), C.byref(u[4]), C.byref(nf))
        assert rc == 0, rc
        return dict(found=u[0].value, fed_at_sync=u[1].value, sync_offset=u[2].value, coarse_cfo=cfo.value,
                    refined_lts=u[3].value, data_start=u[4].value, noise_floor=nf.value)

    def midframe_search(self, cfg, audio):
        """The preamble check of the SYNCED state (demodulator.cpp:605-657) on rx_buffer = audio ->
        dict(found, sts_start, refined_lts, consume, coarse_cfo)."""
        audio = _f32(audio)
        u = [C.c_uint32(0) for _ in range(4)]
        cfo = C.c_float(0)
        rc = self._fn("midframe_search")(C.byref(cfg), _ptr(audio), C.c_uint32(audio.size), C.byref(u[0]), C.byref(u[1]),
                                         C.byref(u[2]), C.byref(u[3]), C.byref(cfo))
        assert rc == 0, rc
        return dict(found=u[0].value, sts_start=u[1].value, refined_lts=u[2].value, consume=u[3].value, coarse_cfo=cfo.value)

    def sc_metric(self, cfg, audio, offset, noise_floor=0.0):
        """One Schmidl-Cox metric + energy gate -> (corr, P.re, P.im, energy, noise_floor_after, has_energy)."""
        audio = _f32(audio)
        f = [C.c_float(0) for _ in range(4)]
        nf = C.c_float(noise_floor); he = C.c_uint32(0)
        rc = self._fn("sc_metric")(C.byref(cfg), _ptr(audio), C.c_uint32(audio.size), C.c_uint32(offset), C.byref(f[0]),
                                   C.byref(f[1]), C.byref(f[2]), C.byref(f[3]), C.byref(nf), C.byref(he))
        assert rc == 0
        return np.array([x.value for x in f] + [nf.value], np.float32), he.value

    def lts_templates(self, cfg):
        I = np.zeros(2048, np.float32); Q = np.zeros(2048, np.float32)
        m = self._fn("lts_templates")(C.byref(cfg), _ptr(I), _ptr(Q), C.c_uint32(2048))
        assert m > 0
        return I[:m].copy(), Q[:m].copy()

    # -------------------------------------------------------------- FEC
    def ldpc_encode(self, rate, data: bytes) -> bytes:
        out = (C.c_uint8 * 4096)()
        n = self._fn("ldpc_encode")(C.c_uint32(rate), data, C.c_uint32(len(data)), out, C.c_uint32(4096))
        assert n >= 0
        return bytes(out[:n])

    def ldpc_decode_soft(self, rate, llr, max_iters=50):
        llr = _f32(llr)
        out = (C.c_uint8 * 4096)()
        ok, it = C.c_int(0), C.c_int(0)
        n = self._fn("ldpc_decode_soft")(C.c_uint32(rate), C.c_int(max_iters), _ptr(llr), C.c_uint32(llr.size),
                                         out, C.c_uint32(4096), C.byref(ok), C.byref(it))
        assert n >= 0
        return bytes(out[:n]), bool(ok.value), it.value

    def ldpc_decode_batch(self, rate, llr, max_iters=50, want_total=False):
        llr = _f32(llr).reshape(-1, 648)
        n = llr.shape[0]
        nbytes = (INFO_BITS[rate] + 7) // 8
        out = np.zeros((n, nbytes), np.uint8)
        iters = np.zeros(n, np.int32)
        ok = np.zeros(n, np.uint8)
        args = [C.c_uint32(rate), C.c_int(max_iters), _ptr(llr), C.c_uint32(n), _ptr(out, C.c_uint8),
                C.c_uint32(nbytes), _ptr(iters, C.c_int32), _ptr(ok, C.c_uint8)]
        total = None
        if self.prefix == "uo_":
            total = np.zeros((n, 648), np.float32) if want_total else None
            args.append(_ptr(total) if want_total else None)
        rc = self._fn("ldpc_decode_batch")(*args)
        assert rc == 0
        return (out, iters, ok, total) if want_total else (out, iters, ok)

    def ldpc_decode_batch_mt(self, rate, llr, n_threads, max_iters=50):
        """uo_ldpc_decode_batch over worker threads (the timed CPU leg of the LDPC-only bench)."""
        llr = _f32(llr).reshape(-1, 648)
        n = llr.shape[0]
        nbytes = (INFO_BITS[rate] + 7) // 8
        out = np.zeros((n, nbytes), np.uint8)
        iters = np.zeros(n, np.int32)
        ok = np.zeros(n, np.uint8)
        rc = self._fn("ldpc_decode_batch_mt")(C.c_uint32(rate), C.c_int(max_iters), _ptr(llr), C.c_uint32(n), C.c_int(n_threads),
                                              _ptr(out, C.c_uint8), C.c_uint32(nbytes), _ptr(iters, C.c_int32), _ptr(ok, C.c_uint8))
        assert rc == 0
        return out, iters, ok

    def make_llr_batch(self, rate, n, esn0_db, seed=0x5EED, c0=0):
        """BPSK-over-AWGN LLRs of codewords c0 .. c0+n-1 (twin of ultra_hip_make_llr_batch) -> (llr [n][648], payload)."""
        llr = np.zeros((n, 648), np.float32)
        payload = np.zeros((n, INFO_BITS[rate] // 8), np.uint8)
        rc = self._fn("make_llr_batch")(C.c_uint32(rate), C.c_uint64(seed), C.c_uint64(c0), C.c_uint32(n), C.c_float(esn0_db),
                                        _ptr(llr), _ptr(payload, C.c_uint8))
        assert rc == 0, rc
        return llr, payload

    # -------------------------------------------------------------- DSP
    def fft_forward(self, x):
        x = np.ascontiguousarray(x, np.complex64)
        out = np.zeros_like(x)
        self._fn("fft_forward")(C.c_uint32(x.size), _ptr(x.view(np.float32)), _ptr(out.view(np.float32)))
        return out

    def fft_inverse(self, x):
        x = np.ascontiguousarray(x, np.complex64)
        out = np.zeros_like(x)
        self._fn("fft_inverse")(C.c_uint32(x.size), _ptr(x.view(np.float32)), _ptr(out.view(np.float32)))
        return out

    def nco(self, freq, fs, n):
        out = np.zeros(n, np.complex64)
        self._fn("nco")(C.c_float(freq), C.c_float(fs), C.c_uint32(n), _ptr(out.view(np.float32)))
        return out

    # ------------------------------------------------------ demodulator
    def demod_tables(self, cfg):
        di = np.zeros(128, np.int32); pi = np.zeros(128, np.int32)
        ps = np.zeros(128, np.complex64); ii = np.zeros((128, 3), np.int32)
        ia = np.zeros(128, np.float32); ss = np.zeros(128, np.complex64)
        cnt = np.zeros(4, np.uint32)
        rc = self._fn("demod_tables")(C.byref(cfg), _ptr(di, C.c_int32), _ptr(pi, C.c_int32),
                                      _ptr(ps.view(np.float32)), _ptr(ii, C.c_int32), _ptr(ia),
                                      _ptr(ss.view(np.float32)), _ptr(cnt, C.c_uint32))
        assert rc == 0
        nd, np_, ni, ns = (int(v) for v in cnt)
        return dict(data_idx=di[:nd].copy(), pilot_idx=pi[:np_].copy(), pilot_seq=ps[:np_].copy(),
                    interp=ii[:ni].copy(), interp_alpha=ia[:ni].copy(), sync_seq=ss[:ns].copy())

    def demod_synced(self, cfg, audio, cfo_hz=0.0, stages=False):
        """SYNCED-entry symbol loop.  Returns (llr, stage dict | None)."""
        audio = _f32(audio)
        g = geometry(cfg)
        S, N, nd = g.symbol_samples, cfg.fft_size, g.n_data_carriers
        nsym = audio.size // S
        cap = nsym * g.llrs_per_symbol
        llr = np.zeros(cap, np.float32)
        per = 2 * S + 2 * N + 2 * N + 2 * nd + nd + 8
        st = np.zeros(nsym * per, np.float32) if stages else None
        n = self._fn("demod_synced")(C.byref(cfg), _ptr(audio), C.c_uint32(nsym), C.c_float(cfo_hz),
                                     _ptr(llr), C.c_uint32(cap), _ptr(st) if stages else None)
        assert n == cap, (n, cap)
        if not stages:
            return llr, None
        st = st.reshape(nsym, per)
        o = 0
        d = {}
        for name, ln, cx in (("bb", S, True), ("freq", N, True), ("H", N, True), ("eq", nd, True),
                             ("nv", nd, False), ("scal", 8, False)):
            w = 2 * ln if cx else ln
            blk = np.ascontiguousarray(st[:, o:o + w])
            d[name] = blk.view(np.complex64) if cx else blk
            o += w
        return llr, d

    def demod_presynced(self, cfg, audio, cfo_hz=0.0, cfo_phase=0.0):
        """processPresynced after setFrequencyOffsetWithPhase(cfo_hz, cfo_phase); cfo_hz = None: the offset was never
        set (training-symbol estimate, ofdm_sync.cpp:278-380)."""
        audio = _f32(audio)
        has_cfo = cfo_hz is not None
        cfo_hz = cfo_hz if has_cfo else 0.0
        g = geometry(cfg)
        cap = g.llrs_per_frame + 4096
        llr = np.zeros(cap, np.float32)
        H = np.zeros(cfg.fft_size, np.complex64)
        scal = np.zeros(8, np.float32)
        n = self._fn("demod_presynced")(C.byref(cfg), _ptr(audio), C.c_uint32(audio.size), C.c_int(1 if has_cfo else 0),
                                        C.c_float(cfo_hz), C.c_float(cfo_phase), _ptr(llr), C.c_uint32(cap),
                                        _ptr(H.view(np.float32)), _ptr(scal))
        assert n >= 0
        return llr[:n].copy(), H, scal

    # -------------------------------------------------------- modulator
    def modulate_frame(self, cfg, encoded: bytes):
        cap = 1 << 20
        out = np.zeros(cap, np.float32)
        pre = C.c_uint32(0)
        n = self._fn("modulate_frame")(C.byref(cfg), encoded, C.c_uint32(len(encoded)), _ptr(out),
                                       C.c_uint32(cap), C.byref(pre))
        assert n > 0
        return out[:n].copy(), pre.value

    def modulate_presynced(self, cfg, encoded: bytes):
        cap = 1 << 20
        out = np.zeros(cap, np.float32)
        n = self._fn("modulate_presynced")(C.byref(cfg), encoded, C.c_uint32(len(encoded)), _ptr(out), C.c_uint32(cap))
        assert n > 0
        return out[:n].copy()


class Oracle(_Base):
    prefix = "uo_"

    def __init__(self):
        build_oracle()
        super().__init__(ORACLE_SO)
        self.lib.uo_geometry.argtypes = [C.POINTER(Config), C.POINTER(Geometry)]

    def ldpc_graph(self, rate):
        rp = np.zeros(487, np.uint32); ci = np.zeros(4096, np.uint32)
        k, m = C.c_uint32(0), C.c_uint32(0)
        e = self.lib.uo_ldpc_graph(C.c_uint32(rate), _ptr(rp, C.c_uint32), _ptr(ci, C.c_uint32), C.byref(k), C.byref(m))
        return rp[:m.value + 1].copy(), ci[:e].copy(), k.value, m.value

    def channel_interleaver_perm(self, bits_per_symbol, total=648):
        p = np.zeros(total, np.uint32); inv = np.zeros(total, np.uint32)
        self.lib.uo_channel_interleaver_perm(C.c_uint32(bits_per_symbol), C.c_uint32(total),
                                             _ptr(p, C.c_uint32), _ptr(inv, C.c_uint32))
        return p, inv

    def interleaver_deinterleave(self, rows, cols, x):
        x = _f32(x); out = np.zeros_like(x)
        self.lib.uo_interleaver_deinterleave(C.c_uint32(rows), C.c_uint32(cols), _ptr(x), C.c_uint32(x.size), _ptr(out))
        return out

    def demod_decode_batch(self, cfg, audio, cfo_hz=None, cfo_phase=None, n_threads=1, decode=True,
                           want_llr=True, want_state=True):
        audio = _f32(audio)
        g = geometry(cfg)
        audio = audio.reshape(-1, audio.shape[-1]) if audio.ndim > 1 else audio.reshape(-1, g.frame_samples)
        n, stride = audio.shape
        llr = np.zeros((n, g.llrs_per_frame), np.float32) if want_llr else None
        state = np.zeros((n, 8), np.float32) if want_state else None
        by = np.zeros((n, g.decoded_bytes), np.uint8) if decode else None
        it = np.zeros(n, np.int32) if decode else None
        ok = np.zeros(n, np.uint8) if decode else None
        cfo = _f32(cfo_hz) if cfo_hz is not None else None
        cph = _f32(cfo_phase) if cfo_phase is not None else None
        rc = self.lib.uo_demod_decode_batch(
            C.byref(cfg), _ptr(audio), C.c_size_t(stride), _ptr(cfo) if cfo is not None else None,
            _ptr(cph) if cph is not None else None, C.c_size_t(n), C.c_int(n_threads),
            _ptr(llr) if want_llr else None, _ptr(state) if want_state else None,
            _ptr(by, C.c_uint8) if decode else None, _ptr(it, C.c_int32) if decode else None,
            _ptr(ok, C.c_uint8) if decode else None)
        assert rc == 0
        return dict(llr=llr, state=state, bytes=by, iters=it, ok=ok)

    def channel_apply_cfo(self, x, cfo_hz, sample_rate=48000):
        """WattersonChannel::applyCFO of a fresh channel (hf_channel.hpp:161-232) -> shifted copy."""
        x = _f32(x).copy()
        assert self.lib.uo_channel_apply_cfo(C.c_float(cfo_hz), C.c_uint32(sample_rate), _ptr(x), C.c_uint32(x.size)) == 0
        return x

    def watterson(self, x, snr_db, delay_ms, doppler_hz, seed, g1=0.707, g2=0.707, fading=1, multipath=1, noise=1):
        x = _f32(x); out = np.zeros_like(x)
        self.lib.uo_watterson(C.c_float(snr_db), C.c_float(delay_ms), C.c_float(doppler_hz), C.c_float(g1),
                              C.c_float(g2), C.c_int(fading), C.c_int(multipath), C.c_int(noise),
                              C.c_uint64(seed), _ptr(x), C.c_uint32(x.size), _ptr(out))
        return out

    def make_batch(self, cfg, n, seed=0x5EED, f0=0, n_threads=None, channel="awgn", snr_db=30.0,
                   delay_ms=0.5, doppler_hz=0.1):
        """Synthetic frames at the configured entry point → (audio [n][frame_samples], payload [n][k//8])."""
        g = geometry(cfg)
        pb = g.ldpc_k // 8
        audio = np.zeros((n, g.frame_samples), np.float32)
        payload = np.zeros((n, pb), np.uint8)
        kind = dict(none=0, awgn=1, watterson=2)[channel]
        nt = n_threads or min(os.cpu_count() or 1, 16)
        rc = self.lib.uo_make_batch(C.byref(cfg), C.c_uint64(seed), C.c_uint64(f0), C.c_uint32(n), C.c_int(nt),
                                    C.c_int(kind), C.c_float(snr_db), C.c_float(delay_ms), C.c_float(doppler_hz),
                                    _ptr(audio), _ptr(payload, C.c_uint8), C.c_uint32(pb))
        assert rc == 0, rc
        return audio, payload


class Ref(_Base):
    prefix = "ref_"

    def __init__(self):
        build_oracle()
        if not REF_SO.exists():
            raise FileNotFoundError(f"{REF_SO} not built (needs /root/reference in this container)")
        super().__init__(REF_SO)

    def channel_apply_cfo(self, x, cfo_hz, sample_rate=48000):
        x = _f32(x).copy()
        assert self.lib.ref_channel_apply_cfo(C.c_float(cfo_hz), _ptr(x), C.c_uint32(x.size)) == 0
        return x

    def watterson(self, x, snr_db, delay_ms, doppler_hz, seed, g1=0.707, g2=0.707, fading=1, multipath=1, noise=1):
        x = _f32(x); out = np.zeros_like(x)
        self.lib.ref_watterson(C.c_float(snr_db), C.c_float(delay_ms), C.c_float(doppler_hz), C.c_float(g1),
                               C.c_float(g2), C.c_int(fading), C.c_int(multipath), C.c_int(noise),
                               C.c_uint32(seed), _ptr(x), C.c_uint32(x.size), _ptr(out))
        return out

    def demod_process(self, cfg, audio, chunk=960):
        """Full reference receive incl. Schmidl-Cox search → (llr, sync_offset, cfo_hz)."""
        audio = _f32(audio)
        cap = 1 << 16
        llr = np.zeros(cap, np.float32)
        so, cfo = C.c_uint32(0), C.c_float(0)
        n = self.lib.ref_demod_process(C.byref(cfg), _ptr(audio), C.c_uint32(audio.size), C.c_uint32(chunk),
                                       _ptr(llr), C.c_uint32(cap), C.byref(so), C.byref(cfo))
        return llr[:n].copy(), so.value, cfo.value

    def demod_process_coarse(self, cfg, audio, chunk=960):
        """Full reference receive → (llr, sync_offset, coarse_cfo, final_cfo, fed_at_sync)."""
        audio = _f32(audio)
        cap = 1 << 16
        llr = np.zeros(cap, np.float32)
        so, fs = C.c_uint32(0), C.c_uint32(0)
        cc, fc = C.c_float(0), C.c_float(0)
        n = self.lib.ref_demod_process_coarse(C.byref(cfg), _ptr(audio), C.c_uint32(audio.size), C.c_uint32(chunk),
                                              _ptr(llr), C.c_uint32(cap), C.byref(so), C.byref(cc), C.byref(fc),
                                              C.byref(fs))
        return llr[:n].copy(), so.value, cc.value, fc.value, fs.value

    def demod_stream(self, cfg, audio, chunks):
        """A live stream through OFDMDemodulator::process + getSoftBits, one call per entry of `chunks` (sample counts,
        0 = an empty call) → (ready [n_calls] u8, synced [n_calls] u8, drained [n_calls] u32, soft bits concatenated)."""
        audio = _f32(audio)
        chunks = np.ascontiguousarray(chunks, np.uint32)
        assert int(chunks.sum()) <= audio.size
        n = chunks.size
        ready = np.zeros(n, np.uint8); synced = np.zeros(n, np.uint8); drained = np.zeros(n, np.uint32)
        cap = 1 << 18
        soft = np.zeros(cap, np.float32)
        total = self.lib.ref_demod_stream(C.byref(cfg), _ptr(audio), _ptr(chunks, C.c_uint32), C.c_uint32(n), _ptr(ready, C.c_uint8),
                                          _ptr(synced, C.c_uint8), _ptr(drained, C.c_uint32), _ptr(soft), C.c_uint32(cap))
        assert 0 <= total <= cap
        return ready, synced, drained, soft[:total].copy()

    def demod_decode_batch(self, cfg, audio, cfo_hz=None):
        """The reference's own classes over SYNCED-entry frames, one thread → dict(bytes, iters, ok)."""
        audio = _f32(audio)
        g = geometry(cfg)
        n, stride = audio.shape
        by = np.zeros((n, g.decoded_bytes), np.uint8); it = np.zeros(n, np.int32); ok = np.zeros(n, np.uint8)
        cfo = _f32(cfo_hz) if cfo_hz is not None else None
        rc = self.lib.ref_demod_decode_batch(C.byref(cfg), _ptr(audio), C.c_size_t(stride),
                                             _ptr(cfo) if cfo is not None else None, C.c_uint32(n),
                                             _ptr(by, C.c_uint8), C.c_uint32(g.decoded_bytes), _ptr(it, C.c_int32),
                                             _ptr(ok, C.c_uint8))
        assert rc == 0
        return dict(bytes=by, iters=it, ok=ok)

    def demod_decode_batch_mt(self, cfg, audio, n_threads, cfo_hz=None):
        """The same over worker threads (one OFDMDemodulator / LDPCDecoder per thread): the reference's CPU path on all
        host cores."""
        audio = _f32(audio)
        g = geometry(cfg)
        n, stride = audio.shape
        by = np.zeros((n, g.decoded_bytes), np.uint8); it = np.zeros(n, np.int32); ok = np.zeros(n, np.uint8)
        cfo = _f32(cfo_hz) if cfo_hz is not None else None
        rc = self.lib.ref_demod_decode_batch_mt(C.byref(cfg), _ptr(audio), C.c_size_t(stride),
                                                _ptr(cfo) if cfo is not None else None, C.c_uint32(n), C.c_int(n_threads),
                                                _ptr(by, C.c_uint8), C.c_uint32(g.decoded_bytes), _ptr(it, C.c_int32),
                                                _ptr(ok, C.c_uint8))
        assert rc == 0
        return dict(bytes=by, iters=it, ok=ok)

    def receive_batch_mt(self, cfg, audio, n_threads, chunk=960):
        """Raw streams [n][n_samples] -> the reference's whole receive per stream (fresh demodulator, `chunk` samples per
        process() call, decodeSoft of the first 648 soft bits) -> dict(bytes, iters, ok, found, sync_offset)."""
        audio = _f32(audio)
        g = geometry(cfg)
        n, stride = audio.shape
        by = np.zeros((n, g.decoded_bytes), np.uint8); it = np.zeros(n, np.int32); ok = np.zeros(n, np.uint8)
        found = np.zeros(n, np.uint8); so = np.zeros(n, np.uint32)
        rc = self.lib.ref_receive_batch_mt(C.byref(cfg), _ptr(audio), C.c_size_t(stride), C.c_uint32(stride), C.c_uint32(chunk),
                                           C.c_uint32(n), C.c_int(n_threads), _ptr(by, C.c_uint8), C.c_uint32(g.decoded_bytes),
                                           _ptr(it, C.c_int32), _ptr(ok, C.c_uint8), _ptr(found, C.c_uint8), _ptr(so, C.c_uint32))
        assert rc == 0
        return dict(bytes=by, iters=it, ok=ok, found=found, sync_offset=so)

    def demod_synced_public(self, cfg, audio, cfo_hz=0.0):
        audio = _f32(audio)
        g = geometry(cfg)
        nsym = audio.size // g.symbol_samples
        cap = nsym * g.llrs_per_symbol
        llr = np.zeros(cap, np.float32)
        n = self.lib.ref_demod_synced_public(C.byref(cfg), _ptr(audio), C.c_uint32(nsym), C.c_float(cfo_hz),
                                             _ptr(llr), C.c_uint32(cap))
        assert n == cap
        return llr

    def demod_synced_setcfo(self, cfg, audio, set_at, cfo_new_hz, cfo0_hz=None):
        """SYNCED-entry frame whose offset is replaced by setFrequencyOffset before symbol `set_at` -> LLRs."""
        audio = _f32(audio)
        g = geometry(cfg)
        nsym = audio.size // g.symbol_samples
        cap = nsym * g.llrs_per_symbol
        llr = np.zeros(cap, np.float32)
        n = self.lib.ref_demod_synced_setcfo(C.byref(cfg), _ptr(audio), C.c_uint32(nsym), C.c_int(cfo0_hz is not None),
                                             C.c_float(cfo0_hz or 0.0), C.c_uint32(set_at), C.c_float(cfo_new_hz),
                                             _ptr(llr), C.c_uint32(cap))
        assert n == cap, (n, cap)
        return llr

    def harness_awgn(self, cfg, payload: bytes, snr_db, noise_seed):
        cap = 1 << 20
        out = np.zeros(cap, np.float32)
        n = C.c_uint32(0)
        pre = self.lib.ref_harness_awgn(C.byref(cfg), payload, C.c_uint32(len(payload)), C.c_float(snr_db),
                                        C.c_uint32(noise_seed), _ptr(out), C.c_uint32(cap), C.byref(n))
        assert pre > 0
        return out[:n.value].copy(), pre

    def channel_interleaver(self, bits_per_symbol, x, total=648, inverse=True):
        x = _f32(x); out = np.zeros(total, np.float32)
        fn = self.lib.ref_channel_interleaver_deinterleave if inverse else self.lib.ref_channel_interleaver_interleave
        fn(C.c_uint32(bits_per_symbol), C.c_uint32(total), _ptr(x), C.c_uint32(x.size), _ptr(out))
        return out

    def channel_interleaver_perm(self, bits_per_symbol, total=648):
        """(perm, inv) read off the reference's ChannelInterleaver::interleave: out[perm[i]] = in[i]."""
        out = self.channel_interleaver(bits_per_symbol, np.arange(total, dtype=np.float32), total, inverse=False)
        inv = out.astype(np.uint32)
        perm = np.empty(total, np.uint32); perm[inv] = np.arange(total, dtype=np.uint32)
        return perm, inv

    def interleaver_deinterleave(self, rows, cols, x):
        x = _f32(x); out = np.zeros_like(x)
        self.lib.ref_interleaver_deinterleave(C.c_uint32(rows), C.c_uint32(cols), _ptr(x), C.c_uint32(x.size), _ptr(out))
        return out


_ORACLE = None


def oracle() -> Oracle:
    global _ORACLE
    if _ORACLE is None:
        _ORACLE = Oracle()
    return _ORACLE


def geometry(cfg) -> Geometry:
    g = Geometry()
    rc = oracle().lib.uo_geometry(C.byref(cfg), C.byref(g))
    assert rc == 0, rc
    return g


def have_ref() -> bool:
    if REF_SO.exists():
        return True
    if REFERENCE_ROOT.is_dir():
        try:
            build_oracle()
        except Exception:
            return False
    return REF_SO.exists()
