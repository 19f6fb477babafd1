"""The header-only C++ adapter (include/ultra_hip_waveform.hpp) RUN on the GPU: a small program, compiled
on the box with g++ against libultra_hip.so, receives the whole frames of tests/golden/fullsync.npz the
way a reference harness would (960-sample chunks -> process -> getSoftBits -> decodeSoft) and its
outputs are compared with what the compiled reference produced."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

from _util import beq, cfg_from_array
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

SRC = r'''
#include "ultra_hip_waveform.hpp"
#include <cstdio>
#include <vector>
using namespace ultra_hip;
int main(int argc, char** argv) {
    // argv: in.f32 n_samples out_llr.f32 out_bytes.bin fft carriers cp_mode guard pilot_spacing use_pilots mod rate n_data
    FILE* f = std::fopen(argv[1], "rb");
    const size_t n = std::stoul(argv[2]);
    std::vector<float> audio(n);
    if (std::fread(audio.data(), 4, n, f) != n) return 2;
    std::fclose(f);
    ModemConfig c;
    c.fft_size = std::stoul(argv[5]); c.num_carriers = std::stoul(argv[6]);
    c.cp_mode = static_cast<decltype(c.cp_mode)>(std::stoi(argv[7])); c.symbol_guard = std::stoul(argv[8]);
    c.pilot_spacing = std::stoul(argv[9]); c.use_pilots = std::stoi(argv[10]) != 0;
    c.modulation = static_cast<Modulation>(std::stoi(argv[11])); c.code_rate = static_cast<CodeRate>(std::stoi(argv[12]));
    HipOfdmCoxWaveform rx(c);                                   // no frame length: symbols are demodulated as they arrive
    bool ready = false;
    for (size_t i = 0; i < n; i += 960) {                       // tools/test_nvis_mode.cpp:88-99
        const size_t len = std::min<size_t>(960, n - i);
        ready = rx.process(SampleSpan(audio.data() + i, len));
    }
    if (!ready) return 3;
    std::vector<float> soft = rx.getSoftBits();                 // first 648
    HipLDPCDecoder dec(c.code_rate);
    Bytes out = dec.decodeSoft(std::span<const float>(soft.data(), soft.size()));
    FILE* g = std::fopen(argv[3], "wb"); std::fwrite(soft.data(), 4, soft.size(), g); std::fclose(g);
    g = std::fopen(argv[4], "wb"); std::fwrite(out.data(), 1, out.size(), g);
    const int meta[3] = {dec.lastDecodeSuccess() ? 1 : 0, dec.lastIterations(), (int)rx.getLastSyncOffset()};
    std::fwrite(meta, 4, 3, g); std::fclose(g);
    float cfo = rx.coarseCFO(); g = std::fopen(argv[3], "ab"); std::fwrite(&cfo, 4, 1, g); std::fclose(g);
    return 0;
}
'''


@pytest.mark.parametrize("name", ["cfg3_qam16_r34", "cfg2_dqpsk_r12"])
def test_cpp_adapter_receives_reference_frames(tmp_path, oracle, name):
    g = np.load(GOLDEN / "fullsync.npz")
    cfg = cfg_from_array(g[f"{name}__cfg"])
    src = tmp_path / "rx.cpp"
    src.write_text(SRC)
    exe = tmp_path / "rx"
    lib = ROOT / "projectultra_amd"
    subprocess.check_call(["g++", "-O1", "-std=c++20", f"-I{ROOT / 'include'}", str(src), f"-L{lib}", "-lultra_hip",
                           f"-Wl,-rpath,{lib}", "-o", str(exe)])
    for t, (a, meta, cfo, want) in enumerate(zip(g[f"{name}__audio"], g[f"{name}__meta"], g[f"{name}__cfo"], g[f"{name}__llr"])):
        fin, fl, fb = tmp_path / f"in{t}.f32", tmp_path / f"llr{t}.f32", tmp_path / f"bytes{t}.bin"
        a.astype(np.float32).tofile(fin)
        args = [str(exe), str(fin), str(a.size), str(fl), str(fb)] + [str(int(x)) for x in (
            cfg.fft_size, cfg.num_carriers, cfg.cp_mode, cfg.symbol_guard, cfg.pilot_spacing, cfg.use_pilots,
            cfg.modulation, cfg.code_rate, cfg.n_data_symbols)]
        r = subprocess.run(args, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.returncode, r.stderr[-400:])
        got = np.fromfile(fl, np.float32)
        assert beq(got[:648], want[:648]), name
        assert np.float32(got[648]).tobytes() == np.float32(cfo).tobytes()
        raw = np.fromfile(fb, np.uint8)
        nbytes = raw.size - 12
        ok, iters, sync_off = np.frombuffer(raw[nbytes:].tobytes(), np.int32)
        ob, oi, ook = oracle.ldpc_decode_batch(int(cfg.code_rate), want[:648].reshape(1, 648))
        assert np.array_equal(raw[:nbytes], ob[0]) and ok == ook[0] and iters == oi[0]
        assert sync_off == meta[1]


SRC_CHIRP = r'''
#include "ultra_hip_waveform.hpp"
#include <cstdio>
#include <vector>
using namespace ultra_hip;
int main(int argc, char** argv) {
    // argv: in.f32 n_samples out.f32 fft carriers cp_mode guard pilot_spacing use_pilots mod rate
    FILE* f = std::fopen(argv[1], "rb");
    const size_t n = std::stoul(argv[2]);
    std::vector<float> audio(n);
    if (std::fread(audio.data(), 4, n, f) != n) return 2;
    std::fclose(f);
    ModemConfig c;
    c.fft_size = std::stoul(argv[4]); c.num_carriers = std::stoul(argv[5]);
    c.cp_mode = static_cast<decltype(c.cp_mode)>(std::stoi(argv[6])); c.symbol_guard = std::stoul(argv[7]);
    c.pilot_spacing = std::stoul(argv[8]); c.use_pilots = std::stoi(argv[9]) != 0;
    c.modulation = static_cast<Modulation>(std::stoi(argv[10])); c.code_rate = static_cast<CodeRate>(std::stoi(argv[11]));
    HipOfdmWaveform w(c);
    SyncResult r;
    std::vector<float> soft;
    if (w.detectSync(SampleSpan(audio.data(), n), r, 0.15f)) {            // the harness flow: tools/test_nvis_mode.cpp
        w.setFrequencyOffset(r.cfo_hz);
        if (!w.process(SampleSpan(audio.data() + r.start_sample, n - r.start_sample))) return 3;
        soft = w.getSoftBits();
    }
    FILE* g = std::fopen(argv[3], "wb");
    const float head[4] = {r.detected ? 1.0f : 0.0f, (float)r.start_sample, r.cfo_hz, r.correlation};
    std::fwrite(head, 4, 4, g); std::fwrite(soft.data(), 4, soft.size(), g); std::fclose(g);
    return 0;
}
'''


def test_cpp_adapter_chirp_waveform(tmp_path, oracle):
    """HipOfdmWaveform::detectSync (GPU dual-chirp detection) -> setFrequencyOffset -> process -> getSoftBits from a
    C++ program, against the oracle's detection and PRESYNCED demodulation of the same samples."""
    from _util import chirp_initial_phase, chirp_streams, make_config
    cfg = make_config(512, "DQPSK", "R1_2", entry=1)
    src = tmp_path / "rxc.cpp"
    src.write_text(SRC_CHIRP)
    exe = tmp_path / "rxc"
    lib = ROOT / "projectultra_amd"
    subprocess.check_call(["g++", "-O1", "-std=c++20", f"-I{ROOT / 'include'}", str(src), f"-L{lib}", "-lultra_hip",
                           f"-Wl,-rpath,{lib}", "-o", str(exe)])
    streams = chirp_streams(oracle, cfg, np.random.default_rng(33), n=2)
    for t, x in enumerate(streams):
        fin, fo = tmp_path / f"c{t}.f32", tmp_path / f"o{t}.f32"
        x.tofile(fin)
        args = [str(exe), str(fin), str(x.size), str(fo)] + [str(int(v)) for v in (
            cfg.fft_size, cfg.num_carriers, cfg.cp_mode, cfg.symbol_guard, cfg.pilot_spacing, cfg.use_pilots,
            cfg.modulation, cfg.code_rate)]
        r = subprocess.run(args, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.returncode, r.stderr[-400:])
        got = np.fromfile(fo, np.float32)
        o = oracle.chirp_detect(x)
        assert got[0] == o["success"] and np.float32(got[2]).tobytes() == np.float32(o["cfo_hz"]).tobytes()
        assert np.float32(got[3]).tobytes() == np.float32(max(o["up_correlation"], o["down_correlation"])).tobytes()
        if o["success"]:
            assert int(got[1]) == o["start_sample"]
            want, _, _ = oracle.demod_presynced(cfg, x[o["start_sample"]:], o["cfo_hz"],
                                                chirp_initial_phase(o["cfo_hz"], o["start_sample"], cfg.sample_rate))
            assert beq(got[4:], want)
        else:
            assert got.size == 4


SRC_FRAME = r'''
#include "ultra_hip_waveform.hpp"
#include <cstdio>
#include <string>
#include <vector>
using namespace ultra_hip;
int main(int argc, char** argv) {
    // argv: soft.f32 n out.bin rate bits_per_symbol(0 = interleaving off, "default" = no setter call at all)
    FILE* f = std::fopen(argv[1], "rb");
    const size_t n = std::stoul(argv[2]);
    std::vector<float> soft(n);
    if (std::fread(soft.data(), 4, n, f) != n) return 2;
    std::fclose(f);
    HipRxFrameDecoder dec;
    dec.setDataMode(static_cast<CodeRate>(std::stoi(argv[4])), true);
    if (std::string(argv[5]) != "default") {
        const size_t bps = std::stoul(argv[5]);
        dec.setInterleavingEnabled(bps != 0);
        if (bps) dec.setInterleaverConfig(bps);
    }
    HipRxFrameResult r = dec.decodeSoftBits(std::span<const float>(soft.data(), n));
    FILE* g = std::fopen(argv[3], "wb");
    const int head[6] = {r.success, r.is_ping, r.frame_type, r.codewords_ok, r.codewords_failed, dec.getExpectedCodewords()};
    std::fwrite(head, 4, 6, g); std::fwrite(r.frame_data.data(), 1, r.frame_data.size(), g); std::fclose(g);
    return 0;
}
'''


@pytest.mark.parametrize("rate,bps", [(0, 0), (4, 176), (0, "default")])
def test_cpp_frame_decoder(tmp_path, oracle, rate, bps):
    """HipRxFrameDecoder::decodeSoftBits (RxPipeline's decode half) from a C++ program, against the oracle.
    "default": no setter is called — like a default-constructed RxPipeline (ChannelInterleaver(60, 648), interleaving
    on: rx_pipeline.cpp:13-18) the decoder deinterleaves with 60 bits per symbol."""
    arg, bps = str(bps), (60 if bps == "default" else bps)
    from _util import v2_frame_cases
    src = tmp_path / "fd.cpp"
    src.write_text(SRC_FRAME)
    exe = tmp_path / "fd"
    lib = ROOT / "projectultra_amd"
    subprocess.check_call(["g++", "-O1", "-std=c++20", f"-I{ROOT / 'include'}", str(src), f"-L{lib}", "-lultra_hip",
                           f"-Wl,-rpath,{lib}", "-o", str(exe)])
    for t, (name, soft) in enumerate(v2_frame_cases(oracle, rate, np.random.default_rng(3), bps)[::2]):
        fin, fo = tmp_path / f"s{t}.f32", tmp_path / f"r{t}.bin"
        soft.tofile(fin)
        r = subprocess.run([str(exe), str(fin), str(soft.size), str(fo), str(rate), arg], capture_output=True,
                           text=True, timeout=300)
        assert r.returncode == 0, (name, r.returncode, r.stderr[-400:])
        raw = fo.read_bytes()
        head = np.frombuffer(raw[:24], np.int32)
        w = oracle.v2_decode_frame(rate, soft, bps)
        assert list(head) == [w[k] for k in ("success", "is_ping", "frame_type", "codewords_ok", "codewords_failed",
                                             "expected_codewords")], (name, head, w)
        assert raw[24:] == w["frame_data"], name


def test_cpp_monte_carlo_harness(tmp_path):
    """tools/nvis_mode_hip.cpp — the loop of the reference's tools/test_nvis_mode.cpp as one GPU batch per mode,
    host side in C++ over the C-ABI only.  Its counters equal those of the Python host for the same seed (same
    kernels, same generators), and the eight NVIS modes decode at 30 dB as they do in the reference tool."""
    import re
    import torch
    from projectultra_amd import CodeRate, Modulation, ReceiveContext, presets
    exe = tmp_path / "nvis_mode_hip"
    lib = ROOT / "projectultra_amd"
    subprocess.check_call(["g++", "-O2", "-std=c++20", f"-I{ROOT / 'include'}", str(ROOT / "tools" / "nvis_mode_hip.cpp"),
                           f"-L{lib}", "-lultra_hip", f"-Wl,-rpath,{lib}", "-o", str(exe)])
    trials, seed = 2048, 7
    r = subprocess.run([str(exe), "--snr", "30", "--trials", str(trials), "--seed", str(seed)], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-400:]
    rows = [tuple(int(v) for v in m.groups()) for m in re.finditer(r"^MODE (\d+) (\d+) (\d+) (\d+) (\d+) (\d+) (\d+)$", r.stdout, re.M)]
    assert len(rows) == 8
    for mod, rate, frames, ferr, berr, lfail, iters in rows:
        assert frames == trials
        assert ferr <= trials // 50, (mod, rate, ferr)                       # the reference tool reports 100 % at 30 dB
        mc = presets.nvis_mode().with_mode(Modulation(mod), CodeRate(rate))
        mc.pilot_spacing = 4 if mc.use_pilots else 2
        ctx = ReceiveContext(mc)
        audio, payload = ctx.make_batch(trials, seed=seed, channel="awgn", snr_db=30.0, delay_ms=0.0, doppler_hz=0.0)
        c = ctx.count_errors(ctx.demod_decode(audio), payload)
        torch.cuda.synchronize()
        c = [int(v) for v in c.tolist()]
        assert (c[0], c[1], c[2], c[4], c[5]) == (frames, ferr, berr, lfail, iters), (mod, rate, c)


def test_cpp_sweep_harness_equals_python_driver(tmp_path):
    """tools/sweep_hip.cpp — the configs[3] / configs[4] sweeps as a native harness (C++ over the C-ABI, RCCL
    communicator from ncclCommInitAll, one all-reduce per point).  Same generators, per-point seeds and sharding as
    projectultra_amd/sweep.py: every counter of every point equals the Python driver's."""
    import re
    from projectultra_amd import CodeRate, Modulation
    from projectultra_amd.sweep import ldpc_snr_sweep, mode_sweep
    exe = tmp_path / "sweep_hip"
    lib = ROOT / "projectultra_amd"
    subprocess.check_call(["g++", "-O2", "-std=c++20", f"-I{ROOT / 'include'}", str(ROOT / "tools" / "sweep_hip.cpp"),
                           f"-L{lib}", "-lultra_hip", f"-Wl,-rpath,{lib}", "-ldl", "-pthread", "-o", str(exe)])
    pat = re.compile(r"^POINT (\S+) (-?[\d.]+) (\d+) (\d+) (\d+) (\d+) (\d+) (\d+) (\d+)", re.M)
    keys = ("frames", "frame_errors", "bit_errors", "info_bits", "ldpc_fail", "iters_sum", "undetected_errors")

    r = subprocess.run([str(exe), "--config", "cfg4", "--trials", "20000", "--snr-from", "-6", "--snr-to", "0", "--snr-step", "2",
                        "--seed", "77", "--batch", "7000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-400:] + r.stderr[-400:]
    rows = pat.findall(r.stdout)
    want = ldpc_snr_sweep(CodeRate.R1_4, [-6.0, -4.0, -2.0, 0.0], 20000, seed=77, batch=5000)
    assert len(rows) == 4
    for row, p in zip(rows, want):
        assert float(row[1]) == p.snr_db and [int(v) for v in row[2:]] == [p.counters[k] for k in keys], (row, p.counters)

    r = subprocess.run([str(exe), "--config", "cfg5", "--trials", "2048", "--snr-from", "0", "--snr-to", "12", "--snr-step", "12",
                        "--seed", "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-400:] + r.stderr[-400:]
    rows = pat.findall(r.stdout)
    want = mode_sweep(None, [0.0, 12.0], frames_per_point=2048, seed=5)
    assert len(rows) == 60
    for row, p in zip(rows, want):
        assert row[0] == p.label.replace(" ", "_") and [int(v) for v in row[2:]] == [p.counters[k] for k in keys], (row, p.counters)



SRC_STREAM = r'''
#include "ultra_hip_waveform.hpp"
#include <cstdio>
#include <vector>
using namespace ultra_hip;
int main(int argc, char** argv) {
    // argv: in.f32 n_samples chunks.u32 n_calls out.bin fft carriers cp_mode guard pilot_spacing use_pilots mod rate
    const size_t n = std::stoul(argv[2]), n_calls = std::stoul(argv[4]);
    std::vector<float> audio(n);
    std::vector<uint32_t> chunks(n_calls);
    FILE* f = std::fopen(argv[1], "rb"); if (std::fread(audio.data(), 4, n, f) != n) return 2; std::fclose(f);
    f = std::fopen(argv[3], "rb"); if (std::fread(chunks.data(), 4, n_calls, f) != n_calls) return 2; std::fclose(f);
    ModemConfig c;
    c.fft_size = std::stoul(argv[6]); c.num_carriers = std::stoul(argv[7]);
    c.cp_mode = static_cast<decltype(c.cp_mode)>(std::stoi(argv[8])); c.symbol_guard = std::stoul(argv[9]);
    c.pilot_spacing = std::stoul(argv[10]); c.use_pilots = std::stoi(argv[11]) != 0;
    c.modulation = static_cast<Modulation>(std::stoi(argv[12])); c.code_rate = static_cast<CodeRate>(std::stoi(argv[13]));
    HipOfdmCoxWaveform w(c);                                    // what WaveformFactory::create(OFDM_COX, config) would hand out
    std::vector<uint32_t> trace;                                // per call: ready, synced, soft bits handed out
    std::vector<float> soft;
    size_t pos = 0;
    for (size_t i = 0; i < n_calls; ++i) {
        const bool ready = w.process(SampleSpan(audio.data() + pos, chunks[i]));
        pos += chunks[i];
        uint32_t drained = 0;
        if (ready) { std::vector<float> sb = w.getSoftBits(); drained = (uint32_t)sb.size(); soft.insert(soft.end(), sb.begin(), sb.end()); }
        trace.push_back(ready ? 1u : 0u); trace.push_back(w.isSynced() ? 1u : 0u); trace.push_back(drained);
    }
    FILE* g = std::fopen(argv[5], "wb");
    std::fwrite(trace.data(), 4, trace.size(), g); std::fwrite(soft.data(), 4, soft.size(), g); std::fclose(g);
    return 0;
}
'''


@pytest.mark.parametrize("name", ["cfg3_qam16_r34", "cfg2_dqpsk_r12"])
def test_cox_waveform_live_stream(tmp_path, name):
    """HipOfdmCoxWaveform as a LIVE stream, call by call against the compiled reference's OFDMDemodulator::process +
    getSoftBits (tests/golden/stream.npz): two frames in one stream (frame-complete exit, re-acquisition on the
    leftover buffer), the idle-call exit, the 250-symbol timeout; return values, isSynced() and every soft bit."""
    from _util import STREAM_SCENARIOS, build_stream
    from oracle.bindings import geometry
    g = np.load(GOLDEN / "fullsync.npz")
    want = np.load(GOLDEN / "stream.npz")
    cfg = cfg_from_array(g[f"{name}__cfg"])
    geo = geometry(cfg)
    src = tmp_path / "st.cpp"
    src.write_text(SRC_STREAM)
    exe = tmp_path / "st"
    lib = ROOT / "projectultra_amd"
    subprocess.check_call(["g++", "-O1", "-std=c++20", f"-I{ROOT / 'include'}", str(src), f"-L{lib}", "-lultra_hip",
                           f"-Wl,-rpath,{lib}", "-o", str(exe)])
    pre = int(g[f"{name}__meta"][0][0])
    for sc, recipe in STREAM_SCENARIOS.items():
        audio, chunks = build_stream(g[f"{name}__audio"], recipe(geo.symbol_samples, pre))
        fin, fc, fo = tmp_path / f"{sc}.f32", tmp_path / f"{sc}.u32", tmp_path / f"{sc}.out"
        audio.tofile(fin); chunks.tofile(fc)
        args = [str(exe), str(fin), str(audio.size), str(fc), str(chunks.size), str(fo)] + [str(int(x)) for x in (
            cfg.fft_size, cfg.num_carriers, cfg.cp_mode, cfg.symbol_guard, cfg.pilot_spacing, cfg.use_pilots,
            cfg.modulation, cfg.code_rate)]
        r = subprocess.run(args, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (sc, r.returncode, r.stderr[-400:])
        raw = np.fromfile(fo, np.uint32)
        trace = raw[:3 * chunks.size].reshape(-1, 3)
        soft = raw[3 * chunks.size:].view(np.float32)
        assert np.array_equal(trace[:, 0], want[f"{name}__{sc}__ready"]), (name, sc, "ready")
        assert np.array_equal(trace[:, 1], want[f"{name}__{sc}__synced"]), (name, sc, "synced")
        assert np.array_equal(trace[:, 2], want[f"{name}__{sc}__drained"]), (name, sc, "drained")
        assert beq(soft, want[f"{name}__{sc}__soft"]), (name, sc, "soft bits")


SRC_ADAPTIVE = r'''
#include "ultra_hip_waveform.hpp"
#include <cstdio>
#include <vector>
using namespace ultra_hip;
int main(int argc, char** argv) {
    // argv: in.f32 n_samples out.f32 fft carriers cp_mode guard pilot_spacing use_pilots mod rate use_rls lms_mu rls_lambda
    FILE* f = std::fopen(argv[1], "rb");
    const size_t n = std::stoul(argv[2]);
    std::vector<float> audio(n);
    if (std::fread(audio.data(), 4, n, f) != n) return 2;
    std::fclose(f);
    ModemConfig c;
    c.fft_size = std::stoul(argv[4]); c.num_carriers = std::stoul(argv[5]);
    c.cp_mode = static_cast<decltype(c.cp_mode)>(std::stoi(argv[6])); c.symbol_guard = std::stoul(argv[7]);
    c.pilot_spacing = std::stoul(argv[8]); c.use_pilots = std::stoi(argv[9]) != 0;
    c.modulation = static_cast<Modulation>(std::stoi(argv[10])); c.code_rate = static_cast<CodeRate>(std::stoi(argv[11]));
    c.adaptive_eq_enabled = true;                               // ModemConfig's own fields (types.hpp:170-174)
    c.adaptive_eq_use_rls = std::stoi(argv[12]) != 0; c.lms_mu = std::stof(argv[13]); c.rls_lambda = std::stof(argv[14]);
    HipOfdmCoxWaveform rx(c);
    bool ready = false;
    for (size_t i = 0; i < n; i += 960) ready = rx.process(SampleSpan(audio.data() + i, std::min<size_t>(960, n - i)));
    if (!ready) return 3;
    std::vector<float> soft = rx.getSoftBits();                 // the frame's first 648, as OFDMNvisWaveform hands them out
    FILE* g = std::fopen(argv[3], "wb"); std::fwrite(soft.data(), 4, soft.size(), g); std::fclose(g);
    return 0;
}
'''


@pytest.mark.parametrize("name", ["live_lms_qam256", "live_rls_qam64"])
def test_cpp_adapter_with_the_adaptive_equaliser(tmp_path, name):
    """ModemConfig::adaptive_eq_enabled through the C++ adapter: whole frames the compiled reference received with the switch
    on (OFDMDemodulator::process in 960-sample chunks; tests/golden/adaptive.npz, live_*) — to_c_config carries the five fields
    into ultra_hip_config, the live path keeps the weights in the stream's tracker record from call to call."""
    g = np.load(GOLDEN / "adaptive.npz")
    cfg = cfg_from_array(g[f"{name}__cfg"])
    src = tmp_path / "rxa.cpp"
    src.write_text(SRC_ADAPTIVE)
    exe = tmp_path / "rxa"
    lib = ROOT / "projectultra_amd"
    subprocess.check_call(["g++", "-O1", "-std=c++20", f"-I{ROOT / 'include'}", str(src), f"-L{lib}", "-lultra_hip",
                           f"-Wl,-rpath,{lib}", "-o", str(exe)])
    for t, (a, want) in enumerate(zip(g[f"{name}__audio"], g[f"{name}__llr"])):
        fin, fl = tmp_path / f"in{t}.f32", tmp_path / f"llr{t}.f32"
        a.astype(np.float32).tofile(fin)
        args = [str(exe), str(fin), str(a.size), str(fl)] + [str(int(x)) for x in (
            cfg.fft_size, cfg.num_carriers, cfg.cp_mode, cfg.symbol_guard, cfg.pilot_spacing, cfg.use_pilots,
            cfg.modulation, cfg.code_rate, cfg.adaptive_eq_use_rls)] + [repr(float(np.float32(cfg.lms_mu))), repr(float(np.float32(cfg.rls_lambda)))]
        r = subprocess.run(args, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.returncode, r.stderr[-400:])
        got = np.fromfile(fl, np.float32)
        assert got.size == 648, (name, t, got.size)
        bad = np.flatnonzero(got.view(np.uint32) != want[:648].view(np.uint32))
        assert bad.size == 0, (name, t, bad[:8], got[bad[:4]], want[bad[:4]])

