"""The header-only C++ adapter (include/ultra_hip_waveform.hpp) compiles stand-alone against the
C-ABI, and — where the reference tree is available — inside it, deriving from ultra::IWaveform."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = r'''
#include "ultra_hip_waveform.hpp"
#include <type_traits>
int main() {
    ultra_hip::ModemConfig c;
    ultra_hip::HipOfdmWaveform w(c);
    w.configure(ultra_hip::Modulation::DQPSK, ultra_hip::CodeRate::R1_2);
    ultra_hip::HipRxFrameDecoder fd;
    fd.setDataMode(ultra_hip::CodeRate::R3_4, true);
    fd.setInterleaverConfig(176);
#ifdef ULTRA_HIP_WITH_REFERENCE
    ultra::WaveformPtr p = std::make_unique<ultra_hip::HipOfdmWaveform>(c);   // through the plugin pointer type
    (void)p;
    // the Schmidl-Cox flavour — what WaveformFactory::create(OFDM_COX) returns — as a complete ultra::IWaveform
    auto make_cox = [](const ultra::ModemConfig& cfg) -> ultra::WaveformPtr { return std::make_unique<ultra_hip::HipOfdmCoxWaveform>(cfg); };
    (void)make_cox;
    static_assert(std::is_base_of_v<ultra::IWaveform, ultra_hip::HipOfdmCoxWaveform> && !std::is_abstract_v<ultra_hip::HipOfdmCoxWaveform>);
#endif
    return (w.getSamplesPerSymbol() == 564 && w.getMinSamplesForFrame() == 13 * 564 && w.getCarrierCount() == 30) ? 0 : 1;
}
'''


def test_adapter_standalone(tmp_path, hiplib):
    src = tmp_path / "a.cpp"
    src.write_text(SRC)
    exe = tmp_path / "a"
    lib = ROOT / "projectultra_amd"
    subprocess.check_call(["g++", "-std=c++20", f"-I{ROOT / 'include'}", str(src), f"-L{lib}", "-lultra_hip",
                           f"-Wl,-rpath,{lib}", "-o", str(exe)])
    assert subprocess.run([str(exe)]).returncode == 0


def test_adapter_inside_reference_tree(tmp_path):
    ref = Path("/root/reference")
    if not (ref / "src" / "waveform" / "waveform_interface.hpp").exists():
        pytest.skip("reference tree not present")
    src = tmp_path / "a.cpp"
    src.write_text(SRC)
    subprocess.check_call(["g++", "-std=c++20", "-DULTRA_HIP_WITH_REFERENCE", f"-I{ROOT / 'include'}",
                           f"-I{ref / 'include'}", f"-I{ref / 'src'}", "-fsyntax-only", str(src)])


@pytest.mark.parametrize("name", ["hip_ofdm_demodulator.cpp", "hip_ldpc_decoder.cpp", "hip_waveform_factory.cpp"])
def test_link_time_drop_ins_compile_inside_the_reference_tree(tmp_path, name):
    """projectultra_amd/host/*.cpp DEFINE the reference's own classes (ultra::OFDMDemodulator, ultra::LDPCDecoder and the
    interleavers, WaveformFactory::create) against the reference's headers: every member the headers declare must be defined
    with the declared signature — the compiler checks that here; tests/test_gpu_pimpl.py runs the result."""
    ref = Path("/root/reference")
    if not (ref / "include" / "ultra" / "ofdm.hpp").exists():
        pytest.skip("reference tree not present")
    obj = tmp_path / (name + ".o")
    subprocess.check_call(["g++", "-std=c++20", "-O0", "-w", "-DULTRA_HIP_WITH_REFERENCE", f"-I{ROOT / 'include'}", f"-I{ref / 'include'}",
                           f"-I{ref / 'src'}", "-c", str(ROOT / "projectultra_amd" / "host" / name), "-o", str(obj)])
    syms = subprocess.run(["nm", "-C", "--defined-only", str(obj)], capture_output=True, text=True).stdout
    want = {"hip_ofdm_demodulator.cpp": ["ultra::OFDMDemodulator::process(", "ultra::OFDMDemodulator::processPresynced(", "ultra::OFDMDemodulator::getSoftBits()",
                                         "ultra::OFDMDemodulator::getData()", "ultra::OFDMDemodulator::setFrequencyOffsetWithPhase(", "ultra::OFDMDemodulator::setTimingOffset(",
                                         "ultra::OFDMDemodulator::hasPendingData() const", "ultra::OFDMDemodulator::getLastSyncOffset() const", "ultra::OFDMDemodulator::reset()",
                                         "ultra::OFDMDemodulator::getChannelQuality() const", "ultra::OFDMDemodulator::getConstellationSymbols() const",
                                         "ultra::ChannelEstimator::equalize("],
            "hip_ldpc_decoder.cpp": ["ultra::LDPCDecoder::decodeSoft(", "ultra::LDPCDecoder::decode(", "ultra::LDPCDecoder::lastIterations() const",
                                     "ultra::LDPCDecoder::setMaxIterations(", "ultra::LDPCDecoder::setRate(", "ultra::Interleaver::deinterleave(",
                                     "ultra::ChannelInterleaver::ChannelInterleaver(", "ultra::ChannelInterleaver::interleave("],
            "hip_waveform_factory.cpp": ["ultra::WaveformFactory::create(ultra::protocol::WaveformMode)", "ultra::WaveformFactory::createMCDPSK(int)"]}[name]
    for w in want:
        assert w in syms, (name, w)


def test_host_interleavers_equal_the_oracle(tmp_path, oracle):
    """The drop-in's ChannelInterleaver / Interleaver (host permutations that live in the replaced ldpc_decoder.cpp) against the
    oracle's, through a small program linked with libultra_hip.so only (ultra_hip_channel_interleaver_step needs no GPU)."""
    ref = Path("/root/reference")
    if not (ref / "include" / "ultra" / "fec.hpp").exists():
        pytest.skip("reference tree not present")
    import numpy as np
    src = tmp_path / "il.cpp"
    src.write_text(r'''
#include "ultra/fec.hpp"
#include <cstdio>
int main() {
    for (size_t bps : {60, 30, 90, 116, 176, 220}) {
        ultra::ChannelInterleaver ci(bps);
        std::vector<float> v(648); for (size_t i = 0; i < 648; ++i) v[i] = float(i);
        auto d = ci.deinterleave(v); auto a = ci.interleave(v);
        std::printf("%zu %zu:", bps, ci.getSymbolSeparation());
        for (float x : d) std::printf(" %d", int(x));
        std::printf(" |");
        for (float x : a) std::printf(" %d", int(x));
        std::printf("\n");
    }
    ultra::Interleaver il(6, 108);
    std::vector<float> v(648); for (size_t i = 0; i < 648; ++i) v[i] = float(i);
    auto d = il.deinterleave(v);
    std::printf("rc:"); for (float x : d) std::printf(" %d", int(x)); std::printf("\n");
    return 0;
}''')
    exe = tmp_path / "il"
    lib = ROOT / "projectultra_amd"
    subprocess.check_call(["g++", "-std=c++20", "-w", "-DULTRA_HIP_WITH_REFERENCE", f"-I{ROOT / 'include'}", f"-I{ref / 'include'}", f"-I{ref / 'src'}",
                           str(src), str(ROOT / "projectultra_amd" / "host" / "hip_ldpc_decoder.cpp"), f"-L{lib}", "-lultra_hip", f"-Wl,-rpath,{lib}", "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines()
    for line, bps in zip(out[:6], (60, 30, 90, 116, 176, 220)):
        head, rest = line.split(":")
        de, inter = rest.split("|")
        perm, inv = oracle.channel_interleaver_perm(bps)
        got_d = np.array(de.split(), dtype=np.int64); got_i = np.array(inter.split(), dtype=np.int64)
        want_d = np.empty(648, np.int64); want_d[inv] = np.arange(648)        # out[inverse[i]] = in[i]
        want_i = np.empty(648, np.int64); want_i[perm] = np.arange(648)       # out[perm[i]] = in[i]
        assert (got_d == want_d).all() and (got_i == want_i).all(), bps
    got = np.array(out[6].split(":")[1].split(), dtype=np.int64)
    want = oracle.interleaver_deinterleave(6, 108, np.arange(648, dtype=np.float32)).astype(np.int64)
    assert (got == want).all()
