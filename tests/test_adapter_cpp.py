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
