// frame_kernel.h — v2 wire format on the receive path (scope row f4, second half): what
// RxPipeline::processFrame does with the soft bits of a frame (src/gui/modem/rx_pipeline.cpp:283-346):
// detectPing (:446-472), then decodeFrame (:348-444) — CW0 -> v2::parseHeader (src/protocol/frame_v2.cpp:
// 1175-1229: magic, type, CRC-16 of the control frame or of the data header) -> count the remaining
// codewords -> CodewordStatus::reassemble / reassembleCodewords (:952-982,1023-1044).
//
// The codewords themselves are decoded by ldpc_decode_kernel (all codewords of all frames in one batch,
// the per-codeword channel deinterleaver fused into its LLR load); this kernel is the byte work behind it:
// one lane per frame, a few hundred byte operations each.
#ifndef ULTRA_FRAME_KERNEL_H
#define ULTRA_FRAME_KERNEL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ultra_hip {
namespace dev {

// ControlFrame::calculateCRC (frame_v2.cpp:111-124): CRC-16/CCITT-FALSE, bit by bit
__device__ __forceinline__ unsigned crc16_ccitt(const uint8_t* d, int n) {
    unsigned crc = 0xFFFFu;
    for (int i = 0; i < n; ++i) {
        crc ^= (unsigned)d[i] << 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) crc = ((crc & 0x8000u) ? ((crc << 1) ^ 0x1021u) : (crc << 1)) & 0xFFFFu;
    }
    return crc;
}

__device__ __forceinline__ bool v2_is_control(unsigned t) {   // isControlFrame, frame_v2.hpp:212-217
    return t == 0x10 || t == 0x11 || t == 0x16 || t == 0x17 || t == 0x20 || t == 0x21 || t == 0x40;
}

// results: 8 int32 per frame = ultra_hip_frame_result {success, is_ping, frame_type, codewords_ok,
// codewords_failed, expected_codewords, frame_len, status}.  Decoded codeword (f, c) sits at index
// f * idx_frame + c * idx_cw of bytes[..][decoded_bytes] / okv[..].
__global__ __launch_bounds__(64) void frame_assemble_kernel(
    const float* __restrict__ soft, size_t frame_stride, unsigned n_soft, int n_frames,
    const uint8_t* __restrict__ bytes, const uint8_t* __restrict__ okv, int decoded_bytes, int bytes_per_cw,
    size_t idx_frame, size_t idx_cw, int32_t* __restrict__ results, uint8_t* __restrict__ frame_data,
    size_t data_stride) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    int32_t* res = results + (size_t)f * 8;
    int success = 0, is_ping = 0, type = 0x10, n_ok = 0, n_fail = 0, expected_out = 0, len = 0, status = 0;
    auto finish = [&]() {
        res[0] = success; res[1] = is_ping; res[2] = type; res[3] = n_ok; res[4] = n_fail; res[5] = expected_out;
        res[6] = len; res[7] = status;
    };
    if (n_soft == 0) { finish(); return; }
    // detectPing: hard bytes with soft > 0 -> bit 1; "ULTR" or its bit inversion
    if (n_soft >= 32) {
        const float* s = soft + (size_t)f * frame_stride;
        unsigned w = 0;
        for (int i = 0; i < 32; ++i) w = (w << 1) | (s[i] > 0.0f ? 1u : 0u);
        if (w == 0x554C5452u || w == 0xAAB3ABADu) { success = 1; is_ping = 1; type = 0x01; status = 5; finish(); return; }
    }
    const int num_cw = (int)(n_soft / 648u);
    if (num_cw == 0) { finish(); return; }
    const uint8_t* cw0 = bytes + ((size_t)f * idx_frame) * decoded_bytes;
    // decodeSingleCodeword: success and at least bytes_per_cw decoded bytes (always true: ceil(k/8) >= floor(k/8))
    if (!okv[(size_t)f * idx_frame]) { n_fail = 1; finish(); return; }
    n_ok = 1;
    // RxPipeline::parseHeader = identifyCodeword HEADER (magic) + v2::parseHeader (needs 20 bytes)
    bool valid = bytes_per_cw >= 20 && cw0[0] == 0x55 && cw0[1] == 0x4C;
    int expected = 0, payload_len = 0;
    bool control = false;
    if (valid) {
        control = v2_is_control(cw0[2]);
        if (control) {
            valid = (((unsigned)cw0[18] << 8) | cw0[19]) == crc16_ccitt(cw0, 18);
            expected = 1;
        } else {
            expected = cw0[12];
            payload_len = ((int)cw0[13] << 8) | cw0[14];
            valid = (((unsigned)cw0[15] << 8) | cw0[16]) == crc16_ccitt(cw0, 15);
        }
    }
    // total_cw == 0 indexes an empty vector in the reference (undefined behaviour): treated as an invalid header
    if (!valid || expected == 0) { status = 1; finish(); return; }
    type = cw0[2];
    if (num_cw < expected) { expected_out = expected; status = 2; finish(); return; }
    const int expected_size = control ? 20 : 17 + payload_len + 2;
    uint8_t* out = frame_data + (size_t)f * data_stride;
    {
        const int to_copy = expected_size < bytes_per_cw ? expected_size : bytes_per_cw;
        for (int b = 0; b < to_copy; ++b) out[b] = cw0[b];
        len = to_copy;
    }
    bool all_ok = true;
    for (int c = 1; c < expected; ++c) {
        const size_t idx = (size_t)f * idx_frame + (size_t)c * idx_cw;
        if (!okv[idx]) { ++n_fail; all_ok = false; continue; }
        ++n_ok;
        const int remaining = expected_size - len;
        if (remaining == 0) continue;
        const uint8_t* cw = bytes + idx * decoded_bytes;
        const int skip = (cw[0] == 0xD5) ? 2 : 0;                // marker + index, or the old format without them
        const int room = bytes_per_cw - skip;
        const int to_copy = remaining < room ? remaining : room;
        for (int b = 0; b < to_copy; ++b) out[len + b] = cw[skip + b];
        len += to_copy;
    }
    if (all_ok) { success = 1; status = 4; }
    else { status = 3; len = 0; }
    finish();
}

}  // namespace dev
}  // namespace ultra_hip
#endif
