// lanczos_hls.hpp -- LANCZOS_MODE_HLS: the SEMANTICS of the reference's HLS pipeline (lanczos(), lanczos.cpp:86-98) instead of
// those of its software model.  What differs from the default modes (every item is the reference's, cited):
//   * vertical pass first, then horizontal                           lanczos.cpp:21-51 (fillColBuffer, fillRowBuffer), :97
//   * weights from the ROM of kernel.cpp:40-59: L(k/N) at k = |o*D - i*N|, exactly 0 at whole-pixel distances and at k = a*N
//   * rows above / samples left of the image are zeros               worker.cpp:176-188, :256-265
//     rows below / samples right of it repeat the last one           worker.cpp:147-153 (push(saturate)), :244
//   * de-ringing: each pass clamps its sum to [min, max] of the two centre taps   worker.cpp:66-74, :103-111
//   * the vertical result stays a real number between the passes (num_t), only the final store truncates  worker.cpp:118-130
// The hardware computes in ap_fixed (BIT_PRECISION fractional bits) with a fixed-point phase stepper; this mode computes the
// same algorithm in f64 with exact stepping (floor(o*D/N)) and, on request (lanczos_desc.reserved[0] = BIT_PRECISION, 8-bit
// samples), with the quantisation the ap_fixed declarations imply (lanczos.h:74-81, defaults AP_TRN / AP_WRAP):
//   ROM entries cut to BP fractional bits (host tables)      kernel_t  = ap_fixed<8+BP,8>,  kernel.cpp:42
//   V pass: kernel_t x byte is exact, the sum wraps at 10 integer bits                     worker.cpp:58-64
//   H pass: kernel_t x num_el_t has 2 BP fractional bits, every `acc +=` keeps BP of them  worker.cpp:95-101
// (all values are multiples of 2^-2BP below 2^10: exact in f64 for BP <= 20).  What stays different from the hardware: its
// ROM comes out of hls::sinpi in kernel_t arithmetic (unknown low bits), its phase stepper is fixed point.
// PARITY UNPINNED: the HLS path cannot be built here (Xilinx headers absent) and the reference holds no outputs of it; the
// checker is oracle/lanczos_hls_model.c, bit for bit.
//
// One workgroup = 64 output pixels x 8 output rows of one frame:
//   1. for every (row of the tile, input column the tile's horizontal windows touch, channel): the clamped vertical sum, f64,
//      into LDS (the reference's buf1[IN_WIDTH][ROW_WORKERS], lanczos.cpp:75, here only the columns this tile needs)
//   2. every output sample: the clamped horizontal sum over its 2a columns of that LDS tile, truncated, stored.
// Not a roofline kernel: f64 throughout, ~30 us per 1080p -> 4K frame.  It exists for results, the marching kernel for speed.
#pragma once
#include "lanczos_kernels_common.hpp"
#include "lanczos_taps.hpp"

namespace lz {

constexpr int kHlsTileW = 64;   // output pixels per workgroup
constexpr int kHlsTileH = 8;    // output rows per workgroup
constexpr int kHlsMaxCols = kHlsTileW + 2 * kMaxA + 2;  // input columns one tile can touch (scale > 1)
constexpr int kHlsThreads = 256;

template <typename T>
__global__ __launch_bounds__(kHlsThreads) void k_hls(FrameGeom g, TapTables t) {
    __shared__ double vbuf[kHlsTileH][kHlsMaxCols * 4];  // [tile row][input column - q0][channel]

    const int taps = 2 * g.a, A = g.a, C = g.channels, bp = g.hls_bp;
    const double fx_s = __builtin_ldexp(1.0, bp), fx_is = __builtin_ldexp(1.0, -bp);
    auto wrap10 = [](double x) {  // AP_WRAP of ap_fixed<10+BP,10>: never taken for clamped 8-bit data, kept for fidelity
        if (x >= -512.0 && x < 512.0) return x;
        double r = __builtin_fmod(x + 512.0, 1024.0);
        if (r < 0) r += 1024.0;
        return r - 512.0;
    };
    const int tiles_x = (g.out_w + kHlsTileW - 1) / kHlsTileW;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int frame = blockIdx.y;
    const int x0 = tx * kHlsTileW;
    const int x1 = x0 + kHlsTileW < g.out_w ? x0 + kHlsTileW : g.out_w;
    const int y0 = g.out_row0 + ty * kHlsTileH;
    const int y1 = y0 + kHlsTileH < g.out_row0 + g.out_rows ? y0 + kHlsTileH : g.out_row0 + g.out_rows;
    const int q0 = t.h_first[x0];                       // may be negative: zero samples left of the image
    const int q1 = t.h_first[x1 - 1] + taps - 1;        // may exceed in_w - 1: the last sample again
    const int nq = q1 - q0 + 1;
    const uint8_t* in_f = g.in + (size_t)frame * g.in_frame_stride;
    uint8_t* out_f = g.out + (size_t)frame * g.out_frame_stride;

    // ---- 1. vertical pass (ColWorkers::exec + compute, worker.cpp:138-155, :45-78) of the columns q0..q1
    for (int it = threadIdx.x; it < (y1 - y0) * nq * C; it += kHlsThreads) {
        const int c = it % C, qi = (it / C) % nq, r = it / (C * nq);
        const int y = y0 + r, q = q0 + qi;
        double v = 0.0;
        if (q >= 0) {
            const int qc = q > g.in_w - 1 ? g.in_w - 1 : q;
            const int first = t.v_first[y];
            const double* w = t.v_w + (size_t)y * taps;
            double acc = 0, lo = 0, hi = 0;
            for (int k = 0; k < taps; k++) {
                const int rr = first + k;   // < 0: zero row (worker.cpp:176-188); > in_h-1: the last row again (:147-153)
                const int rc = rr > g.in_h - 1 ? g.in_h - 1 : rr;
                const double px = rr < 0 ? 0.0 : (double)((const T*)(in_f + (size_t)(rc - g.in_row0) * g.in_pitch))[qc * C + c];
                acc += w[k] * px;                                     // worker.cpp:58-64
                if (k == A - 1) lo = hi = px;
                if (k == A) {
                    lo = px < lo ? px : lo;
                    hi = px > hi ? px : hi;
                }
            }
            if (bp > 0) acc = wrap10(acc);
            v = acc < lo ? lo : (acc > hi ? hi : acc);               // worker.cpp:66-74
        }
        vbuf[r][qi * C + c] = v;
    }
    __syncthreads();

    // ---- 2. horizontal pass (RowWorkers::exec + compute_ + clamp_to_byte, worker.cpp:225-236, :81-130)
    const int tw = x1 - x0;
    for (int it = threadIdx.x; it < (y1 - y0) * tw * C; it += kHlsThreads) {
        const int c = it % C, xi = (it / C) % tw, r = it / (C * tw);
        const int x = x0 + xi;
        const int first = t.h_first[x];
        const double* w = t.h_w + (size_t)x * taps;
        double acc = 0, lo = 0, hi = 0;
        for (int k = 0; k < taps; k++) {
            const int q = first + k;            // < 0: zero (worker.cpp:256-265); > in_w-1: clamped in step 1 (:244)
            const double px = vbuf[r][(q - q0) * C + c];
            if (bp > 0) acc = wrap10(__builtin_floor((acc + w[k] * px) * fx_s) * fx_is);  // num_el_t acc += kernel_t * num_el_t (AP_TRN)
            else acc += w[k] * px;                                    // worker.cpp:95-101
            if (k == A - 1) lo = hi = px;
            if (k == A) {
                lo = px < lo ? px : lo;
                hi = px > hi ? px : hi;
            }
        }
        const double v = acc < lo ? lo : (acc > hi ? hi : acc);      // worker.cpp:103-111
        ((T*)(out_f + (size_t)(y0 + r - g.out_row0) * g.out_pitch))[x * C + c] = (T)__builtin_floor(v);  // :118-130
    }
}

}  // namespace lz
