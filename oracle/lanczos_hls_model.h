/*
 * oracle/lanczos_hls_model.h -- CPU model of the reference's HLS path SEMANTICS (lanczos(), lanczos.cpp:86-98).
 *
 * TEST INFRASTRUCTURE ONLY, like lanczos_oracle.h.   *** PARITY UNPINNED ***
 * The HLS path cannot be compiled here (it needs Xilinx ap_fixed.h / hls_stream.h / hls_math.h, absent from the
 * reference tree and from the image; no stand-ins are written), the reference holds no outputs of it, and its
 * arithmetic is fixed point (ap_fixed<10+BP,10> accumulators, ap_fixed<8+BP,8> weights from hls::sinpi, a fixed-point
 * phase stepper that is inexact for non-integer scales).  What is restated here is the ALGORITHM in ideal arithmetic
 * (double), which is what LANCZOS_MODE_HLS of the product implements:
 *
 *   order        vertical pass first, then horizontal                      lanczos.cpp:21-51, :68-83, :97
 *   window       input indices floor(o*D/N)-a+1 .. floor(o*D/N)+a          worker.cpp:138-155, :170-198 (priming), :225-236
 *   weights      ROM[|o*D - i*N|], ROM[k] = L(k/N) for k < a*N, ROM[a*N] = 0,
 *                L(x) = a/pi^2 * sinpi(x) * sinpi(x/a) / x^2, L(0) = 1       kernel.cpp:12-18, :40-59
 *   borders      above / left of the image: zero samples                   worker.cpp:176-188, :256-265
 *                below / right of the image: the last row / sample again    worker.cpp:147-153 (push(saturate)), :244
 *   de-ring      every pass clamps its sum to [min, max] of the two centre taps (window slots a-1, a)
 *                                                                          worker.cpp:66-74, :103-111
 *   between      the vertical result stays a (clamped) real number          worker.cpp:45-78 (num_t out)
 *   store        truncation to the integer sample                          worker.cpp:118-130 (clamp_to_byte)
 *   accumulation acc += kern[i] * px[i], i ascending                        worker.cpp:58-64, :95-101
 * Fixed-point emulation (the _fx entry, bit_precision = BP > 0, 8-bit samples): what the ap_fixed declarations of
 * lanczos.h:74-81 imply under their defaults (AP_TRN: quantisation by truncation towards minus infinity, AP_WRAP):
 *   ROM entries keep BP fractional bits                  kernel_t = ap_fixed<8+BP,8>, kernel.cpp:42
 *   vertical pass: kernel_t x byte products are exact, the sum lives in num_el_t = ap_fixed<10+BP,10>   worker.cpp:58-64
 *   horizontal pass: kernel_t x num_el_t has 2 BP fractional bits, every `acc +=` stores BP of them      worker.cpp:95-101
 * Differences from the hardware that REMAIN: the ROM is truncated from the ideal L(k/N), not from hls::sinpi's fixed-point
 * result (kernel.cpp:12-18 evaluates the formula in kernel_t arithmetic: unknown low bits), and the phase stepping is exact
 * (floor(o*D/N) instead of the fixed-point fractional test worker.cpp:140,234).  Still PARITY UNPINNED.
 */
#ifndef LANCZOS_HLS_MODEL_H
#define LANCZOS_HLS_MODEL_H

#include <stdint.h>

#include "lanczos_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ROM entry k of kernel.cpp:40-45 for (a, N): L(k/N), exactly 0 at k = a*N and wherever sinpi(k/N) is exactly 0 */
double oracle_hls_rom(int k, int a, int scale_n);
/* weight of input index i for output index o: ROM[|o*D - i*N|] (kernel.cpp:50-59) */
double oracle_hls_weight(int i, int o, int a, int scale_n, int scale_d);

/* interleaved [H][W][C] in -> [OUT_H][OUT_W][C] out; cfg as for the software-path oracle */
int oracle_hls_expected_hwc_u8(const oracle_cfg* cfg, const uint8_t* in, uint8_t* out, int threads);
int oracle_hls_expected_hwc_u16(const oracle_cfg* cfg, const uint16_t* in, uint16_t* out, int threads);
/* the same with BIT_PRECISION fractional bits (1..20; 0 = the ideal arithmetic above), 8-bit samples */
int oracle_hls_expected_hwc_u8_fx(const oracle_cfg* cfg, const uint8_t* in, uint8_t* out, int threads, int bit_precision);
double oracle_hls_weight_fx(int i, int o, int a, int scale_n, int scale_d, int bit_precision);

#ifdef __cplusplus
}
#endif
#endif
