// lanczos_kernels_common.hpp -- parameter blocks and device helpers shared by the HIP kernels.
// The translation unit is compiled with -ffp-contract=off: a*b+c written with operators is TWO
// roundings everywhere (what the reference's x86-64 build does, full_TB.h:60,73); fused multiply-adds
// appear only where they are spelled __builtin_fmaf / __builtin_fma.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

// Cache policy of the output stores: non-temporal + system scope (nt sc1).  The output is written once and never read
// by the kernel; letting it allocate in L2 evicts the input rows that neighbouring workgroups re-read.  Measured
// (interleaved A/B, config 2): default 122 us, nt 112 us, nt+sc0 112 us, nt+sc1 110 us; nt on the input LOADS: 125 us.
#ifndef LZ_STORE_AUX
#define LZ_STORE_AUX 18
#endif

namespace lz {

// one marching workgroup's share: strip `tx` of frame `frame`, input-row-indexed output rows m in [m_b, m_e)
struct WgEntry {
    int frame, tx, m_b, m_e;
};

struct FrameGeom {
    // all pitches/strides in BYTES; dims in pixels; sample = one channel of one pixel
    const uint8_t* in;   // first input row held by the caller (= full-frame row in_row0)
    uint8_t* out;        // first output row to write (= full-frame row out_row0)
    unsigned long long in_frame_stride, out_frame_stride;
    int in_pitch, out_pitch;
    int in_w, in_h, out_w, out_h;  // full frame
    int channels, a;
    int in_row0, in_rows; // full-frame index of the first row behind `in`, and how many rows it holds
    int out_row0, out_rows;  // strip to produce
    int skip_rows;        // output rows < skip_rows are left to the in-place prefix kernel
    int frames;
    unsigned long long* stamps;  // diagnostic builds only: per-wave cycle sums (see k_march STAMP)
    // k_march only: the grid is 1-D, `n_main` marching workgroups (`wg_per_frame` per frame) followed by
    // `prefix_blocks_per_frame` workgroups per frame that produce output rows [0, prefix_K) (the in-place prefix, see
    // k_prefix); prefix_K == 0: a separate k_prefix launch does that
    int n_main, wg_per_frame, prefix_blocks_per_frame, prefix_K, prefix_M, prefix_M2;
    int hls_bp;           // k_hls only: BIT_PRECISION of the fixed-point emulation (0 = ideal arithmetic)
    int debug_skip;       // ablation bits for profiling builds (0 in production): 1 H-pass, 2 fix-up, 4 V-pass, 8 stores, 16 loads
    const WgEntry* wg_tab; // k_march only: [n_main][wg_segs] share of every marching workgroup, indexed by the hardware block id
    int wg_segs;           // segments (table entries) per workgroup
};

struct TapTables {
    const int32_t* h_first;  // [out_w]
    const double* h_w;       // [out_w][2a]
    const int32_t* v_first;  // [out_h]
    const double* v_w;       // [out_h][2a]
    // double weights of the exact chains of the specialised kernels, in MEMORY on purpose: [0][k] integer phase
    // (FastConsts::wi), [ph][k] phase ph (FastConsts::wd).  As kernel arguments the compiler loads them once and holds 24+
    // SGPRs through the whole march for a path that runs for one sample in ten thousand.
    const double* x_w;       // [1 + kFastMaxS... rows of kMaxTaps] see lanczos_fast.hpp
};

// full_TB.h:29-37: x > max -> max; x < 0 -> 0; else truncate
template <typename T>
__device__ __forceinline__ T store_convert(double x) {
    constexpr double kMax = sizeof(T) == 1 ? 255.0 : 65535.0;
    if (x > kMax) return (T)kMax;
    if (x < 0) return (T)0;
    return (T)(unsigned)x;
}

}  // namespace lz
