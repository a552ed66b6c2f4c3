// lanczos_generic.hpp -- table-driven kernels for ANY rational scale > 1, channel count and a.
//
//   k_generic  : one workgroup = 256 sample columns x 32 output rows.  The horizontal pass of the
//                input rows that tile needs goes to LDS as truncated integers (the reference's
//                between-pass store, full_TB.h:63), the vertical pass reads it back.  All sums are
//                f64 with a separate multiply and add per tap in ascending tap order, so every
//                sample is bit-identical to lanczos_expected() (full_TB.h:79-96).
//   k_prefix   : the first K output rows.  The reference's vertical pass runs in place from the
//                bottom row up (full_TB.h:67-77): for xx < K some taps i > xx read rows that already
//                hold OUTPUT values.  That is a short sequential recurrence per column; one thread
//                walks it for one sample column, entirely in f64.
//
// These are the always-exact fallback; the specialised kernels of lanczos_fast.hpp serve the
// integer-scale configurations the benchmark is quoted on.
#pragma once
#include "lanczos_kernels_common.hpp"
#include "lanczos_taps.hpp"

namespace lz {

constexpr int kGenTileW = 256;   // sample columns per workgroup (= threads)
constexpr int kGenTileH = 32;    // output rows per workgroup
constexpr int kGenMaxHRows = kGenTileH + 2 * kMaxA;  // S > 1: rows spanned <= TH/S + 2a

template <typename T>
__global__ __launch_bounds__(kGenTileW) void k_generic(FrameGeom g, TapTables t) {
    __shared__ T hbuf[kGenMaxHRows][kGenTileW];

    const int taps = 2 * g.a;
    const int C = g.channels;
    const int samples_w = g.out_w * C;
    const int tiles_x = (samples_w + kGenTileW - 1) / kGenTileW;
    const int tx = blockIdx.x % tiles_x;
    const int ty = blockIdx.x / tiles_x;
    const int frame = blockIdx.y;

    const int y_begin = g.out_row0 + ty * kGenTileH;
    int y_end = y_begin + kGenTileH;
    if (y_end > g.out_row0 + g.out_rows) y_end = g.out_row0 + g.out_rows;
    if (y_end <= g.skip_rows) return;  // the whole tile belongs to the prefix kernel (uniform)

    // input rows this tile reads (clamped to the image; out-of-range taps carry weight 0)
    int r_lo = t.v_first[y_begin];
    int r_hi = t.v_first[y_end - 1] + taps - 1;
    if (r_lo < 0) r_lo = 0;
    if (r_hi > g.in_h - 1) r_hi = g.in_h - 1;

    const int j = tx * kGenTileW + threadIdx.x;  // sample column
    const bool active = j < samples_w;
    const uint8_t* in_f = g.in + (size_t)frame * g.in_frame_stride;
    uint8_t* out_f = g.out + (size_t)frame * g.out_frame_stride;

    // ---- horizontal pass (full_TB.h:55-65) for rows r_lo..r_hi of this column ----
    if (active) {
        const int xx = j / C, c = j - xx * C;
        const int first = t.h_first[xx];
        double w[kMaxTaps];
        int idx[kMaxTaps];
#pragma unroll
        for (int k = 0; k < kMaxTaps; k++) {
            if (k < taps) {
                w[k] = t.h_w[(size_t)xx * taps + k];
                int i = first + k;
                i = i < 0 ? 0 : (i > g.in_w - 1 ? g.in_w - 1 : i);  // weight is 0 there
                idx[k] = i * C + c;
            } else {
                w[k] = 0.0;
                idx[k] = 0;
            }
        }
        for (int r = r_lo; r <= r_hi; r++) {
            const T* row = (const T*)(in_f + (size_t)(r - g.in_row0) * g.in_pitch);
            double sum = 0;
#pragma unroll
            for (int k = 0; k < kMaxTaps; k++)
                if (k < taps) sum += (double)row[idx[k]] * w[k];
            hbuf[r - r_lo][threadIdx.x] = store_convert<T>(sum);
        }
    }
    __syncthreads();

    // ---- vertical pass (full_TB.h:67-77, rows >= K: every tap reads an H-pass row) ----
    if (active) {
        for (int y = y_begin; y < y_end; y++) {
            if (y < g.skip_rows) continue;
            const int first = t.v_first[y];
            const double* wv = t.v_w + (size_t)y * taps;
            double sum = 0;
            for (int k = 0; k < taps; k++) {
                int i = first + k;
                i = i < r_lo ? r_lo : (i > r_hi ? r_hi : i);  // weight is 0 outside the image
                sum += (double)hbuf[i - r_lo][threadIdx.x] * wv[k];
            }
            T* orow = (T*)(out_f + (size_t)(y - g.out_row0) * g.out_pitch);
            orow[j] = store_convert<T>(sum);
        }
    }
}

// One thread per (frame, sample column): output rows [0, K) of the in-place vertical pass.
// The host launches it where (M + M2) rows of 32..128 columns fit 60 KB of LDS (deeper prefixes: k_prefix_stream below).  TAPS is a template parameter so the
// tap loops unroll and the H-pass loads of a row are all in flight together.
template <typename T, int TAPS>
__global__ __launch_bounds__(128) void k_prefix(FrameGeom g, TapTables t, int K, int M, int M2) {  // blockDim.x <= 128
    const int C = g.channels;
    const int samples_w = g.out_w * C;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int frame = blockIdx.y;
    if (j >= samples_w) return;
    const uint8_t* in_f = g.in + (size_t)frame * g.in_frame_stride;
    uint8_t* out_f = g.out + (size_t)frame * g.out_frame_stride;

    // H-pass rows 0..M2-1 and output rows 0..M-1 of this column.  In LDS, not in per-thread arrays: the recurrence
    // indexes them with run-time row numbers, and a register array indexed that way lands in scratch memory
    // (the first version of this kernel: 146 VGPRs + 160 B of scratch per lane).
    extern __shared__ __attribute__((aligned(16))) uint8_t prefix_smem[];  // (M2 + M) rows of blockDim.x samples
    const int bw = blockDim.x;   // 128 for the shallow prefixes of the usual scales; fewer columns per block for scales close
                                 // to 1, whose prefix is hundreds of rows deep (K ~ a*S/(S-1))
    T* hs = (T*)prefix_smem;     // [M2][bw]
    T* os = hs + (size_t)M2 * bw; // [M][bw]
    const int tl = threadIdx.x;

    {   // horizontal pass of the rows the prefix reads (full_TB.h:55-65)
        const int xx = j / C, c = j - xx * C;
        const int first = t.h_first[xx];
        int idx[TAPS];
        double w[TAPS];
#pragma unroll
        for (int k = 0; k < TAPS; k++) {
            int i = first + k;
            i = i < 0 ? 0 : (i > g.in_w - 1 ? g.in_w - 1 : i);  // weight is 0 there
            idx[k] = i * C + c;
            w[k] = t.h_w[(size_t)xx * TAPS + k];
        }
#pragma unroll 4
        for (int r = 0; r < M2; r++) {
            const T* row = (const T*)(in_f + (size_t)(r - g.in_row0) * g.in_pitch);
            T v[TAPS];
#pragma unroll
            for (int k = 0; k < TAPS; k++) v[k] = row[idx[k]];
            double sum = 0;
#pragma unroll
            for (int k = 0; k < TAPS; k++) sum += (double)v[k] * w[k];
            hs[r * bw + tl] = store_convert<T>(sum);
        }
    }
    // full_TB.h:69-76, xx descending: a tap at row i > xx sees the value already written there
    for (int xx = M - 1; xx >= 0; xx--) {
        const int first = t.v_first[xx];
        const double* wv = t.v_w + (size_t)xx * TAPS;
        double sum = 0;
#pragma unroll
        for (int k = 0; k < TAPS; k++) {
            int i = first + k;
            i = i < 0 ? 0 : (i > g.in_h - 1 ? g.in_h - 1 : i);  // weight 0 outside
            const T v = i > xx ? os[i * bw + tl] : hs[i * bw + tl];
            sum += (double)v * wv[k];
        }
        os[xx * bw + tl] = store_convert<T>(sum);
    }
    for (int xx = 0; xx < K; xx++) {
        if (xx < g.out_row0 || xx >= g.out_row0 + g.out_rows) continue;
        T* orow = (T*)(out_f + (size_t)(xx - g.out_row0) * g.out_pitch);
        orow[j] = os[xx * bw + tl];
    }
}

// ---- the same rows at ANY depth: the recurrence as a stream ---------------------------------------------------------------
// k_prefix keeps every H row and every written row of its column in LDS ([M2 + M][bw]), which caps the depth of the in-place
// prefix (S = 1: the whole frame height; S -> 1: K ~ a * S / (S - 1) rows).  But full_TB.h:67-77 walks xx downwards and row xx
// only ever reads
//   * H rows first(xx) .. first(xx) + 2a - 1 (those <= xx), and first(xx) = floor(xx / S) - a + 1 never increases as xx falls;
//   * written rows i in (xx, xx + a]   (first + 2a - 1 = floor(xx / S) + a <= xx + a),
// so a ring of 16 H rows (the 2a-row window + a chunk of 8 rows computed ahead, all their loads in flight together) and a ring
// of 8 written rows per column carry the whole recurrence: 24 rows of LDS whatever M is.  Rows < K are stored as they appear.
template <typename T, int TAPS>
__global__ __launch_bounds__(128) void k_prefix_stream(FrameGeom g, TapTables t, int K, int M) {  // blockDim.x == 128
    constexpr int HR = 16, OR = 8, CH = 8;
    static_assert(TAPS <= 8 && CH + TAPS - 1 <= HR && TAPS / 2 + 1 <= OR, "ring sizes");
    const int C = g.channels;
    const int samples_w = g.out_w * C;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int frame = blockIdx.y;
    if (j >= samples_w) return;
    const uint8_t* in_f = g.in + (size_t)frame * g.in_frame_stride;
    uint8_t* out_f = g.out + (size_t)frame * g.out_frame_stride;
    __shared__ T hs[HR][128];   // H row r at [r & 15]
    __shared__ T os[OR][128];   // written row i at [i & 7]
    const int tl = threadIdx.x;
    const int xx_h = j / C, c = j - xx_h * C;
    int idx[TAPS];
    double w[TAPS];
    {
        const int first = t.h_first[xx_h];
#pragma unroll
        for (int k = 0; k < TAPS; k++) {
            int i = first + k;
            i = i < 0 ? 0 : (i > g.in_w - 1 ? g.in_w - 1 : i);  // weight is 0 there
            idx[k] = i * C + c;
            w[k] = t.h_w[(size_t)xx_h * TAPS + k];
        }
    }
    auto vfirst = [&](int xx) { return t.v_first[xx]; };
    int h_lo = vfirst(M - 1) + TAPS;            // H rows [h_lo, ...) of the window are in the ring (none yet)
    if (h_lo > g.in_h) h_lo = g.in_h;
    for (int xx = M - 1; xx >= 0; xx--) {
        const int first = vfirst(xx);
        const int need = first < 0 ? 0 : first;
        while (h_lo > need) {                    // (uniform per column only -- every thread walks its own column; no barriers)
            const int r0 = h_lo - CH < 0 ? 0 : h_lo - CH;
            T v[CH][TAPS];
#pragma unroll
            for (int rr = 0; rr < CH; rr++) {    // horizontal pass of a chunk of rows (full_TB.h:55-65): all loads in flight
                const int r = r0 + rr < h_lo ? r0 + rr : h_lo - 1;
                const T* row = (const T*)(in_f + (size_t)(r - g.in_row0) * g.in_pitch);
#pragma unroll
                for (int k = 0; k < TAPS; k++) v[rr][k] = row[idx[k]];
            }
#pragma unroll
            for (int rr = 0; rr < CH; rr++) {
                if (r0 + rr >= h_lo) break;
                double sum = 0;
#pragma unroll
                for (int k = 0; k < TAPS; k++) sum += (double)v[rr][k] * w[k];
                hs[(r0 + rr) & (HR - 1)][tl] = store_convert<T>(sum);
            }
            h_lo = r0;
        }
        // full_TB.h:69-76: a tap at row i > xx sees the value already written there
        const double* wv = t.v_w + (size_t)xx * TAPS;
        double sum = 0;
#pragma unroll
        for (int k = 0; k < TAPS; k++) {
            int i = first + k;
            i = i < 0 ? 0 : (i > g.in_h - 1 ? g.in_h - 1 : i);  // weight 0 outside
            const T v = i > xx ? os[i & (OR - 1)][tl] : hs[i & (HR - 1)][tl];
            sum += (double)v * wv[k];
        }
        const T o = store_convert<T>(sum);
        os[xx & (OR - 1)][tl] = o;
        if (xx < K && xx >= g.out_row0 && xx < g.out_row0 + g.out_rows)
            ((T*)(out_f + (size_t)(xx - g.out_row0) * g.out_pitch))[j] = o;
    }
}

// ---- the same rows for INTEGER scales, entirely in registers -------------------------------------------------------------
// For an integer scale the tap rows are known at compile time (first = floor(y / S) - A + 1), so the recurrence unrolls
// completely: hs[] / os[] are registers with static indices, no LDS at all.  That matters for more than speed: with no LDS
// and < 80 VGPRs these workgroups fit on a CU BESIDE four resident marching workgroups (which own all 160 KiB of LDS), so
// launched on a second stream just before the marching kernel they run inside its first microseconds instead of as a
// serial 8.5 us kernel + 2.5 us gap behind it (profiles/README.md, round 2).
constexpr int prefix_last_tap(int o, int S, int A) { return o / S + A; }
constexpr int prefix_K(int S, int A) {
    int k = 0;
    for (int o = 0; o < 4 * A * S + 8; o++)
        if (prefix_last_tap(o, S, A) > o) k = o + 1;
    return k;
}
constexpr int prefix_M(int S, int A) {
    int m = prefix_K(S, A);
    for (int o = 0; o < prefix_K(S, A); o++)
        if (prefix_last_tap(o, S, A) + 1 > m) m = prefix_last_tap(o, S, A) + 1;
    return m;
}
constexpr int prefix_M2(int S, int A) {
    int m2 = 0;
    for (int o = 0; o < prefix_M(S, A); o++)
        if (prefix_last_tap(o, S, A) + 1 > m2) m2 = prefix_last_tap(o, S, A) + 1;
    return m2;
}

template <typename T, int S, int A>
__global__ __launch_bounds__(128) void k_prefix_reg(FrameGeom g, TapTables t) {
    constexpr int TAPS = 2 * A, K = prefix_K(S, A), M = prefix_M(S, A), M2 = prefix_M2(S, A);
    const int C = g.channels;
    const int samples_w = g.out_w * C;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int frame = blockIdx.y;
    if (j >= samples_w) return;
    const uint8_t* in_f = g.in + (size_t)frame * g.in_frame_stride;
    uint8_t* out_f = g.out + (size_t)frame * g.out_frame_stride;
    T hs[M2], os[M];
    {   // horizontal pass of the rows the prefix reads (full_TB.h:55-65); the host guarantees in_h >= M2, in_row0 == 0
        const int xx = j / C, c = j - xx * C;
        const int first = t.h_first[xx];
        int idx[TAPS];
        double w[TAPS];
#pragma unroll
        for (int k = 0; k < TAPS; k++) {
            int i = first + k;
            i = i < 0 ? 0 : (i > g.in_w - 1 ? g.in_w - 1 : i);  // weight is 0 there
            idx[k] = i * C + c;
            w[k] = t.h_w[(size_t)xx * TAPS + k];
        }
#pragma unroll
        for (int r = 0; r < M2; r++) {
            const T* row = (const T*)(in_f + (size_t)r * g.in_pitch);
            T v[TAPS];
#pragma unroll
            for (int k = 0; k < TAPS; k++) v[k] = row[idx[k]];
            double sum = 0;
#pragma unroll
            for (int k = 0; k < TAPS; k++) sum += (double)v[k] * w[k];
            hs[r] = store_convert<T>(sum);
        }
    }
    // full_TB.h:69-76, xx descending: a tap at row i > xx sees the value already written there.  Rows < 0 carry weight 0
    // in the table; rows > in_h - 1 cannot occur (in_h >= M2).
#pragma unroll
    for (int xx = M - 1; xx >= 0; xx--) {
        constexpr int dummy = 0;
        (void)dummy;
        const int first = xx / S - A + 1;
        const double* wv = t.v_w + (size_t)xx * TAPS;
        double sum = 0;
#pragma unroll
        for (int k = 0; k < TAPS; k++) {
            const int i = first + k < 0 ? 0 : first + k;
            const T v = i > xx ? os[i < M ? i : M - 1] : hs[i < M2 ? i : M2 - 1];
            sum += (double)v * wv[k];
        }
        os[xx] = store_convert<T>(sum);
    }
#pragma unroll
    for (int xx = 0; xx < K; xx++) {
        if (xx < g.out_row0 || xx >= g.out_row0 + g.out_rows) continue;
        T* orow = (T*)(out_f + (size_t)(xx - g.out_row0) * g.out_pitch);
        orow[j] = os[xx];
    }
}

}  // namespace lz
