// lanczos_rational.hpp -- k_rat: the f32 tile kernel for RATIONAL scales N/D > 1 (4/3, 3/2, 5/2, 5/3 ...), any channel count,
// a = 2..4, u8 / u16.  (Integer scales have the marching kernel; everything this kernel refuses goes to the f64 k_generic.)
//
// The reference derives SCALE_N/SCALE_D with gcd() (lanczos.h:108-114, stb.cpp:9-12) and evaluates x = xx / SCALE in double
// (full_TB.h:57,70): for a non-integer scale neither the tap positions nor the weights are exactly periodic in the output
// index (x is a rounded quotient).  So nothing here is derived from "phases": every output index carries its OWN first tap and
// its own 2a weights, tabulated on the host with the reference's expressions -- double for the exact chains, rounded to f32
// for the fast ones -- and a flag for the indices where x is an exact integer (weights {.., 3e-17, 1, 3e-17, ..}: SURVEY.md Q4).
//
// One workgroup = 128 dword columns (512 B of an output row) x 32 output rows:
//   1. LOAD   the input rows / byte columns the tile's windows touch -> LDS (dword loads when the rows allow it)
//   2. HPASS  thread = one dword column of the H rows: per sample a 2a-tap f32 chain from the LDS row (weights in registers);
//             a sum within eps of an integer is redone with the reference's f64 chain (full_TB.h:58-63), integer-phase samples
//             are copies unless 1 <= v0 <= vlim (then the f64 chain decides); truncated integers go to LDS (full_TB.h:63)
//   3. VPASS  thread = the same dword column of the output: per output row one dword of each of the 2a H rows, f32 chains,
//             one dword store.  LSB1: floor(sum + eps) (within 1 LSB); EXACT: undecided samples redone in f64.
// Rows < K (the in-place prefix, full_TB.h:67-77) are left to k_prefix, as with every other kernel family.
#pragma once
#include <cmath>
#include <mutex>

#include "lanczos_fast.hpp"
#include "lanczos_kernels_common.hpp"
#include "lanczos_taps.hpp"

namespace lz {

constexpr int kRatCols = 128;                            // dword columns per tile
constexpr int kRatGroups = 4;                            // thread groups sharing a tile: group g takes rows g, g+4, ...
constexpr int kRatThreads = kRatCols * kRatGroups;       // 8 waves per workgroup (a 50 KB tile with 2 waves hid no latency)
constexpr int kRatTileRowBytes = kRatCols * 4;           // output bytes per tile row
constexpr int kRatTileH = 32;                            // output rows per tile
constexpr int kRatInPitch = kRatTileRowBytes + 2 * kMaxA * 4 * 2 + 64;  // input bytes a tile row can touch (scale > 1), padded
constexpr int kRatMaxRows = kRatTileH + 2 * kMaxA + 4;   // input rows a tile can touch (scale > 1)

struct RatTables {
    const float* h_wf;      // [out_w][2a] f32 weights
    const float* v_wf;      // [out_h][2a]
    const uint8_t* h_int;   // [out_w] 1: x is an exact integer there
    const uint8_t* v_int;   // [out_h]
    float bias;             // eps: f32 chain error bound over every index of both axes
    float vbias_rne;        // eps - 0.5
    float near2;            // 2 eps
    int vlim;               // integer-phase flip limit (0: never)
    int tight;              // 1: v0 provably survives when both +-2 neighbours are <= 2*v0 (integer_phase_tight)
};

template <typename T, int TAPS, bool EXACT>
__global__ __launch_bounds__(kRatThreads) void k_rat(FrameGeom g, TapTables t, RatTables rt) {
    constexpr int SB = (int)sizeof(T), VEC = 4 / SB, A = TAPS / 2;
    constexpr float MAXV = SB == 1 ? 255.0f : 65535.0f;
    __shared__ __attribute__((aligned(16))) uint8_t tin[kRatMaxRows * kRatInPitch];
    __shared__ __attribute__((aligned(16))) uint32_t hbuf[kRatMaxRows * kRatCols];

    const int C = g.channels, tid = threadIdx.x % kRatCols, grp = threadIdx.x / kRatCols, tidx = threadIdx.x;
    const int samples_w = g.out_w * C;
    const int tiles_x = (samples_w * SB + kRatTileRowBytes - 1) / kRatTileRowBytes;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, frame = blockIdx.y;
    const int y0 = g.out_row0 + ty * kRatTileH;
    const int y1 = y0 + kRatTileH < g.out_row0 + g.out_rows ? y0 + kRatTileH : g.out_row0 + g.out_rows;
    if (y1 <= g.skip_rows) return;  // the whole tile belongs to the prefix kernel
    const int js0 = tx * (kRatTileRowBytes / SB);                       // first sample column of the tile
    const int js1 = js0 + kRatTileRowBytes / SB < samples_w ? js0 + kRatTileRowBytes / SB : samples_w;  // one past the last
    // input window of the tile: pixels p_lo..p_hi, rows r_lo..r_hi (clamped to the image: taps outside carry weight 0)
    int p_lo = t.h_first[js0 / C], p_hi = t.h_first[(js1 - 1) / C] + TAPS - 1;
    p_lo = p_lo < 0 ? 0 : p_lo;
    p_hi = p_hi > g.in_w - 1 ? g.in_w - 1 : p_hi;
    int r_lo = t.v_first[y0], r_hi = t.v_first[y1 - 1] + TAPS - 1;
    r_lo = r_lo < 0 ? 0 : r_lo;
    r_hi = r_hi > g.in_h - 1 ? g.in_h - 1 : r_hi;
    const int nrows = r_hi - r_lo + 1;
    const int b_lo = p_lo * C * SB, nbytes = (p_hi - p_lo + 1) * C * SB;   // byte window of an input row
    const uint8_t* in_f = g.in + (size_t)frame * g.in_frame_stride;
    uint8_t* out_f = g.out + (size_t)frame * g.out_frame_stride;

    // ------------------------------------------------------------------ 1. LOAD
    {
        const int a_lo = b_lo & ~3;                      // dword-aligned start inside the row
        const int ndw = (b_lo + nbytes - a_lo + 3) / 4;
        const bool dwords = (g.in_pitch & 3) == 0 && (((uintptr_t)in_f) & 3) == 0;
        for (int i = tidx; i < nrows * ndw; i += kRatThreads) {
            const int rr = i / ndw, dw = i - rr * ndw;
            const uint8_t* rowp = in_f + (size_t)(r_lo + rr - g.in_row0) * g.in_pitch;
            const int gb = a_lo + 4 * dw;
            uint32_t v = 0;
            if (dwords && gb + 4 <= g.in_pitch) {
                v = *(const uint32_t*)(rowp + gb);
            } else {
#pragma unroll
                for (int b = 0; b < 4; b++)
                    if (gb + b < g.in_pitch) v |= (uint32_t)rowp[gb + b] << (8 * b);
            }
            *(uint32_t*)(tin + rr * kRatInPitch + 4 * dw) = v;   // tin column 0 <-> row byte a_lo
        }
        __syncthreads();
        // ------------------------------------------------------------------ 2. HPASS
        const int tb0 = a_lo;                              // row byte of tin column 0
        uint32_t packed_first = 0;
        (void)packed_first;
        int off[VEC][TAPS];                                // tin byte offset of tap k of sample e (clamped into the window)
        float wf[VEC][TAPS];
        int xxs[VEC];
        bool isint[VEC], live[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            const int j = js0 + tid * VEC + e;
            live[e] = j < js1;
            const int jj = live[e] ? j : js1 - 1;
            const int xx = jj / C, c = jj - xx * C;
            xxs[e] = xx;
            const int first = t.h_first[xx];
            isint[e] = rt.h_int[xx] != 0;
#pragma unroll
            for (int k = 0; k < TAPS; k++) {
                int p = first + k;
                p = p < p_lo ? p_lo : (p > p_hi ? p_hi : p);   // weight is 0 outside the image
                off[e][k] = (p * C + c) * SB - tb0;
                wf[e][k] = rt.h_wf[(size_t)xx * TAPS + k];
            }
        }
        for (int rr = grp; rr < nrows; rr += kRatGroups) {
            const uint8_t* rowl = tin + rr * kRatInPitch;
            uint32_t packed = 0;
#pragma unroll
            for (int e = 0; e < VEC; e++) {
                unsigned v[TAPS];
#pragma unroll
                for (int k = 0; k < TAPS; k++) v[k] = *(const T*)(rowl + off[e][k]);
                unsigned res;
                bool exact_chain;
                if (isint[e]) {
                    res = v[A - 1];                                       // L(0) = 1, the other taps ~1e-17
                    exact_chain = rt.vlim > 0 && res >= 1u && res <= (unsigned)rt.vlim;
                    if (A >= 3 && rt.tight) exact_chain = exact_chain && (v[A - 3] > 2u * res || v[A + 1] > 2u * res);
                } else {
                    float acc = rt.bias;
#pragma unroll
                    for (int k = 0; k < TAPS; k++) acc = __builtin_fmaf(wf[e][k], (float)v[k], acc);
                    const float xc = __builtin_amdgcn_fmed3f(acc, 0.5f, MAXV + 0.5f);  // below 1 / above max the store clamps
                    const float fl = __builtin_floorf(xc);
                    exact_chain = (xc - fl) < rt.near2;
                    res = (unsigned)fl;
                }
                if (exact_chain) {                                        // full_TB.h:58-63, ascending taps, separate mul/add
                    const double* w = t.h_w + (size_t)xxs[e] * TAPS;
                    double sum = 0;
#pragma unroll
                    for (int k = 0; k < TAPS; k++) sum += (double)v[k] * w[k];
                    res = store_convert<T>(sum);
                }
                packed |= res << (8 * SB * e);
            }
            hbuf[rr * kRatCols + tid] = packed;
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ 3. VPASS
    const unsigned col_b = (unsigned)(tx * kRatTileRowBytes + tid * 4);
    if (col_b >= (unsigned)(samples_w * SB)) return;
    const float vbias = (SB == 1 && !EXACT) ? rt.vbias_rne : rt.bias;
    for (int y = (y0 > g.skip_rows ? y0 : g.skip_rows) + grp; y < y1; y += kRatGroups) {
        const int first = t.v_first[y];
        uint32_t rw[TAPS];
#pragma unroll
        for (int k = 0; k < TAPS; k++) {
            int r = first + k;
            r = r < r_lo ? r_lo : (r > r_hi ? r_hi : r);                 // weight is 0 outside the image
            rw[k] = hbuf[(r - r_lo) * kRatCols + tid];
        }
        uint32_t packed = 0;
        bool undecided = false;
        if (rt.v_int[y]) {
            packed = rw[A - 1];
            if (EXACT && rt.vlim > 0) {
#pragma unroll
                for (int e = 0; e < VEC; e++) {
                    const unsigned sm = SB == 1 ? 0xffu : 0xffffu;
                    const unsigned c0 = (packed >> (8 * SB * e)) & sm;
                    bool fl = c0 >= 1u && c0 <= (unsigned)rt.vlim;
                    if (A >= 3 && rt.tight)
                        fl = fl && (((rw[A - 3] >> (8 * SB * e)) & sm) > 2u * c0 || ((rw[A + 1] >> (8 * SB * e)) & sm) > 2u * c0);
                    undecided |= fl;
                }
            }
        } else {
            const float* wv = rt.v_wf + (size_t)y * TAPS;
#pragma unroll
            for (int e = 0; e < VEC; e++) {
                float acc = vbias;
#pragma unroll
                for (int k = 0; k < TAPS; k++)
                    acc = __builtin_fmaf(wv[k], (float)((rw[k] >> (8 * SB * e)) & (SB == 1 ? 0xffu : 0xffffu)), acc);
                if (SB == 1 && !EXACT) {
                    packed = __builtin_amdgcn_cvt_pk_u8_f32(acc, e, packed);  // floor(sum + eps): see k_march
                } else {
                    const float xc = __builtin_amdgcn_fmed3f(acc, 0.5f, MAXV + 0.5f);
                    const float fl = __builtin_floorf(xc);
                    if (EXACT) undecided |= (xc - fl) < rt.near2;
                    packed |= (unsigned)fl << (8 * SB * e);
                }
            }
        }
        if (EXACT && undecided) {                                         // full_TB.h:71-75 for this dword
            const double* wvd = t.v_w + (size_t)y * TAPS;
            packed = 0;
#pragma unroll
            for (int e = 0; e < VEC; e++) {
                double sum = 0;
#pragma unroll
                for (int k = 0; k < TAPS; k++) sum += (double)((rw[k] >> (8 * SB * e)) & (SB == 1 ? 0xffu : 0xffffu)) * wvd[k];
                packed |= (unsigned)store_convert<T>(sum) << (8 * SB * e);
            }
        }
        uint8_t* orow = out_f + (size_t)(y - g.out_row0) * g.out_pitch;
        if (col_b + 4 <= (unsigned)(samples_w * SB)) {
            *(uint32_t*)(orow + col_b) = packed;
        } else {                                                          // ragged last dword of the row
            for (unsigned b = 0; col_b + b < (unsigned)(samples_w * SB); b++) orow[col_b + b] = (uint8_t)(packed >> (8 * b));
        }
    }
}

// ---------------------------------------------------------------------------------- host side
struct RatHost {
    std::vector<float> h_wf, v_wf;
    std::vector<uint8_t> h_int, v_int;
    float bias = 0, vbias_rne = 0, near2 = 0;
    int vlim = 0;
    int tight = 0;
    bool ok = false;
};

// x == floor(x) exactly, with the reference's double expression (full_TB.h:57)
inline void rat_int_flags(int out_n, int scale_n, int scale_d, std::vector<uint8_t>* f) {
    f->assign(out_n, 0);
    const double SCALE = (double)scale_n / scale_d;
    for (int o = 0; o < out_n; o++) {
        const double x = (double)o / SCALE;
        (*f)[o] = x == std::floor(x) ? 1 : 0;
    }
}

inline void rat_prepare(const lanczos_desc& d, const AxisTaps& H, const AxisTaps& V, RatHost* r) {
    const int taps = 2 * d.a;
    const double maxv = d.bytes_per_sample == 1 ? 255.0 : 65535.0;
    r->ok = false;
    rat_int_flags(d.out_w, d.scale_n, d.scale_d, &r->h_int);
    rat_int_flags(d.out_h, d.scale_n, d.scale_d, &r->v_int);
    r->h_wf.resize(H.w.size());
    r->v_wf.resize(V.w.size());
    for (size_t i = 0; i < H.w.size(); i++) r->h_wf[i] = (float)H.w[i];
    for (size_t i = 0; i < V.w.size(); i++) r->v_wf[i] = (float)V.w[i];
    int order[kMaxTaps];
    for (int k = 0; k < kMaxTaps; k++) order[k] = k;   // the kernel adds its taps in ascending order
    double eps = 0;
    int vlim = 0;
    bool tight = true, any_full = false;
    auto scan = [&](const AxisTaps& ax, const std::vector<uint8_t>& fl) {
        for (int o = 0; o < ax.out_n; o++) {
            const double* w = &ax.w[(size_t)o * taps];
            if (fl[o]) {
                bool full = ax.first[o] >= 0 && ax.first[o] + taps - 1 <= ax.in_n - 1;
                if (full) {
                    const int v = integer_phase_flip_limit(w, d.a, (int)maxv);
                    if (v > vlim) vlim = v;
                    tight = tight && integer_phase_tight(w, d.a, maxv);
                    any_full = true;
                } else {
                    // clipped integer-phase window: some tiny taps are missing -- the limit of the full window covers it
                    // (dropping terms cannot make a larger excursion); handled by the full-window indices of the same axis
                }
                continue;
            }
            const double e = f32_chain_error_bound_ordered(w, order, taps, maxv);
            if (e > eps) eps = e;
        }
    };
    scan(H, r->h_int);
    scan(V, r->v_int);
    if (vlim == 0) {  // no full-window integer-phase index on either axis (tiny image): be conservative
        bool any = false;
        for (uint8_t f : r->h_int) any |= f != 0;
        for (uint8_t f : r->v_int) any |= f != 0;
        if (any) vlim = (int)maxv;
    }
    if (eps <= 0 || eps > 0.2) return;
    r->bias = (float)eps;
    r->vbias_rne = (float)eps - 0.5f;
    r->near2 = (float)(2.0 * eps) * 1.0001f;
    r->vlim = vlim;
    r->tight = (tight && any_full) ? 1 : 0;
    r->ok = true;
}

inline bool rat_supports(const lanczos_desc& d, const FrameGeom& g) {
    if (d.scale_d == 1) return false;                       // integer scales: the marching / tile kernels
    if (d.scale_n >= 2 * d.scale_d + d.scale_d) {}          // (any ratio > 1 is fine)
    if (g.out_pitch % 4 != 0 || (((uintptr_t)g.out) & 3) != 0 || (g.out_frame_stride & 3) != 0) return false;
    // tile window bounds assume scale > 1: input span of a tile <= its output span + 2a
    return d.scale_n > d.scale_d && g.frames <= 65535;
}



// ======================================================================================================================
// k_ratp -- the same job for the rational scales that are exactly PERIODIC in this frame: the host has checked, index by
// index, that first[o] = D*(o/N) + (o%N)*D/N - a + 1 and that x is an exact integer exactly where o % N == 0 (true for
// 4/3, 3/2, 5/2, 5/4 ...; where the rounding of x = o / SCALE breaks it -- 5/3 -- the table-driven k_rat above serves).
// Then every tap offset is a compile-time constant, as in the integer-scale kernels:
//   HPASS  a thread owns a UNIT of UP periods of one row: UP*D input pixels -> UP*N output pixels; the window is read as
//          aligned dwords, every byte converted once, chains with per-phase f32 weights; integer-phase samples are copies
//   VPASS  a thread owns a dword column and walks the tile period by period with a 2a-row register window: D new rows
//          in, N output rows out, every ring sample converted once
// Exactness exactly as in k_rat: eps-window test -> the reference's f64 chain with the per-index table weights.
template <typename T, int C_, int N_, int D_, int A_>
struct RatPCfg {
    static constexpr int C = C_, N = N_, D = D_, A = A_, TAPS = 2 * A_, SB = (int)sizeof(T), VEC = 4 / SB;
    static constexpr int up_min() {   // periods per H unit: its input AND its output are whole dwords
        for (int u = 1; u <= 4; u++)
            if ((u * N_ * C_ * (int)sizeof(T)) % 4 == 0 && (u * D_ * C_ * (int)sizeof(T)) % 4 == 0) return u;
        return 4;
    }
    static constexpr int UP = up_min();
    static constexpr int P_IN = UP * D, P_OUT = UP * N;            // pixels in / out per unit
    static constexpr int UOD = P_OUT * C * SB / 4;                 // output dwords per unit
    static constexpr int NT = 512;                                  // 8 waves: two on every SIMD
#ifndef LZ_RATP_ROWS
#define LZ_RATP_ROWS 48
#endif
#ifndef LZ_RATP_NVG
#define LZ_RATP_NVG 3
#endif
    static constexpr int TP = LZ_RATP_ROWS / N < 1 ? 1 : LZ_RATP_ROWS / N;  // periods per tile vertically
    static constexpr int TH = TP * N;                              // output rows per tile
    static constexpr int NR = TP * D + TAPS - 1;                   // H rows per tile
    static constexpr int NVG = TP % LZ_RATP_NVG == 0 ? LZ_RATP_NVG : (TP % 2 == 0 ? 2 : 1);  // V thread groups: each walks 1/NVG of the tile's periods
    static constexpr int nuw_pick() {   // units per tile row: the H pass is ONE round of the workgroup, the V groups fit
        int n = NT / NR;
        while (n > 1 && n * UOD * NVG > NT) n--;
        return n < 1 ? 1 : n;
    }
    static constexpr int NUW = nuw_pick();
    static constexpr int NVT = NUW * UOD;                          // dword columns per tile row
    static_assert(TP % NVG == 0, "V groups split the periods evenly");
    static constexpr int WIN_PX = P_IN + TAPS - 1;
    static constexpr int LPB = ((A - 1) * C * SB + 15) / 16 * 16;  // left pad bytes of a tile row in LDS
    static constexpr int IN_PITCH = (LPB + NUW * P_IN * C * SB + A * C * SB + 15) / 16 * 16;
    static constexpr int MIS = (LPB - (A - 1) * C * SB) % 4;       // byte offset of a unit's window in its first dword
    static constexpr int NW = (MIS + WIN_PX * C * SB + 3) / 4;     // dwords per unit window
    static constexpr int WIN_DW0 = (LPB - (A - 1) * C * SB - MIS) / 4;
    static constexpr int UNIT_IN_B = P_IN * C * SB;
    static constexpr int H_PITCH = NVT * 4;
    static constexpr int LDS_BYTES = NR * IN_PITCH + NR * H_PITCH;
    static constexpr float MAXV = SB == 1 ? 255.0f : 65535.0f;
    static constexpr unsigned SMASK = SB == 1 ? 0xffu : 0xffffu;
    static_assert(NVT * NVG <= NT, "one V thread per dword column and group");
};

template <typename T, int C, int N, int D, int A, bool EXACT>
__global__ __launch_bounds__((RatPCfg<T, C, N, D, A>::NT)) void k_ratp(FrameGeom g, TapTables t, RatTables rt, const float* __restrict__ phase_w) {
    using K = RatPCfg<T, C, N, D, A>;
    constexpr int TAPS = K::TAPS, SB = K::SB, VEC = K::VEC;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* tin = smem;
    uint32_t* hbuf = (uint32_t*)(smem + K::NR * K::IN_PITCH);
    const int tid = threadIdx.x;
    const int tile_px_out = K::NUW * K::P_OUT;
    const int tiles_x = (g.out_w + tile_px_out - 1) / tile_px_out;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, frame = blockIdx.y;
    // tiles are aligned to whole periods in FULL-frame rows
    const int y_tile = (g.out_row0 / K::TH + ty) * K::TH;
    const int y_begin = y_tile > g.out_row0 ? y_tile : g.out_row0;
    int y_end = y_tile + K::TH;
    if (y_end > g.out_row0 + g.out_rows) y_end = g.out_row0 + g.out_rows;
    const int y_first = y_begin > g.skip_rows ? y_begin : g.skip_rows;
    if (y_end <= y_first) return;
    const int m0 = y_tile / N * D;          // input row of the tile's first period
    const int r_lo = m0 - A + 1;            // LDS row 0 <-> input row r_lo (may be negative)
    const int P0 = tx * K::NUW * K::P_IN;   // first input pixel owned by the tile
    const int row_bytes = g.in_w * C * SB;
    const uint8_t* in_f = g.in + (size_t)frame * g.in_frame_stride;
    uint8_t* out_f = g.out + (size_t)frame * g.out_frame_stride;

    // ------------------------------------------------------------------ 1. LOAD (zero outside the image: a dropped tap)
    {
        const int tile_gb0 = P0 * C * SB - K::LPB;
        const int gr_min = g.in_row0 > 0 ? g.in_row0 : 0;
        const int gr_max = (g.in_row0 + g.in_rows < g.in_h ? g.in_row0 + g.in_rows : g.in_h) - 1;
        constexpr int DWPR = K::IN_PITCH / 4;
        const bool aligned = (g.in_pitch & 3) == 0 && (((uintptr_t)in_f) & 3) == 0;
        for (int i = tid; i < K::NR * DWPR; i += K::NT) {
            const int rr = i / DWPR, dw = i - rr * DWPR;
            const int gr = r_lo + rr, gb = tile_gb0 + 4 * dw;
            uint32_t v = 0;
            if (gr >= gr_min && gr <= gr_max && gb + 4 > 0 && gb < row_bytes) {
                const uint8_t* rp = in_f + (size_t)(gr - g.in_row0) * g.in_pitch;
                if (aligned && gb >= 0 && gb + 4 <= row_bytes) {
                    v = *(const uint32_t*)(rp + gb);
                } else {
#pragma unroll
                    for (int b = 0; b < 4; b++)
                        if (gb + b >= 0 && gb + b < row_bytes) v |= (uint32_t)rp[gb + b] << (8 * b);
                }
            }
            ((uint32_t*)tin)[i] = v;
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ 2. HPASS
    {
        // phases r and N - r are mirror images (L is even): only phases 1 .. N/2 live in registers, phase N - r reads them
        // back to front.  (ratp_prepare builds the f32 phase table that way and prices the deviation into eps.)
        constexpr int NPH = N / 2;
        float wf[NPH + 1][TAPS];
#pragma unroll
        for (int r = 1; r <= NPH; r++)
#pragma unroll
            for (int k = 0; k < TAPS; k++) wf[r][k] = phase_w[r * kMaxTaps + k];
        for (int idx = tid; idx < K::NR * K::NUW; idx += K::NT) {
            const int row = idx / K::NUW, u = idx - row * K::NUW;
            uint32_t wd[K::NW];
            const uint32_t* wp = (const uint32_t*)(tin + row * K::IN_PITCH) + K::WIN_DW0 + u * (K::UNIT_IN_B / 4);
            static_assert(K::UNIT_IN_B % 4 == 0, "a unit's input pixels are whole dwords");
#pragma unroll
            for (int i = 0; i < K::NW; i++) wd[i] = wp[i];
            auto wsample = [&](int p, int c) -> unsigned {   // sample (window pixel p, channel c): static bit field of wd
                const int b = K::MIS + (p * C + c) * SB;
                return (wd[b >> 2] >> (8 * (b & 3))) & K::SMASK;
            };
            uint32_t ow[K::UOD];
#pragma unroll
            for (int i = 0; i < K::UOD; i++) ow[i] = 0;
            const int xx0 = (tx * K::NUW + u) * K::P_OUT;   // first output pixel of the unit
#pragma unroll
            for (int c = 0; c < C; c++) {
                float fch[K::WIN_PX];
#pragma unroll
                for (int p = 0; p < K::WIN_PX; p++) fch[p] = (float)wsample(p, c);
#pragma unroll
                for (int q = 0; q < K::P_OUT; q++) {
                    const int p = (q * D) / N, ph = q % N;      // window pixel of the first tap, phase
                    unsigned res;
                    bool exact_chain;
                    if (ph == 0) {
                        res = wsample(p + A - 1, c);
                        exact_chain = rt.vlim > 0 && res >= 1u && res <= (unsigned)rt.vlim;
                        if (A >= 3 && rt.tight)
                            exact_chain = exact_chain && (wsample(p + A - 3, c) > 2u * res || wsample(p + A + 1, c) > 2u * res);
                    } else {
                        float acc = rt.bias;
#pragma unroll
                        for (int k = 0; k < TAPS; k++)
                            acc = __builtin_fmaf(ph <= NPH ? wf[ph][k] : wf[N - ph][TAPS - 1 - k], fch[p + k], acc);
                        const float xc = __builtin_amdgcn_fmed3f(acc, 0.5f, K::MAXV + 0.5f);
                        const float fl = __builtin_floorf(xc);
                        exact_chain = (xc - fl) < rt.near2;
                        res = (unsigned)fl;
                    }
                    if (exact_chain) {   // full_TB.h:58-63 with the per-index weights (taps outside the image: weight 0)
                        const int xx = xx0 + q < g.out_w ? xx0 + q : g.out_w - 1;
                        const double* w = t.h_w + (size_t)xx * TAPS;
                        double sum = 0;
#pragma unroll
                        for (int k = 0; k < TAPS; k++) sum += (double)wsample(p + k, c) * w[k];
                        res = store_convert<T>(sum);
                    }
                    const int o = q * C + c;
                    ow[o / VEC] |= res << (8 * SB * (o % VEC));
                }
            }
            uint32_t* hp = hbuf + row * K::NVT + u * K::UOD;
#pragma unroll
            for (int i = 0; i < K::UOD; i++) hp[i] = ow[i];
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ 3. VPASS
    const int vgrp = tid / K::NVT, vcol = tid - vgrp * K::NVT;
    if (vgrp >= K::NVG) return;
    constexpr int PPG = K::TP / K::NVG;          // periods per V group
    const unsigned col_b = (unsigned)(tx * K::NVT * 4 + vcol * 4);
    if (col_b >= (unsigned)(g.out_w * C * SB)) return;
    const bool whole = col_b + 4 <= (unsigned)(g.out_w * C * SB);
    constexpr int NPHV = N / 2;
    float wfv[NPHV + 1][TAPS];
#pragma unroll
    for (int r = 1; r <= NPHV; r++)
#pragma unroll
        for (int k = 0; k < TAPS; k++) wfv[r][k] = phase_w[r * kMaxTaps + k];
    const float vbias = (SB == 1 && !EXACT) ? rt.vbias_rne : rt.bias;
    float win[TAPS + D][VEC];     // rows base .. base + TAPS + D - 1 of the current period (static indices)
    uint32_t rw[TAPS + D];
    const uint32_t* hcol = hbuf + vcol + vgrp * PPG * D * K::NVT;   // LDS row of this group's first period
    auto unpack = [&](int slot, uint32_t w) {
        rw[slot] = w;
#pragma unroll
        for (int e = 0; e < VEC; e++) win[slot][e] = (float)((w >> (8 * SB * e)) & K::SMASK);
    };
#pragma unroll
    for (int k = 0; k < TAPS - 1; k++) unpack(k, hcol[k * K::NVT]);
    for (int per = 0; per < PPG; per++) {
        // LDS rows per*D .. per*D + TAPS - 2 (of this group) are in slots 0 .. TAPS-2; bring in the period's D new rows
#pragma unroll
        for (int j = 0; j < D; j++) unpack(TAPS - 1 + j, hcol[(per * D + TAPS - 1 + j) * K::NVT]);
#pragma unroll
        for (int r = 0; r < N; r++) {
            const int s0 = (r * D) / N;                    // window slot of the first tap of output row r of the period
            const int y = y_tile + (vgrp * PPG + per) * N + r;
            uint32_t packed = 0;
            bool undecided = false;
            if (r == 0) {
                packed = rw[s0 + A - 1];
                if (EXACT && rt.vlim > 0) {
#pragma unroll
                    for (int e = 0; e < VEC; e++) {
                        const float c0 = win[s0 + A - 1][e];
                        bool fl = c0 >= 1.0f && c0 <= (float)rt.vlim;
                        if (A >= 3 && rt.tight) fl = fl && (win[s0 + A - 3][e] > 2.0f * c0 || win[s0 + A + 1][e] > 2.0f * c0);
                        undecided |= fl;
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < VEC; e++) {
                    float acc = vbias;
#pragma unroll
                    for (int k = 0; k < TAPS; k++)
                        acc = __builtin_fmaf(r <= NPHV ? wfv[r][k] : wfv[N - r][TAPS - 1 - k], win[s0 + k][e], acc);
                    if (SB == 1 && !EXACT) {
                        packed = __builtin_amdgcn_cvt_pk_u8_f32(acc, e, packed);
                    } else {
                        const float xc = __builtin_amdgcn_fmed3f(acc, 0.5f, K::MAXV + 0.5f);
                        const float fl = __builtin_floorf(xc);
                        if (EXACT) undecided |= (xc - fl) < rt.near2;
                        packed |= (unsigned)fl << (8 * SB * e);
                    }
                }
            }
            if (y >= y_first && y < y_end) {
                if (EXACT && undecided) {
                    const double* wvd = t.v_w + (size_t)y * TAPS;
                    packed = 0;
#pragma unroll
                    for (int e = 0; e < VEC; e++) {
                        double sum = 0;
#pragma unroll
                        for (int k = 0; k < TAPS; k++) sum += (double)win[s0 + k][e] * wvd[k];
                        packed |= (unsigned)store_convert<T>(sum) << (8 * SB * e);
                    }
                }
                uint8_t* orow = out_f + (size_t)(y - g.out_row0) * g.out_pitch;
                if (whole) {
                    *(uint32_t*)(orow + col_b) = packed;
                } else {
                    for (unsigned b = 0; col_b + b < (unsigned)(g.out_w * C * SB); b++) orow[col_b + b] = (uint8_t)(packed >> (8 * b));
                }
            }
        }
        // slide: the next period starts D rows further down
#pragma unroll
        for (int k = 0; k < TAPS - 1; k++) {
            rw[k] = rw[k + D];
#pragma unroll
            for (int e = 0; e < VEC; e++) win[k][e] = win[k + D][e];
        }
    }
}

// the instantiated periodic configurations: (sample type, channels, N, D, a)
#define LZ_RATP_CONFIGS(X) \
    X(uint8_t, 3, 4, 3, 3) \
    X(uint8_t, 3, 3, 2, 3) \
    X(uint8_t, 3, 5, 2, 3) \
    X(uint8_t, 3, 5, 4, 3) \
    X(uint8_t, 4, 4, 3, 3) \
    X(uint8_t, 4, 3, 2, 3) \
    X(uint8_t, 1, 4, 3, 3) \
    X(uint8_t, 1, 3, 2, 3) \
    X(uint8_t, 3, 4, 3, 2) \
    X(uint8_t, 3, 3, 2, 2) \
    X(uint16_t, 4, 3, 2, 3)

struct RatPHost {
    bool ok = false;
    float phase_w[8 * kMaxTaps];   // [phase][tap], phases < 8
    float bias = 0, vbias_rne = 0, near2 = 0;
};

// Is the frame exactly periodic on both axes, and what does using ONE weight set per phase cost in eps?
inline void ratp_prepare(const lanczos_desc& d, const AxisTaps& H, const AxisTaps& V, const RatHost& r, RatPHost* p) {
    p->ok = false;
    if (!r.ok || d.scale_d == 1 || d.scale_n > 7) return;
    const int N = d.scale_n, D = d.scale_d, a = d.a, taps = 2 * a;
    const double maxv = d.bytes_per_sample == 1 ? 255.0 : 65535.0;
    const AxisTaps* ref_ax = nullptr;
    for (const AxisTaps* ax : {&H, &V})
        if (ax->out_n > N * (a + 2) && ax->in_n > 2 * a + D + 2) ref_ax = ax;
    if (!ref_ax) return;
    for (int i = 0; i < 8 * kMaxTaps; i++) p->phase_w[i] = 0.0f;
    const int o_ref = N * (a + 1);   // an interior period: every tap in range
    for (int ph = 0; ph < N; ph++)   // phases above N/2 are the mirror images of those below (what the kernel reads)
        for (int k = 0; k < taps; k++)
            p->phase_w[ph * kMaxTaps + k] = ph <= N / 2 ? (float)ref_ax->w[(size_t)(o_ref + ph) * taps + k]
                                                        : (float)ref_ax->w[(size_t)(o_ref + N - ph) * taps + (taps - 1 - k)];
    double dev = 0;
    auto check = [&](const AxisTaps& ax, const std::vector<uint8_t>& fl) {
        for (int o = 0; o < ax.out_n; o++) {
            if (ax.first[o] != D * (o / N) + ((o % N) * D) / N - a + 1) return false;
            if ((fl[o] != 0) != (o % N == 0)) return false;
            if (o % N == 0) continue;
            for (int k = 0; k < taps; k++) {
                const int i = ax.first[o] + k;
                if (i < 0 || i > ax.in_n - 1) continue;   // dropped tap: the kernel meets a zero sample there
                const double dd = std::fabs(ax.w[(size_t)o * taps + k] - (double)p->phase_w[(o % N) * kMaxTaps + k]);
                // (the f32 rounding of the phase weight is already inside r.bias for the reference index; take the
                //  deviation against the f32 phase weight in full -- conservative)
                if (dd > dev) dev = dd;
            }
        }
        return true;
    };
    if (!check(H, r.h_int) || !check(V, r.v_int)) return;
    const double eps = (double)r.bias + 1.02 * dev * maxv * taps;
    if (eps > 0.2) return;
    p->bias = (float)eps;
    p->vbias_rne = (float)eps - 0.5f;
    p->near2 = (float)(2.0 * eps) * 1.0001f;
    p->ok = true;
}

inline bool ratp_has(const lanczos_desc& d) {
#define X(T, C, N, D, A) \
    if (d.bytes_per_sample == (int)sizeof(T) && d.channels == C && d.scale_n == N && d.scale_d == D && d.a == A) return true;
    LZ_RATP_CONFIGS(X)
#undef X
    return false;
}

inline hipError_t ratp_launch(const lanczos_desc& d, const FrameGeom& g, const TapTables& t, const RatTables& rt,
                              const float* phase_w_dev, hipStream_t stream) {
#define X(T, C, N, D, A)                                                                                               \
    if (d.bytes_per_sample == (int)sizeof(T) && d.channels == C && d.scale_n == N && d.scale_d == D && d.a == A) {     \
        using K = RatPCfg<T, C, N, D, A>;                                                                              \
        const int tiles_x = (g.out_w + K::NUW * K::P_OUT - 1) / (K::NUW * K::P_OUT);                                   \
        const int ty0 = g.out_row0 / K::TH, ty1 = (g.out_row0 + g.out_rows - 1) / K::TH;                               \
        dim3 grid(tiles_x * (ty1 - ty0 + 1), g.frames);                                                                \
        static bool attr_done[2] = {false, false};                                                                     \
        const bool ex = d.mode == LANCZOS_MODE_EXACT;                                                                  \
        {                                                                                                              \
            std::lock_guard<std::mutex> lock(launch_cache_mutex());                                                    \
            if (!attr_done[ex]) {                                                                                      \
                hipError_t e = ex ? hipFuncSetAttribute((const void*)k_ratp<T, C, N, D, A, true>,                      \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES)      \
                                  : hipFuncSetAttribute((const void*)k_ratp<T, C, N, D, A, false>,                     \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);     \
                if (e != hipSuccess) return e;                                                                         \
                attr_done[ex] = true;                                                                                  \
            }                                                                                                          \
        }                                                                                                              \
        if (ex) hipLaunchKernelGGL((k_ratp<T, C, N, D, A, true>), grid, dim3(K::NT), K::LDS_BYTES, stream, g, t, rt, phase_w_dev);  \
        else hipLaunchKernelGGL((k_ratp<T, C, N, D, A, false>), grid, dim3(K::NT), K::LDS_BYTES, stream, g, t, rt, phase_w_dev);    \
        return hipGetLastError();                                                                                      \
    }
    LZ_RATP_CONFIGS(X)
#undef X
    return hipErrorNotSupported;
}

}  // namespace lz
