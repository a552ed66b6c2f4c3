// lanczos_fast.hpp -- specialised fused H+V kernels for integer scales (the benchmark configurations).
//
// One workgroup produces a tile of TH = MR*S output rows x TWP_OUT output pixels:
//   1. LOAD   the MR+2a-1 input rows the tile needs -> LDS (16-byte coalesced row segments, zero
//             outside the image: a dropped tap of full_TB.h:59,72 is a zero contribution)
//   2. HPASS  every thread owns "units" of P input pixels -> P*S output pixels of one row; the window
//             of P+2a-1 pixels is read as aligned dwords, every byte converted once (v_cvt_f32_ubyteN),
//             2a-tap fmaf chains with per-phase f32 weights held in SGPRs; results go to LDS as TRUNCATED
//             integers (the reference's between-pass store, full_TB.h:63); integer-phase samples are
//             byte copies (v_perm_b32), the others are inserted with v_cvt_pk_u8_f32.
//             The horizontal pass must be bit-exact (an error there can double up in the vertical pass),
//             so every sample the f32 chain cannot decide is queued on an LDS worklist:
//               - a sum within +-eps of an integer (eps = proven f32 error bound),
//               - an integer-phase sample 1 <= v0 <= vlim, where the reference's double sum
//                 v0 + O(1e-17) terms can land one ulp below v0 and truncate to v0-1
//   3. FIXUP  the worklist is processed densely (one entry per lane): the exact f64 chain, separate
//             multiply and add, ascending taps (full_TB.h:58-63)
//   4. VPASS  every thread owns one dword column of the output and walks down its share of the tile with
//             a 2a-row register window; integer-phase rows are dword copies, the others 2a fmaf + one
//             v_cvt_pk_u8_f32 per sample.
//             LSB1 mode: f32 result stored (within 1 LSB of the reference by the error bound).
//             EXACT mode: rows with an undecidable sample are recomputed in f64 (wave-uniform branch).
// Output rows < skip_rows (the in-place prefix, full_TB.h:67-77) are left to k_prefix.
#pragma once
#include <cmath>
#include <cstring>
#include <mutex>

#include "lanczos_kernels_common.hpp"
#include "lanczos_taps.hpp"

namespace lz {

constexpr int kFastMaxS = 4;

struct FastConsts {
    float wf[kFastMaxS][kMaxTaps];  // [phase][tap] f32 weights (phase 0 is the integer phase: unused)
    double wi[kMaxTaps];            // integer-phase double weights L(a-1-k): {..,-1.6e-17,3.2e-17,1,3.2e-17,..}
    double wd[kFastMaxS][kMaxTaps]; // [phase][tap] double weights of an interior output index
    int phase_exact_h;              // 1: EVERY horizontal output index carries exactly wd[index % S] on its in-range taps
                                    //    (x = xx/S is exact in double, e.g. S = 2), so the exact chain needs no table
    float bias;                     // eps: f32-chain error bound, added to every sum
    float vbias_rne;                // eps - 0.5: bias under which the RNE byte convert is floor(sum + eps)
    float near2;                    // 2*eps: fract(sum+eps) below this = undecided
    // the same three for the PAIRED chain of the marching kernel (S = 2: symmetric half-phase weights, 3 exact pair sums +
    // 3 fmafs): eps_p < eps.  Equal to the plain ones when the weights are not symmetric (never used then).
    float bias_p, vbias_rne_p, near2_p;
    int vlim;                       // integer-phase flip limit (0: the double chain never leaves v0)
    int skip_last;                  // 1: the last integer-phase tap (x - i = -a, ~1e-33) can never change the sum
    int tight;                      // 1: the only non-negligible negative integer-phase taps sit at +-2 pixels and
                                    //    are < 2^-55, so neighbours <= 2*v0 prove that v0 stays (see fast_prepare)
    int tight2;                     // 1: additionally the +-1 taps are positive and twice the +-2 taps and half the spacing
                                    //    below v0 is >= 3.4 v0 |L(2)|: the second-stage filter of k_march applies
    // 16-bit samples at S = 2, marching kernel: the split-weight chain (lanczos_taps.hpp: split_chain_prepare).  [phase][tap];
    // only [1][0..a-1] are used (paired chain).
    float wsh[kFastMaxS][kMaxTaps], wsl[kFastMaxS][kMaxTaps];
    float bias_s, near2_s;          // eps of that chain (start value of its lo half) and 2 * eps
    int split_ok;                   // 1: the fields above are valid (always for the instantiated 16-bit 2x configurations)
};

// per-configuration tile shape: MR input rows advanced per tile, NGRP vertical thread groups in the V pass
template <typename T, int C, int S, int A>
struct FastShape {
    static constexpr int MR = S == 2 ? 30 : (S == 3 ? 20 : 15);
    static constexpr int NGRP = S == 2 ? 2 : 1;
};

// P_ / UPR_ = 0: the default unit geometry (the tile kernel's); the marching kernel's role-split configurations pass
// wider units (P_ = 8 with UPR_ = 16: the same 128-pixel strip, half the units, 30 % fewer byte->float conversions)
template <typename T, int C_, int S_, int A_, int P_ = 0, int UPR_ = 0>
struct FastCfg {
    static constexpr int C = C_, S = S_, A = A_;
    static constexpr int SB = (int)sizeof(T);
    static constexpr int TAPS = 2 * A;
    static constexpr int P = P_ > 0 ? P_ : (SB == 1 ? (C == 1 ? 8 : 4) : 2);  // input pixels per H unit
    static constexpr int UPR = UPR_ > 0 ? UPR_ : 32;                          // H units per tile row
    static constexpr int TWP_IN = P * UPR;
    static constexpr int TWP_OUT = TWP_IN * S;
    static constexpr int TWS_OUT = TWP_OUT * C;               // output samples per tile row
    static constexpr int TWB_OUT = TWS_OUT * SB;              // ... bytes
    static constexpr int VEC = 4 / SB;                        // samples per V dword
    static constexpr int NVT = TWB_OUT / 4;                   // V threads per group (dword columns)
    static constexpr int NGRP = FastShape<T, C, S, A>::NGRP;
    static constexpr int NT = ((NVT * NGRP + 63) / 64) * 64;  // workgroup size
    static constexpr int MR = FastShape<T, C, S, A>::MR;      // input rows advanced per tile
    static constexpr int MRG = MR / NGRP;                     // ... per V group
    static constexpr int TH = MR * S;                         // output rows per tile
    static constexpr int NR = MR + TAPS - 1;                  // H-pass rows per tile
    static constexpr int LPB = ((A - 1) * C * SB + 15) / 16 * 16;   // left pad bytes (16-B aligned)
    static constexpr int IN_PITCH = (LPB + TWP_IN * C * SB + A * C * SB + 15) / 16 * 16;
    static constexpr int H_PITCH = TWB_OUT;
    static constexpr int WIN_PX = P + TAPS - 1;               // pixels one unit reads
    static constexpr int WIN_S = WIN_PX * C;
    static constexpr int MIS = (LPB - (A - 1) * C * SB) % 4;  // byte offset of the window in its first dword
    static constexpr int NW = (MIS + WIN_S * SB + 3) / 4;     // dwords per unit window
    static constexpr int UNIT_IN_DW = P * C * SB / 4;
    static constexpr int UNIT_OUT_S = P * S * C;
    static constexpr int UNIT_OUT_DW = UNIT_OUT_S * SB / 4;
    static constexpr int WIN_DW0 = (LPB - (A - 1) * C * SB - MIS) / 4;
    static constexpr int WL_CAP = 4096;
    static constexpr int CPR = IN_PITCH / 16;                 // 16-byte chunks per LDS input row
    static constexpr int NCH = NR * CPR;
    static constexpr int LOAD_IT = (NCH + NT - 1) / NT;
    static constexpr int LDS_TIN = NR * IN_PITCH;
    static constexpr int LDS_HBUF = NR * H_PITCH;
    static constexpr int LDS_WL = WL_CAP * 2;
    static constexpr int LDS_BYTES = LDS_TIN + LDS_HBUF + LDS_WL + 16;
    static constexpr float MAXV = SB == 1 ? 255.0f : 65535.0f;
    static constexpr unsigned SMASK = SB == 1 ? 0xffu : 0xffffu;
    static constexpr int WL_ROW_BITS = NR <= 32 ? 5 : 6;      // worklist entry = row | sample, 16 bits
    static constexpr int WL_SMP_BITS = 16 - WL_ROW_BITS;
    static_assert((P * C * SB) % 4 == 0, "unit must cover whole dwords");
    static_assert(UNIT_OUT_S <= 64, "flag mask is 64 bits");
    static_assert(MR % NGRP == 0, "V groups split the tile evenly");
    static_assert(NR <= 64 && TWS_OUT <= (1 << WL_SMP_BITS), "worklist entry packs row | sample in 16 bits");
    static_assert(H_PITCH % 4 == 0 && IN_PITCH % 16 == 0, "LDS pitches");
};

// sample k (static) of a dword window that starts MIS bytes into wd[0]
template <typename T, int MIS, int NW>
__device__ __forceinline__ unsigned win_sample(const uint32_t (&wd)[NW], int k) {
    const int b = MIS + k * (int)sizeof(T);
    if (sizeof(T) == 1) return (wd[b >> 2] >> (8 * (b & 3))) & 0xffu;
    return (wd[b >> 2] >> (8 * (b & 3))) & 0xffffu;
}

// SWAR: per byte/halfword lane of x, top bit set iff 1 <= lane <= vlim  (vlim < half range)
template <int SB>
__device__ __forceinline__ uint32_t swar_in_1_vlim(uint32_t x, uint32_t addc /* (HALF - vlim - 1) replicated */) {
    constexpr uint32_t LOW = SB == 1 ? 0x7f7f7f7fu : 0x7fff7fffu;
    constexpr uint32_t TOP = SB == 1 ? 0x80808080u : 0x80008000u;
    const uint32_t t = x & LOW;
    const uint32_t nz = (t + LOW) | x;   // top bit: lane != 0
    const uint32_t hi = (t + addc) | x;  // top bit: lane > vlim
    return nz & ~hi & TOP;
}

template <typename T, int C, int S, int A, bool EXACT>
__global__ __launch_bounds__((FastCfg<T, C, S, A>::NT)) void k_fast(FrameGeom g, TapTables t, FastConsts fc) {
    using K = FastCfg<T, C, S, A>;
    constexpr int TAPS = K::TAPS;
    constexpr int SB = K::SB;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* tin = smem;
    uint8_t* hbuf = smem + K::LDS_TIN;
    uint16_t* wl = (uint16_t*)(smem + K::LDS_TIN + K::LDS_HBUF);
    unsigned* wl_count = (unsigned*)(smem + K::LDS_TIN + K::LDS_HBUF + K::LDS_WL);

    const int tid = threadIdx.x;
    const int tiles_x = (g.out_w + K::TWP_OUT - 1) / K::TWP_OUT;
    const int tx = blockIdx.x % tiles_x;
    const int ty = blockIdx.x / tiles_x;
    const int frame = blockIdx.y;

    // tile geometry.  Tiles are aligned to multiples of TH in FULL-frame rows so m0 is exact.
    const int y_tile = (g.out_row0 / K::TH + ty) * K::TH;
    const int y_begin = y_tile > g.out_row0 ? y_tile : g.out_row0;
    int y_end = y_tile + K::TH;
    if (y_end > g.out_row0 + g.out_rows) y_end = g.out_row0 + g.out_rows;
    const int y_first = y_begin > g.skip_rows ? y_begin : g.skip_rows;  // first row this tile stores
    if (y_end <= y_first) return;
    const int m0 = y_tile / S;            // first input row index whose S output rows live here
    const int r_lo = m0 - A + 1;          // LDS row 0 <-> input row r_lo (may be negative)
    const int P0 = tx * K::TWP_IN;        // first input pixel owned by the tile
    const int row_bytes = g.in_w * C * SB;

    const uint8_t* in_f = g.in + (size_t)frame * g.in_frame_stride;
    uint8_t* out_f = g.out + (size_t)frame * g.out_frame_stride;

    // ------------------------------------------------------------------ 1. LOAD
    {
        const int tile_gb0 = P0 * C * SB - K::LPB;  // byte offset in the input row of LDS column 0
        const int gr_min = g.in_row0 > 0 ? g.in_row0 : 0;
        const int gr_max = (g.in_row0 + g.in_rows < g.in_h ? g.in_row0 + g.in_rows : g.in_h) - 1;
        if (tid == 0) *wl_count = 0;
        uint4 v[K::LOAD_IT];
#pragma unroll
        for (int it = 0; it < K::LOAD_IT; it++) {
            const int idx = tid + it * K::NT;
            const int row = idx / K::CPR, ch = idx - row * K::CPR;
            const int gr = r_lo + row;
            const int gb = tile_gb0 + 16 * ch;
            v[it] = make_uint4(0, 0, 0, 0);
            if (idx < K::NCH && gr >= gr_min && gr <= gr_max) {
                const uint8_t* rp = in_f + (size_t)(gr - g.in_row0) * g.in_pitch;
                if (gb >= 0 && gb + 16 <= row_bytes && (((uintptr_t)(rp + gb)) & 15) == 0) {
                    v[it] = *(const uint4*)(rp + gb);
                } else if (gb + 16 > 0 && gb < row_bytes) {  // ragged edge / unaligned pitch: bytewise
                    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int b = 0; b < 16; b++) {
                        const int q = gb + b;
                        if (q >= 0 && q < row_bytes) w[b >> 2] |= (uint32_t)rp[q] << (8 * (b & 3));
                    }
                    v[it] = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < K::LOAD_IT; it++) {
            const int idx = tid + it * K::NT;
            if (idx < K::NCH) *(uint4*)(tin + idx * 16) = v[it];  // row*IN_PITCH + ch*16 == idx*16
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ 2. HPASS
    {
        constexpr int NU = K::NR * K::UPR;
        const uint32_t* tin32 = (const uint32_t*)tin;
        uint32_t* hbuf32 = (uint32_t*)hbuf;
        constexpr uint32_t HALF = SB == 1 ? 0x80u : 0x8000u;
        const uint32_t addc1 = (uint32_t)fc.vlim < HALF - 1 ? HALF - 1 - (uint32_t)fc.vlim : 0;
        const uint32_t addc = SB == 1 ? addc1 * 0x01010101u : addc1 * 0x00010001u;
        for (int idx = tid; idx < NU; idx += K::NT) {
            const int row = idx / K::UPR, u = idx % K::UPR;
            uint32_t wd[K::NW];
            const uint32_t* wp = tin32 + row * (K::IN_PITCH / 4) + K::WIN_DW0 + u * K::UNIT_IN_DW;
#pragma unroll
            for (int i = 0; i < K::NW; i++) wd[i] = wp[i];
            float f[K::WIN_S];
#pragma unroll
            for (int k = 0; k < K::WIN_S; k++) f[k] = (float)win_sample<T, K::MIS, K::NW>(wd, k);

            // f32 chains for the non-integer phases; xb = sum + eps
            float xb[K::UNIT_OUT_S];
            float dmin = 1.0f;
#pragma unroll
            for (int q = 0; q < K::P * S; q++) {
                const int p = q / S, ph = q % S;
                if (ph == 0) continue;
#pragma unroll
                for (int c = 0; c < C; c++) {
                    float acc = fc.bias;
#pragma unroll
                    for (int j = 0; j < TAPS; j++) {
                        const int k = f32_tap_order(j, TAPS);  // outside in: the bound of fc.bias assumes this order
                        acc = __builtin_fmaf(fc.wf[ph][k], f[(p + k) * C + c], acc);
                    }
                    // below 1 / above max the store clamps: nothing to decide there
                    const float xc = __builtin_amdgcn_fmed3f(acc, 0.5f, K::MAXV + 0.5f);
                    const float fl = __builtin_floorf(xc);
                    dmin = __builtin_fminf(dmin, xc - fl);
                    xb[q * C + c] = fl;
                }
            }
            // assemble the output dwords: integer-phase samples are copies of input samples
            uint32_t* hp = hbuf32 + row * (K::H_PITCH / 4) + u * K::UNIT_OUT_DW;
#pragma unroll
            for (int i = 0; i < K::UNIT_OUT_DW; i++) {
                uint32_t w = 0;
                if (SB == 1) {
                    // raw bytes first (one v_perm_b32 when they come from <= 2 window dwords) ...
                    int src[4], d0 = -1, d1 = -1;
                    bool any_raw = false, perm_ok = true;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int o = i * 4 + e, q = o / C, c = o % C;
                        src[e] = -1;
                        if (q % S == 0) {
                            const int b = K::MIS + ((q / S + A - 1) * C + c);
                            src[e] = b;
                            any_raw = true;
                            const int dw = b >> 2;
                            if (d0 < 0 || d0 == dw) d0 = dw;
                            else if (d1 < 0 || d1 == dw) d1 = dw;
                            else perm_ok = false;
                        }
                    }
                    if (any_raw) {
                        if (perm_ok) {
                            uint32_t sel = 0;
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                uint32_t s = 0x0c;  // constant 0
                                if (src[e] >= 0) s = ((src[e] >> 2) == d0 ? 0 : 4) + (src[e] & 3);
                                sel |= s << (8 * e);
                            }
                            w = __builtin_amdgcn_perm(wd[d1 < 0 ? d0 : d1], wd[d0], sel);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; e++)
                                if (src[e] >= 0) w |= ((wd[src[e] >> 2] >> (8 * (src[e] & 3))) & 0xffu) << (8 * e);
                        }
                    }
                    // ... then the computed ones (already integers: the RNE convert is exact, saturates)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int o = i * 4 + e;
                        if ((o / C) % S != 0) w = __builtin_amdgcn_cvt_pk_u8_f32(xb[o], e, w);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const int o = i * 2 + e, q = o / C, c = o % C;
                        const unsigned sv = (q % S == 0) ? win_sample<T, K::MIS, K::NW>(wd, (q / S + A - 1) * C + c)
                                                         : (unsigned)xb[o];
                        w |= sv << (16 * e);
                    }
                }
                hp[i] = w;
            }
            // undecided samples -> worklist
            unsigned long long mask = 0;
            if (fc.vlim > 0) {
                // the P own input pixels are dword aligned inside the window: SWAR test 1 <= v0 <= vlim
                constexpr int OWN_DW0 = (K::MIS + (A - 1) * C * SB) / 4;
                static_assert((K::MIS + (A - 1) * C * SB) % 4 == 0, "own pixels start on a dword");
#pragma unroll
                for (int i = 0; i < K::UNIT_IN_DW; i++) {
                    uint32_t fl = swar_in_1_vlim<SB>(wd[OWN_DW0 + i], addc);
                    if (fl) {
#pragma unroll
                        for (int e = 0; e < K::VEC; e++) {
                            // input sample (i*VEC+e) of the own pixels -> output sample of its integer phase
                            const int si = i * K::VEC + e, p = si / C, c = si % C;
                            const int o = (p * S) * C + c;
                            if (fl & (1u << (8 * SB * e + 8 * SB - 1))) mask |= 1ull << o;
                        }
                    }
                }
            }
            if (dmin < fc.near2) {  // rare: queue every non-integer-phase sample of the unit
#pragma unroll
                for (int o = 0; o < K::UNIT_OUT_S; o++)
                    if ((o / C) % S != 0) mask |= 1ull << o;
            }
            if (mask) {
                const int n = __popcll(mask);
                unsigned base = atomicAdd(wl_count, (unsigned)n);
                const unsigned ent0 = ((unsigned)row << K::WL_SMP_BITS) | (unsigned)(u * K::UNIT_OUT_S);
                while (mask) {
                    const int b = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    if (base < (unsigned)K::WL_CAP) wl[base] = (uint16_t)(ent0 + b);
                    base++;
                }
            }
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ 3. FIXUP (exact H samples)
    {
        const unsigned total = *wl_count;
        const bool overflow = total > (unsigned)K::WL_CAP;
        // If the list overflowed (pathological input) every sample of the tile is redone exactly.
        const unsigned n = overflow ? (unsigned)(K::NR * K::TWS_OUT) : total;
        const T* tinT = (const T*)tin;
        T* hbufT = (T*)hbuf;
        const int j0 = tx * K::TWS_OUT;  // first output sample column of the tile
        for (unsigned i = tid; i < n; i += K::NT) {
            int row, js;
            if (overflow) {
                row = i / K::TWS_OUT;
                js = i % K::TWS_OUT;
            } else {
                const unsigned e = wl[i];
                row = e >> K::WL_SMP_BITS;
                js = e & ((1u << K::WL_SMP_BITS) - 1);
            }
            const int xl = js / C, c = js - xl * C;   // output pixel inside the tile
            const int xx = tx * K::TWP_OUT + xl;
            if (xx >= g.out_w) continue;
            const int fl = xl / S;                      // floor(x) - P0
            const T* rp = tinT + (row * K::IN_PITCH + K::LPB) / SB + (fl - A + 1) * C + c;
            double sum = 0;
            if (xl - fl * S == 0) {
                // integer phase: x - i is an exact integer, the weights are the same everywhere
#pragma unroll
                for (int k = 0; k < TAPS; k++) sum += (double)rp[k * C] * fc.wi[k];
            } else {
                const double* w = t.h_w + (size_t)xx * TAPS;
#pragma unroll
                for (int k = 0; k < TAPS; k++) sum += (double)rp[k * C] * w[k];
            }
            hbufT[row * (K::H_PITCH / SB) + js] = store_convert<T>(sum);
        }
        (void)j0;
    }
    __syncthreads();

    // ------------------------------------------------------------------ 4. VPASS
    {
        // vertical group of this thread: wave-uniform (NVT is a whole number of waves when NGRP > 1), so
        // row indices, row-range tests and row base addresses below stay on the scalar unit
        static_assert(K::NGRP == 1 || K::NVT % 64 == 0, "V groups must be whole waves");
        const int grp = K::NGRP == 1 ? (tid < K::NVT ? 0 : 1)
                                     : __builtin_amdgcn_readfirstlane(tid >> 6) * 64 / K::NVT;
        const int col = tid - grp * K::NVT;     // dword column
        const unsigned col_b = (unsigned)(tx * K::TWB_OUT + col * 4);
        const bool col_ok = grp < K::NGRP && col_b + 4 <= (unsigned)(g.out_w * C * SB);
        // buffer descriptor over this frame's strip: 32-bit per-lane offsets, hardware range check
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
            out_f, 0, (unsigned)(g.out_rows * g.out_pitch), 0x00020000);
        if (col_ok) {
            constexpr int HP = K::H_PITCH / 4;
            const int ml0 = grp * K::MRG;       // first local input-row step of this group
            const uint32_t* hb = (const uint32_t*)hbuf + col + ml0 * HP;
            float win[TAPS][K::VEC];
            uint32_t raw[TAPS];
            auto unpack = [&](int slot, uint32_t w) {
                raw[slot] = w;
#pragma unroll
                for (int e = 0; e < K::VEC; e++) win[slot][e] = (float)((w >> (8 * SB * e)) & K::SMASK);
            };
#pragma unroll
            for (int k = 0; k < TAPS - 1; k++) unpack(k, hb[k * HP]);
            // LSB1: the store is floor(sum + eps); with the sum biased by eps - 0.5 the hardware's
            // round-to-nearest-even convert IS that floor (no tie can occur: fract(sum+eps) is never 0
            // for a decided sample, and undecided ones are within 1 LSB either way)
            float vbias = (SB == 1 && !EXACT) ? fc.vbias_rne : fc.bias;
            // pin the phase weights in SGPRs for the whole walk (otherwise hipcc re-issues the kernarg
            // s_load + s_waitcnt inside every row block)
            float wv[S][TAPS];
#pragma unroll
            for (int ph = 1; ph < S; ph++)
#pragma unroll
                for (int k = 0; k < TAPS; k++) {
                    wv[ph][k] = fc.wf[ph][k];
                    asm volatile("" : "+s"(wv[ph][k]));
                }
            asm volatile("" : "+s"(vbias));

            for (int mm = 0; mm < K::MRG; mm += TAPS) {
                const uint32_t* hb_mm = hb + mm * HP;
                // uniform: first output row of this outer step, as a byte offset into the strip
                const int y_mm = y_tile + (ml0 + mm) * S;
#pragma unroll
                for (int i = 0; i < TAPS; i++) {
                    if (mm + i >= K::MRG) break;
                    unpack((i + TAPS - 1) % TAPS, hb_mm[(i + TAPS - 1) * HP]);
#pragma unroll
                    for (int ph = 0; ph < S; ph++) {
                        const int y = y_mm + i * S + ph;  // floor(y/S) = m0 + ml0 + mm + i
                        uint32_t packed;
                        bool undecided = false;
                        if (ph == 0) {
                            packed = raw[(i + A - 1) % TAPS];
                            if (EXACT && fc.vlim > 0) {
#pragma unroll
                                for (int e = 0; e < K::VEC; e++)
                                    undecided |= (((packed >> (8 * SB * e)) & K::SMASK) - 1u < (unsigned)fc.vlim);
                            }
                        } else {
                            packed = 0;
                            float accs[K::VEC];
#pragma unroll
                            for (int e = 0; e < K::VEC; e++) {
                                float acc = vbias;
#pragma unroll
                                for (int j = 0; j < TAPS; j++) {
                                    const int k = f32_tap_order(j, TAPS);
                                    acc = __builtin_fmaf(wv[ph][k], win[(i + k) % TAPS][e], acc);
                                }
                                accs[e] = acc;
                            }
                            if (SB == 1 && !EXACT) {
#pragma unroll
                                for (int e = 0; e < 4; e++) packed = __builtin_amdgcn_cvt_pk_u8_f32(accs[e], e, packed);
                            } else {
#pragma unroll
                                for (int e = 0; e < K::VEC; e++) {
                                    const float xc = __builtin_amdgcn_fmed3f(accs[e], 0.5f, K::MAXV + 0.5f);
                                    const float fl = __builtin_floorf(xc);
                                    if (EXACT) undecided |= (xc - fl) < fc.near2;
                                    packed |= (unsigned)fl << (8 * SB * e);
                                }
                            }
                        }
                        if (EXACT) {
                            if (__any(undecided)) {  // wave-uniform: redo this row's dword in f64 (full_TB.h:71-75)
                                const double* wv = t.v_w + (size_t)(y < g.out_h ? y : g.out_h - 1) * TAPS;
                                packed = 0;
#pragma unroll
                                for (int e = 0; e < K::VEC; e++) {
                                    double sum = 0;
#pragma unroll
                                    for (int k = 0; k < TAPS; k++) sum += (double)win[(i + k) % TAPS][e] * wv[k];
                                    packed |= (unsigned)store_convert<T>(sum) << (8 * SB * e);
                                }
                            }
                        }
                        if (y >= y_first && y < y_end)  // uniform; row offset rides in the scalar soffset
                            __builtin_amdgcn_raw_buffer_store_b32(packed, orsrc, col_b,
                                                                  (y - g.out_row0) * g.out_pitch, LZ_STORE_AUX);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------- host side
// The launch helpers cache per-device facts (occupancy answers, "dynamic LDS attribute set") in function-local statics.
// Contexts on different host threads share them, so every access happens under this one process-wide mutex
// (uncontended: tens of nanoseconds per launch).
inline std::mutex& launch_cache_mutex() {
    static std::mutex m;
    return m;
}

inline bool fast_prepare(const lanczos_desc& d, const AxisTaps& H, const AxisTaps& V, FastConsts* fc) {
    const int S = d.scale_n, a = d.a, taps = 2 * a;
    if (d.scale_d != 1 || S < 2 || S > kFastMaxS) return false;
    // phase weights from an interior output index (all 2a taps in range), either axis will do
    const AxisTaps* ax = nullptr;
    for (const AxisTaps* cand : {&H, &V})
        if (cand->in_n >= 2 * a + 2 && cand->out_n > S * a + S) ax = cand;
    if (!ax) return false;
    const double maxv = d.bytes_per_sample == 1 ? 255.0 : 65535.0;
    double eps = 0;
    for (int ph = 0; ph < kFastMaxS; ph++)
        for (int k = 0; k < kMaxTaps; k++) fc->wf[ph][k] = 0.0f;
    for (int k = 0; k < kMaxTaps; k++) fc->wi[k] = 0.0;
    for (int ph = 0; ph < S; ph++) {
        const int o = S * a + ph;
        const double* w = &ax->w[(size_t)o * taps];
        for (int k = 0; k < taps; k++) fc->wf[ph][k] = (float)w[k];
        if (ph == 0)
            for (int k = 0; k < taps; k++) fc->wi[k] = w[k];
        for (int k = 0; k < kMaxTaps; k++) fc->wd[ph][k] = k < taps ? w[k] : 0.0;
        if (ph != 0) {
            const double e = f32_chain_error_bound(w, taps, maxv);
            if (e > eps) eps = e;
        }
    }
    if (S == 3) {
        // phase 2/3 mirrors phase 1/3 (L is even; the doubles differ by the rounding of x = o/3 only): the kernels keep ONE
        // set of f32 weights for both.  Normally (float) of the two doubles is the same number; where it is not, the
        // difference is part of the chain's error bound.
        double dev = 0;
        for (int k = 0; k < taps; k++) {
            dev += std::fabs((double)fc->wf[1][taps - 1 - k] - (double)fc->wf[2][k]) * maxv;
            fc->wf[2][k] = fc->wf[1][taps - 1 - k];
        }
        eps += 1.02 * dev;
    }
    // per-index weights differ from the phase weights by the rounding of x = o/S (a few ulp of x):
    // far below the f32 slack, but count it: |dw| <= |L'| * ulp(x) <= 4 * 2^-52 * out_n
    eps += 4.0 * 2.220446049250313e-16 * (H.out_n > V.out_n ? H.out_n : V.out_n) * maxv * taps;
    fc->bias = (float)eps;
    fc->vbias_rne = (float)eps - 0.5f;
    fc->near2 = (float)(2.0 * eps) * 1.0001f;
    {   // paired chain (S = 2): acc = fma(w[k], v[k] + v[2a-1-k], acc), k = 0..a-1 (outside in); pair sums are exact
        double eps_p = eps;
        if (S == 2) {
            const double* w = &ax->w[(size_t)(S * a + 1) * taps];
            bool sym = true;
            for (int k = 0; k < a; k++) sym = sym && (float)w[k] == (float)w[taps - 1 - k];
            if (!sym) return false;  // cannot happen: L is even and x = m + 1/2 is exact
            int order[kMaxTaps];
            for (int k = 0; k < a; k++) order[k] = k;
            // the exact-sum reference uses w[k] for v[k] and w[2a-1-k] for its partner: count that quantisation too
            double wq_extra = 0;
            for (int k = 0; k < a; k++) wq_extra += std::fabs((double)(float)w[k] - w[taps - 1 - k]) * maxv;
            eps_p = f32_chain_error_bound_ordered(w, order, a, 2.0 * maxv) + 1.02 * wq_extra +
                    4.0 * 2.220446049250313e-16 * (H.out_n > V.out_n ? H.out_n : V.out_n) * maxv * taps;
        }
        fc->bias_p = (float)eps_p;
        fc->vbias_rne_p = (float)eps_p - 0.5f;
        fc->near2_p = (float)(2.0 * eps_p) * 1.0001f;
    }
    {   // split-weight chain of the marching kernel's 16-bit 2x instances (paired chain: taps 0..a-1)
        fc->split_ok = 0;
        fc->bias_s = fc->near2_s = 0.0f;
        for (int ph = 0; ph < kFastMaxS; ph++)
            for (int k = 0; k < kMaxTaps; k++) fc->wsh[ph][k] = fc->wsl[ph][k] = 0.0f;
        if (d.bytes_per_sample == 2 && S == 2) {
            const double per_index = 4.0 * 2.220446049250313e-16 * (H.out_n > V.out_n ? H.out_n : V.out_n) * maxv * taps;
            const double* w1 = &ax->w[(size_t)(S * a + 1) * taps];
            int order[kMaxTaps];
            for (int k = 0; k < a; k++) order[k] = k;
            SplitChain sc;
            const bool ok = split_chain_prepare(w1, order, a, 2.0 * maxv, &sc);
            double partner = 0;  // the exact chain uses w[2a-1-k] for the partner sample: the same number up to its last bits
            for (int k = 0; k < a; k++) partner += std::fabs(w1[k] - w1[taps - 1 - k]) * maxv;
            const double eps_s = sc.eps + 1.02 * partner + per_index;
            if (ok && eps_s < 0.005) {
                for (int k = 0; k < a; k++) fc->wsh[1][k] = sc.wh[k], fc->wsl[1][k] = sc.wl[k];
                fc->bias_s = (float)eps_s;
                fc->near2_s = (float)(2.0 * eps_s) * 1.0001f;
                fc->split_ok = 1;
            }
            if (!fc->split_ok) return false;  // (cannot happen for a = 2..4: the kernels' 16-bit 2x instances rely on it)
        }
    }
    fc->vlim = integer_phase_flip_limit(fc->wi, a, (int)maxv);
    if (fc->vlim >= (d.bytes_per_sample == 1 ? 126 : 32766)) return false;  // SWAR test needs vlim < half range
    // Tight filter precondition (integer_phase_tight, lanczos_taps.cpp)
    fc->tight = integer_phase_tight(fc->wi, a, maxv) ? 1 : 0;
    fc->tight2 = (fc->tight && d.bytes_per_sample == 1 && integer_phase_tight2(fc->wi, a, fc->vlim)) ? 1 : 0;
    // Integer-phase chain, last tap (x - i = -a): by then the running sum is v0 +- a few ulp >= 0.5, so its spacing is
    // >= 2^-54; a term below 2^-55 cannot move it (strictly less than half the spacing: no tie either).  The centre
    // weight is exactly 1.0 (sinc(0)*sinc(0)), so that product is the sample itself.
    fc->skip_last = (std::fabs(fc->wi[taps - 1]) * maxv < std::ldexp(1.0, -55) && fc->wi[a - 1] == 1.0) ? 1 : 0;
    if (eps > 0.2) return false;  // f32 cannot even guarantee +-1 LSB
    {   // are the table rows of the horizontal axis the phase weights, bit for bit?  (out-of-range taps are 0 in the
        // table and meet zero samples in the kernels: a +-0.0 term either way)
        bool same = true;
        for (int xx = 0; xx < H.out_n && same; xx++)
            for (int k = 0; k < taps; k++) {
                const int i = H.first[xx] + k;
                if (i < 0 || i >= H.in_n) continue;
                if (std::memcmp(&H.w[(size_t)xx * taps + k], &fc->wd[xx % S][k], sizeof(double)) != 0) {
                    same = false;
                    break;
                }
            }
        fc->phase_exact_h = same ? 1 : 0;
    }
    return true;
}

template <typename T, int C, int S, int A>
inline hipError_t fast_launch_t(const lanczos_desc& d, const FrameGeom& g, const TapTables& t,
                                const FastConsts& fc, hipStream_t stream) {
    using K = FastCfg<T, C, S, A>;
    const int tiles_x = (g.out_w + K::TWP_OUT - 1) / K::TWP_OUT;
    const int ty0 = g.out_row0 / K::TH;
    const int ty1 = (g.out_row0 + g.out_rows - 1) / K::TH;
    dim3 grid(tiles_x * (ty1 - ty0 + 1), g.frames);
    std::lock_guard<std::mutex> cache_lock(launch_cache_mutex());
    static bool attr_done[2][64] = {};
    const bool exact = d.mode == LANCZOS_MODE_EXACT;
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    if (!attr_done[exact][dev]) {
        hipError_t e = exact ? hipFuncSetAttribute((const void*)k_fast<T, C, S, A, true>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES)
                             : hipFuncSetAttribute((const void*)k_fast<T, C, S, A, false>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done[exact][dev] = true;
    }
    if (exact)
        hipLaunchKernelGGL((k_fast<T, C, S, A, true>), grid, dim3(K::NT), K::LDS_BYTES, stream, g, t, fc);
    else
        hipLaunchKernelGGL((k_fast<T, C, S, A, false>), grid, dim3(K::NT), K::LDS_BYTES, stream, g, t, fc);
    return hipGetLastError();
}

// The instantiated configurations: (sample type, channels, scale, a) -- every integer scale 2..4 of the reference's params.h
// space (lanczos.h:9-31: NUM_CHANNELS 1 / 3 / 4, LANCZOS_A 2..4) for 8-bit samples, scales 2 and 3 with a = 3, 4 for 16-bit ones.
// Four groups = four translation units (csrc/lanczos_inst.hip compiled with -DLZ_INST_GROUP=0..3, in parallel).
#define LZ_FAST_CONFIGS_G0(X) /* 8-bit, 2x */ \
    X(uint8_t, 3, 2, 3) X(uint8_t, 3, 2, 2) X(uint8_t, 3, 2, 4) X(uint8_t, 4, 2, 2) X(uint8_t, 4, 2, 3) X(uint8_t, 4, 2, 4) \
    X(uint8_t, 1, 2, 2) X(uint8_t, 1, 2, 3) X(uint8_t, 1, 2, 4)
#define LZ_FAST_CONFIGS_G1(X) /* 8-bit, 3x */ \
    X(uint8_t, 3, 3, 3) X(uint8_t, 3, 3, 2) X(uint8_t, 3, 3, 4) X(uint8_t, 4, 3, 2) X(uint8_t, 4, 3, 3) X(uint8_t, 4, 3, 4) \
    X(uint8_t, 1, 3, 2) X(uint8_t, 1, 3, 3) X(uint8_t, 1, 3, 4)
#define LZ_FAST_CONFIGS_G2(X) /* 8-bit, 4x */ \
    X(uint8_t, 3, 4, 3) X(uint8_t, 3, 4, 2) X(uint8_t, 3, 4, 4) X(uint8_t, 4, 4, 2) X(uint8_t, 4, 4, 3) X(uint8_t, 4, 4, 4) \
    X(uint8_t, 1, 4, 2) X(uint8_t, 1, 4, 3) X(uint8_t, 1, 4, 4)
#define LZ_FAST_CONFIGS_G3(X) /* 16-bit */ \
    X(uint16_t, 4, 2, 4) X(uint16_t, 3, 2, 3) X(uint16_t, 4, 2, 3) X(uint16_t, 3, 2, 4) \
    X(uint16_t, 3, 3, 3) X(uint16_t, 3, 3, 4) X(uint16_t, 4, 3, 3) X(uint16_t, 4, 3, 4)
#define LZ_FAST_CONFIGS(X) LZ_FAST_CONFIGS_G0(X) LZ_FAST_CONFIGS_G1(X) LZ_FAST_CONFIGS_G2(X) LZ_FAST_CONFIGS_G3(X)

inline bool fast_supports(const lanczos_desc& d, const FrameGeom& g) {
    if (d.scale_d != 1) return false;
    if (g.out_pitch % 4 != 0 || (((uintptr_t)g.out) & 3) != 0 || (g.out_frame_stride & 3) != 0) return false;
    if ((size_t)g.out_pitch * g.out_rows >= (1ull << 31) || (size_t)g.in_pitch >= (1ull << 30)) return false;
#define X(T, C, S, A) \
    if (d.bytes_per_sample == (int)sizeof(T) && d.channels == C && d.scale_n == S && d.a == A) return true;
    LZ_FAST_CONFIGS(X)
#undef X
    return false;
}

// one dispatcher per group, defined in lanczos_inst.hip (hipErrorNotSupported: not one of the group's configurations)
#define LZ_DECLARE_FAST_GROUP(G)                                                                                         \
    hipError_t fast_launch_g##G(const lanczos_desc& d, const FrameGeom& g, const TapTables& t, const FastConsts& fc, hipStream_t stream);
LZ_DECLARE_FAST_GROUP(0)
LZ_DECLARE_FAST_GROUP(1)
LZ_DECLARE_FAST_GROUP(2)
LZ_DECLARE_FAST_GROUP(3)
#undef LZ_DECLARE_FAST_GROUP

inline hipError_t fast_launch(const lanczos_desc& d, const FrameGeom& g, const TapTables& t, const FastConsts& fc,
                              hipStream_t stream) {
    if (d.bytes_per_sample == 2) return fast_launch_g3(d, g, t, fc, stream);
    if (d.scale_n == 2) return fast_launch_g0(d, g, t, fc, stream);
    if (d.scale_n == 3) return fast_launch_g1(d, g, t, fc, stream);
    if (d.scale_n == 4) return fast_launch_g2(d, g, t, fc, stream);
    return hipErrorNotSupported;
}

}  // namespace lz
