// lanczos_march.hpp -- the production kernel: a workgroup MARCHES down a column strip of the frame.
//
// This is the MI355X replacement for the reference's strip scheduler + workers + cyclic line buffer
// (process_channel lanczos.cpp:68-83, ColWorkers/RowWorkers worker.cpp:134-284, CyclicBuffer
// cyclic_buffer.h:4-69): a ring of horizontally-resampled rows lives in LDS and is advanced MS input rows
// per "tick"; nothing but the input and output frames ever touches HBM.
//
// Workgroup = one strip of TWP_OUT output pixels x a share of rows that the host hands it through a table indexed by
// the hardware block id (march_build_table): one to three segments, each a run of rows of one (strip, frame) pair.  Inside a
// segment ONE barrier per tick; between two barriers every wave runs, back to back and independent of the other waves:
//   PREFETCH  one 16-byte buffer load per lane of the input rows two ticks ahead (out-of-image lanes read
//             zeros through the descriptor's range check: a dropped tap is a zero contribution)
//   HPASS     tick t+1: one unit (P input pixels -> P*S output pixels of one row) per thread from the LDS
//             input rows: aligned dword reads, v_cvt_f32_ubyteN, pair sums + fmaf chains with VGPR weights,
//             v_perm_b32 / v_cvt_pk_u8_f32 packing, truncated integers into the LDS ring (the reference's
//             between-pass store, full_TB.h:63)
//   FIXUP     samples the f32 chain cannot decide are put on a wave-private LDS list (integer-phase candidates: ballot +
//             mbcnt compaction; near-integer units: written by the first lanes directly) and redone densely with the exact
//             f64 chain (full_TB.h:58-63), whose phase weights live in LDS:
//               - a sum within +-eps of an integer (eps = proven f32 error bound)
//               - an integer-phase sample whose double sum v0 + O(1e-17) may truncate to v0-1:
//                 1 <= v0 <= vlim and a +-2 neighbour brighter than 2*v0 (SWAR tests on the packed bytes)
//   VPASS     tick t: one dword column per thread, 2a-row register window over the ring; integer-phase rows
//             are dword copies, the others pair sums + fmaf + one v_cvt_pk_u8_f32 per sample; buffer stores with the
//             row offset in the scalar operand
// Config 2: 40.6 KB of LDS per workgroup, 68 VGPRs / 96 SGPRs: 4 workgroups (24 waves) per CU.
//
// What bounds it: the memory system.  The kernel's traffic alone -- the same store and load segments, nothing
// else -- takes 176 us per 32 frames of config 2 (0.707 of 8 TB/s; reads and writes do not overlap in the DRAM); the kernel
// takes 203-212 us, of which its arithmetic adds 15-25 us on top of its own memory skeleton.
//
// Round 4 tried the three structural levers that were left, each an interleaved A/B with bit-identical (or, for the first, +-1 LSB)
// output; the patches and their measurements are under profiles/ (experiments/round4_*.patch, round4[c-r]_*.txt):
//   * the V pass on the matrix cores (ds_read_b64_tr_b8 of the row-major ring -> f16 MFMA with hi + lo weights -> permlane
//     transposes -> 16-byte stores): parity-green, 220-264 us against 215 -- per useful output its v_perm conversions, byte
//     converts and MFMA issue slots cost what the FMA chains cost, and its natural store shape (16 rows x 16 bytes) runs at 1.2 TB/s;
//   * input rows requested a whole tick ahead, made affordable by that V pass (no 30-register window): 228 against 217 us;
//     requested in front of the V stores and committed behind them with a counted s_waitcnt vmcnt(12): 238 against 220 us;
//   * the V window as 16-bit integer lanes (v_fma_mix_f32 on f16 denormals): 65 instead of 72 VGPRs, bit-identical, 219.5
//     against 214.9 us in LSB1 mode -- kept only for the EXACT instances, whose float window spilled inside the row loop.
// Around the tick loop (all measured, profiles/README.md):
//   * the table deals whole frames to every XCD (neighbouring strips share an L2: every input line is fetched from HBM once)
//     and gives the workgroup slots a CU fills first -- whose waves are older and win the SIMD arbitration -- more rows;
//   * the output stores are non-temporal (the output must not evict the input rows neighbours re-read);
//   * the phase weights live in VGPRs (an SGPR source puts a VALU op in gfx950's slow issue class);
//   * per-lane indices are rebuilt every tick from an opaque copy of the thread id (held across the segment loop they cost a
//     fourth workgroup per CU);
//   * RIDE variant (small batches): the in-place prefix rows (k_prefix) are extra workgroups at the end of this grid;
//   * EXACT variant: the V pass keeps the H pass's exactness tests; undecided rows are redone in f64 after the row loop.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <algorithm>
#include <cstring>
#include <utility>
#include <vector>

#include "lanczos_cache.hpp"
#include "lanczos_env.hpp"
#include "lanczos_fast.hpp"

// SGPR budget: gfx950 admits min(8, floor(800 / (ceil(sgpr/16)*16 + 16))) waves per SIMD (MI355X_MICROARCH.md,
// "Residency"); at the compiler's free choice (106) that is 6, and four 6-wave workgroups then only fit when their
// waves happen to spread evenly over the SIMDs (measured: 3 resident, not the 4 the occupancy API answers).
// 96 SGPRs -> 7 waves per SIMD: 4 workgroups per CU, +5..9 % (interleaved A/B on one device).
#define LZ_MARCH_SGPR_ATTR __attribute__((amdgpu_num_sgpr(96)))
#ifndef LZ_MARCH_MIXV
#define LZ_MARCH_MIXV 1
#endif
#ifndef LZ_MARCH_SPLIT       // 16-bit samples, EXACT instances: split-weight chains (0: A/B builds only -- the single f32 chain)
#define LZ_MARCH_SPLIT 1
#endif
#ifndef LZ_MARCH_SPLIT_LSB1  // A/B builds only: the split-weight H chain in the LSB1 instances too (see MarchCfg::SPLIT)
#define LZ_MARCH_SPLIT_LSB1 0
#endif
#ifndef LZ_MARCH_MIXV_LSB1   // A/B builds only: the 16-bit-lane window in the LSB1 instances too (2 % slower there, see vpass)
#define LZ_MARCH_MIXV_LSB1 0
#endif

// Ablation bits (FrameGeom::debug_skip, LANCZOS_DEBUG_SKIP) exist only in builds made with -DLZ_PROFILE_BITS: in the production
// build every test of them is a compile-time false (the per-row `no_store` test alone was two scalar instructions and a
// branch per output row).
#ifdef LZ_PROFILE_BITS
#define LZ_DBG(g, bits) (((g).debug_skip & (bits)) != 0)
#else
#define LZ_DBG(g, bits) false
#endif

namespace lz {

// Tick geometry: NGRP groups of whole waves share a tick's rows in the V pass; MS input rows per tick.
// (Shapes that were built, measured and dropped -- role-specialised H / V waves with a carried register window, 3-wave
// workgroups with 6-row ticks, 8-wave workgroups, LDS-DMA input tiles, a second-stage integer-phase filter ... -- live in
// profiles/experiments/round3_march_variants.patch with their measurements in profiles/README.md.)
template <typename T, int C, int S, int A>
struct MarchShape {
    static constexpr int NGRP = 2;                                  // V groups (whole waves each)
    static constexpr int MS = 2 * A * NGRP;                         // input rows per tick
};
template <int A>
struct MarchShape<uint8_t, 3, 3, A> {  // 288 dword columns = 4.5 waves per V group (padded to 5): one group
    static constexpr int NGRP = 1;
    static constexpr int MS = 12;
};

template <typename T, int C_, int S_, int A_>
struct MarchCfg {
    using SH = MarchShape<T, C_, S_, A_>;
    using F = FastCfg<T, C_, S_, A_>;  // unit geometry (shared with the tile kernel)
    static constexpr int C = C_, S = S_, A = A_;
    static constexpr int SB = F::SB, TAPS = F::TAPS, P = F::P, UPR = F::UPR;
    static constexpr int NGRP = SH::NGRP;
    static constexpr int MS = SH::MS;
    static constexpr int MRG = MS / NGRP;
    static constexpr int NVT = F::TWB_OUT / 4;                       // V threads per group (dword columns)
    static constexpr int NU = MS * UPR;                              // H units per tick
    static constexpr int NVT_PAD = NGRP == 1 ? NVT : ((NVT + 63) / 64) * 64;  // V groups start on wave boundaries
    static constexpr int NT_V = NVT_PAD * NGRP;
    static constexpr int NT = (((NT_V > NU ? NT_V : NU) + 63) / 64) * 64;
    static constexpr int NWAVES = NT / 64;
    // ring rows: two ticks + the window.  A power of two where 32 rows do (index = row & 31); otherwise the next multiple of
    // 8 and a modulo -- at a = 4 that is 40 rows instead of 64: 24 KiB less LDS, two resident workgroups instead of one
    static constexpr int RS = (2 * MS + TAPS - 1) <= 32 ? 32 : ((2 * MS + TAPS - 1 + 7) / 8) * 8;
    static constexpr bool RS_POW2 = (RS & (RS - 1)) == 0;
    // slot of ring row `rel` = row - hb (>= -RS: the first ticks read rows above the chunk, whose results are never stored)
    static __device__ __forceinline__ int ring_slot(int rel) { return RS_POW2 ? (rel & (RS - 1)) : (rel + RS) % RS; }
    static constexpr int IN_PITCH = F::IN_PITCH, H_PITCH = F::H_PITCH, CPR = F::CPR;
    static constexpr int NCH = MS * CPR;                             // 16-byte chunks of one tick's input
    static constexpr int NLT = NT;                                   // threads that move the input rows
    static constexpr int LOAD_IT = (NCH + NLT - 1) / NLT;
    static constexpr int TIN_BYTES = LOAD_IT * NLT * 16;             // >= MS*IN_PITCH: every loading lane commits a chunk
    // worklist entries per wave: one round of candidates.  (A list for all VEC rounds with a single dense pass
    // measured 13 % SLOWER, interleaved A/B on one device: the extra 7 KiB of LDS costs residency.)
    static constexpr int WL_ROUND = 64 * F::UNIT_IN_DW;               // most entries one round can add
    static constexpr int WLW = WL_ROUND + 96;                         // sparse flags: all rounds share ONE dense pass
    static constexpr int NLISTS = (NU + 63) / 64;                     // only waves that run the H pass keep a list
    static constexpr int LDS_TIN = 2 * TIN_BYTES;                    // double buffered
    static constexpr int LDS_HBUF = RS * H_PITCH;
    static constexpr int LDS_WL = NLISTS * WLW * 2;
    // Scales whose per-index double weights are not the phase weights bit for bit (S = 3: x = xx / 3.0 is rounded) read them
    // from the global tap table in the (rare) fix-up; a slice of that table in LDS (18 KiB) bought nothing once the event path
    // was short and cost config 3 its third workgroup per CU (178 -> 161 us at 24 frames).
    // the exact chain's phase weights (TapTables::x_w: integer phase + S-1 interior phases, 256 B): a fix-up that fetches them
    // from global memory holds its workgroup's barrier for a memory round trip
    static constexpr int LDS_XW = kFastMaxS * kMaxTaps * 8;
    static constexpr int LDS_BYTES = LDS_TIN + LDS_HBUF + LDS_WL + LDS_XW;
    static constexpr int NNI = F::UNIT_OUT_S - P * C;                // non-integer-phase samples of a unit
    // S = 2: the half-phase weights are symmetric (L is even and x = m + 1/2 exactly), so a chain is 3 exact pair sums and
    // 3 fmafs -- the same 6 instructions, half the rounding steps: eps (and with it the near-integer fix-up rate) halves
    static constexpr bool SYM = S == 2;
    // the input rows of tick t + 3 are requested a whole tick (a V pass, the barrier, an H pass) before they are committed to
    // LDS instead of one H pass before: +4 staging registers through the V pass.  Pays where LDS, not registers, limits
    // residency (16-bit samples: config 5 634 -> 617 us); on config 2 the four registers cost the fourth workgroup (224 -> 272 us)
    static constexpr bool EARLY = SB == 2;
    // wave priority in the H pass / the V pass (0 elsewhere).  Round 1: (3, 2) beats equal priorities by 11-13 % (the H pass
    // feeds the ring every other wave's next V pass waits for).  Again with inputs from HBM (profiles/round2h_ab_wave_priority
    // .txt): config 2 (3,2) 210.5 / (3,1) 210.6 / (2,3) 212.2 / (3,3) 216.2 / (0,0) 217.4 us; config 3, whose V pass is
    // two thirds of the work, (2,3) 209.2 against (3,2) 213.8
    static constexpr int PRIO_H = S == 3 ? 2 : 3, PRIO_V = S == 3 ? 3 : 2;
    // S = 3: phase 2/3 mirrors phase 1/3 -- one set of weight registers serves both (fast_prepare)
    static constexpr bool MIRROR = S == 3;
    // H pass, u8: the chain carries eps - 0.5 and the RNE byte convert is the truncating store (3-op near-integer test:
    // fract(|acc|), subtract, unsigned min)
    static constexpr bool RNE_H = SB == 1;
    // near-integer flags per SAMPLE instead of per unit: with 16-bit samples the f32 window is 2 eps ~ 0.03 (eps scales
    // with the sample range), a quarter of all units hold a flagged sample, and redoing every sample of such a unit in
    // f64 was a third of config 5's time; 8-bit configurations flag one sample in 10^4 and keep the cheaper unit flag
    // 16-bit samples, S = 2: every weight split into an exactly-summing coarse part and a small remainder (lanczos_taps.hpp:
    // split_chain_prepare): the near-integer window shrinks from 2 eps = 0.025 to 5e-4 and the redo rate from one sample in 40 to
    // one in 2 000, so the H flag goes back to one per UNIT (a max and a min per sample instead of a compare, a select and an or
    // into a 64-bit mask) -- at +a fmafs and +5 epilogue instructions per computed sample.  Used by the EXACT instances, whose V
    // pass found an undecided sample in nearly every row of 128 and redid it in f64: config 5, 8 frames, gradient 2 021 -> 987 us,
    // noise 3 443 -> 2 366, blocks 1 904 -> 1 285, bit-identical.  The LSB1 instances (V pass untested) keep the single chain: with
    // the split H chain gradient 613.5 -> 623.5 us, blocks 620 -> 613, noise 885 -> 780 (profiles/round4z_ab_split_weight_chain_*).
    // (S = 3, whose chains are 2a fmafs long, does not fit its registers: 61 -> 168 VGPRs and spills.)
    static constexpr bool SPLIT = LZ_MARCH_SPLIT && SB == 2 && S == 2;   // (the kernel uses it where EXACT || LZ_MARCH_SPLIT_LSB1)
    static constexpr bool NEAR_PER_SAMPLE = SB == 2;                     // (... and the per-unit flag with it)
    // Register budget (2nd launch bound = waves per SIMD the compiler must leave room for).  6-wave workgroups land 2+2+1+1 on
    // the four SIMDs from a varying start, so four of them only fit reliably when a SIMD may hold SEVEN waves: 72 VGPRs.
    // Configurations the compiler leaves just above that step are told to stay under it -- the EXACT variants too, whose 9
    // spilled dwords sit on the cold f64 redo path (84 VGPRs, 388 us -> 72 VGPRs, 330 us per 32 frames of config 2).
    // The 8-bit RGB 3x a = 3 kernels (config 3: three 6-wave workgroups per CU by LDS) are told to leave room for SIX waves per SIMD
    // (80 VGPRs): the EXACT instance took 85 and lost its third workgroup per CU (472 -> 334 us per 32 frames, 7 spilled dwords
    // included); the LSB1 instance, at 75, gains the hoisted lane constants below (194.1 -> 192.5 us).  profiles/round4z_ab_config3_registers_and_residency.txt
#ifndef LZ_MARCH_C3_MIN_WAVES
#define LZ_MARCH_C3_MIN_WAVES 6
#endif
#ifndef LZ_MARCH_AUTO_MIN_WAVES   // A/B builds only (1): EVERY instance is told to leave room for the workgroups its LDS footprint allows
#define LZ_MARCH_AUTO_MIN_WAVES 0
#endif
    // waves per SIMD that nb resident workgroups need (uneven landing: see march_launch_t), and the largest nb the LDS allows
    // whose need the register files can meet at all (8 waves per SIMD; 7 under the 96-SGPR cap, 6 without it)
    static constexpr int need_waves(int nb) { return NWAVES % 4 == 0 ? nb * NWAVES / 4 : (nb * NWAVES + 3) / 4 + 1; }
    static constexpr int auto_min_waves() {
        const int lim = (160 * 1024 / LDS_BYTES) * NWAVES <= 20 ? 6 : 7;
        int nb = 160 * 1024 / LDS_BYTES;
        while (nb > 1 && need_waves(nb) > lim) nb--;
        return need_waves(nb) <= lim ? need_waves(nb) : 1;
    }
    static constexpr bool TUNED_C2 = SB == 1 && S == 2 && A == 3 && NT == 384;   // 4 x 6 waves per CU: 7 per SIMD
    static constexpr bool TUNED_C3 = SB == 1 && S == 3 && C == 3 && A >= 3;       // 3 x 6 waves per CU: 6 per SIMD (a = 2 fits by itself)
    // The instances whose own register count left a CU short of the workgroups its LDS admits, and that are faster when told to
    // make room (sweep of all 35 instances in both modes, profiles/round4z_sweep_min_waves.txt: -6 ... -25 %; the others move
    // by +-2 % or lose -- 8-bit RGB 4x a = 3 EXACT +12 % -- and keep the compiler's choice)
    static constexpr bool AUTO_OK = SB == 1 && ((S == 2 && A == 4 && C != 4) || (S == 3 && C == 4 && A == 3) || (S == 3 && C == 1 && A >= 3) ||
                                                (S == 4 && C == 1 && A == 3) || (S == 4 && C == 3 && A == 2));
    static constexpr int MIN_WAVES = TUNED_C2 ? 7 : (TUNED_C3 ? LZ_MARCH_C3_MIN_WAVES : ((LZ_MARCH_AUTO_MIN_WAVES || AUTO_OK) ? auto_min_waves() : 1));
    // Per-lane address parts of the input loads and of the H unit held in registers for a whole segment (a tick adds one scalar)
    // instead of being rebuilt every tick from the thread id (~35 VALU instructions, six of them quarter-rate multiplies).
    // Worth 1.5-2 % where the three registers fit under the occupancy step (config 2: 218 -> 214.5 us, profiles/
    // round3k_ab_hoisted_lane_constants.txt; config 3 once its kernels were told to stay at 80 VGPRs: 194.1 -> 192.5 us).
    static constexpr bool HOIST = (TUNED_C2 || TUNED_C3) && MIN_WAVES > 1;
    static_assert(MS % NGRP == 0, "V groups split a tick evenly");
    static_assert(MRG * S <= 64, "the EXACT-mode redo mask has one bit per output row of a V group");
    static_assert(NGRP == 1 || NVT_PAD % 64 == 0, "V groups must be whole waves");
    static_assert(RS >= 2 * MS + TAPS - 1, "ring holds two ticks plus the window");
    // worklist entry (16 bits) = row | unit | sample
    static constexpr int WL_SMP_BITS = F::UNIT_OUT_S <= 32 ? 5 : 6;
    static constexpr int WL_UNIT_BITS = UPR <= 16 ? 4 : (UPR <= 32 ? 5 : 6);
    static_assert(F::UNIT_OUT_S <= 64 && UPR <= 64 && MS <= (1 << (16 - WL_SMP_BITS - WL_UNIT_BITS)), "worklist entry fits 16 bits");
};

// RIDE: the launch also carries the in-place prefix rows as extra workgroups behind the marching ones (small batches:
// no separate k_prefix launch).  That variant rebuilds its per-lane indices every tick to stay inside the register
// budget; the batch variant (RIDE = false) holds them in registers, which is 2-3 % faster when the chip is full.
template <typename T, int C, int S, int A, bool EXACT, bool STAMP = false, bool RIDE = false>
__device__ __forceinline__ void march_body(const FrameGeom& g, const TapTables& t, const FastConsts& fc) {
    using K = MarchCfg<T, C, S, A>;
    using F = typename K::F;
    constexpr int TAPS = K::TAPS, SB = K::SB;
    constexpr bool SPLIT = MarchCfg<T, C, S, A>::SPLIT && (EXACT || LZ_MARCH_SPLIT_LSB1);
    constexpr bool NEAR_PER_SAMPLE = MarchCfg<T, C, S, A>::NEAR_PER_SAMPLE && !SPLIT;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* hbuf = smem + K::LDS_TIN;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    uint16_t* wlw = (uint16_t*)(smem + K::LDS_TIN + K::LDS_HBUF) + wave * K::WLW;  // this wave's list

    // =================================================================== the in-place prefix rows (extra workgroups)
    // Output rows [0, K) are a short sequential recurrence per sample column (full_TB.h:67-77; k_prefix).  As a launch of
    // their own they cost 8 us + a 2-5 us gap per step whatever the batch: a chain of dependent memory round trips.  Here
    // they ride at the END of the marching grid: these workgroups have the highest ids, are dispatched last and run in
    // the slots the first marching workgroups free, beside the tail of the march.  One column per thread, f64, the
    // arithmetic of k_prefix; everything is read through a laundered kernel-argument pointer so that nothing this path
    // needs is carried in SGPRs by the marching workgroups (that cost 14 spill reloads per tick when tried).
    if (RIDE && (int)blockIdx.x >= g.n_main) {
        const uint8_t* ka = (const uint8_t*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        const FrameGeom& G = *(const FrameGeom*)ka;
        const TapTables& TT = *(const TapTables*)(ka + ((sizeof(FrameGeom) + 7) & ~(size_t)7));
        static_assert(alignof(FrameGeom) == 8 && alignof(TapTables) == 8, "kernel-argument layout: g, then t, both 8-byte aligned");
        const int pw = (int)blockIdx.x - G.n_main;
        const int pframe = pw / G.prefix_blocks_per_frame;
        const int j = (pw - pframe * G.prefix_blocks_per_frame) * K::NT + tid;
        if (j >= G.out_w * C) return;
        const int PK = G.prefix_K, PM = G.prefix_M, PM2 = G.prefix_M2;
        T* hs = (T*)smem;                 // [PM2][NT]
        T* os = hs + PM2 * K::NT;         // [PM][NT]
        const uint8_t* pin = G.in + (size_t)pframe * G.in_frame_stride;
        uint8_t* pout = G.out + (size_t)pframe * G.out_frame_stride;
        {   // horizontal pass of the rows the prefix reads (full_TB.h:55-65)
            const int xx = j / C, c = j - xx * C;
            const int first = TT.h_first[xx];
            int idx[TAPS];
            double w[TAPS];
#pragma unroll
            for (int k = 0; k < TAPS; k++) {
                int i = first + k;
                i = i < 0 ? 0 : (i > G.in_w - 1 ? G.in_w - 1 : i);  // weight is 0 there
                idx[k] = i * C + c;
                w[k] = TT.h_w[(size_t)xx * TAPS + k];
            }
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
            for (int r = 0; r < PM2; r++) {  // (kept scalar and rolled: this path must not set the kernel's register count)
                const T* rowp = (const T*)(pin + (size_t)(r - G.in_row0) * G.in_pitch);
                T v[TAPS];
#pragma unroll
                for (int k = 0; k < TAPS; k++) v[k] = rowp[idx[k]];
                double sum = 0;
#pragma unroll
                for (int k = 0; k < TAPS; k++) sum += (double)v[k] * w[k];
                hs[r * K::NT + tid] = store_convert<T>(sum);
            }
        }
        // full_TB.h:69-76, xx descending: a tap at row i > xx sees the value already written there
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
        for (int xx = PM - 1; xx >= 0; xx--) {
            const int first = TT.v_first[xx];
            const double* wvp = TT.v_w + (size_t)xx * TAPS;
            double sum = 0;
#pragma unroll
            for (int k = 0; k < TAPS; k++) {
                int i = first + k;
                i = i < 0 ? 0 : (i > G.in_h - 1 ? G.in_h - 1 : i);  // weight 0 outside
                const T v = i > xx ? os[i * K::NT + tid] : hs[i * K::NT + tid];
                sum += (double)v * wvp[k];
            }
            os[xx * K::NT + tid] = store_convert<T>(sum);
        }
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
        for (int xx = 0; xx < PK; xx++) {
            if (xx < G.out_row0 || xx >= G.out_row0 + G.out_rows) continue;
            ((T*)(pout + (size_t)(xx - G.out_row0) * G.out_pitch))[j] = os[xx * K::NT + tid];
        }
        return;
    }

    // Which strip, which frame, which rows: one table entry per hardware block id, built on the host (march_build_table):
    //   * XCD-aware placement.  Workgroups go round-robin to the 8 XCDs by block id (verified: profiles/round1c census) and
    //     every XCD has its own L2.  A strip's 16-byte halo chunks pull in its neighbours' 128-byte lines, so with the natural
    //     order (neighbouring strips on different XCDs) every input line is fetched 5/3 times (measured: 164 MB read for
    //     99.5 MB of input).  The ids one XCD receives cover whole frames: neighbours share an L2 (160 -> 98.9 MB per launch).
    //   * rank-aware shares.  Inside an XCD the dispatcher fills the CUs breadth first (block id / 8 / 32 = rank of the
    //     workgroup on its CU) and the SIMD arbiter serves older waves first: with equal shares the 1st..4th workgroup of a CU
    //     took 179 / 189 / 197 / 207 us, the three of a CU that got only three 152 / 164 / 179 us, and the launch ended in a
    //     70 us tail of half-empty CUs (profiles/round2e_census_32frames.txt).  The table gives faster slots more rows.
    // A workgroup's share is up to g.wg_segs SEGMENTS (consecutive table entries; m_b >= m_e: empty): with one workgroup per CU
    // slot a share may run from the end of one (strip, frame) pair into the start of the next.  Everything below that depends
    // on the segment is a plain variable the lambdas capture by reference; load_segment() sets them.
    const int y_lo = g.out_row0 > g.skip_rows ? g.out_row0 : g.skip_rows;  // first output row stored at all
    const int y_hi = g.out_row0 + g.out_rows;                               // one past the last
    const int row_bytes = g.in_w * C * SB;
    const int gr_min = g.in_row0 > 0 ? g.in_row0 : 0;
    const int gr_max = (g.in_row0 + g.in_rows < g.in_h ? g.in_row0 + g.in_rows : g.in_h) - 1;
    // rows: m = floor(y/S) is the input row an output row hangs on.  The segment owns m in [m_b, m_e).
    int tx = 0, m_b = 0, m_e = 0;
    int hb = 0;      // first H row (= input row) the segment needs
    int h_last = 0;  // last one
    int ticks = 0;
    int tile_gb0 = 0;  // byte offset in the input row of LDS column 0
    // ---- input rows of one tick -> registers -> LDS.  Rows are 16-byte multiples (checked on the host), so a
    // chunk is never partly inside the image; outside lanes get an out-of-range offset and read zeros.
    __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(g.in), 0, (unsigned)(g.in_rows * g.in_pitch), 0x00020000);
    __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (unsigned)(g.out_rows * g.out_pitch), 0x00020000);
    auto load_segment = [&](int seg) -> bool {  // uniform
        const WgEntry we = g.wg_tab[(size_t)blockIdx.x * g.wg_segs + seg];  // scalar loads
        // (readfirstlane: inside the segment loop the compiler no longer proves these uniform by itself, and everything
        // derived from them -- row counters, the buffer descriptors -- would move into vector registers: 72 -> 83 VGPRs)
        const int frame = __builtin_amdgcn_readfirstlane(we.frame);
        tx = __builtin_amdgcn_readfirstlane(we.tx), m_b = __builtin_amdgcn_readfirstlane(we.m_b), m_e = __builtin_amdgcn_readfirstlane(we.m_e);
        if (m_b >= m_e) return false;
        hb = m_b - (A - 1);
        h_last = m_e - 1 + A;
        ticks = (h_last - hb + 1 + K::MS - 1) / K::MS;
        tile_gb0 = tx * F::TWP_IN * C * SB - F::LPB;
        irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(g.in + (size_t)frame * g.in_frame_stride), 0,
                                                  (unsigned)(g.in_rows * g.in_pitch), 0x00020000);
        orsrc = __builtin_amdgcn_make_buffer_rsrc(g.out + (size_t)frame * g.out_frame_stride, 0,
                                                  (unsigned)(g.out_rows * g.out_pitch), 0x00020000);
        return true;
    };

    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 pre[K::LOAD_IT];
    // Per-lane address parts live in registers for the whole segment; a tick adds one scalar.  (Rebuilding them every tick from
    // the thread id cost ~35 VALU instructions per tick, six of them quarter-rate multiplies (v_mul_lo_u32, v_mul_hi_i32,
    // v_mad_u64_u32) -- 12 % of a kernel whose VALU pipes are busy 100 % of the time, profiles/round3_sq_counters.json.)
    //   ld_const: row * in_pitch + byte offset of the lane's 16-byte chunk; 0x80000000 for a lane that never loads (beyond the
    //   tile, left / right of the image).  offset = ld_const + (first row of the tick - in_row0) * in_pitch.  Rows above the
    //   buffer give a negative sum = a huge unsigned offset, rows below it an offset >= num_records: the descriptor's range check
    //   returns zeros for both (a dropped tap), and 0x80000000 plus any such scalar stays out of range (buffers < 2^30 bytes:
    //   march_supports).
    unsigned ld_const[K::LOAD_IT];
    auto set_load_consts = [&]() {  // per segment (tile_gb0 changes with the strip)
        if (!K::HOIST) return;
        int t1 = tid;
        asm volatile("" : "+v"(t1));
#pragma unroll
        for (int it = 0; it < K::LOAD_IT; it++) {
            const int idx = t1 + it * K::NLT;
            const int row = idx / K::CPR, ch = idx - row * K::CPR;
            const int gb = tile_gb0 + 16 * ch;
            const bool ok = !LZ_DBG(g, 16) && idx < K::NCH && gb >= 0 && gb < row_bytes;
            ld_const[it] = ok ? (unsigned)(row * g.in_pitch + gb) : 0x80000000u;
            asm volatile("" : "+v"(ld_const[it]));
        }
    };
    auto issue_loads_to = [&](int tick, u32x4 (&dst)[K::LOAD_IT]) {
        if (K::HOIST) {
            // uniform: past the segment's last tick nothing is fetched (0x40000000 puts every lane out of range)
            const unsigned s_off = tick < ticks ? (unsigned)((hb + tick * K::MS - g.in_row0) * g.in_pitch) : 0x40000000u;
#pragma unroll
            for (int it = 0; it < K::LOAD_IT; it++) dst[it] = __builtin_amdgcn_raw_buffer_load_b128(irsrc, ld_const[it] + s_off, 0, 0);
            return;
        }
        int t1 = tid;
        asm volatile("" : "+v"(t1));  // (indices rebuilt from an opaque copy of the thread id: nothing held across the phases)
#pragma unroll
        for (int it = 0; it < K::LOAD_IT; it++) {
            const int idx = t1 + it * K::NLT;
            const int row = idx / K::CPR, ch = idx - row * K::CPR;
            const int gr = hb + tick * K::MS + row;
            const int gb = tile_gb0 + 16 * ch;
            const bool ok = !LZ_DBG(g, 16) && tick < ticks && idx < K::NCH && gr >= gr_min && gr <= gr_max &&
                            gr <= h_last && gb >= 0 && gb < row_bytes;
            const unsigned off = ok ? (unsigned)((gr - g.in_row0) * g.in_pitch + gb) : 0xffffffffu;
            dst[it] = __builtin_amdgcn_raw_buffer_load_b128(irsrc, off, 0, 0);
        }
    };
    auto issue_loads = [&](int tick) { issue_loads_to(tick, pre); };
    auto commit_from = [&](int buf, const u32x4 (&src)[K::LOAD_IT]) {
#pragma unroll
        for (int it = 0; it < K::LOAD_IT; it++)
            *(u32x4*)(smem + buf * K::TIN_BYTES + (tid + it * K::NLT) * 16) = src[it];
    };
    auto commit_loads = [&](int buf) {
#pragma unroll
        for (int it = 0; it < K::LOAD_IT; it++)
            *(u32x4*)(smem + buf * K::TIN_BYTES + (tid + it * K::NLT) * 16) = pre[it];
    };

    // ---- phase weights, pinned in VGPRs for the whole march.  Measured on gfx950 (scripts/probes/probe_valu3.hip): a VALU
    // instruction with an SGPR source issues at ~1.9 ns per wave, the same instruction with VGPR / inline-constant
    // sources at ~1.1 ns -- so the (S-1)*2a weights cost registers, not the constant bus.
    // S = 3: phase 2/3 is the mirror image of phase 1/3 (fast_prepare makes wf[2][k] == wf[1][2a-1-k] bit for bit and
    // prices the difference into eps): one set of registers serves both
    constexpr int NPHW = K::MIRROR ? 2 : S;
    float wv_[NPHW][TAPS];
#pragma unroll
    for (int ph = 1; ph < NPHW; ph++)
#pragma unroll
        for (int k = 0; k < TAPS; k++) {
            wv_[ph][k] = fc.wf[ph][k];
            asm volatile("" : "+v"(wv_[ph][k]));
        }
    auto wv = [&](int ph, int k) -> float { return (K::MIRROR && ph == 2) ? wv_[1][TAPS - 1 - k] : wv_[ph][k]; };
    float wsh_[NPHW][TAPS], wsl_[NPHW][TAPS];  // SPLIT: coarse / remainder halves of the H weights
    float sbias = fc.bias_s;
    if (SPLIT) {
#pragma unroll
        for (int ph = 1; ph < NPHW; ph++)
#pragma unroll
            for (int k = 0; k < (K::SYM ? A : TAPS); k++) {
                wsh_[ph][k] = fc.wsh[ph][k], wsl_[ph][k] = fc.wsl[ph][k];
                asm volatile("" : "+v"(wsh_[ph][k]), "+v"(wsl_[ph][k]));
            }
        asm volatile("" : "+v"(sbias));
    }
    auto wsh = [&](int ph, int k) -> float { return (K::MIRROR && ph == 2) ? wsh_[1][TAPS - 1 - k] : wsh_[ph][k]; };
    auto wsl = [&](int ph, int k) -> float { return (K::MIRROR && ph == 2) ? wsl_[1][TAPS - 1 - k] : wsl_[ph][k]; };
    constexpr uint32_t HALF = SB == 1 ? 0x80u : 0x8000u;
    const uint32_t addc1 = (uint32_t)fc.vlim < HALF - 1 ? HALF - 1 - (uint32_t)fc.vlim : 0;
    // the chain bias and the SWAR masks live in VGPRs for the same reason (a literal is a constant-bus read too)
    // SB == 1: the H chain carries eps - 0.5 (see the byte convert in hpass); SYM: the paired chain's (smaller) eps
    float hbias = K::RNE_H ? (K::SYM ? fc.vbias_rne_p : fc.vbias_rne) : (K::SYM ? fc.bias_p : fc.bias);
    const float near2 = K::SYM ? fc.near2_p : fc.near2;
    constexpr uint32_t LOW = SB == 1 ? 0x7f7f7f7fu : 0x7fff7fffu;
    constexpr uint32_t TOP = SB == 1 ? 0x80808080u : 0x80008000u;
    const uint32_t addc = SB == 1 ? addc1 * 0x01010101u : addc1 * 0x00010001u;
    asm volatile("" : "+v"(hbias));

    // =================================================================== HPASS + FIXUP of one tick
    // The H unit of this thread.  Its output sits at byte tid * UNIT_OUT bytes of a ring-row block (a ring row is UPR units wide:
    // row * H_PITCH + u * UNIT_OUT_DW * 4 == tid * UNIT_OUT_DW * 4), one full-rate multiply per tick, and that offset doubles as
    // the row test (unit rows r < n <=> offset < n * H_PITCH).  The input window's LDS offset (rows are IN_PITCH apart, not a
    // multiple of the unit) is kept in a register.
    static_assert(K::H_PITCH == K::UPR * F::UNIT_OUT_DW * 4, "a ring row is exactly UPR units");
    unsigned h_in_const = 0;
    if (K::HOIST) {
        const int row = tid / K::UPR, u = tid % K::UPR;
        h_in_const = (unsigned)((row * (K::IN_PITCH / 4) + F::WIN_DW0 + u * F::UNIT_IN_DW) * 4);
        asm volatile("" : "+v"(h_in_const));
    }
    auto hpass = [&](int tick) {
        const uint8_t* tin = smem + (tick & 1) * K::TIN_BYTES;
        const int h0 = hb + tick * K::MS;           // first H row of the tick
        const int rows_here = h_last - h0 + 1 < K::MS ? h_last - h0 + 1 : K::MS;  // uniform; <= 0: nothing left
        int t3 = tid;
        if (!K::HOIST) asm volatile("" : "+v"(t3));
        const unsigned h_out_const = (unsigned)t3 * (unsigned)(F::UNIT_OUT_DW * 4);  // (HOIST: tid < 1024, one v_mul_u32_u24)
        const bool unit_ok = rows_here > 0 && h_out_const < (unsigned)(rows_here * K::H_PITCH) && (K::NU >= K::NT || t3 < K::NU) &&
                             !LZ_DBG(g, 1);
        uint32_t im = 0;      // undecided integer-phase samples: bit (8*e*SB + i) <-> own input sample i*VEC + e
        bool near = false;    // some non-integer-phase sample of the unit is within eps of an integer
        unsigned long long nearmask = 0;  // NEAR_PER_SAMPLE: which ones (bit = output sample of the unit)
        if (unit_ok) {
            const uint32_t* tin32 = (const uint32_t*)tin;
            uint32_t wd[F::NW];
            const uint32_t* wp = K::HOIST ? (const uint32_t*)(tin + h_in_const)
                                          : tin32 + (t3 / K::UPR) * (K::IN_PITCH / 4) + F::WIN_DW0 + (t3 % K::UPR) * F::UNIT_IN_DW;
#pragma unroll
            for (int i = 0; i < F::NW; i++) wd[i] = wp[i];
            // Output dwords start as the integer-phase samples (copies of input bytes: one v_perm_b32 when they come
            // from <= 2 window dwords); the computed samples are inserted as they are produced.  One CHANNEL at a
            // time, fenced, so that only WIN_PX converted samples are live at once (registers, not ILP, are scarce:
            // <= 64 VGPRs lets every SIMD hold 8 waves, which is what makes 4 workgroups per CU always placeable).
            // ring slot of the unit's row = (slot of the tick's first row + row) mod RS: one add and an unsigned min
            unsigned hofs = h_out_const + (unsigned)(K::ring_slot(h0 - hb) * K::H_PITCH);
            hofs = hofs < hofs - (unsigned)(K::RS * K::H_PITCH) ? hofs : hofs - (unsigned)(K::RS * K::H_PITCH);
            uint32_t* hp = (uint32_t*)(hbuf + hofs);
            uint32_t ow[F::UNIT_OUT_DW];
#pragma unroll
            for (int i = 0; i < F::UNIT_OUT_DW; i++) {
                uint32_t w = 0;
                if (SB == 1) {
                    int src[4], d0 = -1, d1 = -1;
                    bool any_raw = false, perm_ok = true;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int o = i * 4 + e, q = o / C, c = o % C;
                        src[e] = -1;
                        if (q % S == 0) {
                            const int b = F::MIS + ((q / S + A - 1) * C + c);
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
                                uint32_t sl = 0x0c;  // constant 0
                                if (src[e] >= 0) sl = ((src[e] >> 2) == d0 ? 0 : 4) + (src[e] & 3);
                                sel |= sl << (8 * e);
                            }
                            w = __builtin_amdgcn_perm(wd[d1 < 0 ? d0 : d1], wd[d0], sel);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; e++)
                                if (src[e] >= 0) w |= ((wd[src[e] >> 2] >> (8 * (src[e] & 3))) & 0xffu) << (8 * e);
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const int o = i * 2 + e, q = o / C, c = o % C;
                        if (q % S == 0) w |= win_sample<T, F::MIS, F::NW>(wd, (q / S + A - 1) * C + c) << (16 * e);
                    }
                }
                ow[i] = w;
            }
            float dmin = 1.0f;
            uint32_t dminu = 0x7f800000u;  // SB == 1: smallest fract(acc) - 0.5 seen, compared as unsigned bit patterns
            unsigned pend[K::P * S];       // 16-bit samples, even C: the even channel's sample waits for its dword partner
#pragma unroll
            for (int c = 0; c < C; c++) {
                float fch[F::WIN_PX];
#pragma unroll
                for (int t = 0; t < F::WIN_PX; t++) {
                    fch[t] = (float)win_sample<T, F::MIS, F::NW>(wd, t * C + c);
                    // SYM: keep the conversion a v_cvt_f32_ubyteN.  Left alone, the compiler rewrites (float)a + (float)b as
                    // (float)(a + b) with SDWA byte adds + v_cvt_f32_u32 -- fewer instructions, 45 % more time (measured)
                    if (K::SYM) asm volatile("" : "+v"(fch[t]));
                }
#pragma unroll
                for (int q = 0; q < K::P * S; q++) {
                    const int p = q / S, ph = q % S;
                    if (ph == 0) continue;
                    const int o = q * C + c;
                    if (SPLIT) {
                        // hi: exact (every product and partial sum is an integer multiple of 2^-q below 2^24).  lo: an f32 chain
                        // of numbers 2^-(q+1) as large, started at fract(hi) + eps.  floor(hi) + floor(lo) is then floor(sum + eps')
                        // with eps' in (0, 2 eps); fract(lo) < 2 eps = undecided
                        float ah = 0.0f;
#pragma unroll
                        for (int k = 0; k < A; k++) ah = __builtin_fmaf(wsh(ph, k), fch[p + k] + fch[p + TAPS - 1 - k], ah);
                        const float fh = __builtin_amdgcn_fractf(ah);   // exact: a multiple of 2^-q in [0, 1)
                        float al = fh + sbias;
#pragma unroll
                        for (int k = 0; k < A; k++) al = __builtin_fmaf(wsl(ph, k), fch[p + k] + fch[p + TAPS - 1 - k], al);
                        const float jt = __builtin_floorf(al);
                        const float r = (ah - fh) + jt;             // exact: integers below 2^24
                        // (al - floor(al) is exact wherever it is small.  A sum below 1 stores 0 whatever it is: r <= 0 puts 1 - r >= 1
                        // in the minimum instead -- black regions and empty channels, whose sums are exactly 0 and whose lo halves are
                        // exactly eps, must not flag)
                        dmin = __builtin_fminf(dmin, __builtin_fmaxf(al - jt, 1.0f - r));
                        unsigned uv;                                // the hardware convert saturates (negative -> 0); a C++ cast of a
                        asm("v_cvt_u32_f32 %0, %1" : "=v"(uv) : "v"(r));  // negative float is undefined, hence the instruction itself
                        if (C % 2 == 0) {
                            // the two halves of an output dword are the same pixel's channels c, c + 1: one saturating pack
                            if (c % 2 == 0) pend[q] = uv;
                            else ow[o / 2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_u16(pend[q], uv));
                        } else {
                            ow[o / 2] |= (uv < 65535u ? uv : 65535u) << (16 * (o % 2));
                        }
                        continue;
                    }
                    float acc = hbias;
                    if (K::SYM) {
#pragma unroll
                        for (int k = 0; k < A; k++)  // outside in; the pair sums are exact (<= 2 * max sample)
                            acc = __builtin_fmaf(wv(ph, k), fch[p + k] + fch[p + TAPS - 1 - k], acc);
                    } else {
#pragma unroll
                        for (int j = 0; j < TAPS; j++) {
                            const int k = f32_tap_order(j, TAPS);  // outside in: the bound of fc.bias assumes this order
                            acc = __builtin_fmaf(wv(ph, k), fch[p + k], acc);
                        }
                    }
                    if (K::RNE_H) {
                        // acc = sum + (eps - 0.5) + chain error, |chain error| < eps: the saturating round-to-nearest-even
                        // byte convert is floor(sum + eps') with eps' in (0, 2 eps) -- the reference's truncating store --
                        // unless sum lies within 2 eps below an integer, i.e. fract(acc) in [0.5, 0.5 + 2 eps): undecided.
                        // fract and the subtraction are exact; a negative difference has its sign bit set and loses the
                        // unsigned minimum.  (Sums outside [0, max] may flag spuriously: the exact chain clamps like the store.)
                        // acc < 0 (sum + eps' < 1/2): the store is 0 whatever the sum.  fract(|acc|) keeps those off the undecided
                        // list for free (the |.| is a source modifier; v_max_f32 is a slow-class op on gfx950): black regions
                        // have acc = eps - 0.5 exactly, fract(|acc|) = 0.5 - eps, g < 0.  (Negative sums within 2 eps below
                        // -(k + 1/2) flag spuriously: the exact chain clamps like the store.)
                        const float g = __builtin_amdgcn_fractf(__builtin_fabsf(acc)) - 0.5f;
                        const uint32_t gu = __builtin_bit_cast(uint32_t, g);
                        dminu = gu < dminu ? gu : dminu;
                        ow[o / 4] = __builtin_amdgcn_cvt_pk_u8_f32(acc, o % 4, ow[o / 4]);
                    } else {
                        // 16-bit samples: acc = sum + eps.  Below 1/2 the store is 0 whatever the sum (and black regions, whose
                        // sums are exactly 0, must not look "within eps of an integer"); the float -> u32 convert truncates
                        // and an unsigned min saturates: 2.5 slow-class ops per sample instead of 4
                        const float m = __builtin_fmaxf(acc, 0.5f);
                        if (NEAR_PER_SAMPLE) nearmask |= (unsigned long long)(__builtin_amdgcn_fractf(m) < near2) << o;
                        else dmin = __builtin_fminf(dmin, __builtin_amdgcn_fractf(m));
                        unsigned uv = (unsigned)m;                      // v_cvt_u32_f32: truncation
                        if (C % 2 == 0) {
                            // the two halves of an output dword are the same pixel's channels c, c + 1 (an even channel waits for
                            // its partner): one saturating pack instead of two unsigned minima, a shift and two ors
                            if (c % 2 == 0) pend[q] = uv;
                            else ow[o / 2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_u16(pend[q], uv));
                        } else {
                            uv = uv < 65535u ? uv : 65535u;
                            ow[o / 2] |= uv << (16 * (o % 2));
                        }
                    }
                }
                // (16-bit instances, whose residency LDS limits, without this fence: 624.9 against 624.5 us on config 5 -- that
                // configuration is not VALU-bound; profiles/round4z_ab_split_weight_chain_config5.txt)
                __builtin_amdgcn_sched_barrier(0);
            }
            near = NEAR_PER_SAMPLE ? nearmask != 0 : (K::RNE_H ? dminu < __builtin_bit_cast(uint32_t, near2) : dmin < near2);
            if (SPLIT) near = dmin < fc.near2_s;
            // widest aligned LDS stores the unit allows
            if (F::UNIT_OUT_DW % 2 == 0) {
#pragma unroll
                for (int i = 0; i + 1 < F::UNIT_OUT_DW; i += 2) *(uint2*)(hp + i) = make_uint2(ow[i], ow[i + 1]);
            } else {
#pragma unroll
                for (int i = 0; i < F::UNIT_OUT_DW; i++) hp[i] = ow[i];
            }
            // integer-phase candidates (SWAR on the unit's own input dwords)
            if (fc.vlim > 0) {
                constexpr int OWN_B0 = F::MIS + (A - 1) * C * SB;   // window byte offset of the own pixels
                constexpr int OWN_DW0 = OWN_B0 / 4;
                static_assert(OWN_B0 % 4 == 0 && F::UNIT_IN_DW <= 7, "own pixels are whole dwords");
                uint32_t loose[F::UNIT_IN_DW], any = 0;
#pragma unroll
                for (int i = 0; i < F::UNIT_IN_DW; i++) {
                    {   // 1 <= v0 <= vlim, per byte/halfword lane (swar_in_1_vlim with the masks in registers)
                        const uint32_t x = wd[OWN_DW0 + i], t7 = x & LOW;
                        loose[i] = ((t7 + LOW) | x) & ~((t7 + addc) | x) & TOP;
                    }
                    any |= loose[i];
                }
                if (!fc.tight) {
#pragma unroll
                    for (int i = 0; i < F::UNIT_IN_DW; i++) im |= (loose[i] >> (8 * SB - 1)) << i;
                } else if (__any(any != 0)) {
                    // tight filter: the chain can only leave v0 through the negative taps at +-2 pixels; if both
                    // neighbours are <= 2*v0 (< 3.44 * 2^ceil(log2 v0), the proven bound) v0 stays
#pragma unroll
                    for (int i = 0; i < F::UNIT_IN_DW; i++) {
                        auto window_word = [&](int bo) -> uint32_t {  // 4 bytes at window byte offset bo (static)
                            return (bo & 3) == 0 ? wd[bo >> 2]
                                                 : __builtin_amdgcn_alignbyte(wd[(bo >> 2) + 1], wd[bo >> 2], bo & 3);
                        };
                        const uint32_t x = wd[OWN_DW0 + i];
                        const uint32_t c2 = ((x & LOW) << 1) | TOP;
                        const uint32_t nm = window_word(OWN_B0 + 4 * i - 2 * C * SB);
                        const uint32_t np = window_word(OWN_B0 + 4 * i + 2 * C * SB);
                        const uint32_t pm = (c2 - (nm & LOW)) & ~nm;   // top bit: n(-2) <= 2*v0
                        const uint32_t pp = (c2 - (np & LOW)) & ~np;   // top bit: n(+2) <= 2*v0
                        const uint32_t tight = loose[i] & ~(pm & pp) & TOP;
                        im |= (tight >> (8 * SB - 1)) << i;
                    }
                    // (A second stage -- necessary conditions on the +-1 neighbours too, proven in lanczos_taps.cpp:
                    // integer_phase_tight2 and checked exhaustively in tests/test_integer_phase_filter.py -- takes the candidates on
                    // noise from 23 % to 8 % of the integer-phase samples, but its ~170 SWAR instructions per unit cost more than
                    // the fix-ups they save: noise 326 -> 345 us on config 2.  profiles/experiments/round3_march_variants.patch.)
                }
            }
        }
#ifdef LZ_MARCH_DIAG_NONEAR   // timing diagnostics only (results are then wrong): no near-integer flags
        near = false, nearmask = 0;
#endif
        if (LZ_DBG(g, 32 | 256 | 512)) {  // profiling bits: 32 no flags at all, 256 no near flags, 512 no integer flags
            if (LZ_DBG(g, 32 | 512)) im = 0;
            if (LZ_DBG(g, 32 | 256)) near = false, nearmask = 0;
        }

        // ---- wave-private compaction of the undecided samples, then the exact chain, densely
        if (__any(im != 0 || near) && !LZ_DBG(g, 2)) {
            int cnt = 0;  // uniform
            auto flush = [&]() {
                const T* tinT = (const T*)tin;
                T* hbufT = (T*)hbuf;
                for (int i = lane; i < cnt; i += 64) {
                    const unsigned e = wlw[i];
                    const int erow = e >> (K::WL_SMP_BITS + K::WL_UNIT_BITS), eu = (e >> K::WL_SMP_BITS) & ((1 << K::WL_UNIT_BITS) - 1),
                              o = e & ((1 << K::WL_SMP_BITS) - 1);
                    const int q = o / C, c = o - q * C;          // output pixel / channel inside the unit
                    const int xl = eu * (K::P * S) + q;          // output pixel inside the strip
                    const int xx = tx * F::TWP_OUT + xl;
                    if (xx >= g.out_w) continue;
                    const int fl = xl / S;                        // floor(x) - P0
                    const T* rp = tinT + (erow * K::IN_PITCH + F::LPB) / SB + (fl - A + 1) * C + c;
                    double sum = 0;
                    const double* xw = (const double*)(smem + K::LDS_TIN + K::LDS_HBUF + K::LDS_WL);  // exact-chain weights
                    if (xl - fl * S == 0) {                       // integer phase: the same weights everywhere
                        if (fc.skip_last) {  // flagged samples have v0 >= 1: the ~1e-33 tap at x-i = -a is inert, L(0) is 1
#pragma unroll
                            for (int k = 0; k < TAPS - 1; k++)
                                sum += k == A - 1 ? (double)rp[k * C] : (double)rp[k * C] * xw[k];
                        } else {
#pragma unroll
                            for (int k = 0; k < TAPS; k++) sum += (double)rp[k * C] * xw[k];
                        }
                    } else if (fc.phase_exact_h) {
                        // every index of a phase has the same weights: kernel arguments, no table gather on the
                        // critical path (a wave that finds a near-integer sum holds its workgroup's barrier)
                        const int ph = xl - fl * S;
#pragma unroll
                        for (int k = 0; k < TAPS; k++) sum += (double)rp[k * C] * xw[ph * kMaxTaps + k];
                    } else {
                        const double* w = t.h_w + (size_t)xx * TAPS;
#pragma unroll
                        for (int k = 0; k < TAPS; k++) sum += (double)rp[k * C] * w[k];
                    }
                    const int slot = K::ring_slot(h0 + erow - hb);
                    hbufT[slot * (K::H_PITCH / SB) + xl * C + c] = store_convert<T>(sum);
                }
                cnt = 0;
            };
            const int row = t3 / K::UPR, u = t3 % K::UPR;  // (only needed on this rare path)
            const unsigned ent0 = ((unsigned)row << (K::WL_SMP_BITS + K::WL_UNIT_BITS)) | ((unsigned)u << K::WL_SMP_BITS);
            auto append = [&](bool flag, int o) {  // o wave-uniform: output sample of the unit
                const unsigned long long m = __ballot(flag);
                if (m == 0) return;
                if (flag)
                    wlw[cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0))] =
                        (uint16_t)(ent0 + o);
                cnt += __popcll(m);
            };
            // integer-phase candidates: one round per byte/halfword lane e of the own dwords (runtime loop so the
            // exact chain in flush() is not replicated per sample position); a round adds <= 64*UNIT_IN_DW entries
            if (__any(im != 0)) {
#pragma unroll 1
                for (int e = 0; e <= F::VEC; e++) {  // the extra trip only drains the list (one flush() site)
                    if (e < F::VEC) {
#pragma unroll
                        for (int i = 0; i < F::UNIT_IN_DW; i++) {
                            const int si = i * F::VEC + e, p = si / C, c = si - p * C;  // own input sample -> its integer phase
                            append((im >> (8 * SB * e + i)) & 1, (p * S) * C + c);
                        }
                    }
                    if (cnt > K::WLW - K::WL_ROUND || (e == F::VEC && cnt > 0)) flush();
                }
            }
            if (NEAR_PER_SAMPLE) {
                if (__any(near)) {  // every flagged non-integer-phase sample
#pragma unroll 1
                    for (int o = 0; o < F::UNIT_OUT_S; o++) {
                        if ((o / C) % S == 0) continue;
                        append((bool)((nearmask >> o) & 1), o);
                        if (cnt > K::WLW - 64) flush();
                    }
                    if (cnt > 0) flush();
                }
            } else {
                // rare (one unit in ~10^3), but a wave that handles one holds its workgroup's barrier: no ballot rounds -- the
                // first NNI lanes write the flagged unit's non-integer-phase samples straight into the list
                unsigned long long nm = __ballot(near);
                if (nm) {
                    const int qi = lane / C;
                    const int o_lane = ((qi / (S - 1)) * S + qi % (S - 1) + 1) * C + lane % C;  // lane-th non-integer-phase sample
                    do {
                        const int src = __builtin_ctzll(nm);
                        nm &= nm - 1;
                        const unsigned e0 = (unsigned)__builtin_amdgcn_readlane((int)ent0, src);
                        if (lane < K::NNI) wlw[cnt + lane] = (uint16_t)(e0 + o_lane);
                        cnt += K::NNI;
                        if (cnt > K::WLW - K::NNI) flush();
                    } while (nm);
                    if (cnt > 0) flush();
                }
            }
        }
    };

    // =================================================================== VPASS of one tick
    const int grp_w = K::NGRP == 1 ? 0 : wave * 64 / K::NVT_PAD;  // the wave's V group (NGRP > 1: groups are whole waves)
    float vbias = SB == 1 ? (K::SYM ? fc.vbias_rne_p : fc.vbias_rne) : (K::SYM ? fc.bias_p : fc.bias);
    asm volatile("" : "+v"(vbias));

    // MIXV (8-bit, symmetric half-phase weights, EXACT instances): the V window is kept as 16-bit integer lanes instead of floats.
    // A ring dword splits into its even and odd bytes (two VGPRs per row instead of four floats and the packed copy: 18 registers
    // fewer through the row loop); a pair sum is ONE v_add_u32 for two samples (<= 510 per lane, no carry); and a 16-bit integer n
    // IS the f16 denormal n * 2^-24, which v_fma_mix_f32 widens exactly: fma(w * 2^24, n * 2^-24, acc) is the same single
    // rounding as fmaf(w, (float)n, acc) -- results are bit for bit those of the float window (profiles/round2_probe_mix.txt,
    // profiles/round4g_*: checked against the float window's output on the GPU).  For the LSB1 instances the float window is 2 %
    // faster (profiles/round4g_ab_mix_window_and_counted_commit.txt); the EXACT instances, whose exactness tests push the float
    // version over the 72-register step (9 dwords spilled INSIDE the row loop, each reload a memory round trip in front of the
    // row's store), run without spills this way.
    constexpr bool MIXV = LZ_MARCH_MIXV && SB == 1 && K::SYM && (EXACT || LZ_MARCH_MIXV_LSB1) && K::MIN_WAVES > 1;
    float wmix[A];
    uint32_t mixmask = 0x00ff00ffu;
    if (MIXV) {
#pragma unroll
        for (int k = 0; k < A; k++) {
            wmix[k] = wv_[1][k] * 16777216.0f;  // exact
            asm volatile("" : "+v"(wmix[k]));
        }
        asm volatile("" : "+v"(mixmask));
    }
    auto mix_lo = [](float w, uint32_t p, float acc) -> float {
        float r;
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(w), "v"(p), "v"(acc));
        return r;
    };
    auto mix_hi = [](float w, uint32_t p, float acc) -> float {
        float r;
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(w), "v"(p), "v"(acc));
        return r;
    };
    uint32_t wev[TAPS], wod[TAPS];   // MIXV: even / odd bytes of the window rows

    float win[TAPS][F::VEC];
    uint32_t raw[TAPS];
    auto vpass = [&](int tick) {
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        const int grp = K::NGRP == 1 ? (t2 < K::NVT ? 0 : 1) : grp_w;
        const int col = t2 - grp * K::NVT_PAD;
        const unsigned col_b = (unsigned)(tx * F::TWB_OUT + col * 4);
        const bool col_ok = grp < K::NGRP && col < K::NVT && col_b + 4 <= (unsigned)(g.out_w * C * SB);
        if (!col_ok || LZ_DBG(g, 4)) return;
        constexpr int HP = K::H_PITCH / 4;
        // m handled this tick: [m_lo, m_lo + MS) with m_lo = m_b - (2a-1) + tick*MS; this group's share:
        const int m_g = m_b - (TAPS - 1) + tick * K::MS + grp * K::MRG;
        if (m_g + K::MRG <= m_b || m_g >= m_e) return;  // uniform
        const uint32_t* hcol = (const uint32_t*)hbuf + col;
        auto unpack = [&](int slot_i, uint32_t w) {
            if (MIXV) {
                wev[slot_i] = w & mixmask;
                wod[slot_i] = (w >> 8) & mixmask;
                return;
            }
            raw[slot_i] = w;
#pragma unroll
            for (int e = 0; e < F::VEC; e++) {
                win[slot_i][e] = (float)((w >> (8 * SB * e)) & F::SMASK);
                if (K::SYM) asm volatile("" : "+v"(win[slot_i][e]));  // see hpass: no integer pair sums
            }
        };
        auto ring = [&](int r) { return hcol[K::ring_slot(r - hb) * HP]; };  // H row r of this column
        auto rawrow = [&](int slot_i) -> uint32_t { return MIXV ? (wev[slot_i] | (wod[slot_i] << 8)) : raw[slot_i]; };
        // rows m_g-a+1 .. m_g+a-1 seed the window; rows before hb were never produced: only read for m < m_b,
        // whose outputs are not stored
#pragma unroll
        for (int k = 0; k < TAPS - 1; k++) unpack(k, ring(m_g - A + 1 + k));
        // every output row of this group's share lies inside the chunk and the stored range: no per-row tests
        const bool interior = m_g >= m_b && m_g + K::MRG <= m_e && m_g * S >= y_lo && (m_g + K::MRG) * S <= y_hi;
        const bool no_store = LZ_DBG(g, 8);
        int soff = (m_g * S - g.out_row0) * g.out_pitch;  // scalar byte offset of the current output row
        // two instances of the row loop: the interior one (the common case) carries no range tests at all --
        // the per-row scalar compare/select chains cost more issue slots than the arithmetic they guarded
        unsigned long long redo_mask = 0;  // EXACT: output rows (relative to m_g * S) to redo in f64
        auto rows = [&](auto checked_c) {
        constexpr bool CHECKED = decltype(checked_c)::value;
        for (int mm = 0; mm < K::MRG; mm += TAPS) {
#pragma unroll
            for (int i = 0; i < TAPS; i++) {
                if (mm + i >= K::MRG) break;
                const int m = m_g + mm + i;
                unpack((i + TAPS - 1) % TAPS, ring(m + A));
#pragma unroll
                for (int ph = 0; ph < S; ph++) {
                    const int y = m * S + ph;
                    uint32_t packed;
                    bool undecided = false;
                    if (ph == 0) {
                        packed = rawrow((i + A - 1) % TAPS);
                        if (EXACT && fc.vlim > 0 && SB == 1 && A >= 3 && fc.tight) {
                            // the H pass's byte-parallel tests on the packed ring dwords: 1 <= v0 <= vlim (and only in
                            // waves that hold such a sample at all) a row two above or below brighter than 2*v0
                            const uint32_t x = packed, t7 = x & LOW;
                            const uint32_t lo = ((t7 + LOW) | x) & ~((t7 + addc) | x) & TOP;
                            if (__any(lo != 0)) {
                                const uint32_t c2 = (t7 << 1) | TOP;
                                const uint32_t nm = rawrow((i + A - 3) % TAPS), np = rawrow((i + A + 1) % TAPS);
                                const uint32_t pm = (c2 - (nm & LOW)) & ~nm;   // top bit: row m-2 <= 2*v0
                                const uint32_t pp = (c2 - (np & LOW)) & ~np;   // top bit: row m+2 <= 2*v0
                                undecided = (lo & ~(pm & pp) & TOP) != 0;
                            }
                        } else if (EXACT && fc.vlim > 0) {
                            // the integer-phase chain can leave v0 only if 1 <= v0 <= vlim AND (fc.tight, proven in
                            // fast_prepare for these weights) a row two above or below is brighter than 2*v0 -- without
                            // the second test nearly every row of a natural image would take the f64 path
                            const float vl = (float)fc.vlim;
#pragma unroll
                            for (int e = 0; e < F::VEC; e++) {
                                auto wf = [&](int slot_i) -> float {   // (MIXV keeps no float window: the sample comes out of the split rows)
                                    return MIXV ? (float)((rawrow(slot_i) >> (8 * SB * e)) & F::SMASK) : win[slot_i][e];
                                };
                                const float c0 = wf((i + A - 1) % TAPS);
                                bool fl = c0 >= 1.0f && c0 <= vl;
                                if (A >= 3 && fc.tight)
                                    fl = fl && (wf((i + A - 3) % TAPS) > 2.0f * c0 || wf((i + A + 1) % TAPS) > 2.0f * c0);
                                undecided |= fl;
                            }
                        }
                    } else if (LZ_DBG(g, 1024)) {  // profiling bit 1024: computed rows are copies too (loads + stores, no V arithmetic)
                        packed = rawrow((i + A - 1) % TAPS);
                    } else {
                        packed = 0;
                        float accs[F::VEC];
                        if (SPLIT && EXACT) {
                            // 16-bit samples: the H pass's split-weight chain (exact coarse half, small remainder started at
                            // fract(hi) + eps): the undecided window is 5e-4 instead of 0.025, where nearly every row of 128
                            // samples held an undecided one and was redone in f64
                            float ftmin = 1.0f;
                            unsigned uvv[F::VEC];
#pragma unroll
                            for (int e = 0; e < F::VEC; e++) {
                                float ah = 0.0f;
#pragma unroll
                                for (int k = 0; k < A; k++)
                                    ah = __builtin_fmaf(wsh(ph, k), win[(i + k) % TAPS][e] + win[(i + TAPS - 1 - k) % TAPS][e], ah);
                                const float fh = __builtin_amdgcn_fractf(ah);
                                float al = fh + sbias;
#pragma unroll
                                for (int k = 0; k < A; k++)
                                    al = __builtin_fmaf(wsl(ph, k), win[(i + k) % TAPS][e] + win[(i + TAPS - 1 - k) % TAPS][e], al);
                                const float jt = __builtin_floorf(al);
                                const float r = (ah - fh) + jt;
                                ftmin = __builtin_fminf(ftmin, __builtin_fmaxf(al - jt, 1.0f - r));   // (r <= 0: stores 0 whatever the sum)
                                asm("v_cvt_u32_f32 %0, %1" : "=v"(uvv[e]) : "v"(r));   // saturating: negative -> 0
                            }
                            packed = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_u16(uvv[0], uvv[F::VEC - 1]));
                            undecided = ftmin < fc.near2_s;
                        } else {
                        if (MIXV) {
#pragma unroll
                            for (int e = 0; e < 4; e++) accs[e] = vbias;
#pragma unroll
                            for (int k = 0; k < A; k++) {  // outside in, as the float chain
                                const uint32_t pe = wev[(i + k) % TAPS] + wev[(i + TAPS - 1 - k) % TAPS];
                                const uint32_t po = wod[(i + k) % TAPS] + wod[(i + TAPS - 1 - k) % TAPS];
                                accs[0] = mix_lo(wmix[k], pe, accs[0]);
                                accs[1] = mix_lo(wmix[k], po, accs[1]);
                                accs[2] = mix_hi(wmix[k], pe, accs[2]);
                                accs[3] = mix_hi(wmix[k], po, accs[3]);
                            }
                        }
#pragma unroll
                        for (int e = 0; e < F::VEC && !MIXV; e++) {
                            float acc = vbias;
                            if (K::SYM) {
#pragma unroll
                                for (int k = 0; k < A; k++)
                                    acc = __builtin_fmaf(wv(ph, k), win[(i + k) % TAPS][e] + win[(i + TAPS - 1 - k) % TAPS][e], acc);
                            } else {
#pragma unroll
                                for (int j = 0; j < TAPS; j++) {
                                    const int k = f32_tap_order(j, TAPS);
                                    acc = __builtin_fmaf(wv(ph, k), win[(i + k) % TAPS][e], acc);
                                }
                            }
                            accs[e] = acc;
                        }
                        if (SB == 1) {
                            // floor(sum + eps): the sum is biased by eps - 0.5 and the hardware's saturating
                            // round-to-nearest-even byte convert does the rest (a tie needs fract(sum+eps) == 0,
                            // an undecided sample, which is within 1 LSB either way)
#pragma unroll
                            for (int e = 0; e < 4; e++) packed = __builtin_amdgcn_cvt_pk_u8_f32(accs[e], e, packed);
                            if (EXACT) {  // the H pass's test: fract(|acc|) - 0.5 in [0, 2 eps) = undecided
                                uint32_t gmin = 0x7f800000u;
#pragma unroll
                                for (int e = 0; e < 4; e++) {
                                    const uint32_t gu = __builtin_bit_cast(uint32_t, __builtin_amdgcn_fractf(__builtin_fabsf(accs[e])) - 0.5f);
                                    gmin = gu < gmin ? gu : gmin;
                                }
                                undecided = gmin < __builtin_bit_cast(uint32_t, near2);
                            }
                        } else if (SB == 2 && !EXACT) {
                            // floor(sum + eps) clamped to [0, 65535]: max with 0, truncating convert, saturating pack
                            const unsigned u0 = (unsigned)__builtin_fmaxf(accs[0], 0.0f), u1 = (unsigned)__builtin_fmaxf(accs[1], 0.0f);
                            packed = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_u16(u0, u1));
                        } else {
#pragma unroll
                            for (int e = 0; e < F::VEC; e++) {
                                const float xc = __builtin_amdgcn_fmed3f(accs[e], 0.5f, F::MAXV + 0.5f);
                                const float fl = __builtin_floorf(xc);
                                if (EXACT) undecided |= (xc - fl) < near2;
                                packed |= (unsigned)fl << (8 * SB * e);
                            }
                        }
                        }  // !(SPLIT && EXACT)
                    }
                    bool redo_row = false;
                    if (EXACT) {
                        // a row with an undecided sample is redone in f64 AFTER the row loop, from the ring (one copy of
                        // the f64 code, no doubles live in this loop); it is not stored here
                        redo_row = __any(undecided);  // wave-uniform
                        if (redo_row) redo_mask |= 1ull << ((mm + i) * S + ph);
                    }
                    if ((!CHECKED || (m >= m_b && m < m_e && y >= y_lo && y < y_hi)) && !no_store && !redo_row)  // uniform
                        __builtin_amdgcn_raw_buffer_store_b32(packed, orsrc, col_b, soff, LZ_STORE_AUX);
                    soff += g.out_pitch;
                }
            }
        }
        };
        if (interior) rows(std::integral_constant<bool, false>{});
        else rows(std::integral_constant<bool, true>{});
        if (EXACT) {
#pragma unroll 1
            while (redo_mask) {  // uniform: the exact vertical chain of full_TB.h:71-75 for one output row
                const int r = __builtin_ctzll(redo_mask);
                redo_mask &= redo_mask - 1;
                const int y = m_g * S + r, m = y / S;
                if (!(m >= m_b && m < m_e && y >= y_lo && y < y_hi) || no_store) continue;
                // (the phase weights out of LDS instead of the table row: measured, no faster -- profiles/round4n_*)
                const double* wvd = t.v_w + (size_t)y * TAPS;
                uint32_t rw[TAPS];
#pragma unroll
                for (int k = 0; k < TAPS; k++) rw[k] = ring(m - A + 1 + k);
                uint32_t packed = 0;
                if (r % S == 0 && fc.skip_last) {
                    // an integer-phase row (on noise: nearly every one of them holds a candidate): the H fix-up's short chain -- the
                    // centre weight is exactly 1 (the product is the sample itself), the ~1e-33 tap at x - i = -a is inert
                    // (FastConsts::skip_last), the weights are the same for every row (LDS) -- 4 multiplies and 5 adds instead of 6 + 6
                    const double* xw = (const double*)(smem + K::LDS_TIN + K::LDS_HBUF + K::LDS_WL);
#pragma unroll 1
                    for (int e = 0; e < F::VEC; e++) {
                        double sum = 0;
#pragma unroll
                        for (int k = 0; k < TAPS - 1; k++) {
                            const double v = (double)((rw[k] >> (8 * SB * e)) & F::SMASK);
                            sum += k == A - 1 ? v : v * xw[k];
                        }
                        packed |= (unsigned)store_convert<T>(sum) << (8 * SB * e);
                    }
                } else {
#pragma unroll 1
                    for (int e = 0; e < F::VEC; e++) {
                        double sum = 0;
#pragma unroll
                        for (int k = 0; k < TAPS; k++) sum += (double)((rw[k] >> (8 * SB * e)) & F::SMASK) * wvd[k];
                        packed |= (unsigned)store_convert<T>(sum) << (8 * SB * e);
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b32(packed, orsrc, col_b, (y - g.out_row0) * g.out_pitch, LZ_STORE_AUX);
            }
        }
    };

    // diagnostic build only: residency census -- when and where (XCC / SE / CU) this workgroup ran
    unsigned long long census_t0 = 0;
    unsigned census_hw = 0, census_xcc = 0;
    if (STAMP) {
        census_t0 = __builtin_amdgcn_s_memrealtime();
        census_hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
        census_xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (31 << 11));
    }
    if (tid < kFastMaxS * kMaxTaps)  // exact-chain phase weights (published by the prologue's first barrier)
        ((double*)(smem + K::LDS_TIN + K::LDS_HBUF + K::LDS_WL))[tid] = t.x_w[tid];
    // =================================================================== the march, segment by segment
    unsigned long long tsum[5] = {0, 0, 0, 0, 0};  // diagnostic build only (STAMP): where a wave's cycles go
    int ticks_total = 0;
#pragma clang loop unroll(disable)
    for (int seg = 0; seg < g.wg_segs; seg++) {
    if (!load_segment(seg)) continue;  // uniform
    ticks_total += ticks;
    set_load_consts();
    {   // both ticks' loads in flight at once (one memory round trip instead of two per chunk)
        u32x4 pre0[K::LOAD_IT];
        issue_loads_to(0, pre0);
        issue_loads(1);
        commit_from(0, pre0);
        commit_loads(1);
        if (K::EARLY) issue_loads(2);
    }
    __syncthreads();
    hpass(0);
    __syncthreads();
    auto stamp = [&]() -> unsigned long long {
        unsigned long long tt = 0;
        if (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt)::"memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        return tt;
    };
    for (int tick = 0; tick < ticks; tick++) {
        const unsigned long long t0 = stamp();
        if (!K::EARLY) issue_loads(tick + 2);  // lands in the buffer HPASS(tick) has finished with
        const unsigned long long t1 = stamp();
        // Wave priority by phase (measured, interleaved on one device: H=3/V=2/else=0 is 11-13 % faster than all
        // equal): the H pass feeds the ring every other wave's next V pass waits for, so it goes first.
        __builtin_amdgcn_s_setprio(K::PRIO_H);
        if (tick + 1 < ticks) hpass(tick + 1);
        __builtin_amdgcn_s_setprio(0);
        const unsigned long long t2 = stamp();
        // Register staging: commit BEFORE the V pass issues its stores (the staging registers must not live through the V pass:
        // committing behind it -- the rows get the H pass and the V pass to arrive, no wait -- spills 5 dwords in this loop at the
        // 72-VGPR budget: 207 -> 235 us, profiles/round3n_ab_commit_behind_v_pass.txt).
        // (LDS-DMA input tiles -- buffer_load ... lds, no staging registers, a counted vmcnt wait behind the V pass -- measured
        // 7-10 % slower on config 2 wherever the DMA is issued: profiles/round3a_ab_prefix_riding_and_ldsdma_placement.txt.)
        commit_loads(tick & 1);
        if (K::EARLY) issue_loads(tick + 3);  // committed after the NEXT tick's H pass (see MarchCfg::EARLY)
        const unsigned long long t3 = stamp();
        __builtin_amdgcn_s_setprio(K::PRIO_V);
        vpass(tick);
        __builtin_amdgcn_s_setprio(0);
        const unsigned long long t4 = stamp();
        if (!LZ_DBG(g, 128)) __syncthreads();  // (profiling bit 128: no barrier -- results are then wrong)
        const unsigned long long t5 = stamp();
        if (STAMP) {
            tsum[0] += t1 - t0; tsum[1] += t2 - t1; tsum[2] += t3 - t2; tsum[3] += t4 - t3; tsum[4] += t5 - t4;
        }
    }
    }  // segments
    if (STAMP && g.stamps && lane == 0) {
        unsigned long long* dst = g.stamps + ((size_t)blockIdx.x * K::NWAVES + wave) * 6;
        for (int i = 0; i < 5; i++) dst[i] = tsum[i];
        dst[5] = (unsigned long long)ticks_total;
        if (wave == 0) {  // second half of the buffer: one record per workgroup
            unsigned long long* c = g.stamps + (size_t)16384 * 8 * 3 + (size_t)blockIdx.x * 3;
            c[0] = census_t0;
            c[1] = __builtin_amdgcn_s_memrealtime();
            c[2] = ((unsigned long long)census_xcc << 32) | census_hw;
        }
    }
}

// The kernel proper, twice: with the 96-SGPR cap (7 waves per SIMD: what four 6-wave workgroups per CU need) and without it.
// Instances that cannot hold 7 waves per SIMD anyway (their VGPRs or their LDS decide) gain nothing from the cap and, where the
// row loop is long, pay for it with scalar registers spilled to VGPR lanes INSIDE the loop: the EXACT 8-bit RGB 3x instances
// carried ~500 v_readlane_b32 (config 3, EXACT: 547 -> 474 us; profiles/round4z_ab_config3_registers_and_residency.txt).
template <typename T, int C, int S, int A, bool EXACT, bool STAMP = false, bool RIDE = false>
__global__ __launch_bounds__((MarchCfg<T, C, S, A>::NT), (MarchCfg<T, C, S, A>::MIN_WAVES)) LZ_MARCH_SGPR_ATTR void k_march(FrameGeom g, TapTables t, FastConsts fc) {
    march_body<T, C, S, A, EXACT, STAMP, RIDE>(g, t, fc);
}
template <typename T, int C, int S, int A, bool EXACT, bool STAMP = false, bool RIDE = false>
__global__ __launch_bounds__((MarchCfg<T, C, S, A>::NT), (MarchCfg<T, C, S, A>::MIN_WAVES)) void k_march_ws(FrameGeom g, TapTables t, FastConsts fc) {
    march_body<T, C, S, A, EXACT, STAMP, RIDE>(g, t, fc);
}
// which of the two an instance runs.  Workgroups land on the four SIMDs of a CU unevenly (a 6-wave workgroup as 2+2+1+1 from
// a varying start), so a CU that is to hold W waves needs room for ceil(W / 4) + 1 on every SIMD; without the cap (106 SGPRs)
// a SIMD holds 6.  Instances whose LDS footprint keeps a CU at W <= 20 waves therefore lose nothing by running uncapped, and
// the cap costs them spilled scalars: config 3 (3 x 5 waves) 206.4 -> 195.7 us, config 5 (2 x 8 waves) 604.8 -> 599.3 us.
template <typename T, int C, int S, int A, bool EXACT>
constexpr bool march_wide_sgpr() {
#ifdef LZ_MARCH_NO_WIDE_SGPR   // A/B builds only
    return false;
#else
    using K = MarchCfg<T, C, S, A>;
    return (160 * 1024 / K::LDS_BYTES) * K::NWAVES <= 20;
#endif
}
template <typename T, int C, int S, int A, bool EXACT, bool STAMP = false, bool RIDE = false>
inline const void* march_kernel_fn() {
    if constexpr (march_wide_sgpr<T, C, S, A, EXACT>()) return (const void*)k_march_ws<T, C, S, A, EXACT, STAMP, RIDE>;
    else return (const void*)k_march<T, C, S, A, EXACT, STAMP, RIDE>;
}
template <typename T, int C, int S, int A, bool EXACT, bool STAMP = false, bool RIDE = false>
inline void march_kernel_launch(dim3 grid, dim3 block, size_t lds, hipStream_t stream, const FrameGeom& g, const TapTables& t, const FastConsts& fc) {
    if constexpr (march_wide_sgpr<T, C, S, A, EXACT>()) hipLaunchKernelGGL((k_march_ws<T, C, S, A, EXACT, STAMP, RIDE>), grid, block, lds, stream, g, t, fc);
    else hipLaunchKernelGGL((k_march<T, C, S, A, EXACT, STAMP, RIDE>), grid, block, lds, stream, g, t, fc);
}

// the marching kernel moves whole 16-byte chunks: rows, frames and the base must be 16-byte multiples
inline bool march_supports(const FrameGeom& g) {
    return g.in_pitch % 16 == 0 && (((uintptr_t)g.in) & 15) == 0 && (g.in_frame_stride & 15) == 0 &&
           (size_t)g.in_pitch * g.in_rows < (1ull << 30);  // (the hoisted load offsets' out-of-range markers need the two top bits)
}

// Chunk height: ONE resident round of workgroups.  With more workgroups than the chip holds at once the second
// round runs part-empty and the launch takes two wave lifetimes; so the (strip, frame) pairs are cut into
// floor(slots / pairs) chunks each (at least one, a whole number of ticks, at least two ticks).
// A chunk of R rows marches over R + 2a - 1 H rows (its window reaches a - 1 rows above and a rows below), i.e.
// ceil((R + 2a - 1) / MS) ticks: R is rounded up to "a whole number of ticks minus the window" so that no tick is spent on
// window rows alone (16 x 1080p: 271-row chunks in 23 ticks where 276-row chunks took 24).
inline int march_chunk_rows(int m_rows, int strips, int frames, int ms, int slots, int taps) {
    const int target_env = env().march_wgs;
    const int pairs = strips * frames;
    auto ticks_of = [&](int chunks) { return ((m_rows + chunks - 1) / chunks + taps - 1 + ms - 1) / ms; };
    int chunks = 1;
    if (target_env > 0) {
        chunks = target_env / pairs;
        if (chunks < 1) chunks = 1;
    } else {
        // rounds of resident workgroups x (ticks of a chunk + its prologue, about two ticks): one full round where the batch
        // allows it; when floor(slots / pairs) chunks would leave the machine part-empty (pairs between slots / 2 and slots:
        // 32 frames of 720p, 320 pairs on 512 slots ran on 62 % of the chip), more and shorter chunks in two or three rounds
        long best = -1;
        for (int c = 1; c <= 512; c++) {
            if (c > 1 && ticks_of(c) < 3) break;
            const long rounds = ((long)pairs * c + slots - 1) / slots;
            const long cost = rounds * (ticks_of(c) + 2);
            if (best < 0 || cost < best) best = cost, chunks = c;
        }
    }
    int ticks = ticks_of(chunks);
    if (ticks < 3) ticks = 3;
    return ticks * ms - (taps - 1);
}

// ---------------------------------------------------------------------------------------------------------------------
// The workgroup table (see the kernel: "Which strip, which frame, which rows").
//
// Hardware facts it encodes (MI355X, profiles/round2e_census_32frames.txt; wrong facts only cost speed, never results):
//   block id b runs on XCD b % 8; inside an XCD the j-th block (j = b / 8) is the (j / CU_X)-th workgroup of CU j % CU_X
//   (CU_X = CUs per XCD = 32): the dispatcher fills the CUs breadth first.  A CU's earlier workgroups run faster (older
//   waves win the SIMD arbitration), and a CU that received one workgroup fewer runs each of them faster still.
// speed[] below = rows per unit time of such a slot relative to the mean, from measured lifetimes with equal shares;
// LANCZOS_RANK_WEIGHTS="w0:w1:w2:w3/v0:v1:v2" overrides (full CUs; CUs one short), LANCZOS_RANK_WEIGHTS=0 = equal shares.
struct MarchSlotSpeed {
    double full[8], shorty[8];
};
inline MarchSlotSpeed march_slot_speed(int nb, int nwaves) {
    MarchSlotSpeed w;
    for (int i = 0; i < 8; i++) w.full[i] = w.shorty[i] = 1.0;
    if (nb == 4) {
        const double f[4] = {1.060, 1.008, 0.964, 0.918}, s3[3] = {1.246, 1.156, 1.064};
        for (int i = 0; i < 4; i++) w.full[i] = f[i];
        for (int i = 0; i < 3; i++) w.shorty[i] = s3[i];
    } else if (nb == 2 && nwaves % 4 != 0) {
        // 6-wave workgroups, two per CU (config 3: 123 -> 117 us at 16 frames); 8-wave workgroups put two waves on every SIMD
        // and showed no rank effect (config 5, 4 frames: 315 us with equal shares, 320 with these)
        w.full[0] = 1.04, w.full[1] = 0.96, w.shorty[0] = 1.20;
    } else if (nb == 3) {
        w.full[0] = 1.05, w.full[1] = 1.0, w.full[2] = 0.95, w.shorty[0] = 1.18, w.shorty[1] = 1.10;
    }
    if (env().has_rank_weights) {
        const char* e = env().rank_weights.c_str();
        if (atof(e) == 0.0 && e[0] == '0') {
            for (int i = 0; i < 8; i++) w.full[i] = w.shorty[i] = 1.0;
        } else {
            const char* p = e;
            double* dst = w.full;
            int i = 0;
            while (*p) {
                char* end = nullptr;
                const double v = strtod(p, &end);
                if (end == p) break;
                if (i < 8) dst[i++] = v;
                p = end;
                if (*p == ',' || *p == ':') p++;
                else if (*p == ';' || *p == '/') p++, dst = w.shorty, i = 0;
            }
        }
    }
    return w;
}

// rows of a chunk that end on a tick boundary: ticks * MS - (2a - 1)
inline int march_align_rows(int rows, int ms, int taps) {
    int t = (rows + taps - 1 + ms / 2) / ms;  // nearest
    if (t < 2) t = 2;
    return t * ms - (taps - 1);
}

// Fills `tab` ([block id][segs]) for `strips` x `frames` pairs whose rows [m_lo, m_hi) are shared out among the marching
// workgroups.  Returns the number of marching workgroups; *segs_out = table entries per workgroup.
//   mode A (large batches): exactly one workgroup per CU slot.  The pairs are dealt to the XCDs in consecutive runs (neighbouring
//     strips share an L2); inside an XCD the pairs' rows form one line that is cut into consecutive shares proportional to
//     the slots' speeds.  A share may run from one pair into the next: up to 3 segments, cuts moved off pair boundaries'
//     neighbourhood and onto tick boundaries.
//   mode B: whole chunks (one segment), equal, or -- one resident round of >= 2 chunks per pair -- fast slots paired with
//     slow ones and the pair's rows split by speed.
inline int march_build_table(std::vector<WgEntry>& tab, int* segs_out, int strips, int frames, int m_lo, int m_hi, int ms, int taps,
                             int nb, int cus, int nwaves, bool* balanced_out) {
    const int m_rows = m_hi - m_lo, slots = nb * cus, pairs = strips * frames;
    const int nx = 8, cu_x = cus / nx > 0 ? cus / nx : 1;
    const MarchSlotSpeed sp = march_slot_speed(nb, nwaves);
    const int segs_env = env().march_segs;  // 0: never mode A
    const int ticks_pair = (m_rows + taps - 1 + ms - 1) / ms;
    // what mode B would do with this shape
    const int rows_u = march_chunk_rows(m_rows, strips, frames, ms, slots, taps);
    const int chunks = (m_rows + rows_u - 1) / rows_u;
    const int n_b = pairs * chunks;
    bool b_balanced = n_b <= slots && chunks >= 2 && cus % nx == 0;
    for (int x = 0; x < nx && b_balanced; x++)
        if (((n_b - x + nx - 1) / nx) % chunks != 0) b_balanced = false;
    if (b_balanced) {
        bool any = false;
        for (int i = 0; i < 8; i++) any = any || sp.full[i] != 1.0 || sp.shorty[i] != 1.0;
        b_balanced = any;
    }
    // Mode B keeps the strips of a frame on the same rows at the same time (their 384-byte input segments are one
    // 5760-byte image row: the DRAM sees whole rows) and wins where it fills the machine in one balanced round (config 2,
    // 32 frames: 217 us against 232 for mode A); mode A wins where B would need two rounds or leave slots empty
    // (config 3, 32 frames: 245 -> 235 us; config 5, 8 frames: 633 -> 621).
    const bool b_good = b_balanced && n_b * 10 >= slots * 9;
    // ---------------------------------------------------------------- mode A
    if (segs_env != 0 && (!b_good || segs_env > 0) && cus % nx == 0 && pairs >= 8 * nx && (pairs % nx == 0 || pairs >= 32 * nx) &&
        (long)pairs * ticks_pair >= (long)slots * 10 && pairs <= 2 * slots - 2 * nx) {
        const int n = slots, n_x = slots / nx, min_seg = 3 * ms - (taps - 1);
        std::vector<std::vector<WgEntry>> share(n);
        int pair0 = 0, max_segs = 1;
        for (int x = 0; x < nx; x++) {
            const int np = pairs / nx + (x < pairs % nx ? 1 : 0);
            const long R = (long)np * m_rows;
            double tot = 0;
            for (int j = 0; j < n_x; j++) tot += (j / cu_x < 8) ? sp.full[j / cu_x] : 1.0;
            long prev = 0;  // start of the current share on the XCD's line of rows
            double acc = 0;
            for (int j = 0; j < n_x; j++) {
                acc += (j / cu_x < 8) ? sp.full[j / cu_x] : 1.0;
                long cut = j + 1 == n_x ? R : (long)(R * (acc / tot) + 0.5);
                if (cut < prev) cut = prev;
                if (j + 1 < n_x) {
                    const long q = cut / m_rows;
                    long r = cut - q * m_rows;
                    if (r < min_seg) r = 0;                                  // no sliver at the top of a pair ...
                    else if (m_rows - r < min_seg) r = m_rows;               // ... nor at its bottom
                    else {
                        // the segment that ends here starts at the share's start or at the pair's top: make it whole ticks
                        const long seg0 = prev > q * m_rows ? prev - q * m_rows : 0;
                        long len = r - seg0;
                        long t = (len + taps - 1 + ms / 2) / ms;
                        if (t < 3) t = 3;
                        r = seg0 + t * ms - (taps - 1);
                        if (m_rows - r < min_seg) r = m_rows;
                    }
                    cut = q * m_rows + r;
                    if (cut < prev) cut = prev;
                    if (cut > R) cut = R;
                }
                // [prev, cut) -> segments
                std::vector<WgEntry>& sh = share[(size_t)j * nx + x];
                for (long pos = prev; pos < cut;) {
                    const long q = pos / m_rows, r0 = pos - q * m_rows;
                    const long r1 = (cut - q * m_rows) < m_rows ? (cut - q * m_rows) : m_rows;
                    const int pair = pair0 + (int)q;
                    sh.push_back(WgEntry{pair / strips, pair % strips, m_lo + (int)r0, m_lo + (int)r1});
                    pos = q * m_rows + r1;
                }
                if ((int)sh.size() > max_segs) max_segs = (int)sh.size();
                prev = cut;
            }
            pair0 += np;
        }
        tab.assign((size_t)n * max_segs, WgEntry{0, 0, 0, 0});
        for (int b = 0; b < n; b++)
            for (size_t k = 0; k < share[b].size(); k++) tab[(size_t)b * max_segs + k] = share[b][k];
        *segs_out = max_segs;
        *balanced_out = true;
        return n;
    }
    // ---------------------------------------------------------------- mode B
    *segs_out = 1;
    const int n = n_b;
    tab.assign(n, WgEntry{0, 0, 0, 0});
    // pairs (frame-major: neighbouring strips consecutive) are dealt to the XCDs in consecutive runs; XCD x owns the
    // hardware ids x, x + 8, ...: (n - x + 7) / 8 of them
    const bool balanced = b_balanced;
    *balanced_out = balanced;
    int pair0 = 0, lid0 = 0;
    for (int x = 0; x < nx; x++) {
        const int n_x = (n - x + nx - 1) / nx;
        if (!balanced) {
            // equal shares: the ids of XCD x are renumbered consecutively behind XCD x-1's; chunk-major inside a frame
            for (int j = 0; j < n_x; j++) {
                const int lid = lid0 + j, frame = lid / (strips * chunks), q = lid - frame * strips * chunks;
                const int tx = q % strips, c = q / strips;
                const int b = m_lo + c * rows_u, e = b + rows_u < m_hi ? b + rows_u : m_hi;
                tab[(size_t)j * nx + x] = WgEntry{frame, tx, b, e};
            }
            lid0 += n_x;
            continue;
        }
        // slot speeds of this XCD's workgroups
        std::vector<std::pair<double, int>> sj(n_x);
        for (int j = 0; j < n_x; j++) {
            const int rank = j / cu_x, c = j % cu_x;
            const int on_cu = (n_x - c + cu_x - 1) / cu_x;
            double v = 1.0;
            if (rank < 8) v = on_cu >= nb ? sp.full[rank] : (on_cu == nb - 1 ? sp.shorty[rank] : 1.0);
            sj[j] = {v, j};
        }
        const int groups = n_x / chunks;
        // (Band-synchronous shares -- all strips of a (frame, chunk) band on the same rows -- were measured and dropped:
        // profiles/round2g_ab_band_sync.txt.)
        std::sort(sj.begin(), sj.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) {
            return a.first != b.first ? a.first > b.first : a.second < b.second;
        });
        // groups of `chunks` slots with near-equal speed sums: deal the sorted slots out in snake order
        std::vector<std::vector<std::pair<double, int>>> grp(groups);
        for (int i = 0; i < n_x; i++) {
            const int round = i / groups, pos = i % groups;
            grp[(round & 1) ? groups - 1 - pos : pos].push_back(sj[i]);
        }
        for (int gi = 0; gi < groups; gi++) {
            const int pair = pair0 + gi, frame = pair / strips, tx = pair % strips;
            auto& mem = grp[gi];
            std::sort(mem.begin(), mem.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.second < b.second; });
            double tot = 0;
            for (auto& m : mem) tot += m.first;
            int b = m_lo;
            double acc = 0;
            for (size_t k = 0; k < mem.size(); k++) {
                acc += mem[k].first;
                int e;
                if (k + 1 == mem.size()) e = m_hi;
                else {
                    const int want = (int)(m_lo + m_rows * acc / tot + 0.5) - b;
                    e = b + march_align_rows(want, ms, taps);
                    if (e > m_hi) e = m_hi;
                }
                tab[(size_t)mem[k].second * nx + x] = WgEntry{frame, tx, b, e};
                b = e;
            }
        }
        pair0 += groups;
    }
    return n;
}

// retire lists and the table cache: lanczos_cache.hpp (host-only, tested without a GPU)
using WgTabCache = WgTabCacheT<WgEntry>;

template <typename T, int C, int S, int A>
inline hipError_t march_launch_t(const lanczos_desc& d, const FrameGeom& g_in, const TapTables& t,
                                 const FastConsts& fc, hipStream_t stream, bool* prefix_fused, bool query_only, WgTabCache* cache) {
    using K = MarchCfg<T, C, S, A>;
    using F = typename K::F;
    FrameGeom g = g_in;
    // g.prefix_K > 0 asks for the in-place prefix rows to ride on this launch (extra workgroups behind the marching
    // ones); that needs a main launch at all and the row arrays to fit the workgroup's LDS -- otherwise the caller
    // launches k_prefix
    if (g.prefix_K > 0 &&
        (env().separate_prefix || (size_t)(g.prefix_M + g.prefix_M2) * K::NT * sizeof(T) > (size_t)K::LDS_BYTES ||
         g.out_row0 + g.out_rows <= (g.out_row0 > g.skip_rows ? g.out_row0 : g.skip_rows)))
        g.prefix_K = g.prefix_M = g.prefix_M2 = 0;
    *prefix_fused = g.prefix_K > 0;
    const int strips = (g.out_w + F::TWP_OUT - 1) / F::TWP_OUT;
    const int y_lo = g.out_row0 > g.skip_rows ? g.out_row0 : g.skip_rows;
    const int y_hi = g.out_row0 + g.out_rows;
    const bool exact = d.mode == LANCZOS_MODE_EXACT;
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    int nb = 1, cus = 256;
    {   // process-wide facts about this kernel instance on this device.  The lock covers only these arrays: the table cache below
        // belongs to the context, whose calls are serialised by its own mutex.
        std::lock_guard<std::mutex> cache_lock(launch_cache_mutex());
        static int slots[2][64] = {};
        static int cus_of[64] = {};
        static bool attr_done[2][2][64] = {};  // [exact][riding][device]: dynamic LDS size set
        if (cus_of[dev] == 0 && (hipDeviceGetAttribute(&cus_of[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus_of[dev] < 1))
            cus_of[dev] = 256;
        if (slots[exact][dev] == 0) {
            int n = 0;
            hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, exact ? march_kernel_fn<T, C, S, A, true>() : march_kernel_fn<T, C, S, A, false>(),
                                                                        K::NT, K::LDS_BYTES);
            if (e != hipSuccess || n < 1) n = 1;
            {   // The API counts waves as if they spread evenly over the four SIMDs.  They do not: a workgroup whose wave count is
                // not a multiple of 4 lands unevenly from a varying start (6 waves: 2+2+1+1), and n workgroups only fit reliably
                // when every SIMD has room for ceil(n * waves / 4) + 1.  Where the instance's registers do not leave that room the
                // last workgroup per CU does not become resident, the table would be laid out for slots that do not exist and the
                // launch would run a part-empty second round (the EXACT 8-bit RGB 3x instance: 85 VGPRs = 5 waves per SIMD, three
                // 6-wave workgroups need 6: 472 us per 32 frames of config 3, 334 once it was told to stay at 80 VGPRs).
                hipFuncAttributes fa;
                const void* fn = exact ? march_kernel_fn<T, C, S, A, true>() : march_kernel_fn<T, C, S, A, false>();
                if (hipFuncGetAttributes(&fa, fn) == hipSuccess && fa.numRegs > 0 && fa.numRegs <= 512) {
                    const int cap_v = 512 / ((fa.numRegs + 7) & ~7) < 8 ? 512 / ((fa.numRegs + 7) & ~7) : 8;
                    const bool wide = exact ? march_wide_sgpr<T, C, S, A, true>() : march_wide_sgpr<T, C, S, A, false>();
                    const int cap_s = wide ? 6 : 7;   // 106 / 96 SGPRs (MI355X_MICROARCH.md, "Residency")
                    const int cap = cap_v < cap_s ? cap_v : cap_s;
                    auto need = [](int nn) { return K::NWAVES % 4 == 0 ? nn * K::NWAVES / 4 : (nn * K::NWAVES + 3) / 4 + 1; };
                    const int n_api = n;
                    while (n > 1 && need(n) > cap) n--;
                    if (env().verbose)
                        fprintf(stderr, "lanczos: k_march<%d B,%d ch,x%d,a=%d,%s> %d VGPRs: %d waves per SIMD, %d workgroups/CU by the API, %d resident\n",
                                (int)sizeof(T), C, S, A, exact ? "exact" : "lsb1", fa.numRegs, cap, n_api, n);
                } else {
                    (void)hipGetLastError();
                }
            }
            // (With the 96-SGPR cap the API's answer holds.  At the compiler's own ~105 SGPRs a CU admitted one 6-wave workgroup
            // fewer than the API said -- waves land unevenly on the SIMDs and the SGPR file caps waves per SIMD,
            // MI355X_MICROARCH.md "Residency" -- and the grid's second round ran part-empty.)
            slots[exact][dev] = n * cus_of[dev];
            if (env().verbose)
                fprintf(stderr, "lanczos: k_march<%d B,%d ch,x%d,a=%d> %d threads, %d B LDS: %d workgroups/CU x %d CUs\n",
                        (int)sizeof(T), C, S, A, K::NT, K::LDS_BYTES, n, cus_of[dev]);
        }
        for (int ride = 0; ride < 2; ride++) {
            if (attr_done[exact][ride][dev]) continue;
            const void* fn = exact ? (ride ? march_kernel_fn<T, C, S, A, true, false, true>() : march_kernel_fn<T, C, S, A, true>())
                                   : (ride ? march_kernel_fn<T, C, S, A, false, false, true>() : march_kernel_fn<T, C, S, A, false>());
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
            if (e != hipSuccess) return e;
            attr_done[exact][ride][dev] = true;
        }
        cus = cus_of[dev];
        nb = slots[exact][dev] / cus > 0 ? slots[exact][dev] / cus : 1;
    }
    {
        int pf = (nb * cus) / (2 * strips);
        cache->max_one_round_frames = pf;
        cache->wg_per_cu = nb;
        while (pf > 0 && (2 * strips * pf) % 16 != 0) pf--;   // every XCD gets whole pairs of chunks (march_build_table: balanced)
        cache->pref_frames = pf;
    }
    // Measured (config 2, ms per step riding / separate): 1 frame 0.0198 / 0.0253, 2: 0.0281 / 0.0335, 4: 0.0464 / 0.0525,
    // 8: 0.0687 / 0.0729, 16: 0.117-0.120 / 0.114-0.116, 32: 0.232 / 0.218 -- past about one prefix workgroup per CU they slow
    // the march down more than the launch they replace (profiles/round3a_ab_prefix_riding_and_ldsdma_placement.txt).
    g.prefix_blocks_per_frame = g.prefix_K > 0 ? (g.out_w * C + K::NT - 1) / K::NT : 0;
    if (g.prefix_blocks_per_frame * g.frames > cus) {
        g.prefix_K = g.prefix_M = g.prefix_M2 = g.prefix_blocks_per_frame = 0;
        *prefix_fused = false;
    }
    // a pure query (the caller wants *prefix_fused and pref_frames) touches neither the table cache nor the stream
    if (query_only || y_lo >= y_hi) return hipSuccess;
    // the workgroup table of this launch shape (built once per context and shape)
    const int m_lo = y_lo / S, m_hi = (y_hi - 1) / S + 1;
    const long long key[8] = {(long long)sizeof(T) | ((long long)C << 8) | ((long long)S << 16) | ((long long)A << 24) | ((long long)exact << 32),
                              strips, g.frames, m_lo, m_hi, nb, cus, dev};
    WgTabCache::Item* item = cache->find(key);
    if (!item) {
        std::vector<WgEntry> tab;
        int segs = 1;
        bool balanced = false;
        const int n = march_build_table(tab, &segs, strips, g.frames, m_lo, m_hi, K::MS, K::TAPS, nb, cus, K::NWAVES, &balanced);
        hipError_t e = hipSuccess;
        item = cache->insert(key, tab, n, segs, balanced, stream, &e);
        if (!item) return e;
        if (env().verbose)
            fprintf(stderr, "lanczos: k_march table: %d workgroups x %d segment(s) for %d strips x %d frames, rows [%d, %d): %s shares\n", item->n,
                    item->segs, strips, g.frames, m_lo, m_hi, item->balanced ? "rank-aware" : "equal");
    }
    if (stream != item->upload_stream && hipEventQuery(item->uploaded) != hipSuccess) {
        hipError_t e = hipStreamWaitEvent(stream, item->uploaded, 0);  // another stream: behind the upload
        if (e != hipSuccess) return e;
    }
    g.wg_tab = item->dev;
    g.wg_segs = item->segs;
    g.wg_per_frame = 0;
    g.n_main = item->n;
    note_stream(item->streams, stream);
    dim3 grid(g.n_main + g.prefix_blocks_per_frame * g.frames);
    if (*prefix_fused) {
        if (exact) march_kernel_launch<T, C, S, A, true, false, true>(grid, dim3(K::NT), K::LDS_BYTES, stream, g, t, fc);
        else march_kernel_launch<T, C, S, A, false, false, true>(grid, dim3(K::NT), K::LDS_BYTES, stream, g, t, fc);
        return hipGetLastError();
    }
#ifdef LZ_PROFILE_BITS
    if (!exact && g.stamps && sizeof(T) == 1 && C == 3 && S == 2 && A == 3) {  // diagnostic build, one configuration
        using KS = MarchCfg<uint8_t, 3, 2, 3>;
        hipLaunchKernelGGL((k_march<uint8_t, 3, 2, 3, false, true>), grid, dim3(KS::NT), KS::LDS_BYTES, stream, g, t, fc);
        return hipGetLastError();
    }
#endif
    if (exact) march_kernel_launch<T, C, S, A, true>(grid, dim3(K::NT), K::LDS_BYTES, stream, g, t, fc);
    else march_kernel_launch<T, C, S, A, false>(grid, dim3(K::NT), K::LDS_BYTES, stream, g, t, fc);
    return hipGetLastError();
}

#define LZ_DECLARE_MARCH_GROUP(G)                                                                                                  \
    hipError_t march_launch_g##G(const lanczos_desc& d, const FrameGeom& g, const TapTables& t, const FastConsts& fc, hipStream_t stream, \
                                 bool* prefix_fused, WgTabCache* cache, bool query_only);
LZ_DECLARE_MARCH_GROUP(0)
LZ_DECLARE_MARCH_GROUP(1)
LZ_DECLARE_MARCH_GROUP(2)
LZ_DECLARE_MARCH_GROUP(3)
#undef LZ_DECLARE_MARCH_GROUP

inline hipError_t march_launch(const lanczos_desc& d, const FrameGeom& g, const TapTables& t, const FastConsts& fc,
                               hipStream_t stream, bool* prefix_fused, WgTabCache* cache, bool query_only = false) {
    *prefix_fused = false;
    if (d.bytes_per_sample == 2) return march_launch_g3(d, g, t, fc, stream, prefix_fused, cache, query_only);
    if (d.scale_n == 2) return march_launch_g0(d, g, t, fc, stream, prefix_fused, cache, query_only);
    if (d.scale_n == 3) return march_launch_g1(d, g, t, fc, stream, prefix_fused, cache, query_only);
    if (d.scale_n == 4) return march_launch_g2(d, g, t, fc, stream, prefix_fused, cache, query_only);
    return hipErrorNotSupported;
}

}  // namespace lz
