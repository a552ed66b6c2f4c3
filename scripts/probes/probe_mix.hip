// probe: mixed-precision FMA on 16-bit integer lanes (gfx950).  Question (round 2, second half): the byte -> float
// conversions (v_cvt_f32_ubyteN, slow issue class, 71 per wave and tick in k_march) could disappear if the chains ran on
// 16-bit lanes:  a byte pair (b0, b2) = dword & 0x00ff00ff is two u16 integers; a u16 integer n IS the f16 denormal n * 2^-24;
//     p   = v_pk_add_u16(A, B)                         -- two exact pair sums per instruction (<= 510)
//     acc = v_fma_mix_f32(w * 2^24, p.lo|hi (f16), acc) -- f32 FMA whose f16 source is widened exactly
// needs (1) f16 denormal sources honoured by v_fma_mix_f32, (2) both instructions in the fast issue class.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
template <int MODE>
__global__ void k(float* out, int iters, float sa, unsigned seed) {
    float x[12];
    unsigned u[12];
    float va = sa + threadIdx.x * 1e-9f;
    unsigned mask = 0x00ff00ffu + (seed >> 31);
#pragma unroll
    for (int i = 0; i < 12; i++) { x[i] = threadIdx.x + i; u[i] = ((threadIdx.x * 2654435761u + i * 40503u) ^ seed) & 0x01ff01ffu; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < 12; i++) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(va), "v"(x[(i + 1) % 12]));
                if (MODE == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "+v"(x[i]) : "v"(va), "v"(0x3c003c00u + u[i]));
                if (MODE == 2) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(x[i]) : "v"(va), "v"(u[i]));
                if (MODE == 3) asm volatile("v_pk_add_u16 %0, %1, %2" : "=v"(u[i]) : "v"(u[(i + 1) % 12]), "v"(mask));
                if (MODE == 4) asm volatile("v_pk_add_u16 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(u[i]) : "v"(u[(i + 1) % 12]), "v"(mask));
                if (MODE == 5) asm volatile("v_and_b32 %0, %1, %2" : "=v"(u[i]) : "v"(u[(i + 1) % 12]), "v"(mask));
                if (MODE == 6) asm volatile("v_lshrrev_b32 %0, 8, %1" : "=v"(u[i]) : "v"(u[(i + 1) % 12]));
                if (MODE == 7) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(x[i]) : "v"(u[i]));
                if (MODE == 8) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(x[i]) : "v"(0x3c003c00u + u[i]), "v"(0x3c003c00u));
                if (MODE == 9) asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(u[i]) : "v"(u[(i + 1) % 12]), "v"(mask));
                if (MODE == 10) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u[i]) : "v"(x[i]));
                if (MODE == 11) asm volatile("v_sat_pk_u8_i16 %0, %1" : "=v"(u[i]) : "v"(u[(i + 1) % 12]));
                if (MODE == 12) asm volatile("v_and_b32 %0, 0x00ff00ff, %1" : "=v"(u[i]) : "v"(u[(i + 1) % 12]));
                if (MODE == 13) asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(u[i]) : "v"(u[(i + 1) % 12]));
                if (MODE == 14) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(x[i]) : "v"(u[i]));
                if (MODE == 15) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(u[i]) : "v"(u[(i + 1) % 12]), "v"(mask));
                if (MODE == 16) asm volatile("v_and_b32 %0, %1, %2" : "=v"(u[i]) : "v"(u[(i + 1) % 12]), "s"(seed));
                if (MODE == 17) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,1,1]" : "=v"(x[i]) : "v"(va), "v"(u[i]), "v"(u[(i + 1) % 12]));
                if (MODE == 18) asm volatile("v_lshlrev_b32 %0, 8, %1" : "=v"(u[i]) : "v"(u[(i + 1) % 12]));
                if (MODE == 19) asm volatile("v_pk_lshrrev_b16 %0, 8, %1" : "=v"(u[i]) : "v"(u[(i + 1) % 12]));
                if (MODE == 20) asm volatile("v_max_f32 %0, %1, %0" : "+v"(x[i]) : "v"(va));
                if (MODE == 21) asm volatile("v_pk_max_u16 %0, %1, %2" : "=v"(u[i]) : "v"(u[(i + 1) % 12]), "v"(mask));
                if (MODE == 22) asm volatile("v_pk_sub_u16 %0, %1, %2" : "=v"(u[i]) : "v"(u[(i + 1) % 12]), "v"(mask));
                if (MODE == 23) asm volatile("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(u[i]) : "v"(u[(i + 1) % 12]), "v"(mask));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) s += x[i] + u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// numerics: sum of 3 pair sums of bytes times weights through the 16-bit-lane route vs the plain float route, lo and hi lanes
__global__ void knum(const unsigned* a, const unsigned* b, const float* w, float* plain, float* mix, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned da = a[i], db = b[i];
    const float s24 = 16777216.0f;
    // three "rows" per side made from shifted copies of the two dwords
    unsigned ra[3] = {da, (da >> 8) | (db << 24), (da >> 16) | (db << 16)};
    unsigned rb[3] = {db, (db >> 8) | (da << 24), (db >> 16) | (da << 16)};
    for (int lane = 0; lane < 2; lane++) {
        float acc_p = 0.25f, acc_m = 0.25f;
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
            const float fa = (float)((ra[kk] >> (16 * lane)) & 255), fb = (float)((rb[kk] >> (16 * lane)) & 255);
            acc_p = __builtin_fmaf(w[kk], fa + fb, acc_p);
        }
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
            unsigned ea = ra[kk] & 0x00ff00ffu, eb = rb[kk] & 0x00ff00ffu, p;
            asm volatile("v_pk_add_u16 %0, %1, %2" : "=v"(p) : "v"(ea), "v"(eb));
            const float ws = w[kk] * s24;
            if (lane == 0) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "+v"(acc_m) : "v"(ws), "v"(p));
            else asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(acc_m) : "v"(ws), "v"(p));
        }
        plain[2 * i + lane] = acc_p;
        mix[2 * i + lane] = acc_m;
    }
}
int main() {
    float* d;
    hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* names[] = {"v_fma_f32 v,v,v", "v_fma_mix_f32 f16.lo normal", "v_fma_mix_f32 f16.hi denormal", "v_pk_add_u16", "v_pk_add_u16 op_sel",
                           "v_and_b32 v,v", "v_lshrrev_b32 8", "v_cvt_f32_ubyte1", "v_dot2_f32_f16", "v_pk_add_f16 (denormal)", "v_cvt_pk_u8_f32",
                           "v_sat_pk_u8_i16", "v_and_b32 literal", "v_bfe_u32", "v_cvt_f32_f16", "v_pk_fma_f16", "v_and_b32 sgpr",
                           "v_fma_mix_f32 two f16 srcs", "v_lshlrev_b32 8", "v_pk_lshrrev_b16 8", "v_max_f32 v,v", "v_pk_max_u16", "v_pk_sub_u16", "v_pk_mul_lo_u16"};
    const int iters = 500;
    for (int wps : {8, 4, 2, 1}) {
        const int blocks = 256 * wps;
        printf("--- %d waves per SIMD\n", wps);
#define RUN(M)                                                                                                  \
    {                                                                                                           \
        float best = 1e9;                                                                                       \
        for (int rep = 0; rep < 3; rep++) {                                                                     \
            hipEventRecord(e0);                                                                                 \
            hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 12345u);                 \
            hipEventRecord(e1);                                                                                 \
            hipEventSynchronize(e1);                                                                            \
            float ms;                                                                                           \
            hipEventElapsedTime(&ms, e0, e1);                                                                   \
            if (ms < best) best = ms;                                                                           \
        }                                                                                                       \
        double winst = (double)blocks * 4 * iters * 48;                                                         \
        printf("%-34s %8.1f us  %.3f ns per wave-instr per SIMD\n", names[M], best * 1e3, best * 1e6 / (winst / 1024)); \
    }
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17)
        RUN(18) RUN(19) RUN(20) RUN(21) RUN(22) RUN(23)
    }
    // numerics
    const int n = 1 << 20;
    unsigned *ha = new unsigned[n], *hb = new unsigned[n], *da, *db;
    float hw[3] = {0.02431708f, -0.13508514f, 0.60792710f}, *dw, *dp, *dd;
    unsigned s = 777;
    for (int i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; ha[i] = s; s = s * 1664525u + 1013904223u; hb[i] = s; }
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dw, 12); hipMalloc(&dp, n * 8); hipMalloc(&dd, n * 8);
    hipMemcpy(da, ha, n * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb, n * 4, hipMemcpyHostToDevice); hipMemcpy(dw, hw, 12, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(knum, dim3(n / 256), dim3(256), 0, 0, da, db, dw, dp, dd, n);
    float *hp = new float[2 * n], *hd = new float[2 * n];
    hipMemcpy(hp, dp, n * 8, hipMemcpyDeviceToHost); hipMemcpy(hd, dd, n * 8, hipMemcpyDeviceToHost);
    int same = 0; double maxd = 0;
    for (int i = 0; i < 2 * n; i++) { if (hp[i] == hd[i]) same++; double dlt = fabs((double)hp[i] - hd[i]); if (dlt > maxd) maxd = dlt; }
    printf("numerics: 16-bit-lane route == plain route bit for bit in %d of %d cases, max |diff| %.3g (sample: plain %.6f mix %.6f ; %.6f %.6f)\n", same, 2 * n, maxd, hp[5], hd[5], hp[6], hd[6]);
    return 0;
}
