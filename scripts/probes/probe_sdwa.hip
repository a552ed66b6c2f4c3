// probe: SDWA byte selects on f32 VALU ops (gfx950).  Question (round 2): can the byte -> float conversions of the H and V
// passes (71 v_cvt_f32_ubyteN per wave and tick, slow issue class) be dropped by feeding the packed bytes to the arithmetic
// directly?  A byte selected by SDWA (zero-extended) IS the f32 denormal b * 2^-149, so
//     p = v_add_f32_sdwa(dwordA BYTE_i, dwordB BYTE_j)      -- exact pair sum, still in the 2^-149 domain
//     acc = v_fmac_f32(w * 2^126, p, acc)                    -- an ordinary product, a normal float scaled by 2^-23
// would need no conversion at all, provided (1) denormal inputs are honoured (not flushed) and (2) the SDWA forms issue
// in the fast class.  This probe measures (2) and checks (1).  (v_fmac_f32 has no SDWA form on gfx950: llvm-mc refuses it.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
template <int MODE>
__global__ void k(float* out, int iters, float sa, unsigned seed) {
    float x[12];
    unsigned u[12];
    float va = sa + threadIdx.x * 1e-9f;
#pragma unroll
    for (int i = 0; i < 12; i++) { x[i] = threadIdx.x + i; u[i] = (threadIdx.x * 2654435761u + i * 40503u) ^ seed; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < 12; i++) {
                if (MODE == 0) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[i]) : "v"(va));
                if (MODE == 1) asm volatile("v_add_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2" : "=v"(x[i]) : "v"(u[i]), "v"(u[(i + 1) % 12]));
                if (MODE == 2) asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2" : "=v"(x[i]) : "v"(u[i]), "v"(u[(i + 1) % 12]));
                if (MODE == 3) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(x[i]) : "v"(u[i]));
                if (MODE == 4) asm volatile("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(x[i]) : "v"(u[i]), "v"(va));
                if (MODE == 5) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(va), "v"(__builtin_bit_cast(float, u[i] & 0x1ffu)));  // denormal operand
                if (MODE == 6) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(va), "v"(va));
                if (MODE == 7) asm volatile("v_add_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(x[i]) : "v"(x[(i + 1) % 12]), "v"(va));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) s += x[i] + u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// numerics: sum of 3 pairs of bytes times weights through the denormal route vs the plain float route
__global__ void knum(const unsigned* a, const unsigned* b, const float* w, float* plain, float* den, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned da = a[i], db = b[i];
    float acc_p = 0.25f, acc_d = 0.25f * 1.1920928955078125e-7f;  // bias, bias * 2^-23
    const float s126 = 8.507059173023462e37f;  // 2^126
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float fa = (float)((da >> (8 * k)) & 255), fb = (float)((db >> (8 * (k + 1))) & 255);
        acc_p = __builtin_fmaf(w[k], fa + fb, acc_p);
    }
    float p0, p1, p2;
    asm volatile("v_add_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1" : "=v"(p0) : "v"(da), "v"(db));
    asm volatile("v_add_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2" : "=v"(p1) : "v"(da), "v"(db));
    asm volatile("v_add_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_3" : "=v"(p2) : "v"(da), "v"(db));
    acc_d = __builtin_fmaf(w[0] * s126, p0, acc_d);
    acc_d = __builtin_fmaf(w[1] * s126, p1, acc_d);
    acc_d = __builtin_fmaf(w[2] * s126, p2, acc_d);
    plain[i] = acc_p;
    den[i] = acc_d * 8388608.0f;  // * 2^23
}
int main() {
    float* d;
    hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* names[] = {"v_add_f32 v,v", "v_add_f32_sdwa BYTE,BYTE", "v_add_u32_sdwa BYTE,BYTE", "v_cvt_f32_ubyte1", "v_mul_f32_sdwa BYTE,DWORD",
                           "v_fmac_f32 (denormal operand)", "v_fmac_f32 (normal operands)", "v_add_f32_sdwa DWORD,DWORD"};
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
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
    }
    // numerics
    const int n = 1 << 20;
    unsigned *ha = new unsigned[n], *hb = new unsigned[n], *da, *db;
    float hw[3] = {0.02431708f, -0.13508514f, 0.60792710f}, *dw, *dp, *dd;
    unsigned s = 777;
    for (int i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; ha[i] = s; s = s * 1664525u + 1013904223u; hb[i] = s; }
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dw, 12); hipMalloc(&dp, n * 4); hipMalloc(&dd, n * 4);
    hipMemcpy(da, ha, n * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb, n * 4, hipMemcpyHostToDevice); hipMemcpy(dw, hw, 12, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(knum, dim3(n / 256), dim3(256), 0, 0, da, db, dw, dp, dd, n);
    float *hp = new float[n], *hd = new float[n];
    hipMemcpy(hp, dp, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hd, dd, n * 4, hipMemcpyDeviceToHost);
    int same = 0; double maxd = 0;
    for (int i = 0; i < n; i++) { if (hp[i] == hd[i]) same++; double dlt = fabs((double)hp[i] - hd[i]); if (dlt > maxd) maxd = dlt; }
    printf("numerics: denormal route == plain route bit for bit in %d of %d cases, max |diff| %.3g (sample: plain %.6f denormal %.6f)\n", same, n, maxd, hp[5], hd[5]);
    return 0;
}
