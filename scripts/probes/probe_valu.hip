// probe: VALU issue rate on gfx950 -- plain v_fma_f32 vs v_pk_fma_f32 vs v_cvt_pk_u8_f32, waves/SIMD swept
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
    float2v aa = {a, a}, bb = {b, b};
    unsigned u0 = 0, u1 = 0, u2 = 0, u3 = 0;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                p0 = __builtin_elementwise_fma(p0, aa, bb); p1 = __builtin_elementwise_fma(p1, aa, bb);
                p2 = __builtin_elementwise_fma(p2, aa, bb); p3 = __builtin_elementwise_fma(p3, aa, bb);
                p4 = __builtin_elementwise_fma(p4, aa, bb); p5 = __builtin_elementwise_fma(p5, aa, bb);
                p6 = __builtin_elementwise_fma(p6, aa, bb); p7 = __builtin_elementwise_fma(p7, aa, bb);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                u0 = __builtin_amdgcn_cvt_pk_u8_f32(x0, 0, u0); u1 = __builtin_amdgcn_cvt_pk_u8_f32(x1, 1, u1);
                u2 = __builtin_amdgcn_cvt_pk_u8_f32(x2, 2, u2); u3 = __builtin_amdgcn_cvt_pk_u8_f32(x3, 3, u3);
                x0 += 1.f; x1 += 1.f; x2 += 1.f; x3 += 1.f;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x +
                                                 p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y + u0 + u1 + u2 + u3;
}
int main() {
    float* d; hipMalloc(&d, 256 * 1024 * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int wps = 1; wps <= 8; wps *= 2) {
        for (int mode = 0; mode < 3; mode++) {
            int blocks = 256 * wps;  // 256 threads = 4 waves per block -> wps waves per SIMD when blocks = 256*wps
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (mode == 0) k<0><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
                if (mode == 1) k<1><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
                if (mode == 2) k<2><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double winst = (double)blocks * 4 * iters * (mode == 2 ? 128 : 64);  // wave-instructions
            double per_simd_cycle = winst / 1024 / (ms * 1e-3 * 2.4e9);
            printf("waves/SIMD=%d mode=%s  %.3f ms  wave-instr/cycle/SIMD=%.3f  (cycles per wave-instr %.2f)\n", wps,
                   mode == 0 ? "v_fma_f32   " : mode == 1 ? "v_pk_fma_f32" : "cvt_pk+add  ", ms, per_simd_cycle, 1.0 / per_simd_cycle);
        }
    }
    return 0;
}
