// probe: does an SGPR (constant-bus) source slow a VALU instruction on gfx950?  asm volatile bodies, 12 independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f2 __attribute__((ext_vector_type(2)));
#define R4(x) x x x x
template <int MODE>
__global__ void k(float* out, int iters, float sa, float sb) {
    float x[12];
    f2 p[12];
    float va = sa + threadIdx.x * 1e-9f, vb = sb;
    f2 pa = {va, va}, pb = {vb, vb};
    unsigned u[12];
    const f2 sp = {sa, sa};
#pragma unroll
    for (int i = 0; i < 12; i++) { x[i] = threadIdx.x + i; p[i] = f2{x[i], x[i] + 1}; u[i] = threadIdx.x * 2654435761u + i; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < 12; i++) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(x[i]) : "v"(va), "v"(vb));
                if (MODE == 1) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(x[i]) : "s"(sa), "v"(vb));
                if (MODE == 2) asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(x[i]) : "v"(va));
                if (MODE == 3) asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(x[i]) : "s"(sa));
                if (MODE == 4) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[i]) : "v"(va));
                if (MODE == 5) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[i]) : "s"(sa));
                if (MODE == 6) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(p[i]) : "v"(pa), "v"(pb));
                if (MODE == 7) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(p[i]) : "s"(sp), "v"(pb));
                if (MODE == 8) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(x[i]) : "v"(u[i]));
                if (MODE == 9) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u[i]) : "v"(x[i]));
                if (MODE == 10) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(va), "v"(x[(i + 1) % 12]));
                if (MODE == 11) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[i]) : "v"(va));
                if (MODE == 12) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i]));
                if (MODE == 13) asm volatile("v_fma_f32 %0, 0.5, %0, 1.0" : "+v"(x[i]));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) s += x[i] + p[i].x + p[i].y + u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* d;
    hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* names[] = {"v_fma_f32 v,v,v", "v_fma_f32 s,v,v", "v_fmac_f32 v,v", "v_fmac_f32 s,v", "v_add_f32 v,v", "v_add_f32 s,v",
                           "v_pk_fma_f32 v,v,v", "v_pk_fma_f32 s,v,v", "v_cvt_f32_ubyte1", "v_cvt_pk_u8_f32", "v_fma_f32 v,v2,v (3 distinct)",
                           "v_mul_f32 v,v", "v_fma_f32 x,x,x", "v_fma_f32 0.5,x,1.0"};
    const int iters = 500;
    for (int wps : {8, 4, 2, 1}) {
        const int blocks = 256 * wps;
        printf("--- %d waves per SIMD\n", wps);
#define RUN(M)                                                                                                  \
    {                                                                                                           \
        float best = 1e9;                                                                                       \
        for (int rep = 0; rep < 3; rep++) {                                                                     \
            hipEventRecord(e0);                                                                                 \
            hipLaunchKernelGGL(k<M>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);                  \
            hipEventRecord(e1);                                                                                 \
            hipEventSynchronize(e1);                                                                            \
            float ms;                                                                                           \
            hipEventElapsedTime(&ms, e0, e1);                                                                   \
            if (ms < best) best = ms;                                                                           \
        }                                                                                                       \
        double winst = (double)blocks * 4 * iters * 48;                                                         \
        printf("%-30s %8.1f us  %.3f ns per wave-instr per SIMD\n", names[M], best * 1e3, best * 1e6 / (winst / 1024)); \
    }
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13)
    }
    return 0;
}
