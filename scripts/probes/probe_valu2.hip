// probe: settle the VALU cost model on gfx950.  Fully unrolled chains (no inner branch), the marching kernel's workgroup
// shape (384 threads, 39 KiB LDS -> 4 workgroups/CU) and the plain 256-thread / 8 waves-per-SIMD shape; scalar v_fma_f32 vs
// v_pk_fma_f32 vs VOP2 v_fmac; plus a shader-clock estimate from s_sleep against the 100 MHz s_memrealtime counter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE, int CH>
__global__ void k(float* out, int iters, float a, float b) {
    extern __shared__ uint8_t smem[];
    float x[CH];
    f2 p[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) { x[i] = threadIdx.x + i; p[i] = f2{x[i], x[i] + 1}; }
    const f2 aa = {a, a}, bb = {b, b};
    if (threadIdx.x == 0) smem[0] = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < CH; i++) {
                if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);
                if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], aa, bb);
                if (MODE == 2) asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(x[i]) : "s"(a));
                if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(aa), "v"(bb));
                if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(aa));
                if (MODE == 5) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[i]) : "s"(a));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < CH; i++) s += x[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_clock(unsigned long long* o) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 200; i++) __builtin_amdgcn_s_sleep(127);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    o[0] = r1 - r0;
    o[1] = c1 - c0;
}

int main() {
    float* d;
    hipMalloc(&d, 4096 * 384 * 4);
    unsigned long long* dc;
    hipMalloc(&dc, 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_clock<<<1, 64>>>(dc);
    unsigned long long hc[2];
    hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
    printf("s_sleep: 200 x s_sleep(127) = %llu realtime ticks (10 ns) , %llu s_memtime ticks -> s_memtime runs at %.1f MHz; "
           "if s_sleep(127) = 8128 cycles the shader clock is %.0f MHz\n",
           hc[0], hc[1], hc[1] / (hc[0] * 0.01), 200 * 8128.0 / (hc[0] * 0.01));
    const char* names[] = {"v_fma_f32 (compiler)", "pk fma (compiler)", "v_fmac_f32 VOP2 sgpr", "v_pk_fma_f32 asm", "v_pk_mul_f32 asm", "v_add_f32 VOP2 sgpr"};
    auto time = [&](auto kern, int blocks, int threads, int lds, int iters) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, d, iters, 1.0001f, 0.5f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        return best;
    };
    const int iters = 500;
    for (int shape = 0; shape < 3; shape++) {
        const int threads = shape == 0 ? 256 : 384, lds = shape == 0 ? 1024 : 39 * 1024;
        const int blocks = shape == 0 ? 2048 : (shape == 1 ? 1024 : 960);
        const double waves = (double)blocks * threads / 64;
        printf("--- %d blocks x %d threads, %d B LDS  (%.1f waves per SIMD)\n", blocks, threads, lds, waves / 1024);
#define RUN(M, CH)                                                                                           \
    {                                                                                                         \
        float ms = time(k<M, CH>, blocks, threads, lds, iters);                                               \
        double winst = waves * iters * 4 * CH;                                                                \
        printf("%-22s chains=%2d  %8.1f us   %.3f ns per wave-instr per SIMD\n", names[M], CH, ms * 1e3,      \
               ms * 1e6 / (winst / 1024));                                                                    \
    }
        RUN(0, 8) RUN(0, 12) RUN(1, 8) RUN(2, 12) RUN(3, 8) RUN(3, 12) RUN(4, 12) RUN(5, 12)
    }
    return 0;
}
