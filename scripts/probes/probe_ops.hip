// probe: issue cost (cycles per wave-instruction per SIMD, 8 waves/SIMD) of the VALU ops the kernels use
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ void k(float* out, int iters, float sa, float sb, unsigned ua) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    unsigned u0 = threadIdx.x * 2654435761u, u1 = u0 ^ 0x55aa55aa, u2 = u0 + 77, u3 = u1 + 99;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { REP8(asm volatile("v_fmac_f32 %0, %4, %0\n v_fmac_f32 %1, %4, %1\n v_fmac_f32 %2, %4, %2\n v_fmac_f32 %3, %4, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(sa));) }
        if (MODE == 1) { REP8(asm volatile("v_fma_f32 %0, %4, %0, %5\n v_fma_f32 %1, %4, %1, %5\n v_fma_f32 %2, %4, %2, %5\n v_fma_f32 %3, %4, %3, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(sa), "v"(sb));) }
        if (MODE == 2) { REP8(asm volatile("v_med3_f32 %0, %0, 0.5, %4\n v_med3_f32 %1, %1, 0.5, %4\n v_med3_f32 %2, %2, 0.5, %4\n v_med3_f32 %3, %3, 0.5, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(sb));) }
        if (MODE == 3) { REP8(asm volatile("v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));) }
        if (MODE == 4) { REP8(asm volatile("v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte2 %2, %5\n v_cvt_f32_ubyte3 %3, %5" : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"(u0), "v"(u1));) }
        if (MODE == 5) { REP8(asm volatile("v_cvt_pk_u8_f32 %0, %4, 0, %0\n v_cvt_pk_u8_f32 %1, %5, 1, %1\n v_cvt_pk_u8_f32 %2, %6, 2, %2\n v_cvt_pk_u8_f32 %3, %7, 3, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3));) }
        if (MODE == 6) { REP8(asm volatile("v_perm_b32 %0, %0, %1, %4\n v_perm_b32 %1, %1, %2, %4\n v_perm_b32 %2, %2, %3, %4\n v_perm_b32 %3, %3, %0, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "s"(ua));) }
        if (MODE == 7) { REP8(asm volatile("v_lshl_or_b32 %0, %0, 8, %1\n v_lshl_or_b32 %1, %1, 8, %2\n v_lshl_or_b32 %2, %2, 8, %3\n v_lshl_or_b32 %3, %3, 8, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        if (MODE == 8) { REP8(asm volatile("v_add_u32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_or_b32 %2, %2, %3\n v_sub_u32 %3, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        if (MODE == 9) { REP8(asm volatile("v_cvt_u32_f32 %0, %4\n v_cvt_u32_f32 %1, %5\n v_cvt_u32_f32 %2, %6\n v_cvt_u32_f32 %3, %7" : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3));) }
        if (MODE == 10) { REP8(asm volatile("v_sub_f32 %0, %0, %1\n v_min_f32 %1, %1, %2\n v_sub_f32 %2, %2, %3\n v_min_f32 %3, %3, %0" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));) }
        if (MODE == 11) { REP8(asm volatile("v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));) }
        if (MODE == 12) { REP8(asm volatile("v_min3_f32 %0, %0, %1, %2\n v_min3_f32 %1, %1, %2, %3\n v_min3_f32 %2, %2, %3, %0\n v_min3_f32 %3, %3, %0, %1" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));) }
        if (MODE == 13) { REP8(asm volatile("v_cvt_f32_u32 %0, %4\n v_cvt_f32_u32 %1, %5\n v_cvt_f32_u32 %2, %6\n v_cvt_f32_u32 %3, %7" : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));) }
        if (MODE == 14) { REP8(asm volatile("v_bfi_b32 %0, %0, %1, %2\n v_bfi_b32 %1, %1, %2, %3\n v_alignbyte_b32 %2, %2, %3, 2\n v_alignbyte_b32 %3, %3, %0, 1" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        if (MODE == 15) { REP8(asm volatile("v_add_f64 %0, %0, %1\n v_mul_f64 %1, %1, %0" : "+v"(*(double*)&x0), "+v"(*(double*)&x2) :);) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + u0 + u1 + u2 + u3;
}
const char* names[] = {"v_fmac_f32 (VOP2, sgpr)", "v_fma_f32 (VOP3, sgpr+2v)", "v_med3_f32", "v_floor_f32", "v_cvt_f32_ubyteN", "v_cvt_pk_u8_f32",
                       "v_perm_b32 (sgpr sel)", "v_lshl_or_b32", "add/and/or/sub u32", "v_cvt_u32_f32", "v_sub/min f32", "v_fract_f32",
                       "v_min3_f32", "v_cvt_f32_u32", "v_bfi/alignbyte", "f64 add/mul"};
int main() {
    float* d; hipMalloc(&d, 2048 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 500, blocks = 2048;
    float ms[16];
    for (int mode = 0; mode < 16; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
#define L(M) if (mode == M) k<M><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f, 0x07060100u);
            L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15)
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        hipEventElapsedTime(&ms[mode], e0, e1);
    }
    for (int mode = 0; mode < 16; mode++) {
        double winst = (double)blocks * 4 * iters * (mode == 15 ? 16 : 32);
        printf("%-28s rel to fmac: %.2f   (%.3f ms)\n", names[mode], (ms[mode] / winst) / (ms[0] / ((double)blocks * 4 * iters * 32)), ms[mode]);
    }
    return 0;
}
