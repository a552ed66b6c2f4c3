// probe: store throughput of ONE compute unit (is the 11-13 B/clk per CU seen chip-wide an HBM share or a CU limit?)
// grid = n workgroups of 384 threads (6 waves), each storing the marching kernel's pattern (768-byte rows, dword per lane)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int W = 11520;
template <int WIDE>
__global__ __launch_bounds__(384) void k(uint8_t* out, int rows) {
    const int tid = threadIdx.x;
    uint8_t* base = out + (size_t)blockIdx.x * rows * W;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (unsigned)((size_t)rows * W), 0x00020000);
    if (WIDE == 0) {
        const int grp = tid / 192, col = tid % 192;
        for (int t = 0; t + 24 <= rows; t += 24) {
            int soff = (t + grp * 12) * W;
#pragma unroll
            for (int r = 0; r < 12; r++) {
                __builtin_amdgcn_raw_buffer_store_b32((unsigned)(tid + r), rs, (unsigned)(col * 4), soff, 0);
                soff += W;
            }
        }
    } else {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const int rr = tid / 48, cc = tid % 48;  // 8 rows x 48 16-byte columns per pass
        for (int t = 0; t + 8 <= rows; t += 8) {
            u4 v = {(unsigned)tid, 1u, 2u, 3u};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, (unsigned)(cc * 16 + rr * W), t * W, 0);
        }
    }
}
int main() {
    uint8_t* d;
    const int rows = 2160;
    const size_t per_wg = (size_t)rows * W;
    hipMalloc(&d, per_wg * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wide = 0; wide < 2; wide++)
        for (int n : {1, 2, 4, 8, 32, 64}) {
            float best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                hipEventRecord(e0);
                if (wide) hipLaunchKernelGGL(k<1>, dim3(n), dim3(384), 0, 0, d, rows);
                else hipLaunchKernelGGL(k<0>, dim3(n), dim3(384), 0, 0, d, rows);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double bytes = (double)n * rows * 768;
            printf("%s  %2d workgroup(s): %8.1f us  %7.1f GB/s total  %6.1f GB/s per workgroup (= per CU)\n", wide ? "x4   " : "dword", n,
                   best * 1e3, bytes / (best * 1e-3) / 1e9, bytes / n / (best * 1e-3) / 1e9);
        }
    return 0;
}
