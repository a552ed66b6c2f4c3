// probe: do VALU work and the marching kernel's store stream overlap on a CU, or do their times add?
// Shape of k_march<u8,3,2,3>: 384-thread workgroups, 40 KiB LDS (4 per CU), 768-byte strips, 24 output rows per tick,
// thread = dword column of a 12-row group, one barrier per tick.  Per row: NF fmas (12 independent chains), 1 store.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
constexpr int W = 11520, H = 2160, FR = 16;

template <int AUX, int WIDE>
__global__ __launch_bounds__(384) void k(uint8_t* out, int rows_per_wg, int nf, int do_store, int do_barrier, float w) {
    extern __shared__ uint8_t smem[];
    const int tid = threadIdx.x;
    const int strips = W / 768;
    const int tx = blockIdx.x % strips, chunk = blockIdx.x / strips, frame = blockIdx.y;
    uint8_t* fbase = out + (size_t)frame * W * H;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(fbase, 0, (unsigned)((size_t)W * H), 0x00020000);
    const int grp = tid / 192, col = tid % 192;
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; i++) acc[i] = (float)(tid + i);
    if (tid == 0) smem[0] = 1;
    for (int t = 0; t < rows_per_wg; t += 24) {
        int soff = ((chunk * rows_per_wg + t + grp * 12) * W);
        if (WIDE == 0) {
#pragma unroll
            for (int r = 0; r < 12; r++) {
                for (int j = 0; j < nf; j += 12) {
#pragma unroll
                    for (int i = 0; i < 12; i++) acc[i] = __builtin_fmaf(acc[i], w, 1.0f);
                }
                if (do_store) {
                    unsigned pk = __builtin_amdgcn_cvt_pk_u8_f32(acc[r], 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(pk | (unsigned)r, rs, (unsigned)(tx * 768 + col * 4), soff, AUX);
                }
                soff += W;
            }
        } else {
            // same bytes, 4x fewer store instructions: a thread owns 16 bytes of 3 rows (layout differs: timing only)
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int r = 0; r < 12; r++) {
                for (int j = 0; j < nf; j += 12) {
#pragma unroll
                    for (int i = 0; i < 12; i++) acc[i] = __builtin_fmaf(acc[i], w, 1.0f);
                }
                if (do_store && (r & 3) == 3) {
                    // 48 lanes cover one 768-byte row: tid -> (row, 16-byte column) inside a 4-row block per group
                    const int rr = col / 48, cc = col % 48;
                    u4 v = {__float_as_uint(acc[r]), __float_as_uint(acc[r - 1]), __float_as_uint(acc[r - 2]), __float_as_uint(acc[r - 3])};
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (unsigned)(tx * 768 + cc * 16 + rr * W), soff - 3 * W, AUX);
                }
                soff += W;
            }
        }
        if (do_barrier) __syncthreads();
    }
    if (acc[0] + acc[5] == 1234.5f) out[tid] = 1;
}

int main(int argc, char** argv) {
    uint8_t* d;
    size_t bytes = (size_t)W * H * FR;
    hipMalloc(&d, bytes);
    hipMemset(d, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int lds = argc > 1 ? atoi(argv[1]) : 39 * 1024;
    auto run = [&](const char* name, auto kern, int nf, int st, int bar) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        float best = 1e9;
        for (int rep = 0; rep < 6; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(15 * 4, FR), dim3(384), lds, 0, d, 540, nf, st, bar, 0.999f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-28s nf=%3d store=%d barrier=%d : %7.1f us\n", name, nf, st, bar, best * 1e3);
    };
    for (int bar = 0; bar < 2; bar++) {
        for (int nf : {0, 24, 36, 48}) {
            run("dword", k<0, 0>, nf, 0, bar);
            run("dword", k<0, 0>, nf, 1, bar);
        }
        run("dword nt(aux=2)", k<2, 0>, 36, 1, bar);
        run("dword sc0(aux=1)", k<1, 0>, 36, 1, bar);
        run("dword sc1(aux=16)", k<16, 0>, 36, 1, bar);
        run("x4", k<0, 1>, 0, 1, bar);
        run("x4", k<0, 1>, 36, 1, bar);
        run("x4 nt", k<2, 1>, 36, 1, bar);
    }
    return 0;
}
