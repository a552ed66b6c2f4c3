// probe: HBM write bandwidth for the marching kernel's store pattern vs alternatives (16 x 3840x2160x3 frames)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int W = 11520, H = 2160, FR = 16;
// mode 0: our pattern: WG = 768-byte strip x 360 rows; thread = dword column; 6 waves = 2 row groups; barrier / 24 rows
// mode 1: same without barrier
// mode 2: strip 768 B, thread = 16 bytes (48 lanes per row, wave = 64 lanes -> 1.33 rows), no barrier
// mode 3: full-row contiguous: WG = 256 threads x 16 B = 4096 B chunks streaming (plain memset order)
// mode 4: strip of 3072 B (4x wider), thread = dword column, 768 threads
template <int MODE>
__global__ void k(uint8_t* out, int strip_b, int rows_per_wg) {
    const int tid = threadIdx.x;
    if (MODE == 3) {
        size_t i = ((size_t)blockIdx.x * blockDim.x + tid) * 16;
        size_t total = (size_t)W * H * FR;
        for (; i < total; i += (size_t)gridDim.x * blockDim.x * 16) *(uint4*)(out + i) = make_uint4(tid, 1, 2, 3);
        return;
    }
    const int strips = W / strip_b;
    const int tx = blockIdx.x % strips, chunk = blockIdx.x / strips, frame = blockIdx.y;
    uint8_t* base = out + (size_t)frame * W * H + (size_t)chunk * rows_per_wg * W + tx * strip_b;
    if (MODE == 0 || MODE == 1 || MODE == 4) {
        const int ncol = strip_b / 4;
        const int grp = tid / ncol, col = tid % ncol;
        for (int t = 0; t < rows_per_wg; t += 24) {
            for (int r = 0; r < 12; r++) {
                const int y = t + grp * 12 + r;
                *(uint32_t*)(base + (size_t)y * W + col * 4) = tid + y;
            }
            if (MODE == 0) __syncthreads();
        }
    } else if (MODE == 2) {
        const int lanes_per_row = strip_b / 16;  // 48
        const int rows_per_pass = blockDim.x / lanes_per_row;  // 8 for 384 threads
        const int r0 = tid / lanes_per_row, c = tid % lanes_per_row;
        for (int y = r0; y < rows_per_wg; y += rows_per_pass) *(uint4*)(base + (size_t)y * W + c * 16) = make_uint4(tid, y, 2, 3);
    }
}
int main() {
    uint8_t* d; size_t bytes = (size_t)W * H * FR;
    hipMalloc(&d, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 5; mode++) {
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0);
            if (mode == 0) k<0><<<dim3(15 * 6, FR), 384>>>(d, 768, 360);
            if (mode == 1) k<1><<<dim3(15 * 6, FR), 384>>>(d, 768, 360);
            if (mode == 2) k<2><<<dim3(15 * 6, FR), 384>>>(d, 768, 360);
            if (mode == 3) k<3><<<2048, 256>>>(d, 0, 0);
            if (mode == 4) k<4><<<dim3((W / 2304) * 18, FR), 1024 + 128>>>(d, 2304, 120);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("mode %d: %.1f us  %.2f TB/s\n", mode, best * 1e3, bytes / (best * 1e-3) / 1e12);
    }
    return 0;
}
