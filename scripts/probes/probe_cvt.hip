// probe: rounding/saturation of v_cvt_pk_u8_f32 and v_cvt_u32_f32 on gfx950 (informs the pack path)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float* x, unsigned* y, unsigned* z, int n) {
    int i = threadIdx.x;
    if (i < n) {
        y[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 1, 0xAABBCCDDu);
        z[i] = (unsigned)x[i];
    }
}
int main() {
    float h[] = {0.f, 0.49f, 0.5f, 0.51f, 0.99f, 1.0f, 1.5f, 2.5f, 3.5f, 254.5f, 254.999f, 255.0f, 255.4f, 255.5f, 255.9f, 256.f, 300.f, -0.1f, -0.5f, -0.9f, -3.f, 1e9f, NAN, 127.9999f};
    int n = sizeof(h) / 4;
    float* dx; unsigned *dy, *dz; unsigned y[64], z[64];
    hipMalloc(&dx, 256); hipMalloc(&dy, 256); hipMalloc(&dz, 256);
    hipMemcpy(dx, h, n * 4, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dx, dy, dz, n);
    hipMemcpy(y, dy, n * 4, hipMemcpyDeviceToHost); hipMemcpy(z, dz, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) printf("%12.4f -> pk_u8 byte1=%3u (dword %08x)  cvt_u32=%u\n", h[i], (y[i] >> 8) & 255, y[i], z[i]);
    return 0;
}
