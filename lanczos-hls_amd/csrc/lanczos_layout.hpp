// lanczos_layout.hpp -- planar <-> interleaved on the device.
//
// The reference's software model and its testbench keep PLANAR frames, `byte img_in[NUM_CHANNELS][IN_HEIGHT][IN_WIDTH]`
// (full_TB.h:20-21), and convert from / to the interleaved stb buffer with two host loops (full_TB.h:127-138 and
// 146-165; the packed pixel is R | G<<8 | B<<16, worker.cpp:10-43).  These kernels are that glue for callers who hold
// planar arrays: pure byte movement, HBM-bound -- every thread moves 4 bytes per plane with aligned dword accesses and
// re-packs them in registers.
#pragma once
#include "lanczos_kernels_common.hpp"

namespace lz {

// planar [C][H][W] -> interleaved [H][W][C], T = uint8_t / uint16_t.  One thread: PX = 4/sizeof(T) consecutive pixels.
template <typename T, int C>
__global__ __launch_bounds__(256) void k_planar_to_interleaved(const T* __restrict__ planar, T* __restrict__ inter, int w, int h,
                                                               size_t frame_samples) {
    constexpr int PX = 4 / (int)sizeof(T);
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * PX;
    const int y = blockIdx.y;
    if (x0 >= w) return;
    const T* pf = planar + (size_t)blockIdx.z * frame_samples;
    T* of = inter + (size_t)blockIdx.z * frame_samples;
    const size_t plane = (size_t)w * h;
    const size_t row = (size_t)y * w;
    // the dword path needs the whole group inside the row and 4-byte aligned addresses on both sides
    const bool fast = x0 + PX <= w && ((row + x0) * sizeof(T)) % 4 == 0 && (plane * sizeof(T)) % 4 == 0 &&
                      (((row + x0) * C) * sizeof(T)) % 4 == 0 && (((uintptr_t)pf | (uintptr_t)of) & 3) == 0;
    if (fast) {
        uint32_t pl[C];
#pragma unroll
        for (int c = 0; c < C; c++) pl[c] = *(const uint32_t*)(pf + c * plane + row + x0);
        uint32_t ow[C];
#pragma unroll
        for (int j = 0; j < C; j++) {
            uint32_t v = 0;
#pragma unroll
            for (int e = 0; e < PX; e++) {
                const int k = j * PX + e, p = k / C, c = k % C;  // output sample k of the group = pixel p, channel c
                v |= ((pl[c] >> (8 * sizeof(T) * p)) & (sizeof(T) == 1 ? 0xffu : 0xffffu)) << (8 * sizeof(T) * e);
            }
            ow[j] = v;
        }
        uint32_t* dst = (uint32_t*)(of + (row + x0) * C);
#pragma unroll
        for (int j = 0; j < C; j++) dst[j] = ow[j];
    } else {
        for (int p = 0; p < PX && x0 + p < w; p++)
#pragma unroll
            for (int c = 0; c < C; c++) of[(row + x0 + p) * C + c] = pf[c * plane + row + x0 + p];
    }
}

// interleaved [H][W][C] -> planar [C][H][W]
template <typename T, int C>
__global__ __launch_bounds__(256) void k_interleaved_to_planar(const T* __restrict__ inter, T* __restrict__ planar, int w, int h,
                                                               size_t frame_samples) {
    constexpr int PX = 4 / (int)sizeof(T);
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * PX;
    const int y = blockIdx.y;
    if (x0 >= w) return;
    const T* sf = inter + (size_t)blockIdx.z * frame_samples;
    T* pf = planar + (size_t)blockIdx.z * frame_samples;
    const size_t plane = (size_t)w * h;
    const size_t row = (size_t)y * w;
    const bool fast = x0 + PX <= w && ((row + x0) * sizeof(T)) % 4 == 0 && (plane * sizeof(T)) % 4 == 0 &&
                      (((row + x0) * C) * sizeof(T)) % 4 == 0 && (((uintptr_t)pf | (uintptr_t)sf) & 3) == 0;
    if (fast) {
        uint32_t iw[C];
        const uint32_t* src = (const uint32_t*)(sf + (row + x0) * C);
#pragma unroll
        for (int j = 0; j < C; j++) iw[j] = src[j];
#pragma unroll
        for (int c = 0; c < C; c++) {
            uint32_t v = 0;
#pragma unroll
            for (int p = 0; p < PX; p++) {
                const int k = p * C + c;  // sample k of the group
                v |= ((iw[k / PX] >> (8 * sizeof(T) * (k % PX))) & (sizeof(T) == 1 ? 0xffu : 0xffffu)) << (8 * sizeof(T) * p);
            }
            *(uint32_t*)(pf + c * plane + row + x0) = v;
        }
    } else {
        for (int p = 0; p < PX && x0 + p < w; p++)
#pragma unroll
            for (int c = 0; c < C; c++) pf[c * plane + row + x0 + p] = sf[(row + x0 + p) * C + c];
    }
}

template <typename T, int C>
inline hipError_t layout_launch_t(bool to_interleaved, const void* src, void* dst, int w, int h, int frames, hipStream_t stream) {
    constexpr int PX = 4 / (int)sizeof(T);
    const int groups = (w + PX - 1) / PX;
    dim3 grid((groups + 255) / 256, h, frames);
    const size_t frame_samples = (size_t)w * h * C;
    if (to_interleaved)
        hipLaunchKernelGGL((k_planar_to_interleaved<T, C>), grid, dim3(256), 0, stream, (const T*)src, (T*)dst, w, h, frame_samples);
    else
        hipLaunchKernelGGL((k_interleaved_to_planar<T, C>), grid, dim3(256), 0, stream, (const T*)src, (T*)dst, w, h, frame_samples);
    return hipGetLastError();
}

inline hipError_t layout_launch(bool to_interleaved, const void* src, void* dst, int w, int h, int channels, int bytes_per_sample,
                                int frames, hipStream_t stream) {
#define X(T, C) \
    if (bytes_per_sample == (int)sizeof(T) && channels == C) return layout_launch_t<T, C>(to_interleaved, src, dst, w, h, frames, stream);
    X(uint8_t, 1) X(uint8_t, 3) X(uint8_t, 4) X(uint16_t, 1) X(uint16_t, 3) X(uint16_t, 4)
#undef X
    return hipErrorNotSupported;
}

}  // namespace lz
