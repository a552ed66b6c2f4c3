// probe: does the memory floor of the marching kernel's traffic (probe_store2: 768-byte store segments 6.0 TB/s, 416-byte
// load segments 4.9 TB/s, both 5.66 TB/s) depend on the STRIP WIDTH?  Same 32 x 4K RGB8 output frames / 1080p input frames,
// buffers cycled past the Infinity Cache, one workgroup per (strip, chunk, frame), barrier per 24 output rows; strips of
// 384 / 768 / 1536 / 3072 output bytes (= half as many input bytes + a 32-byte halo), thread = dword column.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <vector>
constexpr int OW = 11520, OH = 2160, IW = 5760, IH = 1080, FR = 32, SETS = 3;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: stores only, 1: loads only, 2: both
template <int MODE>
__global__ __launch_bounds__(1024) void k(const uint8_t* in, uint8_t* out, int strip_b, int rows, int groups) {
    const int tid = threadIdx.x;
    const int strips = OW / strip_b;
    const int tx = blockIdx.x % strips, chunk = blockIdx.x / strips, frame = blockIdx.y;
    uint8_t* obase = out + (size_t)frame * OW * OH + (size_t)chunk * rows * OW + tx * strip_b;
    const uint8_t* ibase = in + (size_t)frame * IW * IH + (size_t)chunk * (rows / 2) * IW + tx * (strip_b / 2);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, (unsigned)(rows * OW), 0x00020000);
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(ibase), 0, (unsigned)((rows / 2) * IW), 0x00020000);
    const int ncol = strip_b / 4, grp = tid / ncol, col = tid % ncol;
    const int rpg = 24 / groups;                       // output rows per group and tick
    const int cpr = (strip_b / 2 + 32 + 15) / 16;      // 16-byte chunks per input row segment (with halo)
    const int nload = 12 * cpr;
    unsigned acc = tid;
    for (int t = 0; t < rows / 24; t++) {
        u32x4 v = {0, 0, 0, 0};
        if (MODE >= 1) {
            for (int i = tid; i < nload; i += blockDim.x) {
                const int r = i / cpr, c = i % cpr;
                u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(irsrc, (unsigned)((t * 12 + r) * IW + c * 16), 0, 0);
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            }
        }
        if (MODE != 1 && grp < groups) {
            for (int r = 0; r < rpg; r++) {
                const int y = t * 24 + grp * rpg + r;
                __builtin_amdgcn_raw_buffer_store_b32(acc + y, orsrc, (unsigned)(col * 4), y * OW, 18);
            }
        }
        if (MODE >= 1) acc += v.x + v.y + v.z + v.w;
        __syncthreads();
    }
    if (MODE >= 1 && acc == 0x12345678u) out[0] = 1;
}

int main() {
    const size_t ob = (size_t)OW * OH * FR, ib = (size_t)IW * IH * FR;
    uint8_t *o[SETS], *in[SETS];
    for (int s = 0; s < SETS; s++) {
        if (hipMalloc(&o[s], ob) != hipSuccess || hipMalloc(&in[s], ib + 4096) != hipSuccess) return 1;
        (void)hipMemset(in[s], s + 1, ib);
    }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    struct Cfg { int strip_b, threads, groups, chunks; };
    // chunks chosen so that strips x chunks x 32 frames is ~960-1024 workgroups of equal LDS-free weight
    const Cfg cfgs[] = {{384, 192, 2, 1}, {768, 384, 2, 2}, {1536, 768, 2, 4}, {3072, 768, 1, 8}, {768, 384, 2, 4}, {1536, 768, 2, 2}};
    const char* mnames[] = {"stores", "loads ", "both  "};
    for (const Cfg& c : cfgs) {
        const int strips = OW / c.strip_b, rows = OH / c.chunks;
        for (int mode = 0; mode < 3; mode++) {
            std::vector<float> ts;
            for (int rep = 0; rep < 9; rep++) {
                const int s = rep % SETS;
                (void)hipEventRecord(e0);
                const dim3 grid(strips * c.chunks, FR);
                if (mode == 0) k<0><<<grid, c.threads>>>(in[s], o[s], c.strip_b, rows, c.groups);
                if (mode == 1) k<1><<<grid, c.threads>>>(in[s], o[s], c.strip_b, rows, c.groups);
                if (mode == 2) k<2><<<grid, c.threads>>>(in[s], o[s], c.strip_b, rows, c.groups);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 2) ts.push_back(ms);
            }
            std::sort(ts.begin(), ts.end());
            const float med = ts[ts.size() / 2];
            const double bytes = mode == 0 ? (double)ob : (mode == 1 ? (double)ib : (double)(ob + ib));
            printf("strip %4d B x %d chunks (%4d workgroups of %4d threads)  %s median %7.1f us  %.2f TB/s\n", c.strip_b, c.chunks,
                   strips * c.chunks * FR, c.threads, mnames[mode], med * 1e3, bytes / (med * 1e-3) / 1e12);
        }
    }
    return 0;
}
