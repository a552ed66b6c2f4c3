// hls_compat.hpp -- the reference's literal entry point `void lanczos(stream_t in, stream_t out)`
// (lanczos.h:121-126, lanczos.cpp:86-98) implemented over the C ABI, for callers shaped like main.cpp/sim_tb.
//
// Provides the few Xilinx-shaped types that caller needs, written from scratch (this is NOT the Vivado HLS
// library): a FIFO `hls::stream<T>` with read()/write()/empty()/size(), and a packed pixel `lz_packed<N>` that
// holds NUM_CHANNELS bytes with channel i in bits [8i+7:8i] (pack_blob/unpack_blob, worker.cpp:10-43;
// `R | G<<8 | B<<16`, full_TB.h:130) -- the same byte order as stb's interleaved RGB on little-endian.
//
// Compile-time parameters come from the same macros the reference's params.h defines (lanczos.h:9-31):
//   IN_WIDTH IN_HEIGHT OUT_WIDTH OUT_HEIGHT NUM_CHANNELS LANCZOS_A
// Unlike the reference (globals c, r, `static pos`: single shot, lanczos.cpp:17-18,54) this lanczos() can be
// called repeatedly, and from several threads.  Also here: the kernel.h call surface (kernel.h:2-8) --
// lanczos_kernel(input_idx_t, output_idx_t, scale_t) and raw_lanczos_kernel(kernel_t).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <vector>

extern "C" {
#include "../../include/lanczos_hip.h"
}

namespace hls {
template <typename T>
class stream {
   public:
    void write(const T& v) { q_.push_back(v); }
    T read() {
        T v = q_.front();
        q_.pop_front();
        return v;
    }
    void read(T& v) { v = read(); }
    bool empty() const { return q_.empty(); }
    size_t size() const { return q_.size(); }

   private:
    std::deque<T> q_;
};
}  // namespace hls

template <int NCH>
struct lz_packed {  // stands where the reference has `typedef ap_uint<8*NUM_CHANNELS> byte_t` (lanczos.h:90)
    uint8_t channel[NCH];
};

#if defined(IN_WIDTH) && defined(IN_HEIGHT) && defined(OUT_WIDTH) && defined(OUT_HEIGHT) && defined(NUM_CHANNELS) && \
    defined(LANCZOS_A)
typedef lz_packed<NUM_CHANNELS> byte_t;
typedef hls::stream<byte_t>& stream_t;  // lanczos.h:121

// SCALE_N / SCALE_D: the reference reduces OUT_WIDTH / IN_WIDTH by their gcd (lanczos.h:108-114); params.h may also name them
constexpr int lz_compat_gcd(int a, int b) { return b == 0 ? a : lz_compat_gcd(b, a % b); }
#ifndef SCALE_N
#define SCALE_N (OUT_WIDTH / lz_compat_gcd(OUT_WIDTH, IN_WIDTH))
#endif
#ifndef SCALE_D
#define SCALE_D (IN_WIDTH / lz_compat_gcd(OUT_WIDTH, IN_WIDTH))
#endif

// ---- kernel.h:2-8.  The reference's kernel_t is ap_fixed<8+BIT_PRECISION,8>; here the weights are the software twin's doubles
// (full_TB.h:51-53), which is what the resample itself uses.
typedef int input_idx_t;
typedef int output_idx_t;
typedef float scale_t;
typedef double kernel_t;
// kernel.cpp:12-18: a/pi^2 * sinpi(x) * sinpi(x/a) / x^2, 1 at 0 -- the Lanczos window L(x) itself
inline kernel_t raw_lanczos_kernel(kernel_t x) { return ::lanczos_kernel((double)x, LANCZOS_A); }
// kernel.cpp:50-67: the weight of input sample `in_idx` for output sample `out_idx`, i.e. the ROM entry |out*SCALE_D - in*SCALE_N|
// (= L at that distance / SCALE_N); like the reference's, this ignores `scale` and reads SCALE_N / SCALE_D
inline kernel_t lanczos_kernel(input_idx_t in_idx, output_idx_t out_idx, scale_t /*scale*/) {
    return lanczos_kernel_idx(in_idx, out_idx, SCALE_N, SCALE_D, LANCZOS_A);
}

// the adapter's context: created on first use (C++11: a function-local static is initialised exactly once, also under concurrent
// first calls), destroyed when the program ends
struct lz_compat_context {
    lanczos_ctx* ctx = nullptr;
    lz_compat_context() {
        if (lanczos_create(&ctx, 0) != LANCZOS_OK) ctx = nullptr;
    }
    ~lz_compat_context() {
        if (ctx) lanczos_destroy(ctx);
    }
    lz_compat_context(const lz_compat_context&) = delete;
    lz_compat_context& operator=(const lz_compat_context&) = delete;
};
inline lanczos_ctx* lz_compat_ctx() {
    static lz_compat_context holder;
    return holder.ctx;
}

inline void lanczos(stream_t streamin, stream_t streamout) {
    lanczos_ctx* ctx = lz_compat_ctx();   // (calls on it are serialised by the library)
    if (!ctx) {
        std::fprintf(stderr, "lanczos(): no HIP device\n");
        std::abort();  // the reference's lanczos() is void: no error path
    }
    std::vector<uint8_t> in((size_t)IN_WIDTH * IN_HEIGHT * NUM_CHANNELS), out((size_t)OUT_WIDTH * OUT_HEIGHT * NUM_CHANNELS);
    for (size_t i = 0; i < (size_t)IN_WIDTH * IN_HEIGHT; i++) {  // drain IN_W*IN_H packed pixels (full_TB.h:127-138)
        byte_t p = streamin.read();
        for (int c = 0; c < NUM_CHANNELS; c++) in[i * NUM_CHANNELS + c] = p.channel[c];
    }
    int rc = lanczos_u8(ctx, in.data(), IN_WIDTH, IN_HEIGHT, NUM_CHANNELS, out.data(), OUT_WIDTH, OUT_HEIGHT, LANCZOS_A);
    if (rc != LANCZOS_OK) {
        std::fprintf(stderr, "lanczos(): %s\n", lanczos_strerror(rc));
        std::abort();
    }
    for (size_t i = 0; i < (size_t)OUT_WIDTH * OUT_HEIGHT; i++) {  // raster order, like stream_out (lanczos.cpp:53-65)
        byte_t p;
        for (int c = 0; c < NUM_CHANNELS; c++) p.channel[c] = out[i * NUM_CHANNELS + c];
        streamout.write(p);
    }
}
#endif
