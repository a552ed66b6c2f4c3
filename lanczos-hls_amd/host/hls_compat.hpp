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
// called repeatedly.
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

inline void lanczos(stream_t streamin, stream_t streamout) {
    static lanczos_ctx* ctx = nullptr;
    if (!ctx && lanczos_create(&ctx, 0) != LANCZOS_OK) {
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
