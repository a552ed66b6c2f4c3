// sim_tb_example.cpp -- a caller shaped like the reference's testbench (full_TB.h:99-180) linked against the
// MI355X library through hls_compat.hpp: pack pixels into stream_in, call lanczos(stream_in, stream_out), read
// stream_out back.  Build: see Makefile target `sim_tb_example`; run on a machine with a GPU:
//   ./sim_tb_example in.png observed.png
#define IN_WIDTH 256
#define IN_HEIGHT 256
#define OUT_WIDTH 512
#define OUT_HEIGHT 512
#define NUM_CHANNELS 3
#define LANCZOS_A 2
#define SCALE_N 2
#define SCALE_D 1
#include "hls_compat.hpp"
#include "image_io.h"

int main(int argc, char* argv[]) {
    if (argc < 3) return EXIT_FAILURE;
    int width, height, channels;
    hls::stream<byte_t> stream_in, stream_out;
    uint8_t* img = lz_image_load(argv[1], &width, &height, &channels, NUM_CHANNELS);
    if (img == NULL) {
        printf("Image was not loaded successfully.\n");
        return EXIT_FAILURE;
    }
    if (width != IN_WIDTH || height != IN_HEIGHT) {
        printf("Image has wrong dimensions (%i x %i).\n", width, height);
        return EXIT_FAILURE;
    }
    printf("Scale:%d/%d, WIDTHS %d -> %d\n", SCALE_N, SCALE_D, IN_WIDTH, OUT_WIDTH);
    for (int i = 0; i < IN_WIDTH * IN_HEIGHT; i++) {
        byte_t pixel;
        for (int j = 0; j < NUM_CHANNELS; j++) pixel.channel[j] = img[i * NUM_CHANNELS + j];
        stream_in.write(pixel);
    }
    lanczos(stream_in, stream_out);
    std::vector<uint8_t> ob((size_t)OUT_WIDTH * OUT_HEIGHT * NUM_CHANNELS);
    for (int i = 0; i < OUT_WIDTH * OUT_HEIGHT; i++) {
        byte_t r_pixel;
        stream_out.read(r_pixel);
        for (int j = 0; j < NUM_CHANNELS; j++) ob[(size_t)i * NUM_CHANNELS + j] = r_pixel.channel[j];
    }
    return lz_image_write_png(argv[2], OUT_WIDTH, OUT_HEIGHT, NUM_CHANNELS, ob.data(), OUT_WIDTH * NUM_CHANNELS) ? 0 : EXIT_FAILURE;
}
