/* image_convert.c -- `image_convert <in> <out.(png|ppm|pgm)> [channels]`: decode with image_io, encode with image_io.
 * A test tool (tests/test_image_io.py): built once on the own codec and, where the reference tree is present, once on its
 * stb headers (-DLANCZOS_WITH_STB), so the two decoders can be compared file by file. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "image_io.h"

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s <in> <out.png|ppm|pgm> [channels]\n", argv[0]);
        return 2;
    }
    const int want = argc > 3 ? atoi(argv[3]) : 3;
    int w = 0, h = 0, c = 0;
    uint8_t* img = lz_image_load(argv[1], &w, &h, &c, want);
    if (!img) {
        fprintf(stderr, "load failed: %s\n", lz_image_last_error());
        return 1;
    }
    const size_t n = strlen(argv[2]);
    const int ok = (n > 4 && !strcmp(argv[2] + n - 4, ".png")) ? lz_image_write_png(argv[2], w, h, want, img, w * want)
                                                               : lz_image_write_pnm(argv[2], w, h, want, img, w * want);
    printf("%dx%d channels_in_file=%d\n", w, h, c);
    lz_image_free(img);
    return ok ? 0 : 1;
}
