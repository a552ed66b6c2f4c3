/*
 * image_io_selftest.c -- CPU-only driver for host/image_io.c (no GPU, no liblanczos_hip): built with
 * -fsanitize=address,undefined by `make -C lanczos-hls_amd san` and run by tests/test_sanitizers.py.
 * Round-trips PNG and PNM for 1/2/3/4 channels and odd sizes, checks channel conversion against the stb
 * conventions, and feeds truncated / hostile headers to the loaders (they must fail cleanly).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "image_io.h"

static int fail(const char* what) {
    fprintf(stderr, "image_io_selftest: %s (%s)\n", what, lz_image_last_error());
    return 1;
}

static int write_bytes(const char* path, const void* p, size_t n) {
    FILE* f = fopen(path, "wb");
    if (!f) return 0;
    fwrite(p, 1, n, f);
    return fclose(f) == 0;
}

int main(int argc, char** argv) {
    const char* dir = argc > 1 ? argv[1] : "/tmp";
    char path[1024];
    unsigned s = 12345u;
    for (int comp = 1; comp <= 4; comp++) {
        for (int k = 0; k < 3; k++) {
            const int w = 1 + 37 * k + comp, h = 1 + 11 * k;
            uint8_t* img = (uint8_t*)malloc((size_t)w * h * comp);
            if (!img) return fail("malloc");
            for (size_t i = 0; i < (size_t)w * h * comp; i++) {
                s = s * 1664525u + 1013904223u;
                img[i] = (uint8_t)(s >> 24);
            }
            snprintf(path, sizeof(path), "%s/lz_selftest_%d_%d.png", dir, comp, k);
            if (!lz_image_write_png(path, w, h, comp, img, 0)) return fail("write_png");
            int rw = 0, rh = 0, rc = 0;
            uint8_t* back = lz_image_load(path, &rw, &rh, &rc, 0);
            if (!back || rw != w || rh != h || rc != comp || memcmp(back, img, (size_t)w * h * comp)) return fail("png round trip");
            lz_image_free(back);
            for (int want = 1; want <= 4; want++) { /* every channel conversion runs (stb conventions) */
                back = lz_image_load(path, &rw, &rh, &rc, want);
                if (!back) return fail("png load with conversion");
                if (want == comp && memcmp(back, img, (size_t)w * h * comp)) return fail("identity conversion");
                if (want == 4 && comp == 3 && back[3] != 255) return fail("alpha must be 255");
                lz_image_free(back);
            }
            remove(path);
            if (comp == 1 || comp == 3) {
                snprintf(path, sizeof(path), "%s/lz_selftest_%d_%d.pnm", dir, comp, k);
                if (!lz_image_write_pnm(path, w, h, comp, img, 0)) return fail("write_pnm");
                back = lz_image_load(path, &rw, &rh, &rc, 0);
                if (!back || rw != w || rh != h || rc != comp || memcmp(back, img, (size_t)w * h * comp)) return fail("pnm round trip");
                lz_image_free(back);
                remove(path);
            }
            free(img);
        }
    }
    /* hostile / truncated inputs must be refused, not crash */
#define BAD(lit) {lit, sizeof(lit) - 1}
    static const struct { const char* p; size_t n; } bad[] = {
        BAD("P6\n99999999999999999999 1\n255\nxxx"), BAD("P6\n4 4\n255\nabc"), BAD("P5\n0 0\n255\n"),
        BAD("P6\n3 3\n65535\n"), BAD("P6"), BAD(""), BAD("\x89PNG\r\n\x1a\n"),
        BAD("\x89PNG\r\n\x1a\n\0\0\0\rIHDR\0\0\0\1\0\0\0\1\x08\x02\0\0\0garbage-garbage"),
    };
    for (size_t i = 0; i < sizeof(bad) / sizeof(bad[0]); i++) {
        snprintf(path, sizeof(path), "%s/lz_selftest_bad_%zu", dir, i);
        if (!write_bytes(path, bad[i].p, bad[i].n)) return fail("write hostile file");
        int w = 0, h = 0, c = 0;
        uint8_t* p = lz_image_load(path, &w, &h, &c, 3);
        remove(path);
        if (p) return fail("hostile input was accepted");
    }
    if (lz_image_load("/nonexistent/none.png", NULL, NULL, NULL, 3)) return fail("missing file accepted");
    printf("image_io_selftest ok\n");
    return 0;
}
