/*
 * oracle/selftest.c -- sanitizer driver for the CPU checker (TEST INFRASTRUCTURE ONLY; SURVEY.md 5:
 * "-fsanitize=address,undefined on the CPU restatement").  Built by `make -C oracle asan` into
 * oracle/_san/oracle_selftest and run by tests/test_sanitizers.py.
 *
 * Runs the restatement of full_TB.h:29-96 on small shapes that exercise every loop bound (tap ranges clipped on
 * all four sides, images smaller than the tap window, 1/3/4 channels, a = 2..4, rational scales, u8 and u16,
 * single- and multi-threaded) and prints one FNV-1a-64 digest per case; the test compares them with the digests of
 * the ordinary (non-sanitized) build, so a sanitizer report or a changed result both fail.
 */
#include <stdio.h>
#include <stdlib.h>

#include "lanczos_hls_model.h"
#include "lanczos_oracle.h"

static const int kShapes[][6] = {
    /* in_w, in_h, channels, scale_n, scale_d, a */
    {1, 1, 3, 2, 1, 3}, {2, 3, 1, 2, 1, 4}, {3, 2, 4, 3, 1, 3}, {5, 4, 3, 2, 1, 2}, {4, 7, 3, 3, 2, 3},
    {7, 1, 3, 2, 1, 3}, {1, 9, 4, 4, 1, 2}, {16, 12, 3, 2, 1, 3}, {12, 9, 3, 4, 3, 3}, {33, 17, 1, 5, 2, 4},
    {64, 40, 3, 2, 1, 3}, {40, 30, 4, 2, 1, 4},
};

int main(void) {
    for (size_t s = 0; s < sizeof(kShapes) / sizeof(kShapes[0]); s++) {
        oracle_cfg c;
        c.in_w = kShapes[s][0];
        c.in_h = kShapes[s][1];
        c.channels = kShapes[s][2];
        c.scale_n = kShapes[s][3];
        c.scale_d = kShapes[s][4];
        c.a = kShapes[s][5];
        c.out_w = c.in_w * c.scale_n / c.scale_d;
        c.out_h = c.in_h * c.scale_n / c.scale_d;
        const size_t n_in = (size_t)c.in_w * c.in_h * c.channels, n_out = (size_t)c.out_w * c.out_h * c.channels;
        for (int threads = 1; threads <= 3; threads += 2) {
            uint8_t* in8 = (uint8_t*)malloc(n_in);
            uint8_t* out8 = (uint8_t*)malloc(n_out ? n_out : 1);
            uint16_t* in16 = (uint16_t*)malloc(n_in * 2);
            uint16_t* out16 = (uint16_t*)malloc(n_out ? n_out * 2 : 2);
            if (!in8 || !out8 || !in16 || !out16) return 2;
            oracle_lcg_fill_u8(in8, n_in, 12345u + (unsigned)s);
            oracle_lcg_fill_u16(in16, n_in, 777u + (unsigned)s);
            if (oracle_expected_hwc_u8(&c, in8, out8, threads) != 0) return 3;
            printf("u8 %dx%dx%d %d/%d a%d t%d %016llx\n", c.in_w, c.in_h, c.channels, c.scale_n, c.scale_d, c.a, threads,
                   (unsigned long long)oracle_fnv1a64(out8, n_out));
            if (oracle_expected_hwc_u16(&c, in16, out16, threads) != 0) return 3;
            printf("u16 %dx%dx%d %d/%d a%d t%d %016llx\n", c.in_w, c.in_h, c.channels, c.scale_n, c.scale_d, c.a, threads,
                   (unsigned long long)oracle_fnv1a64(out16, n_out * 2));
            if (oracle_outofplace_hwc_u8(&c, in8, out8) != 0) return 3;
            printf("oop %dx%dx%d %d/%d a%d t%d %016llx K=%d\n", c.in_w, c.in_h, c.channels, c.scale_n, c.scale_d, c.a,
                   threads, (unsigned long long)oracle_fnv1a64(out8, n_out), oracle_inplace_rows(&c));
            if (oracle_hls_expected_hwc_u8(&c, in8, out8, threads) != 0) return 3;
            printf("hls8 %dx%dx%d %d/%d a%d t%d %016llx\n", c.in_w, c.in_h, c.channels, c.scale_n, c.scale_d, c.a, threads,
                   (unsigned long long)oracle_fnv1a64(out8, n_out));
            if (oracle_hls_expected_hwc_u16(&c, in16, out16, threads) != 0) return 3;
            printf("hls16 %dx%dx%d %d/%d a%d t%d %016llx\n", c.in_w, c.in_h, c.channels, c.scale_n, c.scale_d, c.a, threads,
                   (unsigned long long)oracle_fnv1a64(out16, n_out * 2));
            free(in8);
            free(out8);
            free(in16);
            free(out16);
        }
    }
    return 0;
}
