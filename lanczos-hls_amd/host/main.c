/*
 * main.c -- the harness: what main.cpp:15-19 + sim_tb (full_TB.h:99-180) do around the resample entry point,
 * with the reference's compile-time parameters (params.h) turned into command-line arguments.
 *
 *   lanczos_upscale <in.(png|ppm|pgm)> <out.(png|ppm|pgm)> [--scale N[/D]] [--a A] [--channels C]
 *                   [--exact | --hls] [--device D] [--repeat K]
 *                   [--devices 0-7 | 0,2,5] [--frames F] [--split frames|rows] [--root]   (several GPUs of one node, plain C:
 *                   the image is replicated into a batch of F frames and the batch -- or every frame's rows -- is split
 *                   over the devices by lanczos_resample_multi_host; the first result frame is written.  --root: the batch
 *                   is resident on the first device instead and travels by one RCCL scatter / gather over xGMI
 *                   (lanczos_resample_multi_root); compute-only and exchange-inclusive rates are printed separately,
 *                   SURVEY.md 8e "Reporting")
 *
 * load (interleaved u8 HWC, like stbi_load, full_TB.h:107) -> checks with the reference's messages and
 * EXIT_FAILURE (full_TB.h:110-123) -> "Scale:%d/%d, WIDTHS %d -> %d" (full_TB.h:124) -> the resample, through
 * the C ABI (include/lanczos_hip.h; replaces lanczos(stream_in, stream_out), full_TB.h:140) -> write the image
 * (like stbi_write_png, full_TB.h:172).  Host code is plain C; everything GPU is behind the extern "C" shim.
 * There is no software fallback: without a GPU the program reports the error and fails.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/lanczos_hip.h"
#include "image_io.h"

static int ends_with(const char* s, const char* suf) {
    size_t n = strlen(s), m = strlen(suf);
    return n >= m && strcmp(s + n - m, suf) == 0;
}

int main(int argc, char* argv[]) {
    const char *in_path = NULL, *out_path = NULL;
    int scale_n = 2, scale_d = 1, a = 3, want_channels = 3, exact = 0, hls = 0, device = 0, repeat = 1;
    int devices[64], n_devices = 0, frames = 1, split = LANCZOS_SPLIT_FRAMES, root = 0;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--scale") && i + 1 < argc) {
            scale_d = 1;
            if (sscanf(argv[++i], "%d/%d", &scale_n, &scale_d) < 1) scale_n = 0;
        } else if (!strcmp(argv[i], "--a") && i + 1 < argc) {
            a = atoi(argv[++i]);
        } else if (!strcmp(argv[i], "--channels") && i + 1 < argc) {
            want_channels = atoi(argv[++i]);
        } else if (!strcmp(argv[i], "--device") && i + 1 < argc) {
            device = atoi(argv[++i]);
        } else if (!strcmp(argv[i], "--repeat") && i + 1 < argc) {
            repeat = atoi(argv[++i]);
        } else if (!strcmp(argv[i], "--exact")) {
            exact = 1;
        } else if (!strcmp(argv[i], "--hls")) {
            hls = 1;
        } else if (!strcmp(argv[i], "--root")) {
            root = 1;
        } else if (!strcmp(argv[i], "--frames") && i + 1 < argc) {
            frames = atoi(argv[++i]);
        } else if (!strcmp(argv[i], "--split") && i + 1 < argc) {
            split = !strcmp(argv[++i], "rows") ? LANCZOS_SPLIT_ROWS : LANCZOS_SPLIT_FRAMES;
        } else if (!strcmp(argv[i], "--devices") && i + 1 < argc) { /* "0-7" or "0,2,5" */
            const char* p = argv[++i];
            int lo, hi;
            if (sscanf(p, "%d-%d", &lo, &hi) == 2) {
                for (int v = lo; v <= hi && n_devices < 64; v++) devices[n_devices++] = v;
            } else {
                while (*p && n_devices < 64) {
                    devices[n_devices++] = atoi(p);
                    while (*p && *p != ',') p++;
                    if (*p == ',') p++;
                }
            }
        } else if (!in_path) {
            in_path = argv[i];
        } else if (!out_path) {
            out_path = argv[i];
        }
    }
    if (!in_path || !out_path) {
        fprintf(stderr, "usage: %s <in.png|ppm> <out.png|ppm> [--scale N[/D]] [--a A] [--channels C] [--exact|--hls] "
                        "[--device D] [--repeat K] [--devices 0-7|0,2,5] [--frames F] [--split frames|rows] [--root]\n", argv[0]);
        return EXIT_FAILURE;
    }
    printf("Running full TB (%s)\n", lanczos_version());  /* main.cpp:16 */

    int width = 0, height = 0, channels = 0;
    uint8_t* img = lz_image_load(in_path, &width, &height, &channels, want_channels);
    if (img == NULL) { /* full_TB.h:110-113 */
        printf("Image was not loaded successfully.\n");
        return EXIT_FAILURE;
    }
    lanczos_desc d;
    int rc = lanczos_desc_init(&d, width, height, want_channels, 1, scale_n, scale_d, a);
    if (rc != LANCZOS_OK) { /* the reference rejects what its compiled-in parameters do not cover (full_TB.h:115-123) */
        printf("Image has wrong dimensions (%i x %i) or parameters: %s.\n", width, height, lanczos_strerror(rc));
        return EXIT_FAILURE;
    }
    d.mode = hls ? LANCZOS_MODE_HLS : (exact ? LANCZOS_MODE_EXACT : LANCZOS_MODE_LSB1);
    if (frames < 1) frames = 1;
    printf("Scale:%d/%d, WIDTHS %d -> %d\n", d.scale_n, d.scale_d, d.in_w, d.out_w); /* full_TB.h:124 */

    struct timespec t0, t1;
    uint8_t* out = NULL;
    lanczos_ctx* ctx = NULL;
    if (n_devices > 0) {
        /* several devices, still plain C: a batch of `frames` copies of the image, split by frames or by rows */
        const size_t in_fb = lanczos_in_frame_bytes(&d), out_fb = lanczos_out_frame_bytes(&d);
        uint8_t* batch_in = (uint8_t*)malloc(in_fb * frames);
        out = (uint8_t*)malloc(out_fb * frames);
        lanczos_multi* m = NULL;
        rc = lanczos_multi_create(&m, devices, n_devices);
        if (rc != LANCZOS_OK || !out || !batch_in) {
            printf("Cannot use the GPUs: %s\n", lanczos_strerror(rc));
            return EXIT_FAILURE;
        }
        for (int f = 0; f < frames; f++) memcpy(batch_in + (size_t)f * in_fb, img, in_fb);
        if (root) {
            /* the batch lives on the FIRST device; the other devices get their shares by one RCCL scatter and return them by one
             * gather (over xGMI).  Two rates, never mixed: the resample step alone and the whole call. */
            void *d_in = NULL, *d_out = NULL;
            double cms = 0, tms = 0, cms_sum = 0, tms_sum = 0;
            rc = lanczos_device_alloc(devices[0], &d_in, in_fb * frames);
            if (rc == LANCZOS_OK) rc = lanczos_device_alloc(devices[0], &d_out, out_fb * frames);
            if (rc == LANCZOS_OK) rc = lanczos_device_copy(devices[0], d_in, batch_in, in_fb * frames, 1);
            if (rc == LANCZOS_OK) rc = lanczos_resample_multi_root(m, &d, d_in, d_out, frames, split, &cms, &tms); /* warm-up */
            for (int k = 0; k < repeat && rc == LANCZOS_OK; k++) {
                rc = lanczos_resample_multi_root(m, &d, d_in, d_out, frames, split, &cms, &tms);
                cms_sum += cms;
                tms_sum += tms;
            }
            if (rc == LANCZOS_OK) rc = lanczos_device_copy(devices[0], out, d_out, out_fb * frames, 0);
            if (rc != LANCZOS_OK) {
                int he = 0, re = 0, at = 0;
                lanczos_multi_last_error(m, &he, &re, &at);
                printf("lanczos failed: %s (hip error %d, rccl error %d at message %d)\n", lanczos_strerror(rc), he, re, at);
                return EXIT_FAILURE;
            }
            const double mpix = frames * d.out_w * (double)d.out_h / 1e6;
            printf("%dx%d->%dx%d_%d|%d_%d: %d frames resident on device %d, %d device(s), split by %s:\n"
                   "  compute only            %.3f ms per batch (%.1f Mpix/s)\n"
                   "  root scatter + gather   %.3f ms per batch (%.1f Mpix/s)\n", d.in_w, d.in_h, d.out_w, d.out_h, d.scale_n,
                   d.scale_d, d.a, frames, devices[0], lanczos_multi_devices(m), split == LANCZOS_SPLIT_ROWS ? "rows" : "frames",
                   cms_sum / repeat, mpix / (cms_sum / repeat) * 1e3, tms_sum / repeat, mpix / (tms_sum / repeat) * 1e3);
            lanczos_device_free(devices[0], d_in);
            lanczos_device_free(devices[0], d_out);
            goto multi_done;
        }
        rc = lanczos_resample_multi_host(m, &d, batch_in, out, frames, split); /* warm-up: plans, staging buffers */
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int k = 0; k < repeat && rc == LANCZOS_OK; k++) rc = lanczos_resample_multi_host(m, &d, batch_in, out, frames, split);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (rc != LANCZOS_OK) {
            printf("lanczos failed: %s\n", lanczos_strerror(rc));
            return EXIT_FAILURE;
        }
        const double ms = ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) / 1e6) / repeat;
        printf("%dx%d->%dx%d_%d|%d_%d: %d frames over %d device(s), split by %s: %.3f ms per batch incl. PCIe copies "
               "(%.1f Mpix/s)\n", d.in_w, d.in_h, d.out_w, d.out_h, d.scale_n, d.scale_d, d.a, frames,
               lanczos_multi_devices(m), split == LANCZOS_SPLIT_ROWS ? "rows" : "frames", ms,
               frames * d.out_w * (double)d.out_h / ms / 1e3);
    multi_done:
        for (int f = 1; f < frames; f++)
            if (memcmp(out, out + (size_t)f * out_fb, out_fb) != 0) {
                printf("frame %d differs from frame 0\n", f);
                return EXIT_FAILURE;
            }
        lanczos_multi_destroy(m);
        free(batch_in);
    } else {
        out = (uint8_t*)malloc(lanczos_out_frame_bytes(&d));
        rc = lanczos_create(&ctx, device);
        if (rc != LANCZOS_OK || !out) {
            printf("Cannot use the GPU: %s\n", lanczos_strerror(rc));
            return EXIT_FAILURE;
        }
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int k = 0; k < repeat && rc == LANCZOS_OK; k++) rc = lanczos_resample_host(ctx, &d, img, out, 1);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (rc != LANCZOS_OK) {
            printf("lanczos failed: %s (hip error %d)\n", lanczos_strerror(rc), lanczos_last_hip_error(ctx));
            return EXIT_FAILURE;
        }
        const double ms = ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) / 1e6) / repeat;
        printf("%dx%d->%dx%d_%d|%d_%d: %.3f ms per frame incl. PCIe copies (%.1f Mpix/s), kernel family %d\n", d.in_w,
               d.in_h, d.out_w, d.out_h, d.scale_n, d.scale_d, d.a, ms, d.out_w * (double)d.out_h / ms / 1e3,
               lanczos_last_kernel(ctx));
    }

    int ok = ends_with(out_path, ".png") ? lz_image_write_png(out_path, d.out_w, d.out_h, want_channels, out,
                                                              d.out_w * want_channels)
                                         : lz_image_write_pnm(out_path, d.out_w, d.out_h, want_channels, out,
                                                              d.out_w * want_channels);
    if (!ok) {
        printf("Could not write %s\n", out_path);
        return EXIT_FAILURE;
    }
    if (ctx) lanczos_destroy(ctx);
    lz_image_free(img);
    free(out);
    return 0;
}
