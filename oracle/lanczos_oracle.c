/*
 * oracle/lanczos_oracle.c -- CPU restatement of the reference software path.
 * TEST INFRASTRUCTURE ONLY (see lanczos_oracle.h).  Compile: gcc -O2 -ffp-contract=off.
 *
 * Every function cites the lines of /root/reference/LanczosUpscaler/full_TB.h it follows.
 * The arithmetic is deliberately kept in the reference's shape: double accumulation, a
 * separate multiply and add per tap (no FMA), libm sin() on M_PI * x, ascending tap order,
 * clamp-then-truncate stores, a truncated integer intermediate between the passes and an
 * in-place, bottom-to-top vertical pass.
 */
#define _GNU_SOURCE
#include "lanczos_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* lanczos.h:63-64 */
#define ORC_MIN(a, b) ((a) < (b) ? (a) : (b))
#define ORC_MAX(a, b) ((a) > (b) ? (a) : (b))

/* full_TB.h:39-44 */
static double orc_sinc(double x) {
    if (x == 0) {
        return 1;
    }
    return sin(x) / x;
}

/* full_TB.h:51-53 -- note the evaluation order (M_PI * x) / a */
double oracle_lanczos_kernel(double x, int a) {
    return orc_sinc(M_PI * x) * orc_sinc(M_PI * x / a);
}

/* full_TB.h:29-37 (UINT8_MAX clamp, then truncating cast) */
uint8_t oracle_double_to_uint8(double x) {
    if (x > 255) {
        return 255;
    } else if (x < 0) {
        return 0;
    } else {
        return (uint8_t)x;
    }
}

/* the same conversion for the 16-bit generalisation (not in the reference) */
static uint16_t orc_double_to_uint16(double x) {
    if (x > 65535) {
        return 65535;
    } else if (x < 0) {
        return 0;
    } else {
        return (uint16_t)x;
    }
}

/* lanczos.h:112  SCALE = ((double)SCALE_N/SCALE_D) */
static double orc_scale(const oracle_cfg* c) { return (double)c->scale_n / c->scale_d; }

static int orc_check(const oracle_cfg* c) {
    if (!c) return -1;
    if (c->in_w <= 0 || c->in_h <= 0 || c->out_w <= 0 || c->out_h <= 0) return -1;
    if (c->channels <= 0 || c->a <= 0 || c->scale_n <= 0 || c->scale_d <= 0) return -1;
    if (c->out_h < c->in_h) return -1; /* the H pass writes rows 0..IN_H-1 of the output plane */
    return 0;
}

#define ORC_DEFINE(T, SFX, CONV)                                                                   \
    /* full_TB.h:55-65 lanczos_interpolate_row; `stride` generalises the planar unit stride so  \
     * the same loop runs on one channel of an interleaved row. */                               \
    static void orc_row_##SFX(const oracle_cfg* c, const T* in, T* out, int stride) {             \
        const double SCALE = orc_scale(c);                                                         \
        for (int xx = 0; xx < c->out_w; xx++) {                                                    \
            double x = (double)xx / SCALE;                                                         \
            double sum = 0;                                                                        \
            for (int i = ORC_MAX(0, floor(x) - c->a + 1); i <= ORC_MIN(c->in_w - 1, floor(x) + c->a); \
                 i++) {                                                                            \
                sum += in[(size_t)i * stride] * oracle_lanczos_kernel(x - i, c->a);               \
            }                                                                                      \
            out[(size_t)xx * stride] = CONV(sum);                                                  \
        }                                                                                          \
    }                                                                                              \
    /* full_TB.h:67-77 lanczos_interpolate_col: IN PLACE, xx descending; `img` points at        \
     * (row 0, this column/channel), rows are `pitch` elements apart. */                         \
    static void orc_col_##SFX(const oracle_cfg* c, T* img, size_t pitch) {                        \
        const double SCALE = orc_scale(c);                                                         \
        for (int xx = c->out_h - 1; xx >= 0; xx--) {                                               \
            double x = (double)xx / SCALE;                                                         \
            double sum = 0;                                                                        \
            for (int i = ORC_MAX(0, floor(x) - c->a + 1); i <= ORC_MIN(c->in_h - 1, floor(x) + c->a); \
                 i++) {                                                                            \
                sum += img[(size_t)i * pitch] * oracle_lanczos_kernel(x - i, c->a);               \
            }                                                                                      \
            img[(size_t)xx * pitch] = CONV(sum);                                                   \
        }                                                                                          \
    }

ORC_DEFINE(uint8_t, u8, oracle_double_to_uint8)
ORC_DEFINE(uint16_t, u16, orc_double_to_uint16)

/* ---- threading: rows of the H pass and columns of the V pass are independent ---- */
typedef struct {
    const oracle_cfg* c;
    const void* in;
    void* out;
    int bytes;   /* 1 or 2 */
    int planar;  /* 1: [C][H][W], 0: [H][W][C] */
    int phase;   /* 0: H pass, 1: V pass */
    int lo, hi;  /* unit range: H pass = (row, channel) pairs; V pass = (col, channel) pairs */
} orc_job;

static void orc_run_range(const orc_job* j) {
    const oracle_cfg* c = j->c;
    const int C = c->channels;
    for (int u = j->lo; u < j->hi; u++) {
        if (j->phase == 0) {
            /* full_TB.h:83-87: for each input row i, channel ch */
            int i = u / C, ch = u % C;
            if (j->planar) {
                size_t ioff = ((size_t)ch * c->in_h + i) * c->in_w;
                size_t ooff = ((size_t)ch * c->out_h + i) * c->out_w;
                if (j->bytes == 1)
                    orc_row_u8(c, (const uint8_t*)j->in + ioff, (uint8_t*)j->out + ooff, 1);
                else
                    orc_row_u16(c, (const uint16_t*)j->in + ioff, (uint16_t*)j->out + ooff, 1);
            } else {
                size_t ioff = (size_t)i * c->in_w * C + ch;
                size_t ooff = (size_t)i * c->out_w * C + ch;
                if (j->bytes == 1)
                    orc_row_u8(c, (const uint8_t*)j->in + ioff, (uint8_t*)j->out + ooff, C);
                else
                    orc_row_u16(c, (const uint16_t*)j->in + ioff, (uint16_t*)j->out + ooff, C);
            }
        } else {
            /* full_TB.h:89-93: for each output column i1, channel ch */
            int col = u / C, ch = u % C;
            if (j->planar) {
                size_t off = (size_t)ch * c->out_h * c->out_w + col;
                if (j->bytes == 1)
                    orc_col_u8(c, (uint8_t*)j->out + off, (size_t)c->out_w);
                else
                    orc_col_u16(c, (uint16_t*)j->out + off, (size_t)c->out_w);
            } else {
                size_t off = (size_t)col * C + ch;
                if (j->bytes == 1)
                    orc_col_u8(c, (uint8_t*)j->out + off, (size_t)c->out_w * C);
                else
                    orc_col_u16(c, (uint16_t*)j->out + off, (size_t)c->out_w * C);
            }
        }
    }
}

static void* orc_thread(void* p) {
    orc_run_range((const orc_job*)p);
    return NULL;
}

static int orc_run(const oracle_cfg* c, const void* in, void* out, int bytes, int planar, int threads) {
    if (orc_check(c) || !in || !out) return -1;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    for (int phase = 0; phase < 2; phase++) {
        int units = (phase == 0 ? c->in_h : c->out_w) * c->channels;
        if (threads == 1) {
            orc_job j = {c, in, out, bytes, planar, phase, 0, units};
            orc_run_range(&j);
            continue;
        }
        pthread_t tid[256];
        orc_job jobs[256];
        int started = 0;
        for (int t = 0; t < threads; t++) {
            int lo = (int)((long long)units * t / threads);
            int hi = (int)((long long)units * (t + 1) / threads);
            jobs[t] = (orc_job){c, in, out, bytes, planar, phase, lo, hi};
            if (pthread_create(&tid[t], NULL, orc_thread, &jobs[t]) != 0) {
                orc_run_range(&jobs[t]); /* degrade to inline execution */
                tid[t] = 0;
            } else {
                started |= 1;
            }
        }
        for (int t = 0; t < threads; t++)
            if (tid[t]) pthread_join(tid[t], NULL);
        (void)started;
    }
    return 0;
}

/* full_TB.h:79-96 */
int oracle_expected_planar_u8(const oracle_cfg* c, const uint8_t* in, uint8_t* out, int threads) {
    if (orc_check(c) || !out) return -1;
    /* img_out_ex has static storage (full_TB.h:21): rows >= IN_H start as zero */
    memset(out, 0, (size_t)c->channels * c->out_h * c->out_w);
    return orc_run(c, in, out, 1, 1, threads);
}

int oracle_expected_hwc_u8(const oracle_cfg* c, const uint8_t* in, uint8_t* out, int threads) {
    if (orc_check(c) || !out) return -1;
    memset(out, 0, (size_t)c->channels * c->out_h * c->out_w);
    return orc_run(c, in, out, 1, 0, threads);
}

int oracle_expected_hwc_u16(const oracle_cfg* c, const uint16_t* in, uint16_t* out, int threads) {
    if (orc_check(c) || !out) return -1;
    memset(out, 0, (size_t)c->channels * c->out_h * c->out_w * 2);
    return orc_run(c, in, out, 2, 0, threads);
}

/* NOT a reference behaviour: the V pass reading only H-pass values (clean, out of place). */
int oracle_outofplace_hwc_u8(const oracle_cfg* c, const uint8_t* in, uint8_t* out) {
    if (orc_check(c) || !in || !out) return -1;
    const int C = c->channels;
    const size_t pitch = (size_t)c->out_w * C;
    uint8_t* h = (uint8_t*)calloc((size_t)c->in_h * pitch, 1);
    if (!h) return -2;
    for (int i = 0; i < c->in_h; i++)
        for (int ch = 0; ch < C; ch++)
            orc_row_u8(c, in + (size_t)i * c->in_w * C + ch, h + (size_t)i * pitch + ch, C);
    const double SCALE = orc_scale(c);
    for (int xx = 0; xx < c->out_h; xx++) {
        double x = (double)xx / SCALE;
        for (size_t col = 0; col < pitch; col++) {
            double sum = 0;
            for (int i = ORC_MAX(0, floor(x) - c->a + 1); i <= ORC_MIN(c->in_h - 1, floor(x) + c->a); i++)
                sum += h[(size_t)i * pitch + col] * oracle_lanczos_kernel(x - i, c->a);
            out[(size_t)xx * pitch + col] = oracle_double_to_uint8(sum);
        }
    }
    free(h);
    return 0;
}

/* Rows xx whose taps reach a row i > xx read an already-written output row (full_TB.h:67-77).
 * K = first xx from which that no longer happens for any later row. */
int oracle_inplace_rows(const oracle_cfg* c) {
    if (orc_check(c)) return -1;
    const double SCALE = orc_scale(c);
    int K = 0;
    for (int xx = 0; xx < c->out_h; xx++) {
        double x = (double)xx / SCALE;
        int top = (int)ORC_MIN(c->in_h - 1, floor(x) + c->a);
        if (top > xx) K = xx + 1;
    }
    return K;
}

uint64_t oracle_fnv1a64(const void* data, size_t n) {
    const uint8_t* p = (const uint8_t*)data;
    uint64_t h = 1469598103934665603ULL;
    for (size_t i = 0; i < n; i++) {
        h ^= p[i];
        h *= 1099511628211ULL;
    }
    return h;
}

/* s = s*1664525u + 1013904223u (uint32); v = s >> 24 */
void oracle_lcg_fill_u8(uint8_t* dst, size_t n, uint32_t seed) {
    uint32_t s = seed;
    for (size_t i = 0; i < n; i++) {
        s = s * 1664525u + 1013904223u;
        dst[i] = (uint8_t)(s >> 24);
    }
}

void oracle_lcg_fill_u16(uint16_t* dst, size_t n, uint32_t seed) {
    uint32_t s = seed;
    for (size_t i = 0; i < n; i++) {
        s = s * 1664525u + 1013904223u;
        dst[i] = (uint16_t)(s >> 16);
    }
}
