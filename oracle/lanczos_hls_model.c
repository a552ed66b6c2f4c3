/*
 * oracle/lanczos_hls_model.c -- see lanczos_hls_model.h.  TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
 * Compile with -ffp-contract=off (the product's table builder uses the same expressions under the same flag, so the
 * device's f64 chains reproduce these sums bit for bit).
 */
#define _GNU_SOURCE
#include "lanczos_hls_model.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* kernel.cpp:12-18 raw_lanczos_kernel at x = k/N, and the ROM of kernel.cpp:40-45.  hls::sinpi is exact at integers:
 * the weights at whole pixel distances are exactly 0 (the software model's sin(M_PI*x) leaves ~1e-17 there). */
double oracle_hls_rom(int k, int a, int scale_n) {
    if (k < 0) k = -k;
    if (k == 0) return 1.0;
    if (k >= a * scale_n) return 0.0;          /* ROM[LANCZOS_A*SCALE_N] = 0, kernel.cpp:44 */
    if (k % scale_n == 0) return 0.0;          /* sinpi(integer) == 0 */
    const double x = (double)k / scale_n;
    const double s1 = sin(M_PI * x);
    const double s2 = sin(M_PI * x / a);
    const double c = (double)a / (M_PI * M_PI);
    return c * s1 * s2 / (x * x);
}

/* kernel.cpp:50-59: index |output_idx*SCALE_D - input_idx*SCALE_N| */
double oracle_hls_weight(int i, int o, int a, int scale_n, int scale_d) {
    long long k = (long long)o * scale_d - (long long)i * scale_n;
    if (k < 0) k = -k;
    if (k > (long long)a * scale_n) return 0.0; /* ideal stepping: cannot happen inside the window.  A window that lags behind
                                                 * (fixed-point stepper, bp > 0) can ask past the ROM's end -- undefined in the
                                                 * reference's C simulation; here: the ROM's last entry, 0 */
    return oracle_hls_rom((int)k, a, scale_n);
}

typedef struct {
    const oracle_cfg* c;
    const void* in;
    void* out;
    int bytes;
    int y0, y1;
    int bp; /* BIT_PRECISION of the fixed-point emulation, 0 = ideal arithmetic */
} hls_job;

/* ---- fixed-point emulation (lanczos.h:74-81; ap_fixed defaults: AP_TRN quantisation = truncation towards minus infinity,
 * AP_WRAP overflow).  Everything is a multiple of 2^-bp (or 2^-2bp inside one accumulation step) of magnitude < 2^10, exact in
 * double for bp <= 20.
 *   kernel_t  = ap_fixed<8+BP, 8>    a ROM entry keeps BP fractional bits
 *   num_el_t  = ap_fixed<10+BP, 10>  accumulators and the value between the passes keep BP fractional bits, 10 integer bits */
static double fx_trn(double x, int bp) { return ldexp(floor(ldexp(x, bp)), -bp); }
static double fx_wrap10(double x, int bp) { /* two's-complement wrap of a multiple of 2^-bp into [-512, 512) */
    (void)bp;
    if (x >= -512.0 && x < 512.0) return x;
    double r = fmod(x + 512.0, 1024.0);
    if (r < 0) r += 1024.0;
    return r - 512.0;
}
double oracle_hls_weight_fx(int i, int o, int a, int scale_n, int scale_d, int bp) {
    const double w = oracle_hls_weight(i, o, a, scale_n, scale_d);
    return bp > 0 ? fx_trn(w, bp) : w; /* (kernel_t)raw_lanczos_kernel(...), kernel.cpp:42 */
}

#define HLS_MAX_TAPS 16

/* floor of the window position of every output index.  bp == 0: floor(o * D / N).  bp > 0: the reference's fixed-point stepper
 * (worker.cpp:140 ColWorkers::exec, :234 RowWorkers::exec): after output o the workers shift one input sample in when
 *     fractional_t(num_el_t(1/SCALE) * (o + 1)) < fractional_t(1/SCALE)
 * with num_el_t = ap_fixed<10+BP,10> (1/SCALE cut to BP fractional bits, AP_TRN), an exact product with the integer counter,
 * and fractional_t = ap_ufixed<BP,0> (the fractional bits, AP_WRAP) -- lanczos.h:74-82, :112. */
static int* hls_positions(int out_n, int scale_n, int scale_d, int bp) {
    int* pos = (int*)malloc(sizeof(int) * (size_t)(out_n > 0 ? out_n : 1));
    if (!pos) return NULL;
    if (bp <= 0) {
        for (int o = 0; o < out_n; o++) pos[o] = (int)(((long long)o * scale_d) / scale_n);
        return pos;
    }
    const long long mod = 1ll << bp;
    const long long Q = (long long)floor(ldexp(1.0 / ((double)scale_n / scale_d), bp));
    const long long qcmp = Q % mod;
    long long frac = 0;
    int steps = 0;
    for (int o = 0; o < out_n; o++) {
        pos[o] = steps;
        frac = (frac + Q) % mod;
        if (frac < qcmp) steps++;
    }
    return pos;
}

/* one output row: vertical pass of every input column (ColWorkers::exec, worker.cpp:138-155 with compute()
 * worker.cpp:45-78), then the horizontal pass along it (RowWorkers::exec, worker.cpp:225-236 with compute_() :81-115 and
 * clamp_to_byte :118-130). */
#define HLS_DEFINE(T, SFX)                                                                                             \
    static void hls_rows_##SFX(const hls_job* j) {                                                                    \
        const oracle_cfg* c = j->c;                                                                                    \
        const int C = c->channels, a = c->a, taps = 2 * a, W = c->in_w, H = c->in_h, bp = j->bp;                      \
        const T* in = (const T*)j->in;                                                                                 \
        T* out = (T*)j->out;                                                                                           \
        double* vrow = (double*)malloc(sizeof(double) * (size_t)W * C);                                                \
        int* posy = hls_positions(c->out_h, c->scale_n, c->scale_d, bp);                                               \
        int* posx = hls_positions(c->out_w, c->scale_n, c->scale_d, bp);                                               \
        for (int y = j->y0; y < j->y1; y++) {                                                                          \
            /* window rows floor(y*D/N)-a+1 .. +a; rows < 0 are the zero priming (worker.cpp:176-188), rows > H-1     \
             * the saturated push (worker.cpp:147-153); with bp > 0 the window position comes from the fixed-point stepper */ \
            const int fy = posy[y];                                                                                    \
            double wv[HLS_MAX_TAPS];                                                                                    \
            int rr[HLS_MAX_TAPS];                                                                                       \
            for (int k = 0; k < taps; k++) {                                                                           \
                const int r = fy - a + 1 + k;                                                                          \
                wv[k] = oracle_hls_weight_fx(r, y, a, c->scale_n, c->scale_d, bp);                                     \
                rr[k] = r;                                                                                             \
            }                                                                                                          \
            for (int i = 0; i < W * C; i++) {                                                                          \
                double px[HLS_MAX_TAPS];                                                                                \
                for (int k = 0; k < taps; k++) {                                                                       \
                    const int r = rr[k];                                                                               \
                    px[k] = r < 0 ? 0.0 : (double)in[(size_t)(r > H - 1 ? H - 1 : r) * W * C + i];                    \
                }                                                                                                      \
                double acc = 0; /* kernel_t x byte: the product has BP fractional bits, nothing is cut (worker.cpp:58-64) */ \
                for (int k = 0; k < taps; k++) acc += wv[k] * px[k];                                                   \
                if (bp > 0) acc = fx_wrap10(acc, bp);                                                                  \
                const double lo = ORC_HLS_MIN(px[a - 1], px[a]), hi = ORC_HLS_MAX(px[a - 1], px[a]);                   \
                vrow[i] = acc < lo ? lo : (acc > hi ? hi : acc);           /* worker.cpp:66-74 */                      \
            }                                                                                                          \
            for (int x = 0; x < c->out_w; x++) {                                                                       \
                const int fx = posx[x];                                                                                \
                double wh[HLS_MAX_TAPS];                                                                                \
                for (int k = 0; k < taps; k++) wh[k] = oracle_hls_weight_fx(fx - a + 1 + k, x, a, c->scale_n, c->scale_d, bp); \
                for (int ch = 0; ch < C; ch++) {                                                                       \
                    double px[HLS_MAX_TAPS];                                                                            \
                    for (int k = 0; k < taps; k++) {                                                                   \
                        const int q = fx - a + 1 + k;  /* left: zeros (worker.cpp:256-265); right: last again (:244) */ \
                        px[k] = q < 0 ? 0.0 : vrow[(size_t)(q > W - 1 ? W - 1 : q) * C + ch];                          \
                    }                                                                                                  \
                    double acc = 0; /* kernel_t x num_el_t has 2 BP fractional bits; `acc +=` stores BP of them (AP_TRN), :95-101 */ \
                    if (bp > 0) {                                                                                      \
                        for (int k = 0; k < taps; k++) acc = fx_wrap10(fx_trn(acc + wh[k] * px[k], bp), bp);           \
                    } else {                                                                                           \
                        for (int k = 0; k < taps; k++) acc += wh[k] * px[k];                                           \
                    }                                                                                                  \
                    const double lo = ORC_HLS_MIN(px[a - 1], px[a]), hi = ORC_HLS_MAX(px[a - 1], px[a]);               \
                    const double v = acc < lo ? lo : (acc > hi ? hi : acc);/* worker.cpp:103-111 */                    \
                    out[((size_t)y * c->out_w + x) * C + ch] = (T)floor(v);/* worker.cpp:118-130: truncation */        \
                }                                                                                                      \
            }                                                                                                          \
        }                                                                                                              \
        free(vrow);                                                                                                    \
        free(posy);                                                                                                    \
        free(posx);                                                                                                    \
    }

#define ORC_HLS_MIN(a, b) ((a) < (b) ? (a) : (b))
#define ORC_HLS_MAX(a, b) ((a) > (b) ? (a) : (b))
HLS_DEFINE(uint8_t, u8)
HLS_DEFINE(uint16_t, u16)

static void* hls_thread(void* p) {
    const hls_job* j = (const hls_job*)p;
    if (j->bytes == 1) hls_rows_u8(j);
    else hls_rows_u16(j);
    return NULL;
}

static int hls_run(const oracle_cfg* c, const void* in, void* out, int bytes, int threads, int bp) {
    if (bp < 0 || bp > 20 || (bp > 0 && bytes != 1)) return -3;
    if (!c || !in || !out || c->in_w <= 0 || c->in_h <= 0 || c->out_w <= 0 || c->out_h <= 0 || c->channels <= 0 ||
        c->a <= 0 || 2 * c->a > HLS_MAX_TAPS || c->scale_n <= 0 || c->scale_d <= 0)
        return -1;
    if (threads < 1) threads = 1;
    if (threads > c->out_h) threads = c->out_h;
    if (threads > 256) threads = 256;
    pthread_t th[256];
    hls_job jobs[256];
    for (int t = 0; t < threads; t++) {
        jobs[t].c = c;
        jobs[t].in = in;
        jobs[t].out = out;
        jobs[t].bytes = bytes;
        jobs[t].bp = bp;
        jobs[t].y0 = (int)((long long)c->out_h * t / threads);
        jobs[t].y1 = (int)((long long)c->out_h * (t + 1) / threads);
    }
    if (threads == 1) {
        hls_thread(&jobs[0]);
        return 0;
    }
    for (int t = 0; t < threads; t++)
        if (pthread_create(&th[t], NULL, hls_thread, &jobs[t]) != 0) return -2;
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    return 0;
}

int oracle_hls_expected_hwc_u8(const oracle_cfg* cfg, const uint8_t* in, uint8_t* out, int threads) {
    return hls_run(cfg, in, out, 1, threads, 0);
}
int oracle_hls_expected_hwc_u16(const oracle_cfg* cfg, const uint16_t* in, uint16_t* out, int threads) {
    return hls_run(cfg, in, out, 2, threads, 0);
}
int oracle_hls_expected_hwc_u8_fx(const oracle_cfg* cfg, const uint8_t* in, uint8_t* out, int threads, int bit_precision) {
    return hls_run(cfg, in, out, 1, threads, bit_precision);
}
