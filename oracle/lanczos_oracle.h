/*
 * oracle/lanczos_oracle.h -- CPU restatement of the reference's *software* Lanczos path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under lanczos-hls_amd/ (the product) may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / the timed CPU baseline.
 *
 * What it restates (all in /root/reference/LanczosUpscaler/full_TB.h):
 *   double_to_uint8          :29-37   clamp to [0,255] then C-cast (truncate)
 *   sinc                     :39-44   x==0 ? 1 : sin(x)/x
 *   lanczos_kernel(double)   :51-53   sinc(pi*x) * sinc(pi*x/a), no |x|<a window test
 *   lanczos_interpolate_row  :55-65   H pass, truncating u8 store
 *   lanczos_interpolate_col  :67-77   V pass, IN PLACE, descending output row
 *   lanczos_expected         :79-96   H pass over rows 0..IN_H-1, then V pass per column
 * with SCALE = (double)SCALE_N/SCALE_D (lanczos.h:112) and MIN/MAX (lanczos.h:63-64).
 *
 * Parity pin: checked bit-for-bit against (a) the five FNV-1a-64 digests recorded in
 * SURVEY.md 8(c) and (b) the reference's own lines compiled by oracle/build_ref.sh
 * (tests/test_oracle.py, tests/golden/).  Build with -ffp-contract=off.
 */
#ifndef LANCZOS_ORACLE_H
#define LANCZOS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int in_w, in_h;     /* IN_WIDTH, IN_HEIGHT   */
    int out_w, out_h;   /* OUT_WIDTH, OUT_HEIGHT */
    int channels;       /* NUM_CHANNELS          */
    int a;              /* LANCZOS_A             */
    int scale_n, scale_d; /* SCALE_N, SCALE_D    */
} oracle_cfg;

/* full_TB.h:51-53 with LANCZOS_A as an argument */
double oracle_lanczos_kernel(double x, int a);
/* full_TB.h:29-37 */
uint8_t oracle_double_to_uint8(double x);

/* full_TB.h:79-96 on the reference's planar layout: in[C][IN_H][IN_W] -> out[C][OUT_H][OUT_W].
 * threads <= 1: the reference's single-threaded loop order.  threads > 1: H-pass rows and V-pass
 * columns are split across pthreads (rows/columns are independent, so results are identical). */
int oracle_expected_planar_u8(const oracle_cfg* cfg, const uint8_t* in, uint8_t* out, int threads);

/* Same computation on the stb interleaved HWC layout the harness hands over (full_TB.h:127-138
 * planar<->interleaved glue folded in). */
int oracle_expected_hwc_u8(const oracle_cfg* cfg, const uint8_t* in, uint8_t* out, int threads);

/* 16-bit generalisation (clamp 65535, truncation).  NOT expressible in the reference (byte is 8 bit,
 * full_TB.h:18,30): "parity unpinned" -- it is this same code path templated on the sample type. */
int oracle_expected_hwc_u16(const oracle_cfg* cfg, const uint16_t* in, uint16_t* out, int threads);

/* Out-of-place V pass (what a naive resampler would do) -- used by tests to show where the
 * in-place quirk (rows < K) matters.  Not a reference behaviour. */
int oracle_outofplace_hwc_u8(const oracle_cfg* cfg, const uint8_t* in, uint8_t* out);

/* First output row not affected by the in-place V pass: K = min{y : y - floor(y/S) >= a}. */
int oracle_inplace_rows(const oracle_cfg* cfg);

/* Helpers shared by tests: FNV-1a-64 and the LCG byte generator of SURVEY.md 8(c). */
uint64_t oracle_fnv1a64(const void* data, size_t n);
void oracle_lcg_fill_u8(uint8_t* dst, size_t n, uint32_t seed);
void oracle_lcg_fill_u16(uint16_t* dst, size_t n, uint32_t seed);

#ifdef __cplusplus
}
#endif
#endif
