// tests/native/split_chain_check.hip -- the split-weight H chain of the marching kernel's 16-bit instances (lanczos_march.hpp:
// SPLIT; lanczos_taps.cpp: split_chain_prepare; lanczos_fast.hpp: fast_prepare), emulated on the CPU with the constants the
// library itself hands the kernel (S = 2: the paired chain; the 3x instances keep the single chain).  Host code only (hipcc compiles it without a GPU); fmaf() is the single-rounding FMA the
// kernel's v_fma_f32 is.  Checked per (a, S) over adversarial and random 16-bit windows:
//   1. the hi chain is EXACT (every partial sum equals the double sum of its exact products),
//   2. the lo chain, started at fract(hi) + eps, ends in (t*, t* + 2 eps) for the real t* = sum - floor(hi) (the claim the
//      near-integer test rests on),
//   3. wherever the kernel would not flag the sample, its stored value is the reference's (full_TB.h:58-63: double chain,
//      separate multiply and add, ascending taps, truncating store),
//   4. the flag rate on uniform noise is what DESIGN.md quotes (well below 1 %).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include "lanczos_hip.h"
#include "lanczos_fast.hpp"
using namespace lz;

static int fails = 0;
#define EXPECT(c, ...)                                      \
    do {                                                    \
        if (!(c)) {                                         \
            if (fails < 20) { printf("FAILED line %d: %s  ", __LINE__, #c); printf(__VA_ARGS__); printf("\n"); } \
            fails++;                                        \
        }                                                   \
    } while (0)

static unsigned store16(double sum) { return sum < 0 ? 0u : (sum > 65535.0 ? 65535u : (unsigned)sum); }

struct Stats { long n = 0, flagged = 0; double max_dev = 0; };

// one computed sample: window v[0..taps-1] (taps ascending), phase ph
static void one(const FastConsts& fc, const double* wd, int a, int S, int ph, const unsigned* v, Stats* st) {
    const int taps = 2 * a;
    // ---- the kernel's arithmetic (lanczos_march.hpp, SPLIT): exact hi chain, lo chain started at fract(hi) + eps
    float ah = 0.0f;
    double hi_exact = 0;
    for (int k = 0; k < a; k++) {
        const float ps = (float)v[k] + (float)v[taps - 1 - k];
        ah = fmaf(fc.wsh[ph][k], ps, ah);
        hi_exact += (double)fc.wsh[ph][k] * (double)ps;   // exact in double (13 + 17 bits), and so are the partial sums
        EXPECT((double)ah == hi_exact, "a=%d S=%d k=%d", a, S, k);
    }
    const float ih = floorf(ah), fh = ah - ih;
    EXPECT((double)fh == hi_exact - std::floor(hi_exact), "fract(hi) exact");
    float al = fh + fc.bias_s;
    for (int k = 0; k < a; k++) al = fmaf(fc.wsl[ph][k], (float)v[k] + (float)v[taps - 1 - k], al);
    const float tt = al, jt = floorf(al), r = ih + jt;
    // (the kernel flags per unit and exempts units whose samples are all 0; per sample that is: a sum below 1 never needs the flag)
    const bool flag = (tt - jt) < fc.near2_s && r >= 1.0f;
    unsigned uv = r < 0 ? 0u : (unsigned)r;
    uv = uv < 65535u ? uv : 65535u;
    // ---- the reference (full_TB.h:58-63) and the real sum
    double sum = 0;
    long double real = 0;
    for (int k = 0; k < taps; k++) sum += (double)v[k] * wd[k], real += (long double)v[k] * (long double)wd[k];
    const long double t_star = real - (long double)ih;              // what t approximates (ih is exact)
    const long double dev = (long double)tt - t_star;               // must lie in (0, 2 eps)
    EXPECT(dev > 0 && dev < (long double)fc.near2_s, "a=%d S=%d ph=%d dev=%Lg near2=%g", a, S, ph, dev, (double)fc.near2_s);
    if ((double)dev > st->max_dev) st->max_dev = (double)dev;
    st->n++;
    if (flag) st->flagged++;
    else EXPECT(uv == store16(sum), "a=%d S=%d ph=%d kernel %u reference %u (sum %.9f)", a, S, ph, uv, store16(sum), sum);
}

int main() {
    std::mt19937_64 rng(12345);
    for (int S : {2})
        for (int a : {2, 3, 4}) {
            lanczos_desc d{};
            d.in_w = 3840, d.in_h = 2160, d.channels = 4, d.bytes_per_sample = 2, d.scale_n = S, d.scale_d = 1, d.a = a;
            d.out_w = d.in_w * S, d.out_h = d.in_h * S, d.out_rows = d.out_h, d.mode = LANCZOS_MODE_LSB1;
            if (validate(&d) != LANCZOS_OK) return 2;
            AxisTaps H, V;
            build_axis(d.in_w, d.out_w, S, 1, a, &H);
            build_axis(d.in_h, d.out_h, S, 1, a, &V);
            FastConsts fc;
            if (!fast_prepare(d, H, V, &fc) || !fc.split_ok) {
                printf("a=%d S=%d: fast_prepare refused\n", a, S);
                return 1;
            }
            const int taps = 2 * a;
            Stats st;
            for (int ph = 1; ph < S; ph++) {
                const double* wd = &H.w[(size_t)(S * 700 + ph) * taps];   // an interior index of that phase (its own doubles)
                unsigned v[kMaxTaps];
                // adversarial: every tap at 0 or 65535 (all 2^taps corner windows: the extreme partial sums of both halves)
                for (unsigned m = 0; m < (1u << taps); m++) {
                    for (int k = 0; k < taps; k++) v[k] = (m >> k) & 1 ? 65535u : 0u;
                    one(fc, wd, a, S, ph, v, &st);
                }
                // flat windows (sums a hair below / above an integer: the weights of a phase do not add up to exactly 1)
                for (unsigned c = 0; c < 65536; c += 17) {
                    for (int k = 0; k < taps; k++) v[k] = c;
                    one(fc, wd, a, S, ph, v, &st);
                }
                Stats noise;
                for (int it = 0; it < 2000000; it++) {
                    const unsigned long long x = rng(), y = rng();
                    for (int k = 0; k < taps; k++) v[k] = (unsigned)((k < 4 ? x >> (16 * k) : y >> (16 * (k - 4))) & 0xffff);
                    one(fc, wd, a, S, ph, v, &noise);
                }
                // smooth content: a ramp with small noise
                for (int it = 0; it < 500000; it++) {
                    const int base = (int)(rng() % 65000), slope = (int)(rng() % 400) - 200;
                    for (int k = 0; k < taps; k++) {
                        int x = base + slope * k + (int)(rng() % 16);
                        v[k] = (unsigned)(x < 0 ? 0 : (x > 65535 ? 65535 : x));
                    }
                    one(fc, wd, a, S, ph, v, &st);
                }
                printf("a=%d S=%d phase %d: eps %.3e (plain chain: %.3e), noise flag rate %.4f %% of %ld, largest t - t* %.3e\n", a, S, ph,
                       (double)fc.bias_s, (double)(S == 2 ? fc.bias_p : fc.bias), 100.0 * noise.flagged / noise.n, noise.n,
                       noise.max_dev > st.max_dev ? noise.max_dev : st.max_dev);
                EXPECT(noise.flagged < noise.n / 200, "flag rate");
                EXPECT(fc.bias_s < 0.1f * (S == 2 ? fc.bias_p : fc.bias), "the split chain's eps is an order of magnitude below the plain one's");
            }
        }
    printf(fails ? "split chain: %d FAILED\n" : "split chain: all cases ok\n", fails);
    return fails ? 1 : 0;
}
