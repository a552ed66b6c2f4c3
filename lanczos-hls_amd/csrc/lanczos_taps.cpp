// lanczos_taps.cpp -- host-side tap tables and descriptor arithmetic.  See lanczos_taps.hpp.
// Compiled with -ffp-contract=off: the doubles computed here must be the ones the reference's
// software model computes (full_TB.h:51-53), they feed the exact device paths.
#include "lanczos_taps.hpp"

#include <cmath>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace lz {

// full_TB.h:39-44
double sinc(double x) {
    if (x == 0) return 1;
    return std::sin(x) / x;
}

// full_TB.h:51-53: sinc(M_PI * x) * sinc(M_PI * x / LANCZOS_A), evaluated left to right
double kernel(double x, int a) { return sinc(M_PI * x) * sinc(M_PI * x / a); }

int gcd(int a, int b) { return b != 0 ? gcd(b, a % b) : a; }

void build_axis(int in_n, int out_n, int scale_n, int scale_d, int a, AxisTaps* t) {
    t->in_n = in_n;
    t->out_n = out_n;
    t->a = a;
    t->first.assign(out_n, 0);
    t->w.assign((size_t)out_n * 2 * a, 0.0);
    const double SCALE = (double)scale_n / scale_d;  // lanczos.h:112
    for (int o = 0; o < out_n; o++) {
        const double x = (double)o / SCALE;           // full_TB.h:57 / :70
        const int first = (int)(std::floor(x) - a + 1);
        t->first[o] = first;
        for (int k = 0; k < 2 * a; k++) {
            const int i = first + k;
            // taps outside [0, in_n-1] are dropped by the loop bounds of full_TB.h:59/:72;
            // a zero weight contributes an exact +-0 to the sum, which is the same thing
            t->w[(size_t)o * 2 * a + k] = (i >= 0 && i <= in_n - 1) ? kernel(x - i, a) : 0.0;
        }
    }
}

// kernel.cpp:12-18 at x = k/N, tabulated like init_lanczos_kernel (kernel.cpp:40-45).  Same expressions, same order and
// the same -ffp-contract=off as oracle/lanczos_hls_model.c (the checker), so the device's f64 chains match it bit for bit.
double hls_rom(int k, int a, int scale_n) {
    if (k < 0) k = -k;
    if (k == 0) return 1.0;
    if (k >= a * scale_n) return 0.0;  // ROM[LANCZOS_A*SCALE_N] = 0 (kernel.cpp:44)
    if (k % scale_n == 0) return 0.0;  // hls::sinpi of an integer is exactly 0
    const double x = (double)k / scale_n;
    const double s1 = std::sin(M_PI * x);
    const double s2 = std::sin(M_PI * x / a);
    const double c = (double)a / (M_PI * M_PI);
    return c * s1 * s2 / (x * x);
}

void build_axis_hls(int in_n, int out_n, int scale_n, int scale_d, int a, AxisTaps* t, int bit_precision) {
    t->in_n = in_n;
    t->out_n = out_n;
    t->a = a;
    t->first.assign(out_n, 0);
    t->w.assign((size_t)out_n * 2 * a, 0.0);
    // Where the 2a-row window of output o starts.  Ideal arithmetic: floor(o * D / N) - a + 1.  With BIT_PRECISION > 0 the
    // reference's own stepper decides (worker.cpp:140, :234): after every output the workers evaluate
    //     fractional_t(num_el_t(1/SCALE) * (pos + 1)) < fractional_t(1/SCALE)
    // and shift one input sample in when it holds.  num_el_t = ap_fixed<10+BP,10> cuts 1/SCALE to BP fractional bits (AP_TRN),
    // the product with the integer counter is exact, fractional_t = ap_ufixed<BP,0> keeps its fractional bits (AP_WRAP) -- so
    // the window lags wherever Q / 2^BP = trunc(1/SCALE) has drifted below o / SCALE by a whole sample (S = 3: from the first
    // step on; exact for S = 2, 4; S = 1: fractional_t(1.0) wraps to 0 and the window never moves -- the literal behaviour).
    long long Q = 0, qcmp = 0, frac = 0;
    const long long mod = 1ll << (bit_precision > 0 ? bit_precision : 0);
    if (bit_precision > 0) {
        Q = (long long)std::floor(std::ldexp(1.0 / ((double)scale_n / scale_d), bit_precision));  // num_el_t(1/SCALE), lanczos.h:112
        qcmp = Q % mod;                                                                          // fractional_t(1/SCALE)
    }
    int steps = 0;
    for (int o = 0; o < out_n; o++) {
        const int first = (bit_precision > 0 ? steps : (int)(((long long)o * scale_d) / scale_n)) - a + 1;
        if (bit_precision > 0) {
            frac = (frac + Q) % mod;       // fractional bits of num_el_t(1/SCALE) * (o + 1)
            if (frac < qcmp) steps++;      // step_cond
        }
        t->first[o] = first;
        for (int k = 0; k < 2 * a; k++) {
            long long idx = (long long)o * scale_d - (long long)(first + k) * scale_n;  // kernel.cpp:56
            if (idx < 0) idx = -idx;
            // (a lagging window can ask for an entry past the ROM's last one, ROM[a * N] = 0 -- undefined in the reference's C
            // simulation; here it reads as that last entry)
            double w = idx > (long long)a * scale_n ? 0.0 : hls_rom((int)idx, a, scale_n);
            if (bit_precision > 0) w = std::ldexp(std::floor(std::ldexp(w, bit_precision)), -bit_precision);  // (kernel_t)..., kernel.cpp:42
            t->w[(size_t)o * 2 * a + k] = w;
        }
    }
}

PrefixInfo prefix_info(const AxisTaps& v) {
    PrefixInfo p;
    const int taps = 2 * v.a;
    auto last_tap = [&](int o) {
        int last = v.first[o] + taps - 1;
        return last > v.in_n - 1 ? v.in_n - 1 : last;
    };
    for (int o = 0; o < v.out_n; o++)
        if (last_tap(o) > o) p.K = o + 1;
    p.M = p.K;
    for (int o = 0; o < p.K; o++)
        if (last_tap(o) + 1 > p.M) p.M = last_tap(o) + 1;
    p.M2 = 0;
    for (int o = 0; o < p.M && o < v.out_n; o++)
        if (last_tap(o) + 1 > p.M2) p.M2 = last_tap(o) + 1;
    return p;
}

int validate(const lanczos_desc* d) {
    if (!d) return LANCZOS_ERR_BAD_ARG;
    if (d->in_w <= 0 || d->in_h <= 0 || d->out_w <= 0 || d->out_h <= 0) return LANCZOS_ERR_BAD_ARG;
    if (d->channels != 1 && d->channels != 3 && d->channels != 4) return LANCZOS_ERR_BAD_ARG;
    if (d->bytes_per_sample != 1 && d->bytes_per_sample != 2) return LANCZOS_ERR_BAD_ARG;
    if (d->a < 2 || d->a > kMaxA) return LANCZOS_ERR_BAD_ARG;
    if (d->scale_n <= 0 || d->scale_d <= 0) return LANCZOS_ERR_BAD_ARG;
    if (d->mode != LANCZOS_MODE_LSB1 && d->mode != LANCZOS_MODE_EXACT && d->mode != LANCZOS_MODE_HLS) return LANCZOS_ERR_BAD_ARG;
    // reserved[0] = BIT_PRECISION of the HLS mode's fixed-point emulation: 0 (ideal arithmetic) .. 20, 8-bit samples only
    if (d->reserved[0] < 0 || d->reserved[0] > 20) return LANCZOS_ERR_BAD_ARG;
    if (d->reserved[0] > 0 && (d->mode != LANCZOS_MODE_HLS || d->bytes_per_sample != 1)) return LANCZOS_ERR_BAD_ARG;
    // reserved[1..2] must be 0 (a descriptor built by hand has to be zero-initialised: later versions give these fields a meaning)
    if (d->reserved[1] != 0 || d->reserved[2] != 0) return LANCZOS_ERR_BAD_ARG;
    // the harness rejects images whose size is not the compiled-in one (full_TB.h:115-118);
    // here: the output must be the input scaled by N/D (integer division, as OUT_WIDTH = IN_WIDTH*3)
    if ((long long)d->in_w * d->scale_n / d->scale_d != d->out_w) return LANCZOS_ERR_BAD_ARG;
    if ((long long)d->in_h * d->scale_n / d->scale_d != d->out_h) return LANCZOS_ERR_BAD_ARG;
    if (d->out_w > (1 << 20) || d->out_h > (1 << 20)) return LANCZOS_ERR_BAD_ARG;
    if (d->out_row0 < 0 || d->out_rows < 0) return LANCZOS_ERR_BAD_ARG;
    if (d->out_rows == 0 && d->out_row0 != 0) return LANCZOS_ERR_BAD_ARG;
    if (d->out_row0 + d->out_rows > d->out_h) return LANCZOS_ERR_BAD_ARG;
    // S < 1: the reference itself is out of bounds there (full_TB.h:85 writes img_out[j][i] for i < IN_HEIGHT into an array of
    // OUT_HEIGHT rows).  S == 1 is well defined -- every sample sits on an integer phase and the in-place vertical pass
    // (full_TB.h:67-77) is ONE recurrence over the whole frame height per column -- and runs wherever that recurrence fits the
    // in-place-prefix kernels at any frame height (k_prefix / k_prefix_stream).
    if (d->scale_n < d->scale_d) return LANCZOS_ERR_UNSUPPORTED;
    return LANCZOS_OK;
}

double f32_chain_error_bound_ordered(const double* w, const int* order, int n, double maxv) {
    return f32_chain_error_bound_from(w, order, n, maxv, 0.5 + 1e-3);
}

double f32_chain_error_bound_from(const double* w, const int* order, int n, double maxv, double start) {
    const double u = std::ldexp(1.0, -24);  // f32 unit roundoff
    double P = start, err = 0, wq = 0;
    for (int j = 0; j < n; j++) {
        const int k = order[j];
        const double wf = (double)(float)w[k];
        P += maxv * std::fabs(wf);
        err += u * (P + err);
        wq += std::fabs(wf - w[k]) * maxv;
    }
    return 1.02 * (err + wq);  // 2 % slack
}

double f32_chain_error_bound(const double* w, int ntaps, double maxv) {
    // acc_0 = bias; acc_j = fl(wf[k_j] * v[k_j] + acc_{j-1}) with ONE rounding per fmaf:
    //   |acc_j - (wf*v + acc_{j-1})| <= u * |wf*v + acc_{j-1}| <= u * P_j,   P_j = |bias| + maxv * sum_{i<=j} |wf[k_i]| (+ earlier errors)
    // plus the weights themselves: |wf - w| * maxv per tap, taken exactly.
    const double u = std::ldexp(1.0, -24);  // f32 unit roundoff
    double P = 0.5 + 1e-3, err = 0, wq = 0;
    for (int j = 0; j < ntaps; j++) {
        const int k = f32_tap_order(j, ntaps);
        const double wf = (double)(float)w[k];
        P += maxv * std::fabs(wf);
        err += u * (P + err);
        wq += std::fabs(wf - w[k]) * maxv;
    }
    return 1.02 * (err + wq);  // 2 % slack
}

bool split_chain_prepare(const double* w, const int* order, int n, double maxs, SplitChain* sc) {
    // Largest q for which every prefix of the hi chain is an integer multiple of 2^-q below 2^(24-q): with samples in [0, maxs]
    // a prefix lies between -(sum of the negative wh so far) * maxs and +(sum of the positive ones) * maxs.
    const double u = std::ldexp(1.0, -24);
    for (int q = 12; q >= 3; q--) {
        const double sc2 = std::ldexp(1.0, q);
        double wh[kMaxTaps], pos = 0, neg = 0;
        bool ok = true;
        for (int j = 0; j < n && ok; j++) {
            const int k = order[j];
            wh[k] = std::nearbyint(w[k] * sc2) / sc2;
            (wh[k] > 0 ? pos : neg) += std::fabs(wh[k]);
            ok = (pos > neg ? pos : neg) * maxs * sc2 <= 16777216.0;
        }
        if (!ok) continue;
        double wl[kMaxTaps], sum_wl = 0;
        for (int k = 0; k < kMaxTaps; k++) sc->wh[k] = sc->wl[k] = 0.0f, wl[k] = 0.0;
        for (int j = 0; j < n; j++) {
            const int k = order[j];
            wl[k] = w[k] - wh[k];                       // exact: both are doubles of the same magnitude class, |wl| <= 2^-(q+1)
            sc->wh[k] = (float)wh[k];                   // exact: <= 13 significant bits
            sc->wl[k] = (float)wl[k];
            sum_wl += std::fabs((double)sc->wl[k]);
        }
        // the lo chain starts at fl(fract(hi) + eps) -- below 1.01, one rounding of its own -- and is compared with the real
        // fract(hi) + eps + sum of wl * v; weight quantisation included
        (void)sum_wl;
        const double start = 1.01;
        sc->q = q;
        sc->eps = f32_chain_error_bound_from(wl, order, n, maxs, start) + 1.02 * u * start;
        return sc->eps < 0.005;  // (the caller's eps must stay below the 0.01 that `start` allows for)
    }
    return false;
}

bool integer_phase_tight(const double* wi, int a, double maxv) {
    // Lower-bound chain (integer_phase_flip_limit): v0 can only be left through a negative tiny term n*|w| reaching half
    // the spacing below v0, which is >= 2^-54 * v0.  If the negative taps are exactly d = +2 and d = -2 with |w| < 2^-55,
    // then n <= 2*v0 gives n*|w| < 2^-54 * v0: v0 stays.
    bool ok = a >= 3;
    const int taps = 2 * a;
    const double negl = std::ldexp(1.0, -70) / maxv;  // contributes < 2^-70: cannot reach any half spacing >= 2^-54
    for (int k = 0; k < taps && ok; k++) {
        const int dist = a - 1 - k;                    // x - i of this tap
        const double w = wi[k];
        if (dist == 2 || dist == -2) {
            if (!(w < 0 && -w < std::ldexp(1.0, -55) * (1.0 - 1e-9))) ok = false;
        } else if (w < 0 && -w > negl) {
            ok = false;
        }
    }
    return ok;
}

bool integer_phase_tight2(const double* wi, int a, int vlim) {
    // Preconditions of the second-stage filter (integer_phase_tight already holds: negative taps only at +-2, < 2^-55).
    // The double chain of an integer phase is  s = (((t(-a+1..-3)) + n(-2) w(2)) + n(-1) w(1)) + v0 + n(+1) w(-1) + n(+2) w(-2) + ...
    // with every rounding monotone.  With l2 = |w(+-2)|, l1 = w(+-1) = 2 l2 (1 + d), hb(v0) = half the spacing below v0:
    //   A-safe : n(-2) <= 2 n(-1) + 3 v0   =>  the tiny sum in front of v0 is >= -hb         => the sum is >= v0 after v0
    //   B-safe1: n(+2) <= 3 v0             =>  n(+2) l2 <= hb                                 => it cannot pull a sum >= v0 below v0
    //   B-safe2: n(+2) + 4 v0 + 2 <= 2 n(+1)  =>  n(+2) l2 <= n(+1) l1 - ulp(v0)/2 + hb       => the same after n(+1)'s push
    // (a tie lands on v0, whose mantissa is even).  A flip therefore needs  not A-safe  or  (not B-safe1 and not B-safe2).
    // Needed: both +-2 taps equal, both +-1 taps equal and positive, |l1 / l2 - 2| < 1e-6 (it is ~1e-16), every other tap
    // (|d| >= 3) non-negative or negligible, and hb(v0) >= 3.4 v0 l2 for all v0 <= vlim (it is 3.44 v0 l2 at powers of two).
    if (a < 3 || vlim < 1 || vlim > 85) return false;  // 3 v0 and 4 v0 + 2 n must fit the 16-bit lanes comfortably
    const int taps = 2 * a;
    auto at = [&](int dist) { return wi[a - 1 - dist]; };  // tap of x - i = dist
    const double l2 = -at(2), l1 = at(1);
    if (!(l2 > 0 && l1 > 0) || at(-2) != at(2) || at(-1) != at(1) || at(0) != 1.0) return false;
    if (std::fabs(l1 / l2 - 2.0) > 1e-6) return false;
    for (int k = 0; k < taps; k++) {
        const int dist = a - 1 - k;
        if (dist >= -2 && dist <= 2) continue;
        if (wi[k] < 0 && -wi[k] * 255.0 > std::ldexp(1.0, -70)) return false;
    }
    for (int v0 = 1; v0 <= vlim; v0++) {
        const int e = std::ilogb((double)v0);
        const bool pow2 = (v0 & (v0 - 1)) == 0;
        const double hb = std::ldexp(1.0, pow2 ? e - 54 : e - 53);
        if (!(hb >= 3.4 * v0 * l2 * (1.0 + 1e-9))) return false;
        // B-safe2 at a power of two: n(+1)'s push is rounded down by up to ulp/2 = 2 hb, so n(+2) l2 <= n(+1) l1 - hb is
        // needed, and "4 v0" stands for hb / l2 there
        if (pow2 && !(hb <= 3.9 * v0 * l2)) return false;
    }
    return true;
}

int integer_phase_flip_limit(const double* wi, int a, int maxv) {
    // Lower-bound chain (rounding is monotone): drop the positive tiny terms.  The negative terms in
    // front of the centre accumulate to -nb; each negative term behind it is applied on its own.  If
    // every one of these is smaller than half the spacing below v0 the sum never leaves v0.
    double nb = 0, na = 0;
    for (int k = 0; k < a - 1; k++)
        if (wi[k] < 0) nb += -wi[k] * maxv;
    for (int k = a; k < 2 * a; k++)
        if (wi[k] < 0 && -wi[k] * maxv > na) na = -wi[k] * maxv;
    const double need = (nb > na ? nb : na) * (1.0 + 1e-9);
    int limit = 0;
    for (int v0 = 1; v0 <= maxv; v0++) {
        int e = std::ilogb((double)v0);
        const bool pow2 = (v0 & (v0 - 1)) == 0;
        const double half_below = std::ldexp(1.0, pow2 ? e - 54 : e - 53);
        if (!(need < half_below)) limit = v0;
    }
    return limit;
}

}  // namespace lz
