// lanczos_taps.hpp -- host-side tap tables and descriptor arithmetic (no HIP, no GPU needed).
//
// The reference evaluates its weights per tap with two libm sin() calls inside the hot loop
// (full_TB.h:51-53,60,73) or, in the HLS path, from a ROM of a*SCALE_N+1 entries indexed by
// |out*SCALE_D - in*SCALE_N| (kernel.cpp:40-59).  Here the weights of one axis are tabulated once
// per (size, scale, a) on the host -- with exactly the double expressions of the software model, so
// the exact (f64) device paths reproduce its sums bit for bit -- and stay resident on the device.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/lanczos_hip.h"

namespace lz {

constexpr int kMaxA = 4;
constexpr int kMaxTaps = 2 * kMaxA;

// full_TB.h:39-44 and :51-53
double sinc(double x);
double kernel(double x, int a);

struct AxisTaps {
    int in_n = 0, out_n = 0, a = 0;
    std::vector<int32_t> first;  // [out_n]      floor(x) - a + 1 (may be negative)
    std::vector<double> w;       // [out_n][2a]  L(x - (first+k)); 0 where first+k is outside [0,in_n-1]
};

// x = (double)o / ((double)n/d)  (full_TB.h:57,70 with SCALE of lanczos.h:112)
void build_axis(int in_n, int out_n, int scale_n, int scale_d, int a, AxisTaps* t);

// LANCZOS_MODE_HLS: first = floor(o*D/N) - a + 1 (exact integer stepping), weights from the ROM of kernel.cpp:40-59,
// ROM[k] = a/pi^2 * sinpi(k/N) * sinpi(k/(aN)) / (k/N)^2 at k = |o*D - i*N| (1 at k = 0, exactly 0 at whole-pixel
// distances and at k = a*N).  NO zeroing of out-of-range taps: the HLS borders substitute samples, not weights.
double hls_rom(int k, int a, int scale_n);
// bit_precision > 0: ROM entries truncated to that many fractional bits (kernel_t = ap_fixed<8+BP,8>, lanczos.h:80, AP_TRN)
void build_axis_hls(int in_n, int out_n, int scale_n, int scale_d, int a, AxisTaps* t, int bit_precision = 0);

struct PrefixInfo {
    int K = 0;   // output rows [0,K) read rows i > xx, i.e. already-written OUTPUT rows (full_TB.h:67-77)
    int M = 0;   // those reads reach output rows < M  (M >= K)
    int M2 = 0;  // rows [K,M) are ordinary rows; they read H-pass rows < M2
};
PrefixInfo prefix_info(const AxisTaps& v);

int validate(const lanczos_desc* d);
int gcd(int a, int b);  // stb.cpp:9-12 (the reference reduces SCALE_N/SCALE_D with it, lanczos.h:110)

// f32 error bound of an n-tap fmaf chain against the exact sum, for samples <= maxv and the given
// weights: used as the half-width of the "too close to an integer to trust f32" window.
// Order in which the f32 chains of the specialised kernels add their taps: from the outside in
// (0, n-1, 1, n-2, ...), i.e. roughly ascending |weight|, so that the partial sums -- and with them the rounding
// error of every fmaf -- stay small until the last two steps.
constexpr int f32_tap_order(int step, int ntaps) { return (step & 1) ? ntaps - 1 - step / 2 : step / 2; }
// Rigorous bound on |f32 chain - real sum| for samples in [0,maxv], taps added in f32_tap_order, start value |bias| <= 0.5
double f32_chain_error_bound(const double* w, int ntaps, double maxv);
// The same bound for a chain that adds w[order[0]], w[order[1]], ... w[order[n-1]] in that order (n taps used)
double f32_chain_error_bound_ordered(const double* w, const int* order, int n, double maxv);
// ... with a start value of magnitude <= start instead of 0.5
double f32_chain_error_bound_from(const double* w, const int* order, int n, double maxv, double start);

// Split-weight chain for 16-bit samples, where one f32 accumulator cannot hold 16 integer bits and a useful fraction (its
// error bound, 0.013 for Lanczos-4, puts one sample in 40 on the exact-redo list).  Every weight is split w = wh + wl with wh a
// multiple of 2^-q chosen so that wh * v and ALL partial sums of the hi chain are exact in f32 (integers * 2^-q below 2^24);
// wl = w - wh (|wl| <= 2^-(q+1)) runs through an ordinary f32 chain whose magnitudes, and with them its rounding errors, are
// 2^-(q+1) of the plain chain's.  The kernel starts the lo chain at fract(hi) + eps and stores floor(hi) + floor(lo).
//   w[k], k = order[0..n-1]: the taps in the order the chain adds them; maxs: largest (pair) sample.
//   eps: rigorous bound on |lo - eps - (real sum - floor(hi))| (the per-index / mirrored-weight terms are the caller's).
struct SplitChain {
    int q = 0;
    float wh[kMaxTaps], wl[kMaxTaps];  // indexed by tap k
    double eps = 0;
};
bool split_chain_prepare(const double* w, const int* order, int n, double maxs, SplitChain* sc);

// Largest centre sample v0 for which the integer-phase double chain can still end below v0
// (SURVEY.md Q4); every v0 above it provably comes out unchanged.  wi[k] = L(a-1-k), k = 0..2a-1.
int integer_phase_flip_limit(const double* wi, int a, int maxv);
// Tight-filter precondition on the integer-phase weights wi[k] = L(a-1-k): the only non-negligible negative taps sit at
// +-2 samples and are < 2^-55 in magnitude.  Then a centre sample v0 whose +-2 neighbours are both <= 2*v0 provably
// survives the double chain (the negative excursion stays below half the spacing under v0).
bool integer_phase_tight(const double* wi, int a, double maxv);
bool integer_phase_tight2(const double* wi, int a, int vlim);

}  // namespace lz
