// lanczos_api.hip -- the extern "C" boundary (include/lanczos_hip.h) over the HIP kernels.
//
// Replaces, for the resample path only: lanczos() (lanczos.cpp:86-98) with its strip scheduler
// process_channel (lanczos.cpp:68-83), the Col/Row workers (worker.cpp:134-284), the weight ROM
// (kernel.cpp:40-67) and the cyclic line buffer (cyclic_buffer.h).  Here: host-tabulated taps that stay
// resident on the device, one fused H+V kernel launch per batch of frames, and a tiny launch IN FRONT of it (same
// stream) for the in-place prefix rows of large batches; small batches carry those rows inside the main launch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/lanczos_hip.h"
#include "lanczos_env.hpp"
#include "lanczos_fast.hpp"
#include "lanczos_layout.hpp"
#include "lanczos_march.hpp"
#include "lanczos_generic.hpp"
#include "lanczos_hls.hpp"
#include "lanczos_rational.hpp"
#include "lanczos_kernels_common.hpp"
#include "lanczos_taps.hpp"

namespace {

struct PlanKey {
    int in_w, in_h, out_w, out_h, channels, bps, sn, sd, a, hls;
    bool operator<(const PlanKey& o) const { return memcmp(this, &o, sizeof(PlanKey)) < 0; }
};

struct Plan {
    PlanKey key;
    std::vector<hipStream_t> streams;  // streams this plan's tables were used on (retirement)
    lz::AxisTaps H, V;
    lz::PrefixInfo prefix;
    lz::TapTables dev{};       // device copies
    void* dev_block = nullptr;  // one allocation behind `dev`
    void* host_block = nullptr; // page-locked source of the asynchronous upload (alive as long as the plan: a captured graph replays the copy)
    hipEvent_t uploaded = nullptr;       // recorded behind the upload ...
    hipStream_t upload_stream = nullptr; // ... on this stream (the stream of the plan's first call)
    lz::FastConsts fast{};      // phase weights etc. for the specialised kernels
    bool fast_ok = false;
    lz::RatHost rat;            // rational scales: per-index f32 weights, integer-phase flags (k_rat)
    lz::RatTables rat_dev{};
    lz::RatPHost ratp;          // exactly periodic rational scales: per-phase weights (k_ratp)
    lz::RatTables ratp_dev{};
    const float* ratp_w_dev = nullptr;
};

}  // namespace

struct lanczos_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::map<PlanKey, Plan*> plans;
    lz::LruOrder<PlanKey> plan_order;  // the cache is bounded (kMaxPlans): the least recently used plan is retired
    lz::RetireList retired_plans;
    std::mutex mu;
    int last_kernel = LANCZOS_KERNEL_NONE;
    int last_hip = 0;
    int force = LANCZOS_KERNEL_NONE;
    // staging for lanczos_resample_host
    void* stage_in = nullptr;
    void* stage_out = nullptr;
    size_t stage_in_bytes = 0, stage_out_bytes = 0;
    // timing: event triples (start, middle, end) around a call's kernels; ev_prefix_first[i]: the in-place prefix kernel ran
    // in FRONT of the main kernel (start..middle = prefix, middle..end = main) instead of behind it
    bool timing = false;
    std::vector<hipEvent_t> ev;
    std::vector<char> ev_prefix_first;
    int ev_used = 0;
    int launches = 0;
    double main_ms = 0, prefix_ms = 0;
#ifdef LZ_PROFILE_BITS
    void* stamp_buf = nullptr;   // diagnostic builds: per-wave cycle sums + residency census of k_march (lanczos_diag_report)
#endif
    lz::WgTabCache wg_tabs;  // k_march's workgroup tables (device copies), one per launch shape
    hipStream_t copy_in = nullptr, copy_out = nullptr;  // lanczos_resample_host pipeline
    std::vector<hipEvent_t> pipe_ev;
    // interleaved scratch frames of lanczos_resample_planar_device
    void* planar_in = nullptr;
    void* planar_out = nullptr;
    size_t planar_in_bytes = 0, planar_out_bytes = 0;
};
#ifdef LZ_PROFILE_BITS
static constexpr size_t kStampBytes = 16384 * 8 * 6 * 8;
#endif
static constexpr size_t kMaxPlans = 32;
#ifndef LZ_SPLIT_RULE
#define LZ_SPLIT_RULE 1
#endif

namespace {

#define LZ_HIP(ctx, call)                                  \
    do {                                                   \
        hipError_t e_ = (call);                            \
        if (e_ != hipSuccess) {                            \
            (ctx)->last_hip = (int)e_;                     \
            return LANCZOS_ERR_HIP;                        \
        }                                                  \
    } while (0)

void free_plan(Plan* p) {   // nothing in flight reads it any more
    if (p->dev_block) (void)hipFree(p->dev_block);
    if (p->host_block) (void)hipHostFree(p->host_block);
    if (p->uploaded) (void)hipEventDestroy(p->uploaded);
    delete p;
}

// The plan of a shape: tap tables built on the host, uploaded ONCE, asynchronously, on the stream of the first call that needs
// them (page-locked source, no device-wide wait: the entry points stay asynchronous on the caller's stream and can be captured
// into a graph in relaxed capture mode; calls on other streams wait for the upload by event).
int get_plan(lanczos_ctx* ctx, const lanczos_desc* d, hipStream_t stream, Plan** out) {
    PlanKey key;
    memset(&key, 0, sizeof(key));
    const bool hls = d->mode == LANCZOS_MODE_HLS;
    key = PlanKey{d->in_w, d->in_h, d->out_w, d->out_h, d->channels, d->bytes_per_sample,
                  d->scale_n, d->scale_d, d->a, hls ? 1 + d->reserved[0] : 0};
    auto it = ctx->plans.find(key);
    if (it != ctx->plans.end()) {
        Plan* p = it->second;
        ctx->plan_order.touch(key);
        if (stream != p->upload_stream && hipEventQuery(p->uploaded) != hipSuccess)
            LZ_HIP(ctx, hipStreamWaitEvent(stream, p->uploaded, 0));  // another stream: behind the upload
        lz::note_stream(p->streams, stream);
        *out = p;
        return LANCZOS_OK;
    }
    ctx->retired_plans.reap(false);
    if (ctx->plans.size() >= kMaxPlans && !ctx->plan_order.empty()) {
        // bounded: a caller that cycles through many shapes does not grow device memory without limit.  The least recently used
        // plan's tables are freed once the launches that read them have drained (events on the streams it was used on).
        const PlanKey old = ctx->plan_order.pop_oldest();
        auto io = ctx->plans.find(old);
        if (io != ctx->plans.end()) {
            Plan* q = io->second;
            ctx->retired_plans.retire({q->dev_block}, {q->host_block}, q->streams);
            if (q->uploaded) (void)hipEventDestroy(q->uploaded);
            delete q;
            ctx->plans.erase(io);
        }
    }
    Plan* p = new (std::nothrow) Plan();
    if (!p) return LANCZOS_ERR_NOMEM;
    p->key = key;
    if (hls) {  // ROM weights, exact stepping, no in-place prefix (lanczos_hls.hpp)
        lz::build_axis_hls(d->in_w, d->out_w, d->scale_n, d->scale_d, d->a, &p->H, d->reserved[0]);
        lz::build_axis_hls(d->in_h, d->out_h, d->scale_n, d->scale_d, d->a, &p->V, d->reserved[0]);
    } else {
        lz::build_axis(d->in_w, d->out_w, d->scale_n, d->scale_d, d->a, &p->H);
        lz::build_axis(d->in_h, d->out_h, d->scale_n, d->scale_d, d->a, &p->V);
        p->prefix = lz::prefix_info(p->V);
    }
    const int taps = 2 * d->a;
    // one device block: h_first | v_first | h_w | v_w  (8-byte aligned sections)
    size_t off_hf = 0;
    size_t off_vf = off_hf + (((size_t)d->out_w * 4 + 7) & ~(size_t)7);
    size_t off_hw = off_vf + (((size_t)d->out_h * 4 + 7) & ~(size_t)7);
    size_t off_vw = off_hw + (size_t)d->out_w * taps * 8;
    size_t off_xw = off_vw + (size_t)d->out_h * taps * 8;
    size_t total = off_xw + (size_t)lz::kFastMaxS * lz::kMaxTaps * 8;
    p->fast_ok = !hls && lz::fast_prepare(*d, p->H, p->V, &p->fast);
    if (!hls && d->scale_d != 1) lz::rat_prepare(*d, p->H, p->V, &p->rat);
    size_t off_hwf = 0, off_vwf = 0, off_hint = 0, off_vint = 0;
    if (p->rat.ok) {  // appended sections: f32 weights and integer-phase flags of both axes
        off_hwf = total;
        off_vwf = off_hwf + (size_t)d->out_w * taps * 4;
        off_hint = off_vwf + (size_t)d->out_h * taps * 4;
        off_vint = off_hint + (((size_t)d->out_w + 7) & ~(size_t)7);
        total = off_vint + (((size_t)d->out_h + 7) & ~(size_t)7);
    }
    size_t off_pw = 0;
    if (p->rat.ok && lz::ratp_has(*d)) lz::ratp_prepare(*d, p->H, p->V, p->rat, &p->ratp);
    if (p->ratp.ok) {
        off_pw = total;
        total += sizeof(p->ratp.phase_w);
    }
    hipError_t e = hipHostMalloc(&p->host_block, total, hipHostMallocDefault);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        delete p;
        return LANCZOS_ERR_HIP;
    }
    struct HostView {   // (the sections below were written against a std::vector; same two members)
        uint8_t* b;
        uint8_t* data() const { return b; }
    } host{(uint8_t*)p->host_block};
    memset(host.data(), 0, total);
    if (p->rat.ok) {
        memcpy(host.data() + off_hwf, p->rat.h_wf.data(), p->rat.h_wf.size() * 4);
        memcpy(host.data() + off_vwf, p->rat.v_wf.data(), p->rat.v_wf.size() * 4);
        memcpy(host.data() + off_hint, p->rat.h_int.data(), p->rat.h_int.size());
        memcpy(host.data() + off_vint, p->rat.v_int.data(), p->rat.v_int.size());
    }
    if (p->ratp.ok) memcpy(host.data() + off_pw, p->ratp.phase_w, sizeof(p->ratp.phase_w));
    if (p->fast_ok) {  // row 0: integer phase, row ph: phase ph
        memcpy(host.data() + off_xw, p->fast.wi, lz::kMaxTaps * 8);
        for (int ph = 1; ph < lz::kFastMaxS; ph++)
            memcpy(host.data() + off_xw + (size_t)ph * lz::kMaxTaps * 8, p->fast.wd[ph], lz::kMaxTaps * 8);
    }
    memcpy(host.data() + off_hf, p->H.first.data(), (size_t)d->out_w * 4);
    memcpy(host.data() + off_vf, p->V.first.data(), (size_t)d->out_h * 4);
    memcpy(host.data() + off_hw, p->H.w.data(), (size_t)d->out_w * taps * 8);
    memcpy(host.data() + off_vw, p->V.w.data(), (size_t)d->out_h * taps * 8);
    e = hipMalloc(&p->dev_block, total);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->uploaded, hipEventDisableTiming);
    if (e != hipSuccess) {
        ctx->last_hip = (int)e;
        free_plan(p);
        return LANCZOS_ERR_HIP;
    }
    e = hipMemcpyAsync(p->dev_block, p->host_block, total, hipMemcpyHostToDevice, stream);  // stream-ordered in front of the first launch
    if (e == hipSuccess) e = hipEventRecord(p->uploaded, stream);
    if (e != hipSuccess) {
        // the copy may have been queued: its two blocks go through the retire list, not straight back to the allocator
        ctx->last_hip = (int)e;
        ctx->retired_plans.retire({p->dev_block}, {p->host_block}, {stream});
        if (p->uploaded) (void)hipEventDestroy(p->uploaded);
        delete p;
        return LANCZOS_ERR_HIP;
    }
    p->upload_stream = stream;
    p->streams.push_back(stream);
    uint8_t* b = (uint8_t*)p->dev_block;
    p->dev.h_first = (const int32_t*)(b + off_hf);
    p->dev.v_first = (const int32_t*)(b + off_vf);
    p->dev.h_w = (const double*)(b + off_hw);
    p->dev.v_w = (const double*)(b + off_vw);
    p->dev.x_w = (const double*)(b + off_xw);
    if (p->rat.ok) {
        p->rat_dev.h_wf = (const float*)(b + off_hwf);
        p->rat_dev.v_wf = (const float*)(b + off_vwf);
        p->rat_dev.h_int = b + off_hint;
        p->rat_dev.v_int = b + off_vint;
        p->rat_dev.bias = p->rat.bias;
        p->rat_dev.vbias_rne = p->rat.vbias_rne;
        p->rat_dev.near2 = p->rat.near2;
        p->rat_dev.vlim = p->rat.vlim;
        p->rat_dev.tight = p->rat.tight;
    }
    if (p->ratp.ok) {
        p->ratp_dev = p->rat_dev;
        p->ratp_dev.bias = p->ratp.bias;
        p->ratp_dev.vbias_rne = p->ratp.vbias_rne;
        p->ratp_dev.near2 = p->ratp.near2;
        p->ratp_w_dev = (const float*)(b + off_pw);
    }
    ctx->plans[key] = p;
    ctx->plan_order.touch(key);
    *out = p;
    return LANCZOS_OK;
}

int whole_or_strip(const lanczos_desc* d, int* row0, int* rows) {
    if (d->out_rows == 0) {
        *row0 = 0;
        *rows = d->out_h;
    } else {
        *row0 = d->out_row0;
        *rows = d->out_rows;
    }
    return LANCZOS_OK;
}

void strip_input_rows(const lz::AxisTaps& V, int a, int in_h, int row0, int rows, const lz::PrefixInfo& pi,
                      int* in_row0, int* in_rows) {
    int lo = V.first[row0];
    int hi = V.first[row0 + rows - 1] + 2 * a - 1;
    if (row0 < pi.K && pi.M2 - 1 > hi) hi = pi.M2 - 1;  // the in-place prefix reads a little deeper
    if (lo < 0) lo = 0;
    if (hi > in_h - 1) hi = in_h - 1;
    *in_row0 = lo;
    *in_rows = hi - lo + 1;
}

}  // namespace

extern "C" {

const char* lanczos_version(void) {
#ifdef LZ_PROFILE_BITS
    return "lanczos-hls_amd 0.1 (gfx950) +profile-bits";   // diagnostic build: ablation bits, stamps (scripts/ablate*.sh check for this)
#else
    return "lanczos-hls_amd 0.1 (gfx950)";
#endif
}

const char* lanczos_strerror(int code) {
    switch (code) {
        case LANCZOS_OK: return "ok";
        case LANCZOS_ERR_BAD_ARG: return "bad argument (null pointer, size, channels, a, or out != in*N/D)";
        case LANCZOS_ERR_UNSUPPORTED: return "unsupported configuration (scale < 1, or a strip that cuts the in-place prefix rows)";
        case LANCZOS_ERR_NO_DEVICE: return "no HIP device";
        case LANCZOS_ERR_HIP: return "HIP runtime error";
        case LANCZOS_ERR_NOMEM: return "out of memory";
        case LANCZOS_ERR_RCCL: return "RCCL call failed (lanczos_multi_last_error)";
        default: return "unknown error";
    }
}

int lanczos_desc_init(lanczos_desc* d, int in_w, int in_h, int channels, int bytes_per_sample, int scale_n,
                      int scale_d, int a) {
    if (!d) return LANCZOS_ERR_BAD_ARG;
    memset(d, 0, sizeof(*d));
    if (scale_n <= 0 || scale_d <= 0) return LANCZOS_ERR_BAD_ARG;
    const int g = lz::gcd(scale_n, scale_d);  // lanczos.h:110: the reference reduces N/D by their gcd
    d->in_w = in_w;
    d->in_h = in_h;
    d->scale_n = scale_n / g;
    d->scale_d = scale_d / g;
    d->out_w = (int)((long long)in_w * d->scale_n / d->scale_d);
    d->out_h = (int)((long long)in_h * d->scale_n / d->scale_d);
    d->channels = channels;
    d->bytes_per_sample = bytes_per_sample;
    d->a = a;
    d->mode = LANCZOS_MODE_LSB1;
    return lz::validate(d);
}

int lanczos_validate(const lanczos_desc* d) { return lz::validate(d); }

int lanczos_inplace_rows(const lanczos_desc* d) {
    if (lz::validate(d) != LANCZOS_OK) return -1;
    if (d->mode == LANCZOS_MODE_HLS) return 0;  // the HLS pipeline has no in-place pass
    lz::AxisTaps V;
    lz::build_axis(d->in_h, d->out_h, d->scale_n, d->scale_d, d->a, &V);
    return lz::prefix_info(V).K;
}

int lanczos_strip_input_rows(const lanczos_desc* d, int out_row0, int out_rows, int* in_row0, int* in_rows) {
    int rc = lz::validate(d);
    if (rc != LANCZOS_OK) return rc;
    if (!in_row0 || !in_rows || out_row0 < 0 || out_rows <= 0 || out_row0 + out_rows > d->out_h)
        return LANCZOS_ERR_BAD_ARG;
    lz::AxisTaps V;
    if (d->mode == LANCZOS_MODE_HLS) {
        lz::build_axis_hls(d->in_h, d->out_h, d->scale_n, d->scale_d, d->a, &V, d->reserved[0]);  // (the fixed-point stepper moves the window)
        strip_input_rows(V, d->a, d->in_h, out_row0, out_rows, lz::PrefixInfo(), in_row0, in_rows);
        return LANCZOS_OK;
    }
    lz::build_axis(d->in_h, d->out_h, d->scale_n, d->scale_d, d->a, &V);
    strip_input_rows(V, d->a, d->in_h, out_row0, out_rows, lz::prefix_info(V), in_row0, in_rows);
    return LANCZOS_OK;
}

size_t lanczos_in_frame_bytes(const lanczos_desc* d) {
    return d ? (size_t)d->in_w * d->in_h * d->channels * d->bytes_per_sample : 0;
}
size_t lanczos_out_frame_bytes(const lanczos_desc* d) {
    return d ? (size_t)d->out_w * d->out_h * d->channels * d->bytes_per_sample : 0;
}

double lanczos_kernel(double x, int a) { return lz::kernel(x, a); }

double lanczos_kernel_idx(int in_idx, int out_idx, int scale_n, int scale_d, int a) {
    // the software model's argument for (output index, input index): x - i with x = out / SCALE
    // (full_TB.h:57,60).  The HLS twin looks up |out*SCALE_D - in*SCALE_N| / SCALE_N (kernel.cpp:56-58),
    // the same point up to the rounding of the division.
    const double SCALE = (double)scale_n / scale_d;
    const double x = (double)out_idx / SCALE;
    return lz::kernel(x - in_idx, a);
}

int lanczos_taps_host(const lanczos_desc* d, int axis, int32_t* first, double* weights) {
    int rc = lz::validate(d);
    if (rc != LANCZOS_OK) return rc;
    if (!first || !weights || (axis != 0 && axis != 1)) return LANCZOS_ERR_BAD_ARG;
    lz::AxisTaps t;
    if (d->mode == LANCZOS_MODE_HLS)
        lz::build_axis_hls(axis == 0 ? d->in_w : d->in_h, axis == 0 ? d->out_w : d->out_h, d->scale_n, d->scale_d, d->a, &t, d->reserved[0]);
    else if (axis == 0)
        lz::build_axis(d->in_w, d->out_w, d->scale_n, d->scale_d, d->a, &t);
    else
        lz::build_axis(d->in_h, d->out_h, d->scale_n, d->scale_d, d->a, &t);
    memcpy(first, t.first.data(), t.first.size() * sizeof(int32_t));
    memcpy(weights, t.w.data(), t.w.size() * sizeof(double));
    return LANCZOS_OK;
}

int lanczos_create(lanczos_ctx** out, int device) {
    if (!out) return LANCZOS_ERR_BAD_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return LANCZOS_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return LANCZOS_ERR_NO_DEVICE;
    (void)lz::env();  // every environment switch is read here, once per process (lanczos_env.hpp)
    lanczos_ctx* ctx = new (std::nothrow) lanczos_ctx();
    if (!ctx) return LANCZOS_ERR_NOMEM;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return LANCZOS_ERR_HIP;
    }
    *out = ctx;
    return LANCZOS_OK;
}

#ifdef LZ_PROFILE_BITS
// Diagnostic builds only (make variant EXTRA=-DLZ_PROFILE_BITS): where a wave of k_march<u8,3,2,3> spends its cycles and when /
// where every workgroup ran, from the stamps of the last LANCZOS_STAMP=1 launch.  Not part of the C ABI of include/lanczos_hip.h.
static void diag_report(lanczos_ctx* ctx, FILE* out) {
    if (!ctx->stamp_buf) return;
    {
        std::vector<unsigned long long> h(kStampBytes / 8);
        if (hipMemcpy(h.data(), ctx->stamp_buf, kStampBytes, hipMemcpyDeviceToHost) == hipSuccess) {
            double sum[5] = {0, 0, 0, 0, 0}, ticks = 0;
            long n = 0;
            for (size_t i = 0; i + 5 < (size_t)16384 * 8 * 3; i += 6)
                if (h[i + 5]) {
                    for (int k = 0; k < 5; k++) sum[k] += (double)h[i + k];
                    ticks += (double)h[i + 5];
                    n++;
                }
            {   // residency census: per CU, time-averaged and peak number of co-resident workgroups
                std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
                const size_t off = (size_t)16384 * 8 * 3;
                long wgs = 0;
                unsigned long long tmin = ~0ull, tmax = 0;
                for (size_t i = off; i + 2 < h.size(); i += 3) {
                    if (!h[i + 1]) continue;
                    const unsigned hw = (unsigned)h[i + 2];
                    const unsigned long long key = ((h[i + 2] >> 32) << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
                    ev[key].push_back({h[i], +1});
                    ev[key].push_back({h[i + 1], -1});
                    if (h[i] < tmin) tmin = h[i];
                    if (h[i + 1] > tmax) tmax = h[i + 1];
                    wgs++;
                }
                double avg = 0;
                int peak = 0;
                for (auto& kv : ev) {
                    std::sort(kv.second.begin(), kv.second.end());
                    int cur = 0;
                    unsigned long long last = tmin, area = 0;
                    for (auto& e : kv.second) {
                        area += (unsigned long long)cur * (e.first - last);
                        last = e.first;
                        cur += e.second;
                        if (cur > peak) peak = cur;
                    }
                    avg += (double)area / (double)(tmax - tmin ? tmax - tmin : 1);
                }
                {   // spread of start times and lifetimes (100 MHz ticks -> us)
                    std::vector<double> st, life;
                    for (size_t i = off; i + 2 < h.size(); i += 3)
                        if (h[i + 1]) {
                            st.push_back((h[i] - tmin) / 100.0);
                            life.push_back((h[i + 1] - h[i]) / 100.0);
                        }
                    std::sort(st.begin(), st.end());
                    std::sort(life.begin(), life.end());
                    if (!st.empty())
                        fprintf(out, "CENSUS start us: p50=%.1f p90=%.1f max=%.1f | lifetime us: min=%.1f p10=%.1f p50=%.1f p90=%.1f max=%.1f\n",
                                st[st.size() / 2], st[st.size() * 9 / 10], st.back(), life.front(), life[life.size() / 10],
                                life[life.size() / 2], life[life.size() * 9 / 10], life.back());
                }
                if (!lz::env().census_dump.empty()) {  // raw records for offline correlation: wg index, start, end, xcc, hw_id
                    FILE* df = fopen(lz::env().census_dump.c_str(), "w");
                    if (df) {
                        for (size_t i = off, k = 0; i + 2 < h.size(); i += 3, k++)
                            if (h[i + 1])
                                fprintf(df, "%zu %llu %llu %u %u\n", k, h[i] - tmin, h[i + 1] - tmin, (unsigned)(h[i + 2] >> 32),
                                        (unsigned)h[i + 2]);
                        fclose(df);
                    }
                }
                if (wgs)
                    fprintf(out, "CENSUS wgs=%ld distinct_cus=%zu span=%.1f us avg_resident_wgs_per_cu=%.2f peak=%d\n", wgs,
                            ev.size(), (tmax - tmin) / 100.0, avg / ev.size(), peak);
            }
            if (n)
                fprintf(out, "STAMP waves=%ld ticks/wave=%.1f cycles/tick: issue=%.0f hpass=%.0f commit=%.0f vpass=%.0f barrier=%.0f\n",
                        n, ticks / n, sum[0] / ticks, sum[1] / ticks, sum[2] / ticks, sum[3] / ticks, sum[4] / ticks);
            for (int w = 0; w < 6; w++) {  // the same by wave index inside the workgroup (k_march<u8,3,2,3>: 6 waves; 0-2 run the H pass)
                double s6[5] = {0, 0, 0, 0, 0}, tk = 0;
                for (size_t i = (size_t)w * 6; i + 5 < (size_t)16384 * 8 * 3; i += 36)
                    if (h[i + 5]) {
                        for (int k = 0; k < 5; k++) s6[k] += (double)h[i + k];
                        tk += (double)h[i + 5];
                    }
                if (tk > 0)
                    fprintf(out, "STAMP wave %d: issue=%.0f hpass=%.0f commit=%.0f vpass=%.0f barrier=%.0f\n", w, s6[0] / tk, s6[1] / tk,
                            s6[2] / tk, s6[3] / tk, s6[4] / tk);
            }
        }
    }
}
#endif

int lanczos_destroy(lanczos_ctx* ctx) {
    if (!ctx) return LANCZOS_ERR_BAD_ARG;
    (void)hipSetDevice(ctx->device);
    // Launches of this context may still be running on the caller's streams: the device is drained before anything they read
    // is freed (the only device-wide wait of the library, at the one point where it is owed).
    (void)hipDeviceSynchronize();
    ctx->retired_plans.reap(true);
    ctx->wg_tabs.release_all();
    for (auto& kv : ctx->plans) free_plan(kv.second);
    ctx->plans.clear();
#ifdef LZ_PROFILE_BITS
    if (ctx->stamp_buf) {
        diag_report(ctx, stderr);
        (void)hipFree(ctx->stamp_buf);
    }
#endif
    for (hipEvent_t e : ctx->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->pipe_ev) (void)hipEventDestroy(e);
    if (ctx->copy_in) (void)hipStreamDestroy(ctx->copy_in);
    if (ctx->copy_out) (void)hipStreamDestroy(ctx->copy_out);
    if (ctx->planar_in) (void)hipFree(ctx->planar_in);
    if (ctx->planar_out) (void)hipFree(ctx->planar_out);
    if (ctx->stage_in) (void)hipFree(ctx->stage_in);
    if (ctx->stage_out) (void)hipFree(ctx->stage_out);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return LANCZOS_OK;
}

int lanczos_timing_enable(lanczos_ctx* ctx, int on) {
    if (!ctx) return LANCZOS_ERR_BAD_ARG;
    ctx->timing = on != 0;
    return LANCZOS_OK;
}

static int timing_flush(lanczos_ctx* ctx) {
    const int used = ctx->ev_used;
    ctx->ev_used = 0;  // whatever happens below, the next call starts from a clean slate
    for (int i = 0; i + 2 < used; i += 3) {
        float a = 0, b = 0;
        LZ_HIP(ctx, hipEventSynchronize(ctx->ev[i + 2]));
        LZ_HIP(ctx, hipEventElapsedTime(&a, ctx->ev[i], ctx->ev[i + 1]));
        LZ_HIP(ctx, hipEventElapsedTime(&b, ctx->ev[i + 1], ctx->ev[i + 2]));
        const bool prefix_first = ctx->ev_prefix_first[i / 3] != 0;  // start..middle = the prefix kernel in front of the main one
        ctx->main_ms += prefix_first ? b : a;
        ctx->prefix_ms += prefix_first ? a : b;
        ctx->launches++;
    }
    return LANCZOS_OK;
}

int lanczos_timing_read(lanczos_ctx* ctx, int* launches, double* main_kernel_ms, double* prefix_kernel_ms) {
    if (!ctx) return LANCZOS_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(ctx->mu);
    (void)hipSetDevice(ctx->device);
    int rc = timing_flush(ctx);
    if (rc != LANCZOS_OK) return rc;
    if (launches) *launches = ctx->launches;
    if (main_kernel_ms) *main_kernel_ms = ctx->main_ms;
    if (prefix_kernel_ms) *prefix_kernel_ms = ctx->prefix_ms;
    ctx->launches = 0;
    ctx->main_ms = ctx->prefix_ms = 0;
    return LANCZOS_OK;
}

int lanczos_last_kernel(const lanczos_ctx* ctx) { return ctx ? ctx->last_kernel : LANCZOS_KERNEL_NONE; }
int lanczos_last_hip_error(const lanczos_ctx* ctx) { return ctx ? ctx->last_hip : 0; }

int lanczos_force_kernel(lanczos_ctx* ctx, int family) {
    if (!ctx || family < LANCZOS_KERNEL_NONE || family > LANCZOS_KERNEL_FAST) return LANCZOS_ERR_BAD_ARG;  // (_HLS follows the mode)
    ctx->force = family;
    return LANCZOS_OK;
}

// In-place prefix rows of an integer scale as a register-only kernel (k_prefix_reg) on `stream`, in FRONT of the marching kernel
// (5 us by events of a 210 us step).  A fork / join onto a side stream so that it overlaps the march was measured and dropped
// (the two event dependencies cost 14 us per step).  hipErrorNotSupported: no instance for this (sample type, scale, a) -- the
// caller falls back to the general k_prefix behind the main kernel.
static hipError_t launch_prefix_front(const lanczos_desc* d, const lz::FrameGeom& g, const Plan* p, hipStream_t stream) {
#define LZ_PREFIX_REG_CONFIGS(X)                                                                                        \
    X(uint8_t, 2, 2) X(uint8_t, 2, 3) X(uint8_t, 2, 4) X(uint8_t, 3, 2) X(uint8_t, 3, 3) X(uint8_t, 3, 4) X(uint8_t, 4, 2) \
    X(uint8_t, 4, 3) X(uint8_t, 4, 4) X(uint16_t, 2, 3) X(uint16_t, 2, 4) X(uint16_t, 3, 3) X(uint16_t, 3, 4)
    bool have = false;
#define X(T, S, A)                                                                                                    \
    if (d->bytes_per_sample == (int)sizeof(T) && d->scale_n == S && d->a == A && p->prefix.K == lz::prefix_K(S, A) && \
        p->prefix.M == lz::prefix_M(S, A) && p->prefix.M2 == lz::prefix_M2(S, A))                                    \
        have = true;
    LZ_PREFIX_REG_CONFIGS(X)
#undef X
    if (!have) return hipErrorNotSupported;
    const int samples_w = d->out_w * d->channels;
    dim3 grid((samples_w + 127) / 128, g.frames);
#define X(T, S, A)                                                                                  \
    if (d->bytes_per_sample == (int)sizeof(T) && d->scale_n == S && d->a == A)                      \
        hipLaunchKernelGGL((lz::k_prefix_reg<T, S, A>), grid, dim3(128), 0, stream, g, p->dev);
    LZ_PREFIX_REG_CONFIGS(X)
#undef X
    return hipGetLastError();
}

// the resample proper; ctx->mu is held by the caller
static int resample_device_locked(lanczos_ctx* ctx, const lanczos_desc* d, const void* d_in, void* d_out, int frames,
                                  size_t in_frame_stride, size_t out_frame_stride, void* stream_v, bool in_split = false) {
    int rc;
    LZ_HIP(ctx, hipSetDevice(ctx->device));
    // NULL is the NULL (legacy default) stream -- NOT the context's private stream: a caller whose producers run
    // on the default stream (torch's default stream is handle 0) must be ordered behind them
    hipStream_t stream = (hipStream_t)stream_v;
    Plan* p = nullptr;
    rc = get_plan(ctx, d, stream, &p);
    if (rc != LANCZOS_OK) return rc;

    int row0, rows;
    whole_or_strip(d, &row0, &rows);
    int in_row0, in_rows;
    strip_input_rows(p->V, d->a, d->in_h, row0, rows, p->prefix, &in_row0, &in_rows);
    const bool has_prefix = row0 < p->prefix.K;
    if (has_prefix) {
        // the prefix recurrence needs rows [0,M) of the output and [0,M2) of the H pass in one place
        if (row0 != 0) return LANCZOS_ERR_UNSUPPORTED;
        // (any depth runs: prefixes too deep for k_prefix's row arrays take the streaming form of the recurrence, k_prefix_stream)
    }

    lz::FrameGeom g{};
    g.in = (const uint8_t*)d_in;
    g.out = (uint8_t*)d_out;
    g.in_pitch = d->in_w * d->channels * d->bytes_per_sample;
    g.out_pitch = d->out_w * d->channels * d->bytes_per_sample;
    g.in_frame_stride = in_frame_stride ? in_frame_stride : (size_t)g.in_pitch * in_rows;
    g.out_frame_stride = out_frame_stride ? out_frame_stride : (size_t)g.out_pitch * rows;
    g.in_w = d->in_w;
    g.in_h = d->in_h;
    g.out_w = d->out_w;
    g.out_h = d->out_h;
    g.channels = d->channels;
    g.a = d->a;
    g.in_row0 = in_row0;
    g.in_rows = in_rows;
    g.out_row0 = row0;
    g.out_rows = rows;
    g.skip_rows = has_prefix ? p->prefix.K : 0;
    g.frames = frames;
    g.hls_bp = d->mode == LANCZOS_MODE_HLS ? d->reserved[0] : 0;
    g.debug_skip = 0;
    g.stamps = nullptr;
#ifdef LZ_PROFILE_BITS
    // diagnostic builds only: ablation bits, and (LANCZOS_STAMP=1) the kernel instance that sums s_memtime deltas per phase
    // and records when / where every workgroup ran; reported by diag_report() when the context is destroyed
    g.debug_skip = lz::env().debug_skip;
    if (lz::env().stamp) {
        if (!ctx->stamp_buf) (void)hipMalloc(&ctx->stamp_buf, kStampBytes), (void)hipMemset(ctx->stamp_buf, 0, kStampBytes);
        g.stamps = (unsigned long long*)ctx->stamp_buf;
    }
#endif
    if (d->mode == LANCZOS_MODE_HLS) {
        if (ctx->force == LANCZOS_KERNEL_FAST || ctx->force == LANCZOS_KERNEL_GENERIC) return LANCZOS_ERR_UNSUPPORTED;
        if (d->channels > 4 || 2 * d->a > 2 * lz::kMaxA) return LANCZOS_ERR_UNSUPPORTED;
        if (frames > 65535) return LANCZOS_ERR_UNSUPPORTED;
    }
    bool prefix_fused = false;
    bool use_fast = p->fast_ok && ctx->force != LANCZOS_KERNEL_GENERIC &&
                    lz::fast_supports(*d, g);
    const bool use_rat = !use_fast && p->rat.ok && ctx->force != LANCZOS_KERNEL_GENERIC && lz::rat_supports(*d, g);
    if (ctx->force == LANCZOS_KERNEL_FAST && !use_fast && !use_rat) return LANCZOS_ERR_UNSUPPORTED;

    // Events are reserved only once nothing but a HIP failure can stop the call; a triple is committed (ev_used += 3)
    // after all three records succeeded, so timing_flush never meets an unrecorded event.
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    if (ctx->timing) {
        if (ctx->ev_used + 3 > 3 * 4096) {
            rc = timing_flush(ctx);
            if (rc != LANCZOS_OK) return rc;
        }
        while ((int)ctx->ev.size() < ctx->ev_used + 3) {
            hipEvent_t e;
            LZ_HIP(ctx, hipEventCreate(&e));
            ctx->ev.push_back(e);
        }
        ctx->ev_prefix_first.resize(ctx->ev.size() / 3 + 1, 0);
        ev0 = ctx->ev[ctx->ev_used];
        ev1 = ctx->ev[ctx->ev_used + 1];
        ev2 = ctx->ev[ctx->ev_used + 2];
        LZ_HIP(ctx, hipEventRecord(ev0, stream));
    }
    if (d->mode == LANCZOS_MODE_HLS) {
        const int tiles_x = (d->out_w + lz::kHlsTileW - 1) / lz::kHlsTileW;
        const int tiles_y = (rows + lz::kHlsTileH - 1) / lz::kHlsTileH;
        dim3 grid(tiles_x * tiles_y, frames);
        if (d->bytes_per_sample == 1)
            hipLaunchKernelGGL(lz::k_hls<uint8_t>, grid, dim3(lz::kHlsThreads), 0, stream, g, p->dev);
        else
            hipLaunchKernelGGL(lz::k_hls<uint16_t>, grid, dim3(lz::kHlsThreads), 0, stream, g, p->dev);
        LZ_HIP(ctx, hipGetLastError());
        ctx->last_kernel = LANCZOS_KERNEL_HLS;
        if (ev0) {  // one kernel: all of it is "main"
            LZ_HIP(ctx, hipEventRecord(ev1, stream));
            LZ_HIP(ctx, hipEventRecord(ev2, stream));
            ctx->ev_prefix_first[ctx->ev_used / 3] = 0;
            ctx->ev_used += 3;
        }
        return LANCZOS_OK;
    }
    bool prefix_front = false;  // the prefix rows went out on this stream in FRONT of the main kernel (register-only kernel)
    if (use_fast) {
        hipError_t e;
        if (lz::env().tile_kernel || !lz::march_supports(g)) {  // LANCZOS_TILE_KERNEL=1: the tile-per-workgroup kernel (A/B measurements)
            e = lz::fast_launch(*d, g, p->dev, p->fast, stream);
        } else {
            lz::FrameGeom gm = g;
            if (!in_split) {   // (with timing on, every sub-launch commits its own event triple: what is timed is what is shipped)
                // A batch of twice the kernel's preferred size or more goes out as launches of that size (two chunks per (strip,
                // frame) pair, rank-aware shares: the fastest shape this kernel has) followed by the rest.  The four-workgroups-per-CU
                // instances already split at one and a half times that size: their one-workgroup-per-slot table degrades quickly
                // (config 2, 48 frames: 400 us in one launch, 333 us as 32 + 16; 40 frames: 289 us either way), that of the
                // two-workgroups-per-CU instances does not (config 3, 40 frames: 251 us in one launch, 272 us as 24 + 16) --
                // profiles/round4y_ab_oversized_batch_split_rule.txt.
                bool dummy = false;
                e = lz::march_launch(*d, gm, p->dev, p->fast, stream, &dummy, &ctx->wg_tabs, /*query_only=*/true);
                const int pf = ctx->wg_tabs.pref_frames;
                const bool split = frames >= 2 * pf || (LZ_SPLIT_RULE && ctx->wg_tabs.wg_per_cu >= 4 && frames >= pf + pf / 2);
                if (e == hipSuccess && pf >= 8 && split) {
                    for (int f0 = 0; f0 < frames; f0 += pf) {
                        const int nf = frames - f0 < pf ? frames - f0 : pf;
                        rc = resample_device_locked(ctx, d, (const uint8_t*)d_in + (size_t)f0 * g.in_frame_stride,
                                                    (uint8_t*)d_out + (size_t)f0 * g.out_frame_stride, nf, g.in_frame_stride,
                                                    g.out_frame_stride, stream_v, /*in_split=*/true);
                        if (rc != LANCZOS_OK) return rc;
                    }
                    return LANCZOS_OK;
                }
            }
            if (has_prefix) {  // ask the marching launch to carry the prefix rows too (no separate k_prefix launch)
                gm.prefix_K = p->prefix.K;
                gm.prefix_M = p->prefix.M;
                gm.prefix_M2 = p->prefix.M2;
                e = lz::march_launch(*d, gm, p->dev, p->fast, stream, &prefix_fused, &ctx->wg_tabs, /*query_only=*/true);
                if (e == hipSuccess && !prefix_fused) {
                    // Large batch: the prefix rows do not ride (960 extra workgroups crowd the march: measured) -- the
                    // register-only kernel goes out first, on the same stream
                    gm.prefix_K = gm.prefix_M = gm.prefix_M2 = 0;
                    if (!lz::env().separate_prefix && d->scale_d == 1 && d->in_h >= p->prefix.M2) {
                        hipError_t pe = launch_prefix_front(d, g, p, stream);
                        if (pe == hipSuccess) {
                            prefix_front = true;
                            if (ev1) LZ_HIP(ctx, hipEventRecord(ev1, stream));  // start..middle = the prefix kernel
                        } else if (pe != hipErrorNotSupported) e = pe;
                    }
                }
            } else {
                e = hipSuccess;
            }
            if (e == hipSuccess) e = lz::march_launch(*d, gm, p->dev, p->fast, stream, &prefix_fused, &ctx->wg_tabs);
        }
        if (e != hipSuccess) {
            ctx->last_hip = (int)e;
            return LANCZOS_ERR_HIP;
        }
        ctx->last_kernel = LANCZOS_KERNEL_FAST;
    } else if (use_rat && p->ratp.ok && !lz::env().no_ratp) {
        hipError_t e = lz::ratp_launch(*d, g, p->dev, p->ratp_dev, p->ratp_w_dev, stream);
        if (e != hipSuccess) {
            ctx->last_hip = (int)e;
            return LANCZOS_ERR_HIP;
        }
        ctx->last_kernel = LANCZOS_KERNEL_FAST;
    } else if (use_rat) {
        const int row_bytes = d->out_w * d->channels * d->bytes_per_sample;
        const int tiles_x = (row_bytes + lz::kRatTileRowBytes - 1) / lz::kRatTileRowBytes;
        const int tiles_y = (rows + lz::kRatTileH - 1) / lz::kRatTileH;
        dim3 grid(tiles_x * tiles_y, frames);
        const bool ex = d->mode == LANCZOS_MODE_EXACT;
#define LZ_RAT(T, TAPS)                                                                                                  \
    do {                                                                                                                 \
        if (ex) hipLaunchKernelGGL((lz::k_rat<T, TAPS, true>), grid, dim3(lz::kRatThreads), 0, stream, g, p->dev, p->rat_dev);  \
        else hipLaunchKernelGGL((lz::k_rat<T, TAPS, false>), grid, dim3(lz::kRatThreads), 0, stream, g, p->dev, p->rat_dev);    \
    } while (0)
        if (d->bytes_per_sample == 1) {
            if (d->a == 2) LZ_RAT(uint8_t, 4);
            else if (d->a == 3) LZ_RAT(uint8_t, 6);
            else LZ_RAT(uint8_t, 8);
        } else {
            if (d->a == 2) LZ_RAT(uint16_t, 4);
            else if (d->a == 3) LZ_RAT(uint16_t, 6);
            else LZ_RAT(uint16_t, 8);
        }
#undef LZ_RAT
        LZ_HIP(ctx, hipGetLastError());
        ctx->last_kernel = LANCZOS_KERNEL_FAST;
    } else {
        const int samples_w = d->out_w * d->channels;
        const int tiles_x = (samples_w + lz::kGenTileW - 1) / lz::kGenTileW;
        const int tiles_y = (rows + lz::kGenTileH - 1) / lz::kGenTileH;
        dim3 grid(tiles_x * tiles_y, frames);
        if (d->bytes_per_sample == 1)
            hipLaunchKernelGGL(lz::k_generic<uint8_t>, grid, dim3(lz::kGenTileW), 0, stream, g, p->dev);
        else
            hipLaunchKernelGGL(lz::k_generic<uint16_t>, grid, dim3(lz::kGenTileW), 0, stream, g, p->dev);
        LZ_HIP(ctx, hipGetLastError());
        ctx->last_kernel = LANCZOS_KERNEL_GENERIC;
    }
    if (ev1 && !prefix_front) LZ_HIP(ctx, hipEventRecord(ev1, stream));  // start..middle = the main kernel

    if (has_prefix && !prefix_fused && !prefix_front) {
        const int samples_w = d->out_w * d->channels;
        // columns per block: the row arrays (M + M2 rows) must fit 60 KB of LDS; deep prefixes (scales close to 1) get fewer
        int bw = 128;
        while (bw > 32 && (size_t)(p->prefix.M + p->prefix.M2) * bw * d->bytes_per_sample > 60 * 1024) bw >>= 1;
        const bool streamed = (size_t)(p->prefix.M + p->prefix.M2) * bw * d->bytes_per_sample > 60 * 1024;
        if (streamed) bw = 128;   // deep prefix (S = 1: the whole frame; S -> 1): rings of 24 rows instead of M + M2 row arrays
        dim3 grid((samples_w + bw - 1) / bw, frames);
#define LZ_PREFIX_STREAM(T, TAPS) \
    hipLaunchKernelGGL((lz::k_prefix_stream<T, TAPS>), grid, dim3(128), 0, stream, g, p->dev, p->prefix.K, p->prefix.M)
        if (streamed) {
            if (d->bytes_per_sample == 1) {
                if (d->a == 2) LZ_PREFIX_STREAM(uint8_t, 4);
                else if (d->a == 3) LZ_PREFIX_STREAM(uint8_t, 6);
                else LZ_PREFIX_STREAM(uint8_t, 8);
            } else {
                if (d->a == 2) LZ_PREFIX_STREAM(uint16_t, 4);
                else if (d->a == 3) LZ_PREFIX_STREAM(uint16_t, 6);
                else LZ_PREFIX_STREAM(uint16_t, 8);
            }
        } else
#define LZ_PREFIX(T, TAPS)                                                                                       \
    hipLaunchKernelGGL((lz::k_prefix<T, TAPS>), grid, dim3(bw), (size_t)(p->prefix.M + p->prefix.M2) * bw * sizeof(T), stream, g, \
                       p->dev, p->prefix.K, p->prefix.M, p->prefix.M2)
        if (d->bytes_per_sample == 1) {
            if (d->a == 2) LZ_PREFIX(uint8_t, 4);
            else if (d->a == 3) LZ_PREFIX(uint8_t, 6);
            else LZ_PREFIX(uint8_t, 8);
        } else {
            if (d->a == 2) LZ_PREFIX(uint16_t, 4);
            else if (d->a == 3) LZ_PREFIX(uint16_t, 6);
            else LZ_PREFIX(uint16_t, 8);
        }
#undef LZ_PREFIX
#undef LZ_PREFIX_STREAM
        LZ_HIP(ctx, hipGetLastError());
    }
    if (ev2) {
        LZ_HIP(ctx, hipEventRecord(ev2, stream));
        ctx->ev_prefix_first[ctx->ev_used / 3] = prefix_front ? 1 : 0;
        ctx->ev_used += 3;
    }
    return LANCZOS_OK;
}

int lanczos_resample_device(lanczos_ctx* ctx, const lanczos_desc* d, const void* d_in, void* d_out, int frames,
                            size_t in_frame_stride, size_t out_frame_stride, void* stream_v) {
    if (!ctx || !d_in || !d_out || frames <= 0) return LANCZOS_ERR_BAD_ARG;
    int rc = lz::validate(d);
    if (rc != LANCZOS_OK) return rc;
    std::lock_guard<std::mutex> lock(ctx->mu);
    return resample_device_locked(ctx, d, d_in, d_out, frames, in_frame_stride, out_frame_stride, stream_v);
}

int lanczos_host_alloc(void** p, size_t bytes) {
    if (!p || bytes == 0) return LANCZOS_ERR_BAD_ARG;
    *p = nullptr;
    return hipHostMalloc(p, bytes, hipHostMallocDefault) == hipSuccess ? LANCZOS_OK : LANCZOS_ERR_NOMEM;
}

int lanczos_host_free(void* p) {
    if (!p) return LANCZOS_OK;
    return hipHostFree(p) == hipSuccess ? LANCZOS_OK : LANCZOS_ERR_HIP;
}

// Host buffers in, host buffers out.  Frames are pushed through in groups on three streams (copy-in, resample,
// copy-out) chained by events, so with page-locked buffers (lanczos_host_alloc, or the caller's own registered
// memory) the PCIe copies of neighbouring groups overlap each other and the kernels; with pageable memory the
// runtime stages the copies and the calls degrade to the serial order -- same results either way.
int lanczos_resample_host(lanczos_ctx* ctx, const lanczos_desc* d, const void* in, void* out, int frames) {
    if (!ctx || !in || !out || frames <= 0) return LANCZOS_ERR_BAD_ARG;
    int rc = lz::validate(d);
    if (rc != LANCZOS_OK) return rc;
    // The staging buffers, the two copy streams and the pipeline events belong to the context: the lock is held
    // for the whole call (calls on one context are serialised; use one context per host thread to overlap them).
    std::lock_guard<std::mutex> lock(ctx->mu);
    LZ_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->stream) return LANCZOS_ERR_HIP;
    Plan* p = nullptr;
    rc = get_plan(ctx, d, ctx->stream, &p);   // (the pipeline's resample calls run on the context's stream)
    if (rc != LANCZOS_OK) return rc;
    int row0, rows, in_row0, in_rows;
    whole_or_strip(d, &row0, &rows);
    strip_input_rows(p->V, d->a, d->in_h, row0, rows, p->prefix, &in_row0, &in_rows);
    const size_t in_frame = (size_t)d->in_w * d->channels * d->bytes_per_sample * in_rows;
    const size_t out_frame = (size_t)d->out_w * d->channels * d->bytes_per_sample * rows;
    const size_t in_bytes = in_frame * frames, out_bytes = out_frame * frames;
    if (ctx->stage_in_bytes < in_bytes) {
        if (ctx->stage_in) (void)hipFree(ctx->stage_in);
        ctx->stage_in = nullptr;
        ctx->stage_in_bytes = 0;
        LZ_HIP(ctx, hipMalloc(&ctx->stage_in, in_bytes));
        ctx->stage_in_bytes = in_bytes;
    }
    if (ctx->stage_out_bytes < out_bytes) {
        if (ctx->stage_out) (void)hipFree(ctx->stage_out);
        ctx->stage_out = nullptr;
        ctx->stage_out_bytes = 0;
        LZ_HIP(ctx, hipMalloc(&ctx->stage_out, out_bytes));
        ctx->stage_out_bytes = out_bytes;
    }
    if (!ctx->copy_in) LZ_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_in, hipStreamNonBlocking));
    if (!ctx->copy_out) LZ_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_out, hipStreamNonBlocking));
    // groups of at most 4 frames, at most 64 groups in flight per call
    int group = frames >= 8 ? 4 : (frames >= 2 ? (frames + 1) / 2 : 1);
    if ((frames + group - 1) / group > 64) group = (frames + 63) / 64;
    const int ngroups = (frames + group - 1) / group;
    while ((int)ctx->pipe_ev.size() < 2 * ngroups) {
        hipEvent_t e;
        LZ_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->pipe_ev.push_back(e);
    }
    // One failing step must not leave copies in flight on the caller's buffers: whatever the outcome, all three
    // streams are drained before the call returns.
    auto pipeline = [&]() -> int {
        for (int gi = 0; gi < ngroups; gi++) {
            const int f0 = gi * group, nf = (f0 + group <= frames) ? group : frames - f0;
            const uint8_t* hin = (const uint8_t*)in + (size_t)f0 * in_frame;
            uint8_t* din = (uint8_t*)ctx->stage_in + (size_t)f0 * in_frame;
            uint8_t* dout = (uint8_t*)ctx->stage_out + (size_t)f0 * out_frame;
            uint8_t* hout = (uint8_t*)out + (size_t)f0 * out_frame;
            LZ_HIP(ctx, hipMemcpyAsync(din, hin, in_frame * nf, hipMemcpyHostToDevice, ctx->copy_in));
            LZ_HIP(ctx, hipEventRecord(ctx->pipe_ev[2 * gi], ctx->copy_in));
            LZ_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->pipe_ev[2 * gi], 0));
            const int rc2 = resample_device_locked(ctx, d, din, dout, nf, 0, 0, ctx->stream);
            if (rc2 != LANCZOS_OK) return rc2;
            LZ_HIP(ctx, hipEventRecord(ctx->pipe_ev[2 * gi + 1], ctx->stream));
            LZ_HIP(ctx, hipStreamWaitEvent(ctx->copy_out, ctx->pipe_ev[2 * gi + 1], 0));
            LZ_HIP(ctx, hipMemcpyAsync(hout, dout, out_frame * nf, hipMemcpyDeviceToHost, ctx->copy_out));
        }
        return LANCZOS_OK;
    };
    rc = pipeline();
    const hipError_t e_in = hipStreamSynchronize(ctx->copy_in);
    const hipError_t e_k = hipStreamSynchronize(ctx->stream);
    const hipError_t e_out = hipStreamSynchronize(ctx->copy_out);
    if (rc != LANCZOS_OK) return rc;
    for (hipError_t e : {e_in, e_k, e_out})
        if (e != hipSuccess) {
            ctx->last_hip = (int)e;
            return LANCZOS_ERR_HIP;
        }
    return LANCZOS_OK;
}

// ---- planar <-> interleaved (full_TB.h:127-138, 146-165) --------------------------------------------------------
static int layout_call(lanczos_ctx* ctx, bool to_interleaved, const void* src, void* dst, int w, int h, int channels,
                       int bytes_per_sample, int frames, void* stream) {
    if (!ctx || !src || !dst || w <= 0 || h <= 0 || frames <= 0) return LANCZOS_ERR_BAD_ARG;
    if ((channels != 1 && channels != 3 && channels != 4) || (bytes_per_sample != 1 && bytes_per_sample != 2))
        return LANCZOS_ERR_BAD_ARG;
    if (h > 65535 || frames > 65535) return LANCZOS_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(ctx->mu);
    LZ_HIP(ctx, hipSetDevice(ctx->device));
    LZ_HIP(ctx, lz::layout_launch(to_interleaved, src, dst, w, h, channels, bytes_per_sample, frames, (hipStream_t)stream));
    return LANCZOS_OK;
}

int lanczos_planar_to_interleaved_device(lanczos_ctx* ctx, const void* d_planar, void* d_interleaved, int w, int h,
                                         int channels, int bytes_per_sample, int frames, void* stream) {
    return layout_call(ctx, true, d_planar, d_interleaved, w, h, channels, bytes_per_sample, frames, stream);
}

int lanczos_interleaved_to_planar_device(lanczos_ctx* ctx, const void* d_interleaved, void* d_planar, int w, int h,
                                         int channels, int bytes_per_sample, int frames, void* stream) {
    return layout_call(ctx, false, d_interleaved, d_planar, w, h, channels, bytes_per_sample, frames, stream);
}

// img_in[C][IN_H][IN_W] -> img_out[C][OUT_H][OUT_W], the arrays lanczos_expected() works on (full_TB.h:20-21,79-96).
// Whole frames only.  The interleaved scratch frames belong to the context: calls on one context must follow each
// other in stream order (one stream at a time).
int lanczos_resample_planar_device(lanczos_ctx* ctx, const lanczos_desc* d, const void* d_in_planar, void* d_out_planar,
                                   int frames, void* stream) {
    if (!ctx || !d_in_planar || !d_out_planar || frames <= 0) return LANCZOS_ERR_BAD_ARG;
    int rc = lz::validate(d);
    if (rc != LANCZOS_OK) return rc;
    if (d->out_rows != 0 && !(d->out_row0 == 0 && d->out_rows == d->out_h)) return LANCZOS_ERR_UNSUPPORTED;
    const size_t in_bytes = lanczos_in_frame_bytes(d) * frames, out_bytes = lanczos_out_frame_bytes(d) * frames;
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        LZ_HIP(ctx, hipSetDevice(ctx->device));
        if (ctx->planar_in_bytes < in_bytes) {
            if (ctx->planar_in) (void)hipFree(ctx->planar_in);
            ctx->planar_in = nullptr;
            ctx->planar_in_bytes = 0;
            LZ_HIP(ctx, hipMalloc(&ctx->planar_in, in_bytes));
            ctx->planar_in_bytes = in_bytes;
        }
        if (ctx->planar_out_bytes < out_bytes) {
            if (ctx->planar_out) (void)hipFree(ctx->planar_out);
            ctx->planar_out = nullptr;
            ctx->planar_out_bytes = 0;
            LZ_HIP(ctx, hipMalloc(&ctx->planar_out, out_bytes));
            ctx->planar_out_bytes = out_bytes;
        }
    }
    rc = lanczos_planar_to_interleaved_device(ctx, d_in_planar, ctx->planar_in, d->in_w, d->in_h, d->channels,
                                              d->bytes_per_sample, frames, stream);
    if (rc != LANCZOS_OK) return rc;
    lanczos_desc whole = *d;
    whole.out_row0 = 0;
    whole.out_rows = 0;
    rc = lanczos_resample_device(ctx, &whole, ctx->planar_in, ctx->planar_out, frames, 0, 0, stream);
    if (rc != LANCZOS_OK) return rc;
    return lanczos_interleaved_to_planar_device(ctx, ctx->planar_out, d_out_planar, d->out_w, d->out_h, d->channels,
                                                d->bytes_per_sample, frames, stream);
}

int lanczos_u8(lanczos_ctx* ctx, const uint8_t* in, int in_w, int in_h, int channels, uint8_t* out, int out_w,
               int out_h, int a) {
    if (in_w <= 0 || in_h <= 0 || out_w <= 0 || out_h <= 0) return LANCZOS_ERR_BAD_ARG;
    lanczos_desc d;
    // SCALE_N/SCALE_D = OUT_WIDTH/IN_WIDTH reduced (lanczos.h:110-114)
    int rc = lanczos_desc_init(&d, in_w, in_h, channels, 1, out_w, in_w, a);
    if (rc != LANCZOS_OK) return rc;
    if (d.out_w != out_w || d.out_h != out_h) return LANCZOS_ERR_BAD_ARG;  // full_TB.h:115-118
    // the reference-shaped single-frame entry point returns lanczos_expected()'s bytes, bit for bit; the batch and
    // device entry points default to LANCZOS_MODE_LSB1 (lanczos_desc_init)
    d.mode = LANCZOS_MODE_EXACT;
    return lanczos_resample_host(ctx, &d, in, out, 1);
}

}  // extern "C"
