// lanczos_multi.hip -- several MI355X of one node behind the same C ABI (include/lanczos_hip.h, "multi-device").
//
// The reference has no parallelism beyond HLS unrolling (ROW_WORKERS rows per strip, lanczos.cpp:72-82); this is the
// scheduler the north star adds: the resample shards with NO data-path collective --
//   LANCZOS_SPLIT_FRAMES  a batch of frames is cut into per-device blocks          (BASELINE config 4)
//   LANCZOS_SPLIT_ROWS    every frame is cut into output row strips + input halo   (BASELINE config 5;
//                         lanczos_strip_input_rows, only strip 0 holds the in-place prefix rows)
// One lanczos_ctx and one host thread per device.  Two data paths:
//   * lanczos_resample_multi_host: every device copies its share straight from / to the caller's host buffers over its OWN
//     PCIe link (no root GPU in the way);
//   * lanczos_resample_multi_root: frames resident on device 0; one exchange step each way over xGMI with RCCL
//     (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd in one process, ncclCommInitAll) -- SURVEY.md 8(e).  librccl is
//     loaded on first use (dlopen), so single-GPU users never touch it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: nothing of librccl is linked (see Rccl below)

#include <cstdio>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "../../include/lanczos_hip.h"
#include "lanczos_exchange.hpp"

namespace {

// every entry below walks over the devices with hipSetDevice: the caller's current device is put back on every return path
struct DeviceGuard {
    int dev = -1;
    DeviceGuard() {
        if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    }
    ~DeviceGuard() {
        if (dev >= 0) (void)hipSetDevice(dev);
    }
};

// ---- RCCL: types and signatures come from the installed header (a drift between it and the calls below is a compile error,
// not a crash at a customer's 8-GPU node); the symbols are still looked up at run time (dlopen on first use), so the library has
// no link-time dependency on librccl and single-GPU users never load it.
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    bool load() {
        if (lib) return true;
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return false;
#define SYM(field, name) *(void**)(&field) = dlsym(h, name)
        SYM(CommInitAll, "ncclCommInitAll");
        SYM(CommDestroy, "ncclCommDestroy");
        SYM(GroupStart, "ncclGroupStart");
        SYM(GroupEnd, "ncclGroupEnd");
        SYM(Send, "ncclSend");
        SYM(Recv, "ncclRecv");
#undef SYM
        if (!(CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv)) {
            // a library without one of the six: not loaded (a second call must not find `lib` set and null pointers behind it)
            CommInitAll = nullptr, CommDestroy = nullptr, GroupStart = nullptr, GroupEnd = nullptr, Send = nullptr, Recv = nullptr;
            dlclose(h);
            return false;
        }
        lib = h;
        return true;
    }
};

// One message list = one RCCL group (lz::exchange_run).  Rank r sends from send_base[r] and receives into recv_base[r] on its own
// communicator and stream.
struct RootOps {
    Rccl* rccl;
    const std::vector<ncclComm_t>* comms;
    const std::vector<hipStream_t>* streams;
    std::vector<const uint8_t*> send_base;
    std::vector<uint8_t*> recv_base;
    int group_start() { return (int)rccl->GroupStart(); }
    int group_end() { return (int)rccl->GroupEnd(); }
    bool has(int rank) const { return rank >= 0 && (size_t)rank < comms->size() && (size_t)rank < send_base.size(); }
    int send(int rank, size_t off, size_t bytes, int peer) {
        if (!has(rank)) return (int)ncclInvalidArgument;   // a list that names a rank this process does not hold
        return (int)rccl->Send(send_base[rank] + off, bytes, ncclUint8, peer, (*comms)[rank], (*streams)[rank]);
    }
    int recv(int rank, size_t off, size_t bytes, int peer) {
        if (!has(rank)) return (int)ncclInvalidArgument;
        return (int)rccl->Recv(recv_base[rank] + off, bytes, ncclUint8, peer, (*comms)[rank], (*streams)[rank]);
    }
};

}  // namespace

struct lanczos_multi {
    std::vector<int> devices;
    std::vector<lanczos_ctx*> ctx;
    // root path
    Rccl rccl;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
    std::vector<void*> d_in, d_out;         // per-device shard buffers (device 0 works in place on the root buffers)
    std::vector<size_t> d_in_bytes, d_out_bytes;
    int last_hip = 0;
    int last_rccl = 0;       // first ncclResult_t a group reported (0 = none)
    int last_rccl_at = 0;    // which message of the list (-1: ncclGroupStart, list size: ncclGroupEnd)
};

namespace {

// shares of a call: the partition functions of the C ABI, applied to every rank
int make_shares(const lanczos_desc* d, int frames, int split, int n, std::vector<lz::ExShare>* sh, lz::ExGeometry* g) {
    g->in_frame = lanczos_in_frame_bytes(d);
    g->out_frame = lanczos_out_frame_bytes(d);
    g->in_pitch = (size_t)d->in_w * d->channels * d->bytes_per_sample;
    g->out_pitch = (size_t)d->out_w * d->channels * d->bytes_per_sample;
    g->frames = frames;
    g->by_rows = split == LANCZOS_SPLIT_ROWS;
    sh->assign(n, lz::ExShare());
    for (int i = 0; i < n; i++) {
        lz::ExShare& s = (*sh)[i];
        s.f0 = 0, s.cnt = frames, s.r0 = 0, s.rows = d->out_h, s.i0 = 0, s.irows = d->in_h;
        if (!g->by_rows) {
            lanczos_partition_frames(frames, n, i, &s.f0, &s.cnt);
            s.in_bytes = (size_t)s.cnt * g->in_frame;
            s.out_bytes = (size_t)s.cnt * g->out_frame;
        } else {
            const int rc = lanczos_partition_rows(d, n, i, &s.r0, &s.rows, &s.i0, &s.irows);
            if (rc != LANCZOS_OK) return rc;
            s.in_bytes = (size_t)frames * s.irows * g->in_pitch;    // strip-major shard: [frame][strip rows]
            s.out_bytes = (size_t)frames * s.rows * g->out_pitch;
        }
    }
    return LANCZOS_OK;
}

}  // namespace

extern "C" {

int lanczos_partition_frames(int frames, int parts, int part, int* first, int* count) {
    if (frames < 0 || parts < 1 || part < 0 || part >= parts || !first || !count) return LANCZOS_ERR_BAD_ARG;
    const int base = frames / parts, rem = frames % parts;     // earlier parts take the remainder
    *first = part * base + (part < rem ? part : rem);
    *count = base + (part < rem ? 1 : 0);
    return LANCZOS_OK;
}

int lanczos_partition_rows(const lanczos_desc* d, int parts, int part, int* out_row0, int* out_rows, int* in_row0,
                           int* in_rows) {
    int rc = lanczos_validate(d);
    if (rc != LANCZOS_OK) return rc;
    if (parts < 1 || part < 0 || part >= parts || !out_row0 || !out_rows || !in_row0 || !in_rows) return LANCZOS_ERR_BAD_ARG;
    if (d->out_rows != 0 && !(d->out_row0 == 0 && d->out_rows == d->out_h)) return LANCZOS_ERR_BAD_ARG;  // whole frames only
    // equal strips; the first one is widened to hold the in-place prefix recurrence (rows [0, M) must be in one place:
    // full_TB.h:67-77) -- K + 2a + 2 rows always cover M
    const int K = lanczos_inplace_rows(d);
    const int min_first = K > 0 ? K + 2 * d->a + 2 : 0;
    auto bound = [&](int i) {
        long long b = (long long)d->out_h * i / parts;
        if (i >= 1 && i < parts && b < min_first) b = min_first < d->out_h ? min_first : d->out_h;
        return (int)b;
    };
    const int r0 = bound(part), r1 = part + 1 == parts ? d->out_h : bound(part + 1);
    *out_row0 = r0;
    *out_rows = r1 > r0 ? r1 - r0 : 0;
    *in_row0 = *in_rows = 0;
    if (*out_rows == 0) return LANCZOS_OK;
    return lanczos_strip_input_rows(d, r0, *out_rows, in_row0, in_rows);
}

int lanczos_multi_create(lanczos_multi** out, const int* devices, int n_devices) {
    if (!out || !devices || n_devices < 1 || n_devices > 64) return LANCZOS_ERR_BAD_ARG;
    *out = nullptr;
    DeviceGuard restore;  // lanczos_create selects each device in turn
    lanczos_multi* m = new (std::nothrow) lanczos_multi();
    if (!m) return LANCZOS_ERR_NOMEM;
    for (int i = 0; i < n_devices; i++) {
        lanczos_ctx* c = nullptr;
        const int rc = lanczos_create(&c, devices[i]);
        if (rc != LANCZOS_OK) {
            for (lanczos_ctx* p : m->ctx) lanczos_destroy(p);
            delete m;
            return rc;
        }
        m->devices.push_back(devices[i]);
        m->ctx.push_back(c);
    }
    *out = m;
    return LANCZOS_OK;
}

int lanczos_multi_destroy(lanczos_multi* m) {
    if (!m) return LANCZOS_ERR_BAD_ARG;
    DeviceGuard restore;
    for (size_t i = 0; i < m->streams.size(); i++) {
        (void)hipSetDevice(m->devices[i]);
        if (m->streams[i]) (void)hipStreamSynchronize(m->streams[i]);
    }
    for (size_t i = 0; i < m->comms.size(); i++)
        if (m->comms[i]) m->rccl.CommDestroy(m->comms[i]);
    for (size_t i = 0; i < m->d_in.size(); i++) {
        (void)hipSetDevice(m->devices[i]);
        if (m->d_in[i]) (void)hipFree(m->d_in[i]);
        if (m->d_out[i]) (void)hipFree(m->d_out[i]);
        if (m->streams[i]) (void)hipStreamDestroy(m->streams[i]);
    }
    for (lanczos_ctx* c : m->ctx) lanczos_destroy(c);
    delete m;
    return LANCZOS_OK;
}

int lanczos_multi_devices(const lanczos_multi* m) { return m ? (int)m->ctx.size() : 0; }

int lanczos_multi_last_error(const lanczos_multi* m, int* hip_error, int* rccl_error, int* rccl_message) {
    if (!m) return LANCZOS_ERR_BAD_ARG;
    if (hip_error) *hip_error = m->last_hip;
    if (rccl_error) *rccl_error = m->last_rccl;
    if (rccl_message) *rccl_message = m->last_rccl_at;
    return LANCZOS_OK;
}

// The exchange of lanczos_resample_multi_root as data (no device needed): phase 0 = scatter (root -> peers, input shares),
// phase 1 = gather (peers -> root, output shares).  Returns the number of messages (also when `cap` is smaller: call with
// cap = 0 to size the array), or a negative LANCZOS_ERR_*.
int lanczos_multi_exchange_plan(const lanczos_desc* d, int frames, int split, int n_devices, int phase, lanczos_xfer* out, int cap) {
    int rc = lanczos_validate(d);
    if (rc != LANCZOS_OK) return -rc;
    if (frames <= 0 || n_devices < 1 || n_devices > 64 || (phase != 0 && phase != 1) || cap < 0 || (cap > 0 && !out)) return -LANCZOS_ERR_BAD_ARG;
    if (split != LANCZOS_SPLIT_FRAMES && split != LANCZOS_SPLIT_ROWS) return -LANCZOS_ERR_BAD_ARG;
    if (d->out_rows != 0 && !(d->out_row0 == 0 && d->out_rows == d->out_h)) return -LANCZOS_ERR_BAD_ARG;
    std::vector<lz::ExShare> sh;
    lz::ExGeometry g;
    rc = make_shares(d, frames, split, n_devices, &sh, &g);
    if (rc != LANCZOS_OK) return -rc;
    std::vector<lz::ExXfer> list;
    if (phase == 0) lz::exchange_scatter_plan(g, sh, &list);
    else lz::exchange_gather_plan(g, sh, &list);
    for (int k = 0; k < (int)list.size() && k < cap; k++)
        out[k] = lanczos_xfer{list[k].src, list[k].dst, list[k].src_off, list[k].dst_off, list[k].bytes};
    return (int)list.size();
}

// Host buffers in, host buffers out: `frames` whole frames back to back (what lanczos_resample_host takes).  One thread per
// device; every device moves only its own share over its own PCIe link.
int lanczos_resample_multi_host(lanczos_multi* m, const lanczos_desc* d, const void* in, void* out, int frames, int split) {
    if (!m || !in || !out || frames <= 0) return LANCZOS_ERR_BAD_ARG;
    int rc = lanczos_validate(d);
    if (rc != LANCZOS_OK) return rc;
    if (split != LANCZOS_SPLIT_FRAMES && split != LANCZOS_SPLIT_ROWS) return LANCZOS_ERR_BAD_ARG;
    if (d->out_rows != 0 && !(d->out_row0 == 0 && d->out_rows == d->out_h)) return LANCZOS_ERR_BAD_ARG;
    const int n = (int)m->ctx.size();
    const size_t in_frame = lanczos_in_frame_bytes(d), out_frame = lanczos_out_frame_bytes(d);
    const size_t in_pitch = (size_t)d->in_w * d->channels * d->bytes_per_sample;
    const size_t out_pitch = (size_t)d->out_w * d->channels * d->bytes_per_sample;
    std::vector<int> rcs(n, LANCZOS_OK);
    std::vector<std::thread> th;
    for (int i = 0; i < n; i++) {
        th.emplace_back([&, i]() {
            lanczos_desc dd = *d;
            dd.out_row0 = dd.out_rows = 0;
            if (split == LANCZOS_SPLIT_FRAMES) {
                int f0, cnt;
                lanczos_partition_frames(frames, n, i, &f0, &cnt);
                if (cnt > 0)
                    rcs[i] = lanczos_resample_host(m->ctx[i], &dd, (const uint8_t*)in + (size_t)f0 * in_frame,
                                                   (uint8_t*)out + (size_t)f0 * out_frame, cnt);
            } else {
                int r0, rows, i0, irows;
                rcs[i] = lanczos_partition_rows(&dd, n, i, &r0, &rows, &i0, &irows);
                if (rcs[i] != LANCZOS_OK || rows == 0) return;
                dd.out_row0 = r0;
                dd.out_rows = rows;
                for (int f = 0; f < frames && rcs[i] == LANCZOS_OK; f++)  // a strip of a host frame is contiguous: one call per frame
                    rcs[i] = lanczos_resample_host(m->ctx[i], &dd, (const uint8_t*)in + (size_t)f * in_frame + (size_t)i0 * in_pitch,
                                                   (uint8_t*)out + (size_t)f * out_frame + (size_t)r0 * out_pitch, 1);
            }
        });
    }
    for (auto& t : th) t.join();
    for (int i = 0; i < n; i++)
        if (rcs[i] != LANCZOS_OK) return rcs[i];
    return LANCZOS_OK;
}

#define LZM_HIP(call)                     \
    do {                                  \
        hipError_t e_ = (call);           \
        if (e_ != hipSuccess) {           \
            m->last_hip = (int)e_;        \
            return LANCZOS_ERR_HIP;       \
        }                                 \
    } while (0)

// Frames resident on the ROOT device (devices[0]) in, results on the root device out.  scatter (ncclSend/ncclRecv group) ->
// every device resamples its share -> gather (second group).  Synchronous.  compute_ms / total_ms (optional): wall time of
// the resample step alone (max over devices) and of the whole call.
//
// STATUS: with more than one device this entry has NOT run on hardware (no multi-GPU node was available to the builders); the
// message lists and the group handling are checked on the CPU (tests/test_multi_entry.py: lanczos_multi_exchange_plan against
// lanczos_partition_*; tests/native/exchange_check.cpp: every send/recv code looked at, the group closed on every path).
int lanczos_resample_multi_root(lanczos_multi* m, const lanczos_desc* d, const void* d_in_root, void* d_out_root, int frames,
                                int split, double* compute_ms, double* total_ms) {
    if (!m || !d_in_root || !d_out_root || frames <= 0) return LANCZOS_ERR_BAD_ARG;
    int rc = lanczos_validate(d);
    if (rc != LANCZOS_OK) return rc;
    if (split != LANCZOS_SPLIT_FRAMES && split != LANCZOS_SPLIT_ROWS) return LANCZOS_ERR_BAD_ARG;
    if (d->out_rows != 0 && !(d->out_row0 == 0 && d->out_rows == d->out_h)) return LANCZOS_ERR_BAD_ARG;
    const int n = (int)m->ctx.size();
    for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++)
            if (m->devices[i] == m->devices[j]) return LANCZOS_ERR_BAD_ARG;  // one rank per physical device
    DeviceGuard restore;  // the caller's current device (its inputs live on devices[0]) is put back on every return path
    m->last_hip = m->last_rccl = m->last_rccl_at = 0;
    if (m->streams.empty()) {
        m->streams.assign(n, nullptr);
        m->d_in.assign(n, nullptr);
        m->d_out.assign(n, nullptr);
        m->d_in_bytes.assign(n, 0);
        m->d_out_bytes.assign(n, 0);
        for (int i = 0; i < n; i++) {
            LZM_HIP(hipSetDevice(m->devices[i]));
            LZM_HIP(hipStreamCreateWithFlags(&m->streams[i], hipStreamNonBlocking));
        }
    }
    if (n > 1 && m->comms.empty()) {
        if (!m->rccl.load()) return LANCZOS_ERR_UNSUPPORTED;  // no librccl on this system
        m->comms.assign(n, nullptr);
        const int r = (int)m->rccl.CommInitAll(m->comms.data(), n, m->devices.data());
        if (r != 0) {
            m->comms.clear();
            m->last_rccl = r;
            m->last_rccl_at = -1;
            return LANCZOS_ERR_RCCL;
        }
    }
    std::vector<lz::ExShare> sh;
    lz::ExGeometry geo;
    rc = make_shares(d, frames, split, n, &sh, &geo);
    if (rc != LANCZOS_OK) return rc;
    const size_t in_frame = geo.in_frame, out_frame = geo.out_frame, in_pitch = geo.in_pitch, out_pitch = geo.out_pitch;
    for (int i = 1; i < n; i++) {  // peer shard buffers
        LZM_HIP(hipSetDevice(m->devices[i]));
        if (m->d_in_bytes[i] < sh[i].in_bytes) {
            if (m->d_in[i]) (void)hipFree(m->d_in[i]);
            m->d_in[i] = nullptr;
            m->d_in_bytes[i] = 0;
            LZM_HIP(hipMalloc(&m->d_in[i], sh[i].in_bytes ? sh[i].in_bytes : 16));
            m->d_in_bytes[i] = sh[i].in_bytes;
        }
        if (m->d_out_bytes[i] < sh[i].out_bytes) {
            if (m->d_out[i]) (void)hipFree(m->d_out[i]);
            m->d_out[i] = nullptr;
            m->d_out_bytes[i] = 0;
            LZM_HIP(hipMalloc(&m->d_out[i], sh[i].out_bytes ? sh[i].out_bytes : 16));
            m->d_out_bytes[i] = sh[i].out_bytes;
        }
    }
    auto now_ms = []() {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6;
    };
    auto sync_all = [&]() -> int {
        for (int i = 0; i < n; i++) {
            LZM_HIP(hipSetDevice(m->devices[i]));
            LZM_HIP(hipStreamSynchronize(m->streams[i]));
        }
        return LANCZOS_OK;
    };
    auto exchange = [&](bool scatter) -> int {
        std::vector<lz::ExXfer> list;
        if (scatter) lz::exchange_scatter_plan(geo, sh, &list);
        else lz::exchange_gather_plan(geo, sh, &list);
        // rank r's buffer: the root's caller buffers (r == 0) or the peer's shard buffer
        RootOps ops{&m->rccl, &m->comms, &m->streams, {}, {}};
        for (int r = 0; r < n; r++) {
            ops.send_base.push_back(r == 0 ? (const uint8_t*)d_in_root : (const uint8_t*)(scatter ? m->d_in[r] : m->d_out[r]));
            ops.recv_base.push_back(r == 0 ? (uint8_t*)d_out_root : (uint8_t*)(scatter ? m->d_in[r] : m->d_out[r]));
        }
        int at = 0;
        const int r = lz::exchange_run(ops, list, &at);
        if (r != 0) {
            m->last_rccl = r;
            m->last_rccl_at = at;
            (void)sync_all();  // whatever was queued before the failure is drained before the caller sees the error
            return LANCZOS_ERR_RCCL;
        }
        return LANCZOS_OK;
    };
    const double t_start = now_ms();
    if (n > 1 && (rc = exchange(true)) != LANCZOS_OK) return rc;   // ---- scatter: root -> peers
    if ((rc = sync_all()) != LANCZOS_OK) return rc;
    const double t_c0 = now_ms();
    // ---- compute: every device its share, on its own stream
    for (int i = 0; i < n; i++) {
        const lz::ExShare& s = sh[i];
        if (s.out_bytes == 0) continue;
        lanczos_desc dd = *d;
        dd.out_row0 = dd.out_rows = 0;
        if (split == LANCZOS_SPLIT_FRAMES) {
            const void* src = i == 0 ? (const void*)((const uint8_t*)d_in_root + (size_t)s.f0 * in_frame) : m->d_in[i];
            void* dst = i == 0 ? (void*)((uint8_t*)d_out_root + (size_t)s.f0 * out_frame) : m->d_out[i];
            rc = lanczos_resample_device(m->ctx[i], &dd, src, dst, s.cnt, 0, 0, m->streams[i]);
        } else {
            dd.out_row0 = s.r0;
            dd.out_rows = s.rows;
            if (i == 0)  // the root works in place on the full frames: frame strides are those of the whole frame
                rc = lanczos_resample_device(m->ctx[0], &dd, (const uint8_t*)d_in_root + (size_t)s.i0 * in_pitch,
                                             (uint8_t*)d_out_root + (size_t)s.r0 * out_pitch, frames, in_frame, out_frame, m->streams[0]);
            else
                rc = lanczos_resample_device(m->ctx[i], &dd, m->d_in[i], m->d_out[i], frames, 0, 0, m->streams[i]);
        }
        if (rc != LANCZOS_OK) {
            (void)sync_all();
            return rc;
        }
    }
    if ((rc = sync_all()) != LANCZOS_OK) return rc;
    const double t_c1 = now_ms();
    if (n > 1 && (rc = exchange(false)) != LANCZOS_OK) return rc;  // ---- gather: peers -> root
    if ((rc = sync_all()) != LANCZOS_OK) return rc;
    if (compute_ms) *compute_ms = t_c1 - t_c0;
    if (total_ms) *total_ms = now_ms() - t_start;
    return LANCZOS_OK;
}

// The exchange machinery of lanczos_resample_multi_root on ONE rank: librccl loaded (dlopen + the six symbols), a one-device
// communicator (ncclCommInitAll on devices[0]), and `messages` self messages (rank 0 -> rank 0, `bytes` each) issued through the
// SAME executor and adapter as the real exchange (lz::exchange_run, RootOps) from one device buffer into another; the bytes are
// compared afterwards.  fail_at >= 0: message `fail_at` names a peer that does not exist -- the call must come back as
// LANCZOS_ERR_RCCL with lanczos_multi_last_error() naming that message, the group closed, the communicator still usable.
// This is what a one-GPU box can execute of the RCCL path (the multi-rank exchange itself needs a multi-GPU node).
int lanczos_multi_exchange_selftest(lanczos_multi* m, int messages, size_t bytes, int fail_at) {
    if (!m || messages < 1 || messages > 1024 || bytes == 0 || bytes > ((size_t)1 << 28) || fail_at >= messages) return LANCZOS_ERR_BAD_ARG;
    DeviceGuard restore;
    m->last_hip = m->last_rccl = m->last_rccl_at = 0;
    if (!m->rccl.load()) return LANCZOS_ERR_UNSUPPORTED;  // no librccl on this system
    const int dev = m->devices[0];
    LZM_HIP(hipSetDevice(dev));
    std::vector<ncclComm_t> comm(1, nullptr);
    std::vector<hipStream_t> stream(1, nullptr);
    uint8_t *a = nullptr, *b = nullptr;
    std::vector<uint8_t> ha(bytes * messages), hb(bytes * messages, 0);
    for (size_t i = 0; i < ha.size(); i++) ha[i] = (uint8_t)(i * 2654435761u >> 13);
    int rc = LANCZOS_OK;
    auto cleanup = [&]() {
        if (stream[0]) (void)hipStreamSynchronize(stream[0]);
        if (comm[0]) (void)m->rccl.CommDestroy(comm[0]);
        if (a) (void)hipFree(a);
        if (b) (void)hipFree(b);
        if (stream[0]) (void)hipStreamDestroy(stream[0]);
    };
    hipError_t he = hipStreamCreateWithFlags(&stream[0], hipStreamNonBlocking);
    if (he == hipSuccess) he = hipMalloc((void**)&a, ha.size());
    if (he == hipSuccess) he = hipMalloc((void**)&b, hb.size());
    if (he == hipSuccess) he = hipMemcpy(a, ha.data(), ha.size(), hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemset(b, 0, hb.size());
    if (he != hipSuccess) {
        m->last_hip = (int)he;
        cleanup();
        return LANCZOS_ERR_HIP;
    }
    const int ri = (int)m->rccl.CommInitAll(comm.data(), 1, &dev);
    if (ri != 0) {
        comm[0] = nullptr;
        m->last_rccl = ri;
        m->last_rccl_at = -1;
        cleanup();
        return LANCZOS_ERR_RCCL;
    }
    std::vector<lz::ExXfer> list;
    for (int k = 0; k < messages; k++)  // reversed placement: message k lands in slot messages-1-k
        list.push_back(lz::ExXfer{0, k == fail_at ? 5 : 0, (size_t)k * bytes, (size_t)(messages - 1 - k) * bytes, bytes});
    RootOps ops{&m->rccl, &comm, &stream, {a, a, a, a, a, a}, {b, b, b, b, b, b}};  // (ranks up to 5: the failing message's "peer")
    int at = 0;
    const int r = lz::exchange_run(ops, list, &at);
    he = hipStreamSynchronize(stream[0]);
    if (r != 0) {
        m->last_rccl = r;
        m->last_rccl_at = at;
        rc = LANCZOS_ERR_RCCL;
        // the group was closed by exchange_run: the communicator must still work -- one more, clean, message
        std::vector<lz::ExXfer> one(1, lz::ExXfer{0, 0, 0, 0, bytes});
        int at2 = 0;
        if (lz::exchange_run(ops, one, &at2) != 0 || hipStreamSynchronize(stream[0]) != hipSuccess) rc = LANCZOS_ERR_HIP;
    } else if (he != hipSuccess) {
        m->last_hip = (int)he;
        rc = LANCZOS_ERR_HIP;
    } else {
        he = hipMemcpy(hb.data(), b, hb.size(), hipMemcpyDeviceToHost);
        if (he != hipSuccess) {
            m->last_hip = (int)he;
            rc = LANCZOS_ERR_HIP;
        } else {
            for (int k = 0; k < messages && rc == LANCZOS_OK; k++)
                if (memcmp(hb.data() + (size_t)(messages - 1 - k) * bytes, ha.data() + (size_t)k * bytes, bytes) != 0) rc = LANCZOS_ERR_HIP;
        }
    }
    cleanup();
    return rc;
}

// ---- plain-C callers without the HIP headers (host/main.c --root): device memory on a chosen device
int lanczos_device_alloc(int device, void** p, size_t bytes) {
    if (!p || bytes == 0) return LANCZOS_ERR_BAD_ARG;
    *p = nullptr;
    DeviceGuard restore;
    if (hipSetDevice(device) != hipSuccess) return LANCZOS_ERR_NO_DEVICE;
    return hipMalloc(p, bytes) == hipSuccess ? LANCZOS_OK : LANCZOS_ERR_NOMEM;
}

int lanczos_device_free(int device, void* p) {
    if (!p) return LANCZOS_OK;
    DeviceGuard restore;
    if (hipSetDevice(device) != hipSuccess) return LANCZOS_ERR_NO_DEVICE;
    return hipFree(p) == hipSuccess ? LANCZOS_OK : LANCZOS_ERR_HIP;
}

int lanczos_device_copy(int device, void* dst, const void* src, size_t bytes, int to_device) {
    if (!dst || !src) return LANCZOS_ERR_BAD_ARG;
    DeviceGuard restore;
    if (hipSetDevice(device) != hipSuccess) return LANCZOS_ERR_NO_DEVICE;
    return hipMemcpy(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost) == hipSuccess ? LANCZOS_OK : LANCZOS_ERR_HIP;
}

}  // extern "C"
