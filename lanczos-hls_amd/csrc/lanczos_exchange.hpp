// lanczos_exchange.hpp -- the root scatter / gather of lanczos_resample_multi_root as DATA: which bytes travel from which
// rank's buffer to which rank's buffer (one exchange step each way, grouped ncclSend / ncclRecv because the halos
// make the shards unequal), and an executor that issues one list as ONE RCCL group with every return code checked.
//
// Pure host code with no HIP or RCCL dependency: tests/native/exchange_check.cpp compiles it with g++ and drives the executor
// with recording / failing stubs, because the multi-GPU path itself cannot run on the one-GPU boxes this repo is tested on.
// The reference has nothing comparable (its only parallelism is HLS unrolling, lanczos.cpp:72-82).
#pragma once
#include <cstddef>
#include <vector>

namespace lz {

// one rank's share of a call (lanczos_partition_frames / lanczos_partition_rows)
struct ExShare {
    int f0 = 0, cnt = 0;              // split by frames: first frame, frame count
    int r0 = 0, rows = 0;             // split by rows: output strip ...
    int i0 = 0, irows = 0;            // ... and the input rows it needs (halo included)
    size_t in_bytes = 0, out_bytes = 0;  // size of the rank's shard buffers
};

// one message: `bytes` from rank `src`'s buffer at src_off to rank `dst`'s buffer at dst_off.  Root offsets index the caller's
// root buffers (all frames, whole frames); peer offsets index the peer's shard buffer (d_in[i] / d_out[i]).
struct ExXfer {
    int src, dst;
    size_t src_off, dst_off, bytes;
};

struct ExGeometry {
    size_t in_frame, out_frame;   // bytes of one whole frame
    size_t in_pitch, out_pitch;   // bytes of one row
    int frames;
    bool by_rows;
};

// root (rank 0) -> peers: every peer's input share.  Split by frames: one contiguous piece per peer.  Split by rows: one piece per
// frame (a strip of a frame is contiguous in the root's frame; the peer stores its strips frame after frame).
inline void exchange_scatter_plan(const ExGeometry& g, const std::vector<ExShare>& sh, std::vector<ExXfer>* out) {
    out->clear();
    for (int i = 1; i < (int)sh.size(); i++) {
        if (sh[i].in_bytes == 0) continue;
        if (!g.by_rows) {
            out->push_back(ExXfer{0, i, (size_t)sh[i].f0 * g.in_frame, 0, sh[i].in_bytes});
        } else {
            const size_t piece = (size_t)sh[i].irows * g.in_pitch;
            for (int f = 0; f < g.frames; f++)
                out->push_back(ExXfer{0, i, (size_t)f * g.in_frame + (size_t)sh[i].i0 * g.in_pitch, (size_t)f * piece, piece});
        }
    }
}

// peers -> root: every peer's output share, to where it belongs in the root's output frames
inline void exchange_gather_plan(const ExGeometry& g, const std::vector<ExShare>& sh, std::vector<ExXfer>* out) {
    out->clear();
    for (int i = 1; i < (int)sh.size(); i++) {
        if (sh[i].out_bytes == 0) continue;
        if (!g.by_rows) {
            out->push_back(ExXfer{i, 0, 0, (size_t)sh[i].f0 * g.out_frame, sh[i].out_bytes});
        } else {
            const size_t piece = (size_t)sh[i].rows * g.out_pitch;
            for (int f = 0; f < g.frames; f++)
                out->push_back(ExXfer{i, 0, (size_t)f * piece, (size_t)f * g.out_frame + (size_t)sh[i].r0 * g.out_pitch, piece});
        }
    }
}

// Issues one list as ONE group.  Ops: int group_start(), int group_end(), int send(rank, offset, bytes, peer),
// int recv(rank, offset, bytes, peer) -- 0 = success (ncclSuccess).  Every code is looked at; after the first failure nothing
// more is queued, but a group that was opened is ALWAYS closed (an open group would swallow every later RCCL call of the
// process).  Returns the first non-zero code, *failed_at = index of the message that failed (-1: group_start, size: group_end).
template <typename Ops>
inline int exchange_run(Ops& ops, const std::vector<ExXfer>& list, int* failed_at) {
    if (failed_at) *failed_at = -2;
    int rc = ops.group_start();
    if (rc != 0) {
        if (failed_at) *failed_at = -1;
        return rc;  // nothing was opened
    }
    int first = 0, where = -2;
    for (size_t k = 0; k < list.size() && first == 0; k++) {
        const ExXfer& x = list[k];
        int r = ops.send(x.src, x.src_off, x.bytes, x.dst);
        if (r == 0) r = ops.recv(x.dst, x.dst_off, x.bytes, x.src);
        if (r != 0) first = r, where = (int)k;
    }
    const int end = ops.group_end();  // always: closes the group even when a send / recv was refused
    if (first == 0 && end != 0) first = end, where = (int)list.size();
    if (failed_at) *failed_at = where;
    return first;
}

}  // namespace lz
