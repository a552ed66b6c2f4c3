/*
 * include/lanczos_hip.h -- C ABI of the MI355X (gfx950) Lanczos-a resampler.
 *
 * This is the drop-in boundary for the reference's resample path.  Plain pointers and sizes only;
 * no C++/torch types.  What each entry point replaces in /root/reference/LanczosUpscaler:
 *
 *   lanczos_resample_host / lanczos_u8     <- void lanczos(stream_t, stream_t)   lanczos.h:121-126,
 *                                             lanczos.cpp:86-98 (called at full_TB.h:140); results are
 *                                             those of the software model lanczos_expected(),
 *                                             full_TB.h:79-96, on the stb interleaved layout of
 *                                             full_TB.h:107 (R | G<<8 | B<<16, worker.cpp:35-43):
 *                                             bit-identical with LANCZOS_MODE_EXACT (what lanczos_u8 and
 *                                             hls_compat.hpp's lanczos() use), within +-1 LSB per sample
 *                                             with LANCZOS_MODE_LSB1 (what lanczos_desc_init sets: the
 *                                             faster default of the batch / device entry points)
 *   lanczos_resample_device                <- the same, for callers that already hold device memory
 *                                             (batches of frames, row strips of one frame)
 *   lanczos_kernel / lanczos_kernel_idx    <- double lanczos_kernel(double)      full_TB.h:51-53 and
 *                                             kernel_t lanczos_kernel(input_idx_t, output_idx_t, scale_t)
 *                                             kernel.h:6, kernel.cpp:61-67
 *   lanczos_desc                           <- the compile-time macros of params.h (lanczos.h:9-31):
 *                                             IN_WIDTH, IN_HEIGHT, OUT_WIDTH, OUT_HEIGHT, NUM_CHANNELS,
 *                                             LANCZOS_A, SCALE_N, SCALE_D -- here run-time arguments
 *   error codes                            <- the EXIT_FAILURE checks of full_TB.h:110-123
 *
 * Semantics (all verified against the reference's compiled software path, tests/):
 *   horizontal pass then vertical pass; taps floor(x)-a+1 .. floor(x)+a with x = out/((double)N/D);
 *   out-of-range taps dropped, no renormalisation; every store clamps to [0,max] and TRUNCATES; the
 *   horizontal result is stored as a truncated integer before the vertical pass; the vertical pass is
 *   IN PLACE bottom-to-top, so the first K output rows read already-written output rows
 *   (full_TB.h:67-77; K = lanczos_inplace_rows()).
 */
#ifndef LANCZOS_HIP_H
#define LANCZOS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LANCZOS_OK 0
#define LANCZOS_ERR_BAD_ARG 1      /* null pointer, non-positive size, out != in*N/D, a or channels unsupported */
#define LANCZOS_ERR_UNSUPPORTED 2  /* valid request this build cannot run (scale < 1; a row strip that starts inside the in-place prefix rows) */
#define LANCZOS_ERR_NO_DEVICE 3    /* no HIP device / device index out of range */
#define LANCZOS_ERR_HIP 4          /* a HIP runtime call failed; see lanczos_last_hip_error() */
#define LANCZOS_ERR_NOMEM 5
#define LANCZOS_ERR_RCCL 6         /* an RCCL call of the root exchange failed; see lanczos_multi_last_error() */

/* parity modes (lanczos_desc.mode) */
#define LANCZOS_MODE_LSB1 0   /* default: horizontal pass bit-exact, vertical pass f32 accumulators;
                                 every output sample within +-1 LSB of the reference software path */
#define LANCZOS_MODE_EXACT 1  /* every output sample bit-identical to the reference software path */
#define LANCZOS_MODE_HLS 2    /* the semantics of the reference's HLS pipeline instead (lanczos.cpp:86-98): vertical pass
                                 first (lanczos.cpp:21-51), ROM weights L(|o*D - i*N| / N) (kernel.cpp:40-59), zero rows /
                                 samples above and left of the image, the last row / sample repeated below and right of it
                                 (worker.cpp:147-153,176-188,244,256-265), each pass's sum clamped to [min,max] of its two
                                 centre taps -- the de-ringing of README.md:4 (worker.cpp:66-74,103-111) -- a real-valued
                                 intermediate, truncating final store (worker.cpp:118-130).  Computed in f64 with exact phase
                                 stepping; the hardware's ap_fixed arithmetic is NOT emulated.  Parity unpinned by the
                                 reference (the HLS path cannot be built without Xilinx headers); checked against
                                 oracle/lanczos_hls_model.c.  No in-place prefix: lanczos_inplace_rows() is 0. */

/* which kernel family served the last call (lanczos_last_kernel) */
#define LANCZOS_KERNEL_NONE 0
#define LANCZOS_KERNEL_GENERIC 1  /* table-driven, any rational scale > 1, f64 throughout (always exact) */
#define LANCZOS_KERNEL_FAST 2     /* specialised: integer scale, LDS-staged tiles, f32 taps + exact fallback */
#define LANCZOS_KERNEL_HLS 3      /* LANCZOS_MODE_HLS: V-then-H with de-ringing clamps, f64 */

typedef struct lanczos_ctx lanczos_ctx; /* opaque: device, stream, cached tap tables, staging buffers */

typedef struct lanczos_desc {
    int32_t in_w, in_h;         /* IN_WIDTH, IN_HEIGHT  (pixels, FULL frame)                     */
    int32_t out_w, out_h;       /* OUT_WIDTH, OUT_HEIGHT (pixels, FULL frame) = in * N / D         */
    int32_t channels;           /* NUM_CHANNELS: 1, 3 or 4, interleaved                            */
    int32_t bytes_per_sample;   /* 1 (u8, the reference) or 2 (u16 generalisation, clamp 65535)    */
    int32_t scale_n, scale_d;   /* SCALE_N / SCALE_D >= 1 (1/1 included: DESIGN.md 1) */
    int32_t a;                  /* LANCZOS_A: 2, 3 or 4                                            */
    int32_t mode;               /* LANCZOS_MODE_*                                                  */
    /* Row strip of the frame to produce (multi-GPU tile sharding).  out_rows == 0 means the whole
     * frame.  With a strip, `in` points at input row lanczos_strip_input_rows().in_row0 and `out`
     * at output row out_row0; both keep the full-frame row pitch. */
    int32_t out_row0, out_rows;
    /* reserved[0] = BIT_PRECISION (params.h; lanczos.h:74-81) of LANCZOS_MODE_HLS's fixed-point emulation: 0 = ideal arithmetic
     * (default), 1..20 = weights cut to BP fractional bits (kernel_t = ap_fixed<8+BP,8>) and the horizontal accumulator cut
     * to BP fractional bits after every tap (num_el_t = ap_fixed<10+BP,10>), AP_TRN / AP_WRAP as the declarations default to.
     * 8-bit samples, HLS mode only.  Parity unpinned (the hardware's ROM comes out of hls::sinpi).  reserved[1..2]: 0. */
    int32_t reserved[3];
} lanczos_desc;

/* ---- descriptor helpers (host only, no GPU needed) ---- */
/* Fill a descriptor for a whole frame; out dims = in * n / d (integer division, as OUT_WIDTH = IN_WIDTH*3). */
int lanczos_desc_init(lanczos_desc* d, int in_w, int in_h, int channels, int bytes_per_sample,
                      int scale_n, int scale_d, int a);
int lanczos_validate(const lanczos_desc* d);
/* K: output rows [0,K) depend on already-written output rows (in-place vertical pass). */
int lanczos_inplace_rows(const lanczos_desc* d);
/* Input rows [*in_row0, *in_row0 + *in_rows) needed to produce output rows [out_row0, out_row0+out_rows). */
int lanczos_strip_input_rows(const lanczos_desc* d, int out_row0, int out_rows, int* in_row0, int* in_rows);
size_t lanczos_in_frame_bytes(const lanczos_desc* d);   /* full frame */
size_t lanczos_out_frame_bytes(const lanczos_desc* d);  /* full frame */

/* ---- weights (host, double) ---- */
double lanczos_kernel(double x, int a);                                                /* full_TB.h:51-53 */
double lanczos_kernel_idx(int in_idx, int out_idx, int scale_n, int scale_d, int a);   /* kernel.h:6     */
/* Tap table of one axis (0 = horizontal, 1 = vertical): for every output index o, first[o] =
 * floor(x)-a+1 and weights[o*2a + k] = L(x - (first[o]+k)), zero where the tap is out of range. */
int lanczos_taps_host(const lanczos_desc* d, int axis, int32_t* first, double* weights);

/* ---- context ---- */
int lanczos_create(lanczos_ctx** ctx, int device);
int lanczos_destroy(lanczos_ctx* ctx);

/* Page-locked host memory for lanczos_resample_host callers that want the copies overlapped. */
int lanczos_host_alloc(void** p, size_t bytes);
int lanczos_host_free(void* p);

/* ---- the resample ---- */
/* Host buffers (what stbi_load returned / what stbi_write_png takes), `frames` frames back to back.
 * Synchronous for the caller; inside, groups of frames are pipelined over copy-in / resample / copy-out
 * streams, which overlaps the PCIe copies when the buffers are page-locked (lanczos_host_alloc). */
int lanczos_resample_host(lanczos_ctx* ctx, const lanczos_desc* d, const void* in, void* out, int frames);
/* Device buffers, asynchronous on `stream` (a hipStream_t; NULL = the default stream, so work queued there
 * by the caller is ordered before the resample).
 * Frame f starts at in + f*in_frame_stride / out + f*out_frame_stride (bytes; 0 = tightly packed). */
int lanczos_resample_device(lanczos_ctx* ctx, const lanczos_desc* d, const void* d_in, void* d_out,
                            int frames, size_t in_frame_stride, size_t out_frame_stride, void* stream);
/* ---- planar frames ----
 * The reference's software model and testbench hold PLANAR frames, byte img_in[NUM_CHANNELS][IN_HEIGHT][IN_WIDTH] /
 * img_out_ex[NUM_CHANNELS][OUT_HEIGHT][OUT_WIDTH] (full_TB.h:20-21), and convert from/to the interleaved stb buffer with
 * host loops (full_TB.h:127-138, 146-165).  The same three steps on the device, asynchronous on `stream`
 * (NULL = default stream); `frames` frames back to back, each [channels][h][w]: */
int lanczos_planar_to_interleaved_device(lanczos_ctx* ctx, const void* d_planar, void* d_interleaved, int w, int h,
                                         int channels, int bytes_per_sample, int frames, void* stream);
int lanczos_interleaved_to_planar_device(lanczos_ctx* ctx, const void* d_interleaved, void* d_planar, int w, int h,
                                         int channels, int bytes_per_sample, int frames, void* stream);
/* img_in[C][IN_H][IN_W] -> img_out[C][OUT_H][OUT_W]: what lanczos_expected(img_in, img_out_ex) computes (full_TB.h:79-96).
 * Whole frames only; the interleaved scratch frames live in the context (one stream at a time per context). */
int lanczos_resample_planar_device(lanczos_ctx* ctx, const lanczos_desc* d, const void* d_in_planar, void* d_out_planar,
                                   int frames, void* stream);
/* The reference's call shape: sizes as plain ints, RGB8 in/out, scale = out_w/in_w reduced.
 * Always LANCZOS_MODE_EXACT: the bytes are those of lanczos_expected() (full_TB.h:79-96). */
int lanczos_u8(lanczos_ctx* ctx, const uint8_t* in, int in_w, int in_h, int channels,
               uint8_t* out, int out_w, int out_h, int a);

/* ---- several devices of one node (lanczos_multi.hip) ----
 * The reference has no parallelism beyond HLS unrolling (ROW_WORKERS, lanczos.cpp:72-82); this is the north star's
 * multi-GPU scheduler.  The resample shards with no data-path collective: */
#define LANCZOS_SPLIT_FRAMES 0  /* a batch of frames in per-device blocks (BASELINE config 4) */
#define LANCZOS_SPLIT_ROWS 1    /* every frame in output row strips + input halo (BASELINE config 5) */
typedef struct lanczos_multi lanczos_multi; /* opaque: one lanczos_ctx per device (+ RCCL communicators for the root path) */
/* partition arithmetic (host only, no GPU): part `part` of `parts` */
int lanczos_partition_frames(int frames, int parts, int part, int* first, int* count);
int lanczos_partition_rows(const lanczos_desc* d, int parts, int part, int* out_row0, int* out_rows, int* in_row0,
                           int* in_rows);
int lanczos_multi_create(lanczos_multi** m, const int* devices, int n_devices);
int lanczos_multi_destroy(lanczos_multi* m);
int lanczos_multi_devices(const lanczos_multi* m);
/* Host buffers (`frames` whole frames back to back): one host thread per device, every device copies only its share over
 * its own PCIe link.  Results are those of lanczos_resample_host on one device. */
int lanczos_resample_multi_host(lanczos_multi* m, const lanczos_desc* d, const void* in, void* out, int frames, int split);
/* Frames resident on the ROOT device (devices[0]): scatter (RCCL ncclSend/ncclRecv group over xGMI) -> resample on every
 * device -> gather.  Synchronous.  compute_ms / total_ms may be NULL.  librccl is loaded on first use (n_devices > 1).
 * EXPERIMENTAL with more than one device: the exchange has not run on multi-GPU hardware yet (no such node was
 * available to the builders).  What is checked without it: the message lists (lanczos_multi_exchange_plan, below) against the
 * partition functions, and the group handling -- every ncclSend / ncclRecv code looked at, an opened group always closed, the
 * first failure reported as LANCZOS_ERR_RCCL with lanczos_multi_last_error().  The caller's current HIP device is preserved. */
int lanczos_resample_multi_root(lanczos_multi* m, const lanczos_desc* d, const void* d_in_root, void* d_out_root, int frames,
                                int split, double* compute_ms, double* total_ms);
/* What the last lanczos_resample_multi_root call saw: the HIP error, the ncclResult_t of the first failing RCCL call and which
 * message of the list it was (-1: ncclGroupStart / ncclCommInitAll, list size: ncclGroupEnd).  Pointers may be NULL. */
int lanczos_multi_last_error(const lanczos_multi* m, int* hip_error, int* rccl_error, int* rccl_message);
/* The root exchange as data (host only, no GPU): message k moves `bytes` from rank `src`'s buffer at src_off to rank `dst`'s
 * buffer at dst_off.  Rank 0's buffers are the caller's root buffers (all frames), a peer's buffer is its shard (split by
 * frames: its frames back to back; split by rows: its strip of every frame, frame after frame).  phase 0 = scatter of the inputs,
 * 1 = gather of the outputs.  Returns the number of messages (call with cap = 0 to size `out`), or -LANCZOS_ERR_*. */
typedef struct lanczos_xfer {
    int src, dst;
    size_t src_off, dst_off, bytes;
} lanczos_xfer;
int lanczos_multi_exchange_plan(const lanczos_desc* d, int frames, int split, int n_devices, int phase, lanczos_xfer* out, int cap);
/* What a ONE-GPU machine can execute of the RCCL path: loads librccl, builds a one-rank communicator on the context's first
 * device and moves `messages` self messages of `bytes` bytes each between two device buffers through the same group executor and
 * send / recv adapter as lanczos_resample_multi_root; the bytes are compared.  fail_at >= 0 makes message `fail_at` name a peer
 * that does not exist: the call then returns LANCZOS_ERR_RCCL and lanczos_multi_last_error() names that message (the group is
 * closed, the communicator proven usable afterwards).  LANCZOS_ERR_UNSUPPORTED: no librccl on this system.  (rccl.h:700,722) */
int lanczos_multi_exchange_selftest(lanczos_multi* m, int messages, size_t bytes, int fail_at);
/* Device memory for plain-C callers that do not include the HIP headers (host/main.c --root). */
int lanczos_device_alloc(int device, void** p, size_t bytes);
int lanczos_device_free(int device, void* p);
int lanczos_device_copy(int device, void* dst, const void* src, size_t bytes, int to_device);

/* ---- measurement / introspection ---- */
/* When enabled, every lanczos_resample_device call brackets its main kernel with HIP events on the
 * launch stream. lanczos_timing_read synchronises, returns and resets the sums. */
int lanczos_timing_enable(lanczos_ctx* ctx, int on);
int lanczos_timing_read(lanczos_ctx* ctx, int* launches, double* main_kernel_ms, double* prefix_kernel_ms);
int lanczos_last_kernel(const lanczos_ctx* ctx);
int lanczos_last_hip_error(const lanczos_ctx* ctx);
/* Force a kernel family for A/B tests: LANCZOS_KERNEL_NONE (auto), _GENERIC or _FAST. */
int lanczos_force_kernel(lanczos_ctx* ctx, int family);
const char* lanczos_strerror(int code);
const char* lanczos_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LANCZOS_HIP_H */
