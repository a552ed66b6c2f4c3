#!/usr/bin/env python3
"""bench.py -- output Mpixels/s of the Lanczos resample hot path on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c5|c1] [--shard frames|strips]
                    [--pattern gradient|noise|blocks|dark] [--mode lsb1|exact] [--frames F] [--regions R]
                    [--no-cpu-baseline] [--no-extras] [--exchange] [--settle-s SECONDS]

A "step" is ONE pass of the fused resample over a batch of F distinct synthetic frames that are already
resident in HBM (F = 32 by default for config 2; the steps cycle through R input / output sets whose inputs add up to more
than twice the 256 MiB Infinity Cache, so every timed step reads its frames from HBM -- see `rotate` below).

N > 1: `python bench.py --gpus N` starts its own N ranks (a child `python -m torch.distributed.run`, started
before this process touches the GPU); launched under torch.distributed.run by someone else (RANK/WORLD_SIZE in
the environment) it is a rank.  The work shards with NO data-path collective:
  --shard frames (default, BASELINE config 4): F frames per GPU per step, weak scaling;
  --shard strips (default for --config c5 at N > 1, BASELINE config 5): every frame is cut into N output row strips
    with an input halo (lanczos_strip_input_rows); rank r resamples strip r of each of the F frames: strong scaling.
Rank 0 prints one JSON line.

Timing: W untimed warm-up steps, then R regions (default 5) of EXACTLY K steps each, every region bracketed by a
barrier + torch.cuda.synchronize() on both sides and reduced with MAX over ranks; `ms_per_step` and `value` are the
MEDIAN region (median of >= 5 repeats); all regions are listed under "timing".

roofline (whole step, not one kernel): achieved = algorithmic bytes per step (input + output bytes of the F frames,
SURVEY.md 8d) / average device time of one step, measured with HIP events on the launch stream by the library
(lanczos_timing_*: from before the marching kernel to after the in-place-prefix kernel, so both kernels and the gap
between them are inside).  `kernel_us` / `prefix_kernel_us` split it.  `traffic` is the PMC figure
(profiles/traffic.json, written by scripts/round_profile.sh) only if it was collected for this exact workload AND
this exact kernel source (fingerprint of lanczos-hls_amd/csrc); otherwise null.

cpu_baseline = the reference's own software path (oracle/_ref, full_TB.h:29-96 compiled where it lay; kind
"reference") when that build travelled with the repo, else the CPU restatement (oracle/, kind "port"); timed on this
host on a bounded sample, rank 0, N = 1 only.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec

CONFIGS = {
    # name: (in_w, in_h, channels, bytes/sample, scale_n, scale_d, a, description)
    "c1": (256, 256, 3, 1, 2, 1, 2, "256x256->512x512 RGB8 2x Lanczos-2"),
    "c2": (1920, 1080, 3, 1, 2, 1, 3, "1920x1080->3840x2160 RGB8 2x Lanczos-3"),
    "c3": (1280, 720, 3, 1, 3, 1, 3, "1280x720->3840x2160 RGB8 3x Lanczos-3"),
    "c5": (3840, 2160, 4, 2, 2, 1, 4, "3840x2160->7680x4320 RGBA16 2x Lanczos-4"),
}


def csrc_fingerprint():
    """sha256 over the kernel sources: ties a committed PMC traffic figure to the code it was measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "lanczos-hls_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp", ".cpp")):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


# Frames per GPU per step.  SURVEY.md 8(d) asks for >= 16 distinct frame pairs per step (more than the 256 MiB Infinity
# Cache holds); the marching kernel cuts every (strip, frame) pair into floor(slots / pairs) chunks, and every chunk pays a
# prologue (two input ticks, the window rows) and every step a prefix launch: measured 16 frames 102-104 us per step,
# 32 frames 199 us (6.2 us per frame against 6.4), 8 frames 60 us.  The 16- and 8-frame steps are reported under
# other_batches.
DEFAULT_FRAMES = {"c1": 32, "c2": 32, "c3": 32, "c5": 8}


def make_frames(torch, pattern, frames, h, w, c, bps, device, seed):
    """Synthetic frames of SURVEY.md 8(d), generated on the device (torch RNG; not the checker's LCG)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    hi = 256 if bps == 1 else 65536
    dt = torch.uint8 if bps == 1 else torch.int32
    if pattern == "noise":
        x = torch.randint(0, hi, (frames, h, w, c), generator=gen, device=device, dtype=dt)
    elif pattern == "dark":
        x = torch.randint(0, hi * 72 // 256, (frames, h, w, c), generator=gen, device=device, dtype=dt)
    elif pattern == "gradient":
        yy = torch.arange(h, device=device).view(1, h, 1, 1) * (hi - 1) // h
        xx = torch.arange(w, device=device).view(1, 1, w, 1) * (hi - 1) // w
        nz = torch.randint(0, max(hi // 64, 1), (frames, h, w, c), generator=gen, device=device, dtype=torch.int32)
        x = ((yy + xx) // 2 + nz).clamp_(0, hi - 1)
    elif pattern == "blocks":
        yy = torch.arange(h, device=device).view(1, h, 1, 1) // 16
        xx = torch.arange(w, device=device).view(1, 1, w, 1) // 16
        cc = torch.arange(c, device=device).view(1, 1, 1, c)
        ff = torch.arange(frames, device=device).view(frames, 1, 1, 1)
        x = ((yy + xx + cc + ff) % 5) * (60 if bps == 1 else 15000)
    else:
        raise SystemExit(f"unknown pattern {pattern}")
    if bps == 1:
        return x.to(torch.uint8).contiguous()
    return x.to(torch.int32).to(torch.int16).contiguous()  # bit pattern of uint16


def cpu_baseline(frame_np, cfg, gpu_out_np):
    """Time the CPU side on this host.  The ONLY place bench.py touches oracle/ (as the thing timed beside the GPU,
    and as the checker of the GPU's first frame) -- never as the thing measured for `value`.
      * reference (oracle/_ref/ref_<shape>.so = full_TB.h:29-96 compiled by oracle/build_ref.sh in the build
        container): single thread, as the reference is written -- one whole frame
      * port (oracle/liblanczos_oracle.so, our restatement): all host cores, one whole frame"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O

    iw, ih, c, bps, sn, sd, a, _ = cfg
    ow, oh = iw * sn // sd, ih * sn // sd
    ocfg = O.cfg(iw, ih, ow, oh, c, a, sn, sd)
    fn = O.expected_hwc_u8 if bps == 1 else O.expected_hwc_u16
    ncores = os.cpu_count() or 1
    mpix = ow * oh / 1e6
    t0 = time.perf_counter()
    want = fn(ocfg, frame_np, ncores)
    t_all = time.perf_counter() - t0
    port_all = {"value": round(mpix / t_all, 3), "unit": "Mpix/s", "cores": ncores, "kind": "port",
                "sample": f"1 frame {iw}x{ih}->{ow}x{oh}, oracle/ (reference software path restated), "
                          f"{ncores} threads: {t_all:.2f} s"}
    res = None
    if bps == 1 and os.path.exists(O.ref_so_path(iw, ih, ow, oh, sn, sd, a, c)):
        # the reference's own code: planar in/out (full_TB.h:20-21), single thread
        planar = np.ascontiguousarray(frame_np.transpose(2, 0, 1))
        t0 = time.perf_counter()
        ref_out = O.ref_expected_planar_u8(ocfg, planar)
        t1 = time.perf_counter() - t0
        res = {"value": round(mpix / t1, 3), "unit": "Mpix/s", "cores": 1, "kind": "reference",
               "sample": f"1 frame {iw}x{ih}->{ow}x{oh}, lanczos_expected() of full_TB.h:79-96 compiled from the "
                         f"reference tree (oracle/_ref), single thread as written: {t1:.2f} s",
               "reference_equals_port": bool(np.array_equal(ref_out.transpose(1, 2, 0), want))}
    else:
        # bounded single-thread sample of the port: the top-left quarter-size crop (same scale, same a)
        qh, qw = max(ih // 2, 4 * a), max(iw // 2, 4 * a)
        qcfg = O.cfg(qw, qh, qw * sn // sd, qh * sn // sd, c, a, sn, sd)
        crop = np.ascontiguousarray(frame_np[:qh, :qw])
        t0 = time.perf_counter()
        fn(qcfg, crop, 1)
        t1 = time.perf_counter() - t0
        res = {"value": round(qcfg.out_w * qcfg.out_h / 1e6 / t1, 3), "unit": "Mpix/s", "cores": 1, "kind": "port",
               "sample": f"{qw}x{qh}->{qcfg.out_w}x{qcfg.out_h} crop of the same frame, oracle/ single thread: {t1:.2f} s"}
    res["all_cores_port"] = port_all
    diff = np.abs(want.astype(np.int64) - gpu_out_np.astype(np.int64))
    res["parity_vs_gpu"] = {"max_abs_diff": int(diff.max()), "mismatching_samples": int(np.count_nonzero(diff)),
                            "samples": int(diff.size)}
    return res


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--regions", type=int, default=5, help="timed regions of K steps each; the median is reported")
    ap.add_argument("--settle-s", type=float, default=0.25,
                    help="seconds of untimed launches before the warm-up steps (device clock ramp); 0 disables")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--shard", default="auto", choices=["auto", "frames", "strips"])
    ap.add_argument("--pattern", default="gradient",
                    help="headline input: gradient (natural-image-like, SURVEY.md 8d), noise (worst case for the "
                         "integer-phase fix-ups), blocks, dark; the others are reported under other_patterns")
    ap.add_argument("--mode", default="lsb1", choices=["lsb1", "exact"])
    ap.add_argument("--rotate", type=int, default=0,
                    help="distinct input/output batch sets the steps cycle through (default: as many as it takes for the inputs "
                         "touched between two uses of a set to exceed twice the 256 MiB Infinity Cache; 1 = the same batch every step)")
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU per step (default: 32, c5: 8)")
    ap.add_argument("--kernel", default="auto", choices=["auto", "generic", "fast"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip other_patterns / host path / c5 strip leg (profiling runs: only the headline launches)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the multi-rank path)")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--planar-path", action="store_true",
                    help="also time planar-in / planar-out frames (device layout conversion either side), reported separately")
    ap.add_argument("--exchange", action="store_true",
                    help="also time root scatter/gather of the frames over RCCL (N > 1), reported separately")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` with no rank environment: start N ranks as a CHILD process tree.  Nothing in this
    process has touched the GPU (no torch import, no HIP call), and it does not exec: it waits and passes the code on."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = os.environ.copy()
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class Workload:
    """One rank's share of a step: device buffers + the resample call (through the C ABI)."""

    def __init__(self, torch, L, ctx, cfg, frames, mode, pattern, device, rank, world, shard, seed, rotate=1):
        iw, ih, c, bps, sn, sd, a, _ = cfg
        self.torch, self.ctx, self.cfg, self.frames = torch, ctx, cfg, frames
        self.full = L.make_desc(iw, ih, c, sn, sd, a, bps, mode)
        self.shard = shard
        self.rotate, self.n, self.cur = max(1, rotate), 0, 0
        x_full = make_frames(torch, pattern, frames, ih, iw, c, bps, device, seed)
        if shard == "strips" and world > 1:
            import lanczos_hls_amd.sharding as sh
            K = L.inplace_rows(self.full)
            shards = sh.strip_shards(self.full.out_h, world, lambda r0, n: L.strip_input_rows(self.full, r0, n),
                                     min_first=K + 2 * a + 2)
            self.row0, self.rows, self.in0, self.in_n = shards[rank]
            self.desc = L.make_desc(iw, ih, c, sn, sd, a, bps, mode, out_row0=self.row0, out_rows=self.rows)
            # (every rank generates the same frames from the same seed and keeps its strip + halo: no scatter needed)
            self.x = x_full[:, self.in0:self.in0 + self.in_n].contiguous()
            del x_full
            self.y = torch.empty((frames, self.rows, self.full.out_w, c), device=device, dtype=self.x.dtype)
        else:
            self.desc = self.full
            self.row0, self.rows = 0, self.full.out_h
            self.x = x_full
            self.y = torch.empty((frames, self.full.out_h, self.full.out_w, c), device=device, dtype=self.x.dtype)
        # input / output sets the steps cycle through (set 0 = the tensors above; see --rotate)
        self.xs, self.ys = [self.x], [self.y]
        for i in range(1, self.rotate):
            xi = make_frames(torch, pattern, frames, ih, iw, c, bps, device, seed + 1000 * i)
            if self.desc is not self.full:
                xi = xi[:, self.in0:self.in0 + self.in_n].contiguous()
            self.xs.append(xi)
            self.ys.append(torch.empty_like(self.y))
        self.stream = torch.cuda.current_stream().cuda_stream
        self.out_pix = frames * self.rows * self.full.out_w                     # this rank, per step
        self.alg_bytes = (self.x.numel() + self.y.numel()) * bps               # this rank, per step (halo included)

    def step(self):
        i = self.n % self.rotate
        self.n += 1
        self.x, self.y = self.xs[i], self.ys[i]   # the set this step reads / writes (what the parity check looks at)
        self.ctx.resample_device(self.desc, self.x.data_ptr(), self.y.data_ptr(), self.frames, 0, 0, self.stream)

    def refill(self, pattern, seed):
        iw, ih, c, bps = self.cfg[0], self.cfg[1], self.cfg[2], self.cfg[3]
        for i in range(self.rotate):
            x_full = make_frames(self.torch, pattern, self.frames, ih, iw, c, bps, self.xs[i].device, seed + 1000 * i)
            if self.desc is not self.full:
                x_full = x_full[:, self.in0:self.in0 + self.in_n]
            self.xs[i].copy_(x_full)


def device_batch_time(torch, wl, n):
    """Device time of n back-to-back steps, from ONE pair of HIP events on the launch stream (torch's current stream is
    the stream the library launches on): every kernel of every step and the gaps between them, nothing else."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        wl.step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 1e3 / n


def device_step_times(ctx, torch, wl, n):
    """Average device time of one step's two kernels (HIP events around every kernel, recorded by the library on the
    launch stream).  The per-step event records themselves open gaps of a few microseconds: this pass only SPLITS the
    step; the step's own time comes from device_batch_time."""
    torch.cuda.synchronize()
    ctx.timing_enable(True)
    ctx.timing_read()  # reset
    for _ in range(n):
        wl.step()
    torch.cuda.synchronize()
    launches, main_ms, prefix_ms = ctx.timing_read()
    ctx.timing_enable(False)
    launches = max(launches, 1)
    return main_ms / launches / 1e3, prefix_ms / launches / 1e3, launches


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    import torch
    import lanczos_hls_amd as L

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU fallback)")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
    red_dev = device if args.backend == "nccl" else torch.device("cpu")  # where the scalar reductions live

    cfg = CONFIGS[args.config]
    iw, ih, c, bps, sn, sd, a, desc_txt = cfg
    frames = args.frames or DEFAULT_FRAMES[args.config]
    mode = L.MODE_EXACT if args.mode == "exact" else L.MODE_LSB1
    shard = args.shard
    if shard == "auto":
        shard = "strips" if (args.config == "c5" and world > 1) else "frames"
    ctx = L.Context(local_rank)
    ctx.force_kernel({"auto": L.KERNEL_NONE, "generic": L.KERNEL_GENERIC, "fast": L.KERNEL_FAST}[args.kernel])
    # frame-sharded ranks hold different frames (seed + rank); strip-sharded ranks cut the SAME frames
    seed = 1234 + (rank if shard == "frames" else 0)
    # The Infinity Cache (256 MiB, memory side) keeps a batch's INPUT resident from one step to the next when every step
    # reads the same frames (the output is stored non-temporally and does not displace it): measured 16 frames 110 us per
    # step with one or two input sets (100 / 200 MB) against 121 us with four.  The steps therefore cycle through R sets
    # whose inputs add up to more than twice that cache: every timed step reads its frames from HBM.
    in_bytes = frames * iw * ih * c * bps
    rotate = args.rotate or min(8, max(1, -(-2 * 256 * 2**20 // in_bytes)))
    wl = Workload(torch, L, ctx, cfg, frames, mode, args.pattern, device, rank, world, shard, seed, rotate)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(v):
        if dist is None:
            return v
        t = torch.tensor([v], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(v):
        if dist is None:
            return v
        t = torch.tensor([v], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def timed_region(w, steps):
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            w.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        return max_over_ranks(dt)

    # Untimed: let the device reach its steady clocks before the W warm-up steps.  A fresh box runs the first
    # ~30 ms of work 10-15 % slower (measured: 150 us per launch with 5 warm-up launches, 132 us after 200).  The same settling
    # precedes every side measurement below (other patterns / batches / configs): timed right behind a refill with three warm
    # steps, "blocks" read 236-242 us per step where the interleaved A/B harness (30 warm steps) reads 213.
    def settle(w, seconds):
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < seconds:
            for _ in range(20):
                w.step()
            torch.cuda.synchronize()

    settle(wl, args.settle_s)
    for _ in range(args.warmup):
        wl.step()

    # Pass A (never part of `value`): K steps between one pair of HIP events on the launch stream -> roofline; then K steps
    # with the library's per-kernel events -> how the step splits into its kernels.
    step_s = device_batch_time(torch, wl, args.steps)
    main_s, prefix_s, launches = device_step_times(ctx, torch, wl, args.steps)

    # Pass B (the timed regions): R x exactly K steps, no instrumentation.  (Replaying one captured step as a HIP
    # graph was measured too: 0.144-0.150 ms per step against 0.114 ms of plain launches.)
    regions = [timed_region(wl, args.steps) for _ in range(max(args.regions, 1))]
    elapsed = sorted(regions)[len(regions) // 2]

    total_pix_step = sum_over_ranks(float(wl.out_pix))
    value = total_pix_step * args.steps / elapsed / 1e6
    achieved = wl.alg_bytes / step_s / 1e9 if step_s > 0 else 0.0   # whole step: every kernel + the gaps between them

    extra = {}
    if not args.no_extras:
        # the other synthetic generators of SURVEY.md 8(d), same launch shape, fewer steps, WHOLE-step device time
        others = {}
        for pat in ("noise", "gradient", "blocks"):
            if pat == args.pattern:
                continue
            wl.refill(pat, 4321 + (rank if shard == "frames" else 0))
            settle(wl, min(args.settle_s, 0.1))
            s_o = sorted(device_batch_time(torch, wl, max(10, args.steps // 2)) for _ in range(3))[1]
            others[pat] = {"step_us": round(s_o * 1e6, 2),
                           "roofline_frac": round(wl.alg_bytes / s_o / 1e9 / HBM_PEAK_GBS, 4)}
        extra["other_patterns"] = others
        if args.mode == "lsb1":
            # LANCZOS_MODE_EXACT -- what lanczos_u8() and the literal lanczos(stream_t, stream_t) drop-in always run: the same
            # frames, the same launch shape, bit-identical results, whole-step device time
            exact = {}
            desc_keep, full_keep = wl.desc, wl.full
            if wl.desc is wl.full:
                wl.desc = wl.full = L.make_desc(cfg[0], cfg[1], cfg[2], cfg[4], cfg[5], cfg[6], cfg[3], L.MODE_EXACT)
            else:   # a row strip of the frame (--config c5 at N > 1)
                wl.desc = L.make_desc(cfg[0], cfg[1], cfg[2], cfg[4], cfg[5], cfg[6], cfg[3], L.MODE_EXACT,
                                      out_row0=wl.row0, out_rows=wl.rows)
            for pat in ("gradient", "noise"):
                wl.refill(pat, 4321 + (rank if shard == "frames" else 0))
                settle(wl, min(args.settle_s, 0.1))
                s_o = sorted(device_batch_time(torch, wl, max(10, args.steps // 2)) for _ in range(3))[1]
                exact[pat] = {"step_us": round(s_o * 1e6, 2),
                              "roofline_frac": round(wl.alg_bytes / s_o / 1e9 / HBM_PEAK_GBS, 4)}
            wl.desc, wl.full = desc_keep, full_keep
            extra["other_modes"] = {"exact": exact}
        wl.refill(args.pattern, seed)
        wl.step()
        torch.cuda.synchronize()
        if shard == "frames":
            # smaller batches of the same workload (views of the same frames): whole-step device time per batch size
            batches = {}
            full_frames = wl.frames
            for f in (full_frames // 2, full_frames // 4):
                if f < 1:
                    continue
                wl.frames = f
                settle(wl, min(args.settle_s, 0.05))
                s_b = sorted(device_batch_time(torch, wl, max(10, args.steps // 2)) for _ in range(3))[1]
                batches[str(f)] = {"step_us": round(s_b * 1e6, 2),
                                   "roofline_frac": round(wl.alg_bytes * f / full_frames / s_b / 1e9 / HBM_PEAK_GBS, 4)}
            wl.frames = full_frames
            extra["other_batches"] = batches
        if wl.rotate > 1:
            # the same batch every step (what rounds 1 and 2a measured): its input stays in the Infinity Cache
            r_keep = wl.rotate
            wl.rotate = 1
            for _ in range(3):
                wl.step()
            s_1 = device_batch_time(torch, wl, max(10, args.steps // 2))
            wl.rotate = r_keep
            extra["same_batch_every_step"] = {"step_us": round(s_1 * 1e6, 2),
                                              "roofline_frac": round(wl.alg_bytes / s_1 / 1e9 / HBM_PEAK_GBS, 4),
                                              "note": "input resident in the 256 MiB Infinity Cache between steps; not the headline"}

    if rank == 0 and world == 1 and not args.no_extras and args.config == "c2":
        # BASELINE configs 3 and 5 beside the headline (whole-step device time, inputs cycled past the Infinity Cache like the
        # headline's), and the headline's workload at twice the batch: same kernels, same method, fewer steps
        others_cfg = {}
        for name, f_o in (("c3", DEFAULT_FRAMES["c3"]), ("c5", DEFAULT_FRAMES["c5"]), ("c2", 2 * frames)):
            cfg_o = CONFIGS[name]
            in_b = f_o * cfg_o[0] * cfg_o[1] * cfg_o[2] * cfg_o[3]
            rot_o = min(8, max(1, -(-2 * 256 * 2**20 // in_b)))
            w_o = Workload(torch, L, ctx, cfg_o, f_o, mode, args.pattern, device, rank, world, "frames", 4000, rot_o)
            settle(w_o, min(args.settle_s, 0.1))
            n_o = max(10, args.steps // 2)
            s_o = sorted(device_batch_time(torch, w_o, n_o) for _ in range(3))[1]
            others_cfg[f"{name}_{f_o}_frames"] = {
                "workload": cfg_o[7], "frames_per_step": f_o, "batch_sets_cycled": rot_o, "step_us": round(s_o * 1e6, 2),
                "algorithmic_bytes_per_step": w_o.alg_bytes, "roofline_frac": round(w_o.alg_bytes / s_o / 1e9 / HBM_PEAK_GBS, 4),
                "Mpix_per_s": round(w_o.out_pix / s_o / 1e6, 1), "kernel": {1: "generic", 2: "fast"}.get(ctx.last_kernel(), "?")}
            del w_o
            torch.cuda.empty_cache()
        extra["other_configs"] = others_cfg

    if args.exchange and dist is not None and shard == "frames":
        # root-inclusive figure: rank 0 scatters every rank's input frames and gathers the outputs (RCCL)
        reps = 3
        xin = [torch.empty_like(wl.x) for _ in range(world)] if rank == 0 else None
        yout = [torch.empty_like(wl.y) for _ in range(world)] if rank == 0 else None
        sync_all()
        t0 = time.perf_counter()
        for _ in range(reps):
            dist.scatter(wl.x, xin, src=0)
            wl.step()
            dist.gather(wl.y, yout, dst=0)
        torch.cuda.synchronize()
        dt = max_over_ranks(time.perf_counter() - t0)
        extra["root_scatter_gather"] = {"value": round(total_pix_step * reps / dt / 1e6, 1),
                                        "unit": "Mpix/s", "note": "rank 0 scatters inputs / gathers outputs over RCCL"}

    if dist is not None and shard == "strips":
        # config 5, root-inclusive: every rank sends its output strips to rank 0 (one exchange step)
        reps = 3
        yout = None
        if rank == 0:
            yout = [torch.empty((frames, r, wl.full.out_w, c), device=device, dtype=wl.y.dtype) for r in
                    [int(v) for v in _all_rows(dist, torch, wl.rows, world, red_dev)]]
        else:
            _all_rows(dist, torch, wl.rows, world, red_dev)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(reps):
            wl.step()
            if rank == 0:
                reqs = [dist.irecv(yout[r], src=r) for r in range(1, world)]
                for q in reqs:
                    q.wait()
            else:
                dist.send(wl.y, dst=0)
        torch.cuda.synchronize()
        dt = max_over_ranks(time.perf_counter() - t0)
        extra["root_gather_inclusive"] = {"value": round(total_pix_step * reps / dt / 1e6, 1), "unit": "Mpix/s",
                                          "note": "compute + every rank's output strips sent to rank 0 (RCCL send/recv)"}

    if dist is not None and args.config != "c5" and not args.no_extras:
        # BASELINE config 5 beside the headline: ONE 8K RGBA16 frame set cut into N row strips (tile sharding)
        c5 = CONFIGS["c5"]
        f5 = 2
        w5 = Workload(torch, L, ctx, c5, f5, mode, args.pattern, device, rank, world, "strips", 777)
        for _ in range(3):
            w5.step()
        r5 = sorted(timed_region(w5, 10) for _ in range(3))[1]
        tot5 = sum_over_ranks(float(w5.out_pix))
        s5 = device_batch_time(torch, w5, 10)
        bytes5 = sum_over_ranks(float(w5.alg_bytes))
        extra["c5_row_strips"] = {
            "workload": c5[7] + f", {f5} frames, each cut into {world} output row strips (+ input halo), one per GPU",
            "value": round(tot5 * 10 / r5 / 1e6, 1), "unit": "Mpix/s", "ms_per_step": round(r5 / 10 * 1e3, 4),
            "scaling": "strong", "rank0_step_us": round(s5 * 1e6, 2),
            "algorithmic_bytes_per_step_all_ranks": int(bytes5)}
        del w5

    if args.planar_path and rank == 0:
        # Callers that hold the reference's planar img[C][H][W] arrays (full_TB.h:20-21): planar -> interleaved,
        # resample, interleaved -> planar, all on the device.  Reported beside the headline, never `value`.
        d = wl.full
        x, y, stream = wl.x, wl.y, wl.stream
        xp = x.permute(0, 3, 1, 2).contiguous()
        yp = torch.empty((frames, c, d.out_h, d.out_w), device=device, dtype=x.dtype)
        for _ in range(3):
            ctx.resample_planar_device(d, xp.data_ptr(), yp.data_ptr(), frames, stream)
        torch.cuda.synchronize()
        reps = max(5, args.steps // 2)
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.resample_planar_device(d, xp.data_ptr(), yp.data_ptr(), frames, stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        extra["planar_path"] = {"value": round(frames * d.out_w * d.out_h / dt / 1e6, 1), "unit": "Mpix/s",
                                "ms_per_batch": round(dt * 1e3, 4),
                                "note": "planar in -> planar out on the device",
                                "equals_interleaved_result": bool(torch.equal(yp, y.permute(0, 3, 1, 2)))}

    if rank == 0 and world == 1 and not args.no_extras:
        # PCIe-inclusive rate: host buffers in, host buffers out, through the pipelined lanczos_resample_host.
        # Never `value` (inputs must be resident in HBM for that); DESIGN.md quotes this figure.
        d = wl.full
        np_dt = "uint8" if bps == 1 else "uint16"
        pin_in = L.PinnedArray((frames, ih, iw, c), np_dt)
        pin_out = L.PinnedArray((frames, d.out_h, d.out_w, c), np_dt)
        src = wl.x.cpu().numpy()
        pin_in.array[...] = src if bps == 1 else src.view("uint16")
        ctx.resample(pin_in.array, sn, sd, a, mode, out=pin_out.array)  # warm-up
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.resample(pin_in.array, sn, sd, a, mode, out=pin_out.array)
        dt = (time.perf_counter() - t0) / reps
        extra["host_path_pinned"] = {"value": round(frames * d.out_w * d.out_h / dt / 1e6, 1), "unit": "Mpix/s",
                                     "ms_per_batch": round(dt * 1e3, 3), "frames": frames,
                                     "note": "PCIe-inclusive: H2D + resample + D2H, three streams, page-locked buffers"}
        pin_in.close()
        pin_out.close()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import numpy as np
        wl.step()
        torch.cuda.synchronize()
        f0 = wl.x[0].cpu().numpy()
        g0 = wl.y[0].cpu().numpy()
        if bps == 2:
            f0, g0 = f0.view(np.uint16), g0.view(np.uint16)
        cpu = cpu_baseline(f0, cfg, g0)

    # HBM traffic per step from the PMC counters: collected by scripts/round_profile.sh (separate --pmc passes,
    # FETCH_SIZE doubled per MI355X_MICROARCH.md) and committed under profiles/.  bench.py cannot run the profiler on
    # itself, so it reports that figure ONLY for this exact workload and this exact kernel source; otherwise null.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            tj = json.load(f)
        ent = tj.get(f"{args.config}:{frames}:{args.pattern}:{args.mode}")
        if ent and ent.get("csrc_fingerprint") == csrc_fingerprint() and world == 1:
            traffic = ent["total_bytes_per_step"]
    except Exception:
        traffic = None

    if rank == 0:
        par = (f"{frames} frames per GPU per step over {world} GPU(s)" if shard == "frames" else
               f"{frames} frames per step, each cut into {world} output row strips with an input halo, one strip per GPU")
        line = {
            "metric": "output Mpixels/s", "value": round(value, 1), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak" if shard == "frames" else "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc_txt, "frames_per_step_per_gpu": frames, "batch_sets_cycled": wl.rotate, "pattern": args.pattern,
                       "parity_mode": args.mode, "kernel": {1: "generic", 2: "fast"}.get(ctx.last_kernel(), "?"),
                       "settle_s_untimed": args.settle_s, "parallelism": par + ", no data-path collective"},
            "timing": {"regions": len(regions), "steps_per_region": args.steps, "statistic": "median",
                       "ms_per_step_all_regions": [round(r / args.steps * 1e3, 4) for r in regions]},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "scope": "whole step on rank 0 (every kernel of the step + the gaps), one HIP-event pair around K steps "
                                  "on the launch stream; kernel_us / prefix_kernel_us: per-kernel events of a second pass",
                         "step_us": round(step_s * 1e6, 2), "kernel_us": round(main_s * 1e6, 2),
                         "prefix_kernel_us": round(prefix_s * 1e6, 2),
                         "frac_from_ms_per_step": round(wl.alg_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                         "algorithmic_bytes_per_step": wl.alg_bytes, "steps_timed": launches},
            "cpu_baseline": cpu,
        }
        line.update(extra)
        if "same_batch_every_step" in extra:   # rounds 1 / 2a quoted this figure (inputs served by the Infinity Cache)
            line["roofline"]["frac_same_batch_every_step"] = extra["same_batch_every_step"]["roofline_frac"]
        print(json.dumps(line), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


def _all_rows(dist, torch, rows, world, red_dev):
    t = torch.zeros(world, dtype=torch.int64, device=red_dev)
    t[dist.get_rank()] = rows
    dist.all_reduce(t)
    return t.tolist()


if __name__ == "__main__":
    main()
