#!/usr/bin/env python3
"""bench.py -- output Mpixels/s of the Lanczos resample hot path on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c5|c1] [--pattern noise|gradient|blocks|dark]
                    [--mode lsb1|exact] [--frames F] [--no-cpu-baseline] [--exchange] [--settle-s SECONDS]

A "step" is ONE launch of the fused resample over a batch of F distinct synthetic frames that are already
resident in HBM (F = 16 by default: 16 x 31.1 MB of compulsory traffic > the 256 MiB Infinity Cache, so
steps that cycle through the same buffers still stream from HBM).  Frames shard across ranks with no
data-path collective (weak scaling: F frames per GPU per step).  Rank 0 prints one JSON line.

roofline.achieved = algorithmic bytes per launch (input + output bytes of the F frames)
divided by the main kernel's average duration, measured with HIP events on the launch stream by the
library (lanczos_timing_*).  cpu_baseline = the CPU restatement of the reference software path
(oracle/, full_TB.h:29-96) timed on this host's cores on a bounded sample, rank 0, N = 1 only.
"""
import argparse
import json
import os
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
    """Time the CPU checker (oracle/ = restatement of full_TB.h:29-96) on this host.  The ONLY place bench.py
    touches oracle/.  Sample: one frame single-threaded (what the reference does), then one frame on all cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O

    iw, ih, c, bps, sn, sd, a, _ = cfg
    ocfg = O.cfg(iw, ih, iw * sn // sd, ih * sn // sd, c, a, sn, sd)
    fn = O.expected_hwc_u8 if bps == 1 else O.expected_hwc_u16
    ncores = os.cpu_count() or 1
    mpix = ocfg.out_w * ocfg.out_h / 1e6
    t0 = time.perf_counter()
    want = fn(ocfg, frame_np, ncores)
    t_all = time.perf_counter() - t0
    res = {"value": round(mpix / t_all, 3), "unit": "Mpix/s", "cores": ncores, "kind": "port",
           "sample": f"1 frame {iw}x{ih}->{ocfg.out_w}x{ocfg.out_h}, oracle/ (reference software path restated), "
                     f"{ncores} threads: {t_all:.2f} s"}
    # single thread = the reference as written (no threading anywhere in it).  Bounded sample: the top-left
    # quarter-size crop of the same frame (same scale, same a; Mpix/s does not depend on the frame size)
    qh, qw = max(ih // 2, 4 * a), max(iw // 2, 4 * a)
    qcfg = O.cfg(qw, qh, qw * sn // sd, qh * sn // sd, c, a, sn, sd)
    crop = np.ascontiguousarray(frame_np[:qh, :qw])
    t0 = time.perf_counter()
    fn(qcfg, crop, 1)
    t1 = time.perf_counter() - t0
    res["single_thread"] = {"value": round(qcfg.out_w * qcfg.out_h / 1e6 / t1, 3), "unit": "Mpix/s", "cores": 1,
                            "sample": f"{qw}x{qh}->{qcfg.out_w}x{qcfg.out_h} crop of the same frame: {t1:.2f} s"}
    diff = np.abs(want.astype(np.int64) - gpu_out_np.astype(np.int64))
    res["parity_vs_gpu"] = {"max_abs_diff": int(diff.max()), "mismatching_samples": int(np.count_nonzero(diff)),
                            "samples": int(diff.size)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--settle-s", type=float, default=0.25,
                    help="seconds of untimed launches before the warm-up steps (device clock ramp); 0 disables")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--pattern", default="gradient",
                    help="headline input: gradient (natural-image-like, SURVEY.md 8d), noise (worst case for the "
                         "integer-phase fix-ups), blocks, dark; the others are reported under other_patterns")
    ap.add_argument("--mode", default="lsb1", choices=["lsb1", "exact"])
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU per step (default: 16, c5: 4)")
    ap.add_argument("--kernel", default="auto", choices=["auto", "generic", "fast"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the multi-rank path)")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--planar-path", action="store_true",
                    help="also time planar-in / planar-out frames (device layout conversion either side), reported separately")
    ap.add_argument("--host-path", action="store_true",
                    help="also time lanczos_resample_host (PCIe copies included, page-locked buffers), reported separately")
    ap.add_argument("--exchange", action="store_true",
                    help="also time root scatter/gather of the frames over RCCL (N > 1), reported separately")
    args = ap.parse_args()

    import torch
    import lanczos_hls_amd as L

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
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
    frames = args.frames or (4 if args.config == "c5" else 16)
    mode = L.MODE_EXACT if args.mode == "exact" else L.MODE_LSB1
    d = L.make_desc(iw, ih, c, sn, sd, a, bps, mode)
    ctx = L.Context(local_rank)
    ctx.force_kernel({"auto": L.KERNEL_NONE, "generic": L.KERNEL_GENERIC, "fast": L.KERNEL_FAST}[args.kernel])

    x = make_frames(torch, args.pattern, frames, ih, iw, c, bps, device, seed=1234 + rank)
    y = torch.empty((frames, d.out_h, d.out_w, c), device=device, dtype=x.dtype)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        ctx.resample_device(d, x.data_ptr(), y.data_ptr(), frames, 0, 0, stream)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # Untimed: let the device reach its steady clocks before the W warm-up steps.  A fresh box runs the first
    # ~30 ms of work 10-15 % slower (measured: 150 us per launch with 5 warm-up launches, 132 us after 200);
    # the timed region below is still exactly K steps.
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.settle_s:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    # Pass A (untimed for `value`): the same K steps with the library's HIP events around every kernel -> roofline.
    sync_all()
    ctx.timing_enable(True)
    ctx.timing_read()  # reset
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    launches, main_ms, prefix_ms = ctx.timing_read()
    ctx.timing_enable(False)

    # Pass B (the timed region): exactly K steps, no instrumentation, barrier + synchronize on both sides.  (Replaying
    # one captured step as a HIP graph was measured too: 0.144-0.150 ms per step against 0.114 ms of plain launches.)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out_pix_step = frames * d.out_w * d.out_h  # per GPU
    value = world * out_pix_step * args.steps / elapsed / 1e6
    alg_bytes_launch = frames * (iw * ih + d.out_w * d.out_h) * c * bps
    avg_main_s = main_ms / max(launches, 1) / 1e3
    achieved = alg_bytes_launch / avg_main_s / 1e9 if avg_main_s > 0 else 0.0

    extra = {}
    # the other synthetic generators of SURVEY.md 8(d), same launch shape, fewer steps -- reported beside the headline
    others = {}
    for pat in ("noise", "gradient", "blocks"):
        if pat == args.pattern:
            continue
        x.copy_(make_frames(torch, pat, frames, ih, iw, c, bps, device, seed=4321 + rank))
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        ctx.timing_enable(True)
        ctx.timing_read()
        n_o = max(3, args.steps // 5)
        for _ in range(n_o):
            step()
        torch.cuda.synchronize()
        l_o, m_o, _p = ctx.timing_read()
        ctx.timing_enable(False)
        k_s = m_o / max(l_o, 1) / 1e3
        others[pat] = {"kernel_us": round(k_s * 1e6, 2), "roofline_frac": round(alg_bytes_launch / k_s / 1e9 / HBM_PEAK_GBS, 4)}
    extra["other_patterns"] = others
    x.copy_(make_frames(torch, args.pattern, frames, ih, iw, c, bps, device, seed=1234 + rank))
    step()
    torch.cuda.synchronize()
    if args.exchange and dist is not None:
        # root-inclusive figure: rank 0 scatters every rank's input frames and gathers the outputs (RCCL)
        reps = 3
        xin = [torch.empty_like(x) for _ in range(world)] if rank == 0 else None
        yout = [torch.empty_like(y) for _ in range(world)] if rank == 0 else None
        sync_all()
        t0 = time.perf_counter()
        for _ in range(reps):
            dist.scatter(x, xin, src=0)
            step()
            dist.gather(y, yout, dst=0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        extra["root_scatter_gather"] = {"value": round(world * out_pix_step * reps / float(t.item()) / 1e6, 1),
                                        "unit": "Mpix/s", "note": "rank 0 scatters inputs / gathers outputs over RCCL"}

    if args.planar_path and rank == 0:
        # Callers that hold the reference's planar img[C][H][W] arrays (full_TB.h:20-21): planar -> interleaved,
        # resample, interleaved -> planar, all on the device.  Reported beside the headline, never `value`.
        xp = x.permute(0, 3, 1, 2).contiguous()
        yp = torch.empty((frames, c, d.out_h, d.out_w), device=device, dtype=x.dtype)
        yi = torch.empty_like(y)
        for _ in range(3):
            ctx.resample_planar_device(d, xp.data_ptr(), yp.data_ptr(), frames, stream)
        torch.cuda.synchronize()
        reps = max(5, args.steps // 2)
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.resample_planar_device(d, xp.data_ptr(), yp.data_ptr(), frames, stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)  # torch's current stream = `stream`
        e0.record()
        for _ in range(reps):
            ctx.interleaved_to_planar_device(y.data_ptr(), yp.data_ptr(), d.out_w, d.out_h, c, bps, frames, stream)
        e1.record()
        torch.cuda.synchronize()
        t_out = e0.elapsed_time(e1) / reps / 1e3
        e0.record()
        for _ in range(reps):
            ctx.planar_to_interleaved_device(xp.data_ptr(), yi.data_ptr(), iw, ih, c, bps, frames, stream)
        e1.record()
        torch.cuda.synchronize()
        t_in = e0.elapsed_time(e1) / reps / 1e3
        ob, ib = frames * d.out_w * d.out_h * c * bps, frames * iw * ih * c * bps
        extra["planar_path"] = {"value": round(frames * d.out_w * d.out_h / dt / 1e6, 1), "unit": "Mpix/s",
                                "ms_per_batch": round(dt * 1e3, 4),
                                "interleaved_to_planar_GBps": round(2 * ob / t_out / 1e9, 1),
                                "planar_to_interleaved_GBps": round(2 * ib / t_in / 1e9, 1),
                                "note": "planar in -> planar out on the device; GB/s = bytes read + written per second"}
        ok = torch.equal(yp, y.permute(0, 3, 1, 2)) if True else None
        extra["planar_path"]["equals_interleaved_result"] = bool(ok)

    if args.host_path and rank == 0:
        # PCIe-inclusive rate: host buffers in, host buffers out, through the pipelined lanczos_resample_host.
        # Never `value` (inputs must be resident in HBM for that); DESIGN.md quotes this figure.
        pin_in = L.PinnedArray((frames, ih, iw, c), x.cpu().numpy().dtype if bps == 1 else "uint16")
        pin_out = L.PinnedArray((frames, d.out_h, d.out_w, c), pin_in.dtype)
        src = x.cpu().numpy()
        pin_in.array[...] = src if bps == 1 else src.view("uint16")
        ctx.resample(pin_in.array, sn, sd, a, mode, out=pin_out.array)  # warm-up
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            ctx.resample(pin_in.array, sn, sd, a, mode, out=pin_out.array)
        dt = (time.perf_counter() - t0) / reps
        extra["host_path_pinned"] = {"value": round(frames * d.out_w * d.out_h / dt / 1e6, 1), "unit": "Mpix/s",
                                     "ms_per_batch": round(dt * 1e3, 3), "frames": frames,
                                     "note": "H2D + resample + D2H, three streams, page-locked buffers"}
        pin_in.close()
        pin_out.close()
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import numpy as np
        f0 = x[0].cpu().numpy()
        g0 = y[0].cpu().numpy()
        if bps == 2:
            f0, g0 = f0.view(np.uint16), g0.view(np.uint16)
        cpu = cpu_baseline(f0, cfg, g0)

    # HBM traffic per launch from the PMC counters (collected by scripts/round_profile.sh in separate --pmc
    # passes, FETCH_SIZE doubled per MI355X_MICROARCH.md; summary committed under profiles/): bench.py itself
    # cannot run the profiler, so it reports the committed figure for this exact workload or null
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            tj = json.load(f)
        key = f"{args.config}:{frames}"
        if key in tj:
            traffic = tj[key]["total_bytes_per_launch"]
    except Exception:
        traffic = None
    if rank == 0:
        line = {
            "metric": "output Mpixels/s", "value": round(value, 1), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc_txt, "frames_per_gpu_per_step": frames, "pattern": args.pattern,
                       "parity_mode": args.mode, "kernel": {1: "generic", 2: "fast"}.get(ctx.last_kernel(), "?"),
                       "settle_s_untimed": args.settle_s,
                       "parallelism": f"frames sharded over {world} GPU(s), no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel_us": round(avg_main_s * 1e6, 2), "prefix_kernel_us":
                             round(prefix_ms / max(launches, 1) * 1e3, 2),
                         "algorithmic_bytes_per_launch": alg_bytes_launch, "launches_timed": launches},
            "cpu_baseline": cpu,
        }
        line.update(extra)
        print(json.dumps(line), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
