#!/usr/bin/env python3
"""Headline benchmark: deferred shading with all 21 brazier lights of the
SunTemple-proxy scene, whole frames (transform, setup, binning, rasterise,
G-buffer resolve, lighting), output left in device memory.

  python bench.py --gpus N --steps K --warmup W

A step is one frame for one camera pose.  With N > 1 (launched through
torch.distributed.run, one rank per GPU) the camera poses of a fly-through are
partitioned over the ranks (scene replicated per GPU, no data-path
collective), so per-GPU work is fixed: weak scaling.  `--partition sortfirst`
instead splits every frame into N horizontal bands and gathers them on rank 0
over RCCL (strong scaling).

Rank 0 prints one JSON line.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

HBM_PEAK_GBPS = 8000.0       # MI355X HBM3E peak
FUSED_BYTES_PER_PIXEL = 12   # k_resolve_light: 8 (visibility buffer) read + 4 (RGBA8) written; texels and per-triangle data excluded
LIGHT_BYTES_PER_PIXEL = 24   # 8 (normal+metal RGBA16F) + 8 (albedo+rough RGBA16F) + 4 (D32) read + 4 (RGBA8) written


def fly_pose(i, n=512):
    """Pose i of the fly-through: translate(0,-1,-90*i/(n-1)) * rotateY(0.35*sin(2*pi*i/128)), glm column-major."""
    a = 0.35 * math.sin(2.0 * math.pi * i / 128.0)
    c, s = math.cos(a), math.sin(a)
    tz = -90.0 * i / (n - 1)
    # columns: X=(c,0,-s), Y=(0,1,0), Z=(s,0,c), T=(0,-1,tz)
    return [c, 0.0, -s, 0.0, 0.0, 1.0, 0.0, 0.0, s, 0.0, c, 0.0, 0.0, -1.0, tz, 1.0]


def measured_traffic(width, height, kernel="k_deferred_light"):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary (profiles/*pmc_traffic*.json) taken at
    this extent, or None.  PMC counters need their own rocprofv3 passes, so they cannot be collected inside this run."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json"))):
        try:
            with open(path) as f:
                j = json.load(f)
        except (OSError, ValueError):
            continue
        if j.get("width") != width or j.get("height") != height:
            continue
        for name, k in j.get("kernels", {}).items():
            if name.startswith(kernel) and "hbm_bytes_per_launch_corrected" in k:
                best = (int(k["hbm_bytes_per_launch_corrected"]), os.path.basename(path))
    return best


def measured_valu_busy(width, height, kernel="k_deferred_light"):
    """Vector-ALU busy fraction of `kernel` from the newest committed PMC summary (profiles/*pmc_valu*.json) taken at this
    extent, or None: the lighting kernel is bound by ALU issue, so this is reported beside the HBM fraction."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_valu*.json"))):
        try:
            with open(path) as f:
                j = json.load(f)
        except (OSError, ValueError):
            continue
        if j.get("width") != width or j.get("height") != height:
            continue
        for name, k in j.get("kernels", {}).items():
            if name.startswith(kernel) and "valu_busy_frac" in k:
                best = (min(1.0, float(k["valu_busy_frac"])), os.path.basename(path))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--mode", default="deferred", choices=["deferred", "forward", "mosaic"])
    ap.add_argument("--partition", default="poses", choices=["poses", "sortfirst"])
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--tex-size", type=int, default=1024)
    ap.add_argument("--shadow", type=int, default=0,
                    help="deferred mode: also render an N x N light depth map per frame (the 'deferred + shadow map' stress composition)")
    ap.add_argument("--fused", action="store_true",
                    help="deferred mode: G-buffer resolve and lighting as one kernel (VKR_FRAME_FUSED_LIGHTING); the roofline "
                         "object then describes k_resolve_light with 12 B/pixel (8 read from the visibility buffer, 4 written)")
    ap.add_argument("--lanes", type=int, default=1,
                    help="frames in flight (vkr_context_set_frames_in_flight): consecutive frames run on streams of their own and "
                         "overlap on the GPU; the lighting kernel's event times then include the kernels it shares the GPU with")
    ap.add_argument("--readback", default="none", choices=["none", "pinned", "pageable"],
                    help="deliver every frame into host memory (the PCIe-inclusive rate; the headline value keeps frames in HBM)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks share GPUs, collectives on host tensors)")
    ap.add_argument("--profile-all", action="store_true", help="time every stage with HIP events (adds host overhead)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
            local_rank %= max(1, torch.cuda.device_count())
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run" % (args.gpus, world), file=sys.stderr)
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from vulkanrenderer_amd import abi, renderer, scenes

    W, H = args.width, args.height
    t0 = time.time()
    packed = scenes.proxy_scene(detail=args.detail, tex_size=args.tex_size)
    ctx = renderer.Context(local_rank)
    scene = renderer.Scene(ctx, packed)
    stats = scene.stats()
    t_scene = time.time() - t0

    from vulkanrenderer_amd.dist import BandCompositor

    kw = dict(deferred=(args.mode == "deferred"), mosaic=(args.mode == "mosaic"), debug=1)
    if args.shadow and args.mode == "deferred":
        kw.update(shadow_size=args.shadow, flags=abi.FRAME_SHADOW_PASS)
    if args.fused and args.mode == "deferred":
        kw["flags"] = kw.get("flags", 0) | abi.FRAME_FUSED_LIGHTING
    sortfirst = args.partition == "sortfirst" and world > 1
    comp = None
    if sortfirst:
        comp = BandCompositor(W, H, dev, rank, world, host_collective=(args.backend == "gloo"))
        rect = comp.rect
        out_t = comp.local
    else:
        rect = None
        out_t = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
    rh = (rect[3] - rect[1]) if rect else H
    lanes = 1 if sortfirst else max(1, args.lanes)  # sort-first gathers after every frame: nothing to overlap with
    ctx.set_frames_in_flight(lanes)
    # one output image per lane, as a swapchain has one image per frame in flight
    outs = [out_t] + [torch.empty_like(out_t) for _ in range(lanes - 1)]
    if args.readback != "none" and not sortfirst:
        outs = [torch.empty(out_t.shape, dtype=torch.uint8, pin_memory=(args.readback == "pinned")) for _ in range(lanes)]
    fos = []
    for o in outs:
        fo = abi.FrameOut()
        fo.rgba8 = o.data_ptr()
        fos.append(fo)

    def frame(i):
        # poses: frame i of the fly-through goes to rank i % world; sort-first: every rank renders its band of frame i
        pose_index = i if sortfirst else (i * world + rank)
        u = renderer.make_uniforms(W, H, fly_pose(pose_index % 512), 1)
        p = abi.make_params(W, H, u, rect=rect, **kw)
        if rh > 0:
            scene.submit(p, fos[i % lanes])
        if sortfirst:
            ctx.wait()  # the band must be complete before RCCL reads it on torch's stream
            comp.gather()

    def barrier():
        ctx.wait()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        frame(i)
    barrier()
    ctx.profile(True)
    renderer._check(renderer.lib().vkr_profile_enable(ctx.h, 1 if args.profile_all else 2))
    barrier()
    t_start = time.perf_counter()
    for i in range(args.steps):
        frame(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    report = ctx.profile_report()
    renderer._check(renderer.lib().vkr_profile_enable(ctx.h, 0))

    frames_total = args.steps * (1 if sortfirst else world)
    pixels_total = frames_total * W * H
    value = pixels_total / elapsed / 1e6

    roofline = None
    fusedk = report.get("k_resolve_light")
    if fusedk and fusedk[1]:
        avg_ms = fusedk[0] / fusedk[1]
        nbytes = FUSED_BYTES_PER_PIXEL * W * rh
        achieved = nbytes / (avg_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None, "kernel": "k_resolve_light",
                    "avg_launch_us": round(avg_ms * 1e3, 2), "algorithmic_bytes_per_launch": nbytes,
                    "note": "fused G-buffer resolve + lighting: replaces k_resolve_gbuf + k_deferred_light (44 B/pixel between them)"}
    light = report.get("k_deferred_light")
    if light and light[1]:
        avg_ms = light[0] / light[1]
        launch_pixels = W * rh
        achieved = LIGHT_BYTES_PER_PIXEL * launch_pixels / (avg_ms * 1e-3) / 1e9
        traffic = measured_traffic(W, rh) if args.mode == "deferred" else None
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic[0] if traffic else None,
                    "traffic_source": ("profiles/" + traffic[1]) if traffic else None, "kernel": "k_deferred_light",
                    "avg_launch_us": round(avg_ms * 1e3, 2), "algorithmic_bytes_per_launch": LIGHT_BYTES_PER_PIXEL * launch_pixels}
        valu = measured_valu_busy(W, rh) if args.mode == "deferred" else None
        if valu:
            roofline["valu_busy_frac"] = valu[0]
            roofline["valu_busy_source"] = "profiles/" + valu[1]

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        osc = oracle.OracleScene(packed)
        u = oracle.make_uniforms(W, H, fly_pose(0), 1)
        p = abi.make_params(W, H, u, **kw)
        # the box gives one GPU's job 16 cores (os.cpu_count() reports the whole machine)
        try:
            cores = min(16, len(os.sched_getaffinity(0)))
        except AttributeError:
            cores = min(16, os.cpu_count() or 1)
        tc = time.perf_counter()
        ref = osc.render(p, want=("rgba8",), threads=cores)["rgba8"]
        dt = time.perf_counter() - tc
        # the oracle frame doubles as a check of the frame the GPU path produces for the same pose
        got = scene.render(abi.make_params(W, H, renderer.make_uniforms(W, H, fly_pose(0), 1), **kw), want=("rgba8",))["rgba8"]
        diff = np.abs(ref.astype(np.int32) - got.astype(np.int32))
        mse = float(np.mean(diff.astype(np.float64) ** 2))
        cpu = {"value": round(W * H / dt / 1e6, 4), "unit": "Mpix/s", "cores": cores, "kind": "port",
               "sample": "1 frame of the same workload (pose 0, %dx%d %s) by the CPU restatement oracle/vkr_oracle.c, %d threads over horizontal bands, %.1f s"
                         % (W, H, args.mode, cores, dt),
               "gpu_frame_vs_oracle": {"max_abs_diff_8bit": int(diff.max()), "psnr_db": (99.0 if mse == 0 else round(10 * math.log10(255.0 ** 2 / mse), 1))}}
        osc.close()

    if rank == 0:
        line = {
            "metric": "Mpix/s (frames/s) Sun Temple deferred 4K at 1/2/4/8 GPUs; % HBM roofline",
            "value": round(value, 2), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if sortfirst else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "frames_per_s": round(frames_total / elapsed, 2),
            "config": {"workload": "SunTemple-proxy v1 %s shading, 21 brazier lights, %dx%d, fly-through poses, whole frame%s" % (
                args.mode, W, H, (" + %dx%d light depth map" % (args.shadow, args.shadow)) if (args.shadow and args.mode == "deferred") else ""),
                       "triangles": stats["triangles"], "vertices": stats["vertices"], "texture_bytes": stats["texture_bytes"],
                       "partition": args.partition, "frames_in_flight": lanes, "readback": args.readback, "scene_setup_s": round(t_scene, 1)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if args.profile_all:
            line["stages_us"] = {k: round(v[0] / v[1] * 1e3, 2) for k, v in report.items()}
        print(json.dumps(line))
    scene.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
