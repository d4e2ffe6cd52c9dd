#!/usr/bin/env python3
"""Headline benchmark: deferred shading with all 21 brazier lights of the
SunTemple-proxy scene at 3840x2160, whole frames (transform, setup, binning,
rasterise, G-buffer resolve, lighting), output left in device memory.

  python bench.py --gpus N --steps K --warmup W

A step is one frame for one camera pose.  With N > 1 there is one process per
GPU: either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in
the environment), or - when those are absent - started by this script itself,
which then only waits for its N children (the parent never touches a GPU).
By default the camera poses of a fly-through are partitioned over the ranks
(scene replicated per GPU, no data-path collective), so per-GPU work is fixed:
weak scaling.  `--partition sortfirst` instead gives every rank a share of the
rows of every frame (bands, or interleaved stripes with --stripe-height) and
gathers them on rank 0 over RCCL inside the renderer library (strong scaling).

`--config 1..5` selects the other BASELINE.json workloads.  The K timed steps
are repeated `--repeats` times; ms_per_step is the median repeat, min and max
are printed beside it.  Rank 0 prints one JSON line.
"""
import argparse
import ctypes as C
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0       # MI355X HBM3E peak
FUSED_BYTES_PER_PIXEL = 12   # k_resolve_light: 8 (visibility buffer) read + 4 (RGBA8) written; texels and per-triangle data excluded
LIGHT_BYTES_PER_PIXEL = 24   # 8 (normal+metal RGBA16F) + 8 (albedo+rough RGBA16F) + 4 (D32) read + 4 (RGBA8) written
IDENTITY = [1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0]

PRESETS = {
    # BASELINE.json's five configurations
    1: dict(mode="forward", width=1280, height=720, pose="identity", shadow=2048,
            name="config 1: forward + single-light 2048^2 shadow map, 1280x720, fixed camera (the reference's start pose)"),
    2: dict(mode="deferred", width=1920, height=1080, name="config 2: deferred shading, 21 brazier lights, 1920x1080"),
    3: dict(mode="views", width=1920, height=1080,
            name="config 3: debug views at 1920x1080 (mip level, linearised depth, depth derivatives, overdraw, overshading), one view per frame in turn"),
    4: dict(mode="deferred", width=3840, height=2160, shadow=4096, partition="sortfirst",
            name="config 4: deferred + 4096^2 light depth pass at 3840x2160, sort-first rows across the GPUs with RCCL gather"),
    5: dict(mode="mosaic", width=1920, height=1080, steps=64,
            name="config 5: 512-pose fly-through at 1920x1080, forward + mosaic post-process, poses partitioned over the GPUs"),
}
VIEWS = (2, 3, 4, 5, 6)


def fly_pose(i, n=512):
    """Pose i of the fly-through: translate(0,-1,-90*i/(n-1)) * rotateY(0.35*sin(2*pi*i/128)), glm column-major."""
    a = 0.35 * math.sin(2.0 * math.pi * i / 128.0)
    c, s = math.cos(a), math.sin(a)
    tz = -90.0 * i / (n - 1)
    # columns: X=(c,0,-s), Y=(0,1,0), Z=(s,0,c), T=(0,-1,tz)
    return [c, 0.0, -s, 0.0, 0.0, 1.0, 0.0, 0.0, s, 0.0, c, 0.0, 0.0, -1.0, tz, 1.0]


# what each profiled kernel is compiled from: a committed counter summary describes the kernel only while these files are
# the ones it was taken on
PROFILED_SOURCES = {"k_deferred_light": ("k_light.hip", "vkr_light.h", "vkr_device.h", "vkr_kernels.h")}


def source_sha256(name):
    import hashlib
    with open(os.path.join(ROOT, "vulkanrenderer_amd", "csrc", name), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def newest_profile(pattern, width, height, kernel, key):
    """(value, file, git commit, stale) of `key` for `kernel` from the committed PMC summary (profiles/<pattern>) taken
    last at this extent - by the time recorded inside the file, not by its name - or None.  PMC counters need their own
    rocprofv3 passes, so they cannot be collected inside this run; `stale` says the kernel's sources have changed since
    the passes ran (hashes recorded in the summary against the files of this tree; a summary without hashes cannot
    vouch for anything), and the caller then reports no number."""
    import glob
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", pattern)):
        try:
            with open(path) as f:
                j = json.load(f)
        except (OSError, ValueError):
            continue
        if j.get("width") != width or j.get("height") != height:
            continue
        taken = j.get("taken_unix", 0)
        for name, k in j.get("kernels", {}).items():
            if name.startswith(kernel) and key in k and (best is None or (taken, path) > best[0]):
                best = ((taken, path), k[key], j)
    if best is None:
        return None
    (_, path), value, j = best
    recorded = j.get("source_sha256") or {}
    try:
        stale = not recorded or any(recorded.get(n) != source_sha256(n) for n in PROFILED_SOURCES.get(kernel, ()))
    except OSError:
        stale = True
    return value, os.path.basename(path), j.get("git_commit"), stale


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed frames per repeat and rank (default 64)")
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--repeats", type=int, default=5, help="the timed region of --steps frames is run this many times; median / min / max are reported")
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3, 4, 5], help="one of BASELINE.json's five configurations (0 = the headline: deferred 4K)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--mode", default=None, choices=["deferred", "forward", "mosaic", "views"])
    ap.add_argument("--pose", default=None, choices=["fly", "identity"], help="fly: a different fly-through pose every frame; identity: the reference's start pose")
    ap.add_argument("--partition", default=None, choices=["poses", "sortfirst"])
    ap.add_argument("--stripe-height", type=int, default=0,
                    help="sort-first: 0 = horizontal bands; N (a multiple of 32) = interleaved stripes of N rows, which balance clustered foliage better")
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--tex-size", type=int, default=1024)
    ap.add_argument("--shadow", type=int, default=None,
                    help="deferred mode: also render an N x N light depth map per frame (the 'deferred + shadow map' stress composition); forward / mosaic: the map's size")
    ap.add_argument("--fused", action="store_true",
                    help="deferred mode: G-buffer resolve and lighting as one kernel (VKR_FRAME_FUSED_LIGHTING); the roofline "
                         "object then describes k_resolve_light with 12 B/pixel (8 read from the visibility buffer, 4 written)")
    ap.add_argument("--lanes", type=int, default=1,
                    help="frames in flight (vkr_context_set_frames_in_flight): consecutive frames run on streams of their own and "
                         "overlap on the GPU; the lighting kernel's event times then include the kernels it shares the GPU with")
    ap.add_argument("--readback", default="none", choices=["none", "pinned", "pageable"],
                    help="deliver every frame into host memory (the PCIe-inclusive rate; the headline value keeps frames in HBM)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fast-path", action="store_true", help="skip the second timed region (fused kernel, two frames in flight)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks share GPUs, rows gathered through host tensors)")
    ap.add_argument("--profile-all", action="store_true", help="time every stage with HIP events (adds host overhead)")
    args = ap.parse_args(argv)
    preset = PRESETS.get(args.config, {})
    for key, default in (("mode", "deferred"), ("width", 3840), ("height", 2160), ("pose", "fly"), ("partition", "poses"), ("shadow", 0), ("steps", 64)):
        if getattr(args, key) is None:
            setattr(args, key, preset.get(key, default))
    if args.config == 5 and "--steps" not in (argv if argv is not None else sys.argv):
        args.steps = max(1, 512 // max(1, args.gpus))  # the 512 poses of the fly-through, once per repeat
    args.preset_name = preset.get("name")
    return args


def launch_children(args, argv):
    """`--gpus N` without a launcher: start N copies of this script, one per GPU, as fresh child processes (this
    parent has not touched a GPU and does not), wait for them, pass rank 0's line through."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def host_physical_cores():
    """(physical cores of the machine, logical CPUs this process may run on)."""
    cores = set()
    try:
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return (len(cores) or None), usable


def cpu_baseline(packed, params_for, W, H, mode_name, scene, np):
    """The CPU restatement timed on this machine's host cores: the oracle file compiled here with -O3 -march=native and
    OpenMP over bands of rows (oracle/Makefile target `native`; timing only - the checker stays the -O2 build), one
    warm-up frame + five timed frames of pose 0, median.  Threads: what OpenMP gives this job (OMP_NUM_THREADS, 16 on a
    one-GPU box), and once more with 8 for comparison across machines."""
    from vulkanrenderer_amd import abi, renderer
    from oracle import oracle
    phys, usable = host_physical_cores()
    native = os.path.join(ROOT, "oracle", "_build", "libvkr_oracle_native.so")
    built = subprocess.call(["make", "-C", os.path.join(ROOT, "oracle"), "-B", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) == 0
    p = params_for(0, oracle.make_uniforms)
    res = {}
    if built and os.path.exists(native):
        L = C.CDLL(native)
        L.ora_scene_create.restype = C.c_void_p
        L.ora_scene_create.argtypes = [C.POINTER(abi.SceneDesc)]
        L.ora_scene_destroy.argtypes = [C.c_void_p]
        L.ora_render_bands.argtypes = [C.c_void_p, C.POINTER(abi.FrameParams), C.POINTER(abi.FrameOut), C.c_uint32]
        L.ora_set_band_threads.argtypes = [C.c_int]
        h = L.ora_scene_create(C.byref(packed.desc))
        img = np.zeros((H, W, 4), np.uint8)
        ref_tri, ref_depth = np.zeros((H, W), np.uint32), np.zeros((H, W), np.uint32)
        fo = abi.FrameOut()
        fo.rgba8 = img.ctypes.data
        fo.tri_id = ref_tri.ctypes.data
        fo.depth_bits = ref_depth.ctypes.data
        if p.flags & abi.FRAME_SHADOW_PASS:  # the composition with a light depth pass: the CPU draws that pass as well
            S = p.shadow_size or 2048
            smap = np.zeros((S, S), np.float32)
            fo.shadow_map = smap.ctypes.data
        omp = L.ora_band_threads()
        for threads in sorted({min(omp, usable), min(8, omp, usable)}, reverse=True):
            L.ora_set_band_threads(threads)
            band = max(32, -(-H // (2 * threads)))
            times = []
            for i in range(6 if threads == min(omp, usable) else 4):
                t0 = time.perf_counter()
                rc = L.ora_render_bands(h, C.byref(p), C.byref(fo), band)
                times.append(time.perf_counter() - t0)
                assert rc == 0
            times = sorted(times[1:])  # the first frame is the warm-up
            res[threads] = times[len(times) // 2]
        L.ora_scene_destroy(h)
        ref = {"rgba8": img, "tri_id": ref_tri, "depth_bits": ref_depth}
        build = "-O3 -march=native -fopenmp, built on this machine"
    else:
        # no compiler here: the -O2 checker build, bands over a Python thread pool
        threads = min(16, usable)
        osc = oracle.OracleScene(packed)
        t0 = time.perf_counter()
        ref = osc.render(p, want=("rgba8", "tri_id", "depth_bits"), threads=threads)
        res[threads] = time.perf_counter() - t0
        osc.close()
        build = "-O2 (no compiler on this machine for the timing build), one frame"
    top = max(res)
    # the oracle frame doubles as a check of the frame the GPU path produces for the same pose
    got = scene.render(params_for(0, renderer.make_uniforms), want=("rgba8", "tri_id", "depth_bits"))
    diff = np.abs(ref["rgba8"].astype(np.int32) - got["rgba8"].astype(np.int32))
    mse = float(np.mean(diff.astype(np.float64) ** 2))
    # the integer buffers must be the oracle's bit for bit
    ints = {"tri_id_mismatches": int((ref["tri_id"] != got["tri_id"]).sum()),
            "depth_bits_mismatches": int((ref["depth_bits"] != got["depth_bits"]).sum())}
    out = {"value": round(W * H / res[top] / 1e6, 4), "unit": "Mpix/s", "cores": top, "kind": "port",
           "sample": "pose 0 of the same workload (%dx%d %s) by the CPU restatement oracle/vkr_oracle.c (%s), OpenMP over bands of rows, "
                     "1 warm-up + 5 timed frames, median %.2f s per frame" % (W, H, mode_name, build, res[top]),
           "host_physical_cores": phys, "host_usable_cpus": usable,
           "gpu_frame_vs_oracle": dict({"max_abs_diff_8bit": int(diff.max()),
                                        "psnr_db": (99.0 if mse == 0 else round(10 * math.log10(255.0 ** 2 / mse), 1))}, **ints)}
    for t, dt in res.items():
        if t != top:
            out["value_at_%d_threads" % t] = round(W * H / dt / 1e6, 4)
    return out


def run(args):
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = max(1, torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            if local_rank >= ndev:
                raise SystemExit("bench.py: rank %d has no GPU of its own (%d visible); use --backend gloo to rehearse with shared GPUs" % (rank, ndev))
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            # gloo announces its connections on stdout; the contract is one JSON line there, so its chatter goes to stderr
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group(backend="gloo")
                dist.barrier()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
            local_rank %= ndev
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; running %d rank(s)" % (args.gpus, world, world), file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from vulkanrenderer_amd import abi, renderer, scenes
    from vulkanrenderer_amd.dist import HostCompositor, RcclCompositor

    W, H = args.width, args.height
    t0 = time.time()
    packed = scenes.proxy_scene(detail=args.detail, tex_size=args.tex_size)
    t_generate = time.time() - t0  # synthesis of the proxy scene in numpy: no part of the library
    ctx = renderer.Context(local_rank)
    t1 = time.time()
    scene = renderer.Scene(ctx, packed)  # vkr_scene_create: upload, mip chains, repacked chains, tables
    t_create = time.time() - t1
    stats = scene.stats()
    t_scene = time.time() - t0

    base_kw = dict(deferred=(args.mode == "deferred"), mosaic=(args.mode == "mosaic"), debug=1)
    if args.shadow:
        base_kw["shadow_size"] = args.shadow
        if args.mode == "deferred":
            base_kw["flags"] = abi.FRAME_SHADOW_PASS
    if args.fused and args.mode == "deferred":
        base_kw["flags"] = base_kw.get("flags", 0) | abi.FRAME_FUSED_LIGHTING
    sortfirst = args.partition == "sortfirst" and world > 1
    stripe_h = args.stripe_height if sortfirst else 0
    if sortfirst and args.mode == "mosaic" and stripe_h:
        raise SystemExit("bench.py: the mosaic pass is not rendered in stripes; use bands")
    comp, rect, stripes = None, None, 0
    # frames in flight; the RCCL compositor queues a frame's exchange on the frame's own lane, so sort-first keeps its
    # lanes (the host-tensor rehearsal waits for every frame and has one)
    lanes = 1 if (sortfirst and args.backend != "nccl") else max(1, args.lanes)
    if sortfirst:
        if args.backend == "nccl":
            comp = RcclCompositor(ctx, W, H, rank, world, dev, stripe_height=stripe_h, lanes=lanes)
        else:
            comp = HostCompositor(W, H, dev, rank, world, stripe_height=stripe_h)
        out_t = comp.local
        my_rows = comp.my_rows
        if stripe_h:
            stripes = abi.stripes(stripe_h, world, rank)
        else:
            first = renderer.sortfirst_rows(H, world, rank, 0)[0]
            rect = (0, first, W, first + my_rows)
    else:
        out_t = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
        my_rows = H
    ctx.set_frames_in_flight(lanes)
    # one output image per lane, as a swapchain has one image per frame in flight
    if sortfirst and args.backend == "nccl":
        outs = list(comp.locals)
    else:
        outs = [out_t] + [torch.empty_like(out_t) for _ in range(lanes - 1)]
    if args.readback != "none" and not sortfirst:
        outs = [torch.empty(out_t.shape, dtype=torch.uint8, pin_memory=(args.readback == "pinned")) for _ in range(lanes)]
    fos = []
    for o in outs:
        fo = abi.FrameOut()
        fo.rgba8 = o.data_ptr()
        fos.append(fo)

    def params_for(frame_index, make_uniforms, whole=False):
        """Parameters of the frame this rank renders at step `frame_index` (whole: without this rank's row partition)."""
        pose_index = frame_index if sortfirst else (frame_index * world + rank)
        pose = IDENTITY if args.pose == "identity" else fly_pose(pose_index % 512)
        kw = dict(base_kw)
        if args.mode == "views":
            kw["debug"] = VIEWS[pose_index % len(VIEWS)]
        u = make_uniforms(W, H, pose, kw["debug"])
        if whole:
            return abi.make_params(W, H, u, **kw)
        return abi.make_params(W, H, u, rect=rect, stripes=stripes, **kw)

    def frame(i):
        # poses: frame i of the fly-through goes to rank i % world; sort-first: every rank renders its rows of frame i
        if my_rows > 0:
            scene.submit(params_for(i, renderer.make_uniforms), fos[i % lanes])
        if sortfirst:
            if args.backend != "nccl":
                ctx.wait()  # the host-tensor rehearsal reads the rows on another stream
                comp.gather()
            else:
                comp.gather(i % lanes)  # RCCL: queued on the frame's own stream behind the frame, no host wait

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
    per_repeat = []
    step = args.warmup
    for _ in range(max(1, args.repeats)):
        barrier()
        t_start = time.perf_counter()
        for i in range(args.steps):
            frame(step + i)
        barrier()
        elapsed = time.perf_counter() - t_start
        step += args.steps
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        per_repeat.append(elapsed)
    report = ctx.profile_report()
    renderer._check(renderer.lib().vkr_profile_enable(ctx.h, 0))
    ranked = sorted(per_repeat)
    elapsed = ranked[len(ranked) // 2]  # median repeat

    frames_total = args.steps * (1 if sortfirst else world)
    pixels_total = frames_total * W * H
    value = pixels_total / elapsed / 1e6

    roofline = None
    fusedk = report.get("k_resolve_light")
    if fusedk and fusedk[1]:
        avg_ms = fusedk[0] / fusedk[1]
        nbytes = FUSED_BYTES_PER_PIXEL * W * my_rows
        achieved = nbytes / (avg_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None, "kernel": "k_resolve_light",
                    "avg_launch_us": round(avg_ms * 1e3, 2), "algorithmic_bytes_per_launch": nbytes,
                    "note": "fused G-buffer resolve + lighting: replaces k_resolve_gbuf + k_deferred_light (44 B/pixel between them)"}
    light = report.get("k_deferred_light")
    if light and light[1]:
        avg_ms = light[0] / light[1]
        launch_pixels = W * my_rows
        achieved = LIGHT_BYTES_PER_PIXEL * launch_pixels / (avg_ms * 1e-3) / 1e9
        traffic = newest_profile("*pmc_traffic*.json", W, my_rows, "k_deferred_light", "hbm_bytes_per_launch_corrected") if args.mode == "deferred" else None
        fresh = traffic is not None and not traffic[3]
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": int(traffic[0]) if fresh else None,
                    "traffic_source": ("profiles/%s (commit %s)" % (traffic[1], traffic[2] or "unrecorded")) if fresh else
                                      ("none: profiles/%s was taken on other kernel sources" % traffic[1]) if traffic else None,
                    "kernel": "k_deferred_light",
                    "avg_launch_us": round(avg_ms * 1e3, 2), "algorithmic_bytes_per_launch": LIGHT_BYTES_PER_PIXEL * launch_pixels,
                    # the fraction is measured against the HBM roofline, which is the graded number; what actually limits the
                    # kernel is vector-ALU issue (21 BRDF evaluations per pixel), see valu_issue_frac
                    "limited_by": "valu-issue"}
        issue = newest_profile("*pmc_valu*.json", W, my_rows, "k_deferred_light", "valu_issue_frac") if args.mode == "deferred" else None
        if issue and not issue[3]:
            roofline["valu_issue_frac"] = round(float(issue[0]), 4)  # VALU instructions x 2 cycles / (SIMDs x kernel cycles), not clamped
            roofline["valu_issue_source"] = "profiles/%s (commit %s)" % (issue[1], issue[2] or "unrecorded")

    # The same workload the fastest way the library renders it - resolve and lighting as one kernel, two frames in
    # flight - timed in this run as well, beside the headline composition (which keeps the graded lighting kernel a
    # launch of its own).  Deferred mode, pose partition only.
    fast = None
    if args.mode == "deferred" and not sortfirst and not args.fused and lanes == 1 and args.readback == "none" and not args.no_fast_path:
        ctx.wait()
        ctx.set_frames_in_flight(2)
        fouts = [out_t, torch.empty_like(out_t)]
        ffos = []
        for o in fouts:
            fo = abi.FrameOut()
            fo.rgba8 = o.data_ptr()
            ffos.append(fo)
        fkw = dict(base_kw)
        fkw["flags"] = fkw.get("flags", 0) | abi.FRAME_FUSED_LIGHTING

        def fast_frame(i):
            pose_index = i * world + rank
            pose = IDENTITY if args.pose == "identity" else fly_pose(pose_index % 512)
            scene.submit(abi.make_params(W, H, renderer.make_uniforms(W, H, pose, 1), **fkw), ffos[i % 2])

        for i in range(args.warmup):
            fast_frame(i)
        ftimes = []
        for _ in range(max(1, args.repeats)):
            barrier()
            t_start = time.perf_counter()
            for i in range(args.steps):
                fast_frame(args.warmup + i)
            barrier()
            dt = time.perf_counter() - t_start
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            ftimes.append(dt)
        ftimes.sort()
        fdt = ftimes[len(ftimes) // 2]
        fast = {"composition": "VKR_FRAME_FUSED_LIGHTING + 2 frames in flight (bench.py --fused --lanes 2)",
                "value": round(args.steps * world * W * H / fdt / 1e6, 2), "unit": "Mpix/s", "ms_per_step": round(fdt / args.steps * 1e3, 4),
                "ms_per_step_min": round(ftimes[0] / args.steps * 1e3, 4), "ms_per_step_max": round(ftimes[-1] / args.steps * 1e3, 4),
                "frames_per_s": round(args.steps * world / fdt, 2)}
        ctx.wait()
        ctx.set_frames_in_flight(1)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(packed, lambda i, mk: params_for(i, mk, whole=True), W, H, args.mode, scene, np)

    if rank == 0:
        workload = args.preset_name or ("SunTemple-proxy v1 %s shading, 21 brazier lights, %dx%d, %s, whole frame%s" % (
            args.mode, W, H, "fly-through poses" if args.pose == "fly" else "fixed camera",
            (" + %dx%d light depth map" % (args.shadow, args.shadow)) if (args.shadow and args.mode == "deferred") else ""))
        line = {
            "metric": "Mpix/s (frames/s) Sun Temple deferred 4K at 1/2/4/8 GPUs; % HBM roofline",
            "value": round(value, 2), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if sortfirst else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "frames_per_s": round(frames_total / elapsed, 2),
            "repeats": len(per_repeat), "ms_per_step_min": round(ranked[0] / args.steps * 1e3, 4),
            "ms_per_step_max": round(ranked[-1] / args.steps * 1e3, 4),
            "config": {"workload": workload if args.preset_name is None else "SunTemple-proxy v1, " + workload,
                       "mode": args.mode, "width": W, "height": H,
                       "triangles": stats["triangles"], "vertices": stats["vertices"], "texture_bytes": stats["texture_bytes"],
                       "partition": args.partition if world > 1 else "none", "stripe_height": stripe_h, "backend": args.backend if world > 1 else None,
                       "frames_in_flight": lanes, "readback": args.readback, "scene_setup_s": round(t_scene, 2),
                       "scene_generate_s": round(t_generate, 2), "scene_create_s": round(t_create, 2)},
            "roofline": roofline, "cpu_baseline": cpu, "fast_path": fast,
        }
        if args.profile_all:
            line["stages_us"] = {k: round(v[0] / v[1] * 1e3, 2) for k, v in report.items()}
        print(json.dumps(line), flush=True)
    if comp is not None and hasattr(comp, "close"):
        ctx.wait()
        comp.close()
    scene.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_children(args, sys.argv[1:]))
    run(args)


if __name__ == "__main__":
    main()
