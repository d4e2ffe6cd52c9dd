"""GPU tests of the scene hand-off (vkr_scene_create / vkr_scene_load), the GPU
mip-chain builder, error behaviour, and parity + size-independent properties
on the SunTemple-proxy scene."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import IDENTITY, pose_matrix

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    from vulkanrenderer_amd import renderer
    c = renderer.Context(0)
    yield c
    c.close()


def params(W, H, pose=IDENTITY, shadow=256, rect=None, **kw):
    from vulkanrenderer_amd import abi, renderer
    u = renderer.make_uniforms(W, H, pose, kw.get("debug", 1))
    return abi.make_params(W, H, u, shadow_size=shadow, rect=rect, **kw)


def test_mip_chain_matches_vkcmdblit_restatement(ctx, micro, micro_oracle):
    """k_mip_level against the oracle's linear-blit mip chain (utils/vkimage.cpp:163-216), every texture and level."""
    from vulkanrenderer_amd import renderer
    sc = renderer.Scene(ctx, micro)
    for t in range(len(micro.textures)):
        lv = 0
        while True:
            try:
                ref = micro_oracle.texture_level(t, lv)
            except IndexError:
                break
            assert np.array_equal(sc.texture_level(t, lv), ref), "texture %d level %d" % (t, lv)
            lv += 1
        assert lv == int(np.log2(micro.textures[t]["rgba8"].shape[0])) + 1
    sc.close()


def test_repacked_material_chains(ctx, micro):
    """Materials whose four shaded textures share a size get two extra chains holding (normal.rgb, metalness.r)
    and (baseColor.rgb, roughness.r); every level is a byte copy of the source chains' levels."""
    from vulkanrenderer_amd import renderer
    sc = renderer.Scene(ctx, micro)
    nt, seen, extra = len(micro.textures), {}, 0
    for base, rough, metal, _alpha, normal, _em in micro.materials:
        for rgb, a in ((normal, metal), (base, rough)):
            if (rgb, a) not in seen:
                seen[(rgb, a)] = nt + extra
                extra += 1
            pid = seen[(rgb, a)]
            levels = int(np.log2(micro.textures[rgb]["rgba8"].shape[0])) + 1
            for lv in range(levels):
                got, src_rgb, src_a = sc.texture_level(pid, lv), sc.texture_level(rgb, lv), sc.texture_level(a, lv)
                assert np.array_equal(got[..., :3], src_rgb[..., :3]) and np.array_equal(got[..., 3], src_a[..., 0]), (pid, lv)
    assert extra >= 2
    with pytest.raises(renderer.VkrError):
        sc.texture_level(nt + extra, 0)
    sc.close()
    # textures of different sizes: no repacked chains, the four-fetch path shades (test_odd_texture_sizes)
    from vulkanrenderer_amd import scenes
    odd = scenes.micro_scene(odd_textures=True)
    sc = renderer.Scene(ctx, odd)
    with pytest.raises(renderer.VkrError):
        sc.texture_level(len(odd.textures), 0)
    sc.close()


def test_baked_obj_renders_like_the_oracle(ctx, tmp_path):
    """OBJ + MTL -> libvkr_bake.so -> .comp5892mesh + copied textures -> vkr_scene_load -> frames equal to the
    oracle's for the same file: the bake step feeds the renderer without the reference's tools."""
    import shutil
    from oracle import oracle
    from vulkanrenderer_amd import abi, bake, baked_model, renderer
    src = tmp_path / "src"
    (src / "tex").mkdir(parents=True)
    (src / "assets-src" / "main").mkdir(parents=True)
    for f in ("model.obj", "model.mtl"):
        shutil.copy(os.path.join(GOLD, "bake", f), str(src / f))
    # the renderer, like the reference's descriptor-set code (main/main.cpp:770-846), needs a normal map per material
    with open(str(src / "model.mtl"), "a") as f:
        f.write("\tnorm tex/a_Normal.png\n")
    rng = np.random.default_rng(21)

    def png(path, size, alpha=None):
        a = rng.integers(0, 256, (size, size, 4), dtype=np.uint8)
        a[..., 3] = 255 if alpha is None else alpha
        renderer.write_png(str(path), a)
    for name in ("a_BaseColor", "r-a_Specular", "m-a_Specular", "a_Normal", "a_Emissive", "leaf_Normal"):
        png(src / "tex" / (name + ".png"), 16)
    yy, xx = np.mgrid[0:16, 0:16]
    png(src / "tex" / "leaf_BaseColor.png", 16, alpha=np.where((xx // 4 + yy // 4) % 2, 255, 20).astype(np.uint8))
    for name in ("r1", "rgb000", "rgba1111"):   # the bake tool's fallback textures (main-bake/main.cpp:44-46)
        png(src / "assets-src" / "main" / (name + ".png"), 2)
    cwd = os.getcwd()
    os.chdir(str(src))   # texture paths in the MTL and the fallbacks are relative to the working directory
    try:
        bake.bake_obj("model.obj", str(tmp_path / "out" / "model.comp5892mesh"), copy_textures=True)
    finally:
        os.chdir(cwd)
    path = str(tmp_path / "out" / "model.comp5892mesh")
    assert sorted(os.listdir(str(tmp_path / "out" / "model-tex")))[0] == "a_BaseColor.png" and len(os.listdir(str(tmp_path / "out" / "model-tex"))) == 10
    sc = renderer.Scene(ctx, path=path)
    ps = abi.PackedScene(*baked_model.load_scene_arrays(path, image_loader=renderer.decode_png))
    osc = oracle.OracleScene(ps)
    assert sc.stats()["triangles"] == ps.triangle_count == 71
    W, H = 160, 90
    pose = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 3.2, 0.6, 2.4, 1]
    for kw in (dict(deferred=True), dict(), dict(debug=2)):
        p = abi.make_params(W, H, renderer.make_uniforms(W, H, pose, kw.get("debug", 1)), shadow_size=256, **kw)
        ref, got = osc.render(p), sc.render(p)
        assert (ref["tri_id"] != 0xFFFFFFFF).mean() > 0.1
        for k in ("tri_id", "depth_bits", "mip_level"):
            assert np.array_equal(ref[k], got[k]), k
        assert np.abs(ref["rgba8"].astype(np.int32) - got["rgba8"].astype(np.int32)).max() <= 1
    sc.close()


def test_mip_chain_odd_sizes(ctx):
    from oracle import oracle
    from vulkanrenderer_amd import abi, renderer
    rng = np.random.default_rng(4)
    tex = [dict(rgba8=rng.integers(0, 256, (6, 20, 4), dtype=np.uint8), space=1, channels=4),
           dict(rgba8=rng.integers(0, 256, (33, 7, 4), dtype=np.uint8), space=0, channels=4),
           dict(rgba8=rng.integers(0, 256, (1, 1, 4), dtype=np.uint8), space=1, channels=4)]
    ps = abi.PackedScene(tex, [(0, 1, 1, 0xFFFFFFFF, 1, 0)], [])
    sc, osc = renderer.Scene(ctx, ps), oracle.OracleScene(ps)
    for t, n in ((0, 5), (1, 6), (2, 1)):
        for lv in range(n):
            assert np.array_equal(sc.texture_level(t, lv), osc.texture_level(t, lv))
        with pytest.raises(renderer.VkrError):
            sc.texture_level(t, n)
    sc.close()


def test_scene_load_equals_scene_create(ctx, micro):
    """The .comp5892mesh + PNG path gives the same frames as handing the arrays over."""
    from vulkanrenderer_amd import renderer
    a = renderer.Scene(ctx, micro)
    b = renderer.Scene(ctx, path=os.path.join(GOLD, "micro.comp5892mesh"))
    assert a.stats() == b.stats()
    for kw in (dict(deferred=True), dict(debug=1), dict(debug=2), dict(debug=6)):
        p = params(120, 68, pose_matrix(0.5, -0.3, -2.0, 0.4, -0.15), **kw)
        ra, rb = a.render(p), b.render(p)
        for k in ra:
            assert np.array_equal(ra[k], rb[k]), k
    a.close(); b.close()


def test_errors(ctx, micro, tmp_path):
    from vulkanrenderer_amd import abi, renderer
    with pytest.raises(renderer.VkrError, match="unable to open"):
        renderer.Scene(ctx, path=str(tmp_path / "nope.comp5892mesh"))
    bad = tmp_path / "bad.comp5892mesh"
    bad.write_bytes(b"\0\0COMP5892Mmesh\0" + b"default".ljust(16, b"\0"))
    with pytest.raises(renderer.VkrError, match="variant"):
        renderer.Scene(ctx, path=str(bad))
    # counts that the rest of the file cannot hold: an error, not an allocation of that size
    head = b"\0\0COMP5892Mmesh\0" + b"21-tan".ljust(16, b"\0")
    for tail in (b"\xff\xff\xff\xff", b"\0\0\0\0" + b"\xff\xff\xff\x7f", b"\0\0\0\0" * 2 + b"\xff\xff\xff\xff",
                 b"\0\0\0\0" * 2 + b"\1\0\0\0" + b"\0\0\0\0" + b"\xff\xff\xff\xff" * 2):
        bad.write_bytes(head + tail)
        with pytest.raises(renderer.VkrError, match="unexpected end of file"):
            renderer.Scene(ctx, path=str(bad))
    m = dict(micro.meshes[0])
    m["indices"] = m["indices"].copy()
    m["indices"][2] = 10 ** 6
    with pytest.raises(renderer.VkrError, match="index"):
        renderer.Scene(ctx, abi.PackedScene(micro.textures, micro.materials, [m]))
    with pytest.raises(renderer.VkrError, match="texture"):
        renderer.Scene(ctx, abi.PackedScene(micro.textures, [(0, 1, 2, 0xFFFFFFFF, 999, 0)], []))
    sc = renderer.Scene(ctx, micro)
    with pytest.raises(renderer.VkrError, match="rectangle"):
        sc.render(params(64, 36, rect=(0, 0, 65, 36), deferred=True))
    with pytest.raises(renderer.VkrError, match="extent"):
        sc.render(params(0, 36, deferred=True))
    sc.close()


def test_non_finite_vertices_are_dropped(ctx, micro):
    """Triangles with a NaN vertex are dropped by triangle setup; infinite and huge coordinates go through the clipper
    without leaving the frame's tiles.  Nothing faults, and NaN triangles leave the image as it was."""
    from vulkanrenderer_amd import abi, renderer
    base = renderer.Scene(ctx, micro)
    nan_mesh = dict(micro.meshes[0])
    nan_mesh["positions"] = np.full_like(micro.meshes[0]["positions"], np.nan)
    wild = dict(micro.meshes[0])
    pos = micro.meshes[0]["positions"].copy()
    pos[0::5, 0] = np.inf
    pos[1::5, 1] = -np.inf
    pos[2::5, 2] = 3.0e38
    pos[3::5] = -1.0e30
    wild["positions"] = pos
    with_nan = renderer.Scene(ctx, abi.PackedScene(micro.textures, micro.materials, [micro.meshes[0], nan_mesh] + list(micro.meshes[1:])))
    with_wild = renderer.Scene(ctx, abi.PackedScene(micro.textures, micro.materials, list(micro.meshes) + [wild]))
    for kw in (dict(deferred=True), dict(), dict(debug=2), dict(debug=5), dict(debug=6), dict(mosaic=True)):
        p = params(160, 90, pose_matrix(0.3, -0.2, -1.5, 0.2), shadow=128, **kw)
        a, b = base.render(p), with_nan.render(p)
        for k in ("rgba8", "depth_bits", "stencil", "mip_level"):
            assert np.array_equal(a[k], b[k]), (kw, k)
        c = with_wild.render(p)
        assert c["rgba8"].shape == a["rgba8"].shape
    base.close(); with_nan.close(); with_wild.close()


def test_empty_scene_renders_clear_colour(ctx, micro):
    from vulkanrenderer_amd import abi, renderer
    sc = renderer.Scene(ctx, abi.PackedScene(micro.textures, micro.materials, []))
    r = sc.render(params(40, 24, deferred=True))
    assert np.all(r["rgba8"] == np.array([89, 89, 89, 26], np.uint8)) and np.all(r["tri_id"] == 0xFFFFFFFF)
    r = sc.render(params(40, 24, debug=5))
    assert np.all(r["stencil"] == 0)
    sc.close()


def test_frames_in_flight(ctx, micro):
    """vkr_render is asynchronous; several frames may be queued before the fence."""
    from vulkanrenderer_amd import abi, renderer
    sc = renderer.Scene(ctx, micro)
    W, H = 96, 54
    poses = [pose_matrix(0.1 * i, -0.2, -0.5 * i, 0.1 * i) for i in range(6)]
    single = [sc.render(params(W, H, p, deferred=True), want=("rgba8",))["rgba8"] for p in poses]
    bufs = [np.zeros((H, W, 4), np.uint8) for _ in poses]
    for p, b in zip(poses, bufs):
        fo = abi.FrameOut()
        fo.rgba8 = b.ctypes.data
        sc.submit(params(W, H, p, deferred=True), fo)
    ctx.wait()
    for a, b in zip(single, bufs):
        assert np.array_equal(a, b)
    sc.close()


def test_frames_in_flight_on_lanes(micro):
    """vkr_context_set_frames_in_flight: consecutive frames run on lanes of their own (stream, scratch, frame
    buffers) and may overlap; every frame equals the one a single-lane context renders."""
    from vulkanrenderer_amd import abi, renderer
    c = renderer.Context(0)
    sc = renderer.Scene(c, micro)
    jobs = []
    for i in range(7):
        W, H = (96, 54) if i % 2 else (160, 90)
        kw = [dict(deferred=True), dict(), dict(debug=6), dict(deferred=True, flags=abi.FRAME_FUSED_LIGHTING), dict(mosaic=True)][i % 5]
        jobs.append(params(W, H, pose_matrix(0.1 * i, -0.2, -0.5 * i, 0.1 * i), shadow=128, **kw))
    single = [sc.render(p, want=("rgba8", "tri_id")) for p in jobs]
    with pytest.raises(renderer.VkrError):
        c.set_frames_in_flight(0)
    with pytest.raises(renderer.VkrError):
        c.set_frames_in_flight(99)
    c.set_frames_in_flight(3)
    c.profile(True)
    for rounds in range(2):  # the second round reuses the lanes' buffers
        outs = []
        for p in jobs:
            img = np.zeros((p.height, p.width, 4), np.uint8)
            tid = np.full((p.height, p.width), 0xFFFFFFFF, np.uint32)  # the prefill Scene.render uses (views 5, 6 write no ids)
            fo = abi.FrameOut()
            fo.rgba8 = img.ctypes.data
            fo.tri_id = tid.ctypes.data
            sc.submit(p, fo)
            outs.append((img, tid))
        with pytest.raises(renderer.VkrError, match="pending"):
            c.set_frames_in_flight(2)
        c.wait()
        for ref, (img, tid) in zip(single, outs):
            assert np.array_equal(ref["rgba8"], img) and np.array_equal(ref["tri_id"], tid)
    rep = c.profile_report()
    assert rep["k_setup"][1] >= 2 * len(jobs)  # stage timers of every lane are summed
    c.set_frames_in_flight(1)
    again = sc.render(jobs[0], want=("rgba8",))["rgba8"]
    assert np.array_equal(again, single[0]["rgba8"])
    sc.close()
    c.close()


def test_unclosed_handles_exit_cleanly():
    """A host that forgets vkr_scene_destroy / vkr_context_destroy (here: a lane context with a frame in flight)
    still exits with status 0: the wrappers release what is open before the interpreter goes down."""
    import subprocess
    import sys
    code = ("import numpy as np\n"
            "from vulkanrenderer_amd import abi, renderer, scenes\n"
            "c = renderer.Context(0); c.set_frames_in_flight(2)\n"
            "s = renderer.Scene(c, scenes.micro_scene())\n"
            "u = renderer.make_uniforms(64, 36, [1,0,0,0, 0,1,0,0, 0,0,1,0, 0,0,0,1], 1)\n"
            "img = np.zeros((36, 64, 4), np.uint8); fo = abi.FrameOut(); fo.rgba8 = img.ctypes.data\n"
            "s.submit(abi.make_params(64, 36, u, deferred=True), fo)\n"
            "print('submitted')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "submitted" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2000:])


def test_c_host_renders_the_same_frames(ctx, tmp_path):
    """tools/vkr_bench (plain C, gcc) drives the library through the C ABI alone: scene file in, frames in flight,
    PNG out; its last frame equals the one the Python binding renders for the same pose."""
    import subprocess
    from vulkanrenderer_amd import abi, renderer
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tools", "vkr_bench")
    subprocess.check_call(["make", "-C", os.path.join(root, "tools")], stdout=subprocess.DEVNULL)  # gcc, a no-op when built
    scene_path = os.path.join(GOLD, "micro.comp5892mesh")
    sc = renderer.Scene(ctx, path=scene_path)
    W, H = 160, 90
    for mode, kw, lanes in (("deferred", dict(deferred=True), 2), ("mosaic", dict(mosaic=True), 3), ("debug6", dict(debug=6), 1)):
        png = str(tmp_path / (mode + ".png"))
        r = subprocess.run([exe, scene_path, "--width", str(W), "--height", str(H), "--frames", "7", "--first", "3", "--mode", mode,
                            "--lanes", str(lanes), "--png", png], capture_output=True, text=True, timeout=240)
        assert r.returncode == 0 and "frames_per_s" in r.stdout, (r.returncode, r.stdout, r.stderr)
        got = renderer.decode_png(png)[::-1]  # the decoder returns rows bottom first
        ref = sc.render(params(W, H, bench.fly_pose(3 + 6), shadow=2048, **kw), want=("rgba8",))["rgba8"]  # the host leaves shadow_size 0 = 2048
        assert np.array_equal(got, ref), mode
    # the same host as one rank of a sort-first run: its rows rendered into device memory, gathered by the RCCL
    # compositor (a communicator of one rank), read back and written out - bands and interleaved stripes
    ref = sc.render(params(W, H, bench.fly_pose(3 + 6), shadow=2048, deferred=True), want=("rgba8",))["rgba8"]
    for stripe in (0, 32):
        png, uid = str(tmp_path / ("sf%d.png" % stripe)), str(tmp_path / ("uid%d" % stripe))
        r = subprocess.run([exe, scene_path, "--width", str(W), "--height", str(H), "--frames", "7", "--first", "3", "--mode", "deferred", "--png", png,
                            "--ranks", "1", "--rank", "0", "--uid-file", uid, "--stripe-height", str(stripe)], capture_output=True, text=True, timeout=240)
        assert r.returncode == 0 and '"ranks": 1' in r.stdout, (r.returncode, r.stdout, r.stderr)
        assert np.array_equal(renderer.decode_png(png)[::-1], ref), "sort-first host, stripe height %d" % stripe
    sc.close()


@pytest.fixture(scope="module")
def proxy_small():
    from vulkanrenderer_amd import scenes
    return scenes.proxy_scene(detail=0.02, tex_size=64)


@pytest.mark.parametrize("mode,kw", [("deferred", dict(deferred=True)), ("deferred-fused", dict(deferred=True, flags=2)),  # VKR_FRAME_FUSED_LIGHTING
                                     ("forward", dict(debug=1)), ("mip", dict(debug=2)),
                                     ("deriv", dict(debug=4)), ("overdraw", dict(debug=5)), ("overshade", dict(debug=6)),
                                     ("mosaic", dict(mosaic=True))])
def test_proxy_scene_parity(ctx, proxy_small, mode, kw):
    """A thinned SunTemple-proxy (all 48 materials, 3 alpha-masked, ~12k triangles) at 480x270 against the oracle."""
    from oracle import oracle
    from vulkanrenderer_amd import renderer
    sc, osc = renderer.Scene(ctx, proxy_small), oracle.OracleScene(proxy_small)
    for pose in (IDENTITY, pose_matrix(0.0, -2.0, -70.0, yaw=np.pi), pose_matrix(-5.0, -3.0, -6.0, yaw=0.9, pitch=-0.3)):
        p = params(480, 270, pose, shadow=512, **kw)
        ref, got = osc.render(p), sc.render(p)
        for k in ("tri_id", "depth_bits", "stencil", "mip_level"):
            assert np.array_equal(ref[k], got[k]), "%s %s: %d pixels differ" % (mode, k, int((ref[k] != got[k]).sum()))
        if mode in ("forward", "mosaic"):
            assert np.array_equal(ref["shadow_map"], got["shadow_map"])
        d = np.abs(ref["rgba8"].astype(np.int32) - got["rgba8"].astype(np.int32))
        assert d.max() <= 1, "%s colour off by %d at %d pixels" % (mode, int(d.max()), int((d > 1).sum()))
        mse = np.mean(d.astype(np.float64) ** 2)
        assert mse == 0 or 10 * np.log10(255.0 ** 2 / mse) >= 50.0
    sc.close(); osc.close()


def test_properties_at_full_resolution(ctx):
    """Size-independent properties at BASELINE.json's sizes (1920x1080 and 3840x2160) on a denser proxy scene."""
    from vulkanrenderer_amd import renderer, scenes
    from vulkanrenderer_amd.dist import band_rect
    packed = scenes.proxy_scene(detail=0.25, tex_size=128)
    sc = renderer.Scene(ctx, packed)
    ntri = packed.triangle_count
    pose = pose_matrix(0.0, -1.0, -20.0, yaw=0.2)
    for (W, H) in ((1920, 1080), (3840, 2160)):
        a = sc.render(params(W, H, pose, shadow=2048, deferred=True), want=("rgba8", "tri_id", "depth_bits"))
        b = sc.render(params(W, H, pose, shadow=2048, deferred=True), want=("rgba8", "tri_id", "depth_bits"))
        for k in a:
            assert np.array_equal(a[k], b[k]), "not deterministic: " + k                      # idempotence
        cov = a["tri_id"] != 0xFFFFFFFF
        assert cov.mean() > 0.99 and a["tri_id"][cov].max() < ntri                              # inside a closed hall
        z = a["depth_bits"].view(np.float32)
        assert z.min() >= 0.0 and z.max() <= 1.0
        # sort-first: 8 bands rendered one by one stitch to the single-pass frame, byte for byte
        bands = [sc.render(params(W, H, pose, shadow=2048, deferred=True, rect=band_rect(r, 8, W, H)), want=("rgba8", "tri_id"))
                 for r in range(8)]
        assert np.array_equal(np.concatenate([x["rgba8"] for x in bands]), a["rgba8"])
        assert np.array_equal(np.concatenate([x["tri_id"] for x in bands]), a["tri_id"])
    W, H = 1920, 1080
    od = sc.render(params(W, H, pose, debug=5), want=("stencil",))["stencil"]
    os_ = sc.render(params(W, H, pose, debug=6), want=("stencil", "depth_bits"))
    dv = sc.render(params(W, H, pose, debug=3), want=("depth_bits", "tri_id"))
    assert np.all(os_["stencil"] <= od) and np.all((os_["stencil"] > 0) == (od > 0)) and od.max() > 3
    assert np.array_equal(os_["depth_bits"], dv["depth_bits"])                                  # same visible surface
    mo = sc.render(params(1920, 1080, pose, shadow=2048, mosaic=True), want=("rgba8",))["rgba8"]
    blocks = mo.reshape(360, 3, 384, 5, 4)
    assert np.all(blocks == blocks[:, :1, :, :1, :])                                            # 5x3 blocks are flat
    sc.close()


def test_frames_in_flight_at_full_resolution():
    """Three lanes, 3840x2160, fly-through poses of the proxy scene, mixed modes: overlapped frames equal the frames
    rendered one at a time, byte for byte (no state is shared between frames in flight but the read-only scene)."""
    from bench import fly_pose
    from vulkanrenderer_amd import abi, renderer, scenes
    c = renderer.Context(0)
    sc = renderer.Scene(c, scenes.proxy_scene(detail=0.25, tex_size=128))
    W, H = 3840, 2160
    kws = [dict(deferred=True), dict(deferred=True, flags=abi.FRAME_FUSED_LIGHTING), dict(), dict(deferred=True), dict(debug=6), dict(deferred=True)]
    jobs = [params(W, H, fly_pose(40 * i), shadow=1024, **kw) for i, kw in enumerate(kws)]
    single = [sc.render(p, want=("rgba8", "depth_bits")) for p in jobs]
    c.set_frames_in_flight(3)
    outs = []
    for p in jobs:
        img, z = np.zeros((H, W, 4), np.uint8), np.zeros((H, W), np.uint32)
        fo = abi.FrameOut()
        fo.rgba8, fo.depth_bits = img.ctypes.data, z.ctypes.data
        c.wait_next_lane()
        sc.submit(p, fo)
        outs.append((img, z))
    c.wait()
    for i, (ref, (img, z)) in enumerate(zip(single, outs)):
        assert np.array_equal(ref["rgba8"], img) and np.array_equal(ref["depth_bits"], z), "frame %d" % i
    sc.close()
    c.close()


def test_full_scene_1080p_against_oracle(ctx):
    """The benchmark's own workload - every triangle of the SunTemple-proxy (smaller textures), fly-through poses of
    bench.py - at 1920x1080 against the oracle: integer buffers bit-exact, colour within 1/255."""
    from bench import fly_pose
    from oracle import oracle
    from vulkanrenderer_amd import abi, renderer, scenes
    packed = scenes.proxy_scene(detail=1.0, tex_size=64)
    sc, osc = renderer.Scene(ctx, packed), oracle.OracleScene(packed)
    W, H = 1920, 1080
    cases = [(0, dict(deferred=True)), (40, dict(deferred=True)), (128, dict(debug=2)), (0, dict(debug=5)), (0, dict(debug=6))]
    try:
        for pi, kw in cases:
            u = renderer.make_uniforms(W, H, fly_pose(pi), kw.get("debug", 1))
            p = abi.make_params(W, H, u, shadow_size=2048, **kw)
            want = ("rgba8", "tri_id", "depth_bits", "mip_level", "stencil")
            ref, got = osc.render(p, want=want), sc.render(p, want=want)
            for k in ("tri_id", "depth_bits", "stencil", "mip_level"):
                assert np.array_equal(ref[k], got[k]), "pose %d %s %s: %d pixels differ" % (pi, kw, k, int((ref[k] != got[k]).sum()))
            d = np.abs(ref["rgba8"].astype(np.int32) - got["rgba8"].astype(np.int32))
            assert d.max() <= 1, "pose %d %s: colour off by %d at %d pixels" % (pi, kw, int(d.max()), int((d > 1).sum()))
            mse = np.mean(d.astype(np.float64) ** 2)
            assert mse == 0 or 10 * np.log10(255.0 ** 2 / mse) >= 50.0
    finally:
        sc.close(); osc.close()


def test_bin_overflow_is_rerendered(micro):
    """Hundreds of screen-filling triangles need far more bin entries than the first guess: the frame is rendered
    again with larger bins and comes out complete."""
    from vulkanrenderer_amd import abi, renderer
    n = 120
    pos, uv, idx = [], [], []
    for q in range(n):
        z = -2.0 - 0.01 * q
        b = len(pos)
        pos += [(-40.0, -40.0, z), (40.0, -40.0, z), (40.0, 40.0, z), (-40.0, 40.0, z)]
        uv += [(0, 0), (1, 0), (1, 1), (0, 1)]
        idx += [b, b + 1, b + 2, b, b + 2, b + 3, b, b + 2, b + 1, b, b + 3, b + 2]   # both windings: one survives the cull
    pos = np.asarray(pos, np.float32); uv = np.asarray(uv, np.float32)
    nrm = np.tile(np.asarray([[0, 0, 1]], np.float32), (len(pos), 1))
    tan = np.tile(np.asarray([[1, 0, 0, 1]], np.float32), (len(pos), 1))
    from vulkanrenderer_amd.baked_model import pack_tbn
    mesh = dict(materialId=0, positions=pos, normals=nrm, texcoords=uv, tangents=tan, tangentsComp=pack_tbn(tan, nrm),
                indices=np.asarray(idx, np.uint32))
    packed = abi.PackedScene(micro.textures, micro.materials, [mesh])
    ctx = renderer.Context(0)
    sc = renderer.Scene(ctx, packed)
    try:
        W, H = 1920, 1080
        u = renderer.make_uniforms(W, H, IDENTITY, 5)
        od = sc.render(abi.make_params(W, H, u, debug=5), want=("stencil",))["stencil"]
        assert od.min() == n and od.max() == n, (int(od.min()), int(od.max()))
        u = renderer.make_uniforms(W, H, IDENTITY, 1)
        r = sc.render(abi.make_params(W, H, u, deferred=True), want=("tri_id",))["tri_id"]
        assert set(np.unique(r).tolist()) <= {0, 1, 2, 3}   # the nearest quad (first four triangles) everywhere
    finally:
        sc.close(); ctx.close()
    # forward path, fresh context: the light's pass (on the helper stream) and the camera's pass both outgrow their first
    # bins; the re-rendered frame equals the same frame rendered once the buffers have grown
    pos2 = pos.copy(); pos2[:, 2] -= 28.0   # in front of the light as well (it looks down -z from z = -12)
    mesh2 = dict(mesh, positions=pos2)
    ctx = renderer.Context(0)
    sc = renderer.Scene(ctx, abi.PackedScene(micro.textures, micro.materials, [mesh2]))
    try:
        W, H = 1920, 1080
        p = abi.make_params(W, H, renderer.make_uniforms(W, H, IDENTITY, 1), shadow_size=2048)
        first = sc.render(p, want=("rgba8", "tri_id", "shadow_map"))
        again = sc.render(p, want=("rgba8", "tri_id", "shadow_map"))
        for k in first:
            assert np.array_equal(first[k], again[k]), k
        assert set(np.unique(first["tri_id"]).tolist()) <= {0, 1, 2, 3}
        assert (first["shadow_map"] < 1.0).mean() > 0.5      # the quads fill most of the light's view
    finally:
        sc.close(); ctx.close()


@pytest.mark.parametrize("quads", [120, 3000, 11000])
def test_overshading_order_in_deep_tiles(micro, quads):
    """Thousands of overlapping quads at random depths inside a few tiles: the overshading counter (fragments that pass
    LESS_OR_EQUAL in submission order, main/main.cpp:2781-2797) depends on the order the tile's bin is walked in.  The
    kernel sorts each bin by primitive itself - in LDS up to 4096 entries, in place in global memory beyond - and the
    counters equal the oracle's for bins of 240, 6000 and 22000 entries."""
    from oracle import oracle
    from vulkanrenderer_amd import abi, renderer
    from vulkanrenderer_amd.baked_model import pack_tbn
    rng = np.random.default_rng(quads)
    pos, uv, idx = [], [], []
    for q in range(quads):
        z = -2.0 - 3.0 * float(rng.random())
        cx, cy = rng.uniform(-0.15, 0.15, 2)
        h = float(rng.uniform(0.05, 0.4))
        b = len(pos)
        pos += [(cx - h, cy - h, z), (cx + h, cy - h, z), (cx + h, cy + h, z), (cx - h, cy + h, z)]
        uv += [(0, 0), (1, 0), (1, 1), (0, 1)]
        idx += [b, b + 1, b + 2, b, b + 2, b + 3]
    pos = np.asarray(pos, np.float32); uv = np.asarray(uv, np.float32)
    nrm = np.tile(np.asarray([[0, 0, 1]], np.float32), (len(pos), 1))
    tan = np.tile(np.asarray([[1, 0, 0, 1]], np.float32), (len(pos), 1))
    # two meshes, so that the file order (the order of the counters) interleaves materials
    half = (quads // 2) * 6
    idx = np.asarray(idx, np.uint32)
    meshes = [dict(materialId=0, positions=pos, normals=nrm, texcoords=uv, tangents=tan, tangentsComp=pack_tbn(tan, nrm), indices=idx[:half]),
              dict(materialId=1, positions=pos, normals=nrm, texcoords=uv, tangents=tan, tangentsComp=pack_tbn(tan, nrm), indices=idx[half:])]
    packed = abi.PackedScene(micro.textures, micro.materials, meshes)
    ctx = renderer.Context(0)
    sc, osc = renderer.Scene(ctx, packed), oracle.OracleScene(packed)
    try:
        W, H = 96, 54
        for debug in (6, 5):
            p = abi.make_params(W, H, renderer.make_uniforms(W, H, IDENTITY, debug), debug=debug)
            ref, got = osc.render(p, want=("stencil", "depth_bits", "rgba8")), sc.render(p, want=("stencil", "depth_bits", "rgba8"))
            assert ref["stencil"].max() >= 5
            for k in ref:
                assert np.array_equal(ref[k], got[k]), "view %d %s: %d pixels differ" % (debug, k, int((ref[k] != got[k]).sum()))
    finally:
        sc.close(); osc.close(); ctx.close()
