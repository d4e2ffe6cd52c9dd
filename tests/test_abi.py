"""The C-ABI library loads and exports every function include/vkr_renderer.h declares."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    src = open(os.path.join(ROOT, "include", "vkr_renderer.h")).read()
    return sorted(set(re.findall(r"VKR_API[^;(]*?\b(vkr_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_boundary():
    names = declared()
    for must in ("vkr_context_create", "vkr_scene_create", "vkr_scene_load", "vkr_render", "vkr_wait", "vkr_last_error",
                 "vkr_update_scene_uniforms", "vkr_update_depth_mvp_uniforms"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from vulkanrenderer_amd import renderer
    L = C.CDLL(renderer.LIB_PATH)
    for n in declared():
        assert hasattr(L, n), "libvkr_hip.so does not export %s" % n
    assert sorted(renderer.EXPORTS) == declared()


def test_struct_layouts_match_the_reference_blocks():
    from vulkanrenderer_amd import abi
    # std140 sizes of main/main.cpp:205-227
    assert C.sizeof(abi.SceneUniform) == 208 and abi.SceneUniform.projection.offset == 64
    assert abi.SceneUniform.projCam.offset == 128 and abi.SceneUniform.camPos.offset == 192
    assert C.sizeof(abi.LightUniform) == 32 and C.sizeof(abi.MultipleLights) == 21 * 32 and C.sizeof(abi.DepthMVP) == 64
    assert C.sizeof(abi.MaterialDesc) == 24  # BakedMaterialInfo, six u32


def test_null_arguments_fail_cleanly():
    from vulkanrenderer_amd import renderer
    L = renderer.lib()
    assert L.vkr_render(None, None, None, None) != 0
    assert b"NULL" in L.vkr_last_error()
    assert L.vkr_wait(None) != 0
    assert L.vkr_scene_stats(None, None, None, None) != 0
    L.vkr_context_destroy(None)
    L.vkr_scene_destroy(None)
    assert L.vkr_stage_count() >= 10 and L.vkr_stage_name(0) == b"k_setup"


def test_no_device_is_an_error_not_a_fallback():
    """Without a GPU the context cannot be created: there is no CPU path behind the ABI."""
    import ctypes
    from vulkanrenderer_amd import renderer
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    h = ctypes.c_void_p()
    assert renderer.lib().vkr_context_create(0, ctypes.byref(h)) != 0
    assert not h.value
    with pytest.raises(renderer.VkrError):
        renderer.Context(0)


def test_png_write_then_decode_roundtrip(tmp_path):
    """vkr_write_png_file -> vkr_decode_png_file gives the pixels back (the decoder returns rows bottom first)."""
    import numpy as np
    from vulkanrenderer_amd import renderer
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    p = str(tmp_path / "frame.png")
    renderer.write_png(p, img)
    back = renderer.decode_png(p)
    assert np.array_equal(back[::-1], img)
    renderer.write_png(p, img, flip=True)
    assert np.array_equal(renderer.decode_png(p), img)


def test_c_host_builds_and_fails_cleanly_without_a_gpu():
    """tools/vkr_bench.c: the header is plain C99 and a C host links against the library with gcc alone; without a
    GPU the host reports the library's error and exits 1 (no fallback), without arguments it prints its usage."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "tools")], stdout=subprocess.DEVNULL)
    exe = os.path.join(root, "tools", "vkr_bench")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "usage" in r.stderr
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, os.path.join(root, "tests", "golden", "micro.comp5892mesh")], capture_output=True, text=True, timeout=120)
        assert r.returncode == 1 and "vkr_context_create" in r.stderr, (r.returncode, r.stderr)


def test_sortfirst_rows_partition_the_frame():
    """vkr_sortfirst_rows (host-only): bands and interleaved stripes give every row of the frame to exactly one rank;
    bands agree with the Python-side band_rect."""
    from vulkanrenderer_amd import renderer
    from vulkanrenderer_amd.dist import band_rect, stripe_rows
    for height in (2160, 1080, 720, 36, 5):
        for world in (1, 2, 3, 8):
            owner = np.zeros(height, np.int32)
            for r in range(world):
                first, count = renderer.sortfirst_rows(height, world, r, 0)
                assert (first, first + count) == band_rect(r, world, 64, height)[1::2]
                owner[first:first + count] += 1
            assert np.all(owner == 1)
            for sh in (32, 64, 128):
                owner[:] = 0
                for r in range(world):
                    first, count = renderer.sortfirst_rows(height, world, r, sh)
                    rows = stripe_rows(height, world, r, sh)
                    assert count == len(rows) and (count == 0 or rows[0] == first)
                    owner[rows] += 1
                assert np.all(owner == 1)


def test_workgroup_placement_maps_are_bijections(tmp_path):
    """The screen-walking kernels take their place on the screen from a map of the dispatch index (runs of neighbouring
    workgroups per XCD, vulkanrenderer_amd/csrc/vkr_place.h).  The header is plain C++: compiled for the host, every map
    must cover every grid exactly once - a place taken twice would leave pixels undrawn."""
    import subprocess
    exe = str(tmp_path / "place_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tools", "place_check.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "bijective", out.stdout + out.stderr
