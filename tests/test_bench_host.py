"""Host logic of bench.py that needs no GPU: configuration presets, the fly-through poses, the self-launch of one rank
per GPU, and the look-up of committed counter summaries."""
import json
import os
import sys

import numpy as np

import bench


def test_default_is_the_headline_configuration():
    a = bench.parse_args([])
    assert (a.gpus, a.mode, a.width, a.height, a.pose, a.partition, a.steps, a.warmup) == (1, "deferred", 3840, 2160, "fly", "poses", 64, 8)
    assert a.shadow == 0 and a.lanes == 1 and not a.fused and a.repeats == 5


def test_presets_name_the_five_configurations():
    want = {1: ("forward", 1280, 720, 2048, "identity"), 2: ("deferred", 1920, 1080, 0, "fly"), 3: ("views", 1920, 1080, 0, "fly"),
            4: ("deferred", 3840, 2160, 4096, "fly"), 5: ("mosaic", 1920, 1080, 0, "fly")}
    for c, (mode, w, h, shadow, pose) in want.items():
        a = bench.parse_args(["--config", str(c)])
        assert (a.mode, a.width, a.height, a.shadow, a.pose) == (mode, w, h, shadow, pose), c
        assert a.preset_name.startswith("config %d" % c)
    # the 512-pose fly-through is split over the ranks, and explicit flags win over the preset
    assert bench.parse_args(["--config", "5", "--gpus", "8"]).steps == 64
    assert bench.parse_args(["--config", "5"]).steps == 512
    a = bench.parse_args(["--config", "2", "--width", "640", "--height", "360", "--steps", "7"])
    assert (a.width, a.height, a.steps, a.mode) == (640, 360, 7, "deferred")


def test_fly_poses_are_rigid_and_cover_the_hall():
    for i in (0, 1, 100, 511):
        m = np.array(bench.fly_pose(i), np.float64).reshape(4, 4).T  # glm column-major -> rows
        r = m[:3, :3]
        assert np.allclose(r @ r.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(r) - 1.0) < 1e-12
        assert np.allclose(m[3], [0, 0, 0, 1])
    assert bench.fly_pose(0)[12:15] == [0.0, -1.0, 0.0] and abs(bench.fly_pose(511)[14] + 90.0) < 1e-12


def test_self_launch_starts_one_fresh_rank_per_gpu(monkeypatch):
    started = []

    class FakeProc:
        def __init__(self, cmd, env):
            started.append((cmd, env))

        def wait(self):
            return 0

    monkeypatch.setattr(bench.subprocess, "Popen", lambda cmd, env: FakeProc(cmd, env))
    argv = ["--gpus", "4", "--steps", "3"]
    assert bench.launch_children(bench.parse_args(argv), argv) == 0
    assert len(started) == 4
    ports = set()
    for r, (cmd, env) in enumerate(started):
        assert cmd[0] == sys.executable and os.path.basename(cmd[1]) == "bench.py" and cmd[2:] == argv
        assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["MASTER_ADDR"]) == (str(r), str(r), "4", "127.0.0.1")
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        ports.add(env["MASTER_PORT"])
    assert len(ports) == 1 and 1024 < int(ports.pop()) < 65536


def test_a_failing_rank_fails_the_launch(monkeypatch):
    codes = iter([0, -11, 0])

    class FakeProc:
        def __init__(self, *_a, **_k):
            self.rc = next(codes)

        def wait(self):
            return self.rc

    monkeypatch.setattr(bench.subprocess, "Popen", lambda cmd, env: FakeProc())
    argv = ["--gpus", "3"]
    assert bench.launch_children(bench.parse_args(argv), argv) == 11


def test_newest_counter_summary_of_the_right_extent(tmp_path, monkeypatch):
    """The summary is picked by the time recorded inside it, never by its file name, and only vouches for the kernel
    while the recorded source hashes are those of the tree."""
    prof = tmp_path / "profiles"
    prof.mkdir()
    src = tmp_path / "vulkanrenderer_amd" / "csrc"
    src.mkdir(parents=True)
    for n in bench.PROFILED_SOURCES["k_deferred_light"]:
        (src / n).write_text("// " + n)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    now = {n: bench.source_sha256(n) for n in bench.PROFILED_SOURCES["k_deferred_light"]}
    old = dict(now, **{"k_light.hip": "0" * 64})
    # zz_ sorts last by name but was taken first; the one taken last carries the hashes of the tree
    for name, w, h, v, taken, hashes, commit in (
            ("zz_pmc_traffic_4k.json", 3840, 2160, 1.0, 100, old, "aaaaaaa"), ("r02_pmc_traffic_4k.json", 3840, 2160, 2.0, 300, now, "bbbbbbb"),
            ("r01_pmc_traffic_4k.json", 3840, 2160, 4.0, 200, now, None), ("r03_pmc_traffic_1080.json", 1920, 1080, 3.0, 50, old, "ccccccc"),
            ("r00_pmc_traffic_720.json", 1280, 720, 5.0, None, None, None)):
        j = {"width": w, "height": h, "kernels": {"k_deferred_light<true>": {"x": v}, "k_other": {"x": 9.0}}}
        if taken is not None:
            j.update(taken_unix=taken, source_sha256=hashes, git_commit=commit)
        (prof / name).write_text(json.dumps(j))
    (prof / "r04_pmc_traffic_4k.json").write_text("not json")
    assert bench.newest_profile("*pmc_traffic*.json", 3840, 2160, "k_deferred_light", "x") == (2.0, "r02_pmc_traffic_4k.json", "bbbbbbb", False)
    # taken on other sources: found, but stale
    assert bench.newest_profile("*pmc_traffic*.json", 1920, 1080, "k_deferred_light", "x") == (3.0, "r03_pmc_traffic_1080.json", "ccccccc", True)
    # a summary that records no hashes vouches for nothing
    assert bench.newest_profile("*pmc_traffic*.json", 1280, 720, "k_deferred_light", "x") == (5.0, "r00_pmc_traffic_720.json", None, True)
    assert bench.newest_profile("*pmc_traffic*.json", 640, 360, "k_deferred_light", "x") is None
    # a change of the kernel's source turns the newest summary stale
    (src / "vkr_light.h").write_text("// edited")
    assert bench.newest_profile("*pmc_traffic*.json", 3840, 2160, "k_deferred_light", "x")[3] is True


def test_committed_profiles_feed_the_bench_line():
    """The counter summaries the default line quotes exist for its extent and carry the keys it reads; when they are
    current (taken on the lighting kernel's sources as they are in this tree) the traffic is the algorithmic figure."""
    t = bench.newest_profile("*pmc_traffic*.json", 3840, 2160, "k_deferred_light", "hbm_bytes_per_launch_corrected")
    v = bench.newest_profile("*pmc_valu*.json", 3840, 2160, "k_deferred_light", "valu_issue_frac")
    assert t and abs(t[0] / (24 * 3840 * 2160) - 1.0) < 0.01   # no wasted re-reads: within 1 % of the algorithmic bytes
    assert v and 0.3 < float(v[0]) < 1.0
