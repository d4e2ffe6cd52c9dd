"""The library's file parsers under AddressSanitizer + UndefinedBehaviorSanitizer on mutated inputs (host code only):
Zstandard frames and PNG files, OBJ / MTL text through the bake step, .comp5892mesh files through vkr_scene_load's
reader.  Every input is either decoded or refused; a sanitizer report fails the test."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FUZZ = os.path.join(ROOT, "tools", "fuzz")
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def drivers():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    p = subprocess.run(["make", "-C", FUZZ, "-j3"], capture_output=True, text=True)
    if p.returncode != 0 and "sanitize" in p.stderr and "cannot find" in p.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert p.returncode == 0, p.stderr[-2000:]
    return os.path.join(FUZZ, "_build")


def run(cmd, cwd=None):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, (p.stdout + p.stderr)[-3000:]
    return p.stdout


def test_zstd_and_png_decoders(drivers, tmp_path):
    files = sorted(os.path.join(GOLD, "png", f) for f in os.listdir(os.path.join(GOLD, "png")) if f.endswith(".png"))
    files += sorted(os.path.join(GOLD, "zstd", f) for f in os.listdir(os.path.join(GOLD, "zstd")) if f.endswith(".zst"))
    files.append(os.path.join(GOLD, "bake", "model.obj-zstd"))
    out = run([os.path.join(drivers, "fuzz_decoders"), str(tmp_path)] + files)
    assert "decoded" in out and "rejected" in out


def test_obj_and_mtl_through_the_bake_step(drivers, tmp_path):
    out = run([os.path.join(drivers, "fuzz_bake"), str(tmp_path), os.path.join(GOLD, "bake", "model.obj"), os.path.join(GOLD, "bake", "model.mtl")])
    assert "baked" in out


def test_baked_model_files(drivers, tmp_path):
    # the mutated file lives beside a copy of the textures the good file names
    shutil.copytree(os.path.join(GOLD, "micro_tex"), str(tmp_path / "micro_tex"))
    out = run([os.path.join(drivers, "fuzz_scene_file"), os.path.join(GOLD, "micro.comp5892mesh"), str(tmp_path / "m.comp5892mesh")])
    assert "reached scene creation" in out
