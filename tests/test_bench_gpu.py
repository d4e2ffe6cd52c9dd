"""bench.py end to end on the GPU box at a small size: the one-line contract, the self-launch of several ranks (here
sharing the one GPU through the gloo backend) and both sort-first partitions."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--detail", "0.05", "--tex-size", "64", "--width", "640", "--height", "352", "--steps", "4", "--warmup", "2", "--repeats", "2"]


def run_bench(*flags, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + list(flags), cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]   # rank 0 prints one line, the other ranks none
    return json.loads(lines[0])


def test_one_rank_line_follows_the_contract():
    d = run_bench()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["value"] > 0 and d["vs_baseline"] is None
    assert d["ms_per_step_min"] <= d["ms_per_step"] <= d["ms_per_step_max"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["kernel"] == "k_deferred_light" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["algorithmic_bytes_per_launch"] == 24 * 640 * 352
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["gpu_frame_vs_oracle"]["max_abs_diff_8bit"] <= 1
    assert c["gpu_frame_vs_oracle"]["tri_id_mismatches"] == 0 and c["gpu_frame_vs_oracle"]["depth_bits_mismatches"] == 0
    assert d["fast_path"]["value"] > 0 and "workload" in d["config"]


def test_self_launch_of_two_ranks_pose_partition():
    d = run_bench("--gpus", "2", "--backend", "gloo", "--no-cpu-baseline", "--no-fast-path")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["partition"] == "poses" and d["value"] > 0


@pytest.mark.parametrize("stripe", [0, 32])
def test_self_launch_of_two_ranks_sort_first(stripe):
    d = run_bench("--gpus", "2", "--backend", "gloo", "--partition", "sortfirst", "--stripe-height", str(stripe), "--no-cpu-baseline")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["partition"] == "sortfirst" and d["config"]["stripe_height"] == stripe
    # each rank's lighting launch covers its own rows only: half of the 352 rows as a band, six of the eleven stripes on rank 0
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 24 * 640 * (176 if stripe == 0 else 192)
