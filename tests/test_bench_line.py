"""bench.py as the driver types it: `python bench.py --gpus N ...` with no launcher around it must
print one valid JSON line, for N > 1 through the library's own multi-GPU path."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_three_workers_on_one_gpu_both_reduces():
    # the headline workload cut down in size only (same scene, same split, same reduce)
    common = ["--gpus", "3", "--devices", "0,0,0", "--steps", "2", "--warmup", "1", "--size", "640", "360", "--spp", "96",
              "--no-cpu-baseline"]
    one = _run("--gpus", "1", "--steps", "2", "--warmup", "1", "--size", "640", "360", "--spp", "96", "--no-cpu-baseline",
               "--no-extra")
    assert one["n_gpus"] == 1 and one["config"]["workload"].startswith("c4")
    for reduce in ("p2p",):
        out = _run(*common, "--reduce", reduce)
        assert out["n_gpus"] == 3 and out["config"]["driver"] == "library" and out["config"]["reduce"] == reduce
        assert out["config"]["workload"].startswith("c4") and out["config"]["devices"] == [0, 0, 0]
        assert out["metric"].startswith("Msamples/sec") and out["value"] > 0 and out["scaling"] == "strong"
        assert out["roofline"]["frac"] > 0 and out["roofline"]["kernel_ms"] > 0
        assert out["rtx_render_to_host"]["ms_per_call"] > 0
        # the same image up to the order of three additions per value
        assert abs(out["checksum"] - one["checksum"]) <= 1e-5 * abs(one["checksum"])
    # RCCL needs distinct GPUs: one rank on the one GPU there is
    out = _run("--gpus", "1", "--reduce", "rccl", "--steps", "2", "--warmup", "1", "--size", "640", "360", "--spp", "96",
               "--no-cpu-baseline")
    assert out["config"]["rccl_ranks"] == 1 and out["checksum"] == one["checksum"]


def test_more_gpus_than_visible_is_a_clear_error():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "visible" in (p.stderr + p.stdout)
