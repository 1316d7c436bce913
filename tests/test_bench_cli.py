"""bench.py launch plumbing without a GPU: `python bench.py --gpus 2` must spawn its own ranks (torch.distributed.run on
127.0.0.1), build the process group, run the segmented TrainStep with the gradient reducer's collectives and print ONE JSON line
with n_gpus = 2.  `--dry-run-cpu` routes the same code through the host simulator + gloo on a shrunken model (not a measurement)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus2_self_spawn_dry_run(hostsim_path):
    env = dict(os.environ, SFAMD_LIBRARY=hostsim_path, SF_SIM_THREADS="2")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu", "--steps", "1",
                        "--warmup", "0", "--batch", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 1 and out["scaling"] == "weak"
    assert out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 4
    assert out["data"].startswith("DRY RUN") and out["value"] > 0


import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_multi_gpu_attribution_block(gpu):
    """The N > 1 attribution block of bench.py (`multi_gpu`: overlap log + the per-step time of the same binary with the pathway /
    branch streams off) cannot run on a one-GPU box as such; SF_BENCH_ATTRIB=1 executes it at N = 1: a second
    TrainStep captured with the streams off in the same process, timed, reported -- and never an exception out of bench.py."""
    env = dict(os.environ, SF_BENCH_ATTRIB="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SFAMD_LIBRARY"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "4",
                        "--no-secondary", "--no-cpu-baseline", "--no-kernel-profile"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    one = out["multi_gpu"]["one_stream"]
    assert "error" not in one, one
    assert one["ms_per_step"] > 0 and out["ms_per_step"] > 0
