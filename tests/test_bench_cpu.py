"""bench.py's own N > 1 plumbing, on the CPU: `python bench.py --gpus 2` with no launcher must
start two ranks itself (fresh child processes, before any GPU call in the parent), shard the batch,
broadcast rank 0's quantized buffers and report n_gpus == 2 -- never a silent 1-GPU line.
`--host-only --tiny` is the harness check of that plumbing: no forward, no timing, value = null
(the product has no CPU path to time)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["MIXDQ_DIST_BACKEND"] = "gloo"
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--tiny", "--host-only",
                           *flags], env=env, capture_output=True, text=True, timeout=timeout)


def _line(proc):
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks_and_broadcasts():
    out = _line(_bench("--gpus", "2"))
    assert out["n_gpus"] == 2
    assert out["weight_broadcast_bytes"] > 0
    assert out["scaling"] == "weak" and out["config"]["per_gpu_batch"] == 1
    assert out["config"]["global_batch"] == 2
    assert out["value"] is None and out["host_only"] is True


def test_named_config_3_is_strong_scaling_over_the_ranks():
    out = _line(_bench("--gpus", "2", "--baseline-config", "3"))
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["config"]["per_gpu_batch"] == 32 and out["config"]["global_batch"] == 64
    assert out["config"]["unet_forwards_per_image"] == 4


def test_world_size_mismatch_fails_loudly():
    proc = _bench("--gpus", "2", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert proc.returncode != 0
    assert "--gpus 2 but WORLD_SIZE=1" in proc.stderr + proc.stdout


def test_a_failing_rank_fails_the_parent():
    # rank 1 cannot rendezvous with itself only: give the children a world of 2 but kill rank 1's
    # import by an impossible backend -> the parent must exit non-zero and print no JSON line
    proc = _bench("--gpus", "2", env_extra={"MIXDQ_DIST_BACKEND": "no-such-backend"}, timeout=300)
    assert proc.returncode != 0
    assert not [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
