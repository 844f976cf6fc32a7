"""bench.py's launch contract: `--gpus N` is never silently a one-GPU run."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=300)


def test_gpus_n_without_enough_devices_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        return  # (a multi-GPU box would start the ranks; the refusal is what is tested here)
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2 and "refusing to run fewer ranks" in r.stderr, (r.returncode, r.stderr[-400:])
    assert r.stdout.strip() == ""   # no JSON line that could be mistaken for a result


def test_gpus_n_must_equal_world_size():
    r = _run(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr, (r.returncode, r.stderr[-400:])
