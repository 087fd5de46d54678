"""bench.py's launcher contract without a GPU: `--gpus N` never yields a line whose n_gpus differs from N."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLEAN = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PSGSDF_BENCH_SHARE_GPU")}


def test_gpus_n_without_enough_devices_exits_nonzero_and_prints_nothing(built):
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return      # (a multi-GPU box really runs it: covered by the driver)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=CLEAN)
    assert r.returncode != 0 and r.stdout.strip() == "" and "needs 2 devices" in r.stderr


def test_world_size_that_disagrees_with_gpus_is_refused(built):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=300, cwd=ROOT,
                       env=dict(CLEAN, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and r.stdout.strip() == "" and "WORLD_SIZE" in r.stderr
