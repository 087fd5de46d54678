"""bench.py contract: ONE JSON line on stdout with the keys the driver reads, for the single-GPU loop and for the multi-rank host
program (two ranks sharing GPU 0 over gloo -- a functional check of that code path, not a measurement)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"}


def _port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _one_json(out):
    lines = [l for l in out.strip().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_single_gpu_line(built):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--grid", "64", "--frames", "8"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_json(r.stdout)
    assert KEYS <= set(d) and "cpu_baseline" in d
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["value"] > 0 and d["unit"] == "it/s" and d["scaling"] == "weak"
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["launches_timed"] > 0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0


def test_two_ranks_share_the_gpu(built):
    env = dict(os.environ, PSGSDF_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", PSGSDF_FAULT_DUMP="90")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_port()),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--grid", "64", "--frames", "8"],
                       capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_json(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["value"] > 0
    assert d["config"]["collectives_per_step"] > 10 and "cpu_baseline" not in d
