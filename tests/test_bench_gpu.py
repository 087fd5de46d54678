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
    """two ranks started the way torch.distributed.run would start them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment; the
    driver itself covers the launcher at N > 1), both on GPU 0"""
    port = str(_port())
    procs = []
    for r in range(2):
        env = dict(os.environ, PSGSDF_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", PSGSDF_FAULT_DUMP="90", GLOO_SOCKET_IFNAME="lo",
                   RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--grid", "64", "--frames", "8"],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env))
    try:
        outs = [p.communicate(timeout=120) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    assert outs[1][0].strip() == ""                     # only rank 0 prints
    d = _one_json(outs[0][0])
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["value"] > 0
    assert "cpu_baseline" not in d and d["config"]["collectives_per_step"] > 3      # (cross-rank persistent solve: no per-pass collectives; ~6 exchanges per iteration remain)


def test_strong_scaling_mode_two_ranks_share_the_gpu(built):
    """bench.py --strong: ONE volume cut into z-slabs of equal band count, every rank synthesising and uploading only its own planes
    (psgsdf_plan_slab + psgsdf_upload_volume_slab); functional check with two ranks on GPU 0 through the gloo test transport"""
    port = str(_port())
    procs = []
    for r in range(2):
        env = dict(os.environ, PSGSDF_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", PSGSDF_FAULT_DUMP="120", GLOO_SOCKET_IFNAME="lo",
                   RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--strong", "--steps", "3", "--warmup", "1", "--grid", "48", "--frames", "6", "--model", "SH2",
                                       "--width", "160", "--height", "120"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env))
    try:
        outs = [p.communicate(timeout=150) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    d = _one_json(outs[0][0])
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    rows = d["config"]["band_rows_per_rank"]
    assert len(rows) == 2 and min(rows) > 0 and max(rows) <= 1.35 * min(rows) and d["config"]["collectives_per_step"] > 3
