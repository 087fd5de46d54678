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
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["value"] > 0 and d["unit"] == "it/s" and d["scaling"] == "strong"
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["launches_timed"] > 0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0


@pytest.mark.parametrize("world", [2, 8])
def test_n_ranks_share_the_gpu(built, world):
    """`python bench.py --gpus N` WITHOUT a launcher starts its N ranks itself; here all on GPU 0 through the gloo test
    transport, each rank on its own share of the CUs (PSGSDF_BENCH_SHARE_GPU=1).  `value` is the STRONG scaling of the metric's own scene -- one
    volume cut into N slabs --, `extra.weak` the N stacked copies, `extra.configs4_strong` the SH2 / two-visibility-word
    workload (a small stand-in here); each carries its multi-GPU block: cross-rank solves, fallbacks, the hand-off memory the probe chose,
    collectives per step, the pre-timing self-check (N ranks vs 1 context, e_total to 1e-5) and -- the primary -- the full-size check."""
    env = dict(os.environ, PSGSDF_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", PSGSDF_FAULT_DUMP="400", GLOO_SOCKET_IFNAME="lo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PSGSDF_CU_MASK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--reps", "2", "--grid", "64", "--frames", "8", "--configs4", "48:70"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=560, cwd=ROOT, env=env)
    if r.returncode != 0 and "gave up" in r.stderr:
        # N processes time-share ONE GPU here and every in-kernel wait for another rank is bounded: once in ~80 runs of the whole suite (round 6: 1 of 25 suite
        # runs, 0 of 68 runs of this command alone, before and after the round's changes to the multi-rank solve) a rank is starved past a bound and the run
        # ends with PSGSDF_ERR_DEVICE -- the designed outcome of a starved rank, not a wrong result.  One more attempt; a second failure is a failure.
        first = r.stderr[-4000:]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=560, cwd=ROOT, env=env)
        r.stderr = "(first attempt ended in a bounded wait: " + first + ")\n" + r.stderr
    try:      # (the ranks' stderr is the only trace of a failed multi-rank run: keep it where gpurun / the driver collect files)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        open(os.path.join(ROOT, "gpurun_out", f"bench_share_gpu_{world}_ranks.stderr.log"), "w").write(r.stderr[-200000:])
    except OSError:
        pass
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_json(r.stdout)                             # only rank 0 prints
    assert KEYS <= set(d) and d["n_gpus"] == world and d["value"] > 0 and d["scaling"] == "strong"
    assert "cpu_baseline" not in d and d["config"]["collectives_per_step"] >= 0      # (cross-rank persistent solve, frame rows, scalar folds and halos all travel through the mapped regions: no communicator call per iteration unless the ranks sharing the GPU fall back to the per-pass solve)
    assert "strong scaling: ONE 64^3 volume with 8 keyframes" in d["config"]["parallelism"]
    rows = d["config"]["band_rows_per_rank"]
    assert len(rows) == world and min(rows) > 0 and max(rows) <= 1.5 * min(rows)
    blocks = [(d["multi_gpu"], d["degraded"])] + [(d["extra"][k]["multi_gpu"], d["extra"][k]["degraded"]) for k in ("weak", "configs4_strong")]
    for mg, degraded in blocks:
        assert mg["ranks"] == world and mg["rccl_ranks"] == 0 and mg["cross_rank_ready"] == 1 and mg["cross_rank_solves"] > 0 and mg["persist_fallbacks"] <= 1
        assert mg["hand_off_memory"] == "fine-grained" and mg["probe_stale_records"] == 0 and mg["probe_timeouts"] == 0
        checks = mg["self_check"] if isinstance(mg["self_check"], list) else [mg["self_check"]]
        assert all(c["ok"] for c in checks)
        assert degraded == (mg["persist_fallbacks"] > 0)
    small, full = d["multi_gpu"]["self_check"]
    assert small["rel_diff"] <= 1e-5 and full["rel_diff"] <= 1e-5 and "the measured one" in full["scene"]
    assert d["extra"]["weak"]["scaling"] == "weak" and d["extra"]["weak"]["value"] > 0 and f"{8 * world} keyframes" in d["extra"]["weak"]["workload"]
    assert d["extra"]["configs4_strong"]["scaling"] == "strong" and "2 visibility words" in d["extra"]["configs4_strong"]["workload"]
    assert d["spread"]["reps"] == 2 and d["spread"]["min"] <= d["value"] <= d["spread"]["max"]


def test_more_ranks_than_devices_is_refused(built):
    """`--gpus 2` on the one-GPU box without PSGSDF_BENCH_SHARE_GPU: non-zero exit and no line -- never an n_gpus other than --gpus -- and the
    same when a launcher's WORLD_SIZE disagrees with --gpus"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PSGSDF_BENCH_SHARE_GPU")}
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--grid", "48", "--frames", "6"],
                           capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
        assert r.returncode != 0 and r.stdout.strip() == "" and "needs 2 devices" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--grid", "48", "--frames", "6"],
                       capture_output=True, text=True, timeout=120, cwd=ROOT, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and r.stdout.strip() == "" and "WORLD_SIZE" in r.stderr


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
    assert len(rows) == 2 and min(rows) > 0 and max(rows) <= 1.35 * min(rows) and d["config"]["collectives_per_step"] >= 0


def test_first_contact_probe(built):
    """`python bench.py --probe-only --gpus N` (tools/first_contact.py): communicator, one collective, the hand-off forms per memory kind
    between every neighbour pair, two iterations with every in-kernel exchange against one context -- one JSON line, every phase under a wall-clock
    bound.  Here: four ranks sharing GPU 0 (each on a quarter of the CUs) through the gloo test transport."""
    env = dict(os.environ, PSGSDF_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", GLOO_SOCKET_IFNAME="lo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PSGSDF_CU_MASK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--probe-only", "--gpus", "4"], capture_output=True, text=True, timeout=400, cwd=ROOT, env=env)
    d = _one_json(r.stdout)
    assert r.returncode == 0 and d["ok"] and d["timed_out_phase"] is None, d
    assert d["ranks"] == 4 and set(d["phases"]) == {"rccl_init", "allreduce", "hand_off", "exchanges"}
    assert d["hand_off_memory_kinds_that_pass"] == ["fine", "uncached"]
    for x in d["phases"]["hand_off"]["per_rank"]:
        for kind in ("fine", "uncached"):
            k = x["kinds"][kind]
            assert k["passed_on_all_ranks"] and k["cross_rank_ready"] == 1 and k["stale_mappings"] == 0 and k["all_ranks_stale_records"] == 0 and k["all_ranks_expired_waits"] == 0
            assert k["this_rank"]["tried"] == 1 and k["this_rank"]["stale_records_from_lower"] == 0
    for x in d["phases"]["exchanges"]["per_rank"]:
        assert x["ok"] and x["e_total_rel_diff"] <= 1e-5 and x["cross_rank_solves"] == 2 and x["persist_fallbacks"] == 0 and x["halo_exchanges_by_push_kernels"] > 0


def test_failed_in_kernel_exchange_falls_back_to_the_communicator(built):
    """A first multi-GPU lease must measure, not debug: if the pre-timing self-check fails with the in-kernel exchanges on (here: rank 1's second halo push is
    dropped -- fault injection, development library -- and rank 0's bounded wait expires), every rank falls back to the communicator paths (PSGSDF_XR=0),
    the self-check is repeated, the measurement goes ahead and the line says `degraded` with the reason."""
    env = dict(os.environ, PSGSDF_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", GLOO_SOCKET_IFNAME="lo", PSGSDF_USE_DEV_LIB="1", PSGSDF_XWAIT_LOG2="16",
               PSGSDF_BENCH_FAULT="1:PSGSDF_FAULT_HALO=2", PSGSDF_BENCH_WATCHDOG_S="400", PSGSDF_DESTROY_TIMEOUT_S="3")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PSGSDF_CU_MASK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--reps", "1", "--grid", "64", "--frames", "8", "--no-extra"],
                       capture_output=True, text=True, timeout=450, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_json(r.stdout)
    assert d["degraded"] is True and d["n_gpus"] == 2 and d["value"] > 0
    fb = d["multi_gpu"]["fallback"]
    assert "PSGSDF_XR=0" in fb["exchanges"] and ("gave up" in fb["reason"] or "NaN came back" in fb["reason"]), fb
    assert d["multi_gpu"]["cross_rank_ready"] == 0 and d["multi_gpu"]["collectives_per_step"] > 10      # the per-pass collectives are back
    assert all(c["ok"] for c in d["multi_gpu"]["self_check"])
    assert "falling back to the communicator paths" in r.stderr
