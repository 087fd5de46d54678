"""Host-side pieces of the voxelPS drop-in that need no GPU: the zlib PNG reader against PIL, the generated
marching-cubes table, and the exit codes of the CLI contract (main_ps.cpp:72-75)."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "psgradientsdf_amd", "host", "voxelPS")
GOLD = os.path.join(ROOT, "tests", "golden", "sokrates_small")


def run(*a):
    return subprocess.run([EXE, *a], capture_output=True, text=True, timeout=120)


def test_png_reader_matches_pil(built, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    cases = {"rgb8.png": rng.integers(0, 256, (37, 53, 3), dtype=np.uint8), "gray16.png": rng.integers(0, 65536, (41, 29), dtype=np.uint16),
             "gray8.png": rng.integers(0, 256, (16, 19), dtype=np.uint8), "rgba8.png": rng.integers(0, 256, (9, 11, 4), dtype=np.uint8)}
    for name, arr in cases.items():
        path = str(tmp_path / name)
        Image.fromarray(arr).save(path)                      # PIL picks per-row filters 0-4
        w, h, ch, bd, s = map(int, run("--selftest-png", path).stdout.split())
        assert (w, h) == (arr.shape[1], arr.shape[0]) and ch == (arr.shape[2] if arr.ndim == 3 else 1)
        assert s == int(arr.astype(np.uint64).sum()), name
    for f in ("color000001.png", "depth000001.png"):           # the real-data fixture
        arr = np.asarray(Image.open(os.path.join(GOLD, f)))
        assert int(run("--selftest-png", os.path.join(GOLD, f)).stdout.split()[4]) == int(arr.astype(np.uint64).sum())
    assert run("--selftest-png", str(tmp_path / "missing.png")).returncode == 1


def test_generated_marching_cubes_table(built):
    ntri, faces, vol, rmin, rmax = map(float, run("--selftest-mc").stdout.split())
    assert ntri == 820                                   # the classic 256-case table has 820 triangles in total
    assert faces > 1500
    assert abs(vol - 4 / 3 * np.pi * 7.2 ** 3) < 0.03 * 4 / 3 * np.pi * 7.2 ** 3   # closed, outward-oriented surface of the 7.2-voxel sphere
    assert 7.0 < rmin and rmax < 7.4                       # every vertex lies on the iso-surface (linear interpolation error only)


def test_cli_contract_exit_codes(built, tmp_path):
    assert run("--config_file", str(tmp_path / "nope.json")).returncode == 1
    bad = tmp_path / "bad.json"; bad.write_text(json.dumps({"input": "x/"}))
    r = run("--config_file", str(bad))
    assert r.returncode == 1 and "missing necessary input arguments" in r.stdout
    cfg = tmp_path / "c.json"; cfg.write_text(json.dumps({"input": str(tmp_path) + "/", "output": str(tmp_path) + "/", "datatype": "synth"}))
    r = run("--config_file", str(cfg))
    assert r.returncode == 1 and "No intrinsics file found" in r.stderr
    assert os.path.exists(tmp_path / "saved_config.json")   # ConfigLoader.h:161-165 writes it before anything else can fail
