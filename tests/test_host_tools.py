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
    ntri, faces, vol, rmin, rmax, vol2, area, asum = map(float, run("--selftest-mc").stdout.split())
    assert ntri == 820                                   # the classic 256-case table has 820 triangles in total
    assert faces > 1500
    assert abs(vol - 4 / 3 * np.pi * 7.2 ** 3) < 0.03 * 4 / 3 * np.pi * 7.2 ** 3   # outward-oriented surface of the 7.2-voxel sphere
    assert abs(area - 4 * np.pi * 7.2 ** 2) < 0.03 * 4 * np.pi * 7.2 ** 2
    assert 7.0 < rmin and rmax < 7.4                       # every vertex lies on the iso-surface (linear interpolation error only)
    # watertight and consistently oriented (divergence theorem): the signed volume is the same from any origin, the area vectors cancel
    assert abs(vol - vol2) <= 1e-6 * abs(vol) and asum <= 1e-6 * area


# corner and edge numbering of the reference's marching cubes (third/mesh/MarchingCubes.cpp:511-556 computeLutIndex: bit c is set when the
# corner at these (dx, dy, dz) is inside; :342-505 edge e joins these two corners) -- a convention, restated; the tables are NOT restated
MC_CORNER = [(1, 1, 0), (1, 0, 0), (0, 0, 0), (0, 1, 0), (1, 1, 1), (1, 0, 1), (0, 0, 1), (0, 1, 1)]
MC_EDGE = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]


def test_marching_cubes_table_uses_exactly_the_crossed_edges(built):
    """What the reference's `edgeTable` encodes per case is which cube edges the surface crosses: those whose two corners lie on different
    sides.  Computed here from the inside mask (not copied from the reference), it must be EXACTLY the set of edges the generated triangle
    table uses, case by case; every triangle lives on three distinct crossed edges; inside a cell every triangle side is either shared by two
    triangles with opposite orientation or lies on a cube face (the polygon boundary that the neighbouring cell closes); complementary cases
    use the same edges.  The triangles themselves may be fanned differently from `triTable`."""
    lines = run("--selftest-mc-table").stdout.strip().split("\n")
    assert len(lines) == 256
    table = {}
    for ln in lines:
        v = [int(x) for x in ln.split()]
        assert len(v) == 2 + 3 * v[1]
        table[v[0]] = np.array(v[2:], int).reshape(-1, 3)
    face_of_edge_pair = lambda e0, e1: any(all(MC_CORNER[c][ax] == side for e in (e0, e1) for c in MC_EDGE[e]) for ax in range(3) for side in (0, 1))
    total = 0
    for cs in range(256):
        crossed = {e for e, (a, b) in enumerate(MC_EDGE) if ((cs >> a) & 1) != ((cs >> b) & 1)}
        tris = table[cs]
        assert set(tris.ravel().tolist()) == crossed, cs
        total += len(tris)
        directed = {}
        for t in tris:
            assert len(set(t.tolist())) == 3
            for q in range(3):
                directed[(int(t[q]), int(t[(q + 1) % 3]))] = directed.get((int(t[q]), int(t[(q + 1) % 3])), 0) + 1
        for (p, q), n in directed.items():
            assert n == 1, (cs, p, q)                                   # no side used twice in the same direction
            if (q, p) not in directed:
                assert face_of_edge_pair(p, q), (cs, p, q)              # an unmatched side lies on a cube face
        assert set(table[255 - cs].ravel().tolist()) == crossed
        # triangle count of a case: its surface polygons (closed loops of crossed edges) fanned -> crossed edges - 2 per loop
        if crossed:
            assert 1 <= len(tris) <= 5 or len(tris) == len(crossed) - 2
    assert total == 820


def test_cli_contract_exit_codes(built, tmp_path):
    assert run("--config_file", str(tmp_path / "nope.json")).returncode == 1
    bad = tmp_path / "bad.json"; bad.write_text(json.dumps({"input": "x/"}))
    r = run("--config_file", str(bad))
    assert r.returncode == 1 and "missing necessary input arguments" in r.stdout
    cfg = tmp_path / "c.json"; cfg.write_text(json.dumps({"input": str(tmp_path) + "/", "output": str(tmp_path) + "/", "datatype": "synth"}))
    r = run("--config_file", str(cfg))
    assert r.returncode == 1 and "No intrinsics file found" in r.stderr
    assert os.path.exists(tmp_path / "saved_config.json")   # ConfigLoader.h:161-165 writes it before anything else can fail
