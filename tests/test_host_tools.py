"""Host-side pieces of the voxelPS drop-in that need no GPU: the zlib PNG reader against PIL, the marching-cubes table (the classic one,
against the golden extracted from the reference, and against a first-principles generator), the mesh writer re-derived face by face,
and the exit codes of the CLI contract (main_ps.cpp:72-75)."""
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


def test_marching_cubes_closed_surface(built):
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


def _table(flag):
    lines = run(flag).stdout.strip().split("\n")
    assert len(lines) == 256
    table = {}
    for ln in lines:
        v = [int(x) for x in ln.split()]
        assert len(v) == 2 + 3 * v[1]
        table[v[0]] = np.array(v[2:], int).reshape(-1, 3)
    return table


def test_marching_cubes_table_is_the_references(built):
    """the table the mesh writer walks equals `triTable` of third/mesh/MarchingCubes.cpp:31-290 for all 256 cases -- same triangles, same
    vertex order, same triangle order (golden: tests/golden/mc_tritable.npy, extracted by tests/golden/make_mc_tritable.py)"""
    gold = np.load(os.path.join(ROOT, "tests", "golden", "mc_tritable.npy"))
    table = _table("--selftest-mc-table")
    for cs in range(256):
        want = [int(e) for e in gold[cs] if e >= 0]
        assert table[cs].ravel().tolist() == want, cs
    edge = np.load(os.path.join(ROOT, "tests", "golden", "mc_edgetable.npy"))
    for cs in range(256):       # `edgeTable` names exactly the edges whose corners lie on different sides
        assert int(edge[cs]) == sum(1 << e for e, (a, b) in enumerate(MC_EDGE) if ((cs >> a) & 1) != ((cs >> b) & 1))


@pytest.mark.parametrize("flag", ["--selftest-mc-table", "--selftest-mc-generated"])
def test_marching_cubes_table_uses_exactly_the_crossed_edges(built, flag):
    """What `edgeTable` encodes per case is which cube edges the surface crosses: those whose two corners lie on different sides.  Computed
    here from the inside mask, it must be EXACTLY the set of edges the triangle table uses, case by case; every triangle lives on three distinct
    crossed edges; inside a cell no triangle side is used twice in the same direction, and (generated table: always; classic table: wherever the
    side is not shared) an unmatched side lies on a cube face -- the polygon boundary that the neighbouring cell closes.  Holds for the classic
    table in use and for the first-principles generator of round 1 (kept as a cross-check): two independent derivations of the same surface."""
    table = _table(flag)
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
    assert total == 820


def test_mesh_writer_face_by_face(built, tmp_path):
    """`*_mesh.ply` as the product writes it (host/marching_cubes.hpp) against a numpy re-derivation of third/mesh/MarchingCubes.cpp:314-505,
    559-637 driven by the GOLDEN table: same number of vertices and faces, every vertex position, every colour byte and every face index, in
    the reference's order (non-indexed vertices, three per face; degenerate triangles dropped; cubes with a weight-0 corner skipped; colours
    looked up at idx, idx + 1, idx + 2 -- quirk B10 -- and interpolated in getColor's own end-point order)."""
    n = 20
    path = str(tmp_path / "m.ply")
    assert run("--selftest-mc-ply", str(n), path).returncode == 0
    head, body = open(path).read().split("end_header\n")
    nv = int([l for l in head.split("\n") if l.startswith("element vertex")][0].split()[2]); nf = int([l for l in head.split("\n") if l.startswith("element face")][0].split()[2])
    rows = body.strip().split("\n")
    V = np.array([[float(x) for x in r.split()] for r in rows[:nv]]); Fc = np.array([[int(x) for x in r.split()] for r in rows[nv:]])
    assert len(Fc) == nf and nv == 3 * nf and nf > 300
    # the same volume (selftest_mc_ply), float32 arithmetic as in C
    f32 = np.float32
    k, j, i = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    dx, dy, dz = i.astype(f32) - f32(0.47) * f32(n), j.astype(f32) - f32(0.52) * f32(n), k.astype(f32) - f32(0.45) * f32(n)
    t = (f32(0.31) * f32(n) + f32(0.6) * np.sin(f32(0.9) * i.astype(f32)).astype(f32) * np.cos(f32(0.7) * j.astype(f32) + f32(0.3) * k.astype(f32)).astype(f32)
         - np.sqrt(dx * dx + dy * dy + dz * dz).astype(f32)).astype(f32)
    w = np.where(i + j + k < n // 2, f32(0), f32(1))
    r = ((37 * i + 11 * j) & 255).astype(np.uint8); g = ((5 * j + 91 * k) & 255).astype(np.uint8); b = ((17 * k + 3 * i) & 255).astype(np.uint8)
    tf, wf = t.reshape(-1), w.reshape(-1)
    rf, gf, bf = (np.concatenate([c.reshape(-1), np.zeros(2, np.uint8)]) for c in (r, g, b))
    gold = np.load(os.path.join(ROOT, "tests", "golden", "mc_tritable.npy"))
    vox = f32(0.5) * f32(n) / f32(n); org = np.array([0.1, -0.2, 0.3], f32)
    color_rev = {2, 3, 6, 7}

    def interp(t0, t1, v0, v1):
        if abs(f32(0) - t0) < 1e-7: return v0
        if abs(f32(0) - t1) < 1e-7: return v1
        if abs(t0 - t1) < 1e-7: return v0
        mu = float((f32(0) - t0) / (t1 - t0)); mu = min(max(mu, 0.0), 1.0)
        return (v0.astype(np.float64) + mu * (v1 - v0).astype(np.float64)).astype(f32)
    verts, cols = [], []
    lin = lambda x, y, z: (z * n + y) * n + x
    for z in range(n - 2):
        for y in range(n - 2):
            for x in range(n - 2):
                off = [lin(x + c[0], y + c[1], z + c[2]) for c in MC_CORNER]
                if any(wf[o] == 0 for o in off):
                    continue
                cs = sum(1 << c for c in range(8) if tf[off[c]] > 0)
                if cs in (0, 255):
                    continue
                ep, ec = {}, {}
                for e in {int(v) for v in gold[cs] if v >= 0}:
                    a, bb = MC_EDGE[e]
                    pa = (np.array([x + MC_CORNER[a][0], y + MC_CORNER[a][1], z + MC_CORNER[a][2]], f32) * vox - org).astype(f32)
                    pb = (np.array([x + MC_CORNER[bb][0], y + MC_CORNER[bb][1], z + MC_CORNER[bb][2]], f32) * vox - org).astype(f32)
                    ep[e] = interp(tf[off[a]], tf[off[bb]], pa, pb)
                    o1, o2 = (off[bb], off[a]) if e in color_rev else (off[a], off[bb])
                    c1 = np.array([rf[o1], gf[o1 + 1], bf[o1 + 2]], f32) / f32(255); c2 = np.array([rf[o2], gf[o2 + 1], bf[o2 + 2]], f32) / f32(255)
                    ec[e] = (interp(tf[o1], tf[o2], c1, c2) * f32(255)).astype(f32).astype(np.uint8)
                tri = [int(v) for v in gold[cs] if v >= 0]
                for q in range(0, len(tri), 3):
                    p = [ep[tri[q + s]] for s in range(3)]
                    if np.array_equal(p[0], p[1]) or np.array_equal(p[0], p[2]) or np.array_equal(p[1], p[2]):
                        continue
                    for s in range(3):
                        verts.append(p[s]); cols.append(ec[tri[q + s]])
    verts, cols = np.array(verts, np.float64), np.array(cols, int)
    assert len(verts) == nv
    assert np.array_equal(Fc[:, 0], np.full(nf, 3)) and np.array_equal(Fc[:, 1:].ravel(), np.arange(nv))      # three fresh vertices per face, in order
    assert np.abs(V[:, :3] - verts).max() <= 6e-6 * np.abs(verts).max() + 1e-7                                 # (the file holds 6 significant digits)
    assert np.array_equal(V[:, 3:].astype(int), cols)


def test_cli_contract_exit_codes(built, tmp_path):
    assert run("--config_file", str(tmp_path / "nope.json")).returncode == 1
    bad = tmp_path / "bad.json"; bad.write_text(json.dumps({"input": "x/"}))
    r = run("--config_file", str(bad))
    assert r.returncode == 1 and "missing necessary input arguments" in r.stdout
    cfg = tmp_path / "c.json"; cfg.write_text(json.dumps({"input": str(tmp_path) + "/", "output": str(tmp_path) + "/", "datatype": "synth"}))
    r = run("--config_file", str(cfg))
    assert r.returncode == 1 and "No intrinsics file found" in r.stderr
    assert os.path.exists(tmp_path / "saved_config.json")   # ConfigLoader.h:161-165 writes it before anything else can fail


def lapm_numpy(rgb_u8):
    """modifiedLaplacian (SharpDetector.h:22-37) restated with scipy: the image is the loader's float colour image (byte / 255, ImageLoader.h:181);
    Lx = sepFilter2D(src, kernelX = M = [-1 2 -1], kernelY = G), Ly = sepFilter2D(src, G, M) with G = getGaussianKernel(3, -1) = [1/4 1/2 1/4]
    (OpenCV's fixed table for ksize 3) and OpenCV's default border BORDER_REFLECT_101 (= scipy 'mirror': the edge pixel is not repeated);
    cv::mean(|Lx| + |Ly|).val[0] is the mean of CHANNEL 0, which is blue in the reference's BGR images."""
    from scipy import ndimage
    b = rgb_u8[..., 2].astype(np.float32) * np.float32(1.0 / 255.0)
    M = np.array([-1, 2, -1], np.float32); G = np.array([0.25, 0.5, 0.25], np.float32)
    lx = ndimage.correlate1d(ndimage.correlate1d(b, M, axis=1, mode="mirror"), G, axis=0, mode="mirror")
    ly = ndimage.correlate1d(ndimage.correlate1d(b, G, axis=1, mode="mirror"), M, axis=0, mode="mirror")
    return float((np.abs(lx).astype(np.float64) + np.abs(ly).astype(np.float64)).mean())


def sample_keyframes_numpy(n, max_num):
    """sampleKeyFrame (main_ps.cpp:392-421) behind its call-site guard `keyframes.size() > 40` (:312): max_num - 1 picks at a FLOAT running index
    (idx += step in float32, truncated), then the last frame"""
    idx = list(range(n))
    if not n > max_num or n < max_num:
        return idx
    m = max_num - 1
    step = np.float32(n) / np.float32(m)
    out, pos = [], np.float32(0)
    for _ in range(m):
        out.append(idx[int(pos)]); pos = np.float32(pos + step)
    return out + [idx[-1]]


def test_focus_measure_against_scipy(built, tmp_path):
    """the keyframe selector's focus measure (no test at all) on random, smooth and real images, incl. sizes whose borders matter"""
    from PIL import Image
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:48, 0:64]
    cases = {"noise.png": rng.integers(0, 256, (48, 64, 3), dtype=np.uint8),
             "smooth.png": np.stack([(127 + 100 * np.sin(0.2 * xx + 0.1 * yy)), (127 + 100 * np.cos(0.15 * yy)), (127 + 120 * np.sin(0.31 * xx) * np.cos(0.27 * yy))], -1).astype(np.uint8),
             "tiny.png": rng.integers(0, 256, (3, 4, 3), dtype=np.uint8),
             "flat.png": np.full((20, 30, 3), 77, np.uint8)}
    for name, arr in cases.items():
        path = str(tmp_path / name); Image.fromarray(arr).save(path)
        got = float(run("--selftest-lapm", path).stdout)
        want = lapm_numpy(arr)
        assert abs(got - want) <= 2e-6 * max(1.0, want), (name, got, want)      # (float32 products summed in a different order)
    assert float(run("--selftest-lapm", str(tmp_path / "flat.png")).stdout) == 0.0
    arr = np.asarray(Image.open(os.path.join(GOLD, "color000003.png")).convert("RGB"))
    assert abs(float(run("--selftest-lapm", os.path.join(GOLD, "color000003.png")).stdout) - lapm_numpy(arr)) <= 2e-6
    # the red and green channels do not enter: only channel 0 of the reference's BGR image
    arr2 = cases["noise.png"].copy(); arr2[..., :2] = 0
    Image.fromarray(arr2).save(str(tmp_path / "blue_only.png"))
    assert float(run("--selftest-lapm", str(tmp_path / "blue_only.png")).stdout) == float(run("--selftest-lapm", str(tmp_path / "noise.png")).stdout)


@pytest.mark.parametrize("n,max_num", [(41, 40), (46, 40), (97, 40), (40, 40), (12, 40), (151, 40), (9, 4), (1000, 40)])
def test_keyframe_sampling_against_numpy(built, n, max_num):
    """sampleKeyFrame on n selected frames: the four parallel lists (frame index, stamp, image, pose) stay aligned, the float stepping and the
    forced last frame are the reference's"""
    rows = [x.split(":") for x in run("--selftest-sample", str(n), str(max_num)).stdout.split()]
    want = sample_keyframes_numpy(n, max_num)
    assert [int(r[3]) for r in rows] == want
    assert all(int(r[0]) == 3 * int(r[3]) + 1 and int(r[1]) == int(r[3]) and int(r[2]) == -int(r[3]) for r in rows)
    assert len(rows) == (max_num if n > max_num else n) and int(rows[-1][3]) == n - 1


def test_threaded_writers_print_floats_like_the_reference(built):
    """The round-5 writers format floats with their own "%g" (host/ps_optimizer.hpp format_g6: exact power-of-ten scaling in double, snprintf for the values
    within 1e-6 of a rounding tie and outside the exact powers' reach) into per-thread buffers; the reference (and round 4's writers) with `ostream << float`,
    one std::endl per line.  `voxelPS --selftest-fmt N`: N pseudo-random float bit patterns (every exponent, denormals, infinities, NaNs), N uniform values in
    (-2, 2), N log-uniform over 1e-7 .. 1e7, 3 N floats at and next to seven-digit decimals ending in 5 (the nearest a float gets to a tie of the sixth
    digit), a ladder of values at the rounding edges and the 256 colour levels come out character for character the same both ways."""
    r = subprocess.run([EXE, "--selftest-fmt", "1000000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-500:]
    done, bad = (int(x) for x in r.stdout.strip().split("\n")[-1].split())
    assert bad == 0 and done > 6000000


def test_voxelps_gpus_n_without_a_device_fails_loudly_and_ends_every_rank(built, tmp_path):
    """`voxelPS --gpus N` on a box without a GPU: the ranks cannot create their device volumes -- the launcher reports the first one that ends, stops
    the others and returns its code (no CPU fallback, no rank left behind).  The working case runs on the GPU: tests/test_voxelps_ranks_gpu.py."""
    import json
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present: tests/test_voxelps_ranks_gpu.py covers the launcher")
    gold = os.path.join(ROOT, "tests", "golden", "sokrates_small")
    out = str(tmp_path) + "/"
    json.dump({"input": gold + "/", "output": out, "pose filename": "pose.txt", "datatype": "multiview", "first": 0, "last": 3, "voxel size": 0.004, "grid dim": 32, "model type": "SH1"}, open(out + "config.json", "w"))
    for transport in ("sockets", "rccl"):
        r = subprocess.run([os.path.join(ROOT, "psgradientsdf_amd", "host", "voxelPS"), "--config_file", out + "config.json", "--gpus", "3", "--transport", transport], capture_output=True, text=True, timeout=60)
        assert r.returncode != 0
        assert "stopping the other ranks" in r.stderr or "no RCCL on this node" in r.stderr, r.stderr[-800:]
    r = subprocess.run([os.path.join(ROOT, "psgradientsdf_amd", "host", "voxelPS"), "--config_file", out + "config.json", "--gpus", "2", "--host-writers"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "single-process cross-check" in r.stderr
