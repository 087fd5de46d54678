"""The writers' geometry on the device (psgsdf_extract_mesh / _pointcloud / _sdf, csrc/extract.hip) and the threaded text writers of the voxelPS
drop-in: (1) every file voxelPS writes is BYTE FOR BYTE the file round 4's path wrote (dense download, host marching cubes --
pinned face by face in tests/test_host_tools.py --, iostream formatting, serial PNG decode, normals through the host: `voxelPS --host-writers`);
(2) the point cloud and the sdf block against a numpy restatement of the reference's formulas (OptimizerAux.cpp:456-577)."""
import filecmp
import json
import os
import subprocess

import numpy as np
import pytest

from psgradientsdf_amd import capi, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "psgradientsdf_amd", "host", "voxelPS")
GOLD = os.path.join(ROOT, "tests", "golden", "sokrates_small")


@pytest.mark.parametrize("upsample", [False, True])
def test_voxelps_files_equal_the_host_side_pass_byte_for_byte(built, tmp_path, upsample):
    outs = []
    for flags in ([], ["--host-writers"]):
        out = str(tmp_path / ("host" if flags else "device")) + "/"; os.makedirs(out)
        cfg = {"input": GOLD + "/", "output": out, "pose filename": "pose.txt", "datatype": "multiview", "first": 0, "last": 7, "voxel size": 0.004,
               "truncation factor": 5, "zmin": 0.5, "zmax": 3.5, "sharpness threshold": 0.0, "model type": "SH1", "loss function": "cauchy",
               "reg albedo": 0.0, "reg norm": 10.0, "reg laplacian": 0.0, "max iter": 7, "damping": 10.0 if upsample else 1.0, "converge threshold": 1e-9, "lambda": 0.2,
               "upsample": upsample, "--light": True, "--albedo": True, "--distance": True, "--pose": True, "grid dim": 64 if upsample else 128}
        json.dump(cfg, open(out + "config.json", "w"))
        r = subprocess.run([EXE, "--config_file", out + "config.json"] + flags, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(out)
    names = sorted(f for f in os.listdir(outs[1]) if f not in ("config.json", "saved_config.json"))
    assert sorted(f for f in os.listdir(outs[0]) if f not in ("config.json", "saved_config.json")) == names
    assert {"init_mesh.ply", "init_pointcloud.ply", "init_sdf.sdf", "after_iter_3_mesh.ply", "after_iter_3_pointcloud.ply", "after_iter_6_mesh.ply", "optimizer_doc.txt", "tracking_poses.txt"} <= set(names)
    if upsample:
        assert any(n.startswith("upsample_after_") and n.endswith("_mesh.ply") for n in names)
    for n in names:
        assert filecmp.cmp(outs[0] + n, outs[1] + n, shallow=False), n
    assert os.path.getsize(outs[0] + "after_iter_3_mesh.ply") > 1e5


def pointcloud_by_numpy(v, lin, N, vs):
    """save_pointcloud's formulas (OptimizerAux.cpp:456-511) in float32, operation for operation"""
    f32 = np.float32
    k, rest = np.divmod(lin, N * N); j, i = np.divmod(rest, N)
    g = v["grad"][:, lin].astype(f32)
    z = (g[0] * g[0] + g[1] * g[1]) + g[2] * g[2]
    s = np.sqrt(z, dtype=f32); ok = z > 0
    g = np.where(ok, g / np.where(ok, s, f32(1)), g).astype(f32)
    d = v["dist"][lin]
    return np.stack([vs * i.astype(f32) - d * g[0], vs * j.astype(f32) - d * g[1], vs * k.astype(f32) - d * g[2], g[0], g[1], g[2]], axis=1).astype(f32), (f32(255) * v["rgb"][:, lin]).astype(np.int32).T


@pytest.mark.parametrize("world,transport,mode,N", [(2, "sockets", "iterate", 40), (3, "gloo", "optimize", 24), (4, "sockets", "fuse_rebalance", 40), (8, "sockets", "iterate", 48)])
def test_slab_shares_concatenate_to_the_single_context(built, tmp_path, world, transport, mode, N):
    """Multi-rank contexts: every rank extracts ITS share (collective calls); the shares in rank order must be exactly what ONE context extracts from the
    slabs' stitched volume -- mesh (cells across the cuts: the upper plane of a cut cell is the neighbour's, its albedo arrives by the extraction's own
    exchange), both point clouds, the sdf block; after plain iterations, after the 2x refinement inside psgsdf_optimize, after a slab-parallel fusion and
    re-cut.  "sockets": the engine's built-in node-local transport (psgsdf_comm_init_sockets), also what `voxelPS --gpus N --transport sockets` uses."""
    import test_slab_gpu as ts
    res = ts.run_ranks(tmp_path, "SH1", world, transport, mode, N, 7 if mode == "optimize" else 2, {"SLAB_EXTRACT": "1"}, timeout=200)
    v = {k: ts.stitch(res, k) for k in ("dist", "rgb", "grad")}; v["weight"] = ts.stitch(res, "x_weight")
    dim = [int(x) for x in res[0]["dim"]]; n = dim[0] * dim[1] * dim[2]
    assert dim[0] == (2 * N if mode == "optimize" else N)
    sc = synth.make_scene(N=N, F=5 if mode == "optimize" else 6, W=160, H=120, model="SH1")
    g = capi.GridDesc(); g.dim[:] = dim; g.voxel_size = float(sc.voxel_size) * N / dim[0]; g.shift[:] = [float(x) for x in sc.shift]; g.truncation = 5 * g.voxel_size
    one = capi.load_engine(g, sc.K, capi.default_settings(capi.SH1), 0)
    one.upload_volume(v["dist"], v["grad"], v["weight"], v["rgb"], np.zeros((n, 1), np.uint64), 1)
    # ---- mesh
    xyz, rgb = one.extract_mesh()
    got_xyz = np.concatenate([r["x_mesh_xyz"] for r in res]); got_rgb = np.concatenate([r["x_mesh_rgb"] for r in res])
    assert len(xyz) > 3000 and sum(len(r["x_mesh_xyz"]) > 0 for r in res) >= min(world, 3) - 1        # (really split: several ranks hold faces)
    assert np.array_equal(got_xyz, xyz) and np.array_equal(got_rgb, rgb)
    for r in res:      # the host-side sum every rank places its share with
        assert np.array_equal(r["x_counts"], [len(q["x_mesh_xyz"]) for q in res])
    # ---- point clouds: every fused voxel (against the one context), the band (against numpy: the one context has no band)
    p1, c1 = one.extract_pointcloud(1)
    assert np.array_equal(np.concatenate([r["x_pc1"] for r in res]), p1) and np.array_equal(np.concatenate([r["x_pc1_rgb"] for r in res]), c1) and len(p1) > 1000
    band = np.concatenate([r["band"][:int(r["info"][1] - r["info"][0])] for r in res]).astype(np.int64)
    vs = np.float32(g.voxel_size); lim = np.sqrt(3.0) * float(vs)
    want, wcol = pointcloud_by_numpy(v, band[np.abs(v["dist"][band]).astype(np.float64) < lim], dim[0], vs)
    assert np.array_equal(np.concatenate([r["x_pc0"] for r in res]), want) and np.array_equal(np.concatenate([r["x_pc0_rgb"] for r in res]), wcol) and len(want) > 1000
    # ---- sdf block: the same global box on every rank, the planes in rank order
    lo, d, blk = one.extract_sdf()
    for r in res:
        assert list(r["x_sdf_lo"]) == lo and list(r["x_sdf_dim"]) == d
    assert np.array_equal(np.concatenate([r["x_sdf"] for r in res if r["x_sdf"].size], axis=0), blk)
    one.close()


def test_pointcloud_and_sdf_block_against_numpy(built):
    sc = synth.make_scene(N=48, F=6, W=160, H=120, model="SH1")
    eng = capi.load_engine(sc, sc.K, capi.default_settings(capi.SH1), 0); eng.load_scene(sc)
    eng.init_albedo(); eng.normalize_weights(); eng.iterate(capi.ALL, 2)
    v = eng.download_volume(); band = eng.download_band(); vs = np.float32(eng.info().voxel_size); N = 48
    lim = np.sqrt(3.0) * float(vs)
    for which in (0, 1):
        pn, col = eng.extract_pointcloud(which)
        lin = band[np.abs(v["dist"][band]).astype(np.float64) < lim] if which == 0 else np.nonzero((v["weight"] > 0) & (np.abs(v["dist"]).astype(np.float64) < lim))[0]
        assert len(pn) == len(lin) > 1000
        want, wcol = pointcloud_by_numpy(v, lin, N, vs)
        assert np.array_equal(pn, want), np.abs(pn - want).max()
        assert np.array_equal(col, wcol)
    lo, dim, blk = eng.extract_sdf()
    idx = np.nonzero(np.abs(v["dist"]).astype(np.float64) <= lim)[0]
    k, rest = np.divmod(idx, N * N); j, i = np.divmod(rest, N)
    assert lo == [i.min(), j.min(), k.min()] and dim == [i.max() - i.min() + 1, j.max() - j.min() + 1, k.max() - k.min() + 1]
    full = (-v["dist"]).reshape(N, N, N)
    assert np.array_equal(blk, full[lo[2]:lo[2] + dim[2], lo[1]:lo[1] + dim[1], lo[0]:lo[0] + dim[0]])
    xyz, rgb = eng.extract_mesh()
    assert len(xyz) % 3 == 0 and len(xyz) > 3000 and len(rgb) == len(xyz)
    assert xyz.min() > -1e-3 and xyz.max() < float(vs) * N
