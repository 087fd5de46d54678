"""The C++ mirror of the reference's PsOptimizer / LedOptimizer interface (psgradientsdf_amd/host) driven exactly
like main_ps.cpp drives the reference, against the ctypes path and the oracle."""
import os
import subprocess

import numpy as np
import pytest

from psgradientsdf_amd import capi, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "psgradientsdf_amd", "host", "voxelps_scene")


def dump(sc, d, max_it, conv, upsample, reg_n, reg_l, damping):
    os.makedirs(d, exist_ok=True)
    meta = [*sc.dim, float(sc.voxel_size), *sc.shift, float(sc.truncation), sc.F, sc.W, sc.H, sc.vis_words, sc.model_id,
            max_it, conv, upsample, reg_n, reg_l, damping]
    open(os.path.join(d, "meta.txt"), "w").write(" ".join(repr(float(x)) if isinstance(x, (float, np.floating)) else str(int(x)) for x in meta) + "\n")
    for name, arr, dt in [("K.f32", sc.K, np.float32), ("dist.f32", sc.dist, np.float32), ("grad.f32", sc.grad, np.float32),
                          ("weight.f32", sc.weight, np.float32), ("rgb.f32", sc.rgb, np.float32), ("vis.u64", sc.vis, np.uint64),
                          ("images.f32", sc.images, np.float32), ("poses.f32", sc.poses, np.float32), ("frame_idx.i32", sc.frame_idx, np.int32)]:
        np.ascontiguousarray(arr, dt).tofile(os.path.join(d, name))


@pytest.mark.parametrize("model", ["SH1", "LED"])
def test_cpp_mirror_matches_ctypes_and_oracle(built, tmp_path, model):
    from oracle import oracle
    sc = synth.make_scene(N=40, F=6, W=160, H=120, model=model)
    max_it, conv = 4, 1e-9
    st = capi.default_settings(sc.model_id, max_it=max_it, conv_threshold=conv)
    ind, outd = str(tmp_path / "scene") + "/", str(tmp_path / "out") + "/"
    os.makedirs(outd)
    dump(sc, ind, max_it, conv, 0, st.reg_weight_n, st.reg_weight_l, st.damping)
    r = subprocess.run([EXE, ind, outd], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    d_cpp = np.fromfile(outd + "dist_out.f32", np.float32)
    eng = capi.load_engine(sc, sc.K, st, 0); eng.load_scene(sc)
    recs, ok = eng.optimize(capi.ALL)
    band = eng.download_band()
    d_eng = eng.download_volume()["dist"]
    assert np.array_equal(d_cpp[band], d_eng[band])            # same library, same calls -> same bits
    orc = oracle.Oracle(sc, sc.K, st); orc.load_scene(sc)
    orc.optimize(capi.ALL)
    d_orc = orc.download_volume()["dist"]
    err = np.abs(d_cpp[band] - d_orc[band]) / float(sc.voxel_size)
    assert np.quantile(err, 0.999) <= 1e-4, (np.quantile(err, 0.999), err.max())
    doc = open(outd + "optimizer_doc.txt").read()
    assert "albation study settings" in doc and "after distance optimization" in doc and "relative diff" in doc
    assert os.path.exists(outd + "after_poses_opt_3.txt") and os.path.exists(outd + "after_iter_3_pointcloud.ply")
    poses = np.loadtxt(outd + "final_poses.txt")
    assert poses.shape == (sc.F, 8)
    q = poses[:, 4:8]
    assert np.allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-4)
    head = open(outd + "after_iter_3_pointcloud.ply").read().split("end_header")[0]
    assert "element vertex" in head and "property uchar blue" in head
