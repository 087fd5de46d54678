"""The branches of the voxelPS drop-in that the reference's OTHER two shipped configurations take, end to end on the GPU:

  config_basket_LED.json   LED model, `synth` layout, sharpness threshold 0.03 (the focus measure really selects), reg norm 0.1 / reg laplacian 5,
                           damping 3, `upsample: true` -> upsample_after_5_*, final_refined_*, refined_sdf.sdf         -> test_led_basket_settings
  config_tumrgbd.json      `datatype: "tum"` + associated.txt, depth unit 1/5000, NO pose file -> frame-to-model tracking,
                           the stream starts one frame late (quirk B12)                                                 -> test_tum_layout_with_tracking
  more than 40 keyframes   sampleKeyFrame (main_ps.cpp:312-314,392-421)                                                 -> test_more_than_40_keyframes

What voxelPS prints and writes is compared with the CPU oracle driven through ctypes on the same inputs (the PNGs as the loaders decode them) and
with numpy restatements of the host-side selection rules (tests/test_host_tools.py holds their CPU-only KATs)."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from psgradientsdf_amd import capi, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "psgradientsdf_amd", "host", "voxelPS")
THREADS = min(64, os.cpu_count() or 1)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def energies(doc):
    return [float(x) for x in re.findall(r"total energy: ([-+0-9.eE]+)", doc)]


def printed_list(stdout, title):
    m = re.search(re.escape(title) + r"\s*\n([0-9 ]*)\n", stdout)
    assert m, stdout[-1500:]
    return [int(x) for x in m.group(1).split()]


def relative_poses(sc):
    """camera -> world poses re-expressed in the first camera's frame: pose[0] = Identity, as in the reference's data sets (main keeps
    key_poses[0] = Identity whatever the file says, quirk B1, and a stream without GT poses starts at Identity)"""
    P0i = np.linalg.inv(sc.poses_gt[0].reshape(4, 4).astype(np.float64))
    return [(P0i @ sc.poses_gt[f].reshape(4, 4).astype(np.float64)) for f in range(sc.F)]


def quantised(sc, images, depth_unit):
    """what the PNG writer stores and what ImageLoader.h:130-188 turns it back into"""
    img8 = np.round(np.clip(images, 0, 1) * 255).astype(np.uint8)
    d16 = np.round(sc.depth / depth_unit).astype(np.uint16)
    return img8, d16, img8.astype(np.float32) * np.float32(1.0 / 255.0), d16.astype(np.float32) * np.float32(depth_unit)


def write_pose_file(path, stamps, poses):
    from scipy.spatial.transform import Rotation
    with open(path, "w") as fh:
        for s, P in zip(stamps, poses):
            q = Rotation.from_matrix(P[:3, :3]).as_quat()
            fh.write(f"{s} {P[0, 3]:.9f} {P[1, 3]:.9f} {P[2, 3]:.9f} {q[0]:.9f} {q[1]:.9f} {q[2]:.9f} {q[3]:.9f}\n")


def read_pose_file(path):
    """ImageLoader.h:228-258 in float32 (Eigen::Quaternionf::toRotationMatrix)"""
    out = []
    f32 = np.float32
    for line in open(path).read().strip().split("\n"):
        v = [f32(x) for x in line.split()[1:]]
        t, (qx, qy, qz, qw) = v[:3], v[3:7]
        tx, ty, tz = f32(2) * qx, f32(2) * qy, f32(2) * qz
        twx, twy, twz, txx, txy, txz, tyy, tyz, tzz = tx * qw, ty * qw, tz * qw, tx * qx, ty * qx, tz * qx, ty * qy, tz * qy, tz * qz
        out.append(np.array([[1 - (tyy + tzz), txy - twz, txz + twy, t[0]], [txy + twz, 1 - (txx + tzz), tyz - twx, t[1]], [txz - twy, tyz + twx, 1 - (txx + tyy), t[2]], [0, 0, 0, 1]], f32))
    return out


def select_keyframes(color_u8, threshold, first_is_key=True):
    """main_ps.cpp:222-258: frame 0 is a keyframe; a later (integrated) frame becomes one if its focus measure is not below the threshold or more
    than five frames have passed since the last one"""
    from test_host_tools import lapm_numpy
    keys, dist = [0], 0
    for i in range(1, len(color_u8)):
        if not (np.float32(lapm_numpy(color_u8[i])) < np.float32(threshold)) or dist > 5:
            keys.append(i); dist = 0
        else:
            dist += 1
    return keys


def test_led_basket_settings(built, tmp_path):
    """config_basket_LED.json's settings on a synthetic LED sequence in the `synth` layout: 18 frames of which three are sharp (extra sensor noise
    lifts their focus measure over the 0.03 threshold), the rest become keyframes only through the five-frame rule.  The convergence threshold is
    lowered (1e-3 instead of 5e-3) so that the run passes iteration 5, refines the grid 2x and ends on the reference's divergence exit:
    every file of that path must appear, the keyframes must be the ones the focus measure selects, and the energies of optimizer_doc.txt and the
    final refined_sdf.sdf must be the oracle's on the same frames, keyframes and settings."""
    from PIL import Image
    from oracle import oracle
    from test_configs_gpu import centroid
    F, G = 18, 64
    sc = synth.make_scene(N=48, F=F, W=160, H=120, model="LED", perturb=False)
    rng = np.random.default_rng(77)
    images = sc.images.copy()
    for f in (2, 3, 9):
        images[f] = np.clip(images[f] + rng.normal(0, 0.03, images[f].shape).astype(np.float32), 0, 1)
    img8, d16, color, depth = quantised(sc, images, 1e-3)
    poses = relative_poses(sc)
    inp, out = str(tmp_path / "in") + "/", str(tmp_path / "out") + "/"
    os.makedirs(inp + "depth"); os.makedirs(inp + "rgb"); os.makedirs(out)
    for f in range(F):
        Image.fromarray(d16[f]).save(inp + f"depth/{f + 1:03d}.png"); Image.fromarray(img8[f]).save(inp + f"rgb/{f + 1:03d}.png")
    np.savetxt(inp + "intrinsics.txt", sc.K.reshape(3, 3), fmt="%.6f")
    write_pose_file(inp + "pose.txt", [f"{f + 1:03d}" for f in range(F)], poses)
    vs = float(sc.voxel_size)
    cfg = {"input": inp, "output": out, "pose filename": "pose.txt", "datatype": "synth", "first": 0, "last": F - 1, "voxel size": vs, "truncation factor": 5,
           "zmin": 0.05, "zmax": 3.5, "sharpness threshold": 0.03, "model type": "LED", "loss function": "cauchy", "reg albedo": 0.0, "reg norm": 0.1,
           "reg laplacian": 5.0, "max iter": 30, "damping": 3.0, "converge threshold": 1e-3, "lambda": 0.2, "upsample": True,
           "--light": True, "--albedo": True, "--distance": True, "--pose": True, "grid dim": G}
    json.dump(cfg, open(inp + "config.json", "w"))
    r = subprocess.run([EXE, "--config_file", inp + "config.json"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    # ---- keyframes: the focus measure of SharpDetector.h:22-37 decides
    keys = select_keyframes(list(img8), 0.03)
    assert keys == [0, 2, 3, 9, 16]                                          # three sharp frames + the five-frame rule (frame 16)
    assert printed_list(r.stdout, "selected key frame:") == keys and printed_list(r.stdout, "selected key frame after sampling:") == keys
    measures = [float(x) for x in re.findall(r"the sharpness measure is ([-+0-9.eE]+?)\.\n", r.stdout)]
    from test_host_tools import lapm_numpy
    assert len(measures) == F - 1 and np.allclose(measures, [lapm_numpy(img8[f]) for f in range(1, F)], rtol=2e-5)
    # ---- the files of this path (PsOptimizer.cpp:368-384,397-398,419-423; LedOptimizer.cpp:422,430)
    for f in ("upsample_after_5_pointcloud.ply", "upsample_after_5_mesh.ply", "final_refined_pointcloud.ply", "final_refined_mesh.ply", "refined_sdf.sdf",
              "after_iter_3_mesh.ply", "after_iter_6_mesh.ply", "after_poses_opt_6.txt", "init_sdf.sdf", "optimizer_doc.txt"):
        assert os.path.getsize(out + f) > 0, f
    doc = open(out + "optimizer_doc.txt").read()
    assert "num of key frame: 5" in doc and doc.rstrip().endswith("diverged!")
    # ---- the same run through the oracle
    pf = read_pose_file(inp + "pose.txt")
    g = capi.GridDesc(); g.dim[:] = [G, G, G]; g.voxel_size = vs; g.shift[:] = [float(x) for x in centroid(sc.K.reshape(3, 3), depth[0], pf[0])]; g.truncation = 5 * vs
    st = capi.default_settings(capi.LED)
    st.reg_weight_n, st.reg_weight_l, st.damping, st.upsample, st.max_it, st.conv_threshold = 0.1, 5.0, 3.0, 1, 30, 1e-3
    orc = oracle.Oracle(g, sc.K, st, threads=THREADS)
    orc.volume_init(F)
    for f in range(F):
        orc.integrate_frame(color[f], depth[f], orc.estimate_normals(depth[f]), pf[f], f, z_min=0.05, z_max=3.5)
    kp = np.stack([pf[k] for k in keys]).reshape(-1, 16).copy(); kp[0] = np.eye(4, dtype=np.float32).reshape(16)
    orc.set_keyframes(np.array(keys, np.int32), color[keys], kp); orc.init()
    recs, conv = orc.optimize(capi.ALL)
    assert not conv and len(recs) == 7 and recs[5]["upsampled"] and recs[6]["diverged"] and list(orc.info().dim) == [2 * G] * 3
    e_cli = energies(doc)
    e_orc = []
    for it, rr in enumerate(recs):      # the log lines of an iteration: after light, albedo, distance, pose (LedOptimizer.cpp:343-409) + the line after the refinement
        for s in (1, 0, 2, 3):      # light and albedo are logged with the regulariser energies of the previous iteration (E_n / E_l change at the distance block)
            en, el = (rr["e_n_in"], rr["e_l_in"]) if s < 2 else (rr["e_n"], rr["e_l"])
            e_orc.append(float(np.float32(rr["e_after"][s] + rr["reg_weight_n"] * en + rr["reg_weight_l"] * el)))
    assert len(e_cli) == len(e_orc) + 1                                       # (+ the line the reference writes after the refinement, PsOptimizer.cpp:400-405)
    # that line: the old PS energy and Eikonal term + the refined grid's Laplacian energy under its re-normalised weight (= what iteration 6 starts from)
    assert e_cli[24] == pytest.approx(float(np.float32(recs[5]["e_after"][3] + recs[5]["reg_weight_n"] * recs[5]["e_n"] + recs[6]["reg_weight_l"] * recs[6]["e_l_in"])), rel=2e-4)
    head = np.array(e_cli[:24]); tail = np.array(e_cli[25:])
    assert np.allclose(head, e_orc[:24], rtol=2e-4), np.abs(head / np.array(e_orc[:24]) - 1).max()
    assert np.allclose(tail, e_orc[24:], rtol=2e-3), np.abs(tail / np.array(e_orc[24:]) - 1).max()      # one iteration on the refined grid
    # ---- refined_sdf.sdf = -dist of the final (refined) volume over the box of |d| <= sqrt(3) vs
    lines = open(out + "refined_sdf.sdf").read().split("\n")
    dims = list(map(int, lines[0].split())); vs2 = float(lines[2]); assert vs2 == pytest.approx(vs / 2)
    d3 = orc.download_volume()["dist"].reshape(2 * G, 2 * G, 2 * G)
    near = np.abs(d3) <= np.sqrt(3) * np.float32(vs2)
    kk, jj, ii = np.nonzero(near)
    lo = [ii.min(), jj.min(), kk.min()]; hi = [ii.max(), jj.max(), kk.max()]
    assert dims == [hi[a] - lo[a] + 1 for a in range(3)]
    got = np.array([float(x) for x in lines[3:] if x]); want = -d3[lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1].reshape(-1).astype(np.float64)
    inband = np.abs(want) <= np.sqrt(3) * vs2
    err = np.abs(got - want)[inband] / vs2
    assert np.quantile(err, 0.999) <= 2e-4 and np.median(err) <= 2e-5, (np.quantile(err, 0.999), err.max())      # (6 significant digits in the file; 7 iterations incl. a refinement)


def test_tum_layout_with_tracking(built, tmp_path):
    """config_tumrgbd.json's branch: `datatype: "tum"` (associated.txt, depth unit 1/5000, TumrgbdLoader.h:83-119), `pose filename: " "` -> no GT
    poses -> every frame is tracked against the volume (RigidPointOptimizer) before it is fused (main_ps.cpp:241-258); the size probe eats the
    first associated frame because TumrgbdLoader::reset_counter is a no-op (quirk B12).  tracking_poses.txt against the oracle's tracker replaying
    the stream; keyframes = every frame (threshold 0)."""
    from PIL import Image
    from oracle import oracle
    from test_configs_gpu import centroid
    F, G = 13, 128
    # 0.23 deg between frames, strong relief: at this resolution (5 mm voxels, 320x240) every frame converges within a few Gauss-Newton steps.  (The
    # reference's fixed criterion |xi| < 1e-3 after at most 50 steps, RigidOptimizer.h:41-47, is not met on coarser data: the nearest-voxel look-ups of
    # VolumetricGradSdf::tsdf make the iteration hop between two poses -- such frames are simply not fused, main_ps.cpp:245-246.)
    sc = synth.make_scene(N=96, F=F, W=320, H=240, model="SH1", perturb=False, bump=10.0, arc=3.0, zigzag=False)
    img8, d16, color, depth = quantised(sc, sc.images, 1.0 / 5000.0)
    inp, out = str(tmp_path / "in") + "/", str(tmp_path / "out") + "/"
    os.makedirs(inp + "depth"); os.makedirs(inp + "rgb"); os.makedirs(out)
    stamps = [f"{1305031102.175304 + 0.033 * f:.6f}" for f in range(F)]
    with open(inp + "associated.txt", "w") as fh:
        fh.write("# rgb_stamp rgb_file depth_stamp depth_file\n")
        for f in range(F):
            Image.fromarray(d16[f]).save(inp + f"depth/{stamps[f]}.png"); Image.fromarray(img8[f]).save(inp + f"rgb/{stamps[f]}.png")
            fh.write(f"{stamps[f]} rgb/{stamps[f]}.png {stamps[f]} depth/{stamps[f]}.png\n")
    np.savetxt(inp + "intrinsics.txt", sc.K.reshape(3, 3), fmt="%.6f")
    vs = float(sc.voxel_size)
    n_proc = F - 1                                                            # B12: the probe consumed the first frame
    cfg = {"input": inp, "output": out, "pose filename": " ", "datatype": "tum", "first": 0, "last": n_proc - 1, "voxel size": vs, "truncation factor": 5,
           "zmin": 0.05, "zmax": 3.5, "sharpness threshold": 0.0, "model type": "SH1", "loss function": "cauchy", "reg albedo": 0.0, "reg norm": 10.0,
           "reg laplacian": 0.0, "max iter": 2, "damping": 1.0, "converge threshold": 1e-9, "lambda": 0.2, "upsample": False,
           "--light": True, "--albedo": True, "--distance": True, "--pose": True, "grid dim": G}
    json.dump(cfg, open(inp + "config.json", "w"))
    r = subprocess.run([EXE, "--config_file", inp + "config.json"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "GT poses is not avalible!" in r.stdout
    tp = [ln.split() for ln in open(out + "tracking_poses.txt").read().strip().split("\n")]
    assert [t[0] for t in tp] == stamps[1:]                                   # the stream starts one frame late; stamps are the depth stamps of associated.txt
    assert printed_list(r.stdout, "selected key frame:") == list(range(n_proc))
    assert r.stdout.count("Convergence after") == n_proc - 1                  # every tracked frame converged (and was fused)
    from scipy.spatial.transform import Rotation
    cli = []
    for t in tp:
        v = [float(x) for x in t[1:]]
        P = np.eye(4); P[:3, :3] = Rotation.from_quat(v[3:7]).as_matrix(); P[:3, 3] = v[:3]; cli.append(P.astype(np.float32))
    assert np.array_equal(cli[0], np.eye(4, dtype=np.float32))
    # ---- the oracle replays the stream: tracks every frame from the pose voxelPS had before it, fuses at voxelPS's pose
    eye = np.eye(4, dtype=np.float32)
    g = capi.GridDesc(); g.dim[:] = [G, G, G]; g.voxel_size = vs; g.shift[:] = [float(x) for x in centroid(sc.K.reshape(3, 3), depth[1], eye)]; g.truncation = 5 * vs
    st = capi.default_settings(capi.SH1)
    orc = oracle.Oracle(g, sc.K, st, threads=THREADS)
    orc.volume_init(n_proc)
    # The stop rule |xi| < 1e-3 leaves up to one step (1e-3) between two runs that stop one pass apart, so the replay is made pass for pass:
    # voxelPS reports after how many updates k every frame converged; the oracle makes exactly k updates from the same start (threshold 0) --
    # and must then itself see a step below the threshold.
    k_cli = [int(x) for x in re.findall(r"Convergence after (\d+) iterations", r.stdout)]
    assert len(k_cli) == n_proc - 1 and max(k_cli) <= 12
    worst, agree, per_frame = 0.0, 0, []
    for i in range(n_proc):
        f = i + 1
        if i > 0:
            P, iters, conv = orc.track(depth[f], cli[i - 1], z_min=0.05, z_max=3.5, num_iterations=k_cli[i - 1], conv_threshold=0.0, damping=1.0)
            worst = max(worst, float(np.abs(P - cli[i]).max())); per_frame.append((k_cli[i - 1], float(np.abs(P - cli[i]).max())))
            agree += int(orc.track(depth[f], P, z_min=0.05, z_max=3.5, num_iterations=1, conv_threshold=1e-3, damping=1.0)[2])
        orc.integrate_frame(color[f], depth[f], orc.estimate_normals(depth[f]), cli[i], i, z_min=0.05, z_max=3.5)
    assert agree >= n_proc - 2, agree                                         # (a step that sits on the threshold may fall on either side)
    # most frames agree to 1e-7 .. 1e-5 (6 significant digits in the file); where a pixel's nearest voxel differs between the two volumes -- the
    # fusion gates on FALS normals that differ by rounding -- a frame moves by a fraction of a step
    diffs = np.array([d for _, d in per_frame])
    assert np.median(diffs) <= 5e-5 and (diffs <= 2e-6).sum() >= 4 and worst <= 5e-4, per_frame
    # the tracker follows the true motion (frame 1 of the scene is the world frame here)
    P1i = np.linalg.inv(sc.poses_gt[1].reshape(4, 4).astype(np.float64))
    gt = [(P1i @ sc.poses_gt[f].reshape(4, 4).astype(np.float64)) for f in range(1, F)]
    drift = max(np.abs(cli[i][:3, 3] - gt[i][:3, 3]).max() for i in range(n_proc))
    assert drift < 0.008, drift                                               # < 1.5 voxels over the sweep (the reference's tsdf() extrapolation quirk, DESIGN section 8, biases the tracker)
    e = energies(open(out + "optimizer_doc.txt").read())
    assert len(e) == 8 and e[-1] < e[0] and "num of key frame: 12" in open(out + "optimizer_doc.txt").read()
    assert not os.path.exists(out + "final_refined_mesh.ply")                 # max iter exhausted: the reference writes no final files (quirk B13)


def test_more_than_40_keyframes(built, tmp_path):
    """46 selected frames (threshold 0: every frame) -> sampleKeyFrame keeps 40 (main_ps.cpp:312-314,392-421): the list printed after sampling is
    the numpy restatement's, the optimiser runs on 40 keyframes and the after_poses_opt file carries the stamps of exactly those frames"""
    from PIL import Image
    from test_host_tools import sample_keyframes_numpy
    F, G = 46, 48
    sc = synth.make_scene(N=40, F=F, W=96, H=72, model="SH1", perturb=False)
    img8, d16, color, depth = quantised(sc, sc.images, 1e-3)
    poses = relative_poses(sc)
    inp, out = str(tmp_path / "in") + "/", str(tmp_path / "out") + "/"
    os.makedirs(inp + "depth"); os.makedirs(inp + "rgb"); os.makedirs(out)
    for f in range(F):
        Image.fromarray(d16[f]).save(inp + f"depth/{f + 1:03d}.png"); Image.fromarray(img8[f]).save(inp + f"rgb/{f + 1:03d}.png")
    np.savetxt(inp + "intrinsics.txt", sc.K.reshape(3, 3), fmt="%.6f")
    write_pose_file(inp + "pose.txt", [f"{f + 1:03d}" for f in range(F)], poses)
    cfg = {"input": inp, "output": out, "pose filename": "pose.txt", "datatype": "synth", "first": 0, "last": F - 1, "voxel size": float(sc.voxel_size), "truncation factor": 5,
           "zmin": 0.05, "zmax": 3.5, "sharpness threshold": 0.0, "model type": "SH1", "loss function": "cauchy", "reg albedo": 0.0, "reg norm": 10.0,
           "reg laplacian": 0.0, "max iter": 3, "damping": 1.0, "converge threshold": 1e-9, "lambda": 0.2, "upsample": False,
           "--light": True, "--albedo": True, "--distance": True, "--pose": True, "grid dim": G}
    json.dump(cfg, open(inp + "config.json", "w"))
    r = subprocess.run([EXE, "--config_file", inp + "config.json"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert printed_list(r.stdout, "selected key frame:") == list(range(F))
    want = sample_keyframes_numpy(F, 40)
    assert len(want) == 40 and want[-1] == F - 1 and printed_list(r.stdout, "selected key frame after sampling:") == want
    assert "num of key frame: 40" in open(out + "optimizer_doc.txt").read()
    stamps = [ln.split()[0] for ln in open(out + "after_poses_opt_3.txt").read().strip().split("\n")]
    assert stamps == [f"{k + 1:03d}" for k in want]
    e = energies(open(out + "optimizer_doc.txt").read())
    assert len(e) == 12 and e[-1] < e[0]
