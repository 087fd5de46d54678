"""FALS normal estimation (NormalEstimator.h) and the depth tracker (RigidPointOptimizer.cpp).
CPU: oracle known-answer tests.  GPU: HIP kernels against the oracle."""
import numpy as np
import pytest

from psgradientsdf_amd import capi, synth


def fuse(api, sc, frames, poses):
    api.volume_init(sc.F)
    for f in frames:
        api.integrate_frame(sc.images[f], sc.depth[f], sc.normals_cam[f], poses[f], f, z_min=0.05, z_max=10.0)


def perturbed(P, dt, dw):
    from oracle import oracle
    Q = P.reshape(4, 4).astype(np.float64).copy()
    Q[:3, :3] = Q[:3, :3] @ oracle.so3_exp(np.asarray(dw)).astype(np.float64)
    Q[:3, 3] += np.asarray(dt)
    return Q.astype(np.float32)


def test_oracle_normals_and_tracker_known_answers(built):
    from oracle import oracle
    sc = synth.make_scene(N=48, F=6, W=160, H=120, model="SH1", noise=False, perturb=False)
    o = oracle.Oracle(sc, sc.K, capi.default_settings(capi.SH1))
    # FALS normals of the rendered depth agree with the analytic (inward, camera-frame) normals away from the silhouette
    n = o.estimate_normals(sc.depth[0])
    valid = sc.depth[0] > 0
    core = valid.copy()
    for _ in range(8):                                      # erode: the 11x11 window must stay on the object
        core[1:-1, 1:-1] &= core[:-2, 1:-1] & core[2:, 1:-1] & core[1:-1, :-2] & core[1:-1, 2:]
    assert core.sum() > 500
    cosang = (n * sc.normals_cam[0]).sum(0)[core]
    assert np.median(cosang) > 0.995 and np.quantile(cosang, 0.05) > 0.97, (np.median(cosang), np.quantile(cosang, 0.05))
    assert np.allclose(np.linalg.norm(n[:, core], axis=0), 1.0, atol=1e-5)
    # tracker: a short video-like sweep of a strongly bumpy object (a plain sphere leaves the rotation about its centre
    # unobservable); fuse frames 0..8 at their true poses, then track frame 9 from a 2-voxel translation offset.
    # The fused projective TSDF is ~1 voxel rms noisy at this resolution and the reference's tsdf() extrapolates with
    # (x_voxel - p) although its distances grow outward (VolumetricGradSdf.h:87 vs VolumetricGradSdf.cpp:101), so the
    # iteration does not meet its own |xi|^2 < 1e-6 stop; what it must do is pull the pose towards the truth.
    sc = synth.make_scene(N=64, F=10, W=160, H=120, model="SH1", noise=False, perturb=False, bump=6.0, arc=40.0)
    o = oracle.Oracle(sc, sc.K, capi.default_settings(capi.SH1))
    fuse(o, sc, range(9), sc.poses_gt)
    P_true = sc.poses_gt[9].reshape(4, 4)
    P0 = P_true.copy(); P0[:3, 3] += np.array([0.012, -0.008, 0.006], np.float32)
    P1, iters, conv = o.track(sc.depth[9], P0)
    e0 = np.linalg.norm(P0[:3, 3] - P_true[:3, 3]); e1 = np.linalg.norm(P1[:3, 3] - P_true[:3, 3])
    assert iters > 0 and e1 < 0.7 * e0, (e0, e1, iters)
    assert np.allclose(P1[:3, :3] @ P1[:3, :3].T, np.eye(3), atol=1e-5)
    # no valid depth -> no measurement -> "not converged" after 0 iterations, pose untouched
    P2, it2, conv2 = o.track(np.zeros_like(sc.depth[9]), P0)
    assert it2 == 0 and not conv2 and np.array_equal(P2, P0)


@pytest.mark.gpu
def test_engine_normals_and_tracker_match_oracle(built):
    from oracle import oracle
    sc = synth.make_scene(N=48, F=6, W=160, H=120, model="SH1", bump=6.0)
    st = capi.default_settings(capi.SH1)
    eng = capi.load_engine(sc, sc.K, st, 0); orc = oracle.Oracle(sc, sc.K, st)
    ne, no = eng.estimate_normals(sc.depth[1]), orc.estimate_normals(sc.depth[1])
    ok = np.isfinite(no).all(0) & (sc.depth[1] > 0)
    assert ok.sum() > 1000 and np.abs(ne[:, ok] - no[:, ok]).max() <= 2e-5
    for api in (eng, orc):
        fuse(api, sc, [0, 1, 5], sc.poses_gt)
    P0 = perturbed(sc.poses_gt[0], [0.003, 0.002, -0.002], [-0.003, 0.002, 0.001])
    # nearest-voxel look-ups make the iteration piecewise: compare a few passes (identical look-ups), not 50
    Pe, ie, ce = eng.track(sc.depth[0], P0, num_iterations=3); Po, io, co = orc.track(sc.depth[0], P0, num_iterations=3)
    assert ce == co and ie == io
    assert np.abs(Pe - Po).max() <= 2e-5, np.abs(Pe - Po).max()


@pytest.mark.gpu
@pytest.mark.parametrize("W,H", [(160, 120), (380, 570), (1139, 1709), (7, 5)])
def test_normals_cache_on_the_device_is_the_host_formula_bit_for_bit(built, W, H):
    """NormalEstimator::cache (NormalEstimator.h:52-125) runs on the device since round 5 (it was 0.15-0.5 s of one host core at the demo's native
    1139 x 1709).  Restated here in numpy float64 -- the same products, the (2r+1)^2 box sums as eleven adds per direction in the order k = -r .. r with
    OpenCV's BORDER_REFLECT_101, the cofactor inverse -- it must agree in every bit of the nine float planes; (7, 5): an image smaller than the window."""
    sc = synth.make_scene(N=16, F=2, W=64, H=48, model="SH1")
    K = np.array([[525.3 * W / 640, 0, 0.5 * W - 0.37], [0, 524.1 * H / 480, 0.5 * H + 0.21], [0, 0, 1]], np.float32)
    eng = capi.load_engine(sc, K.reshape(-1), capi.default_settings(capi.SH1), 0)
    got = eng.debug_normals_cache(W, H)
    eng.close()
    f64 = np.float64
    fx_inv, fy_inv, cx, cy = f64(1.0) / f64(K[0, 0]), f64(1.0) / f64(K[1, 1]), f64(K[0, 2]), f64(K[1, 2])
    x0 = (fx_inv * (np.arange(W, dtype=f64) - cx))[None, :] * np.ones((H, 1)); y0 = (fy_inv * (np.arange(H, dtype=f64) - cy))[:, None] * np.ones((1, W))
    nsi = 1.0 / (1.0 + x0 * x0 + y0 * y0)
    a = [x0 * x0 * nsi, x0 * y0 * nsi, x0 * nsi, y0 * y0 * nsi, y0 * nsi, nsi]

    def reflect(i, n):
        i = np.asarray(i).copy()
        if n == 1:
            return np.zeros_like(i)
        while ((i < 0) | (i >= n)).any():
            i = np.where(i < 0, -i, i); i = np.where(i >= n, 2 * n - 2 - i, i)
        return i

    def box(v, r=5):
        t = np.zeros_like(v)
        for k in range(-r, r + 1):
            t = t + v[:, reflect(np.arange(W) + k, W)]
        o = np.zeros_like(v)
        for k in range(-r, r + 1):
            o = o + t[reflect(np.arange(H) + k, H), :]
        return o
    M11, M12, M13, M22, M23, M33 = (box(v) for v in a)
    det = M11 * (M22 * M33) + 2 * M12 * (M23 * M13) - (M13 * (M13 * M22) + M12 * (M12 * M33) + M23 * (M23 * M11))
    di = 1.0 / det
    want = np.stack([x0 * nsi, y0 * nsi, nsi, di * (M22 * M33 - M23 * M23), di * (M13 * M23 - M12 * M33), di * (M12 * M23 - M13 * M22),
                     di * (M11 * M33 - M13 * M13), di * (M12 * M13 - M11 * M23), di * (M11 * M22 - M12 * M12)]).astype(np.float32)
    assert np.array_equal(got, want), np.abs(got.astype(f64) - want).max()
