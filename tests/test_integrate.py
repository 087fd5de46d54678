"""integrate_frame (VolumetricGradSdf::update).  CPU: the oracle's fusion reproduces the analytic
signed distance near the surface and sets the right visibility bits.  GPU: the HIP kernel against the oracle, then the
whole pipeline (fuse -> init -> optimise) on the fused state."""
import numpy as np
import pytest

from psgradientsdf_amd import capi, synth


def fuse(api, sc, frames=None):
    api.volume_init(sc.F)
    for f in (range(sc.F) if frames is None else frames):
        api.integrate_frame(sc.images[f], sc.depth[f], sc.normals_cam[f], sc.poses_gt[f], f, z_min=0.05, z_max=10.0)


def test_oracle_fusion_known_answers(built):
    from oracle import oracle
    sc = synth.make_scene(N=40, F=6, W=160, H=120, model="SH1", noise=False, perturb=False)
    o = oracle.Oracle(sc, sc.K, capi.default_settings(capi.SH1))
    fuse(o, sc)
    v = o.download_volume()
    vs = float(sc.voxel_size)
    near = (np.abs(sc.dist) < 1.5 * vs) & (v["weight"] > 0)
    assert near.sum() > 2000
    # the projective TSDF (z - p_z along the optical axis, NN depth sample) equals the true distance only for
    # head-on views: unbiased over all near voxels, and tight where every observing camera looks along the normal
    err = (v["dist"][near] - sc.dist[near]) / vs
    assert np.abs(np.median(err)) < 0.1, np.median(err)
    idx = np.nonzero(near)[0]
    N = int(sc.dim[0]); kk, rest = np.divmod(idx, N * N); jj, ii = np.divmod(rest, N)
    origin = sc.shift.astype(np.float64) - 0.5 * vs * sc.dim
    x = origin + vs * np.stack([ii, jj, kk], -1)
    nrm = sc.grad[:, idx].T.astype(np.float64); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    vis = o.download_vis_seq(1)[idx, 0]
    headon = np.ones(len(idx), bool)
    P = sc.poses_gt.reshape(-1, 4, 4).astype(np.float64)
    for f in range(sc.F):
        seen = ((vis >> np.uint64(f)) & np.uint64(1)).astype(bool)
        cosz = -(nrm @ P[f, :3, 2])                        # normal vs optical axis
        headon &= ~seen | (cosz > 0.95)
    assert headon.sum() > 150
    assert np.quantile(np.abs(err[headon]), 0.9) < 0.35, np.quantile(np.abs(err[headon]), 0.9)
    # fused gradient points outward like the analytic one
    g = v["grad"][:, near]; ga = sc.grad[:, near]
    cosang = (g * ga).sum(0) / (np.linalg.norm(g, axis=0) * np.linalg.norm(ga, axis=0) + 1e-12)
    assert np.median(cosang) > 0.97
    # visibility: bit f set <=> weight gained in frame f; fused colour is a convex combination of observed colours
    vis = o.download_vis_seq(1)
    assert ((vis[:, 0] != 0) == (v["weight"] > 0)).all()
    assert v["rgb"].min() >= 0 and v["rgb"].max() <= 1.0
    # untouched voxels keep the init values of VolumetricGradSdf::init
    un = v["weight"] == 0
    assert np.all(v["dist"][un] == sc.truncation) and np.all(v["grad"][:, un] == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["SH1", "LED"])
def test_engine_fusion_matches_oracle_and_feeds_the_optimiser(built, model):
    from oracle import oracle
    sc = synth.make_scene(N=48, F=6, W=160, H=120, model=model)
    st = capi.default_settings(sc.model_id)
    eng = capi.load_engine(sc, sc.K, st, 0); orc = oracle.Oracle(sc, sc.K, st)
    for api in (eng, orc):
        fuse(api, sc)
    ve, vo = eng.download_volume(), orc.download_volume()
    assert np.array_equal(ve["weight"], vo["weight"])
    assert np.array_equal(eng.download_vis_seq(1), orc.download_vis_seq(1))
    for k in ("dist", "grad", "rgb"):
        assert np.abs(ve[k] - vo[k]).max() <= 2e-6 * max(1.0, np.abs(vo[k]).max()), k
    # the fused state is a valid input of the hot path: same band, same optimisation result
    for api in (eng, orc):
        api.set_keyframes(sc.frame_idx, sc.images, sc.poses)
        api.init(); api.init_albedo(); api.normalize_weights()
    assert np.array_equal(eng.download_band(), orc.download_band()) and eng.info().n_band > 5000
    re_, ro = eng.iterate(capi.ALL, 2), orc.iterate(capi.ALL, 2)
    for a, b in zip(re_, ro):
        assert abs(a["e_total"] - b["e_total"]) <= 2e-4 * abs(b["e_total"])
    band = eng.download_band()
    d = np.abs(eng.download_volume()["dist"][band] - orc.download_volume()["dist"][band]) / float(sc.voxel_size)
    assert np.quantile(d, 0.999) <= 1e-4, (np.quantile(d, 0.999), d.max())
