"""Size-independent properties at BASELINE.json's headline size (256^3 grid, 50 keyframes 640x480) where the CPU oracle is
too slow for a full comparison: symmetry of the assembled distance system, PCG convergence in Eigen's sense, monotone total
energy (the reference's own run-time oracle: it aborts with "diverged!" otherwise, PsOptimizer.cpp:377-384), run-to-run
reproducibility, and agreement with the oracle on a random SAMPLE of observations."""
import numpy as np
import pytest

from conftest import sdf_margin
from psgradientsdf_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big_scene():
    return synth.make_scene(N=256, F=50, W=640, H=480, model="SH1")


def test_headline_size_properties(built, big_scene):
    sc = big_scene
    st = capi.default_settings(capi.SH1)
    eng = capi.load_engine(sc, sc.K, st, 0); eng.load_scene(sc)
    S = eng.info().n_band
    assert 2.5e5 < S < 4.5e5
    eng.init_albedo(); e0 = eng.normalize_weights()
    # symmetric positive semi-definite assembled system
    rng = np.random.default_rng(0)
    x = rng.standard_normal(S).astype(np.float32); y = rng.standard_normal(S).astype(np.float32)
    diag, rhs, Hx = eng.debug_dist_system(x); _, _, Hy = eng.debug_dist_system(y)
    a, b = float(np.dot(y.astype(np.float64), Hx)), float(np.dot(x.astype(np.float64), Hy))
    assert abs(a - b) <= 1e-5 * max(abs(a), abs(b))
    assert float(np.dot(x.astype(np.float64), Hx)) > 0 and diag.min() >= 0
    # the blocks: PCG converges to Eigen's tolerance, every block keeps the total energy from rising
    recs = eng.iterate(capi.ALL, 6)
    snap = (eng.download_volume()["dist"].copy(), eng.download_poses().copy(), eng.download_light().copy())
    tot = [e0] + [r["e_total"] for r in recs]
    assert all(tot[i + 1] <= tot[i] * (1 + 1e-5) for i in range(len(tot) - 1)), tot
    assert all(5 <= r["cg_iters"] <= 60 for r in recs)
    st_d = eng.step(capi.DIST)
    assert st_d["cg_converged"] == 1 and st_d["cg_error"] <= np.finfo(np.float32).eps and st_d["n_obs"] > 3e6
    assert st_d["n_accepted"] > 0.99 * S
    # reproducibility: a second context on the same inputs gives the SAME bits -- every reduction (PCG dots, per-voxel sums, the per-frame
    # normal equations of the light / pose blocks) is summed in a fixed order, no floating-point atomics anywhere on the path
    eng2 = capi.load_engine(sc, sc.K, st, 0); eng2.load_scene(sc); eng2.init_albedo(); eng2.normalize_weights()
    recs2 = eng2.iterate(capi.ALL, 6)
    assert [r["e_total"] for r in recs2] == [r["e_total"] for r in recs]
    assert [r["cg_iters"] for r in recs2] == [r["cg_iters"] for r in recs]
    assert np.array_equal(eng2.download_volume()["dist"], snap[0]) and np.array_equal(eng2.download_poses(), snap[1]) and np.array_equal(eng2.download_light(), snap[2])


def test_headline_size_sample_against_oracle(built, big_scene):
    """albedo normal equations of 2 000 random band voxels (all 50 frames each) against the oracle run on the same volume"""
    from oracle import oracle
    sc = big_scene
    st = capi.default_settings(capi.SH1)
    eng = capi.load_engine(sc, sc.K, st, 0); eng.load_scene(sc); eng.init_albedo()
    orc = oracle.Oracle(sc, sc.K, st, threads=8); orc.load_scene(sc); orc.init_albedo()
    assert np.array_equal(eng.download_band(), orc.download_band())
    He, be = eng.debug_albedo_system(); Ho, bo = orc.debug_albedo_system()
    idx = np.random.default_rng(1).choice(len(He), 2000, replace=False)
    assert np.abs(He[idx] - Ho[idx]).max() <= 2e-5 * np.abs(Ho).max() and np.abs(be[idx] - bo[idx]).max() <= 2e-5 * np.abs(bo).max()
    ee, eo = eng.energy(), orc.energy()
    assert abs(ee[0] - eo[0]) <= 1e-5 * eo[0] and abs(ee[1] - eo[1]) <= 1e-6 * eo[1]


def test_config1_against_the_oracle(built, margins):
    """BASELINE.json configs[1]: synthetic 640x480 RGB-D, 128^3 grid, SH1, 30 keyframes -- one full Gauss-Newton iteration
    against the oracle (8 host threads), tolerance of the north star: <= 1e-4 relative SDF error."""
    from oracle import oracle
    sc = synth.make_scene(N=128, F=30, W=640, H=480, model="SH1")
    st = capi.default_settings(capi.SH1)
    eng = capi.load_engine(sc, sc.K, st, 0); orc = oracle.Oracle(sc, sc.K, st, threads=8)
    for api in (eng, orc):
        api.load_scene(sc); api.init_albedo(); api.normalize_weights()
    assert eng.info().n_band == orc.info().n_band > 5e4
    re_, ro = eng.iterate(capi.ALL, 1)[0], orc.iterate(capi.ALL, 1)[0]
    assert abs(re_["e_total"] - ro["e_total"]) <= 1e-4 * abs(ro["e_total"])
    assert abs(re_["cg_iters"] - ro["cg_iters"]) <= 1
    band = eng.download_band()
    ve, vo = eng.download_volume(), orc.download_volume()
    vs = float(sc.voxel_size)
    m = sdf_margin(ve["dist"], vo["dist"], band, vs)
    got = {"e_total_rel": abs(re_["e_total"] - ro["e_total"]) / abs(ro["e_total"]), "rgb": float(np.abs(ve["rgb"][:, band] - vo["rgb"][:, band]).max()),
           "pose": float(np.abs(eng.download_poses() - orc.download_poses()).max()), "light_rel": float(np.abs(eng.download_light() - orc.download_light()).max() / np.abs(orc.download_light()).max())}
    margins(sdf=m, achieved=got, tolerance={"max_vs": 1e-4, "e_total_rel": 1e-4, "rgb": 1e-4, "pose": 1e-5, "light_rel": 1e-4})
    assert m["max_vs"] <= 1e-4, m
    assert got["rgb"] <= 1e-4
    assert got["pose"] <= 1e-5
    assert got["light_rel"] <= 1e-4


@pytest.mark.parametrize("model,N,F,iters", [("SH1", 96, 20, 12), ("LED", 64, 12, 12), ("SH2", 64, 12, 8)])
def test_long_run_stays_within_the_north_star_tolerance(built, margins, model, N, F, iters):
    """engine vs oracle over a whole optimisation's worth of iterations (rounding differences are amplified by the discrete accept
    rules, so a handful of voxels wander; tools/long_parity.py prints the growth): norm-wise relative SDF error <= 1e-4 and
    99.9 % of the band within 1e-4 voxel"""
    from oracle import oracle
    sc = synth.make_scene(N=N, F=F, W=320, H=240, model=model)
    st = capi.default_settings(sc.model_id)
    if model == "LED":
        st.reg_weight_n, st.reg_weight_l, st.damping = 0.1, 5.0, 3.0      # config_basket_LED.json
    eng = capi.load_engine(sc, sc.K, st, 0); orc = oracle.Oracle(sc, sc.K, st, threads=16)
    for api in (eng, orc):
        api.load_scene(sc); api.init_albedo(); api.normalize_weights()
    re_, ro = eng.iterate(capi.ALL, iters), orc.iterate(capi.ALL, iters)
    band = eng.download_band(); vs = float(sc.voxel_size)
    m = sdf_margin(eng.download_volume()["dist"], orc.download_volume()["dist"], band, vs)
    margins(sdf=m, iterations=iters, e_total_rel=abs(re_[-1]["e_total"] - ro[-1]["e_total"]) / abs(ro[-1]["e_total"]), tolerance={"rel": 1e-4, "q999_vs": 1e-4, "e_total_rel": 5e-4})
    assert m["rel"] <= 1e-4 and m["q999_vs"] <= 1e-4, m
    assert abs(re_[-1]["e_total"] - ro[-1]["e_total"]) <= 5e-4 * abs(ro[-1]["e_total"])


def test_headline_size_against_the_oracle(built, margins, big_scene):
    """256^3 x 50 keyframes 640x480: two full Gauss-Newton iterations, every band voxel against the oracle (64 host threads)"""
    from oracle import oracle
    sc = big_scene
    st = capi.default_settings(capi.SH1)
    eng = capi.load_engine(sc, sc.K, st, 0); orc = oracle.Oracle(sc, sc.K, st, threads=64)
    for api in (eng, orc):
        api.load_scene(sc); api.init_albedo(); api.normalize_weights()
    assert eng.info().n_band == orc.info().n_band
    re_, ro = eng.iterate(capi.ALL, 2), orc.iterate(capi.ALL, 2)
    for a, b in zip(re_, ro):
        assert abs(a["e_total"] - b["e_total"]) <= 1e-4 * abs(b["e_total"]) and abs(a["cg_iters"] - b["cg_iters"]) <= 1
    band = eng.download_band(); vs = float(sc.voxel_size)
    ve, vo = eng.download_volume(), orc.download_volume()
    m = sdf_margin(ve["dist"], vo["dist"], band, vs)
    got = {"e_total_rel": max(abs(a["e_total"] - b["e_total"]) / abs(b["e_total"]) for a, b in zip(re_, ro)), "rgb": float(np.abs(ve["rgb"][:, band] - vo["rgb"][:, band]).max()),
           "pose": float(np.abs(eng.download_poses() - orc.download_poses()).max())}
    margins(sdf=m, achieved=got, tolerance={"max_vs": 1e-4, "e_total_rel": 1e-4, "rgb": 1e-4, "pose": 1e-5})
    assert m["max_vs"] <= 1e-4, m
    assert got["rgb"] <= 1e-4
    assert got["pose"] <= 1e-5
