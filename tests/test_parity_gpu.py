"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerances (north_star: <= 1e-4 relative SDF error, denominator = voxel size):
  per-kernel normal equations  rel 2e-5 of the largest entry (oracle accumulates in double)
  one sub-step / one iteration |d_gpu - d_cpu| <= 1e-4 * voxel_size, albedo 1e-4 abs, pose 1e-5, light 1e-4 rel
"""
import numpy as np
import pytest

from conftest import sdf_margin
from psgradientsdf_amd import capi, synth

pytestmark = pytest.mark.gpu

MODELS = [("SH1", capi.SH1), ("SH2", capi.SH2), ("LED", capi.LED)]
# The reference stores the per-frame light normal equations in float32 (Eigen::SparseMatrix<float>), so the light
# step is only determined to cond(H_f) * eps_f32.  SH1 blocks have cond ~1e2, SH2 9x9 blocks ~2e4 on these scenes
# (printed by the diagnosis in profiles/r01_notes.md): 2e4 * 6e-8 * a small factor.
LIGHT_RTOL = {"SH1": 1e-4, "SH2": 1e-3, "LED": 1e-4}      # (achieved on the driver box, profiles/r04_parity_margins.json: SH1 2e-6, SH2 2.6e-4 per sub-step / 1.0e-3 after seven iterations, LED 0)
# Every band voxel, not a quantile: round 3 allowed single voxels up to 5e-3 voxel here; the recorded margins (profiles/r04_parity_margins.json) show the
# maximum at 1e-6 .. 5e-5 voxel after three iterations and after seven iterations THROUGH the 2x refinement alike, so the maximum itself is held to the
# north star's 1e-4.  (Only whole optimisations of 12 iterations let a handful of voxels wander further: tests/test_fullsize_gpu.py)
OPT_MAX_VS = 1e-4


# The solver of the light / pose blocks.  "eigen": engine and oracle BOTH run the reference's algorithm -- one global float
# Jacobi-PCG over all frames' blocks (PsOptimizer.cpp:175-234, LedOptimizer.cpp:134-275; csrc/frame_solve.hip, oracle solver_mode 1): the PRIMARY
# comparison.  "ldlt": the engine's default (each block solved directly in double) against the oracle's direct solves: the same deviation on both sides.
# What the DEFAULT engine deviates from the reference's solver is measured in tests/test_frame_solver_gpu.py::test_default_solver_against_the_references.
SOLVERS = ["eigen", "ldlt"]
# Light tolerance per (solver, model).  Measured (profiles/r06_parity_margins.json): running the reference's solver on BOTH sides does not tighten SH2 -- its 9 x 9
# light blocks have cond ~2e4, so the 1e-8 relative difference between the engine's and the oracle's float normal equations alone moves the step by
# cond x 1e-8 ~ 2e-4 whatever solves them, and with six keyframes the reference's CG does not even reach eps within its 2n passes (info() = NoConvergence on
# both sides, update applied regardless: PsOptimizer.cpp:199-201), which leaves the step to the rounding of the last passes.  With 50 keyframes it converges
# (251 passes on both sides) and the two agree to 1.5e-4 (tests/test_frame_solver_gpu.py::test_full_keyframe_count).
LIGHT_RTOL_EIGEN = {"SH1": 1e-4, "SH2": 1e-3, "LED": 1e-4}


def light_rtol(solver, name):
    return LIGHT_RTOL_EIGEN[name] if solver == "eigen" else LIGHT_RTOL[name]


def make_pair(model_name, model_id, N=48, F=6, solver="ldlt", **kw):
    from oracle import oracle
    sc = synth.make_scene(N=N, F=F, W=160, H=120, model=model_name)
    st = capi.default_settings(model_id, **kw)
    eng = capi.load_engine(sc, sc.K, st, 0)
    orc = oracle.Oracle(sc, sc.K, st, solver_mode=1 if solver == "eigen" else 0)
    if solver == "eigen":
        eng.set_frame_solver(1)
        assert eng.get_tuning()["effective"]["frame_solve"] == "eigen"
    for api in (eng, orc):
        api.load_scene(sc)
    return sc, eng, orc


def relmax(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("name,mid", MODELS)
def test_band_and_init(built, name, mid):
    sc, eng, orc = make_pair(name, mid)
    assert eng.debug_sync_stats()["keyframes_compacted"] == 0      # (rendered floats, not 8-bit data: kept as they are)
    assert np.array_equal(eng.download_band(), orc.download_band())
    assert np.allclose(eng.download_light(), orc.download_light(), rtol=1e-6, atol=1e-7)
    eng.init_albedo(); orc.init_albedo()
    band = eng.download_band()
    ve, vo = eng.download_volume(), orc.download_volume()
    assert np.allclose(ve["rgb"][:, band], vo["rgb"][:, band], rtol=0, atol=2e-6)   # float vs partly-double bilinear weights
    ee, eo = eng.energy(), orc.energy()
    assert np.allclose(ee, eo, rtol=1e-5), (ee, eo)


@pytest.mark.parametrize("name,mid", MODELS)
def test_normal_equations(built, margins, name, mid):
    sc, eng, orc = make_pair(name, mid, reg_weight_l=2.0)
    for api in (eng, orc):
        api.init_albedo(); api.normalize_weights()
    got = {}
    He, be = eng.debug_albedo_system(); Ho, bo = orc.debug_albedo_system()
    got["albedo_H"], got["albedo_b"] = relmax(He, Ho), relmax(be, bo)
    for blk, nm in ((capi.LIGHT, "light"), (capi.POSE, "pose")):
        He, be = eng.debug_frame_system(blk); Ho, bo = orc.debug_frame_system(blk)
        got[nm + "_H"], got[nm + "_b"] = relmax(He, Ho), relmax(be, bo)
    x = np.random.default_rng(0).standard_normal(eng.info().n_band).astype(np.float32)
    de, re_, ye = eng.debug_dist_system(x); do, ro, yo = orc.debug_dist_system(x)
    got["dist_diag"], got["dist_rhs"], got["dist_matvec"] = relmax(de, do), relmax(re_, ro), relmax(ye, yo)
    margins(achieved=got, tolerance=2e-5)
    assert all(v < 2e-5 for v in got.values()), got


@pytest.mark.parametrize("solver", SOLVERS)
@pytest.mark.parametrize("name,mid", MODELS)
def test_substeps(built, margins, name, mid, solver):
    sc, eng, orc = make_pair(name, mid, solver=solver)
    worst = {"dist_vs": 0.0, "rgb": 0.0, "grad": 0.0, "pose": 0.0, "light_rel": 0.0, "e_in_rel": 0.0}
    for api in (eng, orc):
        api.init_albedo(); api.normalize_weights()
    vs = float(sc.voxel_size)
    band = eng.download_band()
    order = [capi.LIGHT, capi.ALBEDO, capi.DIST, capi.POSE] if mid == capi.LED else [capi.ALBEDO, capi.LIGHT, capi.DIST, capi.POSE]
    for blk in order:
        se, so = eng.step(blk), orc.step(blk)
        assert se["n_obs"] == so["n_obs"]
        # accepted updates (albedo channels / frames / voxels); the distance update (fused into the solve kernel's epilogue) counts exactly
        assert abs(se["n_accepted"] - so["n_accepted"]) <= (2 if blk == capi.ALBEDO else 0), (blk, se, so)
        assert abs(se["e_in"] - so["e_in"]) <= (2e-4 if name == "SH2" else 2e-5) * abs(so["e_in"]), (blk, se, so)   # SH2: after the ill-conditioned light step
        if blk == capi.DIST:
            assert abs(se["cg_iters"] - so["cg_iters"]) <= 1 and se["cg_converged"] == so["cg_converged"]
        if solver == "eigen" and blk in (capi.LIGHT, capi.POSE):      # Eigen's iterations() / info() of the global solve, and the LED pose gate
            assert abs(se["cg_iters"] - so["cg_iters"]) <= (8 if name == "SH2" and blk == capi.LIGHT else 1), (blk, se, so)
            # (SH2 with six keyframes: the reference's light solve uses up its 2n = 108 passes a few 1e-6 above eps -- NoConvergence, on both sides -- and
            # applies the update regardless, PsOptimizer.cpp:199-201)
            assert se["cg_converged"] == so["cg_converged"] and se["applied"] == so["applied"] == 1, (blk, se, so)
            assert so["cg_converged"] == 1 or (name == "SH2" and blk == capi.LIGHT), (blk, so)
            worst.setdefault("frame_cg_iters", {})[blk] = (se["cg_iters"], so["cg_iters"])
        ve, vo = eng.download_volume(), orc.download_volume()
        got = {"dist_vs": float(np.abs(ve["dist"][band] - vo["dist"][band]).max() / vs), "rgb": float(np.abs(ve["rgb"][:, band] - vo["rgb"][:, band]).max()),
               "grad": float(np.abs(ve["grad"][:, band] - vo["grad"][:, band]).max()), "pose": float(np.abs(eng.download_poses() - orc.download_poses()).max()),
               "light_rel": relmax(eng.download_light(), orc.download_light()), "e_in_rel": float(abs(se["e_in"] - so["e_in"]) / abs(so["e_in"]))}
        worst.update({k: max(worst[k], got[k]) for k in got})
        assert got["dist_vs"] <= 1e-4, blk
        assert got["rgb"] <= 1e-4, blk
        assert got["grad"] <= 2e-4, blk
        assert got["pose"] <= 1e-5, blk
        assert got["light_rel"] <= light_rtol(solver, name), blk
    margins(achieved=worst, tolerance={"dist_vs": 1e-4, "rgb": 1e-4, "grad": 2e-4, "pose": 1e-5, "light_rel": light_rtol(solver, name), "e_in_rel": 2e-4 if name == "SH2" else 2e-5})


@pytest.mark.parametrize("solver", SOLVERS)
@pytest.mark.parametrize("name,mid", MODELS)
def test_iterations(built, margins, name, mid, solver):
    sc, eng, orc = make_pair(name, mid, solver=solver)
    for api in (eng, orc):
        api.init_albedo(); api.normalize_weights()
    re_, ro = eng.iterate(capi.ALL, 3), orc.iterate(capi.ALL, 3)
    vs = float(sc.voxel_size)
    band = eng.download_band()
    for a, b in zip(re_, ro):
        assert np.allclose(a["e_after"], b["e_after"], rtol=2e-4), (a, b)
        assert abs(a["e_total"] - b["e_total"]) <= 2e-4 * abs(b["e_total"])
    ve, vo = eng.download_volume(), orc.download_volume()
    m = sdf_margin(ve["dist"], vo["dist"], band, vs)
    lrel = relmax(eng.download_light(), orc.download_light()); rgb = float(np.abs(ve["rgb"][:, band] - vo["rgb"][:, band]).max())
    margins(sdf=m, light_rel=lrel, rgb=rgb, e_total_rel=max(abs(a["e_total"] - b["e_total"]) / abs(b["e_total"]) for a, b in zip(re_, ro)), tolerance={"q999_vs": 1e-4, "max_vs": 1e-4, "e_total_rel": 2e-4})
    assert m["q999_vs"] <= 1e-4 and m["max_vs"] <= 1e-4, m      # three iterations: EVERY band voxel inside the north star's tolerance
    assert lrel <= 3 * light_rtol(solver, name), lrel


def test_upsample_and_optimize(built):
    sc, eng, orc = make_pair("SH1", capi.SH1, N=32, F=6, upsample=1, max_it=8, conv_threshold=1e-9)
    (re_, ok_e), (ro, ok_o) = eng.optimize(capi.ALL), orc.optimize(capi.ALL)
    assert len(re_) == len(ro) and ok_e == ok_o
    assert [r["upsampled"] for r in re_] == [r["upsampled"] for r in ro]
    assert eng.info().n_band == orc.info().n_band and tuple(eng.info().dim) == tuple(orc.info().dim)
    assert np.array_equal(eng.download_band(), orc.download_band())
    for a, b in zip(re_, ro):
        assert abs(a["e_total"] - b["e_total"]) <= 1e-3 * abs(b["e_total"]), (a, b)


@pytest.mark.parametrize("solver", SOLVERS)
@pytest.mark.parametrize("name,mid", MODELS)
def test_optimize_matches_oracle(built, margins, name, mid, solver):
    """psgsdf_optimize -- the loop voxelPS calls -- against the oracle's restatement of alternatingOptimize (PsOptimizer.cpp:239-428: albedo ->
    light -> distance -> pose; LedOptimizer.cpp:279-478: light -> albedo -> distance -> pose) for all three shading models: initAlbedo and weight
    normalisation inside the call, the per-iteration records with the energy after EVERY block, the converged / diverged flags that end the loop,
    the 2x refinement at iteration 5 with its new Laplacian weight, and the state the loop leaves behind (band of the refined grid, SDF, albedo,
    poses, light).  On these scenes both loops run 7 iterations (the one after the refinement raises the energy: the reference's divergence exit)."""
    kw = dict(upsample=1, max_it=18, conv_threshold=0.0, damping=10.0)      # (damping 10: the loop survives its divergence test up to the refinement on these scenes)
    if mid == capi.LED:
        kw.update(reg_weight_n=0.1, reg_weight_l=5.0)                       # config_basket_LED.json's regularisers
    sc, eng, orc = make_pair(name, mid, N=24, F=12 if name == "SH2" else 5, solver=solver, **kw)      # (SH2: enough keyframes for the reference's own light solve to converge, tests/golden/make_golden.py)
    (re_, ce), (ro, co) = eng.optimize(capi.ALL), orc.optimize(capi.ALL)
    assert len(re_) == len(ro) >= 6 and ce == co
    assert [(r["converged"], r["diverged"], r["upsampled"]) for r in re_] == [(r["converged"], r["diverged"], r["upsampled"]) for r in ro]
    assert sum(r["upsampled"] for r in re_) == 1 and re_[5]["upsampled"] == 1
    for a, b in zip(re_, ro):
        assert abs(a["e_total"] - b["e_total"]) <= 1e-4 * abs(b["e_total"]), (a["e_total"], b["e_total"])
        assert np.allclose(a["e_after"], b["e_after"], rtol=2e-4), (a["e_after"], b["e_after"])
        assert a["cg_iters"] == b["cg_iters"]
        assert abs(a["reg_weight_l"] - b["reg_weight_l"]) <= 1e-4 * abs(b["reg_weight_l"]) and abs(a["reg_weight_n"] - b["reg_weight_n"]) <= 1e-5 * abs(b["reg_weight_n"])
    assert tuple(eng.info().dim) == tuple(orc.info().dim) == (48, 48, 48)
    band = eng.download_band()
    assert np.array_equal(band, orc.download_band())
    vs = float(sc.voxel_size) / 2
    ve, vo = eng.download_volume(), orc.download_volume()
    m = sdf_margin(ve["dist"], vo["dist"], band, vs)
    le, lo = eng.download_light(), orc.download_light()
    got = {"rgb": float(np.abs(ve["rgb"][:, band] - vo["rgb"][:, band]).max()), "pose": float(np.abs(eng.download_poses() - orc.download_poses()).max()),
           "light_rel": float(np.abs(le - lo).max() / np.abs(lo).max()), "e_total_rel": max(abs(a["e_total"] - b["e_total"]) / abs(b["e_total"]) for a, b in zip(re_, ro))}
    margins(sdf=m, achieved=got, iterations=len(re_), tolerance={"q999_vs": 1e-4, "max_vs": OPT_MAX_VS, "rgb": 2e-3 if name == "SH2" else 2e-4, "pose": 1e-5, "light_rel": 5 * light_rtol(solver, name), "e_total_rel": 1e-4})
    assert m["q999_vs"] <= 1e-4 and m["max_vs"] <= OPT_MAX_VS, m      # north star: <= 1e-4 relative SDF error
    assert got["rgb"] <= (2e-3 if name == "SH2" else 2e-4)
    assert got["pose"] <= 1e-5
    assert got["light_rel"] <= 5 * light_rtol(solver, name)


SCHEDULE_CASES = [("SH1", capi.SH1, 12, dict(damping=10.0, reg_weight_n=10.0)), ("SH2", capi.SH2, 16, dict(damping=1.0, reg_weight_n=0.1)),
                  ("LED", capi.LED, 12, dict(damping=10.0, reg_weight_n=10.0, reg_weight_l=5.0))]


@pytest.mark.parametrize("name,mid,N,kw", SCHEDULE_CASES)
def test_optimize_through_the_laplacian_schedule(built, margins, name, mid, N, kw):
    """psgsdf_optimize run to max_it = 20 (scenes / settings on which the reference's loop survives its own divergence test that long): the
    2x refinement at iteration 5 switches the Laplacian regulariser on with a normalised weight, and the schedule switches it off again --
    LED at iteration 15 exactly (LedOptimizer.cpp:461-463), SH after iteration 15 (PsOptimizer.cpp:411-413) -- every record against the oracle's."""
    sc, eng, orc = make_pair(name, mid, N=N, F=4, upsample=1, max_it=20, conv_threshold=0.0, **kw)
    (re_, ce), (ro, co) = eng.optimize(capi.ALL), orc.optimize(capi.ALL)
    assert len(ro) == 20, len(ro)                       # (the scene was chosen for that)
    assert len(re_) == len(ro) and ce == co
    assert [(r["converged"], r["diverged"], r["upsampled"]) for r in re_] == [(r["converged"], r["diverged"], r["upsampled"]) for r in ro]
    wl_e, wl_o = [r["reg_weight_l"] for r in re_], [r["reg_weight_l"] for r in ro]
    assert [w == 0.0 for w in wl_e] == [w == 0.0 for w in wl_o]
    first_off = next(i for i in range(6, 20) if wl_o[i] == 0.0)
    assert wl_o[6] > 0.0 and first_off == (16 if mid == capi.LED else 17)      # a record carries the weight its iteration ran with: the schedule acts when iteration 15 (LED) / 16 (SH) is closed
    for a, b in zip(re_, ro):
        assert abs(a["reg_weight_l"] - b["reg_weight_l"]) <= 2e-3 * abs(b["reg_weight_l"])
        assert abs(a["e_total"] - b["e_total"]) <= 2e-3 * abs(b["e_total"]), (a["e_total"], b["e_total"])
    band = eng.download_band()
    assert np.array_equal(band, orc.download_band())
    vs = float(sc.voxel_size) / 2
    # 20 nonlinear iterations amplify float rounding (tests/test_wholerun_gpu.py has the argument and the curves): the bound is the north star's 1e-4
    # norm-wise or 3x what the ORACLE'S OWN FMA BUILD -- the same source, multiply-adds contracted -- deviates from the oracle on this very run,
    # whichever is larger; the stragglers beyond 1e-4 voxel are counted and recorded (round 4 asserted q99 <= 1e-3 here: a quantile, ten times the bar)
    from oracle import oracle
    st = capi.default_settings(mid, upsample=1, max_it=20, conv_threshold=0.0, **kw)
    fma = oracle.Oracle(sc, sc.K, st, fma=True); fma.load_scene(sc)
    rf, _ = fma.optimize(capi.ALL)
    x, y = eng.download_volume()["dist"][band].astype(np.float64), orc.download_volume()["dist"][band].astype(np.float64)
    rel = float(np.linalg.norm(x - y) / np.linalg.norm(y)); d = np.abs(x - y) / vs
    yard = None
    if len(rf) == len(ro) and np.array_equal(fma.download_band(), band):
        z = fma.download_volume()["dist"][band].astype(np.float64)
        yard = float(np.linalg.norm(z - y) / np.linalg.norm(y))
    margins(sdf_rel=rel, max_vs=float(d.max()), above_1e_4_vs=int((d > 1e-4).sum()), n_band=int(len(d)), oracle_fma_build_rel=yard, iterations=len(ro), tolerance="max(1e-4, 3 x the FMA build's deviation)")
    assert rel <= max(1e-4, 3.0 * (yard if yard is not None else 1.0)), (rel, yard)


@pytest.mark.parametrize("name,mid", [("SH1", capi.SH1), ("LED", capi.LED)])
def test_albedo_regulariser(built, name, mid):
    """"reg albedo" != 0 (Optimizer.cpp:221-245,593-647): energy, regularised albedo step (matrix-free CG on the engine, assembled
    sparse system in the oracle) and two full iterations"""
    from oracle import oracle
    sc = synth.make_scene(N=40, F=6, W=160, H=120, model=name)
    st = capi.default_settings(mid, reg_weight_rho=0.02)
    eng = capi.load_engine(sc, sc.K, st, 0); orc = oracle.Oracle(sc, sc.K, st, threads=4)
    for api in (eng, orc):
        api.load_scene(sc); api.init_albedo()
    ee, eo = eng.energy(), orc.energy()
    assert abs(ee[3] - eo[3]) <= 2e-5 * abs(eo[3])
    for api in (eng, orc):
        api.normalize_weights()
    band = eng.download_band()
    if mid == capi.LED:
        eng.step(capi.LIGHT); orc.step(capi.LIGHT)
    se, so = eng.step(capi.ALBEDO), orc.step(capi.ALBEDO)
    assert so["cg_iters"] > 1 and abs(se["cg_iters"] - so["cg_iters"]) <= 1 and se["cg_converged"] == so["cg_converged"] == 1
    ve, vo = eng.download_volume(), orc.download_volume()
    assert np.abs(ve["rgb"][:, band] - vo["rgb"][:, band]).max() <= 1e-4
    re_, ro = eng.iterate(capi.ALL, 2), orc.iterate(capi.ALL, 2)
    for a, b in zip(re_, ro):
        assert abs(a["e_r"] - b["e_r"]) <= 2e-4 * abs(b["e_r"]) and b["e_r"] > 0
        assert abs(a["e_total"] - b["e_total"]) <= 2e-4 * abs(b["e_total"])
    ve, vo = eng.download_volume(), orc.download_volume()
    assert np.abs(ve["rgb"][:, band] - vo["rgb"][:, band]).max() <= 2e-4


@pytest.mark.parametrize("name,mid", MODELS)
def test_8bit_keyframes(built, margins, name, mid):
    """psgsdf_set_keyframes_u8 (the reference loader's 8-bit RGB + its conversion factor, ImageLoader.h:167-181): the engine samples
    RGBA8 words, the oracle the converted floats -- same tolerances as the float path, and the engine's two paths agree to rounding noise"""
    from oracle import oracle
    sc = synth.make_scene(N=48, F=6, W=160, H=120, model=name, u8=True)
    assert sc.images_u8.dtype == np.uint8 and np.array_equal(sc.images, sc.images_u8.astype(np.float32) * sc.image_scale)
    st = capi.default_settings(mid)
    eng = capi.load_engine(sc, sc.K, st, 0); eng.load_scene(sc, u8=True)
    import os
    os.environ["PSGSDF_IMG_COMPACT"] = "0"          # (read by psgsdf_create) the float keyframes stay float: the engine's float-image instance
    try:
        engf = capi.load_engine(sc, sc.K, st, 0)
    finally:
        del os.environ["PSGSDF_IMG_COMPACT"]
    engf.load_scene(sc, u8=False)
    # default: float keyframes that ARE 8-bit data (what the reference's loader hands its optimiser: ImageLoader.h:181) are recognised and held as RGBA8
    # words -- the very same run as the 8-bit entry point, bit for bit
    engc = capi.load_engine(sc, sc.K, st, 0); engc.load_scene(sc, u8=False)
    assert engc.debug_sync_stats()["keyframes_compacted"] == 1 and engf.debug_sync_stats()["keyframes_compacted"] == 0 and eng.debug_sync_stats()["keyframes_compacted"] == 0
    orc = oracle.Oracle(sc, sc.K, st, threads=4); orc.load_scene(sc, u8=True)
    for api in (eng, engf, engc, orc):
        api.init_albedo(); api.normalize_weights()
    rc_ = engc.iterate(capi.ALL, 2)
    He, be = eng.debug_albedo_system(); Ho, bo = orc.debug_albedo_system()
    assert relmax(He, Ho) < 2e-5 and relmax(be, bo) < 2e-5
    for blk in (capi.LIGHT, capi.POSE):
        He, be = eng.debug_frame_system(blk); Ho, bo = orc.debug_frame_system(blk)
        assert relmax(He, Ho) < 2e-5 and relmax(be, bo) < 2e-5, (name, blk)
    re_, rf, ro = eng.iterate(capi.ALL, 2), engf.iterate(capi.ALL, 2), orc.iterate(capi.ALL, 2)
    # the two engine paths sample identical floats and sum them in the same fixed order
    assert np.allclose([r["e_total"] for r in re_], [r["e_total"] for r in rf], rtol=1e-4 if name == "SH2" else 1e-6) and [r["cg_iters"] for r in re_] == [r["cg_iters"] for r in rf]
    band = eng.download_band(); vs = float(sc.voxel_size)
    ve, vf, vo = eng.download_volume(), engf.download_volume(), orc.download_volume()
    vc = engc.download_volume()
    assert [r["e_total"] for r in rc_] == [r["e_total"] for r in re_] and np.array_equal(vc["dist"], ve["dist"]) and np.array_equal(vc["rgb"], ve["rgb"]) and np.array_equal(engc.download_poses(), eng.download_poses())
    noise = 2e-4 if name == "SH2" else 5e-6        # the two template instances contract different multiply-adds into FMAs (measured: dist 9e-8 voxel, albedo 2.6e-6); SH2: the ill-conditioned light step amplifies it (LIGHT_RTOL above)
    assert np.abs(ve["dist"][band] - vf["dist"][band]).max() <= 10 * noise * vs and np.abs(ve["rgb"][:, band] - vf["rgb"][:, band]).max() <= noise
    assert np.abs(eng.download_poses() - engf.download_poses()).max() <= noise
    for a, b in zip(re_, ro):
        assert abs(a["e_total"] - b["e_total"]) <= 2e-4 * abs(b["e_total"])
    m = sdf_margin(ve["dist"], vo["dist"], band, vs)
    margins(sdf=m, tolerance={"q999_vs": 1e-4, "max_vs": 1e-4})
    assert m["q999_vs"] <= 1e-4 and m["max_vs"] <= 1e-4, m
    # one allocation per keyframe (psgsdf_set_keyframes_frames: the reference's std::vector<cv::Mat>): the very same run as the one-array entry point
    engp = capi.load_engine(sc, sc.K, st, 0)
    engp.upload_volume(sc.dist, sc.grad, sc.weight, sc.rgb, sc.vis, sc.vis_words)
    engp.set_keyframes_frames(sc.frame_idx, [sc.images[f].copy() for f in range(sc.F)], sc.poses); engp.init()
    engp.init_albedo(); engp.normalize_weights()
    rp = engp.iterate(capi.ALL, 2)
    assert [r["e_total"] for r in rp] == [r["e_total"] for r in rc_] and engp.debug_sync_stats()["keyframes_compacted"] == 1
