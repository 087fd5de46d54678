"""Whole-run parity: psgsdf_optimize against the oracle's restatement of alternatingOptimize, both run to their OWN termination
under the reference's own iteration budgets (config_skorates.json / config_basket_LED.json: max iter 100, converge threshold 5e-3;
PsOptimizer.cpp:303-428, LedOptimizer.cpp:343-478) -- the headline 256^3 x 50 scene, configs[0] (the reference's demo frames) and the LED
configuration with its 2x refinement.

What is held EXACTLY: the number of iterations, every iteration's converged / diverged / upsampled flags, the loop's return value, every
iteration's PCG iteration count, the band of the final grid.

What is held to a YARDSTICK: the state after 10-18 nonlinear iterations.  This algorithm amplifies rounding: the accept rules (|dd| < sqrt(3) vs,
0 < rho < 1), the in-image tests and above all the image GRADIENT -- a finite difference of the pixel cell a projection falls into -- are
discontinuous, so a 1-ulp difference in a pose moves some of 5 million projections across a pixel boundary and the next distance step differs at the
1e-3 level there (profiles/r05_notes.md section 2: the growth curves).  How much of that is the algorithm's own doing is MEASURED in the same test:
the oracle's source is built a second time with multiply-adds contracted into FMAs (oracle/Makefile: what -march=native makes of the reference) and
run on the same inputs.  On the demo frames the two builds of the SAME code end 1.8e-2 apart (norm-wise, 2 004 voxels beyond 1e-4 voxel); on the
headline scene the FMA build even leaves the loop one iteration earlier.  The engine is required to stay within 3x of that yardstick or within the
north star's 1e-4, whichever is larger -- measured: it is 3x CLOSER to the oracle than the oracle's FMA build is (demo frames), 100x closer on the
headline scene.  Round 4's explanation of the stragglers ("all of them have left the band's own width") did not survive the whole runs -- a third of
the headline scene's 47 stragglers sit well inside it -- and is not asserted.
"""
import os

import numpy as np
import pytest

from psgradientsdf_amd import capi, synth

pytestmark = pytest.mark.gpu
THREADS = min(64, os.cpu_count() or 1)
K_YARD = 3.0


def margins_of(a, b, band, vs):
    da = a.download_volume(); db = b.download_volume()
    x = da["dist"][band].astype(np.float64); y = db["dist"][band].astype(np.float64)
    d = np.abs(x - y) / vs
    lim = np.sqrt(3.0) * vs
    w = d > 1e-4
    lb = b.download_light()
    return {"rel": float(np.linalg.norm(x - y) / np.linalg.norm(y)), "q999_vs": float(np.quantile(d, 0.999)), "max_vs": float(d.max()), "above_1e-4_vs": int(w.sum()), "above_1e-3_vs": int((d > 1e-3).sum()),
            "of_those_outside_the_band_width_in_both": int((w & (np.abs(x) > lim) & (np.abs(y) > lim)).sum()), "n_band": int(len(d)),
            "rgb": float(np.abs(da["rgb"][:, band] - db["rgb"][:, band]).max()), "pose": float(np.abs(a.download_poses() - b.download_poses()).max()),
            "light_rel": float(np.abs(a.download_light() - lb).max() / np.abs(lb).max())}


def whole_run(make, vs_final, margins, min_iters, e_floor=1e-4, shipped_course_exact=True):
    """Round 6: oracle and FMA yardstick solve the light / pose blocks as the REFERENCE does (solver_mode 1: one global float Jacobi-PCG); `eng` = the engine
    with the same solver (psgsdf_set_frame_solver(1)): the primary comparison.  `eng_d` = the engine as shipped (direct block solves) against the same
    oracle run: recorded as `shipped_engine_vs_reference_solver`, held to the same yardstick."""
    from oracle import oracle  # noqa: F401
    eng, eng_d, orc, fma = make("eng"), make("eng_d"), make("orc"), make("orc_fma")
    (re_, ce), (ro, co), (rf, cf) = eng.optimize(capi.ALL), orc.optimize(capi.ALL), fma.optimize(capi.ALL)
    rd, cd = eng_d.optimize(capi.ALL)
    # ---- exact: the discrete course of the optimisation
    assert len(re_) == len(ro) >= min_iters and ce == co, (len(re_), len(ro), ce, co)
    assert [(r["converged"], r["diverged"], r["upsampled"]) for r in re_] == [(r["converged"], r["diverged"], r["upsampled"]) for r in ro]
    assert [r["cg_iters"] for r in re_] == [r["cg_iters"] for r in ro]
    assert re_[-1]["converged"] or re_[-1]["diverged"]                      # the loop ended itself (not max iter)
    band = eng.download_band()
    assert np.array_equal(band, orc.download_band())
    # ---- the yardstick: the oracle's FMA build against the oracle, iteration by iteration (it may even leave the loop elsewhere)
    e_eng = [abs(a["e_total"] - b["e_total"]) / abs(b["e_total"]) for a, b in zip(re_, ro)]
    e_fma = [abs(a["e_total"] - b["e_total"]) / abs(b["e_total"]) for a, b in zip(rf, ro)]
    e_fma += [max(e_fma + [1.0])] * (len(ro) - len(e_fma))                   # (a shorter FMA run: its course differs from there on)
    yard_run = np.maximum.accumulate(e_fma)
    same_band = np.array_equal(fma.download_band(), band)
    m_eng = margins_of(eng, orc, band, vs_final)
    m_fma = margins_of(fma, orc, band, vs_final) if same_band else None
    margins(engine_vs_oracle=m_eng, oracle_fma_build_vs_oracle=m_fma, iterations=len(ro), result=bool(co), fma_build_iterations=len(rf),
            e_total_rel_by_iteration={"engine": [float(f"{x:.2e}") for x in e_eng], "oracle_fma_build": [float(f"{x:.2e}") for x in e_fma[:len(rf)]]},
            tolerance=f"max({e_floor:g} (e_total) / 1e-4 (SDF), {K_YARD} x the FMA build's deviation)")
    for i, (x, y) in enumerate(zip(e_eng, yard_run)):
        assert x <= max(e_floor, K_YARD * y), (i, x, y)
    yard = m_fma if m_fma else {"rel": 1.0, "rgb": 1.0, "pose": 1.0, "light_rel": 1.0}
    assert m_eng["rel"] <= max(1e-4, K_YARD * yard["rel"]), (m_eng, m_fma)
    assert m_eng["rgb"] <= max(2e-4, K_YARD * yard["rgb"]) and m_eng["pose"] <= max(1e-5, K_YARD * yard["pose"]) and m_eng["light_rel"] <= max(2e-4, K_YARD * yard["light_rel"]), (m_eng, m_fma)
    # ---- the engine as shipped against the same run of the reference's solver
    same_course = (len(rd) == len(ro) and cd == co and [(r["converged"], r["diverged"], r["upsampled"]) for r in rd] == [(r["converged"], r["diverged"], r["upsampled"]) for r in ro]
                   and [r["cg_iters"] for r in rd] == [r["cg_iters"] for r in ro] and np.array_equal(eng_d.download_band(), band))
    e_d = [abs(a["e_total"] - b["e_total"]) / abs(b["e_total"]) for a, b in zip(rd, ro)]
    m_d = margins_of(eng_d, orc, band, vs_final) if same_course else None
    margins(shipped_engine_vs_reference_solver={"same_discrete_course": bool(same_course), "iterations": len(rd), "state": m_d, "e_total_rel_by_iteration": [float(f"{x:.2e}") for x in e_d]})
    if shipped_course_exact:
        assert same_course, (len(rd), len(ro), cd, co)
    if same_course:
        for i, (x, y) in enumerate(zip(e_d, yard_run)):
            assert x <= max(e_floor, K_YARD * y), ("shipped", i, x, y)
        assert m_d["rel"] <= max(1e-4, K_YARD * yard["rel"]), (m_d, m_fma)
    return m_eng, m_fma


def synth_maker(model, N, F, W, H, **kw):
    from oracle import oracle
    sc = synth.make_scene(N=N, F=F, W=W, H=H, model=model)
    st = capi.default_settings(sc.model_id, **kw)

    def make(kind):
        c = capi.load_engine(sc, sc.K, st, 0) if kind.startswith("eng") else oracle.Oracle(sc, sc.K, st, threads=THREADS, fma=(kind == "orc_fma"), solver_mode=1)
        if kind == "eng":
            c.set_frame_solver(1)
        c.load_scene(sc)
        return c
    return make, float(sc.voxel_size)


def test_headline_scene_to_its_own_termination(built, margins):
    """256^3 x 50 keyframes, SH1, config_skorates.json's settings: 18 iterations to the reference's divergence exit.  Measured (round 5): norm-wise SDF
    error 2.4e-5, 47 of 337 126 band voxels beyond 1e-4 voxel (max 9e-3); the oracle's FMA build: 3.0e-3 and a loop that ends one iteration earlier."""
    make, vs = synth_maker("SH1", 256, 50, 640, 480)
    m_eng, _ = whole_run(make, vs, margins, min_iters=12)
    assert m_eng["rel"] <= 1e-4, m_eng               # on this scene the fixed bar holds outright


def test_config0_demo_frames_to_convergence(built, margins):
    """configs[0]: the reference's demo frames 0-20, 128^3 / 4 mm, config_skorates.json with its real `max iter` 100 / 5e-3 -- 16 iterations to the
    divergence exit on the sub-sampled frames.  Real, textured images: the chaotic case (docstring above)."""
    import test_configs_gpu as tc
    from oracle import oracle
    K, color, depth, poses = tc.load_sokrates()
    vs = 0.004
    g = capi.GridDesc(); g.dim[:] = [128, 128, 128]; g.voxel_size = vs; g.shift[:] = [float(x) for x in tc.centroid(K, depth[0], poses[0])]; g.truncation = 5 * vs
    st = capi.default_settings(capi.SH1)
    base = oracle.Oracle(g, K.reshape(-1), st, threads=THREADS)
    base.volume_init(len(poses))
    for f in range(len(poses)):
        base.integrate_frame(color[f], depth[f], base.estimate_normals(depth[f]), poses[f], f, z_min=0.5, z_max=3.5)
    vo = base.download_volume(); vis = base.download_vis_seq(1); base.close()      # ONE fused volume for all three (the fusion's own parity: tests/test_configs_gpu.py)
    key_poses = np.stack(poses).reshape(-1, 16).copy(); key_poses[0] = np.eye(4, dtype=np.float32).reshape(16)     # B1, main_ps.cpp:139
    imgs = np.stack(color)

    def make(kind):
        c = capi.load_engine(g, K.reshape(-1), st, 0) if kind.startswith("eng") else oracle.Oracle(g, K.reshape(-1), st, threads=THREADS, fma=(kind == "orc_fma"), solver_mode=1)
        if kind == "eng":
            c.set_frame_solver(1)
        c.upload_volume(vo["dist"], vo["grad"], vo["weight"], vo["rgb"], vis, 1)
        c.set_keyframes(np.arange(len(poses), dtype=np.int32), imgs, key_poses); c.init()
        return c
    whole_run(make, vs, margins, min_iters=10)


def test_config3_led_128_with_refinement_to_termination(built, margins):
    """configs[3]'s model and settings (config_basket_LED.json: reg norm 0.1, reg laplacian 5, damping 3, upsample, max iter 100) at 128^3 x 30
    keyframes: light -> albedo -> distance -> pose, the 2x refinement to 256^3 after iteration 5, then the divergence exit."""
    make, vs = synth_maker("LED", 128, 30, 640, 480, reg_weight_n=0.1, reg_weight_l=5.0, damping=3.0, upsample=1)
    m_eng, _ = whole_run(make, vs / 2, margins, min_iters=6)
    assert m_eng["rel"] <= 1e-4, m_eng


@pytest.mark.parametrize("model,kw", [("SH1", {}), ("SH2", {}), ("LED", dict(reg_weight_n=0.1, reg_weight_l=5.0, damping=3.0))])
def test_small_scenes_to_termination(built, margins, model, kw):
    """the three shading models at 64^3 x 12 to their own termination (10 / 11 / 35 iterations; SH1 and LED converge, SH2 takes the divergence exit)"""
    make, vs = synth_maker(model, 64, 12, 320, 240, **kw)
    # SH2: the 9x9 light blocks (float32 in the reference, cond ~2e4) make the light step itself sensitive to the last bits of its inputs whatever solves
    # it (tests/test_parity_gpu.py LIGHT_RTOL_EIGEN); the energy floor is the one every SH2 test uses (tests/test_configs_gpu.py config4: 5e-4), and the
    # shipped engine's direct solves need not follow the reference solver's discrete course to the end (recorded either way)
    whole_run(make, vs, margins, min_iters=8, e_floor=5e-4 if model == "SH2" else 1e-4, shipped_course_exact=model != "SH2")
