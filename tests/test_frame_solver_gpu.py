"""The solver of the light and pose blocks.

The reference hands the block-diagonal normal equations of ALL frames to one Eigen::ConjugateGradient<SparseMatrix<float>> (Jacobi preconditioner,
tolerance eps_f32, at most 2n passes; PsOptimizer.cpp:175-234, LedOptimizer.cpp:134-275) and -- LED poses only -- applies the update only when
info() == Success.  The engine's default solves every block directly (LDL^T in double); `psgsdf_set_frame_solver(ctx, 1)` / PSGSDF_FRAME_SOLVE=eigen
runs the reference's solver itself (csrc/frame_solve.hip).  Here:
  * the solver kernel alone against the oracle's eigen_cg on supplied systems (well / ill conditioned, singular, capped);
  * iterations(), error(), info() and the applied flag of whole steps against the oracle's solver_mode 1;
  * the LED gate actually taken: a keyframe that sees six voxels makes its 6 x 6 block (no damping) too ill-conditioned for a float CG to reach eps within
    2n passes -- the reference then leaves ALL poses alone;
  * what the DEFAULT engine (direct solves) deviates from the reference's solver, measured and bounded as what it is.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import sdf_margin
from psgradientsdf_amd import capi, synth

pytestmark = pytest.mark.gpu

MODELS = [("SH1", capi.SH1), ("SH2", capi.SH2), ("LED", capi.LED)]


def orc_frame_cg(H, b, max_it=0):
    from oracle import oracle
    H = np.ascontiguousarray(H, np.float32); b = np.ascontiguousarray(b, np.float32)
    nb, n = b.shape
    x = np.zeros((nb, n), np.float32); it = C.c_int(); err = C.c_double(); ok = C.c_int()
    oracle.lib().orc_debug_frame_cg(None, C.c_int(nb), C.c_int(n), H.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), C.c_int(max_it),
                                    C.byref(it), C.byref(err), C.byref(ok))
    return x, it.value, err.value, bool(ok.value)


def spd_blocks(rng, nb, n, cond):
    H = np.empty((nb, n, n), np.float32)
    for k in range(nb):
        q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        ev = np.exp(rng.uniform(0.0, np.log(cond), n)) * rng.uniform(0.5, 50.0)
        A = (q * ev) @ q.T
        H[k] = ((A + A.T) / 2).astype(np.float32)
    return H


@pytest.fixture(scope="module")
def any_ctx(built):
    sc = synth.make_scene(N=16, F=2, W=64, H=48, model="SH1")
    return capi.load_engine(sc, sc.K, capi.default_settings(capi.SH1), 0)


@pytest.mark.parametrize("n", [3, 4, 6, 9])
def test_solver_kernel_against_eigen_cg(any_ctx, margins, n):
    """k_frames_eigen on supplied block-diagonal systems = the oracle's eigen_cg: the same iterations(), info() and -- the recurrences being the same float
    operations in the same order -- the same x up to the rare dot product whose double sum rounds to the neighbouring float"""
    rng = np.random.default_rng(100 + n)
    worst = 0.0; exact = 0; total = 0; counts = []
    cases = [(1, 10.0, 0), (7, 1e2, 0), (50, 1e3, 0), (100, 2e4, 0), (2048 // n, 1e2, 0), (50, 1e6, 0), (50, 1e3, 5)]
    for nb, cond, cap in cases:
        H = spd_blocks(rng, nb, n, cond); b = rng.standard_normal((nb, n)).astype(np.float32)
        xe, ie, ee, oe = any_ctx.debug_frame_cg(H, b, cap)
        xo, io, eo, oo = orc_frame_cg(H, b, cap)
        counts.append((nb, cond, cap, ie, io, oe, oo))
        assert oe == oo and abs(ie - io) <= max(1, io // 50), counts[-1]
        if cap:
            assert ie == io == cap and not oe
        if ie == io:      # (the same number of passes: the iterates are the same floats, or a rounding apart at the very end)
            rel = float(np.abs(xe - xo).max() / np.abs(xo).max())
            worst = max(worst, rel); exact += int(np.array_equal(xe, xo)); total += 1
            assert rel <= 1e-5, (counts[-1], rel)
        assert abs(ee - eo) <= 1e-6 + 1e-3 * eo
    # a singular block (a frame without observations: H = 0, b = 0) next to regular ones, and the all-zero right-hand side (Eigen: x = 0, Success, 0 passes)
    H = spd_blocks(rng, 5, n, 10.0); b = rng.standard_normal((5, n)).astype(np.float32); H[2] = 0; b[2] = 0
    xe, ie, ee, oe = any_ctx.debug_frame_cg(H, b); xo, io, eo, oo = orc_frame_cg(H, b)
    assert ie == io and oe == oo and np.allclose(xe, xo, rtol=1e-5, atol=1e-7) and np.all(xe[2] == 0)
    xe, ie, ee, oe = any_ctx.debug_frame_cg(H, np.zeros_like(b))
    assert ie == 0 and oe and ee == 0.0 and not xe.any()
    margins(max_rel_dx=worst, bit_identical=f"{exact} of {total}", iterations=[c[3:5] for c in counts])


def pair(name, mid, N=48, F=6, engine_solver=1, edit=None, **kw):
    from oracle import oracle
    sc = synth.make_scene(N=N, F=F, W=160, H=120, model=name)
    if edit:
        edit(sc)
    st = capi.default_settings(mid, **kw)
    eng = capi.load_engine(sc, sc.K, st, 0); eng.set_frame_solver(engine_solver)
    orc = oracle.Oracle(sc, sc.K, st, solver_mode=1, threads=8)
    for api in (eng, orc):
        api.load_scene(sc); api.init_albedo(); api.normalize_weights()
    return sc, eng, orc


@pytest.mark.parametrize("name,mid", MODELS)
def test_full_keyframe_count(built, margins, name, mid):
    """50 keyframes (the metric's count): n = 200 / 450 / 300 unknowns in the one workgroup; Eigen's iterations() of the light and pose solves against the oracle's"""
    sc, eng, orc = pair(name, mid, N=40, F=50)
    order = [capi.LIGHT, capi.ALBEDO, capi.DIST, capi.POSE] if mid == capi.LED else [capi.ALBEDO, capi.LIGHT, capi.DIST, capi.POSE]
    rec = {}
    for it in range(2):
        for blk in order:
            se, so = eng.step(blk), orc.step(blk)
            if blk in (capi.LIGHT, capi.POSE):
                rec[f"{it}:{blk}"] = (se["cg_iters"], so["cg_iters"], se["cg_error"], so["cg_error"])
                assert se["cg_converged"] == so["cg_converged"] == 1 and se["applied"] == so["applied"] == 1
                assert abs(se["cg_iters"] - so["cg_iters"]) <= max(1, so["cg_iters"] // 20), (blk, se, so)
                assert eng.frame_solver_stats(blk)["cg_iters"] == se["cg_iters"]
    band = eng.download_band(); vs = float(sc.voxel_size)
    m = sdf_margin(eng.download_volume()["dist"], orc.download_volume()["dist"], band, vs)
    le, lo = eng.download_light(), orc.download_light()
    lrel = float(np.abs(le - lo).max() / np.abs(lo).max())
    ltol = 1e-3 if name == "SH2" else 1e-5      # (SH2: cond ~2e4 x the 1e-8 between the two sides' float normal equations; measured 1.5e-4.  SH1 9e-7, LED 0)
    margins(sdf=m, light_rel=lrel, frame_cg=rec, tolerance={"max_vs": 1e-4, "light_rel": ltol})
    assert m["max_vs"] <= 1e-4 and lrel <= ltol, (m, lrel)


def _one_frame_sees_six_voxels(sc, f=2, keep=6):
    vis = sc.vis.reshape(-1).copy()
    seen = np.nonzero((vis >> np.uint64(f)) & np.uint64(1))[0]
    band = np.nonzero(np.abs(sc.dist.reshape(-1)) <= np.sqrt(3) * sc.voxel_size)[0]
    sb = np.intersect1d(seen, band)
    drop = np.setdiff1d(seen, sb[len(sb) // 2: len(sb) // 2 + keep])
    vis[drop] &= ~np.uint64(1 << f)
    sc.vis = vis.reshape(sc.vis.shape)


def test_led_pose_update_is_gated_on_info(built, margins):
    """LedOptimizer.cpp:259-273: `if (solver.info() == Eigen::Success) updatePose(delta_xi)`.  One keyframe that sees six voxels, no damping: the global CG
    uses up its 2n = 72 passes above eps -- NoConvergence -- and NO pose moves (all frames: it is one solve).  The engine's solver (mode 1) takes the same
    exit; its default (direct solves) has no such state and applies the step -- the deviation DESIGN.md section 2 lists."""
    sc, eng, orc = pair("LED", capi.LED, N=32, F=6, edit=_one_frame_sees_six_voxels, damping=0.0)
    for blk in (capi.LIGHT, capi.ALBEDO, capi.DIST):
        eng.step(blk); orc.step(blk)
    P0e, P0o = eng.download_poses().copy(), orc.download_poses().copy()
    se, so = eng.step(capi.POSE), orc.step(capi.POSE)
    assert so["cg_iters"] == 72 and so["cg_converged"] == 0 and so["applied"] == 0, so      # (the case was built for that)
    assert se["cg_iters"] == 72 and se["cg_converged"] == 0 and se["applied"] == 0 and se["n_accepted"] == 0, se
    assert np.array_equal(eng.download_poses(), P0e) and np.array_equal(orc.download_poses(), P0o)
    # the loop goes on from there on both sides alike
    re_, ro = eng.iterate(capi.ALL, 1), orc.iterate(capi.ALL, 1)
    assert abs(re_[0]["e_total"] - ro[0]["e_total"]) <= 2e-4 * abs(ro[0]["e_total"])
    # SH models apply the update whatever info() says (PsOptimizer.cpp:232), and so does the LED model without the quirk switch
    sc2, eng2, orc2 = pair("LED", capi.LED, N=32, F=6, edit=_one_frame_sees_six_voxels, damping=0.0, ref_quirks=0)
    for blk in (capi.LIGHT, capi.ALBEDO, capi.DIST):
        eng2.step(blk); orc2.step(blk)
    s2e, s2o = eng2.step(capi.POSE), orc2.step(capi.POSE)
    assert s2e["cg_converged"] == s2o["cg_converged"] == 0 and s2e["applied"] == s2o["applied"] == 1
    # the default engine: direct solves, update applied
    sc3, eng3, _ = pair("LED", capi.LED, N=32, F=6, engine_solver=0, edit=_one_frame_sees_six_voxels, damping=0.0)
    for blk in (capi.LIGHT, capi.ALBEDO, capi.DIST):
        eng3.step(blk)
    P3 = eng3.download_poses().copy()
    s3 = eng3.step(capi.POSE)
    assert s3["applied"] == 1 and not np.array_equal(eng3.download_poses(), P3)
    margins(reference_solver={"iters": se["cg_iters"], "error": se["cg_error"], "applied": se["applied"]}, oracle={"iters": so["cg_iters"], "error": so["cg_error"]})


@pytest.mark.parametrize("name,mid", MODELS)
def test_default_solver_against_the_references(built, margins, name, mid):
    """The engine as shipped (every block solved directly in double) against the oracle running the REFERENCE's solver: what the substitution costs in
    parity.  SH1, LED and every pose block: nothing measurable (the blocks' condition numbers are ~1e2: the float CG determines the step to 1e-6).
    SH2: the 9 x 9 light blocks have cond ~2e4, a float CG that stops at ||r|| <= eps ||b|| leaves the step undetermined to ~1e-3 of itself, and the two
    solvers land on different points of that interval: light 1e-3 relative after ONE iteration, single voxels a few 1e-4 voxel, the norm-wise SDF error
    still inside the north star's 1e-4.  Asserted as what it is; `psgsdf_set_frame_solver(1)` removes it (test_parity_gpu.py, solver = "eigen")."""
    sc, eng, orc = pair(name, mid, engine_solver=0)
    band = eng.download_band(); vs = float(sc.voxel_size)
    rec = {}
    for its in (1, 3):
        eng.iterate(capi.ALL, 1 if its == 1 else 2); orc.iterate(capi.ALL, 1 if its == 1 else 2)
        ve, vo = eng.download_volume(), orc.download_volume()
        m = sdf_margin(ve["dist"], vo["dist"], band, vs)
        le, lo = eng.download_light(), orc.download_light()
        rec[f"after_{its}"] = {"sdf": m, "light_rel": float(np.abs(le - lo).max() / np.abs(lo).max()), "rgb": float(np.abs(ve["rgb"][:, band] - vo["rgb"][:, band]).max()),
                               "pose": float(np.abs(eng.download_poses() - orc.download_poses()).max())}
    tol = {"sdf_rel": 1e-4, "q999_vs": 1e-4, "max_vs": 2e-3 if name == "SH2" else 1e-4, "light_rel": 5e-3 if name == "SH2" else 1e-4, "pose": 1e-5}
    margins(achieved=rec, tolerance=tol)
    for k, r in rec.items():
        assert r["sdf"]["rel"] <= tol["sdf_rel"] and r["sdf"]["q999_vs"] <= tol["q999_vs"] and r["sdf"]["max_vs"] <= tol["max_vs"], (k, r)
        assert r["light_rel"] <= tol["light_rel"] and r["pose"] <= tol["pose"], (k, r)
