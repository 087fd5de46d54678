"""More known-answer tests that pin the CPU oracle ("What's weak" 3): the LED pose block with its A_c term, the
Eikonal row, the Laplacian diagonal, and the assembled distance system against an explicit J^T W J built from per-observation rows.

All of them differentiate the oracle's own forward model numerically (the enabled form of the reference's disabled diagnostic,
PsOptimizerJa.cpp:293-318,514-517) or rebuild a matrix from independently probed pieces -- nothing here compares the oracle with itself."""
import numpy as np
import pytest

from psgradientsdf_amd import capi, synth
from oracle import oracle

from test_oracle_kat import make, visible_obs


def _voxel_state(o, sc, lin):
    i = o.info()
    dim = np.array(i.dim[:], np.int64); org = np.array(i.origin[:], np.float64); vs = float(i.voxel_size)
    idx = np.array([lin % dim[0], (lin // dim[0]) % dim[1], lin // (dim[0] * dim[1])], np.float64)
    xv = org + vs * idx
    g = o.peek_grad(lin).astype(np.float64); n = g / np.linalg.norm(g)
    return xv, n, float(o.peek_dist(lin)), o.peek_rgb(lin).astype(np.float64)


def _numeric_pose_block(o, j, f, P0, ht=2e-5, hw=2e-5):
    num = np.zeros((3, 6))
    for k in range(6):
        res = []
        for s in (+1, -1):
            P = P0.copy()
            if k < 3:
                P[k, 3] += s * ht
            else:
                w = np.zeros(3); w[k - 3] = s * hw
                P[:3, :3] = P[:3, :3] @ oracle.so3_exp(w).astype(np.float64)
            o.poke_pose(f, P.astype(np.float32))
            okr, r, _ = o.probe_residual(j, f)
            assert okr
            res.append(r.astype(np.float64))
        o.poke_pose(f, P0.astype(np.float32))
        num[:, k] = (res[0] - res[1]) / (2 * (ht if k < 3 else hw))
    return num


def test_led_pose_jacobian_shading_terms():
    """LedOptimizerJa.cpp:63-78 on CONSTANT images (image term G = 0), so that the block is the shading part alone:
       translation columns = A_c = -(rho_c L_c / |p|^3) n^T          (the term the reference keeps)
       numeric d r / d t   = A_c + 3 rho_c L_c (n.q) q^T / |p|^5     (q = x_s - t; the fall-off derivative the reference drops)
       rotation columns    = 0 analytically ([p]x p = 0) and numerically (R p = x_s - t does not depend on R)."""
    sc = synth.make_scene(N=32, F=5, W=128, H=96, model="LED", noise=False)
    sc.images = np.full_like(sc.images, 0.5)
    o = oracle.Oracle(sc, sc.K, capi.default_settings(capi.LED)); o.load_scene(sc)
    o.init_albedo(); o.update_grad()
    band = o.download_band(); L = o.download_light().astype(np.float64).reshape(-1)[:3]
    P = o.download_poses().reshape(-1, 4, 4).astype(np.float64)
    checked = 0
    for j, f in visible_obs(o, 15):
        ok, J = o.probe_pose_jacobian(j, f)
        if not ok:
            continue
        xv, n, d, rho = _voxel_state(o, sc, int(band[j]))
        q = (xv - d * n) - P[f, :3, 3]; pn = np.linalg.norm(q)
        A = -np.outer(rho * L, n) / pn ** 3
        assert np.abs(J[:, :3] - A).max() <= 2e-5 * np.abs(A).max(), (j, f, J[:, :3], A)
        assert np.abs(J[:, 3:]).max() <= 1e-6 * np.abs(A).max()
        num = _numeric_pose_block(o, j, f, P[f])
        dropped = 3.0 * np.outer(rho * L, q) * float(n @ q) / pn ** 5
        tol = 0.02 * np.abs(A).max() + 2e-2          # central differences of a float32 residual with a 2e-5 step
        assert np.abs(num[:, :3] - (A + dropped)).max() <= tol, (j, f, num[:, :3], A + dropped)
        assert np.abs(num[:, 3:]).max() <= tol
        assert np.abs(dropped).max() > 5 * tol        # the dropped term is far above the noise floor: the check above resolves it
        checked += 1
    assert checked >= 8


def test_led_pose_jacobian_numeric():
    """full LED pose block on rendered images: numeric d r / d (t, R) = analytic block + the dropped fall-off term (translation only)"""
    sc, o = make("LED")
    o.init_albedo(); o.update_grad()
    band = o.download_band(); L = o.download_light().astype(np.float64).reshape(-1)[:3]
    P = o.download_poses().reshape(-1, 4, 4).astype(np.float64)
    checked = 0
    for j, f in visible_obs(o, 12):
        ok, J = o.probe_pose_jacobian(j, f)
        if not ok:
            continue
        xv, n, d, rho = _voxel_state(o, sc, int(band[j]))
        q = (xv - d * n) - P[f, :3, 3]; pn = np.linalg.norm(q)
        full = J.astype(np.float64).copy()
        full[:, :3] += 3.0 * np.outer(rho * L, q) * float(n @ q) / pn ** 5
        num = _numeric_pose_block(o, j, f, P[f])
        scale = max(np.abs(J).max(), 1e-2)
        assert np.abs(num - full).max() <= 0.08 * scale + 5e-2, (j, f, num, full)
        checked += 1
    assert checked >= 6


@pytest.mark.parametrize("model", ["SH1", "LED"])
def test_eikonal_row_numeric(model):
    """Optimizer.cpp:196-218: residual |g_fd| - 1 and its row over {self, 3 stencil neighbours}; a neighbour outside the band has
    its column dropped (rows[k] = -1) although the residual does depend on it"""
    sc, o = make(model)
    band = o.download_band(); vs = float(sc.voxel_size); h = 1e-2 * vs
    rng = np.random.default_rng(3)
    checked = 0
    for j in rng.choice(len(band), 60, replace=False):
        Jr, res, rows = o.probe_eikonal(int(j))
        assert rows[0] == j
        g = None
        for k in range(4):
            if rows[k] < 0:
                continue
            lin = int(band[rows[k]]); d0 = o.peek_dist(lin); rr = []
            for s in (+1, -1):
                o.poke_dist(lin, d0 + s * h); rr.append(o.probe_eikonal(int(j))[1])
            o.poke_dist(lin, d0)
            num = (rr[0] - rr[1]) / (2 * h)
            assert abs(num - Jr[k]) <= 5e-3 * max(np.abs(Jr).max(), 1.0 / vs * 1e-2) + 1e-3 / vs, (j, k, num, Jr)
            checked += 1
        # the four entries sum to zero when all neighbours are in the band (a constant offset of d leaves |g| unchanged)
        if min(rows) >= 0:
            assert abs(float(Jr.sum())) <= 1e-4 * np.abs(Jr).max()
    assert checked >= 150


def test_laplacian_diagonal_and_quirk_b3():
    """Optimizer.cpp:368-393,540-590: residual (sum of 6 neighbours - 6 d) / vs^2; the reference emits only the diagonal -6 / vs^2
    (B3: the neighbour triplets are built but never pushed) although d res / d d_neighbour = 1 / vs^2"""
    sc, o = make("SH1", reg_weight_l=2.0)
    band = o.download_band(); vs = float(sc.voxel_size); h = 1e-2 * vs
    dim = sc.dim
    for j in np.random.default_rng(4).choice(len(band), 20, replace=False):
        lin = int(band[j]); res, Jd = o.probe_laplacian(int(j))
        assert abs(Jd - (-6.0 / vs ** 2)) <= 1e-5 * 6.0 / vs ** 2
        d0 = o.peek_dist(lin); rr = []
        for s in (+1, -1):
            o.poke_dist(lin, d0 + s * h); rr.append(o.probe_laplacian(int(j))[0])
        o.poke_dist(lin, d0)
        assert abs((rr[0] - rr[1]) / (2 * h) - Jd) <= 2e-3 * abs(Jd)
        ln = lin + 1                                   # the +x neighbour (inside the grid for band voxels of the sphere scene)
        d1 = o.peek_dist(ln); rr = []
        for s in (+1, -1):
            o.poke_dist(ln, d1 + s * h); rr.append(o.probe_laplacian(int(j))[0])
        o.poke_dist(ln, d1)
        assert abs((rr[0] - rr[1]) / (2 * h) - 1.0 / vs ** 2) <= 2e-3 * 6.0 / vs ** 2


@pytest.mark.parametrize("model,quirks", [("SH1", 1), ("LED", 1), ("LED", 0)])
def test_assembled_distance_system_is_jtwj(model, quirks):
    """PsOptimizer.cpp:128-154 / LedOptimizer.cpp:198-228: H = Jd^T W Jd + reg_n Jr^T Jr + reg_l Jl^T Jl, b likewise, rebuilt densely from the
    per-observation rows (orc_probe_dist_jacobian + orc_probe_residual), the Eikonal rows and the Laplacian diagonal, against what the
    oracle's assembler (per-voxel 4x4 blocks scattered through the stencil columns) produces"""
    sc = synth.make_scene(N=20, F=4, W=96, H=72, model=model, noise=False)
    st = capi.default_settings(synth.MODELS[model], reg_weight_l=2.0, ref_quirks=quirks)
    o = oracle.Oracle(sc, sc.K, st); o.load_scene(sc)
    o.init_albedo(); o.normalize_weights()
    i = o.info(); S, F = i.n_band, i.n_frames
    reg_n, reg_l = float(i.reg_weight_n), float(i.reg_weight_l)
    assert 200 < S < 4000 and reg_n > 0 and reg_l > 0
    H = np.zeros((S, S)); b = np.zeros(S)
    for j in range(S):
        for f in range(F):
            ok, r, w = o.probe_residual(j, f)
            okj, J, rows = o.probe_dist_jacobian(j, f)
            if not (ok and okj):
                continue
            for ch in range(3):
                for a in range(4):
                    if rows[a] < 0:
                        continue
                    b[rows[a]] += float(J[a, ch]) * float(w[ch]) * float(r[ch])
                    for q in range(4):
                        if rows[q] >= 0:
                            H[rows[a], rows[q]] += float(J[a, ch]) * float(w[ch]) * float(J[q, ch])
        Jr, res, rows = o.probe_eikonal(j)
        for a in range(4):
            if rows[a] < 0:
                continue
            b[rows[a]] += reg_n * float(Jr[a]) * res
            for q in range(4):
                if rows[q] >= 0:
                    H[rows[a], rows[q]] += reg_n * float(Jr[a]) * float(Jr[q])
        lres, Jl = o.probe_laplacian(j)
        H[j, j] += reg_l * Jl * Jl; b[j] += reg_l * Jl * lres
    assert np.abs(H - H.T).max() <= 1e-9 * np.abs(H).max()
    rng = np.random.default_rng(0)
    x = rng.standard_normal(S).astype(np.float32)
    diag, rhs, y = o.debug_dist_system(x)
    assert np.abs(diag - np.diag(H)).max() <= 2e-6 * np.abs(np.diag(H)).max()
    assert np.abs(rhs - b).max() <= 2e-6 * np.abs(b).max()
    assert np.abs(y - H @ x.astype(np.float64)).max() <= 5e-6 * np.abs(H @ x).max()
    # structure: every row couples only voxels that share a stencil, i.e. |index offset| <= 1 per axis
    dim = np.array(i.dim[:]); band = o.download_band().astype(np.int64)
    ii, jj = np.nonzero(H)
    ci = np.stack([band[ii] % dim[0], (band[ii] // dim[0]) % dim[1], band[ii] // (dim[0] * dim[1])])
    cj = np.stack([band[jj] % dim[0], (band[jj] // dim[0]) % dim[1], band[jj] // (dim[0] * dim[1])])
    assert np.abs(ci - cj).max() <= 1
