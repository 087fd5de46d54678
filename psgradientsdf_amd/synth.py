"""Deterministic synthetic RGB-D photometric-stereo scenes.

Test / benchmark input generator only: a bumpy sphere with a smooth albedo pattern, observed by
F cameras on an orbit, rendered with the same forward model the optimiser inverts
(reference: PsOptimizerJa.cpp:30-40 for SH, LedOptimizerJa.cpp:15-29 for LED), plus the
"analytic" fused voxel state the reference would obtain from VolumetricGradSdf::update
(VolumetricGradSdf.cpp:51-138): dist, outward gradient, weight, colour, per-frame visibility.

Everything is numpy (no GPU, no oracle): the same arrays feed the HIP engine and the CPU oracle.
"""
from __future__ import annotations

import numpy as np

MODELS = {"SH1": 0, "SH2": 1, "LED": 2}


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


class Scene:
    """Plain container; see make_scene."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def _shape_f(x, c, R0, A):
    """f(x)=|x-c|-(R0+A sin(8 theta) sin(8 phi)) and its gradient (float64)."""
    p = x - c
    r = np.linalg.norm(p, axis=-1)
    r = np.maximum(r, 1e-12)
    ct = np.clip(p[..., 2] / r, -1.0, 1.0)
    theta = np.arccos(ct)
    phi = np.arctan2(p[..., 1], p[..., 0])
    s8t, c8t = np.sin(8 * theta), np.cos(8 * theta)
    s8p, c8p = np.sin(8 * phi), np.cos(8 * phi)
    f = r - (R0 + A * s8t * s8p)
    # gradient: e_r - A*(8 c8t s8p e_theta/r + 8 s8t c8p e_phi/(r sin theta))
    st = np.sqrt(np.maximum(1.0 - ct * ct, 1e-12))
    er = p / r[..., None]
    rho = np.maximum(np.hypot(p[..., 0], p[..., 1]), 1e-12)
    eth = np.stack([ct * p[..., 0] / rho, ct * p[..., 1] / rho, -st], -1)
    eph = np.stack([-p[..., 1] / rho, p[..., 0] / rho, np.zeros_like(rho)], -1)
    g = er - A * (8 * c8t * s8p / r)[..., None] * eth - A * (8 * s8t * c8p / (r * st))[..., None] * eph
    return f, g


def _albedo(x, c, L):
    """smooth 3-colour pattern in [0.2, 0.9]"""
    q = (x - c) / L
    k = 2 * np.pi * 1.5
    a = np.stack([
        np.sin(k * q[..., 0] + 0.3) * np.cos(k * q[..., 1]),
        np.sin(k * q[..., 1] + 1.1) * np.cos(k * q[..., 2]),
        np.sin(k * q[..., 2] + 2.3) * np.cos(k * q[..., 0]),
    ], -1)
    return 0.55 + 0.35 * a


def _sh(n, order):
    one = np.ones_like(n[..., 0])
    b = [one, n[..., 0], n[..., 1], n[..., 2]]
    if order == 2:
        b += [n[..., 0] * n[..., 1], n[..., 0] * n[..., 2], n[..., 1] * n[..., 2],
              n[..., 0] ** 2 - n[..., 1] ** 2, n[..., 0] ** 2 - n[..., 2] ** 2]
    return np.stack(b, -1)


def _run_parallel(fn, jobs, workers):
    """numpy releases the GIL inside its kernels: a thread pool over frames / z-chunks scales with the host cores"""
    jobs = list(jobs)
    if workers <= 1 or len(jobs) <= 1:
        for j in jobs:
            fn(j)
        return
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=workers) as ex:
        list(ex.map(fn, jobs))


def make_scene(N=64, F=8, W=160, H=120, model="SH1", seed=1234, extent=0.512, noise=True,
               perturb=True, z_mult=1, dtype=np.float32, bump=1.0, arc=360.0, u8=False, workers=None, zigzag=True, z_planes=None, reuse=None):
    """Build a scene.  N: grid edge (z edge = N*z_mult, z_mult bumpy spheres stacked along z for the
    weak-scaling bench), F keyframes of W x H pixels.

    u8: quantise the rendered keyframes to 8 bits the way a camera / PNG does; the scene then carries `images_u8` (uint8) with
    `image_scale` = 1/255 next to `images` = images_u8 * image_scale (float32, exactly what the reference's loader would hold), and
    everything derived from the images uses the quantised values.

    z_planes = (zlo, zhi): the per-voxel arrays hold only those z-planes of the volume (a rank of a multi-rank run that never touches the whole
    volume: capi.load_scene_slab); the planes are bit-identical to the same planes of the whole scene.  reuse: a scene of the same parameters
    whose keyframes are taken over instead of being rendered again.

    Intrinsics scale with the image so the object always fills the same fraction of the frame:
    fx = fy = 525 * W/640 (TUM-like 640x480 -> 525)."""
    if workers is None:
        import os
        workers = max(1, min(32, (os.cpu_count() or 1)))
    rng_l = np.random.default_rng(seed)
    vs = extent / N
    dim = np.array([N, N, N * z_mult], np.int32)
    shift = np.array([0.013, -0.021, 0.007], np.float64)
    T = 5.0 * vs
    R0 = 0.34 * N * vs
    A = 0.01 * N * vs * bump      # bump > 1: stronger relief (pose tracking needs shape that is not rotation-symmetric)
    centres = [shift + np.array([0, 0, (s - 0.5 * (z_mult - 1)) * N * vs]) for s in range(z_mult)]
    fx = fy = 525.0 * W / 640.0
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    K = np.array([fx, 0, cx, 0, fy, cy, 0, 0, 1], np.float32)
    order = 2 if model == "SH2" else 1
    nb = 9 if model == "SH2" else 4

    # ---- cameras: orbit of radius 1.5*N*vs around the z axis of the (whole) object, +-15 deg zig-zag
    orbit = 1.5 * N * vs
    poses = np.zeros((F, 4, 4), np.float64)
    zspan = 0.5 * (z_mult - 1) * N * vs
    for f in range(F):
        az = np.deg2rad(arc) * f / F      # arc < 360: a short video-like sweep (frame-to-model tracking tests)
        el = np.deg2rad(15.0) * (1 if (f % 2 == 0 or not zigzag) else -1)      # zigzag=False: a smooth camera path (frame-to-model tracking over a stream)
        zc = 0.0 if z_mult == 1 else -zspan + 2 * zspan * ((f * 7) % F) / max(F - 1, 1)
        target = shift + np.array([0, 0, zc])
        pos = target + orbit * np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
        zax = _unit(target - pos)
        up = np.array([0.0, 0.0, 1.0])
        xax = _unit(np.cross(zax, up))
        yax = np.cross(zax, xax)
        poses[f, :3, 0], poses[f, :3, 1], poses[f, :3, 2], poses[f, :3, 3] = xax, yax, zax, pos
        poses[f, 3, 3] = 1.0

    # ---- lights
    light = None
    if model == "LED":
        light = np.array([1.2, 1.0, 0.8]) * orbit ** 2
    else:
        light = np.zeros((F, nb))
        for f in range(F):
            d = _unit(-poses[f, :3, 2] + 0.5 * rng_l.standard_normal(3))
            light[f, 0] = 0.3
            light[f, 1:4] = 0.6 * d
            if order == 2:
                light[f, 4:] = rng_l.uniform(-0.1, 0.1, 5)

    def shade(x, n, f):
        """rendered intensity of surface point x with outward unit normal n in frame f"""
        rho = _albedo(x, shift, extent)
        if model == "LED":
            R, t = poses[f, :3, :3], poses[f, :3, 3]
            v = x - t
            irr = -(n * v).sum(-1) / np.linalg.norm(v, axis=-1) ** 3
            return rho * light[None, :] * irr[..., None]
        irr = (_sh(n, order) * light[f]).sum(-1)
        return rho * irr[..., None]

    def fmin(x):
        """distance-like value to the union of the stacked blobs + gradient of the active one"""
        best_f, best_g = None, None
        for c in centres:
            f_, g_ = _shape_f(x, c, R0, A)
            if best_f is None:
                best_f, best_g = f_, g_
            else:
                m = f_ < best_f
                best_f = np.where(m, f_, best_f)
                best_g = np.where(m[..., None], g_, best_g)
        return best_f, best_g

    # ---- images by sphere tracing
    images = np.zeros((F, H, W, 3), np.float64)
    depth = np.zeros((F, H, W), np.float32)              # metres along the optical axis, 0 = no measurement
    normals_cam = np.zeros((F, 3, H, W), np.float32)     # camera-frame, inward-pointing (VolumetricGradSdf.cpp:123)
    uu, vv = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    dirs_c = _unit(np.stack([(uu - cx) / fx, (vv - cy) / fy, np.ones_like(uu)], -1))
    def render(f):
        R, t = poses[f, :3, :3], poses[f, :3, 3]
        d = dirs_c @ R.T
        # cull by the bounding spheres
        hit_any = np.zeros((H, W), bool)
        t0 = np.full((H, W), np.inf)
        for c in centres:
            oc = t - c
            bq = (d * oc).sum(-1)
            cq = (oc * oc).sum() - (R0 + 1.2 * A) ** 2
            disc = bq * bq - cq
            ok = disc > 0
            tt = -bq - np.sqrt(np.where(ok, disc, 0))
            ok &= tt > 0
            t0 = np.where(ok & (tt < t0), tt, t0)
            hit_any |= ok
        idx = np.nonzero(hit_any)
        if len(idx[0]) == 0:
            return
        dd = d[idx]
        s = t0[idx].copy()
        smax = t0[idx] + 4 * (R0 + 2 * A)
        act = np.arange(len(s))                           # rays still marching: converged / escaped ones drop out
        for _ in range(60):
            x = t + dd[act] * s[act, None]
            fv, gv = fmin(x)
            step = fv / np.linalg.norm(gv, axis=-1)
            s[act] += 0.9 * step
            act = act[(np.abs(step) > 1e-14 * extent) & (s[act] < smax[act])]
            if len(act) == 0:
                break
        x = t + dd * s[:, None]
        fv, gv = fmin(x)
        good = np.abs(fv) < 1e-6 * extent + 1e-9
        n = _unit(gv)
        col = shade(x, n, f)
        col = np.where(good[:, None], col, 0.0)
        images[f][idx] = col
        zc = (s * dirs_c[idx][:, 2])
        depth[f][idx] = np.where(good, zc, 0.0).astype(np.float32)
        ncam = -(n @ R)                                   # R^T n_world, flipped to point into the surface
        for a in range(3):
            normals_cam[f, a][idx] = np.where(good, ncam[:, a], 0.0).astype(np.float32)
    if reuse is None:
        _run_parallel(render, range(F), workers)
        if noise:
            images = images + 0.005 * np.random.default_rng(1).standard_normal(images.shape)
        images = np.clip(images, 0.0, 1.0)
    else:
        images, depth, normals_cam = reuse.images.astype(np.float64), reuse.depth, reuse.normals_cam
    images_u8, image_scale = None, None
    if u8:
        images_u8 = np.rint(images * 255.0).astype(np.uint8)
        image_scale = np.float32(1.0) / np.float32(255.0)
        images = (images_u8.astype(np.float32) * image_scale).astype(np.float64)      # float32 product, as convertTo(CV_32FC3, 1/255) stores it

    # ---- analytic voxel state
    ii = np.arange(N, dtype=np.float64)
    kk = np.arange(N * z_mult, dtype=np.float64)
    origin = shift - 0.5 * vs * dim.astype(np.float64)
    plane = N * N
    chunk = max(1, (1 << 20) // plane)      # z-planes per chunk (bounds memory)
    nz = N * z_mult
    zlo, zhi = (0, nz) if z_planes is None else (int(z_planes[0]), int(z_planes[1]))
    ka, kb = (zlo // chunk) * chunk, min(nz, -(-zhi // chunk) * chunk)      # the chunk-aligned planes that are computed
    nvox = (kb - ka) * plane
    dist = np.empty(nvox, np.float32)
    grad = np.zeros((3, nvox), np.float32)
    weight = np.zeros(nvox, np.float32)
    rgb = np.zeros((3, nvox), np.float32)
    albedo_gt = np.zeros((3, nvox), np.float32)   # true albedo at the surface point of every near-surface voxel
    wpv = (F + 63) // 64
    vis = np.zeros((nvox, wpv), np.uint64)
    rng_d = np.random.default_rng(3)

    def voxel_chunk(job):
        k0, k1, pert_noise = job
        Z, Y, X = np.meshgrid(kk[k0:k1], ii, ii, indexing="ij")
        x = origin + vs * np.stack([X, Y, Z], -1).reshape(-1, 3)
        # far from every blob the clipped distance is +-T whatever the bumps do: skip the trigonometry there
        # (|f| >= ||x-c|-R0| - A, and d = f / |grad f| with the bound gmax below)
        rr = np.stack([np.linalg.norm(x - c, axis=-1) for c in centres], 0)
        sin_t = np.stack([np.hypot(x[:, 0] - c[0], x[:, 1] - c[1]) for c in centres], 0) / np.maximum(rr, 1e-12)
        gmax = 1.0 + 8.0 * A / np.maximum(rr, 1e-12) * (1.0 + 1.0 / np.maximum(sin_t, 1e-6))     # >= |grad f| (the phi term blows up at the poles)
        rc = rr - R0
        cand = np.nonzero(((np.abs(rc) - A) <= 1.05 * T * gmax).any(0))[0]
        d = np.where(rc.min(0) < 0, -2.0 * T, 2.0 * T)
        nrm = np.zeros((len(x), 3))
        if len(cand):
            fv, gv = fmin(x[cand])
            gn = np.linalg.norm(gv, axis=-1)
            d[cand] = fv / gn
            nrm[cand] = gv / gn[:, None]
        sl = slice((k0 - ka) * plane, (k1 - ka) * plane)
        near = np.abs(d) < T
        dn = np.clip(d, -T, T)
        if perturb:
            pert = 0.3 * vs * np.sin(40.0 * x[:, 0] / extent * 2 + 1.0) * np.cos(34.0 * x[:, 1] / extent * 2) * np.sin(28.0 * x[:, 2] / extent * 2 + 0.5)
            pert = pert + 0.03 * vs * pert_noise
            dn = np.where(near, np.clip(d + pert, -T, T), dn)
        dist[sl] = dn
        nidx = np.nonzero(near)[0]
        if len(nidx) == 0:
            return
        xs = x[nidx] - d[nidx, None] * nrm[nidx]
        cnt = np.zeros(len(nidx))
        colsum = np.zeros((len(nidx), 3))
        vbits = np.zeros((len(nidx), wpv), np.uint64)
        for f in range(F):
            R, t = poses[f, :3, :3], poses[f, :3, 3]
            view = _unit(t - xs)
            cosang = (view * nrm[nidx]).sum(-1)
            pc = (xs - t) @ R
            with np.errstate(divide="ignore", invalid="ignore"):
                m = fx * pc[:, 0] / pc[:, 2] + cx
                n_ = fy * pc[:, 1] / pc[:, 2] + cy
            ok = (cosang > 0.25) & (pc[:, 2] > 0) & (m >= 1) & (m < W - 2) & (n_ >= 1) & (n_ < H - 2)
            cnt += ok
            vbits[ok, f >> 6] |= np.uint64(1) << np.uint64(f & 63)
            if ok.any():
                colsum[ok] += np.clip(shade(xs[ok], nrm[nidx][ok], f), 0, 1)
        g = sl.start + nidx
        weight[g] = cnt
        grad[:, g] = (nrm[nidx] * np.maximum(cnt, 1)[:, None]).T
        rgb[:, g] = (colsum / np.maximum(cnt, 1)[:, None]).T
        albedo_gt[:, g] = _albedo(xs, shift, extent).T
        vis[g] = vbits

    # the per-voxel noise comes from ONE generator consumed in chunk order (scenes are bit-reproducible whatever the worker count):
    # draw a batch of chunks' worth, process the batch in parallel, repeat
    starts = list(range(0, N * z_mult, chunk))
    for b0 in range(0, len(starts), workers):
        jobs = []
        for k0 in starts[b0:b0 + workers]:
            k1 = min(N * z_mult, k0 + chunk)
            pn = rng_d.standard_normal((k1 - k0) * plane) if perturb else None      # (drawn for every chunk: the planes of a partial scene are those of the whole one)
            if k0 >= ka and k1 <= kb:
                jobs.append((k0, k1, pn))
        _run_parallel(voxel_chunk, jobs, workers)
    if (ka, kb) != (zlo, zhi):
        cut = slice((zlo - ka) * plane, (zhi - ka) * plane)
        dist, weight, vis = dist[cut].copy(), weight[cut].copy(), vis[cut].copy()
        grad, rgb, albedo_gt = grad[:, cut].copy(), rgb[:, cut].copy(), albedo_gt[:, cut].copy()

    poses_used = poses.copy()
    if perturb:
        rng_p = np.random.default_rng(4)
        for f in range(F):
            xi_t = 0.002 * rng_p.standard_normal(3)
            w = np.deg2rad(0.2) * rng_p.standard_normal(3)
            th = np.linalg.norm(w)
            Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
            Rw = np.eye(3) + np.sin(th) / th * Kx + (1 - np.cos(th)) / th ** 2 * Kx @ Kx
            poses_used[f, :3, :3] = poses[f, :3, :3] @ Rw
            poses_used[f, :3, 3] = poses[f, :3, 3] + xi_t

    return Scene(
        N=N, F=F, W=W, H=H, model=model, model_id=MODELS[model], dim=dim, voxel_size=np.float32(vs),
        shift=shift.astype(np.float32), truncation=np.float32(T), K=K,
        dist=dist, grad=np.ascontiguousarray(grad), weight=weight, rgb=np.ascontiguousarray(rgb),
        vis=np.ascontiguousarray(vis), vis_words=wpv, albedo_gt=albedo_gt,
        images=np.ascontiguousarray(images.astype(np.float32)), images_u8=images_u8, image_scale=image_scale,
        poses=np.ascontiguousarray(poses_used.reshape(F, 16).astype(np.float32)),
        poses_gt=poses.reshape(F, 16).astype(np.float32),
        light_gt=np.asarray(light, np.float32), frame_idx=np.arange(F, dtype=np.int32),
        depth=depth, normals_cam=normals_cam, R0=R0, A=A, extent=extent, z_planes=(zlo, zhi),
    )


def tile_scene(sc, n):
    """Weak-scaling scene: n copies of `sc` stacked along z (grid N x N x n*N), each copy observed by its own copy of
    the F keyframes (poses translated with it), so per-slab work is exactly that of the single scene.  O(size) numpy
    copies only -- no re-rendering."""
    if n == 1:
        return sc
    N, F = int(sc.dim[0]), sc.F
    nz = int(sc.dim[2])
    plane = int(sc.dim[0]) * int(sc.dim[1])
    nvox = plane * nz

    def tz(a):          # [..., nvox] -> [..., n*nvox], copies consecutive along z (slowest index)
        return np.ascontiguousarray(np.concatenate([a] * n, axis=-1))

    wpv = (F * n + 63) // 64
    vis = np.zeros((n * nvox, wpv), np.uint64)
    near = np.nonzero((sc.vis != 0).any(axis=1))[0]
    bits = [((sc.vis[near, f >> 6] >> np.uint64(f & 63)) & np.uint64(1)).astype(bool) for f in range(F)]
    for c in range(n):
        for f in range(F):
            g = c * F + f
            rows = near[bits[f]] + c * nvox
            vis[rows, g >> 6] |= np.uint64(1) << np.uint64(g & 63)
    poses = np.tile(sc.poses.reshape(1, F, 16), (n, 1, 1)).astype(np.float32)
    poses_gt = np.tile(sc.poses_gt.reshape(1, F, 16), (n, 1, 1)).astype(np.float32)
    vs = float(sc.voxel_size)
    for c in range(n):
        dz = np.float32(vs * nz * (c - 0.5 * (n - 1)))
        poses[c, :, 11] += dz
        poses_gt[c, :, 11] += dz
    light = sc.light_gt if sc.model == "LED" else np.tile(sc.light_gt, (n, 1))
    d = dict(sc.__dict__)
    d.update(F=F * n, dim=np.array([sc.dim[0], sc.dim[1], nz * n], np.int32), dist=tz(sc.dist), grad=tz(sc.grad), weight=tz(sc.weight),
             rgb=tz(sc.rgb), albedo_gt=tz(sc.albedo_gt), vis=vis, vis_words=wpv,
             images=np.ascontiguousarray(np.tile(sc.images, (n, 1, 1, 1))),
             images_u8=None if getattr(sc, 'images_u8', None) is None else np.ascontiguousarray(np.tile(sc.images_u8, (n, 1, 1, 1))), poses=poses.reshape(n * F, 16), poses_gt=poses_gt.reshape(n * F, 16),
             light_gt=np.asarray(light, np.float32), frame_idx=np.arange(F * n, dtype=np.int32))
    return Scene(**d)


def tile_scene_lazy(sc, n):
    """tile_scene without ever materialising the n-fold volume: returns (scene, planes) where `scene` carries the grid, the n*F keyframes, poses
    and lights of the tiled scene but NO per-voxel arrays, and `planes(zlo, zhi)` builds the per-voxel arrays of the z-planes [zlo, zhi) of the
    tiled volume on demand (capi.load_scene_slab: a rank only ever holds its own slab).  Same bits as tile_scene(sc, n) restricted to the planes."""
    if n == 1:
        return sc, None
    F = sc.F
    nz = int(sc.dim[2])
    plane = int(sc.dim[0]) * int(sc.dim[1])
    wpv = (F * n + 63) // 64
    poses = np.tile(sc.poses.reshape(1, F, 16), (n, 1, 1)).astype(np.float32)
    poses_gt = np.tile(sc.poses_gt.reshape(1, F, 16), (n, 1, 1)).astype(np.float32)
    vs = float(sc.voxel_size)
    for c in range(n):
        dz = np.float32(vs * nz * (c - 0.5 * (n - 1)))
        poses[c, :, 11] += dz
        poses_gt[c, :, 11] += dz
    light = sc.light_gt if sc.model == "LED" else np.tile(sc.light_gt, (n, 1))
    d = {k: v for k, v in sc.__dict__.items() if k not in ("dist", "grad", "weight", "rgb", "albedo_gt", "vis")}
    d.update(F=F * n, dim=np.array([sc.dim[0], sc.dim[1], nz * n], np.int32), vis_words=wpv,
             images=np.ascontiguousarray(np.tile(sc.images, (n, 1, 1, 1))),
             images_u8=None if getattr(sc, 'images_u8', None) is None else np.ascontiguousarray(np.tile(sc.images_u8, (n, 1, 1, 1))),
             poses=poses.reshape(n * F, 16), poses_gt=poses_gt.reshape(n * F, 16),
             light_gt=np.asarray(light, np.float32), frame_idx=np.arange(F * n, dtype=np.int32))
    tiled = Scene(**d)

    def planes(zlo, zhi):
        dist, grad, weight, rgb, vis = [], [], [], [], []
        k = zlo
        while k < zhi:
            c, kz = divmod(k, nz)
            k1 = min(zhi, (c + 1) * nz)                 # the part of [zlo, zhi) that lies in copy c
            sl = slice(kz * plane, (kz + k1 - k) * plane)
            dist.append(sc.dist[sl]); grad.append(sc.grad[:, sl]); weight.append(sc.weight[sl]); rgb.append(sc.rgb[:, sl])
            v = np.zeros(((k1 - k) * plane, wpv), np.uint64)
            src = sc.vis[sl]
            near = np.nonzero((src != 0).any(axis=1))[0]
            for f in range(F):
                g = c * F + f
                rows = near[((src[near, f >> 6] >> np.uint64(f & 63)) & np.uint64(1)).astype(bool)]
                v[rows, g >> 6] |= np.uint64(1) << np.uint64(g & 63)
            vis.append(v)
            k = k1
        return dict(dist=np.ascontiguousarray(np.concatenate(dist)), grad=np.ascontiguousarray(np.concatenate(grad, axis=1)),
                    weight=np.ascontiguousarray(np.concatenate(weight)), rgb=np.ascontiguousarray(np.concatenate(rgb, axis=1)),
                    vis=np.ascontiguousarray(np.concatenate(vis)))
    return tiled, planes
