"""ctypes binding of the C ABI declared in include/psgsdf.h.

This is host-side plumbing for tests and bench.py (the reference is C++; the C++ mirror of its
PsOptimizer/LedOptimizer interface lives in psgradientsdf_amd/host/).  `Api` is generic over
(shared library, symbol prefix) so that a test harness can bind another implementation of the
same C ABI; the product only ever instantiates it through `load_engine()`, which binds
`libpsgsdf.so` and raises if the HIP library is missing — there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

ALBEDO, LIGHT, DIST, POSE, ALL = 1, 2, 4, 8, 15
SH1, SH2, LED = 0, 1, 2
L2, CAUCHY, HUBER, TUKEY, TRUNC_L2 = 0, 1, 2, 3, 4

_HERE = os.path.dirname(os.path.abspath(__file__))
ENGINE_LIB = os.environ.get("PSGSDF_ENGINE_LIB") or os.path.join(_HERE, "csrc", "libpsgsdf.so")      # (the override is for A/B runs of two BUILDS on one box: tools/)
# the development build (-DPSGSDF_DEV): the same engine plus the fault-injection / ablation knobs the product library does not contain; tests and tools only
ENGINE_LIB_DEV = os.path.join(_HERE, "csrc", "libpsgsdf_dev.so")


class GridDesc(C.Structure):
    _fields_ = [("dim", C.c_int32 * 3), ("voxel_size", C.c_float), ("shift", C.c_float * 3), ("truncation", C.c_float)]


class Settings(C.Structure):
    _fields_ = [("model", C.c_int32), ("loss", C.c_int32), ("lambda_", C.c_float), ("damping", C.c_float),
                ("reg_weight_rho", C.c_float), ("reg_weight_n", C.c_float), ("reg_weight_l", C.c_float),
                ("max_it", C.c_int32), ("conv_threshold", C.c_float), ("upsample", C.c_int32),
                ("ref_quirks", C.c_int32), ("cg_max_it", C.c_int32)]


class StepStats(C.Structure):
    _fields_ = [("block", C.c_int32), ("cg_iters", C.c_int32), ("cg_converged", C.c_int32), ("applied", C.c_int32),
                ("e_in", C.c_double), ("cg_error", C.c_double), ("n_accepted", C.c_int64), ("n_obs", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class IterStats(C.Structure):
    _fields_ = [("e_after", C.c_double * 4), ("e_n", C.c_double), ("e_l", C.c_double), ("e_total", C.c_double),
                ("rel_diff", C.c_double), ("reg_weight_n", C.c_float), ("reg_weight_l", C.c_float),
                ("cg_iters", C.c_int32), ("converged", C.c_int32), ("diverged", C.c_int32), ("upsampled", C.c_int32),
                ("e_r", C.c_double), ("e_n_in", C.c_double), ("e_l_in", C.c_double)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["e_after"] = list(self.e_after)
        return d

    def __getitem__(self, k):      # (a record reads like its dict: the passive observer hands out struct copies, not dicts -- it runs between two launches)
        v = getattr(self, k)
        return list(v) if k == "e_after" else v


class Info(C.Structure):
    _fields_ = [("dim", C.c_int32 * 3), ("voxel_size", C.c_float), ("origin", C.c_float * 3), ("n_frames", C.c_int32),
                ("n_band", C.c_int32), ("light_stride", C.c_int32), ("vis_words", C.c_int32),
                ("reg_weight_n", C.c_float), ("reg_weight_l", C.c_float)]


def default_settings(model=SH1, **kw):
    """config_skorates.json values (cauchy, lambda 0.2, damping 1, reg_n 10, reg_l 0, conv 5e-3)."""
    s = Settings(model=model, loss=CAUCHY, lambda_=0.2, damping=1.0, reg_weight_rho=0.0, reg_weight_n=10.0,
                 reg_weight_l=0.0, max_it=100, conv_threshold=5e-3, upsample=0, ref_quirks=1, cg_max_it=0)
    for k, v in kw.items():
        if k == "lambda":
            k = "lambda_"
        setattr(s, k, v)
    return s


def _fp(a, dtype=np.float32):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a, a.ctypes.data_as(C.c_void_p)


class PsgsdfError(RuntimeError):
    pass


class Api:
    """One context behind the C ABI of include/psgsdf.h.  (The binding is generic over (library, symbol prefix): the test oracle exports the same
    entry points under another prefix and subclasses this in oracle/oracle.py; the product only instantiates it through load_engine.)"""

    def __init__(self, lib: C.CDLL, prefix: str, grid: GridDesc, K, settings: Settings, device: int = 0):
        self._lib, self._p = lib, prefix
        self.ctx = C.c_void_p()
        Karr, Kp = _fp(K)
        self._grid, self._settings = grid, settings
        self._check(self._fn("create")(C.byref(grid), Kp, C.byref(settings), C.c_int(device), C.byref(self.ctx)), "create")

    # -- plumbing
    def _fn(self, name):
        f = getattr(self._lib, self._p + name)
        f.restype = C.c_int
        return f

    def _check(self, rc, what):
        if rc != 0:
            msg = ""
            try:
                g = getattr(self._lib, self._p + "last_error")
                g.restype = C.c_char_p
                msg = (g(self.ctx) or b"").decode()
            except Exception:
                pass
            raise PsgsdfError(f"{self._p}{what} failed rc={rc} {msg}")

    def close(self):
        if self.ctx:
            d = getattr(self._lib, self._p + "destroy")
            d.restype = None
            d(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- inputs
    def upload_volume(self, dist, grad, weight, rgb, vis, words):
        a = [_fp(dist), _fp(grad), _fp(weight), _fp(rgb), _fp(vis, np.uint64)]
        self._check(self._fn("upload_volume")(self.ctx, a[0][1], a[1][1], a[2][1], a[3][1], a[4][1], C.c_int(words)), "upload_volume")

    # -- slab-local upload (multi-rank): no rank touches the whole volume
    def slab_plane_count(self, dist_plane, vis_plane, words):
        a = [_fp(dist_plane), _fp(vis_plane, np.uint64)]; out = C.c_double()
        self._check(self._fn("slab_plane_count")(self.ctx, a[0][1], a[1][1], C.c_int(words), C.byref(out)), "slab_plane_count")
        return out.value

    def plan_slab(self, plane_counts):
        a = _fp(plane_counts, np.float64); z0, z1 = C.c_int(), C.c_int()
        self._check(self._fn("plan_slab")(self.ctx, a[1], C.byref(z0), C.byref(z1)), "plan_slab")
        return z0.value, z1.value

    def rebalance_slabs(self):
        """re-cut a slab-parallel fused volume into slabs of equal band-candidate count and move the planes (collective)"""
        self._check(self._fn("rebalance_slabs")(self.ctx), "rebalance_slabs")

    def upload_volume_slab(self, z0, z1, dist, grad, weight, rgb, vis, words):
        a = [_fp(dist), _fp(grad), _fp(weight), _fp(rgb), _fp(vis, np.uint64)]
        self._check(self._fn("upload_volume_slab")(self.ctx, C.c_int(z0), C.c_int(z1), a[0][1], a[1][1], a[2][1], a[3][1], a[4][1], C.c_int(words)), "upload_volume_slab")

    def load_scene_slab(self, sc, rank, n_ranks, planes=None, u8=None):
        """load_scene for a rank of a multi-rank run that only ever looks at ITS planes of the scene: `planes(zlo, zhi)` returns the dict of
        per-voxel arrays (dist, grad, weight, rgb, vis) of the z-planes [zlo, zhi) -- default: slices of the whole-volume arrays of `sc`
        (a scene generator or a file reader produces just those planes).  `sc` supplies the grid, the keyframes and the poses.
        The cut negotiation: rank r counts the band candidates of the r-th of n equal blocks of planes (any split is allowed)."""
        nx, ny, nz = (int(x) for x in sc.dim); plane = nx * ny
        if planes is None:
            def planes(zlo, zhi):
                sl = slice(zlo * plane, zhi * plane)
                return dict(dist=sc.dist[sl], grad=sc.grad[:, sl], weight=sc.weight[sl], rgb=sc.rgb[:, sl], vis=sc.vis[sl])
        cnt = np.zeros(nz)
        a, b = rank * nz // n_ranks, (rank + 1) * nz // n_ranks
        if b > a:
            p = planes(a, b)
            for k in range(a, b):
                sl = slice((k - a) * plane, (k - a + 1) * plane)
                cnt[k] = self.slab_plane_count(p["dist"][sl], p["vis"][sl], sc.vis_words)
        z0, z1 = self.plan_slab(cnt)
        zlo, zhi = max(0, z0 - 1), min(nz, z1 + 1)
        p = planes(zlo, zhi)
        self.upload_volume_slab(z0, z1, p["dist"], p["grad"], p["weight"], p["rgb"], p["vis"], sc.vis_words)
        has_u8 = getattr(sc, "images_u8", None) is not None
        if u8 is None:
            u8 = has_u8
        if u8:
            self.set_keyframes_u8(sc.frame_idx, sc.images_u8, sc.image_scale, sc.poses)
        else:
            self.set_keyframes(sc.frame_idx, sc.images, sc.poses)
        self.init()
        return z0, z1

    def set_keyframes(self, frame_idx, images, poses):
        F, H, W, _ = images.shape
        a = [_fp(frame_idx, np.int32), _fp(images), _fp(poses)]
        self._check(self._fn("set_keyframes")(self.ctx, C.c_int(F), a[0][1], a[1][1], C.c_int(W), C.c_int(H), a[2][1]), "set_keyframes")

    def set_keyframes_frames(self, frame_idx, image_list, poses):
        """one array [H][W][3] per keyframe (psgsdf_set_keyframes_frames: the host keeps one allocation per image)"""
        imgs = [np.ascontiguousarray(im, np.float32) for im in image_list]
        H, W, _ = imgs[0].shape
        ptrs = (C.c_void_p * len(imgs))(*[im.ctypes.data for im in imgs])
        a = [_fp(frame_idx, np.int32), _fp(poses)]
        self._check(self._fn("set_keyframes_frames")(self.ctx, C.c_int(len(imgs)), a[0][1], ptrs, C.c_int(W), C.c_int(H), a[1][1]), "set_keyframes_frames")

    def set_keyframes_u8(self, frame_idx, images_u8, scale, poses):
        """8-bit RGB keyframes [F][H][W][3] + the loader's conversion factor (colour = byte * scale)"""
        F, H, W = images_u8.shape[:3]
        img = np.ascontiguousarray(images_u8, dtype=np.uint8)
        a = [_fp(frame_idx, np.int32), _fp(poses)]
        self._check(self._fn("set_keyframes_u8")(self.ctx, C.c_int(F), a[0][1], img.ctypes.data_as(C.c_void_p), C.c_float(scale),
                                                    C.c_int(W), C.c_int(H), a[1][1]), "set_keyframes_u8")

    def load_scene(self, sc, u8=None):
        """u8: hand the keyframes over as 8-bit RGB when the scene has them (default: whenever it has)"""
        self.upload_volume(sc.dist, sc.grad, sc.weight, sc.rgb, sc.vis, sc.vis_words)
        has_u8 = getattr(sc, "images_u8", None) is not None
        if u8 is None:
            u8 = has_u8
        if u8:
            if not has_u8:
                raise ValueError("scene has no 8-bit images")
            self.set_keyframes_u8(sc.frame_idx, sc.images_u8, sc.image_scale, sc.poses)
        else:
            self.set_keyframes(sc.frame_idx, sc.images, sc.poses)
        self.init()

    def init(self):
        self._check(self._fn("init")(self.ctx), "init")

    # -- state producer (VolumetricGradSdf::init / update)
    def volume_init(self, max_frames):
        self._check(self._fn("volume_init")(self.ctx, C.c_int(max_frames)), "volume_init")

    def integrate_frame(self, rgb, depth, normals, pose, counter, z_min=0.05, z_max=10.0):
        H, W = depth.shape
        a = [_fp(rgb), _fp(depth), _fp(normals), _fp(pose)]
        self._check(self._fn("integrate_frame")(self.ctx, a[0][1], a[1][1], a[2][1], C.c_int(W), C.c_int(H), a[3][1], C.c_int(counter),
                                                 C.c_float(z_min), C.c_float(z_max)), "integrate_frame")

    def estimate_normals(self, depth):
        H, W = depth.shape
        a = _fp(depth)
        out = np.empty((3, H, W), np.float32)
        self._check(self._fn("estimate_normals")(self.ctx, a[1], C.c_int(W), C.c_int(H), out.ctypes.data_as(C.c_void_p)), "estimate_normals")
        return out

    def track(self, depth, pose, z_min=0.05, z_max=10.0, num_iterations=50, conv_threshold=1e-3, damping=1.0):
        H, W = depth.shape
        a = _fp(depth)
        P = np.ascontiguousarray(pose, np.float32).reshape(16).copy()
        it = C.c_int(); conv = C.c_int()
        self._check(self._fn("track")(self.ctx, a[1], C.c_int(W), C.c_int(H), P.ctypes.data_as(C.c_void_p), C.c_float(z_min), C.c_float(z_max),
                                      C.c_int(num_iterations), C.c_float(conv_threshold), C.c_float(damping), C.byref(it), C.byref(conv)), "track")
        return P.reshape(4, 4), it.value, bool(conv.value)

    def download_vis_seq(self, words):
        i = self.info()
        n = int(i.dim[0]) * int(i.dim[1]) * int(i.dim[2])
        out = np.zeros((n, words), np.uint64)      # (a slab fills the planes it owns)
        f = self._fn("download_vis_seq")
        rc = f(self.ctx, out.ctypes.data_as(C.c_void_p))
        if rc < 0:
            self._check(rc, "download_vis_seq")
        return out

    # -- hot path
    def init_albedo(self):
        self._check(self._fn("init_albedo")(self.ctx), "init_albedo")

    def energy(self):
        out = (C.c_double * 4)()
        self._check(self._fn("energy")(self.ctx, out), "energy")
        return list(out)

    def normalize_weights(self):
        e = C.c_double()
        self._check(self._fn("normalize_weights")(self.ctx, C.byref(e)), "normalize_weights")
        return e.value

    def step(self, block):
        st = StepStats()
        self._check(self._fn("step")(self.ctx, C.c_int(block), C.byref(st)), "step")
        return st.as_dict()

    def iterate(self, flags, n):
        arr = (IterStats * n)()
        self._check(self._fn("iterate")(self.ctx, C.c_int(flags), C.c_int(n), arr), "iterate")
        return [a.as_dict() for a in arr]

    def optimize(self, flags, cap=256, on_iter=None):
        """on_iter(iterations_done, record_dict) -> truthy aborts the loop (psgsdf_iter_cb: the host's hook for the reference's periodic dumps)"""
        arr = (IterStats * cap)()
        n, res = C.c_int(), C.c_int()
        cb = None
        if on_iter is not None:
            CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(IterStats))
            cb = CB(lambda user, done, rec: 1 if on_iter(done, rec.contents.as_dict()) else 0)
        self._check(self._fn("optimize")(self.ctx, C.c_int(flags), arr, C.c_int(cap), C.byref(n), C.byref(res), cb, None), "optimize")
        return [arr[i].as_dict() for i in range(min(n.value, cap))], bool(res.value)

    def set_on_iter_period(self, period):
        self._check(self._fn("set_on_iter_period")(self.ctx, C.c_int(period)), "set_on_iter_period")

    def set_record_observer(self, fn):
        """fn(iterations_done, record) -> truthy ends the loop (record: an IterStats copy, rec["e_total"] / rec.as_dict()); passive (psgsdf_set_record_observer): must not call into the engine"""
        if fn is None:
            self._observer = None
            self._check(self._fn("set_record_observer")(self.ctx, None, None), "set_record_observer")
            return
        CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(IterStats))
        self._observer = CB(lambda user, done, rec: 1 if fn(done, IterStats.from_buffer_copy(rec.contents)) else 0)      # (a 120-byte copy, fields read on demand: the callback sits on the loop's critical path)
        self._check(self._fn("set_record_observer")(self.ctx, self._observer, None), "set_record_observer")

    def upsample2x(self):
        self._check(self._fn("upsample2x")(self.ctx), "upsample2x")

    # -- outputs
    def info(self):
        i = Info()
        self._check(self._fn("get_info")(self.ctx, C.byref(i)), "get_info")
        return i

    def download_volume(self, want_vis=False):
        i = self.info()
        n = int(i.dim[0]) * int(i.dim[1]) * int(i.dim[2])
        dist = np.full(n, np.nan, np.float32); grad = np.full((3, n), np.nan, np.float32)      # (a slab fills its own planes only)
        weight = np.full(n, np.nan, np.float32); rgb = np.full((3, n), np.nan, np.float32)
        vis = np.empty((n, i.vis_words), np.uint64) if want_vis else None
        self._check(self._fn("download_volume")(self.ctx, dist.ctypes.data_as(C.c_void_p), grad.ctypes.data_as(C.c_void_p),
                                                 weight.ctypes.data_as(C.c_void_p), rgb.ctypes.data_as(C.c_void_p),
                                                 vis.ctypes.data_as(C.c_void_p) if want_vis else None), "download_volume")
        return dict(dist=dist, grad=grad, weight=weight, rgb=rgb, vis=vis)

    # -- the writers' geometry, extracted on the device (valid until the next extraction: copied here)
    def extract_mesh(self):
        """(xyz [n_vertices, 3] float32 grid-local, rgb [n_vertices, 3] uint8): three consecutive vertices are one face (psgsdf_extract_mesh)"""
        xyz = C.POINTER(C.c_float)(); rgb = C.POINTER(C.c_uint8)(); n = C.c_int64()
        self._check(self._fn("extract_mesh")(self.ctx, C.byref(xyz), C.byref(rgb), C.byref(n)), "extract_mesh")
        if n.value == 0:
            return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint8)
        return np.ctypeslib.as_array(xyz, shape=(n.value, 3)).copy(), np.ctypeslib.as_array(rgb, shape=(n.value, 3)).copy()

    def extract_pointcloud(self, which=0):
        """(xyz_nxyz [n, 6] float32, rgb [n, 3] int32); which = 0: the band voxels, 1: every fused voxel (psgsdf_extract_pointcloud)"""
        pn = C.POINTER(C.c_float)(); col = C.POINTER(C.c_int32)(); n = C.c_int64()
        self._check(self._fn("extract_pointcloud")(self.ctx, C.c_int(which), C.byref(pn), C.byref(col), C.byref(n)), "extract_pointcloud")
        if n.value == 0:
            return np.zeros((0, 6), np.float32), np.zeros((0, 3), np.int32)
        return np.ctypeslib.as_array(pn, shape=(n.value, 6)).copy(), np.ctypeslib.as_array(col, shape=(n.value, 3)).copy()

    def extract_sdf(self):
        """(lo [3], dim [3], -dist block [dim2, dim1, dim0]) of the crop box |d| <= sqrt(3) vs (psgsdf_extract_sdf)"""
        lo = (C.c_int32 * 3)(); dim = (C.c_int32 * 3)(); v = C.POINTER(C.c_float)()
        self._check(self._fn("extract_sdf")(self.ctx, lo, dim, C.byref(v)), "extract_sdf")
        d = list(dim)
        mi = self.mg_info()      # a slab of a multi-rank run gets the planes of the (global) box it owns
        own = max(0, min(lo[2] + d[2], mi["z1"]) - max(lo[2], mi["z0"])) if mi["n_ranks"] > 1 else d[2]
        if d[0] == 0 or own == 0:
            return list(lo), d, np.zeros((0, max(d[1], 0), max(d[0], 0)), np.float32)
        return list(lo), d, np.ctypeslib.as_array(v, shape=(own, d[1], d[0])).copy()

    def download_band(self, n=None):
        """one rank: the whole band.  A slab of a multi-rank run: pass n = row1 - row0 of mg_info (its own band voxels)"""
        if n is None:
            n = self.info().n_band
        b = np.empty(max(n, 1), np.int32)
        self._check(self._fn("download_band")(self.ctx, b.ctypes.data_as(C.c_void_p)), "download_band")
        return b[:n]

    def download_poses(self):
        i = self.info()
        p = np.empty((i.n_frames, 16), np.float32)
        self._check(self._fn("download_poses")(self.ctx, p.ctypes.data_as(C.c_void_p)), "download_poses")
        return p

    def download_light(self):
        i = self.info()
        shape = (3,) if self._settings.model == LED else (i.n_frames, i.light_stride)
        l = np.empty(shape, np.float32)
        self._check(self._fn("download_light")(self.ctx, l.ctypes.data_as(C.c_void_p)), "download_light")
        return l

    def upload_light(self, light):
        a, p = _fp(light)
        self._check(self._fn("upload_light")(self.ctx, p), "upload_light")

    # -- multi-GPU: attach the context to a rank (before load_scene / init); afterwards every call is collective
    def comm_init(self, rank, n_ranks, unique_id=None):
        """unique_id: the 128 bytes rank 0 got from comm_unique_id() (RCCL)"""
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id)) if unique_id is not None else None
        self._check(self._fn("comm_init")(self.ctx, buf, C.c_int(rank), C.c_int(n_ranks)), "comm_init")

    def comm_init_ext(self, ops, rank, n_ranks):
        """ops: a CommOps struct (caller-supplied transport); the caller keeps it (and its callbacks) alive"""
        self._comm_ops = ops
        self._check(self._fn("comm_init_ext")(self.ctx, C.byref(ops), C.c_int(rank), C.c_int(n_ranks)), "comm_init_ext")

    def debug_overlap_probe(self, reps=10):
        """ms of {distance sweep alone, solve alone, back to back, solve started beside the sweep on a second stream}"""
        out = (C.c_double * 4)()
        self._check(self._fn("debug_overlap_probe")(self.ctx, C.c_int(reps), out), "debug_overlap_probe")
        return list(out)

    def debug_normals_cache(self, width, height):
        """the FALS estimator's per-resolution cache as the device computed it: [9, height, width] float32"""
        out = np.empty((9, height, width), np.float32)
        self._check(self._fn("debug_normals_cache")(self.ctx, C.c_int(width), C.c_int(height), out.ctypes.data_as(C.c_void_p)), "debug_normals_cache")
        return out

    def comm_init_sockets(self, peer_fds, rank, n_ranks):
        """the built-in node-local transport: peer_fds[r] = fileno of a connected stream socket to rank r (the caller keeps the sockets open)"""
        fds = (C.c_int * n_ranks)(*[int(f) for f in peer_fds])
        self._check(self._fn("comm_init_sockets")(self.ctx, fds, C.c_int(rank), C.c_int(n_ranks)), "comm_init_sockets")

    def comm_allreduce_host(self, values):
        """sum over the ranks of a host vector of doubles (collective)"""
        a = np.ascontiguousarray(values, np.float64).copy()
        self._check(self._fn("comm_allreduce_host")(self.ctx, a.ctypes.data_as(C.c_void_p), C.c_int(a.size)), "comm_allreduce_host")
        return a

    def comm_stats(self):
        n = C.c_int64()
        self._check(self._fn("comm_stats")(self.ctx, C.byref(n)), "comm_stats")
        return n.value

    def set_stream(self, stream_ptr):
        self._check(self._fn("set_stream")(self.ctx, C.c_void_p(stream_ptr)), "set_stream")

    def mg_info(self):
        """{S (whole volume), rows held, row0, row1, halo, F, rank, n_ranks, need_lo, need_hi, z0, z1} of a context attached to a rank"""
        out = (C.c_int32 * 12)()
        self._check(self._fn("mg_info")(self.ctx, out), "mg_info")
        return dict(zip(self._MG_INFO_NAMES, list(out)))

    _MG_INFO_NAMES = ["S", "rows", "row0", "row1", "halo", "F", "rank", "n_ranks", "need_lo", "need_hi", "z0", "z1"]

    def get_tuning(self):
        """psgsdf_get_tuning: which environment knobs were set when the context was created, which dev-only ones this build ignores, what they resolved to"""
        import json
        f = getattr(self._lib, self._p + "get_tuning"); f.restype = C.c_int
        n = f(self.ctx, None, C.c_size_t(0))
        if n < 0:
            self._check(n, "get_tuning")
        buf = C.create_string_buffer(n + 1)
        f(self.ctx, buf, C.c_size_t(n + 1))
        return json.loads(buf.value.decode())

    # -- measurement
    def set_profiling(self, on):
        self._check(self._fn("set_profiling")(self.ctx, C.c_int(1 if on else 0)), "set_profiling")

    def watch_kernel(self, name):
        self._check(self._fn("watch_kernel")(self.ctx, name.encode() if name else None), "watch_kernel")

    def reset_kernel_times(self):
        self._check(self._fn("reset_kernel_times")(self.ctx), "reset_kernel_times")

    def kernel_times(self):
        cap = 64
        names = (C.c_char_p * cap)(); ms = (C.c_double * cap)(); n = (C.c_int64 * cap)()
        f = getattr(self._lib, self._p + "kernel_times"); f.restype = C.c_int
        k = f(self.ctx, names, ms, n, C.c_int(cap))
        return {names[i].decode(): (ms[i], n[i]) for i in range(k)}

    # -- test hooks
    def debug_dist_system(self, x=None):
        S = self.info().n_band
        diag = np.empty(S, np.float32); rhs = np.empty(S, np.float32)
        y = np.empty(S, np.float32) if x is not None else None
        xa = np.ascontiguousarray(x, np.float32) if x is not None else None
        self._check(self._fn("debug_dist_system")(self.ctx, diag.ctypes.data_as(C.c_void_p), rhs.ctypes.data_as(C.c_void_p),
                                                   xa.ctypes.data_as(C.c_void_p) if x is not None else None,
                                                   y.ctypes.data_as(C.c_void_p) if x is not None else None), "debug_dist_system")
        return diag, rhs, y

    def debug_time_pcg_pass(self, blocks=0, rows=2, ablate=0, reps=50, stamps=False):
        ms = C.c_double(0)
        st = np.zeros((max(blocks, 1), 8), np.int64) if stamps else None
        self._check(self._fn("debug_time_pcg_pass")(self.ctx, C.c_int(blocks), C.c_int(rows), C.c_int(ablate | (1024 if stamps else 0)), C.c_int(reps), C.byref(ms),
                                                     st.ctypes.data_as(C.c_void_p) if stamps else None), "debug_time_pcg_pass")
        return (ms.value, st) if stamps else ms.value

    def debug_time_pcg_solve(self, passes=16, reps=5):
        ms = C.c_double(0); shape = (C.c_int32 * 2)(); st = (C.c_double * 16)()
        self._check(self._fn("debug_time_pcg_solve")(self.ctx, C.c_int(passes), C.c_int(reps), C.byref(ms), shape, st), "debug_time_pcg_solve")
        return ms.value, (shape[0], shape[1]), [x for x in st]

    def debug_sync_stats(self):
        out = (C.c_int64 * 8)()
        self._check(self._fn("debug_sync_stats")(self.ctx, out), "debug_sync_stats")
        return dict(readbacks_checked=out[0], readbacks_late=out[1], persist_fallbacks=out[2] % 1000000, keyframes_compacted=out[2] // 1000000, speculative_starts=out[3] // 1000000, speculative_undos=out[3] % 1000000,
                    cross_rank_ready=out[4] % 10, halo_pushes=out[4] // 10, cross_rank_solves=out[5], cross_rank_mem_kind=out[6], probe_stale=out[7] // 1000000, probe_timeouts=out[7] % 1000000)

    def debug_rare_rows(self):
        r = C.c_int64(); w = C.c_int64()
        self._check(self._fn("debug_rare_rows")(self.ctx, C.byref(r), C.byref(w)), "debug_rare_rows")
        return r.value, w.value

    def debug_frame_system(self, block):
        i = self.info()
        if block == LIGHT:
            n = i.light_stride; nb = 1 if self._settings.model == LED else i.n_frames
        else:
            n = 6; nb = i.n_frames
        H = np.empty((nb, n, n), np.float64); b = np.empty((nb, n), np.float64)
        self._check(self._fn("debug_frame_system")(self.ctx, C.c_int(block), H.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)), "debug_frame_system")
        return H, b

    def set_frame_solver(self, mode):
        """0 = every frame's light / pose block solved directly (LDL^T in double; the engine's default), 1 = the reference's solver: ONE Eigen-style float
        Jacobi-PCG over the block-diagonal system of all frames (include/psgsdf.h psgsdf_set_frame_solver; the oracle: orc_set_frame_solver = its solver_mode)"""
        self._check(self._fn("set_frame_solver")(self.ctx, C.c_int(mode)), "set_frame_solver")

    def frame_solver_stats(self, block):
        it = C.c_int32(); err = C.c_double(); ok = C.c_int32(); ap = C.c_int32()
        self._check(self._fn("get_frame_solver_stats")(self.ctx, C.c_int(block), C.byref(it), C.byref(err), C.byref(ok), C.byref(ap)), "get_frame_solver_stats")
        return {"cg_iters": it.value, "cg_error": err.value, "cg_converged": ok.value, "applied": ap.value}

    def debug_frame_cg(self, H, b, max_it=0):
        """the mode-1 frame solver alone on the block-diagonal system H [nb][n][n], b [nb][n] -> (x [nb][n], iterations, error, converged)"""
        H = np.ascontiguousarray(H, np.float32); b = np.ascontiguousarray(b, np.float32)
        nb, n = b.shape
        x = np.empty((nb, n), np.float32); it = C.c_int32(); err = C.c_double(); ok = C.c_int32()
        self._check(self._fn("debug_frame_cg")(self.ctx, C.c_int(nb), C.c_int(n), H.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p),
                                               C.c_int(max_it), C.byref(it), C.byref(err), C.byref(ok)), "debug_frame_cg")
        return x, it.value, err.value, bool(ok.value)

    def debug_albedo_system(self):
        S = self.info().n_band
        H = np.empty((S, 3), np.float32); b = np.empty((S, 3), np.float32)
        self._check(self._fn("debug_albedo_system")(self.ctx, H.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)), "debug_albedo_system")
        return H, b


class CommXfer(C.Structure):
    _fields_ = [("ptr_dev", C.c_void_p), ("bytes", C.c_size_t), ("peer", C.c_int)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)
SENDRECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(CommXfer), C.c_int, C.POINTER(CommXfer), C.c_int, C.c_void_p)


class CommOps(C.Structure):
    """psgsdf_comm_ops: a caller-supplied transport for psgsdf_comm_init_ext"""
    _fields_ = [("user", C.c_void_p), ("allreduce_f64", ALLREDUCE_FN), ("sendrecv", SENDRECV_FN)]


def comm_unique_id() -> bytes:
    """psgsdf_comm_unique_id (ncclGetUniqueId): call on rank 0, hand the bytes to every rank's comm_init"""
    buf = (C.c_uint8 * 128)()
    f = engine_lib().psgsdf_comm_unique_id; f.restype = C.c_int
    rc = f(buf)
    if rc != 0:
        raise PsgsdfError(f"psgsdf_comm_unique_id failed rc={rc} (librccl could not be loaded?)")
    return bytes(buf)


def grid_of(sc) -> GridDesc:
    g = GridDesc()
    g.dim[:] = [int(x) for x in sc.dim]
    g.voxel_size = float(sc.voxel_size)
    g.shift[:] = [float(x) for x in sc.shift]
    g.truncation = float(sc.truncation)
    return g


_engine_lib = {}


def engine_lib(dev: bool = False) -> C.CDLL:
    """dlopen the HIP engine; raises (no fallback) if it has not been built.  dev: the development build with the fault-injection knobs (tests / tools)."""
    path = ENGINE_LIB_DEV if dev else ENGINE_LIB
    if path not in _engine_lib:
        if not os.path.exists(path):
            raise PsgsdfError(f"HIP engine {path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _engine_lib[path] = C.CDLL(path)
    return _engine_lib[path]


def load_engine(sc_or_grid, K, settings: Settings, device: int = 0, dev: bool = False) -> Api:
    grid = sc_or_grid if isinstance(sc_or_grid, GridDesc) else grid_of(sc_or_grid)
    return Api(engine_lib(dev or os.environ.get("PSGSDF_USE_DEV_LIB") == "1"), "psgsdf_", grid, K, settings, device)
