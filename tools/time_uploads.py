#!/usr/bin/env python3
"""Wall-clock of the host -> device hand-overs of the boundary (pageable caller memory, as a C++ host has it): psgsdf_set_keyframes at 50 x 640 x 480 floats
(184 MB), psgsdf_upload_volume at 256^3, psgsdf_integrate_frame per frame -- fresh arrays every call (what a decoder hands over) and reused ones."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psgradientsdf_amd import capi, synth

sc = synth.make_scene(N=int(os.environ.get("N", "128")), F=50, W=640, H=480, model="SH1")
eng = capi.load_engine(sc, sc.K, capi.default_settings(capi.SH1), 0)
for rep in range(3):
    t0 = time.perf_counter(); eng.upload_volume(sc.dist, sc.grad, sc.weight, sc.rgb, sc.vis, sc.vis_words); t1 = time.perf_counter()
    print(f"upload_volume {sc.dim[0]}^3: {1e3 * (t1 - t0):.1f} ms ({(sc.dist.nbytes * 8 + sc.vis.nbytes) / 1e6:.0f} MB)")
idx = np.arange(sc.F, dtype=np.int32)
for rep in range(3):
    img = sc.images.copy()      # a fresh allocation, like the host's std::vector
    t0 = time.perf_counter(); eng.set_keyframes(idx, img, sc.poses); t1 = time.perf_counter()
    print(f"set_keyframes fresh array: {1e3 * (t1 - t0):.1f} ms ({img.nbytes / 1e6:.0f} MB)")
for rep in range(3):
    t0 = time.perf_counter(); eng.set_keyframes(idx, sc.images, sc.poses); t1 = time.perf_counter()
    print(f"set_keyframes same array:  {1e3 * (t1 - t0):.1f} ms")
eng.volume_init(sc.F)
for rep in range(6):
    im, dp = sc.images[rep].copy(), sc.depth[rep].copy()
    t0 = time.perf_counter(); eng.integrate_frame(im, dp, None, sc.poses_gt[rep], rep, z_min=0.05, z_max=10.0); t1 = time.perf_counter()
    print(f"integrate_frame fresh arrays: {1e3 * (t1 - t0):.2f} ms")
