#!/usr/bin/env python3
"""Micro-benchmark of the state-producer kernels: integrate_frame, FALS normals, tracker pass at 256^3 / 640x480.
Prints a JSON line with HIP-event kernel times and the HBM roofline of integrate_frame."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psgradientsdf_amd import capi, synth

N, F, W, H = 256, 8, 640, 480
sc = synth.make_scene(N=N, F=F, W=W, H=H, model="SH1")
eng = capi.load_engine(sc, sc.K, capi.default_settings(capi.SH1), 0)
eng.volume_init(F)
eng.set_profiling(True)
for rep in range(3):
    for f in range(F):
        eng.integrate_frame(sc.images[f], sc.depth[f], sc.normals_cam[f], sc.poses_gt[f], f)
for f in range(F):
    eng.estimate_normals(sc.depth[f])
    eng.track(sc.depth[f], sc.poses_gt[f], num_iterations=3)
kt = eng.kernel_times()
nvox = N ** 3
# algorithmic bytes of one fusion sweep: every voxel is projected (no memory) and taps the depth image once if it lands in it
# (4 B, upper bound: all voxels); only voxels inside the frame's truncation band touch their record: 8 float planes + one
# visibility word read and written back (2 x 40 B).  The number of voxels a frame updates is counted from the visibility bits.
vis = eng.download_vis_seq((F + 63) // 64)     # sequence-indexed visibility bits written by the fusion
per_frame = float(sum(int(np.unpackbits(np.ascontiguousarray(vis[:, w]).view(np.uint8)).sum()) for w in range(vis.shape[1]))) / F   # (re-integrating a frame sets the same bit)
out = {"grid": N, "image": [W, H], "voxels_updated_per_frame": per_frame}
for k, (ms, n) in kt.items():
    out[k] = {"avg_ms": ms / n, "launches": n}
b = 4.0 * nvox + 80.0 * per_frame
t = kt["integrate_frame"][0] / kt["integrate_frame"][1] * 1e-3
out["integrate_frame"]["algorithmic_bytes"] = b
out["integrate_frame"]["GBs"] = b / t / 1e9
out["integrate_frame"]["frac_of_8TBs"] = b / t / 1e9 / 8000.0
print(json.dumps(out))
