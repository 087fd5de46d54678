"""Wall time of psgsdf_track (frame-to-model tracking) per Gauss-Newton iteration at 256^3 / 640x480: 50 forced iterations per call."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psgradientsdf_amd import capi, synth
N, F, W, H = 256, 6, 640, 480
sc = synth.make_scene(N=N, F=F, W=W, H=H, model="SH1", zigzag=False, arc=30.0)
eng = capi.load_engine(sc, sc.K, capi.default_settings(capi.SH1), 0)
eng.volume_init(F)
for f in range(F):
    eng.integrate_frame(sc.images[f], sc.depth[f], sc.normals_cam[f], sc.poses_gt[f].reshape(4, 4), f)
pose = sc.poses_gt[1].reshape(4, 4).copy()
eng.track(sc.depth[2], pose.copy(), num_iterations=5, conv_threshold=0.0)
t0 = time.perf_counter(); reps = 10
for _ in range(reps):
    out = eng.track(sc.depth[2], pose.copy(), num_iterations=50, conv_threshold=0.0)
dt = time.perf_counter() - t0
print(json.dumps({"us_per_tracker_iteration": 1e6 * dt / (reps * 50), "ms_per_call_50_iterations": 1e3 * dt / reps}))
