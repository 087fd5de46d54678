#!/usr/bin/env python3
"""Regenerates tests/golden/sokrates_small/: 8 frames of the reference's demo data set (data/sokrates-mvs: 34 RGB-D frames,
1139x1709, depth uint16 mm) sub-sampled 6x (285x190), intrinsics scaled accordingly, the matching lines of pose.txt.
Data only -- used by tests/test_voxelps_gpu.py and tests/test_host_tools.py.  Needs /root/reference (build container)."""
import os
import numpy as np
from PIL import Image

src = "/root/reference/data/sokrates-mvs/"
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sokrates_small") + "/"
os.makedirs(dst, exist_ok=True)
step, frames = 6, [1, 3, 5, 7, 9, 11, 13, 15]
K = np.loadtxt(src + "intrinsics.txt")[:3]
for n, f in enumerate(frames, start=1):
    c = np.asarray(Image.open(src + f"color{f:06d}.png").convert("RGB")); d = np.asarray(Image.open(src + f"depth{f:06d}.png"))
    Image.fromarray(c[step // 2::step, step // 2::step]).save(dst + f"color{n:06d}.png", optimize=True)
    Image.fromarray(d[step // 2::step, step // 2::step].astype(np.uint16)).save(dst + f"depth{n:06d}.png", optimize=True)
K2 = K.copy(); K2[0, 0] /= step; K2[1, 1] /= step; K2[0, 2] = (K[0, 2] - step // 2) / step; K2[1, 2] = (K[1, 2] - step // 2) / step
np.savetxt(dst + "intrinsics.txt", K2, fmt="%.6f")
lines = open(src + "pose.txt").read().strip().split("\n")
open(dst + "pose.txt", "w").write("\n".join(lines[f - 1] for f in frames) + "\n")
