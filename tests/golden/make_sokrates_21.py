#!/usr/bin/env python3
"""Regenerates tests/golden/sokrates_21/: frames 0-20 of the reference's demo data set (data/sokrates-mvs: RGB-D frames of
1139x1709 pixels, depth uint16 mm) -- the frames BASELINE.json's configs[0] names -- sub-sampled 3x by pixel picking (380x570), intrinsics
scaled accordingly, the matching lines of pose.txt.  Data only (used by tests/test_configs_gpu.py).  Needs /root/reference (build container)."""
import os
import numpy as np
from PIL import Image

src = "/root/reference/data/sokrates-mvs/"
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sokrates_21") + "/"
os.makedirs(dst, exist_ok=True)
step, frames = 3, list(range(1, 22))
K = np.loadtxt(src + "intrinsics.txt")[:3]
for n, f in enumerate(frames, start=1):
    c = np.asarray(Image.open(src + f"color{f:06d}.png").convert("RGB")); d = np.asarray(Image.open(src + f"depth{f:06d}.png"))
    Image.fromarray(c[step // 2::step, step // 2::step]).save(dst + f"color{n:06d}.png", optimize=True)
    Image.fromarray(d[step // 2::step, step // 2::step].astype(np.uint16)).save(dst + f"depth{n:06d}.png", optimize=True)
K2 = K.copy(); K2[0, 0] /= step; K2[1, 1] /= step; K2[0, 2] = (K[0, 2] - step // 2) / step; K2[1, 2] = (K[1, 2] - step // 2) / step
np.savetxt(dst + "intrinsics.txt", K2, fmt="%.6f")
lines = open(src + "pose.txt").read().strip().split("\n")
open(dst + "pose.txt", "w").write("\n".join(lines[f - 1] for f in frames) + "\n")
