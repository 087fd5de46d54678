#!/usr/bin/env python3
"""Regenerates tests/golden/sokrates_native_4/: frames 0-3 of the reference's demo data set (data/sokrates-mvs) at their NATIVE resolution
(1139 x 1709 pixels, depth uint16 mm), intrinsics and the matching lines of pose.txt -- the native-resolution leg of BASELINE.json's configs[0]
(tests/test_configs_gpu.py::test_config0_native_resolution; the 21-frame fixture sokrates_21 is sub-sampled 3x to keep the repository small).
Data only: the pixels are decoded and re-encoded, nothing else.  Needs /root/reference (build container)."""
import os
import numpy as np
from PIL import Image

src = "/root/reference/data/sokrates-mvs/"
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sokrates_native_4") + "/"
os.makedirs(dst, exist_ok=True)
frames = [1, 2, 3, 4]
for n, f in enumerate(frames, start=1):
    c = np.asarray(Image.open(src + f"color{f:06d}.png").convert("RGB")); d = np.asarray(Image.open(src + f"depth{f:06d}.png"))
    Image.fromarray(c).save(dst + f"color{n:06d}.png", optimize=True)
    Image.fromarray(d.astype(np.uint16)).save(dst + f"depth{n:06d}.png", optimize=True)
np.savetxt(dst + "intrinsics.txt", np.loadtxt(src + "intrinsics.txt")[:3], fmt="%.6f")
lines = open(src + "pose.txt").read().strip().split("\n")
open(dst + "pose.txt", "w").write("\n".join(lines[f - 1] for f in frames) + "\n")
