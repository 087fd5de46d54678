#!/usr/bin/env python3
"""Which voxels sit outside the north star's 1e-4 voxel after a few iterations, and why?  

Engine and oracle run the SAME sub-steps side by side; after every block the band voxels whose distances differ by more than 1e-4 voxel are
listed with what happened to them in that block on both sides: the distance update they received, whether it passed the accept rule
|delta| < sqrt(3) vs (OptimizerAux.cpp:162-188), their albedo (the accept rule 0 < rho < 1, OptimizerAux.cpp:120-150), how many of their
observations are in the image.  Usage (GPU box): python tools/margin_voxels.py [model] [N] [iterations] [F] [W] [H]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from psgradientsdf_amd import capi, synth  # noqa: E402
from oracle import oracle  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "SH1"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 48
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
F, W, H = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((4, 6), (5, 160), (6, 120)))
mid = synth.MODELS[model]
sc = synth.make_scene(N=N, F=F, W=W, H=H, model=model)
st = capi.default_settings(mid)
eng = capi.load_engine(sc, sc.K, st, 0); orc = oracle.Oracle(sc, sc.K, st, threads=min(32, os.cpu_count() or 1))
for api in (eng, orc):
    api.load_scene(sc); api.init_albedo(); api.normalize_weights()
vs = float(sc.voxel_size); lim = np.sqrt(3.0) * vs
band = eng.download_band()
order = [capi.LIGHT, capi.ALBEDO, capi.DIST, capi.POSE] if mid == capi.LED else [capi.ALBEDO, capi.LIGHT, capi.DIST, capi.POSE]
names = {capi.ALBEDO: "albedo", capi.LIGHT: "light", capi.DIST: "dist", capi.POSE: "pose"}
known = set()
for it in range(iters):
    for blk in order:
        be, bo = eng.download_volume(), orc.download_volume()
        se, so = eng.step(blk), orc.step(blk)
        ae, ao = eng.download_volume(), orc.download_volume()
        d = np.abs(ae["dist"][band] - ao["dist"][band]) / vs
        rho = np.abs(ae["rgb"][:, band] - ao["rgb"][:, band]).max(0)
        bad = np.nonzero(d > 1e-4)[0]
        new = [j for j in bad if j not in known]
        print(f"[{it}] {names[blk]:6s} q999 |dd|/vs {np.quantile(d, 0.999):.2e}  max {d.max():.2e}  > 1e-4: {len(bad)} ({len(new)} new)   max |d rho| {rho.max():.2e}"
              + (f"   cg {se['cg_iters']} / {so['cg_iters']}  accepted {se['n_accepted']} / {so['n_accepted']}" if blk == capi.DIST else ""))
        for j in new[:8]:
            lin = band[j]
            de_, do_ = be["dist"][lin] - ae["dist"][lin], bo["dist"][lin] - ao["dist"][lin]      # the update each side applied (0 = rejected)
            print(f"      row {j} voxel {lin}: |dd| = {d[j]:.2e} vs   update engine {de_ / vs:+.5f} vs  oracle {do_ / vs:+.5f} vs   (accept limit {lim / vs:.3f} vs)"
                  f"   d before {be['dist'][lin] / vs:+.4f} / {bo['dist'][lin] / vs:+.4f}   rho {ae['rgb'][:, lin]} / {ao['rgb'][:, lin]}")
        known.update(new)
