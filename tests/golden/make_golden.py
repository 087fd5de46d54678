#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_small_*.npz: small seeded scenes and the oracle's outputs on them.

The reference ships no golden vectors and cannot be run here, so these fixtures pin the ORACLE against
accidental edits and give the GPU tests a committed expected output; they are data (inputs are regenerated from the seed
by psgradientsdf_amd/synth.py, outputs are stored).  Run from the repo root:  python tests/golden/make_golden.py

Round 6: generated with the oracle's solver_mode 1 -- the light and pose blocks solved as the reference solves them, ONE Eigen-style float Jacobi-PCG over
all frames' blocks (PsOptimizer.cpp:175-234, LedOptimizer.cpp:134-275) -- instead of the per-block direct solves of the earlier fixtures.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from psgradientsdf_amd import capi, synth  # noqa: E402
from oracle import oracle  # noqa: E402

# SH2 with twelve keyframes: with fewer than ~ten the reference's global light solve (9 unknowns per keyframe, cond ~2e4) uses up its 2n passes at a residual of
# 1e-4 .. 1e-6, reports NoConvergence and applies a step that depends on the last bit of its inputs -- a legitimate course of the reference, and a useless fixture
CASES = {"SH1": dict(N=24, F=4, W=96, H=72), "SH2": dict(N=24, F=12, W=96, H=72), "LED": dict(N=24, F=4, W=96, H=72)}


def run(model):
    kw = CASES[model]
    sc = synth.make_scene(model=model, **kw)
    st = capi.default_settings(sc.model_id, reg_weight_l=1.0)
    o = oracle.Oracle(sc, sc.K, st, solver_mode=1)
    o.load_scene(sc)
    o.init_albedo()
    e_tot0 = o.normalize_weights()
    e0 = o.energy()
    recs = o.iterate(capi.ALL, 2)
    band = o.download_band()
    v = o.download_volume()
    return sc, dict(solver_mode=np.array(1), band=band, energy0=np.array(e0), e_total0=e_tot0, e_total=np.array([r["e_total"] for r in recs]),
                    e_after=np.array([r["e_after"] for r in recs]), cg_iters=np.array([r["cg_iters"] for r in recs]),
                    dist=v["dist"][band], rgb=v["rgb"][:, band], grad=v["grad"][:, band], poses=o.download_poses(), light=o.download_light(),
                    scene_checksum=np.array([float(np.abs(sc.dist).sum()), float(sc.images.sum()), float(sc.poses.sum())]))


if __name__ == "__main__":
    for m in CASES:
        sc, out = run(m)
        np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), f"oracle_small_{m}.npz"), **out)
        print(m, "band", len(out["band"]), "E", out["e_total"])
