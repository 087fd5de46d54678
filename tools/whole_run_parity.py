#!/usr/bin/env python3
"""Whole-run parity: psgsdf_optimize against orc_optimize, both run to their OWN termination with the reference's own
iteration budgets (config_skorates.json / config_basket_LED.json: max iter 100, converge threshold 5e-3; PsOptimizer.cpp:303-428,
LedOptimizer.cpp:343-478).  Prints, per configuration, the record counts, the flags, the worst per-iteration e_total deviation, the norm-wise SDF
error, how many band voxels differ by more than 1e-4 voxel and where those sit (inside / outside the band's own width sqrt(3) vs).

    python tools/whole_run_parity.py [headline] [sokrates] [led128] [sh1_64] ...      (GPU box; writes gpurun_out/whole_run_parity.json)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from psgradientsdf_amd import capi, synth  # noqa: E402
from oracle import oracle  # noqa: E402

THREADS = min(64, os.cpu_count() or 1)


def whole_run(eng, orc, vs_final, label):
    t0 = time.time(); re_, ce = eng.optimize(capi.ALL); te = time.time() - t0
    t0 = time.time(); ro, co = orc.optimize(capi.ALL); to = time.time() - t0
    out = {"label": label, "records": [len(re_), len(ro)], "result": [bool(ce), bool(co)], "engine_s": round(te, 2), "oracle_s": round(to, 1)}
    n = min(len(re_), len(ro))
    out["flags_equal"] = [(r["converged"], r["diverged"], r["upsampled"]) for r in re_[:n]] == [(r["converged"], r["diverged"], r["upsampled"]) for r in ro[:n]]
    out["e_total_rel_max"] = max((abs(a["e_total"] - b["e_total"]) / abs(b["e_total"]) for a, b in zip(re_, ro)), default=0.0)
    out["e_total_rel_by_iter"] = [float(f"{abs(a['e_total'] - b['e_total']) / abs(b['e_total']):.2e}") for a, b in zip(re_, ro)]
    out["cg_iters"] = [[r["cg_iters"] for r in re_], [r["cg_iters"] for r in ro]]
    out["last"] = {"engine": {k: re_[-1][k] for k in ("e_total", "rel_diff", "converged", "diverged")}, "oracle": {k: ro[-1][k] for k in ("e_total", "rel_diff", "converged", "diverged")}}
    be, bo = eng.download_band(), orc.download_band()
    out["band_equal"] = bool(np.array_equal(be, bo))
    if out["band_equal"]:
        ve, vo = eng.download_volume(), orc.download_volume()
        a = ve["dist"][be].astype(np.float64); b = vo["dist"][be].astype(np.float64)
        d = np.abs(a - b) / vs_final
        lim = np.sqrt(3.0) * vs_final
        w = d > 1e-4
        out["sdf"] = {"n_band": int(len(d)), "rel": float(np.linalg.norm(a - b) / np.linalg.norm(b)), "q999_vs": float(np.quantile(d, 0.999)), "max_vs": float(d.max()),
                      "above_1e-4_vs": int(w.sum()), "above_1e-3_vs": int((d > 1e-3).sum()),
                      "wanderers_outside_band_width_both": int((w & (np.abs(a) > lim) & (np.abs(b) > lim)).sum()),
                      "wanderers_outside_band_width_either": int((w & ((np.abs(a) > lim) | (np.abs(b) > lim))).sum()),
                      "wanderers_min_abs_d_vs": float(np.minimum(np.abs(a[w]), np.abs(b[w])).min() / vs_final) if w.any() else None,
                      "rel_inside_band_width": float(np.linalg.norm((a - b)[np.abs(b) <= lim]) / np.linalg.norm(b[np.abs(b) <= lim])),
                      "max_vs_inside_band_width": float(d[(np.abs(b) <= lim) & (np.abs(a) <= lim)].max())}
        out["rgb_max"] = float(np.abs(ve["rgb"][:, be] - vo["rgb"][:, be]).max())
        out["pose_max"] = float(np.abs(eng.download_poses() - orc.download_poses()).max())
        lo = orc.download_light()
        out["light_rel"] = float(np.abs(eng.download_light() - lo).max() / np.abs(lo).max())
    return out


def headline():
    sc = synth.make_scene(N=256, F=50, W=640, H=480, model="SH1")
    st = capi.default_settings(capi.SH1)      # config_skorates.json: max iter 100, 5e-3
    eng = capi.load_engine(sc, sc.K, st, 0); orc = oracle.Oracle(sc, sc.K, st, threads=THREADS)
    for api in (eng, orc):
        api.load_scene(sc)
    return whole_run(eng, orc, float(sc.voxel_size), "headline 256^3 x 50, SH1, config_skorates.json settings")


def synth_case(model, N, F, W=320, H=240, **kw):
    sc = synth.make_scene(N=N, F=F, W=W, H=H, model=model)
    st = capi.default_settings(sc.model_id, **kw)
    eng = capi.load_engine(sc, sc.K, st, 0); orc = oracle.Oracle(sc, sc.K, st, threads=THREADS)
    for api in (eng, orc):
        api.load_scene(sc)
    return whole_run(eng, orc, float(sc.voxel_size) / (2 if kw.get("upsample") else 1), f"synthetic {model} {N}^3 x {F} ({W}x{H}) {kw}")


def sokrates():
    import test_configs_gpu as tc
    K, color, depth, poses = tc.load_sokrates()
    vs = 0.004
    g = capi.GridDesc(); g.dim[:] = [128, 128, 128]; g.voxel_size = vs; g.shift[:] = [float(x) for x in tc.centroid(K, depth[0], poses[0])]; g.truncation = 5 * vs
    st = capi.default_settings(capi.SH1)
    eng = capi.load_engine(g, K.reshape(-1), st, 0); orc = oracle.Oracle(g, K.reshape(-1), st, threads=THREADS)
    orc.volume_init(len(poses))
    for f in range(len(poses)):
        orc.integrate_frame(color[f], depth[f], orc.estimate_normals(depth[f]), poses[f], f, z_min=0.5, z_max=3.5)
    vo = orc.download_volume()      # ONE fused volume for both (the fusion's own parity is tests/test_configs_gpu.py's subject)
    eng.upload_volume(vo["dist"], vo["grad"], vo["weight"], vo["rgb"], orc.download_vis_seq(1), 1)
    key_poses = np.stack(poses).reshape(-1, 16).copy(); key_poses[0] = np.eye(4, dtype=np.float32).reshape(16)     # B1, main_ps.cpp:139
    imgs = np.stack(color)
    for api in (eng, orc):
        api.set_keyframes(np.arange(len(poses), dtype=np.int32), imgs, key_poses); api.init()
    return whole_run(eng, orc, vs, "configs[0]: sokrates-mvs frames 0-20 (sub-sampled 3x), 128^3 / 4 mm, config_skorates.json (max iter 100, 5e-3)")


CASES = {
    "headline": headline,
    "sokrates": sokrates,
    "led128": lambda: synth_case("LED", 128, 30, 640, 480, reg_weight_n=0.1, reg_weight_l=5.0, damping=3.0, upsample=1),      # config_basket_LED.json (upsample: true)
    "led128_noup": lambda: synth_case("LED", 128, 30, 640, 480, reg_weight_n=0.1, reg_weight_l=5.0, damping=3.0),
    "sh1_64": lambda: synth_case("SH1", 64, 12),
    "sh2_64": lambda: synth_case("SH2", 64, 12),
    "led_64": lambda: synth_case("LED", 64, 12, reg_weight_n=0.1, reg_weight_l=5.0, damping=3.0),
}

if __name__ == "__main__":
    names = sys.argv[1:] or ["sokrates", "headline", "led128"]
    res = []
    for n in names:
        r = CASES[n]()
        res.append(r)
        print(json.dumps(r), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "whole_run_parity.json"), "w"), indent=1)
