#!/usr/bin/env python3
"""Which of the engine's deviations from the reference's arithmetic feeds the drift of a whole optimisation?  

The engine's device arithmetic differs from the reference's in ways that are each at rounding level for ONE observation: FMA
contraction, v_rcp / v_log in the robust weights, float bilinear weights, float sums over a voxel's observations, Jacobian chains contracted from the
right -- plus two solver substitutions (direct block solves for light / pose, pipelined double recurrences for the distance solve).  A whole run on real
frames amplifies rounding (tests/test_wholerun_gpu.py), and the product ends 6-9e-3 (norm-wise) from the oracle on the reference's demo frames.

This tool runs development builds in which the deviations are switched off -- csrc/Makefile `make strict STRICT=<mask>` (device_common.h PSG_STRICT) and the
run-time solver switches -- in lockstep with the oracle (the reference's solver, solver_mode 1), one Gauss-Newton iteration at a time until the
reference's own stop rule ends the loop, and records after every iteration how far each variant is from the oracle:
    product                         libpsgsdf.so as shipped
    product+solvers                 ... with the reference's light / pose solver and the classic (per-pass, float) distance recurrences
    strict(31)+solvers              every deviation off
    strict(31), product solvers     only the arithmetic strict
    strict(31 ^ bit)+solvers        every deviation off but ONE (which one matters?)
    strict(bit)+solvers             the product's arithmetic with ONE deviation off
and the oracle's own FMA build as the yardstick.

    cd psgradientsdf_amd/csrc && for m in 31 30 29 27 23 15 1 2 4 8 16; do make -s -j16 strict STRICT=$m; done
    python tools/deviations.py [sokrates] [headline]          (GPU box; writes gpurun_out/deviations.json)
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from psgradientsdf_amd import capi  # noqa: E402
from oracle import oracle  # noqa: E402

CSRC = os.path.join(ROOT, "psgradientsdf_amd", "csrc")
BITS = {1: "no FMA contraction", 2: "IEEE weights / logf, r / lambda", 4: "bilinear weights and 1/z through double", 8: "observation sums in double", 16: "Jacobian chains in the reference's order"}
SOLVERS = {"PSGSDF_FRAME_SOLVE": "eigen", "PSGSDF_PCG_PIPELINE": "0", "PSGSDF_PCG_PERSIST": "0"}      # the reference's light / pose solver; the classic per-pass float recurrences


def variants():
    v = [("product", None, {}), ("product+solvers", None, SOLVERS), ("strict31+solvers", 31, SOLVERS), ("strict31, product solvers", 31, {})]
    v += [(f"strict31+solvers, ldlt frame solve", 31, {k: w for k, w in SOLVERS.items() if k != "PSGSDF_FRAME_SOLVE"})]
    v += [(f"strict31+solvers, pipelined distance solve", 31, {"PSGSDF_FRAME_SOLVE": "eigen"})]
    for b, what in BITS.items():
        v.append((f"all strict but: {what} (mask {31 ^ b})", 31 ^ b, SOLVERS))
    for b, what in BITS.items():
        v.append((f"product + only: {what} (mask {b})", b, SOLVERS))
    return v


def engine_of(mask, env, make_ctx):
    path = capi.ENGINE_LIB if mask is None else os.path.join(CSRC, f"libpsgsdf_strict{mask}.so")
    if not os.path.exists(path):
        return None
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return make_ctx(C.CDLL(path))
    finally:
        for k, w in old.items():
            if w is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = w


def run_case(name, scene):
    """scene() -> (load(api), grid, K, settings, vs)"""
    load, grid, K, st, vs = scene()
    threads = min(64, os.cpu_count() or 1)
    # the oracle's course, recorded once: per iteration the band distances, albedo, poses, record
    ref = {}
    for kind in ("orc", "orc_fma"):
        o = oracle.Oracle(grid, K, st, threads=threads, solver_mode=1, fma=(kind == "orc_fma")); load(o)
        o.init_albedo(); o.normalize_weights()
        band = o.download_band(); states = []
        t0 = time.time()
        for it in range(100):
            r = o.iterate(capi.ALL, 1)[0]
            v = o.download_volume()
            states.append({"dist": v["dist"][band].astype(np.float64), "rgb": v["rgb"][:, band].copy(), "poses": o.download_poses().copy(), "rec": r})
            if r["rel_diff"] < st.conv_threshold or r["diverged"]:
                break
        ref[kind] = (band, states)
        print(f"# {name}: {kind} ran {len(states)} iterations in {time.time() - t0:.0f} s", flush=True)
        o.close()
    band, S = ref["orc"]

    def against_oracle(states_x, band_x):
        rows = []
        if not np.array_equal(band_x, band):
            return [{"band_differs": True}]
        for it, (x, y) in enumerate(zip(states_x, S)):
            d = np.abs(x["dist"] - y["dist"]) / vs
            rows.append({"iter": it + 1, "rel": float(np.linalg.norm(x["dist"] - y["dist"]) / np.linalg.norm(y["dist"])), "above_1e-4": int((d > 1e-4).sum()), "max_vs": float(d.max()),
                         "rgb": float(np.abs(x["rgb"] - y["rgb"]).max()), "pose": float(np.abs(x["poses"] - y["poses"]).max()),
                         "e_total_rel": abs(x["rec"]["e_total"] - y["rec"]["e_total"]) / abs(y["rec"]["e_total"]), "cg": [x["rec"]["cg_iters"], y["rec"]["cg_iters"]],
                         "same_flags": bool((x["rec"]["rel_diff"] < st.conv_threshold) == (y["rec"]["rel_diff"] < st.conv_threshold) and x["rec"]["diverged"] == y["rec"]["diverged"])})
        return rows

    out = {"case": name, "oracle_iterations": len(S), "n_band": int(len(band)), "variants": {}}
    out["variants"]["oracle's own FMA build (yardstick)"] = {"curve": against_oracle(ref["orc_fma"][1], ref["orc_fma"][0]), "iterations": len(ref["orc_fma"][1])}
    for label, mask, env in variants():
        eng = engine_of(mask, env, lambda lib: capi.Api(lib, "psgsdf_", grid, K, st, 0))
        if eng is None:
            print(f"# {name}: {label}: library not built, skipped", flush=True); continue
        load(eng); eng.init_albedo(); eng.normalize_weights()
        states = []
        for it in range(len(S) + 5):
            r = eng.iterate(capi.ALL, 1)[0]
            v = eng.download_volume()
            states.append({"dist": v["dist"][band].astype(np.float64), "rgb": v["rgb"][:, band].copy(), "poses": eng.download_poses().copy(), "rec": r})
            if r["rel_diff"] < st.conv_threshold or r["diverged"]:
                break
        curve = against_oracle(states, eng.download_band())
        tun = eng.get_tuning()["effective"]
        out["variants"][label] = {"mask": mask, "env": env, "iterations": len(states), "frame_solve": tun.get("frame_solve"), "curve": curve}
        last = curve[min(len(curve), len(S)) - 1]
        first_bad = next((c["iter"] for c in curve if c.get("above_1e-4", 0) > 0), None)
        print(json.dumps({"case": name, "variant": label, "iterations": len(states), "final": {k: (float(f"{w:.3g}") if isinstance(w, float) else w) for k, w in last.items()}, "first_iteration_with_a_voxel_beyond_1e-4": first_bad}), flush=True)
        eng.close()
    return out


def sokrates_scene():
    import test_configs_gpu as tc
    K, color, depth, poses = tc.load_sokrates()
    vs = 0.004
    g = capi.GridDesc(); g.dim[:] = [128, 128, 128]; g.voxel_size = vs; g.shift[:] = [float(x) for x in tc.centroid(K, depth[0], poses[0])]; g.truncation = 5 * vs
    st = capi.default_settings(capi.SH1)
    base = oracle.Oracle(g, K.reshape(-1), st, threads=min(64, os.cpu_count() or 1))
    base.volume_init(len(poses))
    for f in range(len(poses)):
        base.integrate_frame(color[f], depth[f], base.estimate_normals(depth[f]), poses[f], f, z_min=0.5, z_max=3.5)
    vo = base.download_volume(); vis = base.download_vis_seq(1); base.close()
    key_poses = np.stack(poses).reshape(-1, 16).copy(); key_poses[0] = np.eye(4, dtype=np.float32).reshape(16)
    imgs = np.stack(color)

    def load(api):
        api.upload_volume(vo["dist"], vo["grad"], vo["weight"], vo["rgb"], vis, 1)
        api.set_keyframes(np.arange(len(poses), dtype=np.int32), imgs, key_poses); api.init()
    return load, g, K.reshape(-1), st, vs


def synth_scene(model, N, F, W, H, **kw):
    from psgradientsdf_amd import synth
    sc = synth.make_scene(N=N, F=F, W=W, H=H, model=model)
    st = capi.default_settings(sc.model_id, **kw)
    return (lambda api: api.load_scene(sc)), capi.grid_of(sc), sc.K, st, float(sc.voxel_size)


CASES = {"sokrates": sokrates_scene, "headline": lambda: synth_scene("SH1", 256, 50, 640, 480), "sh1_96": lambda: synth_scene("SH1", 96, 20, 320, 240), "sh2_64": lambda: synth_scene("SH2", 64, 12, 320, 240)}

if __name__ == "__main__":
    names = sys.argv[1:] or ["sokrates"]
    res = [run_case(n, CASES[n]) for n in names]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "deviations.json"), "w"), indent=1)
