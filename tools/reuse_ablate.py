#!/usr/bin/env python3
"""Bound on what ONE projection + sample for the albedo -> light pair could win, before building it.

The albedo sweep and the light sweep that follows it sample the same observations at the same geometry (only the albedo changed in between).  Sharing the sample
means: the albedo sweep stores each observation's colour (12 B) at its slot of the frame-major list (one voxel -> slot look-up), the light sweep reads it
(coalesced: its loop index IS the slot) instead of projecting and gathering four taps.  The development library has both halves as a TIMING ablation
(PSGSDF_ABLATE_REUSE=1: SweepArgs::obs_I; the slot is approximated, results are wrong): this tool measures the two sweeps with and without it, rocprofv3-free,
from the engine's own synchronous event pass, and the whole iteration through psgsdf_iterate.

    python tools/reuse_ablate.py            (GPU box; writes gpurun_out/reuse_ablate.json)

NEEDS THE ENGINE AT COMMIT 42a383b: the ablation hooks (SweepArgs::obs_I, PSGSDF_ABLATE_REUSE) were removed again after the measurement -- the shared sample is a
net loss (profiles/r06_reuse_ablate.json, profiles/r06_notes.md section 4) and the hooks cost the product's sweeps a branch per observation.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from psgradientsdf_amd import capi, synth  # noqa: E402


def run(sc, st, ablate, u8):
    if ablate:
        os.environ["PSGSDF_ABLATE_REUSE"] = "1"
    else:
        os.environ.pop("PSGSDF_ABLATE_REUSE", None)
    eng = capi.load_engine(sc, sc.K, st, 0, dev=True)
    eng.load_scene(sc, u8=u8); eng.init_albedo(); eng.normalize_weights()
    eng.iterate(capi.ALL, 3)
    t0 = time.perf_counter(); eng.iterate(capi.ALL, 40); ms_it = (time.perf_counter() - t0) / 40 * 1e3      # (psgsdf_iterate returns with the last record read back: synchronous)
    eng.set_profiling(True); eng.reset_kernel_times(); eng.iterate(capi.ALL, 5)
    kt = {k: round(1e3 * v[0] / max(v[1], 1), 2) for k, v in eng.kernel_times().items()}
    eng.close()
    return {"ms_per_iteration": round(ms_it, 4), "us_per_launch_sync_pass": {k: kt.get(k) for k in ("sweep_albedo", "sweep_light", "sweep_dist", "sweep_pose", "pcg_solve")}}


if __name__ == "__main__":
    out = {}
    for model in ("SH1", "SH2"):
        for u8 in (False, True):
            sc = synth.make_scene(N=256, F=50, W=640, H=480, model=model, u8=u8)
            st = capi.default_settings(synth.MODELS[model])
            rows = [run(sc, st, ab, u8) for ab in (False, True, False, True)]
            base = [r for r in rows[0::2]]; abl = [r for r in rows[1::2]]
            ms0 = min(r["ms_per_iteration"] for r in base); ms1 = min(r["ms_per_iteration"] for r in abl)
            out[f"{model}{'_u8' if u8 else ''}"] = {"product_path": base, "shared_sample_ablation": abl, "iteration_gain_upper_bound": round(ms0 / ms1 - 1.0, 4)}
            print(model, "u8" if u8 else "f32", json.dumps(out[f"{model}{'_u8' if u8 else ''}"]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "reuse_ablate.json"), "w"), indent=1)
