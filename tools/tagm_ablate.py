#!/usr/bin/env python3
"""us per pass of the self-validating solve under the timing ablations of the development library (wrong results, timing only):
PSGSDF_PCG_ABLATE=1 every gather reads the row's own element, 4 the 8 in-plane columns are not gathered (24 of 54 loads per thread)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psgradientsdf_amd import capi, synth
sc = synth.make_scene(N=256, F=12, W=320, H=240, model="SH1")
out = {}
for name, env in (("production", {}), ("own_element", {"PSGSDF_PCG_ABLATE": "1"}), ("no_in_plane_columns", {"PSGSDF_PCG_ABLATE": "4"}), ("both", {"PSGSDF_PCG_ABLATE": "5"}), ("untagged", {"PSGSDF_PCG_TAGM": "0"}), ("untagged_no_in_plane", {"PSGSDF_PCG_TAGM": "0", "PSGSDF_PCG_ABLATE": "4"})):
    for k in ("PSGSDF_PCG_ABLATE", "PSGSDF_PCG_TAGM"):
        os.environ.pop(k, None)
    os.environ.update(env)
    eng = capi.load_engine(sc, sc.K, capi.default_settings(capi.SH1), 0, dev=True); eng.load_scene(sc); eng.init_albedo(); eng.normalize_weights()
    t = {}
    for passes in (16, 48):
        t[passes] = min(eng.debug_time_pcg_solve(passes=passes, reps=8)[0] for _ in range(2))
    out[name] = dict(us_per_pass=round(1e3 * (t[48] - t[16]) / 32, 2), fixed_us=round(1e3 * t[16] - 16e3 * (t[48] - t[16]) / 32, 1))
    eng.close()
print(json.dumps(out))
