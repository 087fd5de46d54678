"""per-pass time of the persistent distance solve under knob settings: (solve of 48 passes - solve of 16 passes) / 32, median and min over repetitions
usage: python tools/pcg_variants.py "PSGSDF_PCG_PIPELINE=0" "PSGSDF_PCG_PREFETCH=0" ...   (each argument: comma-separated KEY=VALUE list; "" = defaults)"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psgradientsdf_amd import capi, synth
sc = synth.make_scene(N=256, F=50, W=640, H=480, model="SH1")
st = capi.default_settings(sc.model_id)
for spec in (sys.argv[1:] or [""]):
    kv = dict(x.split("=") for x in spec.split(",") if x)
    for k, v in kv.items():
        os.environ[k] = v
    eng = capi.load_engine(sc, sc.K, st, 0, dev="PSGSDF_PCG_ABLATE" in kv)      # (the ablations exist in the development build only)
    eng.load_scene(sc); eng.init_albedo(); eng.normalize_weights()
    t16 = [eng.debug_time_pcg_solve(passes=16, reps=4)[0] for _ in range(8)]
    t48 = [eng.debug_time_pcg_solve(passes=48, reps=4)[0] for _ in range(8)]
    recs = eng.iterate(capi.ALL, 6)
    import time
    t0 = time.perf_counter(); eng.iterate(capi.ALL, 40); dt = time.perf_counter() - t0
    print(json.dumps({"knobs": spec or "defaults", "us_per_pass_median": round(1e3 * (np.median(t48) - np.median(t16)) / 32, 3), "us_per_pass_min": round(1e3 * (min(t48) - min(t16)) / 32, 3),
                      "solve16_ms": round(float(np.median(t16)), 4), "it_per_s": round(40 / dt, 1), "cg_iters": [r["cg_iters"] for r in recs], "e_total": [r["e_total"] for r in recs][-2:]}), flush=True)
    eng.close()
    for k in kv:
        del os.environ[k]
