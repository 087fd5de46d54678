"""Time per pass of the persistent distance solve for several BUILDS of the engine (forced 16 / 48 passes, psgsdf_debug_time_pcg_solve), alternating:
    python tools/time_solve_libs.py libA.so libB.so ...      -> one JSON line {lib: {per_pass_us: [...], ms16: [...]}}
Used for the timing ablations of profiles/r06_notes.md section 5.7 (a build whose sums are wrong still times correctly: the passes are forced)."""
import os, sys, json
sys.path.insert(0, os.getcwd())
from psgradientsdf_amd import capi, synth
sc = synth.make_scene(N=256, F=50, W=640, H=480, model="SH1")
st = capi.default_settings(sc.model_id)
out = {}
for lib in sys.argv[1:]:
    os.environ["PSGSDF_ENGINE_LIB"] = os.path.abspath(lib)
    import importlib; importlib.reload(capi)
    eng = capi.load_engine(sc, sc.K, st, 0); eng.load_scene(sc); eng.init_albedo(); eng.normalize_weights()
    r = {}
    for rep in range(3):
        for p in (16, 48):
            ms, shape, _ = eng.debug_time_pcg_solve(passes=p, reps=10)
            r.setdefault(p, []).append(ms)
    out[lib] = {"per_pass_us": [round(1e3 * (b - a) / 32, 3) for a, b in zip(r[16], r[48])], "ms16": [round(x, 4) for x in r[16]]}
    eng.close()
print(json.dumps(out))
