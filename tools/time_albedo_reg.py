"""Cost of the `reg albedo` term (no shipped configuration enables it): Gauss-Newton iterations/s of the headline scene with and without it, and the
regularised albedo solve's CG iteration count:   python tools/time_albedo_reg.py [grid] [frames]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psgradientsdf_amd import capi, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
F = int(sys.argv[2]) if len(sys.argv) > 2 else 50
sc = synth.make_scene(N=N, F=F, W=640, H=480, model="SH1")
out = {}
for w in (0.0, 0.02):
    st = capi.default_settings(sc.model_id, reg_weight_rho=w)
    eng = capi.load_engine(sc, sc.K, st, 0); eng.load_scene(sc); eng.init_albedo(); eng.normalize_weights()
    eng.iterate(capi.ALL, 2)
    t0 = time.perf_counter(); recs = eng.iterate(capi.ALL, 6); dt = time.perf_counter() - t0
    s = eng.step(capi.ALBEDO)
    out[str(w)] = {"it_per_s": round(6 / dt, 1), "ms_per_it": round(1e3 * dt / 6, 3), "albedo_cg_iters": s.get("cg_iters"), "e_total": recs[-1]["e_total"]}
    eng.close()
print(json.dumps(out))
