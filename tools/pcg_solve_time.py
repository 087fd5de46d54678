"""persistent PCG solve vs the per-pass kernels on the headline band: time per pass, and the two loops side by side (same results?)"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psgradientsdf_amd import capi, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
F = int(sys.argv[2]) if len(sys.argv) > 2 else 50
model = sys.argv[3] if len(sys.argv) > 3 else "SH1"
sc = synth.make_scene(N=N, F=F, W=640, H=480, model=model)
st = capi.default_settings(sc.model_id)
out = {}
res = {}
for persist in ("1", "0"):
    os.environ["PSGSDF_PCG_PERSIST"] = persist
    eng = capi.load_engine(sc, sc.K, st, 0); eng.load_scene(sc); eng.init_albedo(); eng.normalize_weights()
    if persist == "1":
        for passes in (4, 16, 32):
            ms, shape, stamps = eng.debug_time_pcg_solve(passes=passes, reps=5)
            out[f"solve_{passes}_passes_ms"] = ms; out["shape"] = shape
            if passes >= 16:      # stage durations of pass 8 in us: wait, acquire+sum, gather+compute, store drain, release, publish  (first / last workgroup)
                out["stages_us"] = [[round((stamps[o + j + 1] - stamps[o + j]) / 100.0, 2) for j in range(6)] for o in (0, 8)]
        out["us_per_pass_persistent"] = 1e3 * (out["solve_32_passes_ms"] - out["solve_16_passes_ms"]) / 16
        out["per_pass_kernel_ms"] = eng.debug_time_pcg_pass(blocks=0, rows=0, ablate=0, reps=50)
    eng.iterate(capi.ALL, 3)
    t0 = time.perf_counter()
    recs = eng.iterate(capi.ALL, 20)
    dt = time.perf_counter() - t0
    out[f"it_per_s_persist{persist}"] = 20 / dt
    res[persist] = ([r["e_total"] for r in recs], [r["cg_iters"] for r in recs], eng.download_volume()["dist"][eng.download_band()])
    eng.close()
out["e_total_rel_diff"] = float(np.max(np.abs(np.array(res["1"][0]) - np.array(res["0"][0])) / np.abs(res["0"][0])))
out["cg_iters"] = [res["1"][1][:6], res["0"][1][:6]]
out["dist_max_diff_vs"] = float(np.abs(res["1"][2] - res["0"][2]).max() / float(sc.voxel_size))
print(json.dumps(out))
