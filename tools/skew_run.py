import os, sys
sys.path.insert(0, "/root/repo")
from psgradientsdf_amd import capi, synth
sc = synth.make_scene(N=256, F=50, W=640, H=480, model="SH1")
st = capi.default_settings(sc.model_id)
os.environ["PSGSDF_SOLVE_DUMP"] = "gpurun_out/solve_dump.txt"
eng = capi.load_engine(sc, sc.K, st, 0); eng.load_scene(sc); eng.init_albedo(); eng.normalize_weights()
print(eng.debug_time_pcg_solve(passes=16, reps=5)[:2])
