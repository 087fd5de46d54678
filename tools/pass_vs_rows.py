#!/usr/bin/env python3
"""us per PCG pass of the persistent solve against the rows a thread carries (grid size): what part of a pass scales with the work of a thread?"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psgradientsdf_amd import capi, synth
out = []
for N in (96, 128, 160, 192, 224, 256):
    sc = synth.make_scene(N=N, F=12, W=320, H=240, model="SH1")
    eng = capi.load_engine(sc, sc.K, capi.default_settings(capi.SH1), 0); eng.load_scene(sc); eng.init_albedo(); eng.normalize_weights()
    t = {}
    for passes in (16, 48):
        ms, shape, _ = eng.debug_time_pcg_solve(passes=passes, reps=8); t[passes] = ms
    S = eng.info().n_band
    out.append(dict(N=N, band=S, workgroups=shape[0], rows_per_workgroup=shape[1], rows_per_thread=-(-shape[1] // 512), us_per_pass=round(1e3 * (t[48] - t[16]) / 32, 2), fixed_us=round(1e3 * t[16] - 16 * 1e3 * (t[48] - t[16]) / 32, 1)))
    eng.close()
print(json.dumps(out))
