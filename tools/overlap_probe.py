#!/usr/bin/env python3
"""the most an early start of the distance solve (under the distance sweep's tail) could win, measured (psgsdf_debug_overlap_probe).
    python tools/overlap_probe.py [--grid 256] [--frames 50] [--reps 20]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psgradientsdf_amd import capi, synth

ap = argparse.ArgumentParser(); ap.add_argument("--grid", type=int, default=256); ap.add_argument("--frames", type=int, default=50); ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
sc = synth.make_scene(N=a.grid, F=a.frames, W=640, H=480, model="SH1")
eng = capi.load_engine(sc, sc.K, capi.default_settings(capi.SH1), 0); eng.load_scene(sc); eng.init_albedo(); eng.normalize_weights()
eng.iterate(capi.ALL, 3)
rows = []
for rnd in range(3):
    t = eng.debug_overlap_probe(a.reps)
    rows.append(dict(sweep_alone_us=round(1e3 * t[0], 2), solve_alone_us=round(1e3 * t[1], 2), back_to_back_us=round(1e3 * t[2], 2), side_by_side_us=round(1e3 * t[3], 2),
                     most_an_overlap_could_hide_us=round(1e3 * (t[2] - t[3]), 2)))
it = eng.iterate(capi.ALL, 1)
print(json.dumps({"scene": f"{a.grid}^3 x {a.frames}", "rounds": rows, "iteration_us_for_scale": 378}))
