#!/usr/bin/env python3
"""Per-iteration wall time of psgsdf_optimize (the product loop: stop decision every iteration) next to psgsdf_iterate
(no early exit) on the bench scene."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psgradientsdf_amd import capi, synth

sc = synth.make_scene(N=256, F=50, W=640, H=480, model="SH1")
for mode in ("iterate", "optimize"):
    st = capi.default_settings(0)
    st.max_it = 40; st.conv_threshold = 0.0; st.upsample = 0
    eng = capi.load_engine(sc, sc.K, st, 0); eng.load_scene(sc)
    if mode == "iterate":
        eng.init_albedo(); eng.normalize_weights()
        eng.iterate(capi.ALL, 3)
        t = time.perf_counter()
        recs = eng.iterate(capi.ALL, 40)
    else:
        t = time.perf_counter()
        recs, conv = eng.optimize(capi.ALL)      # includes init_albedo + weight normalisation (PsOptimizer.cpp:279-301)
    t = time.perf_counter() - t
    print([r["converged"] + 2 * r["diverged"] for r in recs][-5:])
    print(f"{mode}: {len(recs)} iterations, {1e3 * t / len(recs):.3f} ms/iteration, last E {recs[-1]['e_total']:.6f}")
    eng.close()
