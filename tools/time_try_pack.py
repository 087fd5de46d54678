"""k_try_pack_f32 (psgsdf_set_keyframes' check whether float keyframes are 8-bit data) timed on the headline image stack: run under rocprofv3 --kernel-trace --stats
   python tools/time_try_pack.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psgradientsdf_amd import capi, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for u8 in (False, True):
    sc = synth.make_scene(N=64, F=50, W=640, H=480, model="SH1", u8=u8)
    for r in range(reps):
        eng = capi.load_engine(sc, sc.K, capi.default_settings(capi.SH1), 0)
        t0 = time.perf_counter(); eng.load_scene(sc, u8=False); dt = time.perf_counter() - t0
        print("8-bit data as floats" if u8 else "rendered floats", "load_scene %.1f ms" % (1e3 * dt), "compacted", eng.debug_sync_stats()["keyframes_compacted"], flush=True)
        eng.close()
