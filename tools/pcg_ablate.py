#!/usr/bin/env python3
"""Timing of the fused PCG pass (k_cgf_pass) on the bench scene: grid x rows-in-flight sweep, ablations, stage timeline.
ablation bits: 1 no reduction of the previous pass's partials, 4 no partial store, 32 no x update, 64 no record store."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psgradientsdf_amd import capi, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sc = synth.make_scene(N=N, F=50 if N >= 256 else 12, W=640, H=480, model="SH1")
st = capi.default_settings(0)
eng = capi.load_engine(sc, sc.K, st, 0); eng.load_scene(sc); eng.init_albedo(); eng.normalize_weights()
eng.iterate(capi.ALL, 2)
S = eng.info().n_band
nb = (S + 255) // 256
print("band", S, "one-row grid", nb)
rr, rw = eng.debug_rare_rows()
print(f"rows with rare columns: {rr} ({100.0 * rr / S:.1f} %), 64-row groups containing one: {rw} ({100.0 * rw / ((S + 63) // 64):.1f} %)")
for rows in (1, 2):
    g1 = (nb + rows - 1) // rows
    for blocks in sorted({min(768, g1), min(768, (g1 + 1) // 2), 512, 768}):
        ms, stp = eng.debug_time_pcg_pass(blocks, rows, 0, 100, stamps=True)
        print(f"rows {rows} blocks {blocks}: {ms * 1e3:.2f} us   entry spread {(stp[:, 6].max() - stp[:, 6].min()) * 0.01:.2f} us")
for rows, blocks in ((1, (nb + 1) // 2), (2, (nb + 3) // 4)):
    print("rows", rows, "blocks", blocks, " ".join(f"ab{ab}={1e3 * eng.debug_time_pcg_pass(blocks, rows, ab, 100):.2f}us" for ab in (0, 1, 4, 32, 64, 1 | 4, 1 | 4 | 32 | 64)))
    ms, stp = eng.debug_time_pcg_pass(blocks, rows, 0, 20, stamps=True)
    wall = (stp[:, 7] - stp[:, 6]) * 0.01; cyc = stp[:, 4] - stp[:, 0]
    mhz = np.median(cyc / np.maximum(wall, 1e-9))
    print(f"  launch avg {ms * 1e3:.2f} us; shader clock ~{mhz:.0f} ticks/us; first entry -> last exit {(stp[:, 7].max() - stp[:, 6].min()) * 0.01:.2f} us; entry spread {(stp[:, 6].max() - stp[:, 6].min()) * 0.01:.2f} us")
    d = np.diff(stp[:, :5], axis=1) / mhz
    for j, name in enumerate(["issue loads, fold first batch", "wait + reduce previous pass", "fold second batch, rows, stores", "partial store"]):
        print(f"  {name:34s} median {np.median(d[:, j]):6.2f} us   max {d[:, j].max():6.2f} us")
