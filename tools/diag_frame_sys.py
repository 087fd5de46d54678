"""diagnostic: per-frame relative error of the light / pose normal equations, engine vs oracle"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psgradientsdf_amd import capi, synth
from oracle import oracle
for (N, F, model, W, H) in [(128, 100, "SH2", 640, 480), (128, 100, "SH1", 640, 480), (256, 50, "SH1", 640, 480), (128, 100, "SH1", 320, 240)]:
    sc = synth.make_scene(N=N, F=F, W=W, H=H, model=model)
    st = capi.default_settings(sc.model_id)
    eng = capi.load_engine(sc, sc.K, st, 0); orc = oracle.Oracle(sc, sc.K, st, threads=64)
    for api in (eng, orc):
        api.load_scene(sc); api.init_albedo(); api.normalize_weights()
    for blk in (capi.LIGHT, capi.POSE):
        He, be = eng.debug_frame_system(blk); Ho, bo = orc.debug_frame_system(blk)
        relH = np.abs(He - Ho).reshape(len(He), -1).max(1) / np.abs(Ho).reshape(len(Ho), -1).max(1)
        relb = np.abs(be - bo).max(1) / np.abs(bo).max(1)
        print(N, F, model, W, "blk", blk, "H: max %.2e med %.2e argmax %d | b: max %.2e med %.2e" % (relH.max(), np.median(relH), relH.argmax(), relb.max(), np.median(relb)), "global", np.abs(He - Ho).max() / np.abs(Ho).max(), flush=True)
        f = int(relH.argmax())
        if blk == capi.POSE:
            print("  frame", f, "diag eng", np.diag(He[f]), "\n  diag orc", np.diag(Ho[f]))
