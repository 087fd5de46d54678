import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psgradientsdf_amd import capi, synth
from oracle import oracle
for model, N, F in (("SH1", 96, 20), ("LED", 64, 12), ("SH2", 64, 12)):
    sc = synth.make_scene(N=N, F=F, W=320, H=240, model=model)
    st = capi.default_settings(sc.model_id)
    if model == "LED":
        st.reg_weight_n, st.reg_weight_l, st.damping = 0.1, 5.0, 3.0
    eng = capi.load_engine(sc, sc.K, st, 0); orc = oracle.Oracle(sc, sc.K, st, threads=16)
    for api in (eng, orc):
        api.load_scene(sc); api.init_albedo(); api.normalize_weights()
    band = eng.download_band(); vs = float(sc.voxel_size)
    t = time.time()
    for it in range(1, 13):
        re_, ro = eng.iterate(capi.ALL, 1)[0], orc.iterate(capi.ALL, 1)[0]
        ve, vo = eng.download_volume(), orc.download_volume()
        d = np.abs(ve["dist"][band] - vo["dist"][band]) / vs
        rel = np.linalg.norm((ve["dist"][band] - vo["dist"][band]).astype(np.float64)) / np.linalg.norm(vo["dist"][band].astype(np.float64))
        if it in (1, 2, 4, 8, 12):
            print(f"{model} N={N} it {it:2d}: E_total {re_['e_total']:.6f} vs {ro['e_total']:.6f}  rel-L2 {rel:.2e}  max|dd|/vs {d.max():.2e}  n(>1e-4) {int((d > 1e-4).sum())}/{len(d)}  p99.9 {np.quantile(d, 0.999):.2e}  albedo {np.abs(ve['rgb'][:, band] - vo['rgb'][:, band]).max():.2e}  pose {np.abs(eng.download_poses() - orc.download_poses()).max():.2e}  cg {re_['cg_iters']}/{ro['cg_iters']}", flush=True)
    print("  oracle+engine time %.1f s" % (time.time() - t))
