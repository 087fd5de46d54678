"""psgsdf_optimize against the oracle's optimize for the three shading models: per-iteration records and the final state (exploration for
tests/test_parity_gpu.py::test_optimize_matches_oracle)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from psgradientsdf_amd import capi, synth
from oracle import oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
F = int(sys.argv[2]) if len(sys.argv) > 2 else 5
max_it = int(sys.argv[3]) if len(sys.argv) > 3 else 18
damp = float(sys.argv[4]) if len(sys.argv) > 4 else None
for name, mid in (("SH1", capi.SH1), ("SH2", capi.SH2), ("LED", capi.LED)):
    sc = synth.make_scene(N=N, F=F, W=128, H=96, model=name)
    st = capi.default_settings(mid, upsample=1, max_it=max_it, conv_threshold=0.0)
    if mid == capi.LED:
        st.reg_weight_n, st.reg_weight_l, st.damping = 0.1, 5.0, 3.0
    if damp is not None:
        st.damping = damp
    eng = capi.load_engine(sc, sc.K, st, 0); orc = oracle.Oracle(sc, sc.K, st, threads=8)
    for api in (eng, orc):
        api.load_scene(sc); api.init_albedo(); api.normalize_weights()
    (re_, ce), (ro, co) = eng.optimize(capi.ALL), orc.optimize(capi.ALL)
    out = dict(model=name, damping=st.damping, n_eng=len(re_), n_orc=len(ro), conv=(ce, co), upsampled=[a["upsampled"] for a in re_])
    out["flags_equal"] = [(a["converged"], a["diverged"], a["upsampled"]) for a in re_] == [(b["converged"], b["diverged"], b["upsampled"]) for b in ro]
    out["e_total_rel"] = [abs(a["e_total"] - b["e_total"]) / abs(b["e_total"]) for a, b in zip(re_, ro)]
    out["reg_l"] = [(a["reg_weight_l"], b["reg_weight_l"]) for a, b in zip(re_, ro)][-4:]
    out["cg"] = [(a["cg_iters"], b["cg_iters"]) for a, b in zip(re_, ro)]
    same_band = eng.info().n_band == orc.info().n_band and np.array_equal(eng.download_band(), orc.download_band())
    out["same_band"] = bool(same_band)
    if same_band:
        band = eng.download_band(); vs = float(sc.voxel_size) / (2 if any(a["upsampled"] for a in re_) else 1)
        ve, vo = eng.download_volume(), orc.download_volume()
        d = np.abs(ve["dist"][band] - vo["dist"][band]) / vs
        out["dist_over_vs"] = dict(q50=float(np.quantile(d, 0.5)), q999=float(np.quantile(d, 0.999)), max=float(d.max()))
        out["rgb_max"] = float(np.abs(ve["rgb"][:, band] - vo["rgb"][:, band]).max())
        out["pose_max"] = float(np.abs(eng.download_poses() - orc.download_poses()).max())
        le, lo = eng.download_light(), orc.download_light()
        out["light_rel"] = float(np.abs(le - lo).max() / np.abs(lo).max())
    print(json.dumps(out), flush=True)
