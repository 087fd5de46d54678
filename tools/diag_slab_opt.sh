#!/bin/bash
# two ranks sharing the GPU through psgsdf_optimize with refinement: print what each rank saw
mkdir -p gpurun_out/diag
P=29688
for r in 0 1; do
  HSA_ENABLE_IPC_MODE_LEGACY=0 python tests/_slab_worker_gpu.py $r 2 $P SH1 /tmp/slabopt 0 24 gloo optimize > gpurun_out/diag/slabopt_$r.log 2>&1 &
done
wait
python - <<'PY' > gpurun_out/diag/slabopt_cmp.log 2>&1
import numpy as np, sys
sys.path.insert(0,'.')
from psgradientsdf_amd import capi, synth
sc = synth.make_scene(N=24, F=6, W=160, H=120, model="SH1")
st = capi.default_settings(capi.SH1, upsample=1, max_it=8, conv_threshold=1e-9)
ref = capi.load_engine(sc, sc.K, st, 0); ref.load_scene(sc)
recs, conv = ref.optimize(capi.ALL)
print("ref", len(recs), [round(r["e_total"],6) for r in recs], [r["upsampled"] for r in recs], [(r["converged"], r["diverged"]) for r in recs], list(ref.info().dim))
for r in range(2):
    g = np.load(f"/tmp/slabopt.rank{r}.npz")
    print("rank", r, len(g["e_total"]), np.round(g["e_total"],6), g["dim"], g["info"], g["ncoll"])
PY
