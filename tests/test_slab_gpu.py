"""The engine's multi-rank phase API on real hardware: (a) one rank under the nccl (=RCCL) backend -- device-pointer
aliasing, torch-stream sharing, collectives on engine memory; (b) two ranks sharing the GPU under gloo -- real halo
exchanges and all-reduces between two engine contexts.  Both must reproduce the plain single-context engine."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from psgradientsdf_amd import capi, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("model,world,backend", [("SH1", 1, "nccl"), ("SH1", 2, "gloo"), ("LED", 2, "gloo")])
def test_engine_slab_ranks_match_single_context(built, tmp_path, model, world, backend):
    N, n_iters = 40, 2
    port = free_port(); out = str(tmp_path / "slab")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_slab_worker_gpu.py"), str(r), str(world), str(port), model, out, str(n_iters), str(N), backend],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    try:
        for p in procs:
            o, _ = p.communicate(timeout=150)
            assert p.returncode == 0, o[-3000:]
    finally:
        for p in procs:      # a rank that is still alive here would keep the GPU (and the next test) busy
            if p.poll() is None:
                p.kill()
    sc = synth.make_scene(N=N, F=6, W=160, H=120, model=model)
    st = capi.default_settings(sc.model_id)
    ref = capi.load_engine(sc, sc.K, st, 0); ref.load_scene(sc)
    ref.init_albedo(); e0 = ref.normalize_weights()
    recs = ref.iterate(capi.ALL, n_iters)
    band = ref.download_band(); v = ref.download_volume(); vs = float(sc.voxel_size)
    for r in range(world):
        res = np.load(out + f".rank{r}.npz")
        assert abs(float(res["e0"]) - e0) <= 1e-6 * abs(e0)
        assert np.allclose(res["e_total"], [x["e_total"] for x in recs], rtol=5e-6)
        assert np.all(np.abs(res["cg"] - np.array([x["cg_iters"] for x in recs])) <= 1)
        # the slab phases run the two-kernel PCG, the single context the fused one: same recurrences, different dot-product rounding
        assert np.abs(res["dist"][band] - v["dist"][band]).max() <= 1e-4 * vs
        assert np.abs(res["rgb"][:, band] - v["rgb"][:, band]).max() <= 2e-4   # LED: one global light system, all-reduce order
        assert np.abs(res["poses"] - ref.download_poses()).max() <= 1e-6
        if world > 1:
            assert res["ncoll"] > 20 and 0 < res["info"][2] <= (len(band) + world - 1) // world
