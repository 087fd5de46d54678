"""Run-to-run reproducibility ACROSS PROCESSES: the same scene through psgsdf_iterate and psgsdf_optimize in several fresh
processes must give the same bits -- energies after every sub-step, CG iteration counts, number of iterations, final distances / albedo /
poses / light -- for all three shading models.  The reference is one deterministic host thread (PsOptimizer.cpp:303-425,
LedOptimizer.cpp:279-478); the engine sums everything in a fixed order and takes every host decision from validated read-backs
(engine.hip: deliver).  tools/soak.py is the long version of this test (thousands of contexts, every knob)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(model, reps):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "soak.py"), "--worker", "--model", model, "--proc", "0", "--n", "48", "--frames", "6", "--reps", str(reps),
           "--iters", "4", "--modes", "iterate,optimize", "--variants", "default"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    runs = [json.loads(l[5:]) for l in out.stdout.splitlines() if l.startswith("SOAK ")]
    assert len(runs) == 2 * reps and all(r["err"] is None for r in runs), [r["err"] for r in runs]
    return runs


@pytest.mark.parametrize("model", ["SH1", "SH2", "LED"])
def test_same_bits_in_every_process(built, model):
    first = run(model, 2)
    by_mode = {m: [r["cp"] for r in first if r["mode"] == m] for m in ("iterate", "optimize")}
    for m, vecs in by_mode.items():
        assert len(vecs[0]) > 20 and all(v == vecs[0] for v in vecs), (model, m)
    for _ in range(2):          # two more fresh processes
        for r in run(model, 1):
            assert r["cp"] == by_mode[r["mode"]][0], (model, r["mode"])
