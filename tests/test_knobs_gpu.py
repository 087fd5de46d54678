"""The engine's latency tricks must not change results: PCG stop test by watching the mapped mailbox vs draining the stream
(PSGSDF_PCG_POLL), scalar folds done by the next kernel vs by a kernel of their own (PSGSDF_FOLD_IN_NEXT), albedo update applied by the sweep vs by its own kernel (PSGSDF_FUSE_ALBEDO), launch shape of the
fused PCG pass (PSGSDF_PCG_ROWS / PSGSDF_PCG_BLOCKS), the distance step as one persistent kernel (assembly + solve + update) vs its parts
(PSGSDF_PCG_FUSE_APPLY / PSGSDF_PCG_FUSE_ASM / PSGSDF_PCG_PERSIST).  Each variant runs in its own process (the knobs are read at create time)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, sys
sys.path.insert(0, %r)
import numpy as np
from psgradientsdf_amd import capi, synth
sc = synth.make_scene(N=64, F=8, W=160, H=120, model=sys.argv[1])
import os
st = capi.default_settings(sc.model_id, reg_weight_rho=float(os.environ.get('PSGSDF_TEST_REG_RHO', '0')))
eng = capi.load_engine(sc, sc.K, st, 0); eng.load_scene(sc); eng.init_albedo(); eng.normalize_weights()
recs = eng.iterate(capi.ALL, 3)
recs2, conv = eng.optimize(capi.ALL) if len(sys.argv) > 2 else ([], False)
v = eng.download_volume(); band = eng.download_band()
print(json.dumps(dict(e=[r["e_total"] for r in recs], after=[r["e_after"] for r in recs], cg=[r["cg_iters"] for r in recs],
                      n2=len(recs2), e2=[r["e_total"] for r in recs2],
                      dsum=float(np.abs(v["dist"][band]).astype(np.float64).sum()), rsum=float(v["rgb"][:, band].astype(np.float64).sum()),
                      psum=float(np.abs(eng.download_poses()).astype(np.float64).sum()))))
""" % ROOT


def run(model, env, full=False):
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, "-c", WORKER, model] + (["full"] if full else []), capture_output=True, text=True, env=e, timeout=200)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("model", ["SH1", "LED"])
def test_host_side_knobs_are_bitwise_neutral(built, model):
    ref = run(model, {}, full=True)
    for env in ({"PSGSDF_PCG_POLL": "0"}, {"PSGSDF_FOLD_IN_NEXT": "0"}, {"PSGSDF_PCG_POLL": "0", "PSGSDF_FOLD_IN_NEXT": "0"},
                {"PSGSDF_FUSE_ALBEDO": "0"}, {"PSGSDF_SPECULATE": "0"}, {"PSGSDF_SPECULATE": "0", "PSGSDF_FUSE_ALBEDO": "0"},
                {"PSGSDF_XCD_MAP": "0"}, {"PSGSDF_XCD_MAP": "3"}, {"PSGSDF_XCD_MAP": "99"}, {"PSGSDF_XCD_MAP": "7"}, {"PSGSDF_XCD_MAP": "7", "PSGSDF_XCD_STRIPE": "4"},   # which workgroup does which rows: logical ids carry rows AND partial-sum slots
                {"PSGSDF_FM_SOLVE": "2"}, {"PSGSDF_FM_SOLVE": "2", "PSGSDF_SPECULATE": "0"},      # 2: only the LED light vector keeps its solve kernel
                {"PSGSDF_FM_SOLVE": "0"}, {"PSGSDF_FM_SOLVE": "0", "PSGSDF_SPECULATE": "0"}):      # FM_SOLVE=0: the per-frame light / pose solves as kernels of their own instead of in the sweeps' last workgroups
        got = run(model, env, full=True)
        assert got == ref, (env, got, ref)


@pytest.mark.parametrize("model", ["SH1", "LED"])
def test_regularised_albedo_solve_device_driven_equals_host_driven(built, model):
    """`reg albedo` != 0: the CG over the 3S albedo unknowns driven by the device (round 6: chunks of iterations, alpha / beta / the stop test from the kernels'
    own partial sums) against the host-driven loop (two read-backs per iteration; what multi-rank contexts run): the same sums in the same order, the same
    float recurrences -- every energy, iteration count and the optimised state to the bit, through psgsdf_optimize too."""
    ref = run(model, {"PSGSDF_TEST_REG_RHO": "0.02", "PSGSDF_AREG_DEVICE": "0"}, full=True)
    got = run(model, {"PSGSDF_TEST_REG_RHO": "0.02"}, full=True)
    assert got == ref, (got, ref)
    plain = run(model, {}, full=True)
    assert plain["e"] != ref["e"]      # (the term is on)


def test_per_pass_solve_launch_shapes(built):
    """The per-pass distance solve (bands beyond the persistent kernel's cap, fall-backs) over several trips of a small grid, on XCD-contiguous logical
    workgroup ids (round 6, PSGSDF_XCD_MAP bit 7) and on physical ones: every shape covers every row once -- only the order of the partial sums differs."""
    ref = run("SH1", {"PSGSDF_PCG_PERSIST": "0"})
    for env in ({"PSGSDF_PCG_BLOCKS": "7"}, {"PSGSDF_PCG_BLOCKS": "24"}, {"PSGSDF_PCG_BLOCKS": "24", "PSGSDF_XCD_MAP": "35"}, {"PSGSDF_PCG_BLOCKS": "61"}):
        got = run("SH1", dict(env, PSGSDF_PCG_PERSIST="0"))
        assert all(abs(a - b) <= 1 for a, b in zip(got["cg"], ref["cg"])), (env, got["cg"], ref["cg"])
        assert all(abs(a - b) <= 2e-5 * abs(b) for a, b in zip(got["e"], ref["e"])), (env, got["e"], ref["e"])
        assert abs(got["dsum"] - ref["dsum"]) <= 1e-5 * ref["dsum"]


def test_pcg_launch_shape_only_changes_rounding(built):
    ref = run("SH1", {})
    for env in ({"PSGSDF_PCG_ROWS": "2", "PSGSDF_PCG_BLOCKS": "7"}, {"PSGSDF_PCG_ROWS": "1", "PSGSDF_PCG_BLOCKS": "5"}, {"PSGSDF_PCG_ROWS": "2", "PSGSDF_PCG_BLOCKS": "512"},
                {"PSGSDF_FM_ROWS": "16"}, {"PSGSDF_FM_ROWS": "5"}, {"PSGSDF_FUSE_PCG_INIT": "0"}):   # observations per thread of the frame-major sweeps: summation order only
        got = run("SH1", env)
        assert all(abs(a - b) <= 1 for a, b in zip(got["cg"], ref["cg"]))
        assert all(abs(a - b) <= 2e-5 * abs(b) for a, b in zip(got["e"], ref["e"])), (env, got["e"], ref["e"])
        assert abs(got["dsum"] - ref["dsum"]) <= 1e-5 * ref["dsum"]


@pytest.mark.parametrize("model", ["SH1", "LED"])
def test_distance_step_fusions_are_bitwise_neutral(built, model):
    """The persistent solve kernel that assembles its own rows (register accumulation in assemble_row's order) and applies the distance update
    in its epilogue must give the bits of the same kernel with k_apply_dist behind it, with every value through memory instead of the XCD's L2, and
    with the sums fetched after instead of behind the gathers -- energies, iteration counts and the optimised state, through psgsdf_optimize too.
    The CLASSIC recurrences (round 2's persistent kernel, the same with k_assemble in front, the per-pass kernels that multi-rank fallbacks and
    512^3 bands run) agree with each other to the bit as well; between the two families -- pipelined recurrences in double vs Eigen's in float --
    only rounding differs: same iteration counts (+-1), energies to 2e-5."""
    ref = run(model, {}, full=True)
    for env in ({"PSGSDF_PCG_FUSE_APPLY": "0"}, {"PSGSDF_PCG_XCD_LOCAL": "0"}, {"PSGSDF_PCG_PREFETCH": "0"}):
        got = run(model, env, full=True)
        assert got == ref, (env, got, ref)
    # round 5: the exchanged values carry their own tags in four mantissa bits (2^-48) instead of being ordered behind flags -- same pass counts, the rest to rounding
    untagged = run(model, {"PSGSDF_PCG_TAGM": "0"}, full=True)
    assert untagged["cg"] == ref["cg"] and untagged["n2"] == ref["n2"], (untagged["cg"], ref["cg"])
    assert all(abs(a - b) <= 1e-7 * abs(b) for a, b in zip(untagged["e"] + untagged["e2"], ref["e"] + ref["e2"])), (untagged["e"], ref["e"])
    assert abs(untagged["dsum"] - ref["dsum"]) <= 1e-7 * ref["dsum"] and abs(untagged["psum"] - ref["psum"]) <= 1e-7 * ref["psum"]
    classic = run(model, {"PSGSDF_PCG_PIPELINE": "0"}, full=True)
    for env in ({"PSGSDF_PCG_FUSE_ASM": "0"}, {"PSGSDF_PCG_PERSIST": "0"}, {"PSGSDF_PCG_PIPELINE": "0", "PSGSDF_PCG_XCD_LOCAL": "0"},
                {"PSGSDF_PCG_PERSIST": "0", "PSGSDF_XCD_MAP": "35"}):      # (round 6: the per-pass kernel's XCD-contiguous row blocks, bit 7 of the map, switched off)
        got = run(model, env, full=True)
        assert got == classic, (env, got, classic)
    assert all(abs(a - b) <= 1 for a, b in zip(classic["cg"], ref["cg"])) and classic["n2"] == ref["n2"]
    assert all(abs(a - b) <= 2e-5 * abs(b) for a, b in zip(classic["e"] + classic["e2"], ref["e"] + ref["e2"])), (classic["e"], ref["e"])
    assert abs(classic["dsum"] - ref["dsum"]) <= 1e-5 * ref["dsum"] and abs(classic["psum"] - ref["psum"]) <= 1e-6 * ref["psum"]
