"""CPU-side checks of the drop-in boundary: the HIP library builds for gfx950, loads without a GPU and
exports every symbol include/psgsdf.h declares; argument errors come back as status codes."""
import ctypes

import pytest

from psgradientsdf_amd import capi


def test_engine_exports_every_declared_symbol(built):
    import __graft_entry__ as g
    lib = ctypes.CDLL(capi.ENGINE_LIB)
    names = g._declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
    lib.psgsdf_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.psgsdf_version()


def test_oracle_mirrors_the_abi(built):
    import __graft_entry__ as g
    from oracle import oracle
    lib = oracle.lib()
    # (the oracle's multi-rank mirror is driven through its own phase API by tests/_slab_runner.py: no communicator entry points)
    skip = {"psgsdf_comm_unique_id", "psgsdf_comm_init_ext", "psgsdf_comm_stats", "psgsdf_kernel_times", "psgsdf_reset_kernel_times",
            "psgsdf_set_profiling", "psgsdf_watch_kernel", "psgsdf_debug_time_pcg_pass", "psgsdf_debug_time_pcg_solve", "psgsdf_debug_rare_rows", "psgsdf_debug_sync_stats",
            "psgsdf_slab_plane_count", "psgsdf_plan_slab", "psgsdf_upload_volume_slab", "psgsdf_set_record_observer", "psgsdf_set_on_iter_period"}
    for n in g._declared_symbols():
        if n in skip:
            continue
        assert hasattr(lib, n.replace("psgsdf_", "orc_")), n


def test_comm_entry_points_fail_cleanly_without_a_gpu(built):
    """the communicator entry points are status codes too: no context -> ARG; a NULL id -> ARG; librccl is only bound when asked for"""
    lib = ctypes.CDLL(capi.ENGINE_LIB)
    for f in ("psgsdf_comm_init", "psgsdf_comm_init_ext", "psgsdf_comm_unique_id", "psgsdf_comm_stats"):
        getattr(lib, f).restype = ctypes.c_int
    assert lib.psgsdf_comm_init(None, None, 0, 1) == -1
    assert lib.psgsdf_comm_init_ext(None, None, 0, 1) == -1
    assert lib.psgsdf_comm_unique_id(None) == -1
    assert lib.psgsdf_comm_stats(None, None) == -1
    import subprocess, sys
    code = ("import ctypes; lib = ctypes.CDLL(%r); lib.psgsdf_comm_init(None, None, 0, 1); "
            "print('librccl' in open('/proc/self/maps').read())" % capi.ENGINE_LIB)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert out.stdout.strip() == "False", out.stdout + out.stderr      # loading the engine pulls no RCCL into a single-GPU process


def test_bad_arguments_are_status_codes(built):
    lib = ctypes.CDLL(capi.ENGINE_LIB)
    lib.psgsdf_create.restype = ctypes.c_int
    assert lib.psgsdf_create(None, None, None, 0, None) == -1
    st = capi.default_settings(capi.SH1); st.model = 7
    g = capi.GridDesc(); g.dim[:] = [8, 8, 8]; g.voxel_size = 0.01
    K = (ctypes.c_float * 9)(1, 0, 0, 0, 1, 0, 0, 0, 1)
    ctx = ctypes.c_void_p()
    assert lib.psgsdf_create(ctypes.byref(g), K, ctypes.byref(st), 0, ctypes.byref(ctx)) == -1   # unknown model


def test_no_cpu_fallback_without_library(monkeypatch, tmp_path):
    monkeypatch.setattr(capi, "ENGINE_LIB", str(tmp_path / "missing.so"))
    monkeypatch.setattr(capi, "_engine_lib", None)
    with pytest.raises(capi.PsgsdfError):
        capi.engine_lib()
