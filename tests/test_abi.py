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
    skip = {"psgsdf_comm_init", "psgsdf_comm_unique_id", "psgsdf_kernel_times", "psgsdf_reset_kernel_times",
            "psgsdf_set_profiling", "psgsdf_watch_kernel", "psgsdf_debug_time_pcg_pass", "psgsdf_debug_rare_rows"}
    for n in g._declared_symbols():
        if n in skip:
            continue
        assert hasattr(lib, n.replace("psgsdf_", "orc_")), n


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
