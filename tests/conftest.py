import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# every rendezvous of the multi-process tests stays on the loopback interface (the container hostname may not resolve)
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

# Collection order of the GPU suite: oracle-parity tests first, then the product rows in SURVEY 8's order (configs,
# full size, front end, the C++ host mirror, the voxelPS driver, the z-slab engine, the bench line); tests that compare the engine with
# ITSELF (knobs, run-to-run reproducibility) come last, so that `pytest -x` can never hide a product row behind a self-comparison.
# Files not listed keep pytest's alphabetical order in between.
_FIRST = ["test_parity_gpu.py", "test_golden.py", "test_edge_gpu.py", "test_configs_gpu.py", "test_fullsize_gpu.py", "test_integrate.py", "test_frontend.py",
          "test_host_mirror_gpu.py", "test_voxelps_gpu.py", "test_comm_gpu.py", "test_slab_gpu.py", "test_bench_gpu.py"]
_LAST = ["test_repro_gpu.py", "test_knobs_gpu.py"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        f = os.path.basename(str(item.fspath))
        if f in _FIRST:
            return (0, _FIRST.index(f))
        if f in _LAST:
            return (2, _LAST.index(f))
        return (1, 0)
    items.sort(key=key)     # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def built():
    import __graft_entry__ as g
    g.build()
    return True
