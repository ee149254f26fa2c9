import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")

# Load (and, if the C source is newer, rebuild) the oracle library NOW, while this process has not touched the
# GPU: c_oracle builds through a `make` child process, and a process that has initialised the GPU must not
# start other programs on the GPU pool.
from oracle import c_oracle as _corc  # noqa: E402
_corc.lib()


# The plain-C client of include/hsd.h is compiled here for the same reason (tests/test_gpu_forms.py runs it).
def _build_c_client():
    import subprocess
    import tempfile
    exe = os.path.join(tempfile.mkdtemp(prefix="hsd_abi_"), "abi_client")
    libdir = os.path.join(REPO, "hsd_amd")
    if not os.path.exists(os.path.join(libdir, "libhsd_amd.so")):
        return None
    try:
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"),
                               os.path.join(REPO, "tests", "c", "abi_client.c"), "-o", exe, "-L", libdir, "-lhsd_amd",
                               "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    except (OSError, subprocess.CalledProcessError):
        return None
    return exe


_C_CLIENT = _build_c_client()


@pytest.fixture(scope="session")
def c_client():
    if _C_CLIENT is None:
        pytest.fail("tests/c/abi_client.c did not build (gcc + hsd_amd/libhsd_amd.so are required)")
    return _C_CLIENT


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_graph(name):
    from hsd_amd.graph import read_edgelist, Graph
    g = read_edgelist(os.path.join(GOLDEN, "graphs", f"{name}.edgelist"))
    if name == "robust":      # tests/robust_test/main.py:50 adds edge (5, 6)
        i, j = g.node2idx["5"], g.node2idx["6"]
        rows = np.repeat(np.arange(g.n_node), np.diff(g.rowptr))
        src = np.concatenate([rows, [i]])
        dst = np.concatenate([g.col.astype(np.int64), [j]])
        g = Graph.from_edges(g.nodes, src, dst)
    return g


@pytest.fixture(scope="session")
def engine():
    from hsd_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()
