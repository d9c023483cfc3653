import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the encoder-made recordings behind the realistic_65536 workload are test data: the package reads no files itself
    import numpy as np
    from dcsexplorer_amd import workloads
    workloads.register_recordings(np.load(os.path.join(ROOT, "tests", "golden", "encoder_golden.npz")))


@pytest.fixture(scope="session")
def oracle():
    from oracle.dcs_oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    from oracle.dcs_oracle import Reference, reference_available
    if not reference_available():
        pytest.skip("oracle/_ref/libdcsref.so not built (needs /root/reference; run `make -C oracle ref`)")
    return Reference()


@pytest.fixture(scope="session")
def dcs():
    import dcsexplorer_amd
    dcsexplorer_amd.load_library()
    return dcsexplorer_amd


@pytest.fixture(scope="session")
def gpu_ctx(dcs):
    """a DcsCtx on GPU 0; the test FAILS (not skips) if the HIP path is unavailable"""
    ctx = dcs.Context(0)
    yield ctx
    ctx.close()
