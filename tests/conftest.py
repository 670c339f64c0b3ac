import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_lib():
    """Build (if stale) and load the HIP library; GPU tests call the product only through it."""
    import __graft_entry__ as g
    g.build()
    from openvqe_amd import _lib
    return _lib.lib()


@pytest.fixture(scope="session", autouse=True)
def _host_threads_within_the_cpu_quota():
    """numpy's BLAS starts one worker per visible core (256 on the GPU boxes) while the container's CPU quota is 16: the spinning
    workers use the quota up and every thread of the process is throttled for the rest of each 100-ms period (DESIGN.md section 0,
    task 7).  The checker's linear algebra runs on as many threads as the quota allows."""
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        yield
        return
    from openvqe_amd.common_files.host_threads import usable_cpus
    with threadpool_limits(limits=usable_cpus()):
        yield
