"""does a library variant (stripped device symbol tables / compressed code object) load and run?  time to the first handle.
Usage: python tools/exp_lib_variants.py <path.so>  (run one variant per process)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["OVQE_LIB"] = os.path.abspath(sys.argv[1])
t0 = time.perf_counter()
import numpy as np  # noqa: E402
from openvqe_amd.backend import Statevector  # noqa: E402
t1 = time.perf_counter()
sv = Statevector(14)
t2 = time.perf_counter()
import __graft_entry__ as g  # noqa: E402
g.smoke()
t3 = time.perf_counter()
# a path with rocprim / hipcub kernels and the sector tables: the test_encodings case that crashed the stripped build
import subprocess  # noqa: E402
r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_encodings.py", "tests/test_gpu_sector.py", "-x", "-q", "-m", "gpu"], cwd=ROOT,
                   capture_output=True, text=True, env=dict(os.environ))
print({"lib": sys.argv[1], "size": os.path.getsize(sys.argv[1]), "import_s": t1 - t0, "first_handle_s": t2 - t1, "smoke_s": t3 - t2,
       "pytest_rc": r.returncode, "pytest_tail": r.stdout.strip().splitlines()[-1:]})
