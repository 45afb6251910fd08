import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# parity runs of the compiled reference must be single-threaded: its attack loop is order-nondeterministic
# with more OpenMP threads
os.environ["OMP_NUM_THREADS"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
