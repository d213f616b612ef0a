import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")
    _install_abort_trace()


def _install_abort_trace():
    """Native backtrace on SIGABRT / SIGSEGV / SIGBUS into gpurun_out/abort_trace.txt (tests/_abort_trace.c): an abort() inside a
    runtime library is otherwise invisible (captured stderr is lost, faulthandler prints Python frames only).  Best effort."""
    import ctypes
    import subprocess
    import tempfile
    try:
        so = os.path.join(tempfile.gettempdir(), f"jh_abort_trace_{os.getuid()}.so")
        src = os.path.join(ROOT, "tests", "_abort_trace.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", so, src], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        lib = ctypes.CDLL(so)
        lib.abort_trace_install.argtypes = [ctypes.c_char_p]
        lib.abort_trace_install(os.path.join(out, "abort_trace.txt").encode())
        _install_abort_trace.lib = lib  # keep the handler's code mapped
    except Exception:
        pass


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    d = os.path.join(ROOT, "tests", "golden")
    return {"pico": np.load(os.path.join(d, "pico.npz")), "kat": np.load(os.path.join(d, "kat.npz"))}
