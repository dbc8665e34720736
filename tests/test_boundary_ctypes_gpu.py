"""The drop-in boundary bound the way a maintainer of the reference would bind it
(INTEGRATION.md section 1): a child process with numpy + ctypes only -- no torch -- on the null
stream, device memory from bsc_malloc / bsc_h2d / bsc_d2h / bsc_memset / bsc_free.  Replaces the
Theano protocol of bayesic/algebra.py:42-58 for config 1 (exact Normal-Gamma posterior) and for
dot(X.T, X)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.gpu


def test_integration_stub_without_torch():
    env = dict(os.environ)
    env.pop("PYTHONSTARTUP", None)
    proc = subprocess.run([sys.executable, os.path.join(HERE, "_ctypes_only_binding.py")], env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    log = proc.stdout.decode()
    assert proc.returncode == 0, log
    for marker in ("ok config1", "ok gram", "ok allreduce", "ok torch-free"):
        assert marker in log, log
