"""The C ABI driven by a plain C++/HIP host program (examples/c_abi_demo.cpp): no torch in the process."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_demo_runs_without_torch(gpu):
    exe = os.path.join(ROOT, "examples", "c_abi_demo")
    assert os.path.exists(exe), "examples/c_abi_demo not built (make -C isplib_amd/csrc)"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c_abi_demo ok" in r.stdout
