"""GPU: a plain C program (no Python, no torch in the process) drives the C ABI end to end -- the drop-in boundary
is a real C-ABI shared library, not a torch extension."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_c_client_runs_the_whole_path(tmp_path):
    from surel_plus_amd import _lib
    exe = str(tmp_path / "cabi_smoke")
    src = os.path.join(ROOT, "tests", "cabi", "cabi_smoke.c")
    subprocess.check_call(["gcc", "-O1", "-std=gnu11", src, "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                           "-D__HIP_PLATFORM_AMD__", "-L/opt/rocm/lib", "-lamdhip64", "-ldl", "-Wl,-rpath,/opt/rocm/lib",
                           "-o", exe])
    out = subprocess.run([exe, _lib.LIB_PATH], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "cabi_smoke ok" in out.stdout
