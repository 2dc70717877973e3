"""GPU: a plain C program (no Python, no torch in the process) drives the C ABI end to end -- the drop-in boundary
is a real C-ABI shared library, not a torch extension."""
import os
import subprocess

import pytest

from conftest import ROOT
from gpu_helpers import sp  # noqa: F401  (fixture)

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


def test_integration_md_ctypes_stub_runs_as_printed(sp, monkeypatch):
    """INTEGRATION.md section 3 shows the binding a maintainer of the reference would write (ctypes over the C ABI, no surel_plus_amd
    import): the first code block is executed AS PRINTED and its sjoin() joined against the reference's own golden (train.py:13-45)."""
    import re
    import numpy as np
    import torch
    from conftest import GOLDEN
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text[text.index("## 3. Calling the C ABI directly"):], flags=re.S)
    stub = blocks[0]
    assert "def sjoin(" in stub and "subgacc_sjoin_fill_v2" in stub
    monkeypatch.chdir(root)          # the stub opens "surel_plus_amd/libsubgacc_hip.so"
    ns = {}
    exec(compile(stub, "INTEGRATION.md#3", "exec"), ns)
    g = np.load(os.path.join(GOLDEN, "sjoin_int.npz"))
    indptr = torch.from_numpy(g["z_indptr"].astype(np.int64)).cuda()
    indices = torch.from_numpy(g["z_indices"].astype(np.int32)).cuda()
    data = torch.from_numpy(g["z_data"].astype(np.int32)).cuda()
    table = torch.from_numpy(g["encode"].astype(np.float32)).cuda().contiguous()
    edge = torch.from_numpy(g["edge"].astype(np.int64)).cuda().contiguous()
    max_len = int(np.diff(g["z_indptr"]).max())
    xz, seg = ns["sjoin"](indptr, indices, data, edge, table, max_len)
    torch.cuda.synchronize()
    assert np.array_equal(xz.cpu().numpy(), g["xz_ptr1"])
    assert np.array_equal(seg.cpu().numpy(), g["ind_ptr1"])
