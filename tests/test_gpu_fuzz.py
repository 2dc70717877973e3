"""GPU: randomized parity sweep (tools/fuzz_parity.py) -- graph shape, M, m, bucket, RNG mode, query and batch layout drawn
at random; gset_sampler, the general / fused / strided / lazy SpG paths, walk_sampler and gather against the oracle,
bit for bit.  120 cases here; the tool runs thousands."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_randomized_parity_sweep():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "120", "20261003"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "120 cases, 0 bad" in r.stdout
