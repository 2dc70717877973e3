"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/subgacc.h declares.
No compute call is made here (there is no GPU in the build container)."""
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def L():
    from surel_plus_amd import _lib
    _lib.build()
    return _lib.lib()


def test_header_symbols_are_exported(L):
    from surel_plus_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "subgacc.h")).read()
    declared = set(re.findall(r"\b(subgacc_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"subgacc_status"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name


def test_abi_version_and_argument_errors(L):
    from surel_plus_amd import _lib
    assert L.subgacc_abi_version() == 6           # 6: subgacc_sjoin_fill_v2 (one descriptor for every form of the join); 5: 64-bit key rows (4-hop walks); 2: row / node counts in the join and rng_positions entry points (bounds); 3: hop records in the walk cfg; 4: batched registration of key rows
    assert L.subgacc_key_shift(200, 3) == 8            # SHIFT = 32-clz(M), subg_acc.c:903
    assert L.subgacc_key_shift(100, 4) == 7
    assert L.subgacc_key_shift(200, 8) == _lib.ERR_KEYWIDTH   # 8*8+1 > 64 (subg_acc.c:905-915)
    with pytest.raises(AssertionError, match="hasing key"):
        _lib.check(L.subgacc_key_shift(200, 8))
    assert L.subgacc_key_shift(0, 3) == _lib.ERR_BADARG
    with pytest.raises(TypeError, match="Input parsing error"):
        _lib.check(L.subgacc_key_shift(0, 3))
    assert L.subgacc_scan_workspace_bytes(10) > 0
    assert L.subgacc_uniq_table_bytes(1024) == 1024 * 20


def test_no_cpu_fallback():
    """Without a GPU the product path must fail loudly (never route through the oracle)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    import surel_plus_amd as sp
    with pytest.raises(sp.SubgAccError, match="no HIP device"):
        sp.gset_sampler(np.array([0, 1, 2], np.int32), np.array([1, 0], np.int32), np.arange(2))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "surel_plus_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "oracle/" not in txt, f
