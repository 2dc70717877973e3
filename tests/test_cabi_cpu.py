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
    assert L.subgacc_abi_version() == 7           # 7: headed rows, the per-form join entry points of ABI 1-5 gone; 6: subgacc_sjoin_fill_v2 (one descriptor for every form of the join); 5: 64-bit key rows (4-hop walks); 2: row / node counts in the join and rng_positions entry points (bounds); 3: hop records in the walk cfg; 4: batched registration of key rows
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


def test_join_descriptor_layout_matches_the_header(tmp_path, L):
    """struct subgacc_join_desc (ABI 6) as gcc lays it out from include/subgacc.h == the ctypes mirror, field by field: a binding
    that disagrees about one offset would hand the kernels a wrong pointer.  And the entry point refuses a descriptor of another
    size or with neither / both row layouts before anything is launched (no GPU needed for that)."""
    import ctypes as C
    import subprocess
    from surel_plus_amd import _lib
    names = [f[0] for f in _lib.JoinDesc._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "subgacc.h"\nint main(void) {\n'
                   '  printf("%zu\\n", sizeof(subgacc_join_desc));\n' +
                   "".join(f'  printf("%zu\\n", offsetof(subgacc_join_desc, {n}));\n' for n in names) + "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", str(src), "-I" + os.path.join(ROOT, "include"), "-o", str(exe)])
    out = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert out[0] == C.sizeof(_lib.JoinDesc)
    assert out[1:] == [getattr(_lib.JoinDesc, n).offset for n in names]
    # ... and struct subgacc_walk_cfg (row_pitch joined it this round)
    wnames = [f[0] for f in _lib.WalkCfg._fields_]
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "subgacc.h"\nint main(void) {\n'
                   '  printf("%zu\\n", sizeof(subgacc_walk_cfg));\n' +
                   "".join(f'  printf("%zu\\n", offsetof(subgacc_walk_cfg, {n}));\n' for n in wnames) + "  return 0;\n}\n")
    subprocess.check_call(["gcc", "-std=c11", str(src), "-I" + os.path.join(ROOT, "include"), "-o", str(exe)])
    wout = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert wout[0] == C.sizeof(_lib.WalkCfg) and wout[1:] == [getattr(_lib.WalkCfg, n).offset for n in wnames]
    d = _lib.JoinDesc()
    d.struct_bytes = 8
    assert L.subgacc_sjoin_fill_v2(C.byref(d), None) == _lib.ERR_BADARG
    assert b"descriptor" in L.subgacc_last_error()
    d.struct_bytes = C.sizeof(_lib.JoinDesc)
    assert L.subgacc_sjoin_fill_v2(C.byref(d), None) == _lib.ERR_BADARG          # neither row_off nor row_len
    assert L.subgacc_sjoin_fill_v2(None, None) == _lib.ERR_BADARG
    # SUBGACC_JOIN_OPT_SIZES (the whole join in one call): what it refuses before it launches anything
    buf = (C.c_int64 * 64)()
    here = C.addressof(buf)
    d.row_off, d.n_rows, d.S, d.pair_block = here, 4, 4, 2
    d.options = 2                                                               # an option bit this library does not know
    assert L.subgacc_sjoin_fill_v2(C.byref(d), None) == _lib.ERR_BADARG and b"option" in L.subgacc_last_error()
    d.options = _lib.JOIN_OPT_SIZES
    d.form = _lib.JOIN_COUNTS
    assert L.subgacc_sjoin_fill_v2(C.byref(d), None) == _lib.ERR_BADARG          # the row form only
    d.form = _lib.JOIN_ROWS
    assert L.subgacc_sjoin_fill_v2(C.byref(d), None) == _lib.ERR_BADARG          # no out_seg
    d.out_seg, d.seg = here, here
    assert L.subgacc_sjoin_fill_v2(C.byref(d), None) == _lib.ERR_BADARG          # seg next to out_seg
    d.seg, d.own = None, here
    assert L.subgacc_sjoin_fill_v2(C.byref(d), None) == _lib.ERR_WORKSPACE       # (no output: the size pass alone) no state
    # (round 6: with an output, what the FILL would refuse is refused before the size pass has written out_seg / host_tail)
    d.out_xz = here
    assert L.subgacc_sjoin_fill_v2(C.byref(d), None) == _lib.ERR_BADARG and b"null argument" in L.subgacc_last_error()
    d.ids = d.payload = d.flags = here
    d.payload_kind = 9
    assert L.subgacc_sjoin_fill_v2(C.byref(d), None) == _lib.ERR_BADARG and b"payload kind" in L.subgacc_last_error()
    d.payload_kind = _lib.JOIN_KEY64                                            # 64-bit keys are the payload of STRIDED / HEADED rows
    assert L.subgacc_sjoin_fill_v2(C.byref(d), None) == _lib.ERR_BADARG and b"row layout" in L.subgacc_last_error()
    d.payload_kind = _lib.JOIN_SFPTR
    assert L.subgacc_sjoin_fill_v2(C.byref(d), None) == _lib.ERR_WORKSPACE       # no state
    # ABI 7, headed rows: neither row_off nor row_len, a row_stride -- and not two layouts at once
    d.row_len, d.row_stride = here, 32
    assert L.subgacc_sjoin_fill_v2(C.byref(d), None) == _lib.ERR_BADARG and b"exactly one" in L.subgacc_last_error()
    d.row_off = d.row_len = None
    assert L.subgacc_sjoin_fill_v2(C.byref(d), None) == _lib.ERR_WORKSPACE       # headed rows: accepted up to the missing state
    d.row_stride = 1
    assert L.subgacc_sjoin_fill_v2(C.byref(d), None) == _lib.ERR_BADARG          # a headed row needs a slot behind its length
    d.row_off, d.row_stride = here, 0
    d.size_state, d.size_state_bytes = here, 8
    assert L.subgacc_sjoin_fill_v2(C.byref(d), None) == _lib.ERR_WORKSPACE       # a state too small
    assert L.subgacc_sjoin_workspace_bytes(4) >= 64 + 8 and L.subgacc_sjoin_workspace_bytes(1 << 22) >= 64 + 8 * (1 << 12)
