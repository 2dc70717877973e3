"""ctypes binding of libsubgacc_hip.so (the C ABI declared in include/subgacc.h).

There is no CPU fallback: if the library is missing, or no gfx950 device is visible when a compute entry
point is called, this module raises.  Error codes map onto the exceptions the reference raises for the
same condition (subg_acc/subg_acc.c:658, :688-721, :905-915).
"""
import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# SUBGACC_LIB (dev-only): load an experiment build from elsewhere (tools/ab_*.sh) -- the shipped library is never overwritten
LIB_PATH = os.environ.get("SUBGACC_LIB") or os.path.join(_HERE, "libsubgacc_hip.so")
CSRC = os.path.join(_HERE, "csrc")

OK, ERR_BADARG, ERR_WORKSPACE, ERR_KEYWIDTH, ERR_CAPACITY, ERR_HIP, ERR_LDS, ERR_NODEVICE = 0, -1, -2, -3, -4, -5, -6, -7
RNG_RAND_R, RNG_PHILOX = 0, 1
ORDER_WALK_MAJOR, ORDER_STEP_MAJOR = 0, 1

# every symbol include/subgacc.h declares (tests check the built library exports all of them)
SYMBOLS = (
    "subgacc_abi_version", "subgacc_last_error", "subgacc_device_count", "subgacc_key_shift",
    "subgacc_rng_positions_workspace_bytes", "subgacc_rng_positions", "subgacc_walk_sets", "subgacc_walk_spg",
    "subgacc_compact_rows",
    "subgacc_scan_workspace_bytes", "subgacc_exclusive_scan_i32", "subgacc_compact_sets",
    "subgacc_uniq_table_bytes", "subgacc_uniq_reset", "subgacc_uniq_insert",
    "subgacc_uniq_number_workspace_bytes", "subgacc_uniq_number", "subgacc_uniq_translate", "subgacc_unpack_lp",
    "subgacc_spg_build",
    "subgacc_sjoin_workspace_bytes", "subgacc_sjoin_sizes",
    "subgacc_ppr_slab_bytes", "subgacc_ppr_slab_reset", "subgacc_ppr_topk", "subgacc_ppr_normalize", "subgacc_ppr_encode",
    "subgacc_walk_join", "subgacc_sjoin_sizes_rows",
    "subgacc_encode_sizes", "subgacc_encode_fill", "subgacc_finish_rows",
    "subgacc_batch_sampler_workspace_bytes", "subgacc_batch_sampler", "subgacc_step_prologue",
    "subgacc_hop_records_format", "subgacc_hop_records_build", "subgacc_step_dedup_workspace_bytes",
    "subgacc_step_prologue_dedup", "subgacc_walk_spg_sparse",
    "subgacc_keyrows_register", "subgacc_keyrows_cand_capacity", "subgacc_walk_tags", "subgacc_keyrows_compact", "subgacc_keyrows_translate", "subgacc_rng_replay", "subgacc_walk_keyrows64", "subgacc_worklist_workspace_bytes", "subgacc_worklist_by_root", "subgacc_walk_spg_list",
    "subgacc_sjoin_fill_v2", "subgacc_publish_words", "subgacc_rows_to_headed",
)


class WalkCfg(C.Structure):
    """struct subgacc_walk_cfg"""
    _fields_ = [("num_walks", C.c_int32), ("num_steps", C.c_int32), ("bucket", C.c_int32), ("rng_mode", C.c_int32),
                ("seed", C.c_uint32), ("first_hop_wo", C.c_int32), ("order", C.c_int32),
                ("cap_root_degree", C.c_int32), ("indptr64", C.c_int32), ("emit_walks", C.c_int32),
                ("hop_records", C.c_void_p), ("rec_id_bits", C.c_int32), ("rec_beg_bits", C.c_int32),
                ("walk_pos", C.c_void_p), ("row_pitch", C.c_int32)]


class JoinDesc(C.Structure):
    """struct subgacc_join_desc (include/subgacc.h, ABI 6): what one call of subgacc_sjoin_fill_v2 joins"""
    _fields_ = [("struct_bytes", C.c_int32), ("form", C.c_int32), ("payload_kind", C.c_int32), ("max_len", C.c_int32),
                ("row_off", C.c_void_p), ("row_len", C.c_void_p), ("row_stride", C.c_int64), ("n_rows", C.c_int64),
                ("ids", C.c_void_p), ("payload", C.c_void_p), ("uniq_table", C.c_void_p), ("uniq_capacity", C.c_int64),
                ("own", C.c_void_p), ("partner", C.c_void_p), ("S", C.c_int64), ("seg", C.c_void_p), ("pair_block", C.c_int64),
                ("table", C.c_void_p), ("table_rows", C.c_int64), ("k", C.c_int32), ("num_walks", C.c_int32),
                ("num_steps", C.c_int32), ("options", C.c_int32), ("out_xz", C.c_void_p), ("out_idx", C.c_void_p),
                ("out_segid", C.c_void_p), ("out_counts", C.c_void_p), ("out_pairs", C.c_void_p), ("out_mult", C.c_void_p),
                ("out_cnt", C.c_void_p), ("flags", C.c_void_p), ("out_seg", C.c_void_p), ("size_state", C.c_void_p),
                ("size_state_bytes", C.c_int64), ("host_tail", C.c_void_p)]


JOIN_SFPTR, JOIN_F64, JOIN_KEY32, JOIN_KEY64 = 0, 1, 2, 3      # payload_kind
JOIN_ROWS, JOIN_COUNTS, JOIN_PAIRS = 0, 1, 2                   # form
JOIN_OPT_SIZES = 1                                             # options: size pass + fill in one call


class SubgAccError(RuntimeError):
    pass


_lib = None
# the reference prints "#SubGAcc: ..." statistics to stdout (subg_acc.c:878,1009); SUBGACC_QUIET=1 silences ours
VERBOSE = os.environ.get("SUBGACC_QUIET", "0") != "1"


def build(force=False):
    """hipcc-compile csrc/*.hip for gfx950 into surel_plus_amd/libsubgacc_hip.so (in-tree)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))]
    srcs.append(os.path.join(_HERE, "..", "include", "subgacc.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", CSRC, "-j4", "all"])
    return LIB_PATH


def lib():
    """Load the library (after torch, so that both bind the same libamdhip64 runtime instance)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  -- torch's bundled libamdhip64.so.7 must be the one already in the process
    if not os.path.exists(LIB_PATH):
        raise SubgAccError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"(or `make -C surel_plus_amd/csrc`). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_size_t
    cfgp = C.POINTER(WalkCfg)
    sig = {
        "subgacc_abi_version": (C.c_int, []),
        "subgacc_last_error": (C.c_char_p, []),
        "subgacc_device_count": (C.c_int, []),
        "subgacc_key_shift": (C.c_int, [i32, i32]),
        "subgacc_rng_positions_workspace_bytes": (sz, [i64]),
        "subgacc_rng_positions": (C.c_int, [cfgp, vp, i64, vp, i64, i32, u64, vp, vp, vp, sz, vp]),
        "subgacc_walk_sets": (C.c_int, [cfgp, vp, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp]),
        "subgacc_walk_spg": (C.c_int, [cfgp, vp, vp, i64, vp, i64, i64, vp, vp, vp, i64, vp, vp, vp, vp, vp]),
        "subgacc_compact_rows": (C.c_int, [vp, vp, vp, vp, i64, i32, vp, vp, vp, i64, vp]),
        "subgacc_scan_workspace_bytes": (sz, [i64]),
        "subgacc_exclusive_scan_i32": (C.c_int, [vp, i64, vp, vp, sz, vp]),
        "subgacc_compact_sets": (C.c_int, [vp, vp, vp, vp, i64, i32, vp, vp, vp, i64, i64, vp, vp, vp]),
        "subgacc_uniq_table_bytes": (sz, [i64]),
        "subgacc_uniq_reset": (C.c_int, [vp, i64, vp]),
        "subgacc_uniq_insert": (C.c_int, [vp, i64, vp, i64, i64, vp, vp, vp]),
        "subgacc_uniq_number_workspace_bytes": (sz, [i64, i64]),
        "subgacc_uniq_number": (C.c_int, [vp, i64, vp, i64, vp, i64, vp, i64, vp, sz, vp]),
        "subgacc_uniq_translate": (C.c_int, [vp, i64, vp, i64, vp, i32, vp]),
        "subgacc_unpack_lp": (C.c_int, [vp, i64, vp, i32, i32, vp, vp, vp, i32, vp]),
        "subgacc_spg_build": (C.c_int, [vp, i64, vp, vp, vp, i64, i32, vp, vp, vp, vp]),
        "subgacc_sjoin_workspace_bytes": (sz, [i64]),
        "subgacc_sjoin_sizes": (C.c_int, [vp, i64, vp, vp, i64, vp, vp, vp, sz, vp]),
    }
    f32 = C.c_float
    sig["subgacc_ppr_slab_bytes"] = (sz, [i32, i32])
    sig["subgacc_ppr_slab_reset"] = (C.c_int, [vp, i32, i32, vp])
    sig["subgacc_ppr_topk"] = (C.c_int, [vp, i32, vp, i64, vp, i64, f32, f32, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp])
    sig["subgacc_ppr_normalize"] = (C.c_int, [vp, i32, vp, i64, vp, i64, vp, vp, i32, vp, vp, vp])
    sig["subgacc_ppr_encode"] = (C.c_int, [vp, i64, vp, vp, vp])
    sig["subgacc_walk_join"] = (C.c_int, [vp, i64, i32, vp, vp, vp, i32, vp, i64, vp, vp])
    sig["subgacc_hop_records_format"] = (C.c_int, [i64, i64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)])
    sig["subgacc_hop_records_build"] = (C.c_int, [vp, i32, vp, i64, i64, i32, i32, vp, vp])
    sig["subgacc_step_dedup_workspace_bytes"] = (C.c_size_t, [i64])
    sig["subgacc_step_prologue_dedup"] = (C.c_int, [vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, i64, vp, C.c_size_t, vp, vp])
    sig["subgacc_walk_spg_sparse"] = (C.c_int, [cfgp, vp, vp, i64, vp, i64, vp, vp, vp, i64, vp, vp, vp, vp, vp])
    sig["subgacc_step_prologue"] = (C.c_int, [vp, i64, vp, i64, vp, vp, i64, vp])
    sig["subgacc_batch_sampler_workspace_bytes"] = (sz, [i64])
    sig["subgacc_batch_sampler"] = (C.c_int, [vp, i32, vp, i64, vp, i64, i32, i32, i32, C.c_uint32, vp, i64, vp, vp, sz, vp, vp])
    sig["subgacc_sjoin_sizes_rows"] = (C.c_int, [vp, i64, vp, vp, i64, vp, vp, vp, sz, vp])
    sig["subgacc_encode_sizes"] = (C.c_int, [vp, vp, i64, vp, i32, vp, i32, vp, vp, vp])
    sig["subgacc_encode_fill"] = (C.c_int, [vp, vp, vp, i64, i32, vp, i32, vp, i32, vp, vp, i64, vp, vp, vp, vp, vp, vp])
    sig["subgacc_finish_rows"] = (C.c_int, [vp, vp, vp, i64, i32, i64, vp, i64, vp, vp, vp])
    sig["subgacc_keyrows_register"] = (C.c_int, [vp, vp, i64, i32, i64, vp, i64, vp, i64, vp, vp, vp])
    sig["subgacc_keyrows_cand_capacity"] = (i64, [i64])
    sig["subgacc_walk_tags"] = (C.c_int, [cfgp, vp, vp, i64, vp, i64, i64, vp, vp, vp, vp, i64, vp, i64, vp, vp])
    sig["subgacc_keyrows_compact"] = (C.c_int, [vp, vp, vp, vp, i64, i32, i64, vp, i64, vp, vp, i64, vp, vp, vp, i64, vp, vp, vp])
    sig["subgacc_keyrows_translate"] = (C.c_int, [vp, i64, vp, vp, i64, vp, vp, i64, vp])
    sig["subgacc_walk_keyrows64"] = (C.c_int, [cfgp, vp, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp])
    sig["subgacc_rng_replay"] = (C.c_int, [cfgp, vp, vp, i64, vp, i64, i32, u64, vp, vp, vp, vp])
    sig["subgacc_worklist_workspace_bytes"] = (sz, [i64])
    sig["subgacc_worklist_by_root"] = (C.c_int, [vp, i64, i64, vp, vp, vp, sz, vp])
    sig["subgacc_walk_spg_list"] = (C.c_int, [cfgp, vp, vp, i64, vp, i64, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp])
    sig["subgacc_sjoin_fill_v2"] = (C.c_int, [C.POINTER(JoinDesc), vp])
    sig["subgacc_publish_words"] = (C.c_int, [vp, i64, vp, vp])
    sig["subgacc_rows_to_headed"] = (C.c_int, [vp, i64, vp, vp, i32, i64, vp, vp, vp, vp])
    assert set(sig) == set(SYMBOLS)
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    if L.subgacc_abi_version() != 7:
        raise SubgAccError("libsubgacc_hip.so ABI version mismatch")
    _lib = L
    return L


def check(rc):
    """Turn a negative subgacc_status into the exception the reference raises for that condition."""
    if rc >= 0:
        return rc
    msg = lib().subgacc_last_error().decode("utf-8", "replace")
    if rc == ERR_BADARG:
        raise TypeError(f"Input parsing error. ({msg})")
    if rc == ERR_WORKSPACE:
        raise MemoryError(msg)
    if rc == ERR_KEYWIDTH:
        raise AssertionError(msg)
    if rc == ERR_LDS:
        raise ValueError(msg)
    raise SubgAccError(f"subgacc status {rc}: {msg}")


def require_device():
    """Fail loudly unless a gfx950 GPU is usable from this process."""
    import torch
    if not torch.cuda.is_available():
        raise SubgAccError("no HIP device visible: the SubGAcc hot path only runs on MI355X (gfx950); "
                           "there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def join_fill(form=JOIN_ROWS, payload_kind=JOIN_SFPTR, **fields):
    """One join through subgacc_sjoin_fill_v2 on the current stream: `fields` are the members of subgacc_join_desc by name --
    torch tensors for the pointers (None = NULL), ints for the rest; what a form does not read stays zero."""
    d = JoinDesc()
    d.struct_bytes, d.form, d.payload_kind = C.sizeof(JoinDesc), int(form), int(payload_kind)
    for name, val in fields.items():
        if val is None:
            continue
        setattr(d, name, val.data_ptr() if hasattr(val, "data_ptr") else int(val))
    return check(lib().subgacc_sjoin_fill_v2(C.byref(d), stream_ptr()))


def publish(src, host):
    """queue `host[:] = src` (a few int64 words: sizes, status) on the current stream WITHOUT a copy engine: see subgacc_publish_words.
    (`_lib._READBACK_COPY = True` takes torch's asynchronous copy instead: the A/B switch)"""
    if _READBACK_COPY or src.numel() > 4096 or not src.is_contiguous() or src.dtype != host.dtype or src.element_size() != 8:
        host.copy_(src, non_blocking=True)
    else:
        check(lib().subgacc_publish_words(C.c_void_p(src.data_ptr()), src.numel(), C.c_void_p(host.data_ptr()), stream_ptr()))


_KEPT = []
_KEPT_LOCK = threading.Lock()


def keep_until(event, obj):
    """`obj` (the pinned buffer a publish() kernel writes) stays alive until `event` -- recorded behind that kernel -- has passed,
    even if its owner is dropped first: torch's host allocator knows nothing of a kernel that writes pinned memory and would hand
    the block to somebody else.  (Finished entries leave when the next one arrives.)
    Called from every thread that queues steps (the reference's loader runs four, train.py:88-111): look-then-pop on the shared list
    is one critical section -- without the lock a second thread could pop the entry BEHIND the finished one the first had seen,
    i.e. drop a block a kernel is still going to write.  Entries are queued by different streams, so a finished one may stand behind
    an unfinished one: every finished entry leaves, wherever it stands."""
    with _KEPT_LOCK:
        _KEPT[:] = [(ev, o) for ev, o in _KEPT if not ev.query()]
        _KEPT.append((event, obj))


_READBACK_COPY = False      # True: torch's asynchronous copy instead (how the A/B of profiles/r25_readback_ab.log was run)


def ptr(t):
    """device pointer of a torch tensor (None -> NULL)"""
    if t is None:
        return C.c_void_p(0)
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensor expected"
    return C.c_void_p(t.data_ptr())
