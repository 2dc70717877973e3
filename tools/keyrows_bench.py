"""Dev-only: the passes of csrc/keyrows.hip alone on synthetic key rows (cit2-like: stride 601, ~344 members per row, ~1,400
distinct keys), timed with HIP events; compile-time variants are built into /tmp and selected with SUBGACC_LIB:
    python tools/keyrows_bench.py [rows]            (variants: KRB_VARIANTS="-DKR_NT=0|-DKR_UNROLL=8|-DKR_THREADS=256")"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child(n):
    import torch
    from surel_plus_amd._lib import check, lib, ptr, stream_ptr
    L, dev, stride = lib(), "cuda", 601
    if os.environ.get("KRB_SYNTH") == "1":
        g = torch.Generator(device=dev).manual_seed(1)
        nsize = torch.randint(90, 600, (n,), device=dev, generator=g, dtype=torch.int32)
        pool = torch.randint(1, 1 << 25, (1432,), device=dev, generator=g, dtype=torch.int32)
        # a long-tailed key distribution, like LP rows: most members carry a few dozen hot keys
        pick = (torch.rand(n * stride, device=dev, generator=g) ** 6 * 1432).long().clamp_(max=1431)
        keys = pool[pick].contiguous()
        ids = torch.arange(n * stride, device=dev, dtype=torch.int32)
        del pick
    else:       # the real thing: key rows of the first n roots of the cit2-like graph
        from surel_plus_amd.graphs import preset_graph
        from surel_plus_amd.sampler import sample_sets
        csr = preset_graph("cit2")
        sets = sample_sets(csr, torch.arange(n, dtype=torch.int32, device=dev), 200, 3, rng="philox", fused_rows=True, strided=True,
                           number_rows=False, lazy=True)
        assert sets.keyrows
        nsize, ids, keys = sets.nsize, sets.ids, sets.slot
    row_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(nsize, 0, out=row_off[1:])
    X = int(row_off[-1])
    cap = 1 << 20
    table = torch.empty(L.subgacc_uniq_table_bytes(cap), dtype=torch.uint8, device=dev)
    ccap = L.subgacc_keyrows_cand_capacity(n)
    cand = torch.empty(ccap, dtype=torch.int32, device=dev)
    words = torch.zeros(4, dtype=torch.int64, device=dev)
    flags = words.view(torch.int32)[:4]
    out_i = torch.empty(X, dtype=torch.int32, device=dev)
    out_d = torch.empty(X, dtype=torch.int32, device=dev)
    ukeys = torch.empty(16384, dtype=torch.int64, device=dev)
    ws = torch.empty(L.subgacc_uniq_number_workspace_bytes(cap, 0), dtype=torch.uint8, device=dev)
    st = stream_ptr()
    def timed(fn, reps=5):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps
    def reg():
        check(L.subgacc_uniq_reset(ptr(table), cap, st)); words.zero_()
        check(L.subgacc_keyrows_register(ptr(keys), ptr(nsize), n, stride, 0, ptr(table), cap, ptr(cand), ccap, ptr(words[2:3]), ptr(flags), st))
    t_reg = timed(reg)
    t_reset = timed(lambda: (check(L.subgacc_uniq_reset(ptr(table), cap, st)), words.zero_()))
    reg()
    # (timing only: the numbering proper needs the candidates' exact tags -- subgacc_walk_tags -- or equal coarse tags collide in
    #  out_ukeys; any numbering of the distinct keys fills the dictionary the same way)
    valid = (torch.arange(stride, device=dev)[None, :] < nsize[:, None]).reshape(-1)
    uk = torch.unique(keys.view(n, stride).reshape(-1)[valid].long() & 0xFFFFFFFF)
    ukeys[: uk.numel()] = uk
    words[3] = uk.numel()
    del valid
    t_find = timed(lambda: check(L.subgacc_keyrows_compact(ptr(ids), ptr(keys), ptr(nsize), ptr(row_off), n, stride, 0, ptr(table), cap,
                                                           ptr(ukeys), ptr(words[3:4]), 16384, ptr(out_i), ptr(out_d), None, 0, None, ptr(flags), st)))
    def regcopy():
        words[2:3].zero_()
        check(L.subgacc_keyrows_compact(ptr(ids), ptr(keys), ptr(nsize), ptr(row_off), n, stride, 0, ptr(table), cap,
                                        None, None, 0, ptr(out_i), ptr(out_d), ptr(cand), ccap, ptr(words[2:3]), ptr(flags), st))
    t_rc = timed(regcopy)
    def translate():      # (in place: the payload must be keys again before every run)
        out_d.copy_(keys_packed)
        check(L.subgacc_keyrows_translate(ptr(out_d), X, None, ptr(table), cap, ptr(ukeys), ptr(words[3:4]), 16384, st))
    regcopy()
    keys_packed = out_d.clone()
    t_tr = timed(translate) - timed(lambda: out_d.copy_(keys_packed))
    t_copy = timed(lambda: out_d.copy_(out_i))
    import ctypes
    if hasattr(L, "subgacc_debug_kr_misses"):
        L.subgacc_debug_kr_misses.restype = ctypes.c_longlong
        L.subgacc_debug_kr_misses()
        translate(); torch.cuda.synchronize()
        print("dictionary misses in one translate pass:", L.subgacc_debug_kr_misses(), "of", X, flush=True)
    print(f"rows {n} members {X} cand {int(words[2])} distinct {int(words[3])}: register {t_reg - t_reset:.3f} ms ({4 * X / (t_reg - t_reset) / 1e9:.2f} TB/s)  "
          f"find+copy {t_find:.3f} ms ({16 * X / t_find / 1e9:.2f} TB/s)  register+copy {t_rc:.3f} ms  translate {t_tr:.3f} ms ({8 * X / t_tr / 1e9:.2f} TB/s)  "
          f"[torch copy of {4 * X >> 20} MiB: {t_copy:.3f} ms = {8 * X / t_copy / 1e9:.2f} TB/s]", flush=True)

if __name__ == "__main__":
    if os.environ.get("KRB_CHILD"):
        child(int(sys.argv[1]))
        sys.exit(0)
    n = sys.argv[1] if len(sys.argv) > 1 else "1000000"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from ab import build_variant
    for v in [""] + [x for x in os.environ.get("KRB_VARIANTS", "").split("|") if x]:
        env = dict(os.environ, KRB_CHILD="1", SUBGACC_QUIET="1")
        env.pop("SUBGACC_LIB", None)
        if v:
            env["SUBGACC_LIB"] = build_variant(v, ["keyrows.hip"])
        print(f"[{v or 'shipped'}] ", end="", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), n], env=env)
