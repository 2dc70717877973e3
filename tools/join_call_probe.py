#!/usr/bin/env python3
"""Dev-only: what a join CALL costs per step on a resident float store (cit2-PPR shaped: 100 members per row, 65,536 pairs), form by
form -- CapturedJoin as shipped (one library call: single-launch size pass + fill), the same as a HIP graph, replays alone, and the
five-launch form of rounds 3-4 launched eagerly (memset, two size kernels, fill, read-back; buffers and the descriptor built once).      python tools/join_call_probe.py [steps] [B]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import surel_plus_amd as sp                                    # noqa: E402
from surel_plus_amd import _lib as L                            # noqa: E402
from surel_plus_amd.graphs import ppr_like_spg                  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda:0")
N = 2_927_963
z = ppr_like_spg(N, 100, seed=3, device=dev)
gens = [torch.Generator(device=dev).manual_seed(s) for s in range(16)]
edges = [torch.randint(0, N, (2, B), device=dev, generator=g) for g in gens]
torch.cuda.synchronize()


def timed(name, step, resolve, depth=1):
    for s in range(10):
        resolve(step(s))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pend = []
    for s in range(K):
        pend.append(step(s))
        if len(pend) > depth:
            resolve(pend.pop(0))
    for p in pend:
        resolve(p)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    # host cost alone: queue K steps without ever waiting
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    last = None
    for s in range(50):
        last = step(s)
    th = (time.perf_counter() - t1) / 50
    torch.cuda.synchronize()
    resolve(last)
    print(f"{name:58s} {dt * 1e6:8.1f} us / step   {B / dt / 1e6:8.1f} M pairs/s   host-only {th * 1e6:6.1f} us", flush=True)


# V0: the captured join as shipped
caps = [sp.CapturedJoin(z, B, graph=True) for _ in (0, 1)]
one = [sp.CapturedJoin(z, B) for _ in (0, 1)]
timed("V4 ONE call: OPT_SIZES + fill, host_tail", lambda s: one[s & 1](edges[s & 15]), lambda q: q.finish()[0].shape[0])
streams = [torch.cuda.Stream(device=dev) for _ in (0, 1)]
for st_ in streams:
    st_.wait_stream(torch.cuda.current_stream(dev))
timed("V5 ONE call, two joins in turn on TWO streams", lambda s: one[s & 1](edges[s & 15], stream=streams[s & 1]), lambda q: q.finish()[0].shape[0])
four = one + [sp.CapturedJoin(z, B) for _ in (0, 1)]
timed("V5' four joins in turn on two streams, depth 3", lambda s: four[s & 3](edges[s & 15], stream=streams[s & 1]), lambda q: q.finish()[0].shape[0], depth=3)
timed("V0 the same as a HIP graph", lambda s: caps[s & 1](edges[s & 15]), lambda q: q.finish()[0].shape[0])
timed("V0' the same, edge already in the static buffer", lambda s: caps[s & 1](caps[s & 1].edge), lambda q: q.finish()[0].shape[0])

# V2: replays alone (no copy-in, no wait for the read-back)
timed("V2 graph.replay() alone, nothing resolved", lambda s: caps[s & 1].graph.replay(), lambda q: None)


# V3: lean eager launches
class LeanJoin:
    def __init__(self, z, B):
        S = 2 * B
        self.S, self.z = S, z
        self.buf = torch.empty(S + 3, dtype=torch.int64, device=dev)
        self.seg, self.flags, self.tail = self.buf[:S + 1], self.buf[S + 1:].view(torch.int32), self.buf[S:]
        self.ws = torch.empty(L.lib().subgacc_sjoin_workspace_bytes(S), dtype=torch.uint8, device=dev)
        self.edge = torch.zeros((2, B), dtype=torch.int64, device=dev)
        self.out = torch.empty(S * z.max_len * 2, dtype=torch.float32, device=dev)
        self.host = torch.empty(3, dtype=torch.int64, pin_memory=True)
        self.ev = torch.cuda.Event()
        d = L.JoinDesc()
        d.struct_bytes, d.form, d.payload_kind = C.sizeof(L.JoinDesc), L.JOIN_ROWS, L.JOIN_F64
        d.row_off, d.n_rows, d.ids, d.payload, d.max_len = z.indptr.data_ptr(), z.n_rows, z.indices.data_ptr(), z.data.data_ptr(), z.max_len
        d.own, d.S, d.seg, d.pair_block = self.edge.data_ptr(), S, self.seg.data_ptr(), B
        d.out_xz, d.flags = self.out.data_ptr(), self.flags.data_ptr()
        self.d = d
        self.lib = L.lib()
        self.args = (C.c_void_p(z.indptr.data_ptr()), z.n_rows, C.c_void_p(self.edge.data_ptr()), C.c_void_p(0), S,
                     C.c_void_p(self.seg.data_ptr()), C.c_void_p(self.flags.data_ptr()), C.c_void_p(self.ws.data_ptr()), self.ws.numel())

    def __call__(self, edge, read_back=True):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if edge is not self.edge:
            self.edge.copy_(edge, non_blocking=True)
        self.flags.zero_()
        self.lib.subgacc_sjoin_sizes(*self.args, st)
        self.lib.subgacc_sjoin_fill_v2(C.byref(self.d), st)
        if read_back:
            self.host.copy_(self.tail, non_blocking=True)
        self.ev.record()
        return self

    def finish(self):
        self.ev.synchronize()
        return int(self.host[0])


lean = [LeanJoin(z, B) for _ in (0, 1)]
timed("V3 lean eager: zero + sizes + fill_v2 + read-back", lambda s: lean[s & 1](edges[s & 15]), lambda q: q.finish())
timed("V3' lean eager, no read-back (event only)", lambda s: lean[s & 1](edges[s & 15], False), lambda q: q.ev.synchronize())
timed("V3'' lean eager, edge in place, no read-back", lambda s: lean[s & 1](lean[s & 1].edge, False), lambda q: q.ev.synchronize())
timed("V4 again", lambda s: one[s & 1](edges[s & 15]), lambda q: q.finish()[0].shape[0])
