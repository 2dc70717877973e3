"""GPU (MI355X): round-5 additions, through the C ABI, against the oracle / the reference's goldens / the other forms.
  * a 64-bit key-rows batch that is numbered after the fact is sampled again with the configuration AND the
    rand_r stream positions of its first sampling (calls_before, rng_streams, cap_root_degree), not with defaults."""
import numpy as np
import pytest
import torch

import oracle
from test_gpu_parity import _oracle_spg, dir_graph, sp, sym_graph  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,hops", [(200, 4), (100, 3)])
@pytest.mark.parametrize("kw", [dict(prefix=40), dict(cap_root_degree=False), dict(prefix=7, cap_root_degree=False)])
def test_key_rows_numbered_after_the_fact_describe_the_batch_that_was_joined(sp, M, hops, kw):
    """sets.number() / enc / to_csr() of a key-rows batch (64-bit keys: sampled again with the table form; 32-bit keys: registered
    from the rows, tags from walk_tags) == the table-form sample made with the SAME keyword arguments, rand_r stream included"""
    ptr_, idx = sym_graph(6000, 50000, seed=17, hubs=3)          # hubs: a root of degree > M is where cap_root_degree matters
    csr = sp.DeviceCSR(ptr_, idx)
    q = np.random.default_rng(4).integers(0, 6000, 900).astype(np.int32)
    deg = np.diff(ptr_)
    q[:3] = np.argsort(-deg)[:3]                                   # the hubs themselves are roots
    kw = dict(kw)
    prefix = q[::-1][: kw.pop("prefix", 0)].copy()                 # roots "sampled before" this batch on the same rand_r stream
    if prefix.size:
        from surel_plus_amd.shard import rand_r_calls
        kw["calls_before"] = rand_r_calls(csr.indptr, prefix, M, hops)
        assert kw["calls_before"] > 0
    zk, sk = sp.sample_spg(csr, q, num_walks=M, num_steps=hops, seed=19, rng="rand_r", strided=True, number_rows=False, **kw)
    assert sk.keyrows and sk.key64 == (hops == 4 and M >= 128)
    zt, st_ = sp.sample_spg(csr, q, num_walks=M, num_steps=hops, seed=19, rng="rand_r", strided=True, number_rows=True,
                            key_rows=False, **kw)
    assert not st_.keyrows
    assert torch.equal(sk.nsize, st_.nsize)
    assert sk.c == st_.c and torch.equal(sk.number().ukeys, st_.number().ukeys)
    assert torch.equal(sk.enc_int16(), st_.enc_int16())
    a, b = zk.to_csr(), zt.to_csr()
    assert torch.equal(a.indptr, b.indptr) and torch.equal(a.indices, b.indices) and torch.equal(a.data, b.data)
    # ... and what the reference's one sequential stream gives these roots when `prefix` was sampled in front of them
    o_nsize, o_remap, _ = oracle.gset_sampler(ptr_, idx, np.concatenate([prefix, q]), num_walks=M, num_steps=hops, rng="rand_r", seed=19)
    assert np.array_equal(sk.nsize.cpu().numpy(), o_nsize[prefix.size:])
    skip = int(o_nsize[: prefix.size].sum())
    rows = np.split(o_remap[0][skip:], np.cumsum(o_nsize[prefix.size:])[:-1])
    assert np.array_equal(a.indices.cpu().numpy(), np.concatenate([np.sort(r) for r in rows]))
