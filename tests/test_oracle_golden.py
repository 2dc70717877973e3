"""CPU: the oracle (oracle/) against the golden vectors the REFERENCE produced (oracle/gen_golden.py).
This is what pins the oracle; the GPU parity tests then compare the HIP path with the oracle."""
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, golden_files


def _load(name):
    return np.load(os.path.join(GOLDEN, name))


def test_rand_r_known_answers():
    g = _load("rand_r.npz")
    for k in g.files:
        seed = int(k.split("_")[1])
        assert np.array_equal(oracle.rand_r_stream(seed, 1000), g[k]), k


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    assert [hex(v) for v in oracle.philox4x32_10([0, 0, 0, 0], [0, 0])] == \
        ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert [hex(v) for v in oracle.philox4x32_10([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2)] == \
        ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    assert [hex(v) for v in oracle.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                                                 [0xa4093822, 0x299f31d0])] == \
        ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]
    # ... and for philox2x32-10, the generator of the rng="philox" mode
    assert [hex(v) for v in oracle.philox2x32_10([0, 0], 0)] == ["0xff1dae59", "0x6cd10df2"]
    assert [hex(v) for v in oracle.philox2x32_10([0xFFFFFFFF] * 2, 0xFFFFFFFF)] == ["0x2c3f628b", "0xab4fd7ad"]
    assert [hex(v) for v in oracle.philox2x32_10([0x243f6a88, 0x85a308d3], 0x13198a2e)] == ["0xdd7ce038", "0xf62a4c12"]


@pytest.mark.parametrize("name", golden_files("gset_"))
def test_gset_matches_reference(name):
    g = _load(name)
    out = oracle.gset_sampler(g["indptr"], g["indices"], g["query"], num_walks=int(g["M"]), num_steps=int(g["m"]),
                              bucket=int(g["bucket"]), seed=int(g["seed"]), debug=True)
    assert np.array_equal(out[0], g["nsize"])
    assert np.array_equal(out[1], g["remap"])
    assert np.array_equal(out[2], g["enc"])
    assert np.array_equal(out[3], g["raw"])
    assert out[0].dtype == np.int32 and out[1].dtype == np.int32 and out[2].dtype == np.int16


@pytest.mark.parametrize("name", golden_files("walk_"))
def test_walk_sampler_matches_reference(name):
    g = _load(name)
    walks, nsize, ids, counts = oracle.walk_sampler(g["indptr"], g["indices"], g["query"], num_walks=int(g["M"]),
                                                    num_steps=int(g["m"]), nthread=int(g["nthread"]),
                                                    seed=int(g["seed"]), replacement=bool(g["replacement"]))
    assert np.array_equal(walks, g["walks"])
    assert np.array_equal(nsize, g["nsize"])
    assert np.array_equal(ids, g["ids"])
    assert np.array_equal(counts, g["counts"])


@pytest.mark.parametrize("name", golden_files("spg_"))
def test_spg_build_matches_scipy(name):
    g = _load(name)
    indptr, indices, data = oracle.spg_build(g["nsize"], g["remap"])
    assert np.array_equal(indptr, g["z_indptr"])
    assert np.array_equal(indices, g["z_indices"])
    assert np.array_equal(data, g["z_data"])
    assert np.array_equal(oracle.enc_table(g["enc"]), g["encz"])


@pytest.mark.parametrize("name", golden_files("sjoin_"))
@pytest.mark.parametrize("ptr", [True, False])
def test_sjoin_matches_reference(name, ptr):
    g = _load(name)
    spg = (g["z_indptr"], g["z_indices"], g["z_data"])
    enc = g["encode"] if g["encode"].size else None
    for fn in (oracle.gather, oracle.gather_numpy):
        xz, ind = fn(g["edge"], spg, ptr=ptr, encode=enc)
        assert xz.dtype == np.float32 and ind.dtype == np.int64
        assert np.array_equal(xz, g[f"xz_ptr{int(ptr)}"]), fn.__name__
        assert np.array_equal(ind, g[f"ind_ptr{int(ptr)}"]), fn.__name__


def test_hgather_matches_reference():
    g = _load("hjoin_int.npz")
    xz, ind = oracle.hgather(g["hedge"], (g["z_indptr"], g["z_indices"], g["z_data"]), g["encode"])
    assert np.array_equal(xz, g["xz"])
    assert np.array_equal(ind, g["ind"])


def test_reference_invariants_hold_for_philox():
    """subg_acc/test/test.py:34-45 invariants are RNG independent."""
    g = _load("gset_mixeddeg_s1.npz")
    M, m = int(g["M"]), int(g["m"])
    nsize, remap, enc, raw = oracle.gset_sampler(g["indptr"], g["indices"], g["query"], num_walks=M, num_steps=m,
                                                 seed=5, rng="philox", debug=True)
    assert nsize.sum() == remap.shape[1]
    assert remap[1].max() == enc.shape[0] - 1
    assert (enc[remap[1]][:, 0] == M).sum() == len(g["query"])
    assert np.array_equal(enc[remap[1]], raw)
    off = np.concatenate([[0], np.cumsum(nsize)])
    for i in range(len(nsize)):
        rows = raw[off[i]:off[i + 1]]
        assert np.all(rows[:, 1:].sum(axis=0) == M)
        assert len(set(remap[0, off[i]:off[i + 1]])) == nsize[i]
        assert remap[0, off[i]] == g["query"][i]
    # philox with several threads == philox with one
    again = oracle.gset_sampler(g["indptr"], g["indices"], g["query"], num_walks=M, num_steps=m, seed=5,
                                rng="philox", nthreads=4, debug=True)
    for a, b in zip((nsize, remap, enc, raw), again):
        assert np.array_equal(a, b)


# ------------------------------------------------------------------------------- walk_join (legacy SUREL join)
@pytest.mark.parametrize("name", golden_files("walkjoin_"))
def test_oracle_walk_join_matches_reference(name):
    g = np.load(os.path.join(GOLDEN, name))
    off = np.concatenate([[0], np.cumsum(g["key_len"])])
    key = [g["key_ids"][off[i]:off[i + 1]] for i in range(len(g["key_len"]))]
    out, xrow = oracle.walk_join(g["walks"], key, g["query"], return_idx=True)
    np.testing.assert_array_equal(xrow, g["xrow"])
    np.testing.assert_array_equal(out, g["out"])
    assert out.dtype == g["out"].dtype and out.shape == g["out"].shape


# ------------------------------------------------------------------------------- batch_sampler (legacy SUREL mini-batches)
@pytest.mark.parametrize("name", golden_files("batch_"))
def test_oracle_batch_sampler_matches_reference(name):
    """the reference seeds with seed + getpid() (subg_acc.c:421): the fixture carries the effective seed of its run"""
    g = np.load(os.path.join(GOLDEN, name))
    out = oracle.batch_sampler(g["indptr"], g["indices"], g["query"], num_walks=int(g["M"]), num_steps=int(g["S"]),
                               thld=int(g["thld"]), seed_eff=int(g["seed_eff"]))
    assert out.dtype == np.int32 and np.array_equal(out, g["out"])
    assert len(set(out.tolist())) == len(out) and out[0] == g["query"][0]
