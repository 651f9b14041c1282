"""The decoder of PackedIdRelVec streams that ft_packed.hip runs on the device (reindexer_amd/csrc/ft_packed_decode.h), compiled for the
host (tests/cpp/ft_packed_decode_cpu.cc) and checked on the CPU:
  * against the committed streams of the reference's own packer (tests/golden/ft.npz) and, where oracle/_ref exists, the live packer;
  * against the host decoder PositionPostings::AppendPacked (itself pinned to the same) on random streams, incl. the derived
    (field, tf, first position) entries and the range index;
  * its error behaviour (truncated streams, descending ids, field out of range)."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

from .ft_pack import flat_entries, pack_postings
from .test_bm25_oracle import FT_GOLDEN, _unpack, make_pos_postings

LIB = Path(__file__).resolve().parent / "cpp" / "libft_packed_decode_cpu.so"


@pytest.fixture(scope="module")
def dec():
    if not LIB.exists():
        from reindexer_amd import build
        build.build_cpp_tests()
    L = C.CDLL(str(LIB))
    L.ftpk_count.restype = C.c_uint32
    L.ftpk_count.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
    L.ftpk_write.restype = C.c_uint32
    L.ftpk_write.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32] + [C.c_void_p] * 8 + [C.c_uint32, C.c_void_p]
    return L


def decode(L, data, afp, nf, range_docs=8192):
    data = np.ascontiguousarray(data, np.uint8)
    cnt = np.zeros(4, np.uint32)
    st = L.ftpk_count(data.ctypes.data, data.shape[0], afp, nf, range_docs, cnt.ctypes.data)
    if st:
        return st, None
    n, npos, nent, last = (int(x) for x in cnt)
    n_ranges = last // range_docs + 2 if n else 0
    doc, po, fp = np.zeros(n, np.uint32), np.zeros(n + 1, np.uint32), np.zeros(npos, np.uint64)
    eo, ef, et, e1 = np.zeros(n + 1, np.uint32), np.zeros(nent, np.uint8), np.zeros(nent, np.uint32), np.zeros(nent, np.uint32)
    ro = np.zeros(n_ranges, np.uint32)
    cnt2 = np.zeros(4, np.uint32)
    st = L.ftpk_write(data.ctypes.data, data.shape[0], afp, nf, range_docs, doc.ctypes.data, po.ctypes.data, fp.ctypes.data, eo.ctypes.data,
                      ef.ctypes.data, et.ctypes.data, e1.ctypes.data, ro.ctypes.data, n_ranges, cnt2.ctypes.data)
    assert st == 0 and np.array_equal(cnt, cnt2)
    return 0, dict(doc=doc, pos_off=po, fpos=fp, ent_off=eo, ent_field=ef, ent_tf=et, ent_first=e1, range_off=ro)


def check(L, data, afp, nf, want):
    st, got = decode(L, data, afp, nf)
    assert st == 0
    assert np.array_equal(got["doc"], want["doc"]) and np.array_equal(got["pos_off"], want["pos_off"]) and np.array_equal(got["fpos"], want["fpos"])
    eo, ef, et, e1, ro = flat_entries(want["doc"], want["pos_off"], want["fpos"])
    assert np.array_equal(got["ent_off"], eo) and np.array_equal(got["ent_field"], ef) and np.array_equal(got["ent_tf"], et)
    assert np.array_equal(got["ent_first"], e1) and np.array_equal(got["range_off"], ro)


@pytest.mark.parametrize("name", ["plain", "arrays"])
def test_shared_decoder_and_test_packer_against_reference_golden(dec, name):
    z = np.load(FT_GOLDEN)
    data, afp = z[f"packed_{name}_bytes"], int(z[f"packed_{name}_afp"])
    want = dict(doc=z[f"packed_{name}_doc"], pos_off=z[f"packed_{name}_pos_off"], fpos=z[f"packed_{name}_fpos"])
    nf = int(want["fpos"].max() >> np.uint64(56)) + 1
    check(dec, data, afp, nf, want)
    mine, mine_afp = pack_postings(want["doc"], want["pos_off"], want["fpos"])     # the test-side packer writes the reference's bytes
    assert np.array_equal(mine, data) and (mine_afp == afp or (afp >= len(data) and mine_afp == len(data)))
    st, _ = decode(dec, data[:-1], afp, nf)
    assert st == 1                                                                   # truncated
    st, _ = decode(dec, data, afp, 1 if nf > 1 else 0)
    assert st == 3                                                                   # field out of range


def test_test_packer_against_live_reference_packer():
    from oracle.pyoracle import ref_ft_or_none
    real = ref_ft_or_none(4)
    if real is None:
        pytest.skip("oracle/_ref/libref_ft.so not available")
    rng = np.random.default_rng(5)
    for arr in (False, True):
        for _ in range(3):
            s = make_pos_postings(rng, 200000, 4, int(rng.integers(1, 2000)), 1.0, array_fields=arr, max_pos=1 << 20)
            data, afp = real.pack(s)
            mine, mine_afp = pack_postings(s["doc"], s["pos_off"], s["fpos"])
            assert np.array_equal(mine, data)
            assert mine_afp == afp or (afp >= len(data) and mine_afp == len(data))
    real.close()


@pytest.mark.parametrize("seed", range(6))
def test_shared_decoder_equals_host_decoder_on_random_streams(dec, seed):
    rng = np.random.default_rng(100 + seed)
    nf = int(rng.integers(1, 6))
    n = int(rng.integers(1, 4000))
    s = make_pos_postings(rng, int(rng.choice([5000, 300000, 3_000_000])), nf, n, 1.0, array_fields=bool(seed % 2),
                          max_pos=int(rng.choice([40, 1 << 14, (1 << 28) - 1])))
    if seed % 2:   # arrays appear only from some element on: both layouts in one stream
        cut = int(rng.integers(0, n))
        lo = int(s["pos_off"][cut])
        s["fpos"][:lo] &= ~np.uint64(((1 << 28) - 1) << 28)
        for i in range(cut):   # clearing array indexes may create duplicates / disorder inside a posting: re-sort and de-duplicate
            a, b = int(s["pos_off"][i]), int(s["pos_off"][i + 1])
            s["fpos"][a:b] = np.sort(s["fpos"][a:b])
        keep = np.ones(len(s["fpos"]), bool)
        owner = np.repeat(np.arange(n), np.diff(s["pos_off"].astype(np.int64)))
        keep[1:] = (owner[1:] != owner[:-1]) | (s["fpos"][1:] != s["fpos"][:-1])
        s["fpos"] = s["fpos"][keep]
        po = np.zeros(n + 1, np.int64)
        np.add.at(po, owner[keep] + 1, 1)
        s["pos_off"] = np.cumsum(po).astype(np.uint32)
    data, afp = pack_postings(s["doc"], s["pos_off"], s["fpos"])
    hd, hp, hf = _unpack(data, afp)                                      # the host decoder (pinned to the reference's packer)
    assert np.array_equal(hd, s["doc"]) and np.array_equal(hp, s["pos_off"]) and np.array_equal(hf, s["fpos"])
    check(dec, data, afp, nf, s)


def test_shared_decoder_errors_and_empty(dec):
    st, got = decode(dec, np.zeros(0, np.uint8), 0, 3)
    assert st == 0 and got["doc"].shape[0] == 0 and got["pos_off"].tolist() == [0] and got["range_off"].shape[0] == 0
    # two postings with the same document: ids must ascend strictly
    data, afp = pack_postings([5, 5], [0, 1, 2], [3, 4])
    st, _ = decode(dec, data, afp, 1)
    assert st == 2
    # every proper prefix that cuts an element is "truncated" (a prefix ending on an element boundary is a valid shorter list)
    data, afp = pack_postings([1, 9, 300], [0, 2, 3, 6], [1, 2 | (1 << 56), 7, 1, 5, 9 | (2 << 56)])
    whole, _ = decode(dec, data, afp, 3)
    assert whole == 0
    seen = set()
    for cutlen in range(len(data)):
        st, got = decode(dec, data[:cutlen], afp, 3)
        seen.add(st)
        assert st in (0, 1)
    assert seen == {0, 1}
