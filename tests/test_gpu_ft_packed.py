"""-m gpu: posting lists uploaded as the reference stores them (PackedIdRelVec byte streams) and decoded ON THE DEVICE
(rxgpu_ft_set_words_packed / GpuFtMerger::SetWordsPacked, ft_packed.hip) — SURVEY 8f-4.  Bar: the device arrays (documents, positions,
(field, tf, first position) entries, range index) equal what the host decoder + rxgpu_ft_set_word_positions produce, byte streams of the
reference's own packer included, and merges over device-decoded words equal merges over host-uploaded ones."""
import numpy as np
import pytest

from oracle.pyoracle import FtOracle
from .ft_pack import flat_entries, pack_postings
from .test_bm25_oracle import FT_GOLDEN, _multi_case, make_pos_postings

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hostapi(rxgpu):
    from reindexer_amd import hostapi as h
    h.lib()
    return h


def _same(got, want):
    eo, ef, et, e1, ro = flat_entries(want["doc"], want["pos_off"], want["fpos"])
    assert np.array_equal(got["doc"], want["doc"]) and np.array_equal(got["pos_off"], want["pos_off"]) and np.array_equal(got["fpos"], want["fpos"])
    assert np.array_equal(got["ent_off"], eo) and np.array_equal(got["ent_field"], ef) and np.array_equal(got["ent_tf"], et)
    assert np.array_equal(got["ent_first"], e1) and np.array_equal(got["range_off"], ro)


@pytest.mark.parametrize("host_from,thread_kernels", [(1 << 30, False), (2000, False), (1 << 30, True)])
def test_device_decode_equals_host_upload(hostapi, monkeypatch, host_from, thread_kernels):
    """A batch of words: the reference packer's committed streams (both element layouts), random lists of 1..3000 postings (some with array
    indexes from some element on), an empty word.  host_from = 2000 sends the longer streams through the host decoder: same arrays.
    thread_kernels: the one-thread-per-word kernels of round 2 instead of the wavefront-per-word ones (both stay checked)."""
    if thread_kernels:
        monkeypatch.setenv("RXGPU_FT_PACKED_THREAD", "1")
    nf = 4
    z = np.load(FT_GOLDEN)
    rng = np.random.default_rng(21)
    words, want = [], {}
    for wid, name in enumerate(["plain", "arrays"]):
        words.append((wid, z[f"packed_{name}_bytes"], int(min(int(z[f"packed_{name}_afp"]), 1 << 40))))
        want[wid] = dict(doc=z[f"packed_{name}_doc"], pos_off=z[f"packed_{name}_pos_off"], fpos=z[f"packed_{name}_fpos"])
    for wid in range(2, 60):
        n = int(rng.choice([1, 2, 7, 64, 300, 3000]))
        s = make_pos_postings(rng, int(rng.choice([4000, 70_000, 2_000_000])), nf, n, 1.0, array_fields=bool(wid % 3 == 0),
                              max_pos=int(rng.choice([40, 1 << 14, (1 << 28) - 1])))
        data, afp = pack_postings(s["doc"], s["pos_off"], s["fpos"])
        words.append((wid, data, afp))
        want[wid] = s
    words.append((60, np.zeros(0, np.uint8), 0))
    m = hostapi.GpuFtMerger(nf)
    total = 2_000_001
    m.set_docs(np.ones((total, nf), np.float32), np.ones(nf, np.float32), np.zeros(total, np.uint8))
    m.set_words_packed(words, host_from_bytes=host_from)
    for wid, s in want.items():
        _same(m.get_word(wid), s)
    g = m.get_word(60)
    assert g["doc"].shape[0] == 0
    # the same words again through the classic per-word upload: identical device arrays
    m2 = hostapi.GpuFtMerger(nf)
    m2.set_docs(np.ones((total, nf), np.float32), np.ones(nf, np.float32), np.zeros(total, np.uint8))
    for wid in (0, 1, 5, 17):
        m2.set_word_fpos(wid, want[wid])
        a, b = m.get_word(wid), m2.get_word(wid)
        for k in a:
            assert np.array_equal(a[k], b[k]), (wid, k)
    m.close()
    m2.close()


@pytest.mark.parametrize("case", [(31, 2, 30_000, 20000, (1, 1, 2)), (32, 3, 30_000, 150, (1, 1, 1)), (33, 2, 30_000, 20000, (2, 3, 1))])
def test_merges_over_device_decoded_words(hostapi, oracle, case):
    seed, nf, total, limit, ops = case
    ft = FtOracle(oracle)
    _, words, avg, removed, excluded, terms, store = _multi_case(seed, nf, total, limit, ops, seed % 2 == 1, None, sizes=(200, 2500), nsub_range=(1, 4))
    m = hostapi.GpuFtMerger(nf)
    m.set_docs(words, avg, removed)
    m.set_words_packed([(s["word"],) + pack_postings(s["doc"], s["pos_off"], s["fpos"]) for s in store])
    gterms = [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms]
    cfg = ft.default_config(nf, merge_limit=limit, min_rank=5)
    for exc in (None, excluded):
        wd, wp, wf, wn, wpre = ft.merge_query(cfg, terms, total, words, avg, removed, exc, sort_by_rank=False)
        gd, gp, gf, gn, gpre = m.merge_query(cfg, gterms, exc, sort_by_rank=False)
        assert gpre == wpre and np.array_equal(gd, wd.astype(np.int32)) and np.array_equal(gn, wn) and np.array_equal(gf, wf)
        assert np.array_equal(gp.view(np.uint32), wp.view(np.uint32))
    # a single-term (mergeSimple) query over device-decoded words: reads the derived (field, tf, first position) entries
    t0 = terms[0] if terms[0]["op"] != 3 else terms[1]
    osubs = []
    for s in t0["subs"]:
        eo, ef, et, e1, _ = flat_entries(s["doc"], s["pos_off"], s["fpos"])
        osubs.append(dict(doc=s["doc"], ent_off=eo, ent_field=ef, ent_tf=et, ent_first_pos=e1, proc=s["proc"]))
    wd, wp, wf, wn = ft.merge_simple(cfg, t0["opts"], total, words, avg, removed, None, osubs, sort_by_rank=False)
    gd, gp, gf, gn = m.merge(cfg, t0["opts"], [(s["word"], s["proc"]) for s in t0["subs"]], None, sort_by_rank=False)
    assert np.array_equal(gd, wd.astype(np.int32)) and np.array_equal(gp.view(np.uint32), wp.view(np.uint32)) and np.array_equal(gn, wn)
    m.close()


def test_long_lists_through_the_wave_decoder(hostapi):
    """Lists of 40 000 .. 250 000 postings (0.2 .. 1.5 MB of stream: thousands of 256-byte windows, every staging buffer flushed thousands of
    times, range indexes of hundreds of entries), positions up to 2^28 - 1 (5-byte varints across window borders), array indexes starting
    mid-stream, plus a list of ONE posting with 3000 positions (an element spanning a dozen windows)."""
    nf = 3
    rng = np.random.default_rng(77)
    m = hostapi.GpuFtMerger(nf)
    total = 3_000_001
    m.set_docs(np.ones((total, nf), np.float32), np.ones(nf, np.float32), np.zeros(total, np.uint8))
    want, words = {}, []
    for wid, (n, arrays, max_pos) in enumerate([(40_000, False, 40), (120_000, True, (1 << 28) - 1), (250_000, False, 1 << 14)]):
        s = make_pos_postings(rng, total, nf, n, 1.0, array_fields=arrays, max_pos=max_pos)
        data, afp = pack_postings(s["doc"], s["pos_off"], s["fpos"])
        words.append((wid, data, afp))
        want[wid] = s
    from oracle.pyoracle import make_fpos
    pos = np.sort(rng.choice(1 << 20, 3000, replace=False))
    one = dict(doc=np.array([123456], np.uint32), pos_off=np.array([0, 3000], np.uint32),
               fpos=np.sort(make_fpos(pos, rng.integers(0, nf, 3000), np.zeros(3000, np.int64))).astype(np.uint64))
    data, afp = pack_postings(one["doc"], one["pos_off"], one["fpos"])
    words.append((3, data, afp))
    want[3] = one
    m.set_words_packed(words, host_from_bytes=1 << 40)
    for wid, s in want.items():
        _same(m.get_word(wid), s)
    m.close()


def test_malformed_streams_are_refused_loudly(hostapi):
    nf = 2
    m = hostapi.GpuFtMerger(nf)
    m.set_docs(np.ones((1000, nf), np.float32), np.ones(nf, np.float32), np.zeros(1000, np.uint8))
    good, afp = pack_postings([1, 9, 300], [0, 2, 3, 6], [1, 2 | (1 << 56), 7, 1, 5, 9 | (1 << 56)])
    with pytest.raises(Exception, match="word 7"):
        m.set_words_packed([(3, good, afp), (7, good[:-1], afp)])
    bad_field, afp2 = pack_postings([4], [0, 1], [5 | (2 << 56)])
    with pytest.raises(Exception, match="field"):
        m.set_words_packed([(8, bad_field, afp2)])
    dup, afp3 = pack_postings([5, 5], [0, 1, 2], [3, 4])
    with pytest.raises(Exception, match="ascend"):
        m.set_words_packed([(9, dup, afp3)])
    sixbytes = np.array([0x81, 0x80, 0x80, 0x80, 0x80, 0x80, 0x01, 0x04], np.uint8)   # a "varint" of seven bytes
    with pytest.raises(Exception, match="word 11"):
        m.set_words_packed([(11, sixbytes, 1 << 40)])
    m.set_words_packed([(3, good, afp)])
    assert m.get_word(3)["doc"].tolist() == [1, 9, 300]
    m.close()
