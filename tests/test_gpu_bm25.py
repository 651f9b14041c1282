"""-m gpu: ft_fast BM25 merge on the GPU (ft_merge.hip through rxgpu_ft_* and GpuFtMerger) vs the CPU restatement that is pinned
by the reference's golden debug_rank strings.  Bar: id set equal, uint8 rank equal — here the raw float ranks are
bit-identical as well, because the kernel keeps the reference's fp64/fp32 types operation for operation."""
import numpy as np
import pytest

from oracle.pyoracle import FtOracle
from .test_bm25_oracle import AVG, KATS, N_DOCS, f32, make_postings

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ft(oracle):
    return FtOracle(oracle)


@pytest.fixture(scope="module")
def hostapi(rxgpu):
    from reindexer_amd import hostapi as h
    h.lib()
    return h


def test_gpu_reproduces_reference_golden_ranks(hostapi):
    """The reference's own KATs (ft_generic.cc:326-443) through the GPU: one posting per KAT, rank read back raw."""
    total = N_DOCS + 1
    for kat in KATS:
        m = hostapi.GpuFtMerger(1)
        words = np.full((total, 1), kat["words"], np.float32)
        words[0] = 0
        m.set_docs(words, [AVG])
        # one dictionary word with M postings; doc 1 carries the KAT's tf / first position
        docs = np.arange(1, 1 + kat["M"], dtype=np.uint32)
        s = dict(doc=docs, ent_off=np.arange(kat["M"] + 1, dtype=np.uint32), ent_field=np.zeros(kat["M"], np.uint8),
                 ent_tf=np.full(kat["M"], kat["tf"], np.uint32), ent_first_pos=np.full(kat["M"], kat["pos"], np.uint32))
        m.set_word_flat(7, s)
        cfg = hostapi.default_ft_config(1, min_rank=0, full_match_boost=1.0)
        opts = hostapi.default_ft_opts(1, term_len_boost=np.float32(kat["tlb_in"]))
        ids, proc, field, norm = m.merge(cfg, opts, [(7, np.float32(kat["proc"]))], sort_by_rank=False)
        assert ids[0] == 1
        want = f32(kat["term_rank"])
        # proc comes back normalised (uint8 of the raw rank, all ranks here are < 255)
        assert norm[0] == np.uint8(want), (kat, proc[0])
        m.close()


@pytest.mark.parametrize("nf,limit", [(1, 20000), (3, 20000), (3, 150), (5, 40)])
def test_gpu_merge_equals_restated_merger(hostapi, ft, nf, limit):
    rng = np.random.default_rng(nf * 100 + limit)
    total = 3000
    words = rng.integers(1, 40, (total, nf)).astype(np.float32)
    words[0] = 0
    avg = words[1:].mean(axis=0).astype(np.float32)
    removed = np.zeros(total, np.uint8)
    removed[rng.choice(total, 100, replace=False)] = 1
    excluded = np.zeros(total, np.uint8)
    excluded[rng.choice(total, 100, replace=False)] = 1
    m = hostapi.GpuFtMerger(nf)
    m.set_docs(words, avg, removed)
    subs = []
    for wid, proc in enumerate((100.0, 88.5, 77.0, 61.0)):
        s = make_postings(rng, total, nf, int(rng.integers(200, 1500)))
        s["proc"] = proc
        subs.append(s)
        m.set_word_flat(wid, s)
    fb = [1.0, 0.7, 0.0, 1.3, 0.2][:nf]
    for variant in range(3):
        cfg = ft.default_config(nf, merge_limit=limit)
        opts = ft.default_opts(nf, field_boost=fb, boost=1.0 + 0.25 * variant, term_len_boost=0.8)
        if variant == 2 and nf > 1:
            cfg["summation_ratio"] = 0.5
            opts["need_sum_rank"] = [1] * nf
        for exc in (None, excluded):
            for sort_by_rank in (False, True):
                wd, wp, wf, wn = ft.merge_simple(cfg, opts, total, words, avg, removed, exc, subs, sort_by_rank=sort_by_rank)
                gd, gp, gf, gn = m.merge(cfg, opts, [(i, s["proc"]) for i, s in enumerate(subs)], excluded=exc, sort_by_rank=sort_by_rank)
                assert np.array_equal(gd.astype(np.uint32), wd), (nf, limit, variant, sort_by_rank)
                assert np.array_equal(gn, wn) and np.array_equal(gf, wf)
                assert np.array_equal(gp.view(np.uint32), wp.view(np.uint32))
    m.close()


def test_gpu_merge_large_posting_lists(hostapi, ft):
    """Posting lists that span many scan blocks and exceed mergeLimit: the admission cut must fall at the same document."""
    rng = np.random.default_rng(9)
    total, nf = 200_000, 2
    words = rng.integers(1, 60, (total, nf)).astype(np.float32)
    words[0] = 0
    avg = words[1:].mean(axis=0).astype(np.float32)
    m = hostapi.GpuFtMerger(nf)
    m.set_docs(words, avg)
    subs = []
    for wid, (proc, n) in enumerate(((100.0, 60_000), (90.0, 45_000), (70.0, 3_000))):
        s = make_postings(rng, total, nf, n)
        s["proc"] = proc
        subs.append(s)
        m.set_word_flat(wid, s)
    cfg, opts = ft.default_config(nf), ft.default_opts(nf)
    wd, wp, wf, wn = ft.merge_simple(cfg, opts, total, words, avg, None, None, subs, sort_by_rank=False)
    gd, gp, gf, gn = m.merge(cfg, opts, [(i, s["proc"]) for i, s in enumerate(subs)], sort_by_rank=False)
    assert gd.shape[0] == wd.shape[0] <= 20000
    assert np.array_equal(gd.astype(np.uint32), wd) and np.array_equal(gn, wn) and np.array_equal(gf, wf)
    postings, ms = m.read_stats()
    assert postings == 108_000 and ms > 0
    m.close()


def test_more_than_eight_summed_fields_is_refused_loudly(hostapi):
    """The GPU engine keeps at most 8 per-field ranks for summationRanksByFieldsRatio: more is an error, never a silent truncation."""
    nf = 10
    m = hostapi.GpuFtMerger(nf)
    total = 50
    words = np.ones((total, nf), np.float32)
    m.set_docs(words, np.ones(nf, np.float32))
    rng = np.random.default_rng(1)
    m.set_word_flat(0, make_postings(rng, total, nf, 20))
    cfg = hostapi.default_ft_config(nf, summation_ratio=0.5)
    with pytest.raises(Exception):
        m.merge(cfg, hostapi.default_ft_opts(nf, need_sum_rank=[1] * nf), [(0, 100.0)])
    ok = m.merge(cfg, hostapi.default_ft_opts(nf, need_sum_rank=[1] * 8 + [0, 0]), [(0, 100.0)])
    assert len(ok[0]) > 0
    m.close()


@pytest.mark.parametrize("bm25_type", ["classic", "word_count"])
@pytest.mark.parametrize("nf,limit", [(1, 20000), (3, 150)])
def test_gpu_merge_bm25_classic_and_word_count(hostapi, ft, bm25_type, nf, limit):
    """FTConfig::Bm25Config::bm25Type = classic / wordCount (bm25.h:38-68) evaluated ON THE DEVICE: same documents, order, raw-rank bits and
    uint8 ranks as the restatement, which tests/test_bm25_oracle.py pins against the real Merge<Bm25Classic> / Merge<TermCount>."""
    rng = np.random.default_rng(nf * 7 + limit + len(bm25_type))
    total = 2500
    words = rng.integers(1, 40, (total, nf)).astype(np.float32)
    words[0] = 0
    avg = words[1:].mean(axis=0).astype(np.float32)
    m = hostapi.GpuFtMerger(nf)
    m.set_docs(words, avg)
    subs = []
    for wid, proc in enumerate((100.0, 88.5, 61.0)):
        s = make_postings(rng, total, nf, int(rng.integers(200, 1200)))
        s["proc"] = proc
        subs.append(s)
        m.set_word_flat(wid, s)
    cfg = ft.default_config(nf, merge_limit=limit, bm25_type=bm25_type)
    opts = ft.default_opts(nf, field_boost=[1.0, 0.7, 1.3][:nf], term_len_boost=0.8)
    for sort_by_rank in (False, True):
        wd, wp, wf, wn = ft.merge_simple(cfg, opts, total, words, avg, None, None, subs, sort_by_rank=sort_by_rank)
        gd, gp, gf, gn = m.merge(cfg, opts, [(i, s["proc"]) for i, s in enumerate(subs)], sort_by_rank=sort_by_rank)
        assert np.array_equal(gd.astype(np.uint32), wd) and np.array_equal(gn, wn) and np.array_equal(gf, wf)
        assert np.array_equal(gp.view(np.uint32), wp.view(np.uint32))
    m.close()
