import sys, numpy as np
sys.path.insert(0, '.')
from oracle.pyoracle import FtOracle, Oracle
from reindexer_amd import hostapi
from tests.test_bm25_oracle import _multi_case
ft = FtOracle(Oracle())
nf, total = 2, 120_000
_, words, avg, removed, excluded, terms_all, store = _multi_case(4242, nf, total, 700, (1, 1, 2, 1, 3, 1), False, None, sizes=(1500, 9000), nsub_range=(2, 4))
wide = _multi_case(4243, nf, total, 700, (1, 1), False, None, sizes=(300, 900), nsub_range=(10, 12))
for s in wide[6]:
    s["word"] += 1000
m = hostapi.GpuFtMerger(nf)
m.set_docs(words, avg, removed)
for s in store + wide[6]:
    m.set_word_fpos(s["word"], s)
rng = np.random.default_rng(9)
queries, oracle_terms = [], []
for it in range(12):
    if it % 4 == 3:
        terms = wide[5]
    else:
        pick = sorted(rng.choice(len(terms_all), int(rng.integers(1, 4)), replace=False).tolist())
        terms = [terms_all[i] for i in pick]
        if all(t["op"] == 3 for t in terms):
            terms = [terms_all[0]]
    oracle_terms.append(terms)
    queries.append([dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms])
cfg = ft.default_config(nf, merge_limit=700)
hostapi.set_ft_train_mode(1)
got = m.merge_query_batch(cfg, queries, sort_by_rank=False)
for i, (terms, g, q) in enumerate(zip(oracle_terms, got, queries)):
    w = ft.merge_query(cfg, terms, total, words, avg, removed, None, sort_by_rank=False)
    s = m.merge_query(cfg, q, None, sort_by_rank=False)
    print(i, 'ops', [t['op'] for t in terms], 'nsubs', sum(len(t['subs']) for t in terms), 'batch==oracle', np.array_equal(g[0], w[0].astype(np.int32)), len(g[0]), len(w[0]),
          'single==oracle', np.array_equal(s[0], w[0].astype(np.int32)), len(s[0]), 'pre', g[4], w[4])
print('---- fresh merger, query 8 and 1 alone')
m2 = hostapi.GpuFtMerger(nf)
m2.set_docs(words, avg, removed)
for s in store + wide[6]:
    m2.set_word_fpos(s["word"], s)
for qi in (8, 8, 0, 2):
    w = ft.merge_query(cfg, oracle_terms[qi], total, words, avg, removed, None, sort_by_rank=False)
    hostapi.set_ft_train_mode(1)
    s = m2.merge_query(cfg, queries[qi], None, sort_by_rank=False)
    hostapi.set_ft_train_mode(0)
    d = m2.merge_query(cfg, queries[qi], None, sort_by_rank=False)
    wd = w[0].astype(np.int32)
    print(qi, 'sparse==oracle', np.array_equal(s[0], wd), 'dense==oracle', np.array_equal(d[0], wd), 'same set', np.array_equal(np.sort(s[0]), np.sort(wd)), len(s[0]), len(wd))
    if not np.array_equal(s[0], wd):
        n = min(len(s[0]), len(wd))
        bad = np.flatnonzero(s[0][:n] != wd[:n])
        print('  first mismatch at', bad[:5], s[0][bad[:5]], wd[bad[:5]], 'missing', np.setdiff1d(wd, s[0])[:10], 'extra', np.setdiff1d(s[0], wd)[:10])
        print('  train stats', m2.read_train_stats())
