import sys, numpy as np
sys.path.insert(0, '.')
from oracle.pyoracle import FtOracle, Oracle
from reindexer_amd import hostapi
from tests.test_bm25_oracle import _multi_case
ft = FtOracle(Oracle())
nf, total = 2, 120_000
_, words, avg, removed, excluded, terms_all, store = _multi_case(4242, nf, total, 700, (1, 1, 2, 1, 3, 1), False, None, sizes=(1500, 9000), nsub_range=(2, 4))
m = hostapi.GpuFtMerger(nf)
m.set_docs(words, avg, removed)
for s in store:
    m.set_word_fpos(s["word"], s)
rng = np.random.default_rng(9)
queries, oracle_terms = [], []
for it in range(12):
    if it % 4 == 3:
        continue
    pick = sorted(rng.choice(len(terms_all), int(rng.integers(1, 4)), replace=False).tolist())
    terms = [terms_all[i] for i in pick]
    if all(t["op"] == 3 for t in terms):
        terms = [terms_all[0]]
    oracle_terms.append(terms)
    queries.append([dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms])
cfg = ft.default_config(nf, merge_limit=700)
hostapi.set_ft_train_mode(1)
qi = int(sys.argv[1]) if len(sys.argv) > 1 else 6
print('ops', [t['op'] for t in oracle_terms[qi]], [len(t['subs']) for t in oracle_terms[qi]], [[(len(s['doc']), s['proc']) for s in t['subs']] for t in oracle_terms[qi]])
try:
    s = m.merge_query(cfg, queries[qi], None, sort_by_rank=False)
    w = ft.merge_query(cfg, oracle_terms[qi], total, words, avg, removed, None, sort_by_rank=False)
    print('ok', np.array_equal(s[0], w[0].astype(np.int32)), len(s[0]))
except Exception as e:
    print('stopped:', str(e)[:80])
