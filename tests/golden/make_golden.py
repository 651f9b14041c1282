"""Generates tests/golden/*.npz from the REAL reference engines (oracle/_ref, built in place from
/root/reference by `make -C oracle ref`).  Run in the build container only:

    RX_TARGET_INSTRUCTIONS=avx512 python tests/golden/make_golden.py

The fixtures pin (a) the plain-C oracle and (b) the HIP kernels on machines where /root/reference does not exist.
Reference entry points exercised: L2SqrDistance / InnerProductDistance (tools/distances), CalculateL2Module
(tools/normalize.cc), BruteforceSearch::{AddPointNoLock,RemovePoint,SearchKnn,SearchRange} (hnswlib/bruteforce.cc).
"""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
os.environ.setdefault("RX_TARGET_INSTRUCTIONS", "avx512")

from oracle.pyoracle import Ref, RefBruteforce, RefHnsw  # noqa: E402

OUT = Path(__file__).resolve().parent
DIMS = [1, 7, 16, 33, 64, 100, 128, 200, 512, 768, 1000]


STREAM_PLANS = [(0, [10, 10, 10]), (16, [5, 40, 1, 300]), (3, [1, 1, 2, 2000])]   # (ef, batch sizes); shared with tests/test_golden.py


def main():
    ref = Ref()
    assert ref.simd_level == 3, "golden vectors must be generated with the AVX-512 dispatch"
    rng = np.random.default_rng(20260924)

    # ---- distances + norms
    dist = {}
    for d in DIMS:
        rows = rng.normal(0, 0.25, (48, d)).astype(np.float32)
        q = rng.normal(0, 0.25, d).astype(np.float32)
        dist[f"rows_{d}"] = rows
        dist[f"q_{d}"] = q
        dist[f"l2_{d}"] = ref.dist_many(0, q, rows)
        dist[f"ip_{d}"] = ref.dist_many(1, q, rows)
        norm_in = rows.copy()
        norm_in[0] = norm_in[0] / np.linalg.norm(norm_in[0])  # unit-vector shortcut
        norm_in[1] = 0
        dist[f"norm_in_{d}"] = norm_in
        dist[f"norm_{d}"] = np.array([ref.l2_module(v) for v in norm_in], np.float32)
    np.savez_compressed(OUT / "distances.npz", **dist)

    # ---- brute-force KNN (with swap-deletes, ties, k > N, range)
    cases = {}
    for name, n, d, gen in (("gauss", 800, 128, lambda s: rng.normal(0, 0.25, s).astype(np.float32)),
                            ("ties", 600, 8, lambda s: rng.integers(-1, 2, s).astype(np.float32))):
        rows = gen((n, d))
        labels = ((rng.permutation(n).astype(np.uint64)) << np.uint64(32)) | rng.integers(0, 3, n).astype(np.uint64)
        victims = rng.choice(n, 25, replace=False)
        queries = gen((12, d))
        cases[f"{name}_rows"] = rows
        cases[f"{name}_labels"] = labels
        cases[f"{name}_victims"] = victims
        cases[f"{name}_queries"] = queries
        for metric in (0, 1, 2):
            bf = RefBruteforce(ref, metric, d, n)
            bf.add(rows, labels)
            for v in victims:
                bf.remove(labels[v])
            for qi in range(queries.shape[0]):
                q = queries[qi]
                if metric == 2:
                    q, _ = ref.normalize_copy(q)
                for k in (1, 10, 64, 100, n):
                    dd, ll = bf.search_knn(q, k)
                    cases[f"{name}_m{metric}_q{qi}_k{k}_dist"] = dd
                    cases[f"{name}_m{metric}_q{qi}_k{k}_label"] = ll
                dd, ll = bf.search_knn(q, n)
                radius = np.float32(dd[20])  # strict '<' excludes the 21st itself (and its ties)
                rd, rl = bf.search_range(q, float(radius))
                cases[f"{name}_m{metric}_q{qi}_radius"] = np.array([radius], np.float32)
                cases[f"{name}_m{metric}_q{qi}_range_dist"] = rd
                cases[f"{name}_m{metric}_q{qi}_range_label"] = rl
            bf.close()
    np.savez_compressed(OUT / "bruteforce.npz", **cases)
    # ---- HNSW: graph built by the real engine (seed 100, sequential inserts) + SearchKnn results, before/after deletes
    n, d, M, efc, metric = 1500, 48, 16, 200, 2
    rows = rng.normal(0, 0.25, (n, d)).astype(np.float32)
    labels = (np.arange(n, dtype=np.uint64) << np.uint64(32)) | np.uint64(1)
    h = RefHnsw(ref, metric, d, n, M=M, ef_construction=efc)
    h.add(rows, labels)
    g = h.export(with_vectors=False)
    hz = dict(rows=rows, labels=labels, metric=np.int64(metric), M=np.int64(M), efc=np.int64(efc), maxlevel=np.int64(g["maxlevel"]),
              entry=np.int64(g["entry"]), links0=g["links0"], levels=g["levels"], upper=g["upper"], upper_off=g["upper_off"])
    queries = rng.normal(0, 0.25, (16, d)).astype(np.float32)
    hz["queries"] = queries
    victims = rng.choice(n, 60, replace=False)
    hz["victims"] = victims
    for phase in (0, 1):
        if phase:
            for v in victims:
                h.mark_delete(labels[v])
        for qi in range(queries.shape[0]):
            qn, _ = ref.normalize_copy(queries[qi])
            for k, ef in ((10, 128), (10, 10), (1, 0), (40, 64)):
                dd, ll = h.search_knn(qn, k, ef)
                hz[f"p{phase}_q{qi}_k{k}_ef{ef}_dist"] = dd
                hz[f"p{phase}_q{qi}_k{k}_ef{ef}_label"] = ll
        # streaming sessions of the real engine (BeginStreamingSearch / ContinueStreamingSearch): every batch, sorted by (dist, label)
        for qi in range(4):
            qn, _ = ref.normalize_copy(queries[qi])
            for si, (sef, plan) in enumerate(STREAM_PLANS):
                sess = h.stream(qn, sef)
                for bi, b in enumerate(plan):
                    dd, ll, ex = sess.next(b)
                    o = np.lexsort((ll, dd))
                    hz[f"s{phase}_q{qi}_p{si}_b{bi}_dist"], hz[f"s{phase}_q{qi}_p{si}_b{bi}_label"] = dd[o], ll[o]
                    hz[f"s{phase}_q{qi}_p{si}_b{bi}_exhausted"] = np.bool_(ex)
                sess.close()
    h.close()
    np.savez_compressed(OUT / "hnsw.npz", **hz)
    make_ann_cache_golden()
    make_ft_goldens()
    print("wrote", [p.name for p in OUT.glob("*.npz")])


def make_ann_cache_golden():
    """The reference's ANN disk cache of a small graph, written by the REAL engine (HierarchicalNSW::SaveIndex behind hnsw.cc:56-62's flag,
    through the in-memory IWriter of oracle/ref/ref_shim.cc): the stream, the rows it refers to, and the graph the engine holds after
    re-loading it — for tests/test_ann_cache.py::test_loader_reads_the_golden_reference_cache on machines without the reference tree."""
    from tests.conftest import make_corpus
    ref = Ref()
    z = {}
    for metric in (0, 2):
        n, d, M, efc = 700, 20, 8, 60
        rows = make_corpus(4242 + metric, n, d)
        labels = (np.arange(n, dtype=np.uint64) << np.uint64(32)) | np.uint64(11)
        r = RefHnsw(ref, metric, d, n, M=M, ef_construction=efc)
        r.add(rows, labels)
        for lab in labels[np.random.default_rng(3).choice(n, 35, replace=False)]:
            r.mark_delete(lab)
        blob = np.frombuffer(r.save_index(), np.uint8)
        e = r.export(with_vectors=False)
        z[f"m{metric}_cache"], z[f"m{metric}_rows"], z[f"m{metric}_labels"] = blob, rows, labels
        z[f"m{metric}_meta"] = np.array([metric, n, d, M, efc, e["maxlevel"], e["entry"], e["num_deleted"]], np.int64)
        for key in ("links0", "levels", "deleted", "upper_off"):
            z[f"m{metric}_{key}"] = e[key]
        z[f"m{metric}_upper"] = e["upper"][:int(e["upper_off"][-1])]
        r.close()
    np.savez_compressed(OUT / "ann_cache.npz", **z)


def make_ft_goldens():
    """ft_fast fixtures from the REAL reference code (oracle/_ref/libref_ft.so): PackedIdRelVec byte streams produced by the reference's
    own packer, and results of the real ft::Merger on small multi-term queries."""
    import sys
    sys.path.insert(0, str(ROOT / "tests"))
    from test_bm25_oracle import MULTI_CASES, _multi_case, make_pos_postings
    from oracle.pyoracle import RefFt
    z = {}
    real = RefFt(3)
    rng = np.random.default_rng(99)
    for name, arr in (("plain", False), ("arrays", True)):
        s = make_pos_postings(rng, 2000, 3, 300, 100.0, array_fields=False, max_pos=3000)
        if arr:   # array data starts in the middle of the stream: the first 300 postings carry none (packWithoutArrayIdxs)
            t = make_pos_postings(rng, 2000, 3, 400, 100.0, array_fields=True, max_pos=3000)
            s = dict(doc=np.concatenate([s["doc"], t["doc"] + np.uint32(2000)]),
                     pos_off=np.concatenate([s["pos_off"], t["pos_off"][1:] + s["pos_off"][-1]]).astype(np.uint32),
                     fpos=np.concatenate([s["fpos"], t["fpos"]]), proc=100.0)
        data, afp = real.pack(s)
        z[f"packed_{name}_bytes"], z[f"packed_{name}_afp"] = data, np.uint64(afp)
        for k in ("doc", "pos_off", "fpos"):
            z[f"packed_{name}_{k}"] = s[k]
    real.close()
    for case in MULTI_CASES[:6]:
        seed, nf, total, limit, ops, arr, fbs = case
        _, words, avg, removed, excluded, terms, store = _multi_case(seed, nf, total, limit, ops, arr, fbs)
        real = RefFt(nf)
        real.set_docs(words, avg, removed)
        for s in store:
            real.set_word_fpos(s["word"], s)
        from oracle.pyoracle import FtOracle
        cfg = FtOracle.default_config(nf, merge_limit=limit)
        real.set_config(cfg)
        rterms = [dict(op=t["op"], opts=t["opts"], subs=[(s["word"], s["proc"]) for s in t["subs"]]) for t in terms]
        wd, wp, wf, wn = real.merge(rterms, excluded, rank_sort_type=1)
        z[f"merge{seed}_doc"], z[f"merge{seed}_proc"], z[f"merge{seed}_field"], z[f"merge{seed}_norm"] = wd, wp, wf, wn
        real.close()
    np.savez_compressed(OUT / "ft.npz", **z)


SQ8_DIMS = [1, 5, 63, 64, 65, 100, 128, 768, 1000, 1536]


def make_sq8_goldens():
    """SQ8 (uint8) distance path from the REAL reference (ref_shim.cc: L2SqrDistance<uint8_t> / InnerProductDistance<uint8_t>,
    Quantizer::Quantize, DistCalculator<uint8_t>): pins oracle/oracle_sq8.c where /root/reference does not exist."""
    from oracle.pyoracle import Sq8Ref
    sr = Sq8Ref(Ref())
    rng = np.random.default_rng(2026)
    z = {}
    for d in SQ8_DIMS:
        a = rng.integers(0, 256, (24, d)).astype(np.uint8)
        b = rng.integers(0, 256, (24, d)).astype(np.uint8)
        a[0], b[0] = 255, 0   # the largest lane sums: the float reduction rounds
        z[f"u8_a_{d}"], z[f"u8_b_{d}"] = a, b
        z[f"u8_l2_{d}"] = np.array([sr.l2sqr_u8(x, y) for x, y in zip(a, b)], np.float32)
        z[f"u8_ip_{d}"] = np.array([sr.ip_u8(x, y) for x, y in zip(a, b)], np.float32)
    for metric in (0, 1, 2):
        for d in (8, 100, 768):
            v = rng.normal(0, 0.25, (16, d)).astype(np.float32)
            v[0, : min(4, d)] = [3.0, -3.0, 0.0, 1e-3][: min(4, d)]   # outliers beyond [minQ, maxQ]: the clamp
            min_q, max_q = float(np.quantile(v, 0.02)), float(np.quantile(v, 0.98))
            p = sr.params(min_q, max_q, d)
            key = f"q_m{metric}_d{d}"
            z[key + "_vec"], z[key + "_minmax"] = v, np.array([min_q, max_q], np.float32)
            z[key + "_params"] = np.array([p["alpha"], p["alpha_2"], p["delta"]], np.float32)
            codes, corr, qcodes, qcorr = [], [], [], []
            for x in v:
                c, o = sr.quantize(metric, p, x)
                codes.append(c)
                corr.append(o)
                c, o = sr.quantize(metric, p, x, 1.25)   # the scaled view prepareData feeds for a query
                qcodes.append(c)
                qcorr.append(o)
            z[key + "_codes"], z[key + "_corr"] = np.stack(codes), np.array(corr, np.float32)
            z[key + "_qcodes"], z[key + "_qcorr"] = np.stack(qcodes), np.array(qcorr, np.float32)
            z[key + "_pair"] = np.array([sr.dist_pair(metric, p, codes[i], corr[i], v[i], codes[i + 1], corr[i + 1], v[i + 1])
                                         for i in range(15)], np.float32)
            z[key + "_query"] = np.array([sr.dist_query(metric, p, qcodes[i], qcorr[i], codes[i + 1], corr[i + 1], v[i + 1])
                                          for i in range(15)], np.float32)
    # the quantised engine end to end: float graph -> HierarchicalNSWImpl<uint8_t> (the copy constructor Quantize() uses) -> SearchKnn
    from oracle.pyoracle import RefHnswQ
    n, d = 1200, 32
    for metric in (0, 2):
        rows = rng.normal(0, 0.25, (n, d)).astype(np.float32)
        labels = (np.arange(n, dtype=np.uint64) << np.uint64(32)) | np.uint64(1)
        h = RefHnsw(Ref(), metric, d, n, M=8, ef_construction=60)
        h.add(rows, labels)
        for lab in labels[rng.choice(n, 40, replace=False)]:
            h.mark_delete(lab)
        g = h.export(with_vectors=False)
        hq = RefHnswQ(h, sample_size=1000)
        sq = hq.export()
        key = f"hq_m{metric}"
        z[key + "_rows"], z[key + "_labels"] = rows, labels
        for k in ("links0", "upper_off", "upper", "levels", "deleted"):
            z[f"{key}_{k}"] = g[k]
        z[key + "_meta"] = np.array([g["n"], g["dim"], g["M"], g["maxM0"], g["maxlevel"], g["entry"], g["num_deleted"]], np.int64)
        z[key + "_params"] = np.array([sq["min_q"], sq["max_q"], sq["alpha"], sq["alpha_2"], sq["delta"]], np.float32)
        z[key + "_codes"], z[key + "_corr"] = sq["codes"], sq["corr"]
        queries = rng.normal(0, 0.25, (12, d)).astype(np.float32)
        norms = np.zeros(12, np.float32)
        if metric == 2:
            ref = Ref()
            for i in range(12):
                queries[i], k_ = ref.normalize_copy(queries[i])
                norms[i] = np.float32(1.0) / np.float32(k_)
        z[key + "_queries"], z[key + "_qnorms"] = queries, norms
        res_d, res_l = [], []
        for i in range(12):
            wd, wl = hq.search_knn(queries[i], 10, 32, float(norms[i]) if metric == 2 else None)
            res_d.append(wd)
            res_l.append(wl)
        z[key + "_res_dist"], z[key + "_res_label"] = np.stack(res_d), np.stack(res_l)
        hq.close()
        h.close()
    np.savez_compressed(OUT / "sq8.npz", **z)


if __name__ == "__main__":
    if sys.argv[1:] == ["sq8"]:
        make_sq8_goldens()
    else:
        main()
        make_sq8_goldens()
