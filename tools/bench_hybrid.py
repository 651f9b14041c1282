#!/usr/bin/env python3
"""BASELINE configs[4]: hybrid query = ft_fast BM25 merge over a 5M-document inverted index + cosine KNN over 5M x 512 vectors (k = 100),
fused with RRF (rank_const 60) — the pipeline of hybrid.md's `ORDER BY RRF()` on one MI355X.

    python tools/bench_hybrid.py --docs 5000000 --dim 512 --queries 20 [--out profiles/r2_hybrid_5m.json]

Also imported by bench.py (`run(opts)`), which puts the same leg into the driver-run bench line.

Per query: 1-3 query words (each with an exact and a stem variant, document frequencies 10 % / 3 % / 1 % / 0.3 %), one query vector.
GPU: rxgpu::host::HybridQueryResident (hybrid_query.h) — the Map's search and the Merger's merge run on their own streams with their
results LEFT IN HBM, one kernel applies postProcessResults + the rank fusion there (hybrid_fuse.hip), one list of (row id, rank) comes
back.  The same queries also run through the separate product calls (GpuFtMerger::MergeQuery, GpuBruteforceMap::select, host fusion
hybrid_rerank.h) — the round-2 path — for the per-half parity against the reference's engines and as the comparison figure.
CPU baseline (cpu_baseline.kind "reference" when oracle/_ref is there): the SAME pipeline from the reference's own code compiled in
place — ft::Merger (libref_ft.so), hnswlib::BruteforceSearch over the FULL corpus (libref_oracle.so, AVX-512) and
SelectIteratorContainer::MergerRankedImpl (libref_rank.so) — one core, `cpu_queries` queries, measured (no scaling).
Parity on those queries: FT documents + ranks, KNN ids + distance bits, fused ids + rank bits, each half against the reference's."""
import argparse
import json
import os
import sys
import time
from pathlib import Path
from types import SimpleNamespace

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
os.environ.setdefault("RX_TARGET_INSTRUCTIONS", "avx512")

from bench_bm25 import pos_postings  # noqa: E402
from reindexer_amd import hostapi  # noqa: E402

DEFAULTS = dict(docs=5_000_000, dim=512, queries=64, k=100, cpu_queries=64, device=0, out=None)


def run(o) -> dict:
    import torch
    o = SimpleNamespace(**{**DEFAULTS, **(vars(o) if not isinstance(o, dict) else o)})
    t_all = time.perf_counter()
    rng = np.random.default_rng(20260926)
    total = o.docs + 1   # vdoc 0 = the empty sentinel; vdoc i <-> vector row i
    # ---- full-text side
    words = rng.integers(20, 61, (total, 1)).astype(np.float32)
    words[0] = 0
    avg = words[1:].mean(axis=0).astype(np.float32)
    ftm = hostapi.GpuFtMerger(1, device=o.device)
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
    # ---- vector side: generated on the GPU, handed to the Map in chunks (host mirror + HBM upload, like upserts)
    t0 = time.perf_counter()
    dev = torch.device("cuda", o.device)
    g = torch.Generator(device=dev)
    g.manual_seed(20260926)
    vm = hostapi.GpuBruteforceMap(2, o.dim, total, device=o.device)
    labels = np.arange(total, dtype=np.uint64) << np.uint64(32)
    rows_host = np.empty((total, o.dim), np.float32)
    step = 500_000
    for a in range(0, total, step):
        b = min(total, a + step)
        chunk = torch.empty((b - a, o.dim), dtype=torch.float32, device=dev).normal_(0.0, 0.25, generator=g).cpu().numpy()
        if a == 0:
            chunk[0] = 1.0
        rows_host[a:b] = chunk
        vm.add(chunk, labels[a:b])
    load_s = time.perf_counter() - t0
    keys = torch.empty((o.queries, o.dim), dtype=torch.float32, device=dev).normal_(0.0, 0.25, generator=g).cpu().numpy()
    cfg, opts = hostapi.default_ft_config(1), hostapi.default_ft_opts(1)
    plans = []
    for q in range(o.queries):
        nw = int(rng.integers(1, 4))
        plans.append([int(x) for x in rng.choice(len(vocab), nw, replace=False)])

    def terms_of(q):
        return [dict(op=1, opts=opts, subs=[(w, s["proc"]) for w, s in vocab[p]]) for p in plans[q]]

    def run_split(q):
        """the separate product calls: FT result and KNN result come to the host, fused there"""
        plan = plans[q]
        t = [time.perf_counter()]
        if len(plan) == 1:
            fid, fproc, _, _ = ftm.merge(cfg, opts, [(w, s["proc"]) for w, s in vocab[plan[0]]], sort_by_rank=True)
        else:
            fid, fproc, _, _, _ = ftm.merge_query(cfg, terms_of(q), sort_by_rank=True)
        t.append(time.perf_counter())
        kid, krank = vm.select(keys[q], k=o.k, need_sort=False)
        t.append(time.perf_counter())
        ids, ranks = hostapi.merge_ranked("rrf", [60.0], kid, krank, fid, fproc, union=True, desc=True, metric=2, ft_order="rank")
        t.append(time.perf_counter())
        return (fid, fproc, kid, krank, ids, ranks), np.diff(t)

    def run_resident(q):
        return hostapi.hybrid_query_resident(vm, ftm, cfg, terms_of(q), keys[q], o.k, kind="rrf", params=[60.0], union=True, desc=True)

    run_split(0)   # warm-up (device sync of the Map, buffers)
    run_resident(0)
    ftm.read_stats()
    t0 = time.perf_counter()
    parts = np.zeros(3)
    results = []
    for q in range(o.queries):
        r, dt = run_split(q)
        results.append(r)
        parts += dt
    split_s = time.perf_counter() - t0
    split_postings, split_ft_ms = ftm.read_stats()   # the merges of the split leg ran alone on the device: their kernel time is the FT half's own
    # the same FT merges as ONE launch train (GpuFtMerger::MergeQueryBatch: the query is the kernels' second grid dimension)
    all_terms = [terms_of(q) for q in range(o.queries)]
    ftm.merge_query_batch(cfg, all_terms, sort_by_rank=True)   # warm-up: the batch lanes' scratch
    ftm.read_stats()
    t0 = time.perf_counter()
    batch_res = ftm.merge_query_batch(cfg, all_terms, sort_by_rank=True)
    batch_wall = time.perf_counter() - t0
    batch_postings, batch_ms = ftm.read_stats()
    batch_same = sum(int(np.array_equal(b[0], r[0]) and np.array_equal(b[1], r[1])) for b, r in zip(batch_res, results))
    ftm.read_fuse_stats()
    t0 = time.perf_counter()
    fused = [run_resident(q) for q in range(o.queries)]
    gpu_s = time.perf_counter() - t0
    ft_postings, ft_kernel_ms = ftm.read_stats()
    fuse_calls, fuse_kernel_ms, prep_kernel_ms = ftm.read_fuse_stats()
    same_as_split = sum(int(np.array_equal(f[0], r[4]) and np.array_equal(f[1].view(np.uint32), r[5].view(np.uint32))) for f, r in zip(fused, results))
    out = {"workload": f"hybrid RRF: ft_fast BM25 (1-3 OR terms x 2 sub-terms) over {o.docs} vdocs + cosine KNN k={o.k} over {o.docs} x {o.dim}, union fusion "
                       "(BASELINE configs[4])",
           "load_seconds": load_s,
           "gpu": {"path": "resident: KNN list and FT merge left in HBM, postProcessResults + rank fusion on the device (hybrid_fuse.hip), one download",
                   "queries": o.queries, "queries_per_sec": o.queries / gpu_s, "ms_per_query": gpu_s / o.queries * 1e3,
                   "ms_fusion": fuse_kernel_ms / max(fuse_calls, 1), "fusion_kernel_launches": fuse_calls,
                   "ms_fusion_note": "hybrid_join_kernel: what is left to do once both halves are there (critical path)",
                   "ms_fusion_prepare_overlapped": prep_kernel_ms / max(fuse_calls, 1),
                   "ms_fusion_prepare_note": "hybrid_prepare_kernel (FT only: postProcessResults, id sort, rank classes): enqueued behind the merge, runs "
                                             "while the KNN scan streams the corpus",
                   "ms_ft_kernels": ft_kernel_ms / o.queries, "ft_postings_per_query": ft_postings / o.queries,
                   "boundary_ties_redone_on_host": int(sum(int(f[2]) for f in fused)),
                   "fused_results_avg": float(np.mean([len(f[0]) for f in fused])),
                   "identical_to_split_path_frac": same_as_split / o.queries},
           "ft_half": {"kernel": "ft_ranges / ft_rank_all / ft_adders / ft_finish (+ ft_preselect_apply): the train of ALL the leg's queries at once "
                                 "(MergeQueryBatch: grid.y = query), alone on the device",
                       "queries_per_train": o.queries, "ms_kernels_per_train": batch_ms, "ms_kernels_per_merge": batch_ms / o.queries,
                       "ms_wall_per_merge": batch_wall / o.queries * 1e3, "postings_per_merge": batch_postings / o.queries,
                       "identical_to_single_merges_frac": batch_same / o.queries,
                       "single_merge": {"ms_kernels_per_merge": split_ft_ms / o.queries,
                                        "frac": split_postings * 20 / (split_ft_ms / 1e3) / 1e9 / 8000.0 if split_ft_ms else None,
                                        "what": "the same merges one train each (the split leg)"},
                       "roofline": {"bound": "hbm", "unit": "GB/s", "peak": 8000.0,
                                    "achieved": batch_postings * 20 / (batch_ms / 1e3) / 1e9 if batch_ms else None,
                                    "frac": batch_postings * 20 / (batch_ms / 1e3) / 1e9 / 8000.0 if batch_ms else None,
                                    "bytes_per_posting": 20,
                                    "note": "SURVEY 8d's 20 B per posting; the PMC passes of the 3 x 3 merge (profiles/rd3b_bm25_rocprof.json, FETCH_SIZE with the "
                                            "gfx950 x2 correction) count 89 MB fetched + 16 MB written for 3.9 M postings = 27 B per posting = 1.34 x this model; "
                                            "in the resident leg the same kernels share the device with the KNN scan (ms_ft_kernels there is their "
                                            "stretched elapsed time, not their cost)"}},
           "gpu_split": {"path": "round-2 path: both halves downloaded, fused on the host (hybrid_rerank.h)", "queries_per_sec": o.queries / split_s,
                         "ms_per_query": split_s / o.queries * 1e3, "ms_ft_merge": parts[0] / o.queries * 1e3, "ms_knn_select": parts[1] / o.queries * 1e3,
                         "ms_fusion": parts[2] / o.queries * 1e3}}
    nq = min(o.cpu_queries, o.queries)
    try:
        from oracle import pyoracle
        ref = pyoracle.ref_or_none()
        rft = pyoracle.ref_ft_or_none(1)
        rrank = pyoracle.ref_rank_or_none()
        if ref is None or ref.simd_level != 3 or rft is None or rrank is None:
            raise RuntimeError("oracle/_ref incomplete (needs libref_oracle.so with AVX-512, libref_ft.so, libref_rank.so)")
        rft.set_docs(words, avg, None)
        for pair in vocab:
            for w, s in pair:
                rft.set_word_fpos(w, s)
        rft.set_config(cfg)
        rb = pyoracle.RefBruteforce(ref, 2, o.dim, total)
        rb.add(rows_host, labels)
        orc = pyoracle.Oracle()
        t_ft = t_knn = t_fuse = 0.0
        same_ft = same_knn = same_fused = 0
        for q in range(nq):
            plan = plans[q]
            terms = [dict(op=1, opts=opts, subs=[(w, s["proc"]) for w, s in vocab[p]]) for p in plan]
            t0 = time.perf_counter()
            rd, rp, rf, rn = rft.merge(terms, None, rank_sort_type=1, cap=1 << 17)
            t1 = time.perf_counter()
            qn, _ = orc.normalize_copy(keys[q])      # HnswIndexBase::search normalises the key for cosine (hnsw_index.cc:166-173)
            kd, kl = rb.search_knn(qn, o.k)
            t2 = time.perf_counter()
            fid, fproc, kid, krank, ids, ranks = results[q]
            # FT half: same documents with the same uint8 ranks
            a, b = np.argsort(fid, kind="stable"), np.argsort(rd, kind="stable")
            ft_ok = np.array_equal(fid[a].astype(np.uint32), rd[b].astype(np.uint32)) and np.array_equal(fproc[a].astype(np.float32), rn[b].astype(np.float32))
            # KNN half: row ids best-first; ranks = -distance for cosine (selectRaw, hnsw_index.cc:205-222)
            knn_ok = np.array_equal(kid.astype(np.int64), (kl >> np.uint64(32)).astype(np.int64)) and np.array_equal(
                krank.astype(np.float32).view(np.uint32), (-kd).astype(np.float32).view(np.uint32))
            # fusion by the reference's merger, fed with the reference's two halves
            t3 = time.perf_counter()
            by_id = np.argsort(rd, kind="stable")
            order = np.argsort(-rn.astype(np.float32), kind="stable")
            pos = np.zeros(rd.shape[0], np.uint64)
            pos[order] = rrank.rrf_positions(rn.astype(np.float32)[order])
            wi, wr = rrank.merge("rrf", [60.0], (kl >> np.uint64(32)).astype(np.int32), (-kd).astype(np.float32), rd[by_id].astype(np.int32),
                                 rn[by_id].astype(np.float32), union=True, desc=True, metric=2, ft_positions=pos[by_id])
            t4 = time.perf_counter()
            fused_ok = np.array_equal(wi, fused[q][0]) and np.array_equal(wr.view(np.uint32), fused[q][1].view(np.uint32))   # the RESIDENT path's list
            same_ft += int(ft_ok)
            same_knn += int(knn_ok)
            same_fused += int(fused_ok)
            t_ft += t1 - t0
            t_knn += t2 - t1
            t_fuse += t4 - t3
        per_q = (t_ft + t_knn + t_fuse) / nq
        out["cpu_baseline"] = {"kind": "reference", "cores": 1, "unit": "queries/s", "value": 1.0 / per_q, "ms_ft_merge": t_ft / nq * 1e3,
                               "ms_knn_scan": t_knn / nq * 1e3, "ms_fusion": t_fuse / nq * 1e3,
                               "sample": f"{nq} of the same queries, measured (full {o.docs}-row corpus, no scaling): reference ft::Merger + "
                                         "BruteforceSearch (AVX-512) + MergerRankedImpl, compiled in place (oracle/_ref)"}
        out["parity"] = {"checked": nq, "ft_identical_frac": same_ft / nq, "knn_identical_frac": same_knn / nq, "fused_identical_frac": same_fused / nq,
                         "against": "reference ft::Merger / BruteforceSearch / MergerRankedImpl; the fused list checked is the device fusion's "
                                    "(resident path), the two halves are the split path's"}
        rb.close()
        rft.close()
    except Exception as e:
        out["cpu_baseline"] = {"error": repr(e)}
    out["leg_seconds"] = time.perf_counter() - t_all
    ftm.close()
    vm.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    for k, v in DEFAULTS.items():
        if v is None:
            ap.add_argument("--" + k.replace("_", "-"), default=None)
        else:
            ap.add_argument("--" + k.replace("_", "-"), type=type(v), default=v)
    args = ap.parse_args()
    out = run(args)
    text = json.dumps(out)
    print(text)
    if args.out:
        Path(args.out).write_text(text + "\n")


if __name__ == "__main__":
    main()
