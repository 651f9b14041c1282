#!/usr/bin/env python3
"""HNSW leg (BASELINE configs[2]): cosine, M=16, ef_construction=200, ef=128, k=10 on N x 768.

    python tools/bench_hnsw.py --rows 1000000 --queries 16384 [--out profiles/r2_hnsw_1m.json]
    python tools/bench_hnsw.py --rows 10000000 --no-map-legs --out profiles/r2_hnsw_10m.json      # configs[2] at its true size

Also imported by bench.py (`run(opts)`), which puts the same leg into the driver-run bench line.

The graph is built ON THIS BOX by the product's host builder from `--build-threads` inserting threads (GpuHnswMap<OnInsertions> =
the reference's multithreaded index build, AddPointConcurrent), then
  (a) searched on the MI355X through the C-ABI (rxgpu_hnsw_search_knn, all queries in one batched call; the Map's SearchKnn is the
      nq = 1 case of the same entry point), with the kernel's own counters giving distance evaluations / hops -> algorithmic bytes
      (evals x D x 4 + hops x (1 + 2M) x 4) against the HBM roofline;
  (b) written INTO the reference's own engine (oracle/_ref, ref_hnsw_import_graph: HierarchicalNSWImpl<float> of the reference,
      AVX-512 distances) and searched by ITS SearchKnn on the same graph: single-thread and all-core queries/s
      (cpu_baseline.kind == "reference") and the fraction of queries whose result set equals the GPU's (labels + distance bits);
  (c) recall@k of the GPU result vs exact brute force (the exact scan of the same index on the GPU).
Without oracle/_ref the CPU side falls back to the plain-C restatement (kind "port").
"""
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
os.environ.setdefault("RX_TARGET_INSTRUCTIONS", "avx512")

from reindexer_amd import capi, hostapi  # noqa: E402

DEFAULTS = dict(rows=1_000_000, dim=768, queries=16384, k=10, ef=128, M=16, efc=200, metric="cosine", cpu_queries=1024, recall_queries=10_000,
                map_threads=(1, 0, 64, 256), map_per_thread=64, clusters=2000,
                graph=None, save_graph=None, sq8=True, gpu_only=False, build_threads=0, cpu_threads=0, cpu_per_thread=64, map_legs=True, device=0, out=None, seed=20260924, delete_frac=0.0)


def make_clustered(rows: int, dim: int, clusters: int, seed: int, device: int, return_pick: bool = False):
    """Synthetic corpus = cluster centre + noise (embedding-like: low intrinsic dimension); clusters == 0: i.i.d. N(0, 0.25^2), the
    distribution of the reference's tests (gtests/tools.h:121-129), on which ANY graph index has poor recall at 768 dims.
    Generated on the GPU (a 10M x 768 corpus takes numpy minutes), returned as a host array; deterministic per (seed, device type)."""
    import torch
    dev = torch.device("cuda", device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    centres = torch.empty((max(clusters, 1), dim), dtype=torch.float32, device=dev).normal_(0.0, 0.25, generator=g)
    out = np.empty((rows, dim), np.float32)
    picks = np.zeros(rows, np.int32)
    chunk = 1 << 19
    for a in range(0, rows, chunk):
        n = min(chunk, rows - a)
        if clusters:
            pick = torch.randint(0, clusters, (n,), device=dev, generator=g)
            picks[a:a + n] = pick.cpu().numpy()
            x = centres[pick] + torch.empty((n, dim), dtype=torch.float32, device=dev).normal_(0.0, 0.08, generator=g)
        else:
            x = torch.empty((n, dim), dtype=torch.float32, device=dev).normal_(0.0, 0.25, generator=g)
        out[a:a + n] = x.cpu().numpy()
    return (out, picks) if return_pick else out


def server_times(m):
    """(us on the device, us at the callers) over the queries the Map's resident search kernel has answered so far."""
    import ctypes as C
    a, b = C.c_uint64(0), C.c_uint64(0)
    capi.lib().rxgpu_hnsw_server_times(m.device_index, C.byref(a), C.byref(b))
    return int(a.value), int(b.value)


def sq8_leg(o, ix, metric, g, queries, qnorms, trow, tq, nq, ncpu, ref_float) -> dict:
    """the same graph with SQ8 rows.  The reference's own Quantize() (HierarchicalNSWImpl<uint8_t> copy-built from the float
    engine, sampled minQ / maxQ) produces the codes; the device searches THOSE codes (rxgpu_hnsw_search_knn_sq8) and is compared, labels and
    distance bits, with the quantised engine's SearchKnn; the queries are quantised by the product's host quantiser."""
    from oracle import pyoracle
    t0 = time.perf_counter()
    hq = pyoracle.RefHnswQ(ref_float, sample_size=20000)
    sq = hq.export()
    quant_s = time.perf_counter() - t0
    glabels = g["labels"]
    ix.hnsw_attach_sq8(sq["codes"], sq["corr"], float(sq["alpha_2"]))
    coef = (np.float32(1.0) / qnorms).astype(np.float32) if metric == 2 else np.ones(len(queries), np.float32)
    scales = (np.float32(1.0) / coef).astype(np.float32)   # prepareData: norm = 1.f / normCoef
    qcodes, qcorr = hostapi.sq8_quantize_many(metric, float(sq["min_q"]), float(sq["max_q"]), queries, scales if metric == 2 else None)
    ix.hnsw_search_knn_sq8(qcodes, qcorr, coef, o.k, o.ef)
    ix.hnsw_read_stats()
    ix.profile_enable(True)
    t0 = time.perf_counter()
    dist, row, cnt = ix.hnsw_search_knn_sq8(qcodes, qcorr, coef, o.k, o.ef)
    gpu_s = time.perf_counter() - t0
    launches, kernel_ms = ix.profile_read("hnsw")
    _, redo_ms = ix.profile_read("hnsw_redo")
    _, ties_ms = ix.profile_read("hnsw_ties")
    ix.profile_enable(False)
    evals, hops = ix.hnsw_read_stats()
    tie_reruns = ix.hnsw_read_tie_reruns()
    busy_ms = kernel_ms + redo_ms + ties_ms
    bytes_algo = evals * (o.dim + 4) + hops * (1 + 2 * g["M"]) * 4
    recall = float(np.mean([len(set(trow[i].tolist()) & set(row[i, :int(cnt[i])].tolist())) / o.k for i in range(tq)]))
    t0 = time.perf_counter()
    same = 0
    for i in range(nq):
        wd, wl = hq.search_knn(queries[i], o.k, o.ef, float(qnorms[i]) if metric == 2 else None)
        c = int(cnt[i])
        a = np.lexsort((glabels[row[i, :c]], dist[i, :c]))
        b = np.lexsort((wl, wd))
        same += int(c == len(wl) and np.array_equal(glabels[row[i, :c]][a], wl[b]) and np.array_equal(dist[i, :c][a].view(np.uint32), wd[b].view(np.uint32)))
    cpu_s = time.perf_counter() - t0
    hq.close()
    return {"gpu": {"queries": len(queries), "queries_per_sec": len(queries) / gpu_s, "kernel_ms_total": kernel_ms, "redo_ms": redo_ms,
                    "tie_reruns": tie_reruns, "tie_rerun_ms": ties_ms,
                    "queries_per_sec_kernel_only": len(queries) / (busy_ms / 1e3) if busy_ms else None,
                    "distance_evals_per_query": evals / len(queries), "hops_per_query": hops / len(queries),
                    "roofline": {"bound": "hbm", "kernel": "hnsw_search_kernel<sq8>", "achieved": bytes_algo / (busy_ms / 1e3) / 1e9 if busy_ms else None,
                                 "peak": 8000.0, "unit": "GB/s", "frac": bytes_algo / (busy_ms / 1e3) / 1e9 / 8000.0 if busy_ms else None,
                                 "algorithmic_bytes": bytes_algo, "note": "bytes = evals*(D+4) + hops*(1+2M)*4"}},
            "recall_at_k_vs_exact_float": recall,
            "cpu_baseline": {"kind": "reference", "value": nq / cpu_s, "unit": "queries/s", "cores": 1,
                             "sample": f"{nq} queries through the reference's quantised engine (python call per query included)"},
            "equal_to_reference_frac": same / nq, "equal_to_reference_checked": nq,
            "quantize_seconds_reference": quant_s, "params": {k: float(sq[k]) for k in ("min_q", "max_q", "alpha", "alpha_2", "delta")}}


def run(o) -> dict:
    o = SimpleNamespace(**{**DEFAULTS, **(vars(o) if not isinstance(o, dict) else o)})
    metric = capi.METRICS[o.metric]
    sys.path.insert(0, str(ROOT / "tools"))
    from cpu_scaling import effective_cpus
    ncpu = effective_cpus()   # min(affinity, cgroup CPU quota): what the container may really use
    build_threads = o.build_threads or 2 * ncpu
    t_all = time.perf_counter()
    corpus = make_clustered(o.rows + o.queries, o.dim, o.clusters, o.seed, o.device)
    rows, queries = corpus[:o.rows], corpus[o.rows:]
    labels = np.arange(o.rows, dtype=np.uint64) << np.uint64(32)
    qnorms = np.ones(o.queries, np.float32)   # query_data_norm of the Map's SearchKnn (hnsw_index.cc:168): 1.f / NormalizeCopyVector(...)
    if metric == 2:
        pairs = [hostapi.normalize_copy(q) for q in queries]
        queries = np.stack([a for a, _ in pairs])
        qnorms = (np.float32(1.0) / np.array([b for _, b in pairs], np.float32)).astype(np.float32)
    gen_s = time.perf_counter() - t_all

    m = None
    if o.graph:   # a graph saved by --save-graph on the same corpus (same seed): skip the host build (rocprof passes)
        z = np.load(o.graph)
        meta = z["meta"]
        assert int(meta[0]) == metric and int(meta[1]) == o.rows and int(meta[2]) == o.dim, "graph file is for another corpus"
        order = (z["labels"] >> np.uint64(32)).astype(np.int64)   # internal ids follow arrival order
        vec = rows[order]
        inv = np.array([hostapi.l2_module(r) for r in vec], np.float32) if metric == 2 else None
        g = dict(metric=metric, n=o.rows, dim=o.dim, M=int(meta[3]), maxM0=int(meta[4]), maxlevel=int(meta[5]), entry=int(meta[6]), num_deleted=0,
                 links0=z["links0"], upper_off=z["upper_off"], upper=z["upper"], levels=z["levels"], labels=z["labels"], deleted=z["deleted"],
                 vectors=vec, inv_norms=inv)
        build_s = float(z["build_seconds"])
        build_threads = int(meta[7])
    else:
        t0 = time.perf_counter()
        m = hostapi.GpuHnswMap(metric, o.dim, o.rows, M=o.M, ef_construction=o.efc, multithread=build_threads > 1, device=o.device)
        m.add(rows, labels, threads=build_threads if build_threads > 1 else 0)
        build_s = time.perf_counter() - t0
        g = m.export_graph(with_views=True)   # vectors / 1/|v| in internal-id (= arrival) order, zero-copy views of the builder's storage
        if o.save_graph:
            np.savez(o.save_graph, meta=np.array([metric, o.rows, o.dim, g["M"], g["maxM0"], g["maxlevel"], g["entry"], build_threads], np.int64),
                     links0=g["links0"], upper_off=g["upper_off"], upper=g["upper"], levels=g["levels"], labels=g["labels"], deleted=g["deleted"],
                     build_seconds=np.float64(build_s))
    glabels = g["labels"]
    del corpus, rows
    if o.delete_frac > 0:   # MarkDelete on a random subset (flags only, like hnswalg.h:1303-1339): the search then runs its non-bare form
        dele = np.random.default_rng(o.seed + 1).random(o.rows) < o.delete_frac
        g["deleted"] = np.ascontiguousarray(dele.astype(np.uint8))
        g["num_deleted"] = int(dele.sum())
        o.map_legs = False   # the Map built above knows nothing of these marks (it stays alive: g's vectors are views of its storage)

    # ---- (a) GPU search through the C-ABI
    ix = capi.VectorIndex(metric, o.dim, o.rows, device=o.device)
    ix.upload_rows(0, g["vectors"], g["inv_norms"])
    ix.hnsw_attach_graph(g)
    ix.hnsw_search_knn(queries, o.k, o.ef)   # warm-up at full size (scratch buffers, visited bitsets)
    ix.hnsw_read_stats()
    ix.profile_enable(True)
    t0 = time.perf_counter()
    dist, row, cnt = ix.hnsw_search_knn(queries, o.k, o.ef)
    gpu_s = time.perf_counter() - t0
    launches, kernel_ms = ix.profile_read("hnsw")
    redo_launches, redo_ms = ix.profile_read("hnsw_redo")
    _, ties_ms = ix.profile_read("hnsw_ties")
    ix.profile_enable(False)
    evals, hops = ix.hnsw_read_stats()
    tie_reruns = ix.hnsw_read_tie_reruns()
    bytes_algo = evals * o.dim * 4 + hops * (1 + 2 * g["M"]) * 4
    busy_ms = kernel_ms + redo_ms + ties_ms
    heap_leg = None
    if not o.gpu_only and os.environ.get("RXGPU_HNSW_SORTED", "1") != "0":
        # the same batch on the kernel that replays the reference's binary heaps (what every search used before the sorted list)
        os.environ["RXGPU_HNSW_SORTED"] = "0"
        ix.hnsw_search_knn(queries, o.k, o.ef)
        ix.profile_enable(True)
        t0 = time.perf_counter()
        hd, hr, hc = ix.hnsw_search_knn(queries, o.k, o.ef)
        heap_s = time.perf_counter() - t0
        _, hk_ms = ix.profile_read("hnsw")
        _, hr_ms = ix.profile_read("hnsw_redo")
        ix.profile_enable(False)
        ix.hnsw_read_stats()
        del os.environ["RXGPU_HNSW_SORTED"]
        same = 0
        for i in range(o.queries):
            c = int(cnt[i])
            a, b = np.lexsort((row[i, :c], dist[i, :c])), np.lexsort((hr[i, :c], hd[i, :c]))
            same += int(c == int(hc[i]) and np.array_equal(row[i, :c][a], hr[i, :c][b]) and np.array_equal(dist[i, :c][a].view(np.uint32), hd[i, :c][b].view(np.uint32)))
        heap_leg = {"queries_per_sec": o.queries / heap_s, "kernel_ms_total": hk_ms + hr_ms, "equal_to_sorted_list_frac": same / o.queries}

    # ---- (c) exact ground truth: the exact scan of the SAME index
    tq = min(o.queries, o.recall_queries)   # 10 000 queries
    trow = np.concatenate([ix.search_knn(queries[a:min(a + 1024, tq)], o.k)[1] for a in range(0, tq, 1024)])
    recall = float(np.mean([len(set(trow[i].tolist()) & set(row[i, :int(cnt[i])].tolist())) / o.k for i in range(tq)]))

    out = {
        "deleted_nodes": int(g.get("num_deleted", 0)),
        "workload": f"HNSW {o.metric} M={g['M']} efC={o.efc} ef={o.ef} k={o.k}, {o.rows} x {o.dim} (BASELINE configs[2]"
                    + ("" if o.rows == 10_000_000 else f" scaled to {o.rows} rows") + "), "
                    + (f"{o.clusters} gaussian clusters" if o.clusters else "i.i.d. gaussian"),
        "deviations_from_survey_8d": {
            "corpus": (f"{o.clusters} gaussian clusters (centre N(0, 0.25^2) + N(0, 0.08^2) noise), not the i.i.d. N(0, 0.25^2) rows SURVEY 8(d) names for cfg3: "
                       "on i.i.d. 768-d gaussians every HNSW (the reference's too) answers a small part of the exact neighbours; recall vs the REFERENCE "
                       "engine on the same graph is what north_star bounds (>= 0.99) and is reported as equal_to_reference_frac") if o.clusters else None,
            "build": f"{build_threads} inserting threads (the reference's multithreaded index build, HierarchicalNSWMT), not single-thread insert order 0..N-1: "
                     "a 10M-row single-thread build takes hours; both engines search the SAME graph, so the comparison does not depend on it",
        },
        "corpus_bytes": o.rows * o.dim * 4,
        "build": {"seconds": build_s, "threads": build_threads, "inserts_per_sec": o.rows / build_s if build_s else None,
                  "builder": "rxgpu::host::HnswGraph::AddPointConcurrent (host, the reference's HierarchicalNSWMT build)", "corpus_gen_seconds": gen_s},
        "gpu": {"queries": o.queries, "queries_per_sec": o.queries / gpu_s, "kernel_ms_total": kernel_ms, "launches": launches,
                "queries_per_sec_kernel_only": o.queries / (busy_ms / 1e3) if busy_ms else None,
                "redo_launches": redo_launches, "redo_ms": redo_ms, "tie_reruns": tie_reruns, "tie_rerun_ms": ties_ms, "heap_kernel": heap_leg,
                "distance_evals_per_query": evals / o.queries, "hops_per_query": hops / o.queries,
                "roofline": {"bound": "hbm", "kernel": "hnsw_search_kernel", "achieved": bytes_algo / (busy_ms / 1e3) / 1e9 if busy_ms else None,
                             "peak": 8000.0, "unit": "GB/s", "frac": bytes_algo / (busy_ms / 1e3) / 1e9 / 8000.0 if busy_ms else None,
                             "algorithmic_bytes": bytes_algo, "avg_ms": busy_ms / max(launches, 1),
                             "note": "dependent random 3 KB row gathers; bytes = evals*D*4 + hops*(1+2M)*4 from the kernel's own counters"}},
        "recall_at_k_vs_exact": recall, "recall_queries": tq,
    }
    # the same queries at wider candidate lists: where recall@k vs exact reaches the 0.99 north_star names (profiles/rd6_hnsw_evidence.json has
    # the whole sweep at 1M rows); a failure here never takes the leg down
    try:
        by_ef = {str(o.ef): recall}
        for ef2 in (2 * o.ef, 4 * o.ef):
            _, r2, c2 = ix.hnsw_search_knn(queries[:tq], o.k, ef2)
            by_ef[str(ef2)] = float(np.mean([len(set(trow[i].tolist()) & set(r2[i, :int(c2[i])].tolist())) / o.k for i in range(tq)]))
        out["recall_at_k_vs_exact_by_ef"] = by_ef
    except Exception as e:   # noqa: BLE001
        out["recall_at_k_vs_exact_by_ef"] = {"error": repr(e)}

    if o.gpu_only:
        out["leg_seconds"] = time.perf_counter() - t_all
        ix.close()
        return out

    if m is not None and o.map_legs:
        m.search_knn(queries[0], o.k, o.ef)   # the first search of a Map mirrors the graph into HBM (3 GB at 1M x 768): not a query latency
        for q in queries[1:5]:   # (the first calls launch the resident kernel's first generation and size the call's buffers)
            m.search_knn(q, o.k, o.ef)
        t0 = time.perf_counter()
        for q in queries[5:69]:
            m.search_knn(q, o.k, o.ef)
        out["gpu"]["map_single_query_latency_ms"] = (time.perf_counter() - t0) / 64 * 1e3
        # The reference has no batched API: T planner threads call SearchKnn with one query each (SURVEY 8b "Threading").  The same through
        # GpuHnswMap::SearchKnn, native threads, coalescing on (calls that arrive while the device is busy share one launch); the reference's
        # own T-thread figure over the same graph is cpu_baseline.all_cores below.
        out["gpu"]["map_threads"] = []
        for T in o.map_threads:
            T = T or ncpu
            m.search_knn_mt(queries, o.k, o.ef, T, 2, 10.0)   # warm-up: contexts and buffers of T concurrent callers
            p0 = m.posted_queries()
            tm0 = server_times(m)
            secs, done, batches = m.search_knn_mt(queries, o.k, o.ef, T, o.map_per_thread, 20.0)
            posted = m.posted_queries() - p0   # answered by the resident search kernel: no launch, no batch
            tm1 = server_times(m)
            out["gpu"]["map_threads"].append({"threads": T, "queries": done, "queries_per_sec": done / secs if secs else None, "posted": posted,
                                              "posted_ms_on_device": (tm1[0] - tm0[0]) / posted / 1e3 if posted else None,       # mean duration of a search on the chip
                                              "posted_ms_at_caller": (tm1[1] - tm0[1]) / posted / 1e3 if posted else None,       # ... from the request's store to its answer seen
                                              "device_batches": batches, "avg_batch": (done - posted) / batches if batches else None})
        sess = m.stream(queries[0], o.ef)   # a15: one streaming session, 10 batches of 10 (the planner's post-filter pattern)
        t0 = time.perf_counter()
        got = 0
        for _ in range(10):
            d_, l_, ex_ = sess.next(10)
            got += len(d_)
        out["streaming_session"] = {"batches": 10, "batch": 10, "ef": o.ef, "returned": got, "ms_per_continue": (time.perf_counter() - t0) / 10 * 1e3}
        sess.close()

    # ---- (b) the reference's engine on the SAME graph
    nq = min(o.cpu_queries, o.queries)
    try:
        from oracle import pyoracle
        ref = pyoracle.ref_or_none()
        if ref is not None and ref.simd_level == 3 and hasattr(ref.L, "ref_hnsw_import_graph"):
            r = pyoracle.RefHnsw(ref, metric, o.dim, o.rows, M=g["M"], ef_construction=o.efc)
            t0 = time.perf_counter()
            r.import_graph(g)
            import_s = time.perf_counter() - t0
            t0 = time.perf_counter()
            rd, rl, rc = r.search_knn_many(queries[:nq], o.k, o.ef)
            cpu_s = time.perf_counter() - t0
            same = 0
            for i in range(nq):
                c = int(cnt[i])
                a = np.lexsort((glabels[row[i, :c]], dist[i, :c]))
                b = np.lexsort((rl[i, :int(rc[i])], rd[i, :int(rc[i])]))
                same += int(c == int(rc[i]) and np.array_equal(glabels[row[i, :c]][a], rl[i, :c][b])
                            and np.array_equal(dist[i, :c][a].view(np.uint32), rd[i, :c][b].view(np.uint32)))
            threads = o.cpu_threads or ncpu
            secs, done = r.search_knn_mt(queries, o.k, o.ef, threads, o.cpu_per_thread, deadline_s=30.0)
            out["cpu_baseline"] = {"kind": "reference", "value": nq / cpu_s, "unit": "queries/s", "cores": 1,
                                   "sample": f"{nq} queries, measured; the reference's HierarchicalNSWImpl<float>::SearchKnn (AVX-512) on the SAME graph "
                                             f"(written into the engine by ref_hnsw_import_graph in {import_s:.1f} s)",
                                   "all_cores": {"value": done / secs, "cores": threads, "queries": done,
                                                 "note": "T threads, one query each at a time over the shared index; thread start outside the timed region"}}
            out["equal_to_reference_frac"] = same / nq
            out["equal_to_reference_checked"] = nq
            if o.sq8:
                out["sq8"] = sq8_leg(o, ix, metric, g, queries, qnorms, trow, tq, nq, ncpu, ref_float=r)
            r.close()
        else:
            from oracle.pyoracle import Oracle, oracle_hnsw_search_knn
            orc = Oracle()
            t0 = time.perf_counter()
            res = [oracle_hnsw_search_knn(orc, g, queries[i], o.k, o.ef, g["inv_norms"]) for i in range(nq)]
            cpu_s = time.perf_counter() - t0
            same = sum(int(np.array_equal(np.sort(glabels[row[i, :int(cnt[i])]]), np.sort(res[i][1]))) for i in range(nq))
            out["cpu_baseline"] = {"kind": "port", "value": nq / cpu_s, "unit": "queries/s", "cores": 1, "sample": f"{nq} queries, same graph"}
            out["equal_to_reference_frac"] = same / nq
    except Exception as e:  # the GPU numbers are still reported
        out["cpu_baseline"] = {"error": repr(e)}
    out["leg_seconds"] = time.perf_counter() - t_all
    ix.close()
    if m is not None:
        m.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    for k, v in DEFAULTS.items():
        if k == "map_legs":
            ap.add_argument("--no-map-legs", dest="map_legs", action="store_false")
        elif k == "gpu_only":
            ap.add_argument("--gpu-only", dest="gpu_only", action="store_true", help="stop after the GPU search (rocprof passes)")
        elif k == "sq8":
            ap.add_argument("--no-sq8", dest="sq8", action="store_false")
        elif isinstance(v, tuple):
            ap.add_argument("--" + k.replace("_", "-"), type=lambda t: tuple(int(x) for x in t.split(",")), default=v)
        elif v is None or isinstance(v, str):
            ap.add_argument("--" + k.replace("_", "-"), default=v)
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
