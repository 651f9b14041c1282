#!/usr/bin/env python3
"""How long one rxgpu_hnsw_search_knn call takes as a function of the number of queries in it, and what the Map's query coalescer makes of
T planner threads for several numbers of device batches in flight (GpuHnswMap::SetCoalescerLanes).  One graph, built once.
    python tools/bench_hnsw_nq_sweep.py [--rows 1000000] [--out gpurun_out/hnsw_nq_sweep.json]"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
from reindexer_amd import capi, hostapi   # noqa: E402
import bench_hnsw   # noqa: E402
from cpu_scaling import effective_cpus   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--ef", type=int, default=128)
    ap.add_argument("--lanes", default="1,2,3,4,8,16")
    ap.add_argument("--threads", default="16,64,256")
    ap.add_argument("--short", action="store_true", help="four call sizes only")
    ap.add_argument("--blocking-sync", action="store_true", help="hipDeviceScheduleBlockingSync: host waits do not spin")
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "hnsw_nq_sweep.json"))
    a = ap.parse_args()
    if a.blocking_sync:   # hipDeviceScheduleBlockingSync before the first HIP call of the process: waits sleep on an interrupt instead of spinning
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        print("hipSetDeviceFlags(hipDeviceScheduleBlockingSync) ->", hip.hipSetDeviceFlags(ctypes.c_uint(4)), flush=True)
    nqmax = 4096
    corpus = bench_hnsw.make_clustered(a.rows + nqmax, a.dim, 2000, 20260924, 0)
    rows, queries = corpus[:a.rows], corpus[a.rows:]
    pairs = [hostapi.normalize_copy(q) for q in queries]
    queries = np.stack([p for p, _ in pairs])
    labels = np.arange(a.rows, dtype=np.uint64) << np.uint64(32)
    threads = 2 * effective_cpus()
    t0 = time.perf_counter()
    m = hostapi.GpuHnswMap(2, a.dim, a.rows, M=16, ef_construction=200, multithread=True, device=0)
    m.add(rows, labels, threads=threads)
    out = {"rows": a.rows, "dim": a.dim, "k": a.k, "ef": a.ef, "build_seconds": time.perf_counter() - t0, "call_ms_by_nq": {}, "map_threads": []}
    g = m.export_graph(with_views=True)
    ix = capi.VectorIndex(2, a.dim, a.rows, device=0)
    ix.upload_rows(0, g["vectors"], g["inv_norms"])
    ix.hnsw_attach_graph(g)
    ix.hnsw_search_knn(queries, a.k, a.ef)
    for nq in ((1, 8, 128, 1024) if a.short else (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096)):
        reps = max(3, min(40, 2048 // nq))
        ix.hnsw_search_knn(queries[:nq], a.k, a.ef)
        t0 = time.perf_counter()
        for r in range(reps):
            ix.hnsw_search_knn(queries[(r * nq) % (nqmax - nq + 1):][:nq], a.k, a.ef)
        out["call_ms_by_nq"][nq] = (time.perf_counter() - t0) / reps * 1e3
        print("nq", nq, round(out["call_ms_by_nq"][nq], 3), "ms per call", flush=True)
    # the same with ONE query repeated nq times: every search of the launch walks the same path, so the call lasts as long as one search
    # does with nq - 1 others beside it — the device-side cost of concurrency without the max-over-different-queries effect
    import os
    os.environ["RXGPU_HNSW_SERVER"] = "0"   # (nq = 1 as a launch too)
    out["same_query_call_ms_by_nq"] = {}
    for nq in (1, 4, 16, 64, 128, 256):
        tot = 0.0
        for qi in range(12):
            block = np.repeat(queries[qi:qi + 1], nq, axis=0)
            ix.hnsw_search_knn(block, a.k, a.ef)
            t0 = time.perf_counter()
            for r in range(4):
                ix.hnsw_search_knn(block, a.k, a.ef)
            tot += (time.perf_counter() - t0) / 4
        out["same_query_call_ms_by_nq"][nq] = tot / 12 * 1e3
        print("same query x", nq, round(out["same_query_call_ms_by_nq"][nq], 3), "ms per call", flush=True)
    del os.environ["RXGPU_HNSW_SERVER"]
    ix.close()
    m.search_knn(queries[0], a.k, a.ef)
    for lanes in [int(x) for x in a.lanes.split(",")]:
        m.set_coalescer_lanes(lanes)
        for T in [int(x) for x in a.threads.split(",")]:
            m.search_knn_mt(queries, a.k, a.ef, T, 2, 10.0)
            secs, done, batches = m.search_knn_mt(queries, a.k, a.ef, T, 48, 20.0)
            rec = {"lanes": lanes, "threads": T, "queries_per_sec": done / secs, "avg_batch": done / max(batches, 1)}
            out["map_threads"].append(rec)
            print(rec, flush=True)
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
