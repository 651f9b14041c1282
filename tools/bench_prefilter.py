#!/usr/bin/env python3
"""Pre-filtered brute-force KNN (`WHERE cond AND KNN(...)`) on one MI355X: time per query against the selectivity of the filter.

    python tools/bench_prefilter.py --rows 4000000 --dim 768 --queries 20 [--out profiles/r1_prefilter.json]

For each density the allowed rows are a random sorted subset resident in HBM; rxgpu_search_knn_subset_device is timed with events on the
launch stream.  Bytes that HAVE to move = allowed rows x dim x 4 (+ 4 per id); `gbps` is that figure over the measured time, so it is
directly comparable with the unfiltered scan's roofline fraction.  The host entry points (id list / bitmap incl. upload and the on-device
bitmap expansion) are timed end to end beside it.  Checks in the same run: density 1.0 returns the unfiltered search bit for bit; every
returned row is allowed, its distance equals rxgpu_distances' bits for that row, and lists are (dist,row)-sorted; the bitmap and the list
entry return the same rows."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from bench import make_corpus  # noqa: E402
from reindexer_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--metric", default="ip")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--queries", type=int, default=20)
    ap.add_argument("--densities", default="0.001,0.01,0.1,0.5,1.0")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n, d, kk = args.rows, args.dim, args.k + 1
    corpus = make_corpus(n, d, 20260924, dev)
    gq = torch.Generator(device=dev)
    gq.manual_seed(7)
    queries = torch.empty((args.queries, d), dtype=torch.float32, device=dev).normal_(0.0, 0.25, generator=gq)
    hq = queries.cpu().numpy()
    metric_id = capi.METRICS[args.metric]
    ix = capi.VectorIndex(metric_id, d, device=0)
    ix.adopt_device_rows(corpus.data_ptr(), n, d, None, keepalive=(corpus,))
    stream = torch.cuda.current_stream(dev)
    od = torch.empty((args.queries, kk), dtype=torch.float32, device=dev)
    orow = torch.empty((args.queries, kk), dtype=torch.int32, device=dev)

    def timed(fn, iters):
        fn(0)
        torch.cuda.synchronize(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for i in range(iters):
            fn(i)
        b.record(stream)
        torch.cuda.synchronize(dev)
        return a.elapsed_time(b) / iters

    full_ms = timed(lambda i: ix.search_knn_device(queries.data_ptr() + i * d * 4, 1, kk, od.data_ptr() + i * kk * 4, orow.data_ptr() + i * kk * 4,
                                                   None, stream.cuda_stream), args.queries)
    full_d, full_r = od.clone(), orow.clone()
    out = {"rows": n, "dim": d, "metric": args.metric, "k": args.k, "kk": kk, "queries": args.queries, "arch": capi.device_arch(0),
           "unfiltered": {"ms_per_query": full_ms, "gbps": n * d * 4 / full_ms / 1e6}, "densities": []}
    rng = np.random.default_rng(5)
    for dens in [float(x) for x in args.densities.split(",")]:
        ids = np.arange(n, dtype=np.uint32) if dens >= 1.0 else np.flatnonzero(rng.random(n) < dens).astype(np.uint32)
        tids = torch.from_numpy(ids.view(np.int32)).to(dev)
        assert ix.check_row_list_device(tids.data_ptr(), ids.size, stream.cuda_stream)
        ms = timed(lambda i: ix.search_knn_subset_device(queries.data_ptr() + i * d * 4, 1, kk, tids.data_ptr(), ids.size,
                                                         od.data_ptr() + i * kk * 4, orow.data_ptr() + i * kk * 4, None, stream.cuda_stream),
                   args.queries)
        rows_h = orow.cpu().numpy().view(np.uint32)
        dist_h = od.cpu().numpy()
        allowed = np.zeros(n, bool)
        allowed[ids] = True
        ok_allowed = bool(allowed[rows_h].all())
        ok_sorted = all(bool(np.all((dist_h[q, :-1] < dist_h[q, 1:]) | ((dist_h[q, :-1] == dist_h[q, 1:]) & (rows_h[q, :-1] < rows_h[q, 1:]))))
                        for q in range(args.queries))
        ok_bits = all(np.array_equal(ix.distances(hq[q], rows_h[q]).view(np.uint32), dist_h[q].view(np.uint32)) for q in range(min(4, args.queries)))
        ok_full = None
        if dens >= 1.0:
            ok_full = bool(torch.equal(orow, full_r) and torch.equal(od.view(torch.int32), full_d.view(torch.int32)))
        # host entry points, end to end (upload of the list / of the bitmap + expansion + search + download)
        t0 = time.perf_counter()
        reps = 3
        for i in range(reps):
            ld, lr, _ = ix.search_knn_subset(hq[i], kk, ids)
        t_list = (time.perf_counter() - t0) / reps * 1e3
        words = np.zeros((n + 31) // 32, np.uint32)
        np.bitwise_or.at(words, ids.astype(np.int64) >> 5, np.uint32(1) << (ids & 31))
        t0 = time.perf_counter()
        for i in range(reps):
            bd, br, _, nallowed = ix.search_knn_bitmap(hq[i], kk, words)
        t_bitmap = (time.perf_counter() - t0) / reps * 1e3
        same_entries = bool(np.array_equal(lr, br) and np.array_equal(ld.view(np.uint32), bd.view(np.uint32)) and nallowed == ids.size
                            and np.array_equal(br[0], rows_h[reps - 1]))
        moved = ids.size * (d * 4 + 4)
        out["densities"].append({"density": dens, "allowed_rows": int(ids.size), "ms_per_query": ms, "gbps": moved / ms / 1e6,
                                 "speedup_vs_unfiltered": full_ms / ms, "host_list_ms": t_list, "host_bitmap_ms": t_bitmap,
                                 "parity": {"rows_allowed": ok_allowed, "sorted": ok_sorted, "distance_bits": ok_bits,
                                            "equals_unfiltered": ok_full, "list_equals_bitmap": same_entries}})
    ix.close()
    line = json.dumps(out)
    print(line, flush=True)
    if args.out:
        Path(args.out).write_text(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
