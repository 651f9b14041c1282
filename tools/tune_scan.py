#!/usr/bin/env python3
"""Within-process A/B sweep of the scan-kernel variants on one HBM-resident corpus (GPU box only).

    python tools/tune_scan.py [--rows 10000000] [--dim 768] [--metric ip] [--rounds 3] [--queries 20]

Variants are selected through RXGPU_SCAN_* (see knn_scan.hip: ScanTuning); RXGPU_TUNE_DYNAMIC makes the library
re-read them on every launch so all variants are interleaved in ONE process (cdna_hip_programming.md §5.4 rule 24).
Prints one JSON line per variant: median / min scan-kernel milliseconds from HIP events, and GB/s.
"""
import argparse
import itertools
import json
import os
import statistics
import sys
from pathlib import Path

os.environ["RXGPU_TUNE_DYNAMIC"] = "1"
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import torch  # noqa: E402

from reindexer_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--metric", default="ip")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--queries", type=int, default=20)
    ap.add_argument("--wg", default="2,3,4,6,8")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    corpus = torch.empty((args.rows, args.dim), dtype=torch.float32, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    for s in range(0, args.rows, 1 << 20):
        corpus[s:s + (1 << 20)].normal_(0, 0.25, generator=g)
    q = torch.empty((args.queries, args.dim), dtype=torch.float32, device=dev).normal_(0, 0.25, generator=g)
    kk = 11
    od = torch.empty((args.queries, kk), dtype=torch.float32, device=dev)
    orow = torch.empty((args.queries, kk), dtype=torch.int32, device=dev)
    ix = capi.VectorIndex(args.metric, args.dim)
    ix.adopt_device_rows(corpus.data_ptr(), args.rows, args.dim, None, keepalive=corpus)
    stream = torch.cuda.current_stream(dev).cuda_stream
    variants = [dict(qlds=a, prefetch=b, nt=c, wg=w) for a, b, c in itertools.product((0, 1), (0, 1), (0, 1))
                for w in [int(x) for x in args.wg.split(",")]]
    times = {i: [] for i in range(len(variants))}
    ref_rows = None
    for rnd in range(args.rounds + 1):  # round 0 = warmup
        for i, v in enumerate(variants):
            os.environ["RXGPU_SCAN_QLDS"] = str(v["qlds"])
            os.environ["RXGPU_SCAN_PREFETCH"] = str(v["prefetch"])
            os.environ["RXGPU_SCAN_NT"] = str(v["nt"])
            os.environ["RXGPU_SCAN_WG_PER_CU"] = str(v["wg"])
            ix.profile_enable(True)
            for j in range(args.queries):
                ix.search_knn_device(q.data_ptr() + j * args.dim * 4, 1, kk, od.data_ptr() + j * kk * 4, orow.data_ptr() + j * kk * 4, None, stream)
            torch.cuda.synchronize()
            n, ms = ix.profile_read("scan")
            ix.profile_enable(False)
            if ref_rows is None:
                ref_rows = orow.clone()
            assert torch.equal(ref_rows, orow), f"variant {v} changed the result"
            if rnd:
                times[i].append(ms / n)
    bytes_per = args.rows * args.dim * 4
    lines = []
    for i, v in enumerate(variants):
        med, mn = statistics.median(times[i]), min(times[i])
        lines.append(json.dumps({**v, "median_ms": round(med, 4), "min_ms": round(mn, 4), "GBps_median": round(bytes_per / med / 1e6, 1)}))
    lines.sort(key=lambda s: json.loads(s)["median_ms"])
    text = "\n".join(lines)
    print(text)
    if args.out:
        Path(args.out).write_text(text + "\n")


if __name__ == "__main__":
    main()
