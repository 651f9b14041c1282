#!/usr/bin/env python3
"""A/B of the two bf16 nomination kernels (knn_batched_bf16.hip) inside ONE process, on one corpus and one box: the single-ring kernel
(RXGPU_GEMM_SPLIT=0) against the split-ring one (default), each over the tile-blocked bf16 shadow (default) and the row-major one
(RXGPU_SHADOW_BLOCKED=0), alternating, for every metric.  Kernel time = the library's own HIP events around
the nomination GEMM's filter pass over the whole corpus (the "gemm" profile scope); results are checked against each other.
    python tools/bench_gemm_ab.py [--rows 10000000] [--dim 768] [--batch 256] [--metrics ip,l2,cosine] [--rounds 3] [--iters 4] [--out f.json]"""
import argparse
import json
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import bench  # noqa: E402
from reindexer_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metrics", default="ip,l2,cosine")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--modes", default=None, help="comma list of mode names to run (default: all four)")
    ap.add_argument("--pruned", action="store_true", help="also time the bf16-pruned batch-1 scan under both shadow layouts")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    corpus = bench.make_corpus(a.rows, a.dim, 20260924, dev)
    queries = bench.make_corpus(a.batch, a.dim, 777, dev)
    kk = a.k + 1
    stream = torch.cuda.current_stream(dev).cuda_stream
    result = {"rows": a.rows, "dim": a.dim, "batch": a.batch, "kk": kk, "metrics": {}}
    for metric in a.metrics.split(","):
        mid = capi.METRICS[metric]
        d_inv = (1.0 / torch.linalg.vector_norm(corpus, dim=1)) if mid == 2 else None
        q = queries if mid != 2 else queries / torch.linalg.vector_norm(queries, dim=1, keepdim=True)
        q = q.contiguous()
        def make_index(blocked):
            if blocked:
                os.environ.pop("RXGPU_SHADOW_BLOCKED", None)
            else:
                os.environ["RXGPU_SHADOW_BLOCKED"] = "0"   # read when the shadow is first built (the warm-up below)
            ix_ = capi.VectorIndex(mid, a.dim, device=0)
            ix_.adopt_device_rows(corpus.data_ptr(), a.rows, a.dim, d_inv.data_ptr() if d_inv is not None else None, keepalive=(corpus, d_inv))
            return ix_

        # (name, shadow layout, environment of the launch)
        modes = [("split_ring_blocked_shadow", True, {"RXGPU_GEMM_SPLIT": "1"}), ("split_ring_blocked_shadow_setprio", True, {"RXGPU_GEMM_SPLIT": "1", "RXGPU_GEMM_PRIO": "1"}), ("split_ring_rowmajor_shadow", False, {"RXGPU_GEMM_SPLIT": "1"}),
                 ("single_ring_blocked_shadow", True, {"RXGPU_GEMM_SPLIT": "0"}), ("single_ring_rowmajor_shadow", False, {"RXGPU_GEMM_SPLIT": "0"})]
        if a.modes:
            modes = [m for m in modes if m[0] in a.modes.split(",")]
        index = {}

        def set_mode(env):
            os.environ.pop("RXGPU_GEMM_PRIO", None)
            os.environ.update(env)

        od = {m: torch.empty((a.batch, kk), dtype=torch.float32, device=dev) for m, _, _ in modes}
        orow = {m: torch.empty((a.batch, kk), dtype=torch.int32, device=dev) for m, _, _ in modes}
        times = {m: [] for m, _, _ in modes}
        for mode, blocked, env in modes:   # warm-up of all (shadow, row statistics, LDS attribute)
            if blocked not in index:
                index[blocked] = make_index(blocked)
            set_mode(env)
            index[blocked].search_knn_device(q.data_ptr(), a.batch, kk, od[mode].data_ptr(), orow[mode].data_ptr(), None, stream)
        os.environ.pop("RXGPU_SHADOW_BLOCKED", None)
        torch.cuda.synchronize(dev)
        for ix_ in index.values():
            ix_.profile_enable(True)
        for _ in range(a.rounds):
            for mode, blocked, env in modes:
                ix = index[blocked]
                set_mode(env)
                ix.profile_read("gemm")
                for _ in range(a.iters):
                    ix.search_knn_device(q.data_ptr(), a.batch, kk, od[mode].data_ptr(), orow[mode].data_ptr(), None, stream)
                torch.cuda.synchronize(dev)
                n, ms = ix.profile_read("gemm")
                times[mode].append(ms / max(n, 1))
        first = modes[0][0]
        same = all(bool(torch.equal(orow[m], orow[first]) and torch.equal(od[m].view(torch.int32), od[first].view(torch.int32))) for m, _, _ in modes)
        flops = 2.0 * (128 if a.batch <= 128 else 256) * a.rows * ((a.dim + 63) // 64 * 64)
        shadow = float(a.rows) * ((a.dim + 63) // 64 * 64) * 2
        entry = {"identical_results": same}
        for name, _, _ in modes:
            best = min(times[name])
            entry[name] = {"gemm_ms_per_launch": times[name], "best_ms": best, "mfma_frac_of_2500TF": flops / (best / 1e3) / 1e12 / 2500.0,
                           "hbm_frac_on_shadow": shadow / (best / 1e3) / 1e9 / 8000.0}
        # the opt-in pruned batch-1 scan reads the same shadow (knn_scan_bf16): its time under both layouts
        if a.pruned:
            os.environ["RXGPU_SCAN_BF16"] = "1"
            for blocked, ix in index.items():
                o1 = torch.empty((1, kk), dtype=torch.float32, device=dev)
                r1 = torch.empty((1, kk), dtype=torch.int32, device=dev)
                ix.search_knn_device(q.data_ptr(), 1, kk, o1.data_ptr(), r1.data_ptr(), None, stream)
                torch.cuda.synchronize(dev)
                ix.profile_read("scan_bf16")
                for _ in range(8):
                    ix.search_knn_device(q.data_ptr(), 1, kk, o1.data_ptr(), r1.data_ptr(), None, stream)
                torch.cuda.synchronize(dev)
                n, ms = ix.profile_read("scan_bf16")
                entry["pruned_scan_ms_" + ("blocked" if blocked else "rowmajor")] = ms / max(n, 1)
            os.environ.pop("RXGPU_SCAN_BF16", None)
        for ix_ in index.values():
            ix_.profile_enable(False)
        result["metrics"][metric] = entry
        print(metric, json.dumps(entry), flush=True)
        index.clear()
    os.environ.pop("RXGPU_GEMM_SPLIT", None)
    os.environ.pop("RXGPU_GEMM_PRIO", None)
    if a.out:
        Path(a.out).parent.mkdir(parents=True, exist_ok=True)
        Path(a.out).write_text(json.dumps(result) + "\n")


if __name__ == "__main__":
    main()
