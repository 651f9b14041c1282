#!/usr/bin/env python3
"""Kernel times of the wide-k brute-force search (hybrid leg shape: 5M x 512 cosine, k = 100 -> kk = 101) with and without the sampled
admission bound (RXGPU_SCAN_BOUND_MIN_ROWS, read per call): scan, sample scan, merges, from the library's HIP events."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from reindexer_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=5_000_000)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--metric", type=int, default=2)
    ap.add_argument("--kks", default="11,33,65,101,128")
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    rows = torch.randn((args.rows, args.dim), generator=g, device=dev, dtype=torch.float32)
    inv = (1.0 / rows.norm(dim=1)).contiguous() if args.metric == 2 else None
    q = torch.randn((1, args.dim), generator=g, device=dev, dtype=torch.float32)
    if args.metric == 2:
        q = q / q.norm()
    ix = capi.VectorIndex(args.metric, args.dim, 0)
    ix.adopt_device_rows(rows.data_ptr(), args.rows, args.dim, inv.data_ptr() if inv is not None else None, keepalive=(rows, inv))
    stream = torch.cuda.current_stream(dev).cuda_stream
    out = []
    for kk in [int(x) for x in args.kks.split(",")]:
        od = torch.empty((1, kk), dtype=torch.float32, device=dev)
        orow = torch.empty((1, kk), dtype=torch.int32, device=dev)
        res = {}
        for label, env in (("plain", "0"),):
            os.environ["RXGPU_SCAN_BOUND_MIN_ROWS"] = env
            def run():
                if kk <= 64:
                    ix.search_knn_device(q.data_ptr(), 1, kk, od.data_ptr(), orow.data_ptr(), None, stream)
                else:
                    ix.search_knn_resident(q[0].cpu().numpy(), kk)
            run()
            torch.cuda.synchronize(dev)
            ix.profile_enable(True)
            t0 = time.perf_counter()
            for _ in range(args.iters):
                run()
            torch.cuda.synchronize(dev)
            wall = (time.perf_counter() - t0) / args.iters * 1e3
            prof = {}
            for name in ("scan", "scan_subset", "merge"):
                n, ms = ix.profile_read(name)
                prof[name] = ms / max(n, 1) if n else None
                prof[name + "_launches"] = n
            ix.profile_enable(False)
            res[label] = dict(wall_ms=wall, **prof)
        bytes_ = args.rows * args.dim * 4
        for label in res:
            s = res[label]["scan"]
            res[label]["scan_hbm_frac"] = bytes_ / (s / 1e3) / 8e12 if s else None
        out.append(dict(kk=kk, **res))
        print("PROBE", json.dumps(out[-1]))
    ix.close()


if __name__ == "__main__":
    main()
