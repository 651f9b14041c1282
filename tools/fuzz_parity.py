#!/usr/bin/env python3
"""Randomised differential test of the brute-force engine against the CPU oracle (bit-exact rows and distances).

    python tools/fuzz_parity.py --seconds 120 [--seed 1]

Every round draws a metric, a dimension (tails included), a corpus size, value scales (incl. quantised data with massive ties), a k, a batch
size (so the fused scan, the radix-select path, the f32 and bf16 nomination paths and — with RXGPU_SCAN_BF16=1 — the pruned scan all get
hit), applies a few random upserts / swap-deletes, and compares every query with the oracle; each round also runs the pre-filtered entries
(row list, bitmap, range over the list) over a random subset of the rows against the oracle on the sub-corpus."""
import argparse
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ.setdefault("RX_TARGET_INSTRUCTIONS", "avx512")

from oracle.pyoracle import Oracle  # noqa: E402
from reindexer_amd import capi  # noqa: E402


def lex_topk(d, k):
    order = np.lexsort((np.arange(d.shape[0]), d))[:k]
    return d[order], order.astype(np.uint32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    orc = Oracle()
    t_end = time.time() + args.seconds
    rounds = checked = checked_subset = 0
    while time.time() < t_end:
        metric = int(rng.integers(0, 3))
        d = int(rng.choice([1, 3, 16, 24, 63, 64, 65, 100, 128, 130, 192, 256, 300, 384, 512, 768, 1000, 1024]))
        n = int(rng.choice([1, 2, 7, 63, 64, 65, 300, 2000, 9000, 30000]))
        style = rng.choice(["gauss", "scaled", "quant", "dups"])
        if style == "gauss":
            rows = rng.normal(0, 0.25, (n, d)).astype(np.float32)
        elif style == "scaled":
            rows = (rng.normal(0, 1, (n, d)) * np.exp(rng.uniform(-8, 8, (n, 1)))).astype(np.float32)
        elif style == "quant":
            rows = rng.integers(-2, 3, (n, d)).astype(np.float32)
        else:
            base = rng.normal(0, 1, (max(1, n // 50), d)).astype(np.float32)
            rows = base[rng.integers(0, base.shape[0], n)].copy()
        if metric == 2:
            rows[np.all(rows == 0, axis=1)] = 1.0   # the planner never stores an all-zero vector in a cosine index
        nq = int(rng.choice([1, 1, 2, 5, 33, 70, 130, 256, 300]))
        if n * nq * d > 3e9:
            nq = 2
        queries = (rows[rng.integers(0, n, nq)] + rng.normal(0, 0.05, (nq, d))).astype(np.float32) if rng.random() < 0.5 else \
            rng.normal(0, 0.3, (nq, d)).astype(np.float32)
        if style == "quant":
            queries = np.rint(queries).astype(np.float32)
        if metric == 2:
            queries[np.all(queries == 0, axis=1)] = 1.0
            queries = np.stack([orc.normalize_copy(q)[0] for q in queries])
        k = int(rng.choice([1, 2, 10, 11, 63, 64, 100, 1000]))
        os.environ.pop("RXGPU_SCAN_BF16", None)
        if rng.random() < 0.4:
            os.environ["RXGPU_SCAN_BF16"] = "1"
        inv = orc.l2_modules(rows) if metric == 2 else None
        cap = n + 8
        with capi.VectorIndex(metric, d, cap) as ix:
            ix.upload_rows(0, rows, inv)
            cur, cur_inv = rows.copy(), (inv.copy() if inv is not None else None)
            for phase in range(2):
                if phase == 1:   # a few mutations: overwrite, append, swap-delete
                    for _ in range(int(rng.integers(1, 4))):
                        op = rng.integers(0, 3)
                        if op == 0 or cur.shape[0] < 2:
                            i = int(rng.integers(0, cur.shape[0]))
                            v = rng.normal(0, 0.3, d).astype(np.float32)
                            cur[i] = v
                            if cur_inv is not None:
                                cur_inv[i] = orc.l2_modules(v[None, :])[0]
                            ix.upload_rows(i, cur[i:i + 1], cur_inv[i:i + 1] if cur_inv is not None else None)
                        elif op == 1 and cur.shape[0] < cap:
                            v = rng.normal(0, 0.3, (1, d)).astype(np.float32)
                            iv = orc.l2_modules(v) if cur_inv is not None else None
                            ix.upload_rows(cur.shape[0], v, iv)
                            cur = np.concatenate([cur, v])
                            if cur_inv is not None:
                                cur_inv = np.concatenate([cur_inv, iv])
                        else:
                            victim, last = int(rng.integers(0, cur.shape[0])), cur.shape[0] - 1
                            ix.move_row(last, victim)
                            ix.truncate(last)
                            cur[victim] = cur[last]
                            cur = cur[:last]
                            if cur_inv is not None:
                                cur_inv[victim] = cur_inv[last]
                                cur_inv = cur_inv[:last]
                dist, row, cnt = ix.search_knn(queries, k)
                c = min(k, cur.shape[0])
                for qi in range(nq):
                    want = orc.dist_many(metric, queries[qi], cur, cur_inv)
                    wd, wr = lex_topk(want, c)
                    ok = int(cnt[qi]) == c and np.array_equal(row[qi, :c], wr) and np.array_equal(dist[qi, :c].view(np.uint32), wd.view(np.uint32))
                    if not ok:
                        print("MISMATCH", dict(metric=metric, d=d, n=cur.shape[0], style=str(style), nq=nq, k=k, phase=phase, qi=qi,
                                               pruned=os.environ.get("RXGPU_SCAN_BF16"), seed=args.seed, round=rounds), flush=True)
                        bad = np.flatnonzero(row[qi, :c] != wr)[:5]
                        print(" first diffs at", bad, row[qi, bad], wr[bad], dist[qi, bad], wd[bad], flush=True)
                        sys.exit(1)
                    checked += 1
                # pre-filtered entries over a random row subset of the current corpus: list, bitmap and range-over-list
                m = cur.shape[0]
                dens = float(rng.choice([0.0, 0.01, 0.2, 0.9, 1.0]))
                ids = np.flatnonzero(rng.random(m) < dens).astype(np.uint32) if dens < 1.0 else np.arange(m, dtype=np.uint32)
                qs = queries[: min(nq, 4)]
                sd, sr, sc = ix.search_knn_subset(qs, k, ids)
                words = np.zeros((m + 31) // 32, np.uint32)
                np.bitwise_or.at(words, ids.astype(np.int64) >> 5, np.uint32(1) << (ids & 31))
                bd, br, bc, allowed = ix.search_knn_bitmap(qs, k, words)
                c = min(k, ids.size)
                for qi in range(qs.shape[0]):
                    want = orc.dist_many(metric, qs[qi], cur[ids], cur_inv[ids] if cur_inv is not None else None) if ids.size else np.empty(0, np.float32)
                    wd, wpos = lex_topk(want, c)
                    wr = ids[wpos]
                    ok = (int(sc[qi]) == c and int(bc[qi]) == c and allowed == ids.size
                          and np.array_equal(sr[qi, :c], wr) and np.array_equal(sd[qi, :c].view(np.uint32), wd.view(np.uint32))
                          and np.array_equal(br[qi, :c], wr) and np.array_equal(bd[qi, :c].view(np.uint32), wd.view(np.uint32)))
                    if ok and ids.size:
                        radius = float(np.sort(want)[min(int(rng.integers(0, 40)), want.size - 1)])
                        keep = want < radius
                        rd, rr = ix.search_range_subset(qs[qi], radius, ids, cap=16)
                        gd, gpos = lex_topk(np.where(keep, want, np.inf), int(keep.sum()))
                        ok = np.array_equal(rr, ids[gpos]) and np.array_equal(rd.view(np.uint32), gd.view(np.uint32))
                    if not ok:
                        print("MISMATCH (pre-filter)", dict(metric=metric, d=d, n=m, style=str(style), k=k, phase=phase, qi=qi, dens=dens,
                                                            listed=int(ids.size), seed=args.seed, round=rounds), flush=True)
                        sys.exit(1)
                    checked_subset += 1
        rounds += 1
    print(f"fuzz ok: {rounds} rounds, {checked} queries checked, {checked_subset} pre-filtered queries (list + bitmap + range) checked, seed {args.seed}")


if __name__ == "__main__":
    main()
