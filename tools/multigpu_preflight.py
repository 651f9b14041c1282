#!/usr/bin/env python3
"""Before the first multi-GPU run: does THIS node run the sharded engines?  (no multi-GPU node has been reachable from the build
container, so ncclCommInitAll over > 1 device and the peer copies have never executed — this tool makes their first run say what it found.)

    python tools/multigpu_preflight.py [--gpus N] [--rows 200000] [--dim 128]

On a node with N visible GPUs it
  * lists the devices and the peer-access matrix (hipDeviceCanAccessPeer),
  * creates one sharded index per engine over devices 0..N-1 (brute force: rxgpu_index_create_sharded; HNSW: GpuHnswMap over the device
    list; BM25: rxgpu_ft_create_sharded) — which opens the pooled RCCL communicators — and reports the exchange mode each one got
    (device = ncclAllGather on the devices, host = through the host, with the library's note saying why),
  * runs a few queries per engine against the SAME data on a single-device index and compares: ids / distance bits (brute force), label sets
    per shard merge (HNSW: recall vs exact as well), merged documents + raw ranks (BM25),
  * prints ONE JSON line; exit code 0 iff every engine matched.
With one GPU (this build's test boxes) device 0 is listed twice: same code path, one RCCL rank with two slots."""
import argparse
import ctypes as C
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="devices to use (0: all visible; a single GPU is listed twice)")
    ap.add_argument("--rows", type=int, default=200_000)
    ap.add_argument("--dim", type=int, default=128)
    a = ap.parse_args()
    from reindexer_amd import capi, hostapi
    capi.lib()
    out = {"tool": "multigpu_preflight", "ok": False}
    ndev = capi.device_count()
    n = a.gpus or ndev
    devices = list(range(n)) if n > 1 else [0, 0]
    out["visible_devices"] = ndev
    out["devices_used"] = devices
    out["arch"] = [capi.device_arch(d) for d in sorted(set(devices))]
    hip = C.CDLL("libamdhip64.so")
    peer = []
    for i in sorted(set(devices)):
        row = []
        for j in sorted(set(devices)):
            can = C.c_int(0)
            rc = hip.hipDeviceCanAccessPeer(C.byref(can), i, j) if i != j else 0
            row.append(int(can.value) if i != j and rc == 0 else (1 if i == j else -1))
        peer.append(row)
    out["peer_access"] = peer
    rng = np.random.default_rng(20260930)
    ok = True

    # ---- brute force (e1): row-range shards + one all-gather + merge
    t0 = time.perf_counter()
    rows = rng.normal(0, 0.25, (a.rows, a.dim)).astype(np.float32)
    q = rng.normal(0, 0.25, (8, a.dim)).astype(np.float32)
    leg = {}
    try:
        with capi.ShardedVectorIndex(1, a.dim, a.rows, devices) as sx, capi.VectorIndex(1, a.dim, a.rows, device=devices[0]) as one:
            sx.upload_rows(0, rows)
            one.upload_rows(0, rows, None)
            leg["merge_mode"] = sx.merge_mode
            leg["note"] = sx.merge_note if hasattr(sx, "merge_note") else None
            c0 = sx.collectives
            d1, r1, _ = sx.search_knn(q, 11)
            d0, r0, _ = one.search_knn(q, 11)
            leg["collectives"] = sx.collectives - c0
            leg["identical"] = bool(np.array_equal(r0, r1) and np.array_equal(d0.view(np.uint32), d1.view(np.uint32)))
    except Exception as e:
        leg["error"] = repr(e)
    leg["seconds"] = round(time.perf_counter() - t0, 2)
    ok = ok and leg.get("identical", False)
    out["brute_force"] = leg

    # ---- HNSW (e2): a graph per shard + the same all-gather merge
    t0 = time.perf_counter()
    leg = {}
    try:
        hn = min(a.rows, 40_000)
        labels = np.arange(hn, dtype=np.uint64) << np.uint64(32)
        many = hostapi.GpuHnswMap(0, a.dim, hn, M=8, ef_construction=100, devices=devices)
        many.add(rows[:hn], labels)
        leg["shards"] = many.shard_count
        exact = capi.VectorIndex(0, a.dim, hn, device=devices[0])
        exact.upload_rows(0, rows[:hn], None)
        hit = tot = 0
        same_as_shards = True
        for qi in range(8):
            gd, gl = many.search_knn(q[qi], 10, 64)
            _, er, _ = exact.search_knn(q[qi:qi + 1], 10)
            hit += len(set((gl >> np.uint64(32)).tolist()) & set(er[0].tolist()))
            tot += 10
            per = []
            for s in range(many.shard_count):
                sh = many.shard(s)
                if sh.count:
                    sd, sl = sh.search_knn(q[qi], 10, 64)
                    per.append((sd, sl))
            ad = np.concatenate([p[0] for p in per])
            al = np.concatenate([p[1] for p in per])
            order = np.lexsort((al, ad))[:10]
            same_as_shards = same_as_shards and set(al[order].tolist()) == set(gl.tolist())
        leg["recall_at_10_vs_exact"] = hit / tot
        leg["identical"] = bool(same_as_shards)
        exact.close()
        many.close()
    except Exception as e:
        leg["error"] = repr(e)
    leg["seconds"] = round(time.perf_counter() - t0, 2)
    ok = ok and leg.get("identical", False)
    out["hnsw"] = leg

    # ---- BM25 (e3): document-range shards, two all-gathers inside the launch train
    t0 = time.perf_counter()
    leg = {}
    try:
        total, nf = 100_000, 1
        words = rng.integers(20, 61, (total, nf)).astype(np.float32)
        words[0] = 0
        avg = words[1:].mean(axis=0).astype(np.float32)
        one, many = hostapi.GpuFtMerger(nf, device=devices[0]), hostapi.GpuFtMerger(nf, devices=devices)
        one.set_docs(words, avg)
        many.set_docs(words, avg)
        terms = []
        for t in range(3):
            doc = np.sort(rng.choice(np.arange(1, total), 20_000 // (t + 1), replace=False)).astype(np.uint32)
            po = np.arange(doc.shape[0] + 1, dtype=np.uint32)
            fp = (rng.integers(0, 40, doc.shape[0])).astype(np.uint64)
            s = dict(doc=doc, pos_off=po, fpos=fp, proc=100.0 - 10 * t)
            one.set_word_fpos(t, s)
            many.set_word_fpos(t, s)
            terms.append(dict(op=1, opts=hostapi.default_ft_opts(nf), subs=[(t, s["proc"])]))
        cfg = hostapi.default_ft_config(nf)
        cfg["merge_limit"] = 5000
        lib = capi.lib()
        leg["exchange_mode"] = int(lib.rxgpu_ft_shard_exchange_mode(many.device_index))   # 1 on the devices, 0 through the host
        c0 = int(lib.rxgpu_ft_shard_collectives(many.device_index))
        x, y = one.merge_query(cfg, terms, None, sort_by_rank=False), many.merge_query(cfg, terms, None, sort_by_rank=False)
        leg["collectives"] = int(lib.rxgpu_ft_shard_collectives(many.device_index)) - c0
        leg["identical"] = bool(np.array_equal(x[0], y[0]) and np.array_equal(x[1].view(np.uint32), y[1].view(np.uint32)) and x[4] == y[4])
        leg["merged_documents"] = int(len(x[0]))
        # a phrase over the shards ("w0 w1"~30 OR w2; merge_limit 5000 < w0's 20 000 documents: PhraseMerger's admission cut is settled between the
        # shards) and the same terms with a two-word synonym hung on the first one
        phrase = [dict(terms[0], phrase=0, distance=30), dict(terms[1], phrase=0, distance=30), dict(terms[2], phrase=-1)]
        x, y = one.merge_query(cfg, phrase, None, sort_by_rank=False), many.merge_query(cfg, phrase, None, sort_by_rank=False)
        leg["phrase_identical"] = bool(np.array_equal(x[0], y[0]) and np.array_equal(x[1].view(np.uint32), y[1].view(np.uint32)))
        leg["phrase_documents"] = int(len(x[0]))
        syn = dict(synonyms=[[terms[1], terms[2]]], part_synonyms=[[0]])
        x, y = one.merge_query(cfg, terms[:1], None, sort_by_rank=False, **syn), many.merge_query(cfg, terms[:1], None, sort_by_rank=False, **syn)
        leg["synonym_identical"] = bool(np.array_equal(x[0], y[0]) and np.array_equal(x[1].view(np.uint32), y[1].view(np.uint32)))
        leg["identical"] = bool(leg["identical"] and leg["phrase_identical"] and leg["synonym_identical"])
        one.close()
        many.close()
    except Exception as e:
        leg["error"] = repr(e)
    leg["seconds"] = round(time.perf_counter() - t0, 2)
    ok = ok and leg.get("identical", False)
    out["bm25"] = leg

    out["ok"] = bool(ok)
    print(json.dumps(out))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
