#!/usr/bin/env python3
"""BM25 leg (BASELINE configs[4], ft_fast half): single-term ft_fast merge over a synthetic 5M-vdoc inverted index.

    python tools/bench_bm25.py --docs 5000000 --queries 20 [--out profiles/r1_bm25.json]

Each query = one OR-term with 3 sub-terms (exact word, a stem variant, a typo variant) whose document frequencies follow a
Zipf-like spread (default 20 % / 5 % / 1 % of the corpus), one FT field, default FTConfig (mergeLimit 20000).
Reports postings/s and achieved GB/s of the scoring kernel (20 B per posting: SURVEY §8d) from the library's HIP events,
end-to-end merges/s through GpuFtMerger, the CPU port (oracle/oracle_bm25.c, 1 thread) on the same postings, and parity
(id set + uint8 ranks + raw float ranks identical)."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from reindexer_amd import hostapi  # noqa: E402


def postings(rng, total_docs, frac, max_tf=5):
    mask = rng.random(total_docs) < frac
    mask[0] = False
    doc = np.nonzero(mask)[0].astype(np.uint32)
    n = doc.shape[0]
    return dict(doc=doc, ent_off=np.arange(n + 1, dtype=np.uint32), ent_field=np.zeros(n, np.uint8),
                ent_tf=rng.integers(1, max_tf + 1, n).astype(np.uint32), ent_first_pos=rng.integers(0, 60, n).astype(np.uint32))


def pos_postings(rng, total_docs, frac, proc):
    """Positions-format postings (multi-term merge): 1-3 positions per document in field 0, ascending."""
    mask = rng.random(total_docs) < frac
    mask[0] = False
    doc = np.nonzero(mask)[0].astype(np.uint32)
    n = doc.shape[0]
    k = rng.integers(1, 4, n)
    pos_off = np.zeros(n + 1, np.int64)
    pos_off[1:] = np.cumsum(k)
    start = rng.integers(0, 40, n)
    step = rng.integers(1, 9, int(pos_off[-1]))
    owner = np.repeat(np.arange(n), k)
    csum = np.cumsum(step)
    c0 = csum[pos_off[:-1]] - step[pos_off[:-1]]
    fpos = (start[owner] + (csum - c0[owner])).astype(np.uint64)   # strictly ascending within a posting (field 0, arrayIdx 0)
    return dict(doc=doc, pos_off=pos_off.astype(np.uint32), fpos=fpos, proc=proc)


def run_multi(args):
    """--ops 1,1  (OR OR) / 2,1 (AND OR) ...: multi-term merge through GpuFtMerger.merge_query vs the CPU port."""
    rng = np.random.default_rng(20260925)
    ops = [int(x) for x in args.ops.split(",")]
    total = args.docs + 1
    words = rng.integers(20, 61, (total, 1)).astype(np.float32)
    words[0] = 0
    avg = words[1:].mean(axis=0).astype(np.float32)
    fracs = [float(x) for x in args.fracs.split(",")]
    procs = [100.0, 85.0, 70.0][:len(fracs)]
    m = hostapi.GpuFtMerger(1)
    m.set_docs(words, avg)
    terms_o, terms_g, wid = [], [], 0
    for op in ops:
        subs_o, subs_g = [], []
        for j, fr in enumerate(fracs):
            s = pos_postings(rng, total, fr, procs[j])
            m.set_word_fpos(wid, s)
            subs_o.append(s)
            subs_g.append((wid, procs[j]))
            wid += 1
        o = hostapi.default_ft_opts(1)
        terms_o.append(dict(op=op, opts=o, subs=subs_o))
        terms_g.append(dict(op=op, opts=o, subs=subs_g))
    cfg = hostapi.default_ft_config(1)
    if args.batch_only:   # profiling runs: nothing but trains of Q queries (a single merge is the same kernels with grid.y = 1)
        Q = int(args.batch.split(",")[0])
        for _ in range(max(2, args.queries // Q) + 1):
            m.merge_query_batch(cfg, [terms_g] * Q, sort_by_rank=False)
        npost, kernel_ms = m.read_stats()
        print(json.dumps({"batch_only": Q, "postings": npost, "kernel_ms": kernel_ms}))
        return
    m.merge_query(cfg, terms_g)
    m.read_stats()
    m.read_timing()
    t0 = time.perf_counter()
    for _ in range(args.queries):
        res = m.merge_query(cfg, terms_g, sort_by_rank=False)
    gpu_s = time.perf_counter() - t0
    npost, kernel_ms = m.read_stats()
    host_calls, host_ms = m.read_timing()
    npos = sum(int(s["pos_off"][-1]) for t in terms_o if t["op"] != 3 for s in t["subs"])
    nposting = sum(int(s["doc"].shape[0]) for t in terms_o if t["op"] != 3 for s in t["subs"])
    bytes_per_merge = nposting * (4 + 8 + 9 + 8 + 4 + 4 + 4) + npos * 8   # doc, entry offs, entry, pos offs, words gather, slot gather, mask ; positions
    out = {"workload": f"ft_fast multi-term merge ops={ops}, {args.docs} vdocs, sub-term df fractions {fracs}, 1 field, mergeLimit 20000",
           "postings_per_query": npost / args.queries, "preselected": bool(res[4]), "results": int(res[0].shape[0]),
           "gpu": {"merges_per_sec": args.queries / gpu_s, "ms_per_merge": gpu_s / args.queries * 1e3, "term_pass_ms_per_merge": kernel_ms / args.queries,
                   "ms_per_merge_cpp_boundary": host_ms / max(host_calls, 1),
                   "note_e2e": "ms_per_merge = through this Python harness (ctypes marshalling of the query included); ms_per_merge_cpp_boundary = "
                               "inside GpuFtMerger::MergeQuery, the drop-in boundary (plan, launches, wait, unpack, postProcessResults)",
                   "postings_per_sec_kernel": npost / (kernel_ms / 1e3),
                   "roofline": {"bound": "hbm", "achieved": bytes_per_merge / (kernel_ms / args.queries / 1e3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                "frac": bytes_per_merge / (kernel_ms / args.queries / 1e3) / 1e9 / 8000.0,
                                "note": "per posting 41 B (doc, entry offsets, entry, position offsets, words/slot/mask gathers) + 8 B per position"}}}
    if args.batch:
        # Q merges in ONE launch train (GpuFtMerger::MergeQueryBatch): the same query Q times — every one must equal the single merge
        trains = []
        for Q in [int(x) for x in args.batch.split(",")]:
            m.merge_query_batch(cfg, [terms_g] * Q, sort_by_rank=False)   # warm-up: the batch lanes' scratch
            m.read_stats()
            reps = max(2, args.queries // Q)
            t0 = time.perf_counter()
            for _ in range(reps):
                bres = m.merge_query_batch(cfg, [terms_g] * Q, sort_by_rank=False)
            wall = time.perf_counter() - t0
            bp, bk = m.read_stats()
            same = all(np.array_equal(b[0], res[0]) and np.array_equal(b[1].view(np.uint32), res[1].view(np.uint32)) and np.array_equal(b[3], res[3]) for b in bres)
            trains.append({"queries_per_train": Q, "trains": reps, "kernel_ms_per_merge": bk / (reps * Q), "kernel_ms_per_train": bk / reps,
                           "merges_per_sec_wall": reps * Q / wall, "identical_to_single_merge": bool(same),
                           "roofline": {"bound": "hbm", "achieved": bytes_per_merge / (bk / (reps * Q) / 1e3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                        "frac": bytes_per_merge / (bk / (reps * Q) / 1e3) / 1e9 / 8000.0,
                                        "frac_20B_per_posting": nposting * 20 / (bk / (reps * Q) / 1e3) / 1e9 / 8000.0}})
        out["batched_trains"] = trains
    if args.threads:
        # several planner threads against one index: the merges of concurrent callers run on the handle's lanes (own stream + scratch each)
        conc = []
        for th in [int(x) for x in args.threads.split(",")]:
            m.merge_query_concurrent(cfg, terms_g, th, 2)   # warm-up: the lanes' scratch
            m.read_stats()
            reps = max(4, args.queries)
            n_res, wall_ms = m.merge_query_concurrent(cfg, terms_g, th, reps)
            cp, ck = m.read_stats()
            merges = th * reps + 1   # (+ the reference merge of the driver)
            conc.append({"threads": th, "merges": merges, "merges_per_sec": th * reps / (wall_ms / 1e3), "ms_per_merge_wall": wall_ms / (th * reps),
                         "kernel_ms_per_merge_in_lane": ck / merges,
                         "roofline_wall": {"bound": "hbm", "achieved": bytes_per_merge * th * reps / (wall_ms / 1e3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                           "frac": bytes_per_merge * th * reps / (wall_ms / 1e3) / 1e9 / 8000.0,
                                           "note": "same byte model as gpu.roofline, over the WALL time of the whole concurrent run (host work included)"}})
        out["concurrent"] = conc
    try:
        from oracle.pyoracle import FtOracle, Oracle, ref_ft_or_none
        ft = FtOracle(Oracle())
        t0 = time.perf_counter()
        wd, wp, wf, wn, wpre = ft.merge_query(cfg, terms_o, total, words, avg, None, None, sort_by_rank=False)
        cpu_s = time.perf_counter() - t0
        gd, gp, gf, gn, gpre = res
        same = bool(np.array_equal(gd, wd.astype(np.int32)) and np.array_equal(gn, wn) and np.array_equal(gp.view(np.uint32), wp.view(np.uint32)) and gpre == wpre)
        out["cpu_baseline"] = {"kind": "port", "value": 1.0 / cpu_s, "unit": "merges/s", "cores": 1, "sample": "1 of the same merges",
                               "postings_per_sec": npost / args.queries / cpu_s}
        out["parity"] = {"identical_results": same, "checked": 1}
        real = ref_ft_or_none(1)
        if real is not None:
            real.set_docs(words, avg, None)
            w = 0
            for t in terms_o:
                for s_ in t["subs"]:
                    real.set_word_fpos(w, s_)
                    w += 1
            real.set_config(cfg)
            t0 = time.perf_counter()
            rd, rp, rf, rn = real.merge(terms_g, None, rank_sort_type=1)
            out["cpu_reference"] = {"kind": "reference", "value": 1.0 / (time.perf_counter() - t0), "unit": "merges/s", "cores": 1,
                                    "identical_to_gpu": bool(np.array_equal(rd, gd) and np.array_equal(rn, gn))}
    except Exception as e:
        out["cpu_baseline"] = {"error": repr(e)}
    text = json.dumps(out)
    print(text)
    if args.out:
        Path(args.out).write_text(text + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", default=None, help="multi-term mode: comma list of OpType per term (1 OR, 2 AND, 3 NOT)")
    ap.add_argument("--docs", type=int, default=5_000_000)
    ap.add_argument("--queries", type=int, default=20)
    ap.add_argument("--fracs", default="0.2,0.05,0.01")
    ap.add_argument("--out", default=None)
    ap.add_argument("--threads", default=None, help="multi-term mode: comma list of concurrent caller counts (e.g. 1,2,4,8)")
    ap.add_argument("--batch-only", action="store_true", help="with --batch Q: run nothing but trains of Q queries (for rocprofv3)")
    ap.add_argument("--batch", default=None, help="multi-term mode: comma list of queries per launch train (MergeQueryBatch), e.g. 2,8,32")
    args = ap.parse_args()
    if args.ops:
        return run_multi(args)
    rng = np.random.default_rng(20260924)
    total = args.docs + 1
    words = rng.integers(20, 61, (total, 1)).astype(np.float32)   # 20-60 tokens per doc
    words[0] = 0
    avg = words[1:].mean(axis=0).astype(np.float32)
    fracs = [float(x) for x in args.fracs.split(",")]
    procs = [100.0, 85.0, 70.0][:len(fracs)]
    m = hostapi.GpuFtMerger(1)
    m.set_docs(words, avg)
    n_words = 4   # distinct query words, cycled
    subs_of = []
    for w in range(n_words):
        subs = []
        for j, fr in enumerate(fracs):
            s = postings(rng, total, fr)
            s["proc"] = procs[j]
            wid = w * 16 + j
            subs.append((wid, s))
        subs_of.append(subs)
    # flat upload (the position-record path of hostapi re-expands tf positions; not needed for synthetic postings)
    import ctypes as C
    lib = hostapi.lib()
    lib.rxhost_ft_set_word_flat.argtypes = [C.c_void_p, C.c_uint32, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    for subs in subs_of:
        for wid, s in subs:
            rc = lib.rxhost_ft_set_word_flat(m.h, wid, s["doc"].shape[0], s["doc"].ctypes.data, s["ent_off"].ctypes.data, s["ent_field"].ctypes.data,
                                             s["ent_tf"].ctypes.data, s["ent_first_pos"].ctypes.data)
            assert rc == 0
    cfg, opts = hostapi.default_ft_config(1), hostapi.default_ft_opts(1)
    m.merge(cfg, opts, [(wid, s["proc"]) for wid, s in subs_of[0]])   # warmup
    m.read_stats()
    m.read_timing()
    t0 = time.perf_counter()
    results = []
    for q in range(args.queries):
        subs = subs_of[q % n_words]
        results.append(m.merge(cfg, opts, [(wid, s["proc"]) for wid, s in subs]))
    gpu_s = time.perf_counter() - t0
    npost, kernel_ms = m.read_stats()
    host_calls, host_ms = m.read_timing()

    out = {"workload": f"ft_fast single-term BM25 merge, {args.docs} vdocs, sub-term df fractions {fracs}, 1 field, mergeLimit 20000",
           "postings_per_query": npost / args.queries,
           "gpu": {"merges_per_sec": args.queries / gpu_s, "ms_per_merge": gpu_s / args.queries * 1e3, "term_pass_ms_per_merge": kernel_ms / args.queries,
                   "ms_per_merge_cpp_boundary": host_ms / max(host_calls, 1),
                   "note_e2e": "ms_per_merge = through this Python harness (ctypes marshalling of the query included); ms_per_merge_cpp_boundary = "
                               "inside GpuFtMerger::MergeQuery, the drop-in boundary (plan, launches, wait, unpack, postProcessResults)",
                   "postings_per_sec_kernel": npost / (kernel_ms / 1e3),
                   "roofline": {"bound": "hbm", "achieved": npost * 29 / (kernel_ms / 1e3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                "frac": npost * 29 / (kernel_ms / 1e3) / 1e9 / 8000.0, "bytes_per_posting": 29,
                                "note": "4 doc + 8 entry offsets + 9 entry streamed; 4 words-in-field + 4 slot index gathered (+ the mask bit); SURVEY 8d's 20 B "
                                        "layout + the slot index that replaces the per-document score word"}}}
    try:
        from oracle.pyoracle import FtOracle, Oracle
        ft = FtOracle(Oracle())
        nq = min(args.queries, n_words)
        t0 = time.perf_counter()
        cpu = [ft.merge_simple(cfg, opts, total, words, avg, None, None, [s for _, s in subs_of[q]], sort_by_rank=True) for q in range(nq)]
        cpu_s = time.perf_counter() - t0
        same = 0
        for q in range(nq):
            gd, gp, gf, gn = results[q]
            wd, wp, wf, wn = cpu[q]
            same += int(np.array_equal(np.sort(gd.astype(np.uint32)), np.sort(wd)) and np.array_equal(gn[np.argsort(gd, kind="stable")], wn[np.argsort(wd, kind="stable")]))
        out["cpu_baseline"] = {"kind": "port", "value": nq / cpu_s, "unit": "merges/s", "cores": 1, "sample": f"{nq} of the same merges",
                               "postings_per_sec": npost / args.queries * nq / cpu_s}
        out["parity"] = {"identical_results_frac": same / nq, "checked": nq}
    except Exception as e:
        out["cpu_baseline"] = {"error": repr(e)}
    text = json.dumps(out)
    print(text)
    if args.out:
        Path(args.out).write_text(text + "\n")


if __name__ == "__main__":
    main()
