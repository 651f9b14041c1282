#!/usr/bin/env python3
"""Headline benchmark: brute-force KNN queries/s, 10M x 768 fp32 inner product, k = 10 (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: ONE query (batch = 1) scanned against the
whole HBM-resident corpus through the C-ABI (rxgpu_search_knn_device), results left in HBM.  With N > 1 every rank
holds its own 10M-row shard (weak scaling = BASELINE configs[3]: 80M rows on 8 GPUs); each step then also does the
per-shard top-k all-gather over RCCL and the k-way merge.  Rank 0 prints ONE JSON line.

Extra legs (rank 0, N = 1 only, outside the timed region; each leg is skipped — and says so — once --time-budget is used up):
  roofline      HIP events recorded by the library around every scan-kernel launch on the launch stream
  batched / pruned_scan / prefilter   the other brute-force kernels at the same 10M x 768 shape
  cpu_baseline  the reference's own BruteforceSearch (oracle/_ref, AVX-512 path) built over the FULL corpus on the host: measured,
                un-scaled, 1 thread and all hardware threads (thread start outside the timed region)
  parity        GPU result vs that CPU result on the same FULL corpus (ids must be identical, distance bits too): every query through
                the timed batch-1 kernel, then in one batched call; ip / l2 / cosine each against a reference index of that metric
  hnsw          BASELINE configs[2] (scaled to --hnsw-rows): graph built here by the product's concurrent builder, searched on the GPU and by
                the reference's own engine on the same graph (tools/bench_hnsw.py)
  hybrid        BASELINE configs[4]: BM25 merge + KNN + RRF fusion on the GPU vs the reference's merger / brute force / rank merger
                (tools/bench_hybrid.py)
  ft_packed     index-commit side of the ft half: a dictionary's PackedIdRelVec posting streams decoded on the device vs the per-word host path
                (tools/bench_ft_packed.py)
--scaling strong (or RXGPU_BENCH_SCALING=strong): BASELINE configs[3] with the corpus FIXED at --total-rows (80M) and split over the ranks.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))

from reindexer_amd import capi  # noqa: E402  (no fallback: raises if librxgpu.so is missing)
from reindexer_amd.sharded import ShardedBruteforceGpu  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6.3 TB/s achievable


def pmc_traffic(algo_bytes: float):
    """HBM bytes per scan launch from the committed rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE runs of
    this same command, corrected as the microarch guide prescribes; see tools/summarize_prof.py).  Only reported when
    the profile is for this workload size (PMC cannot be collected from inside the timed run)."""
    best = None
    for p in sorted((ROOT / "profiles").glob("r*_rocprof_summary.json")):
        try:
            summ = json.loads(p.read_text())
        except Exception:
            continue
        # (a kernel launched over inputs of several sizes in the profiled command: the class of its largest launches, not the mean over all)
        for name, e in summ.get("kernels", {}).items():
            t = e.get("largest_class_hbm_traffic_bytes_per_launch") or e.get("hbm_traffic_bytes_per_launch")
            if "knn_scan" in name and t and abs(t / algo_bytes - 1.0) < 0.05:
                best = (t, f"profiles/{p.name}")
    return best


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=10_000_000, help="rows per GPU (default: the BASELINE 10M)")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", default="ip", choices=["l2", "ip", "cosine"])
    ap.add_argument("--cpu-sample-rows", type=int, default=1_000_000, help="only used when oracle/_ref is absent (never on the driver's box: "
                                                                            "the prebuilt reference library travels with the repo)")
    ap.add_argument("--cpu-queries", type=int, default=8, help="queries of the 1-thread CPU leg = queries of the full-size parity check")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the all-core CPU leg (0 = every CPU the container may use: "
                                                               "min(affinity, cgroup quota), tools/cpu_scaling.py)")
    ap.add_argument("--cpu-per-thread", type=int, default=8)
    ap.add_argument("--parity-other-queries", type=int, default=2, help="full-size parity of the two metrics that are not --metric: queries each "
                                                                         "(a reference index per metric is built over all rows; 0 = skip)")
    ap.add_argument("--parity-deadline", type=float, default=150.0, help="no further per-metric reference index is built once the CPU leg "
                                                                          "has run this many seconds")
    ap.add_argument("--cpu-deadline", type=float, default=40.0, help="all-core CPU leg: threads stop STARTING searches after this many seconds")
    ap.add_argument("--scaling", default=os.environ.get("RXGPU_BENCH_SCALING"), choices=["weak", "strong"],
                    help="default: weak at N = 1 (BASELINE configs[1], 10M rows); STRONG at N > 1 — BASELINE configs[3]: the 80M-row corpus FIXED and "
                         "split over the ranks, value = queries/s over that whole corpus")
    ap.add_argument("--total-rows", type=int, default=80_000_000, help="--scaling strong: the fixed corpus, split over the ranks")
    ap.add_argument("--hnsw-rows", type=int, default=10_000_000, help="hnsw leg: graph size — BASELINE configs[2] at its true size when the time budget "
                                                                      "allows the host build, else --hnsw-fallback-rows (0 = skip)")
    ap.add_argument("--hnsw-fallback-rows", type=int, default=1_000_000)
    ap.add_argument("--hnsw-queries", type=int, default=16384)
    ap.add_argument("--hybrid-docs", type=int, default=5_000_000, help="hybrid leg: documents = vectors (0 = skip)")
    ap.add_argument("--ft-packed-words", type=int, default=100_000, help="ft_packed leg: dictionary words whose PackedIdRelVec streams are decoded "
                                                                         "on the device (0 = skip)")
    ap.add_argument("--time-budget", type=float, default=1350.0, help="seconds of wall clock after which remaining extra legs are skipped "
                                                                      "(the 10M-row HNSW host build alone takes 6 - 9 minutes; without it the run takes ~3)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline / parity leg")
    ap.add_argument("--batch", type=int, default=256, help="extra leg (N=1, untimed region): batched queries on the MFMA path; 0 = skip")
    ap.add_argument("--batch-iters", type=int, default=3)
    ap.add_argument("--in-process", action="store_true",
                    help="N GPUs driven by THIS process through the C-ABI alone (rxgpu_index_create_sharded: one RCCL communicator inside "
                         "librxgpu.so, ncclAllGather of the per-shard lists, merge on device 0) — the path the C++ Map uses; no torch.distributed")
    ap.add_argument("--shards-per-gpu", type=int, default=1, help="--in-process: shards per listed device (a 1-GPU box can run the N-shard path)")
    ap.add_argument("--full-json", default=str(ROOT / "gpurun_out" / "bench_full.json"),
                    help="every leg in full goes to this file (and to stderr); stdout carries ONE compact line (< 8 KB)")
    return ap.parse_args()


_DROP_KEYS = {"note", "against", "host", "kernels", "builder", "params", "value_definition", "traffic_source", "ms_fusion_note",
              "ms_fusion_prepare_note", "heap_kernel", "streaming_session", "host_path", "dtype_note", "per_thread_target", "deadline_s",
              "leg_seconds", "corpus_gen_seconds", "index_load_seconds", "numa_interleaved", "peak", "unit", "bound", "launches",
              "algorithmic_bytes_per_launch", "bytes_per_launch", "bytes_read_per_launch", "algorithmic_f32_bytes_per_launch"}
_KEEP_UNITS_AT = {"roofline", "cpu_baseline"}   # the contract's two objects keep every field the contract names


_DROP_KEYS_TIGHT = {"deviations_from_survey_8d", "workload", "sample", "checked", "queries", "recall_queries", "equal_to_reference_checked", "device_batches",
                    "avg_batch", "posted", "fusion_kernel_launches", "what"}


def compact(obj, depth=0, top_key=None, tight=False):
    """The driver keeps the last 8 KB of stdout: the printed line carries every leg's numbers (frac, avg_ms, rates, parity flags), the prose
    and the per-launch byte counts stay in --full-json."""
    if isinstance(obj, dict):
        out = {}
        for k, v in obj.items():
            if depth > 0 and k in _DROP_KEYS and not (depth == 1 and top_key in _KEEP_UNITS_AT and k not in {"note", "host", "index_load_seconds"}):
                continue
            if tight and depth > 1 and k in _DROP_KEYS_TIGHT and top_key not in _KEEP_UNITS_AT and top_key != "config":
                continue
            out[k] = compact(v, depth + 1, k if depth == 0 else top_key, tight)
        return out
    if isinstance(obj, list):
        return [compact(v, depth + 1, top_key, tight) for v in obj]
    if isinstance(obj, float):
        return float(f"{obj:.5g}")
    if isinstance(obj, str) and len(obj) > (48 if tight and top_key not in _KEEP_UNITS_AT and top_key != "config" else 110) and not (top_key == "cpu_baseline" and depth == 2):
        return obj[:(45 if tight and top_key not in _KEEP_UNITS_AT and top_key != "config" else 107)] + "..."
    return obj


def emit(result: dict, args) -> None:
    full = json.dumps(result)
    try:
        Path(args.full_json).parent.mkdir(parents=True, exist_ok=True)
        Path(args.full_json).write_text(full + "\n")
    except OSError:
        pass
    print(full, file=sys.stderr, flush=True)
    small = compact(result)
    small["full_json"] = os.path.relpath(args.full_json, ROOT) if str(args.full_json).startswith(str(ROOT)) else str(args.full_json)
    line = json.dumps(small, separators=(",", ":"))
    if len(line) >= 7600:   # second pass: the numbers of every leg stay, counts / descriptions inside the legs go (they are in --full-json)
        small = compact(result, tight=True)
        small["full_json"] = os.path.relpath(args.full_json, ROOT) if str(args.full_json).startswith(str(ROOT)) else str(args.full_json)
        line = json.dumps(small, separators=(",", ":"))
    for victim in ("ft_packed", "prefilter", "pruned_scan", "hybrid", "hnsw"):   # never reached at today's sizes (~5 KB); a hard guarantee anyway
        if len(line) < 7600:
            break
        small[victim] = {"see": small["full_json"]}
        line = json.dumps(small, separators=(",", ":"))
    print(line, flush=True)


def make_corpus(rows: int, dim: int, seed: int, device) -> torch.Tensor:
    """float32 i.i.d. N(0, 0.25^2) — the distribution of the reference's own tests (gtests/tools.h:121-129)."""
    out = torch.empty((rows, dim), dtype=torch.float32, device=device)
    g = torch.Generator(device=device)
    chunk = 1 << 20
    for i, start in enumerate(range(0, rows, chunk)):
        g.manual_seed(seed * 1_000_003 + i)
        n = min(chunk, rows - start)
        out[start:start + n].normal_(0.0, 0.25, generator=g)
    return out


def host_cpu_info() -> dict:
    """CPU model / hardware threads of the box the baseline ran on, and the reference's SIMD level."""
    from cpu_scaling import effective_cpus, host_limits
    lim = host_limits()
    info = {"hardware_threads": os.cpu_count(), "usable_cpus": effective_cpus(), "cgroup_cpu_max": lim.get("cgroup_v2_cpu_max"),
            "numa_nodes": lim.get("numa_online"), "RX_TARGET_INSTRUCTIONS": os.environ.get("RX_TARGET_INSTRUCTIONS", "avx512")}
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    info["model"] = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return info


def _numa_interleave(on: bool) -> bool:
    """set_mempolicy(MPOL_INTERLEAVE over every allowed node) for pages this thread touches from now on / back to the default policy.
    The reference allocates its row array with malloc and fills it from one thread; on a two-socket host first-touch would put all 30.8 GB
    on one socket and halve what the all-core baseline can read.  Raw syscalls (no libnuma in the image); best effort."""
    try:
        libc = ctypes.CDLL(None, use_errno=True)
        if not on:
            return libc.syscall(ctypes.c_long(238), ctypes.c_long(0), None, ctypes.c_ulong(0)) == 0
        mask = (ctypes.c_ulong * 16)()
        if libc.syscall(ctypes.c_long(239), None, mask, ctypes.c_ulong(1024), None, ctypes.c_ulong(4)) != 0:   # get_mempolicy(MPOL_F_MEMS_ALLOWED)
            return False
        if sum(bin(w).count("1") for w in mask) < 2:
            return False
        return libc.syscall(ctypes.c_long(238), ctypes.c_long(3), mask, ctypes.c_ulong(1024)) == 0   # set_mempolicy(MPOL_INTERLEAVE)
    except Exception:
        return False


def cpu_baseline_and_parity(args, corpus: torch.Tensor, queries: torch.Tensor, metric_id: int, ix):
    """Full-size CPU leg: the reference's BruteforceSearch over ALL rows of the corpus (bruteforce.cc:103-127), measured — nothing is scaled."""
    from oracle import pyoracle  # checker / baseline only
    t_leg = time.perf_counter()
    ref = pyoracle.ref_or_none()
    if ref is None or ref.simd_level != 3 or not hasattr(ref.L, "ref_bf_search_knn_mt"):
        return cpu_baseline_port_sample(args, corpus, queries, metric_id)
    rows = corpus.shape[0]
    nq = min(args.cpu_queries, queries.shape[0])
    from cpu_scaling import effective_cpus
    ncores = effective_cpus()   # what the container may really use: min(affinity, cgroup CPU quota) — NOT os.cpu_count()
    orc = pyoracle.Oracle()
    host_q = queries[:max(nq, 64)].cpu().numpy()
    if metric_id == 2:
        host_q = np.stack([orc.normalize_copy(q)[0] for q in host_q])
    t0 = time.perf_counter()
    interleaved = _numa_interleave(True)
    bf = pyoracle.RefBruteforce(ref, metric_id, args.dim, rows)
    chunk = 1 << 20
    for a in range(0, rows, chunk):
        b = min(rows, a + chunk)
        bf.add(corpus[a:b].cpu().numpy(), np.arange(a, b, dtype=np.uint64) << np.uint64(32))
    _numa_interleave(False)
    load_s = time.perf_counter() - t0

    t0 = time.perf_counter()
    cpu_res = [bf.search_knn(host_q[i], args.k) for i in range(nq)]
    qps_1 = nq / (time.perf_counter() - t0)
    threads = args.cpu_threads or ncores
    secs, done = bf.search_knn_mt(host_q, args.k, threads, args.cpu_per_thread, deadline_s=args.cpu_deadline)
    qps_all = done / secs
    bf.close()
    row_bytes = rows * args.dim * 4
    baseline = {
        "value": qps_1, "unit": "queries/s", "cores": 1, "kind": "reference",
        "sample": f"measured, un-scaled: the reference's hnswlib::BruteforceSearch (AVX-512) over all {rows} rows, {nq} queries, k={args.k}, one thread",
        "gbps_per_core": qps_1 * row_bytes / 1e9,
        "all_cores": {"value": qps_all, "cores": threads, "queries": done, "seconds": secs, "gbps": qps_all * row_bytes / 1e9,
                      "per_thread_target": args.cpu_per_thread, "deadline_s": args.cpu_deadline,
                      "note": "T threads, each scanning the shared index for its own query (the reference's concurrency model, "
                              "gtests/tests/unit/float_vector_index.cc:258-294); threads are created before the clock starts; T = the CPUs "
                              "the container may use (cgroup quota, see host) — more threads only add throttling (profiles/r2b_cpu_scaling.json)"},
        "numa_interleaved": interleaved, "index_load_seconds": load_s, "host": host_cpu_info(),
    }
    # parity on the FULL corpus: GPU through the C-ABI vs the reference engine.  The kernel that `value` times is the batch-1 scan
    # (knn_scan_fixed): the queries go through it one call each (nq = 1); the same queries in ONE call (nq = 8: bf16 nomination +
    # exact re-score) are the second check.  Then the other two metrics over the same resident rows, each against a reference index
    # of that metric built over all rows.
    parity = _compare_with_reference(ix, host_q[:nq], cpu_res, args.k)
    parity.update({"rows": rows, "path": "batch1", "metric": args.metric,
                   "against": "reference BruteforceSearch over the full corpus; every query through the timed batch-1 kernel (nq = 1 per call)"})
    parity["batched_call"] = _compare_with_reference(ix, host_q[:nq], cpu_res, args.k, one_call=True)
    per_metric = {args.metric: {k_: parity[k_] for k_ in ("ids_equal_frac", "max_ulps_dist", "queries")}}
    nq_other = min(args.parity_other_queries, nq)
    for other in ("ip", "l2", "cosine"):
        if other == args.metric or nq_other == 0:
            continue
        if time.perf_counter() - t_leg > args.parity_deadline:
            per_metric[other] = {"skipped": "parity deadline (--parity-deadline)"}
            continue
        per_metric[other] = _parity_other_metric(args, corpus, queries, capi.METRICS[other], nq_other, ref, ncores)
    parity["per_metric"] = per_metric
    return baseline, parity


def _compare_with_reference(ix, host_q, cpu_res, k, one_call=False):
    labels_of = lambda r: r.astype(np.uint64) << np.uint64(32)  # noqa: E731
    nq = host_q.shape[0]
    if one_call:
        dist, row, cnt = ix.search_knn(host_q, k + 1)
    else:
        parts = [ix.search_knn(host_q[i:i + 1], k + 1) for i in range(nq)]
        dist, row = np.concatenate([p_[0] for p_ in parts]), np.concatenate([p_[1] for p_ in parts])
    ids_equal, max_ulps, rec = 0, 0, 0.0
    for i in range(nq):
        wd, wl = cpu_res[i]
        gl = labels_of(row[i, :k])
        ids_equal += int(np.array_equal(gl, wl))
        ulps = np.abs(dist[i, :k].view(np.int32).astype(np.int64) - wd.view(np.int32).astype(np.int64))
        max_ulps = max(max_ulps, int(ulps.max()))
        rec += len(set(gl.tolist()) & set(wl.tolist())) / k
    return {"ids_equal_frac": ids_equal / nq, "max_ulps_dist": max_ulps, "queries": nq, "recall_at_k": rec / nq,
            "path": "nq=%d in one call (bf16 nomination + exact re-score)" % nq if one_call else "batch1"}


def _parity_other_metric(args, corpus, queries, metric_id, nq, ref, threads):
    """The same resident rows under another metric: a reference index of that metric over ALL rows vs the batch-1 scan of that metric.
    Cosine: the query is normalised on the host and 1/|row| comes from the product's own AddNorm arithmetic (hostapi)."""
    from oracle import pyoracle
    from reindexer_amd import hostapi
    rows = corpus.shape[0]
    host_q = queries[:nq].cpu().numpy()
    if metric_id == 2:
        host_q = np.stack([hostapi.normalize_copy(q)[0] for q in host_q])
    inv = np.empty(rows, np.float32) if metric_id == 2 else None
    _numa_interleave(True)
    bf = pyoracle.RefBruteforce(ref, metric_id, args.dim, rows)
    chunk = 1 << 20
    for a in range(0, rows, chunk):
        b = min(rows, a + chunk)
        blk = corpus[a:b].cpu().numpy()
        bf.add(blk, np.arange(a, b, dtype=np.uint64) << np.uint64(32))
        if inv is not None:
            inv[a:b] = hostapi.l2_modules_many(blk, threads)
    _numa_interleave(False)
    cpu_res = [bf.search_knn(host_q[i], args.k) for i in range(nq)]
    bf.close()
    d_inv = torch.from_numpy(inv).to(corpus.device) if inv is not None else None
    with capi.VectorIndex(metric_id, args.dim, device=corpus.device.index or 0) as mx:
        mx.adopt_device_rows(corpus.data_ptr(), rows, args.dim, d_inv.data_ptr() if d_inv is not None else None, keepalive=(corpus, d_inv))
        out = _compare_with_reference(mx, host_q, cpu_res, args.k)
    return {k_: out[k_] for k_ in ("ids_equal_frac", "max_ulps_dist", "queries")}


def cpu_baseline_port_sample(args, corpus: torch.Tensor, queries: torch.Tensor, metric_id: int):
    """Fallback when oracle/_ref is absent: the plain-C port on a row-prefix sample, scaled (linear scan) — labelled as such."""
    from oracle import pyoracle
    s_rows = min(args.cpu_sample_rows, corpus.shape[0])
    nq = min(args.cpu_queries, queries.shape[0])
    host_rows = corpus[:s_rows].cpu().numpy()
    host_q = queries[:nq].cpu().numpy()
    labels = np.arange(s_rows, dtype=np.uint64) << np.uint64(32)
    orc = pyoracle.Oracle()
    inv = orc.l2_modules(host_rows) if metric_id == 2 else None
    if metric_id == 2:
        host_q = np.stack([orc.normalize_copy(q)[0] for q in host_q])
    t0 = time.perf_counter()
    cpu_res = [orc.bf_search_knn(metric_id, host_rows, labels, inv, host_q[i], args.k) for i in range(nq)]
    qps_1 = nq / (time.perf_counter() - t0)
    scale = s_rows / corpus.shape[0]
    baseline = {"value": qps_1 * scale, "unit": "queries/s", "cores": 1, "kind": "port",
                "sample": f"{s_rows}-row prefix, {nq} queries; measured {qps_1:.3f} q/s on the sample, SCALED by {s_rows}/{corpus.shape[0]} rows "
                          "(oracle/_ref absent: plain-C port)", "host": host_cpu_info()}
    with capi.VectorIndex(metric_id, args.dim) as sx:
        d_inv = torch.from_numpy(inv).to(corpus.device) if metric_id == 2 else None
        sx.adopt_device_rows(corpus.data_ptr(), s_rows, corpus.shape[1], d_inv.data_ptr() if d_inv is not None else None, keepalive=(corpus, d_inv))
        dist, row, cnt = sx.search_knn(host_q, args.k + 1)
    ids_equal = sum(int(np.array_equal(labels[row[i, :args.k]], cpu_res[i][1])) for i in range(nq))
    parity = {"ids_equal_frac": ids_equal / nq, "queries": nq, "rows": s_rows, "against": "plain-C port on a row prefix"}
    return baseline, parity


def batched_leg(args, ix, queries: torch.Tensor, device, kk: int):
    """BASELINE configs[1] 'batch=256 (MFMA path)': B queries per call, corpus streamed once per call.
    Reported beside the headline (which stays batch=1): queries/s, the nomination GEMM against its own roofline (batches > 64 queries:
    bf16 matrix cores over the bf16 shadow of the rows — dense bf16 peak 2.5 PFLOP/s, 2 bytes per element from HBM; smaller batches:
    f32-input MFMA, 157.3 TFLOP/s), and agreement of the batched result with the batch-1 exact path on the same queries
    (the nomination only prunes under a rigorous bound; every returned distance is the exact f32 value)."""
    B = min(args.batch, queries.shape[0])
    q = queries[:B].contiguous()
    od = torch.empty((B, kk), dtype=torch.float32, device=device)
    orow = torch.empty((B, kk), dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream(device).cuda_stream
    ix.search_knn_device(q.data_ptr(), B, kk, od.data_ptr(), orow.data_ptr(), None, stream)   # warmup (row statistics, buffers)
    torch.cuda.synchronize(device)
    ix.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.batch_iters):
        ix.search_knn_device(q.data_ptr(), B, kk, od.data_ptr(), orow.data_ptr(), None, stream)
    torch.cuda.synchronize(device)
    t1 = time.perf_counter()
    n_gemm, ms_gemm = ix.profile_read("gemm")
    n_res, ms_res = ix.profile_read("rescore")
    n_fb, ms_fb = ix.profile_read("fallback_scan")
    ix.profile_enable(False)
    # same queries through the batch-1 exact path
    sd = torch.empty((B, kk), dtype=torch.float32, device=device)
    srow = torch.empty((B, kk), dtype=torch.int32, device=device)
    nchk = min(B, 32)
    for i in range(nchk):
        ix.search_knn_device(q.data_ptr() + i * args.dim * 4, 1, kk, sd.data_ptr() + i * kk * 4, srow.data_ptr() + i * kk * 4, None, stream)
    torch.cuda.synchronize(device)
    same_rows = bool(torch.equal(srow[:nchk], orow[:nchk]))
    same_bits = bool(torch.equal(sd[:nchk].view(torch.int32), od[:nchk].view(torch.int32)))
    per_batch = (t1 - t0) / args.batch_iters
    bf16_min = int(os.environ.get("RXGPU_BATCH_BF16_MIN", "2"))
    bf16 = bf16_min > 0 and B >= bf16_min
    mt = (128 if B <= 128 else 256) if bf16 else (32 if B <= 32 else 64 if B <= 64 else 128 if B <= 128 else 256)
    kpad = (args.dim + 63) // 64 * 64 if bf16 else args.dim
    flops = 2.0 * mt * args.rows * kpad       # flops actually issued on the matrix cores (padded to the tile)
    gemm_ms = ms_gemm / max(n_gemm, 1)
    peak = 2500.0 if bf16 else 157.3
    roof = {"bound": "mfma", "achieved": flops / (gemm_ms / 1e3) / 1e12 if n_gemm else None, "peak": peak, "unit": "TFLOP/s",
            "frac": flops / (gemm_ms / 1e3) / 1e12 / peak if n_gemm else None,
            "kernel": ("knn_gemm_bf16_qreg<FILTER>" if os.environ.get("RXGPU_GEMM_QREG", "1") != "0" and os.environ.get("RXGPU_GEMM_SPLIT", "1") != "0"
                       else "knn_gemm_bf16_split<FILTER>" if os.environ.get("RXGPU_GEMM_SPLIT", "1") != "0" else "knn_gemm_bf16_glds<FILTER>") if bf16 else "knn_gemm<FILTER>", "avg_ms": gemm_ms, "launches": n_gemm,
            "dtype": "bf16 nomination (v_mfma_f32_32x32x16_bf16) + exact f32 re-score" if bf16 else "f32 (v_mfma_f32_32x32x2_f32)"}
    if bf16 and n_gemm:
        # north_star: "MFMA only for the batched-query x corpus GEMM case, evidenced ... vs the fp32 roofline": the whole batch (nomination +
        # exact re-score) against the dense fp32-MFMA peak, and the calibration against the vendor GEMM of the same shape on the same chip
        useful = 2.0 * B * args.rows * args.dim
        roof["vs_fp32_mfma_peak"] = {"end_to_end_tflops": useful / per_batch / 1e12, "fp32_mfma_peak_tflops": 157.3, "times_the_peak": useful / per_batch / 1e12 / 157.3}
        roof["vendor_gemm_calibration"] = {"hipblaslt_ms": 4.70, "hipblaslt_ms_minus_product_write": 4.06, "this_kernel_ms_same_process": 4.56,
                                           "shader_clock_ghz_under_both": "1.60-1.70", "file": "profiles/rd6_gemm_vendor.json"}
        shadow = float(args.rows) * kpad * 2
        roof["hbm"] = {"algorithmic_bytes_per_launch": shadow, "achieved": shadow / (gemm_ms / 1e3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                       "frac": shadow / (gemm_ms / 1e3) / 1e9 / 8000.0}
    return {"batch": B, "queries_per_sec": B / per_batch, "ms_per_batch": per_batch * 1e3,
            "roofline": roof,
            "rescore_ms": ms_res / max(n_res, 1), "fallback_scan_ms": ms_fb / max(n_fb, 1),
            "equals_batch1_rows": same_rows, "equals_batch1_dist_bits": same_bits, "checked_queries": nchk}


def pruned_leg(args, ix, queries: torch.Tensor, device, kk: int):
    """Opt-in batch-1 variant (RXGPU_SCAN_BF16=1): the scan reads a bf16 shadow of the rows (2 bytes per element) to PRUNE under a rigorous
    rounding bound, the exact f32 kernels re-score the few dozen survivors — identical rows and distance bits, half the HBM traffic.
    Not the headline: `value` above is the plain f32 scan.  Costs +50 % HBM footprint (the shadow)."""
    stream = torch.cuda.current_stream(device).cuda_stream
    nq = min(16, queries.shape[0])
    od = torch.empty((nq, kk), dtype=torch.float32, device=device)
    orow = torch.empty((nq, kk), dtype=torch.int32, device=device)
    sd, srow = torch.empty_like(od), torch.empty_like(orow)
    for i in range(nq):   # exact path first
        ix.search_knn_device(queries.data_ptr() + i * args.dim * 4, 1, kk, sd.data_ptr() + i * kk * 4, srow.data_ptr() + i * kk * 4, None, stream)
    torch.cuda.synchronize(device)
    os.environ["RXGPU_SCAN_BF16"] = "1"
    try:
        ix.search_knn_device(queries.data_ptr(), 1, kk, od.data_ptr(), orow.data_ptr(), None, stream)   # builds the shadow
        torch.cuda.synchronize(device)
        ix.profile_enable(True)
        iters = 50
        t0 = time.perf_counter()
        for it in range(iters):
            i = it % nq
            ix.search_knn_device(queries.data_ptr() + i * args.dim * 4, 1, kk, od.data_ptr() + i * kk * 4, orow.data_ptr() + i * kk * 4, None, stream)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        n_scan, ms_scan = ix.profile_read("scan_bf16")
        n_f, ms_f = ix.profile_read("filter_approx")
        n_r, ms_r = ix.profile_read("rescore")
        ix.profile_enable(False)
    finally:
        os.environ.pop("RXGPU_SCAN_BF16", None)
    if not n_scan:
        return {"skipped": "dimension not supported by the bf16 scan"}
    scan_ms = ms_scan / n_scan
    kpad = (args.dim + 63) // 64 * 64
    shadow = float(args.rows) * kpad * 2
    algo = float(args.rows) * args.dim * 4
    return {"queries_per_sec": iters / (t1 - t0), "ms_per_query": (t1 - t0) / iters * 1e3,
            "roofline": {"bound": "hbm", "kernel": "knn_scan_bf16", "avg_ms": scan_ms, "launches": n_scan, "peak": 8000.0, "unit": "GB/s",
                         "bytes_read_per_launch": shadow + args.rows * 4.0, "achieved": (shadow + args.rows * 4.0) / (scan_ms / 1e3) / 1e9,
                         "frac": (shadow + args.rows * 4.0) / (scan_ms / 1e3) / 1e9 / 8000.0,
                         "algorithmic_f32_bytes_per_launch": algo, "equivalent_f32_rate": algo / (scan_ms / 1e3) / 1e9,
                         "note": "bytes = bf16 shadow read + one approximate distance written per row; the f32 rows are only gathered for the survivors"},
            "filter_ms": ms_f / max(n_f, 1), "rescore_ms": ms_r / max(n_r, 1),
            "equals_exact_rows": bool(torch.equal(srow, orow)), "equals_exact_dist_bits": bool(torch.equal(sd.view(torch.int32), od.view(torch.int32))),
            "checked_queries": nq}


def prefilter_leg(args, ix, queries: torch.Tensor, device, kk: int):
    """Pre-filtered scan (`WHERE cond AND KNN(...)`): only the rows of a sorted row list compete and only they are read.
    Row lists resident in HBM; per density the device time per query and the rate on the bytes that have to move (allowed x (D*4 + 4)).
    Density 1.0 must reproduce the unfiltered search bit for bit.  Not the headline."""
    stream = torch.cuda.current_stream(device).cuda_stream
    nq = min(8, queries.shape[0])
    sd = torch.empty((nq, kk), dtype=torch.float32, device=device)
    srow = torch.empty((nq, kk), dtype=torch.int32, device=device)
    for i in range(nq):
        ix.search_knn_device(queries.data_ptr() + i * args.dim * 4, 1, kk, sd.data_ptr() + i * kk * 4, srow.data_ptr() + i * kk * 4, None, stream)
    od, orow = torch.empty_like(sd), torch.empty_like(srow)
    g = torch.Generator(device=device)
    g.manual_seed(11)
    out = []
    for dens in (0.01, 0.1, 1.0):
        if dens >= 1.0:
            ids = torch.arange(args.rows, dtype=torch.int32, device=device)
        else:
            ids = (torch.rand(args.rows, device=device, generator=g) < dens).nonzero().flatten().to(torch.int32)
        n_ids = int(ids.numel())
        if n_ids == 0:
            continue

        def run(i):
            ix.search_knn_subset_device(queries.data_ptr() + i * args.dim * 4, 1, kk, ids.data_ptr(), n_ids, od.data_ptr() + i * kk * 4,
                                        orow.data_ptr() + i * kk * 4, None, stream)
        run(0)
        torch.cuda.synchronize(device)
        ix.profile_enable(True)
        for i in range(nq):
            run(i)
        torch.cuda.synchronize(device)
        n_scan, ms_scan = ix.profile_read("scan_subset")
        ix.profile_enable(False)
        ms = ms_scan / max(n_scan, 1)
        moved = float(n_ids) * (args.dim * 4 + 4)
        member = torch.zeros(args.rows, dtype=torch.bool, device=device)
        member[ids.long()] = True
        entry = {"density": dens, "allowed_rows": n_ids, "kernel": "knn_scan_subset", "avg_ms": ms, "launches": n_scan,
                 "bytes_per_launch": moved, "achieved": moved / (ms / 1e3) / 1e9 if n_scan else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": moved / (ms / 1e3) / 1e9 / HBM_PEAK_GBS if n_scan else None,
                 "rows_all_allowed": bool(member[orow.long().flatten()].all())}
        if dens >= 1.0:
            entry["equals_unfiltered_rows"] = bool(torch.equal(orow, srow))
            entry["equals_unfiltered_dist_bits"] = bool(torch.equal(od.view(torch.int32), sd.view(torch.int32)))
        out.append(entry)
        del ids, member
    return {"k": args.k, "queries": nq, "densities": out}


def main_in_process(args, t_start: float) -> None:
    """BASELINE configs[3] behind the C-ABI: ONE process, rxgpu_index_create_sharded over --gpus devices (x --shards-per-gpu), every shard
    adopts rows generated on its own device, a step = rxgpu_search_knn on the sharded handle (host query in, merged global top-k out):
    per-shard scans -> ncclAllGather inside librxgpu.so -> knn_merge_shards on device 0 -> one D2H copy."""
    if int(os.environ.get("RANK", "0")) != 0:
        return   # started under torch.distributed.run by mistake: rank 0 drives every GPU
    metric_id = capi.METRICS[args.metric]
    kk = args.k + 1
    devices = [g for g in range(args.gpus) for _ in range(args.shards_per_gpu)]
    nshards = len(devices)
    if args.scaling is None:
        args.scaling = "strong" if args.gpus > 1 else "weak"
    strong = args.scaling == "strong"
    rows = args.total_rows // nshards if strong else args.rows // args.shards_per_gpu
    rows -= rows % 32   # shard_rows are whole bitmap words
    sx = capi.ShardedVectorIndex(metric_id, args.dim, rows * nshards, devices)
    assert sx.shard_rows == rows, (sx.shard_rows, rows)
    keep = []
    for s, dev in enumerate(devices):
        torch.cuda.set_device(dev)
        device = torch.device("cuda", dev)
        corpus = make_corpus(rows, args.dim, 20260924 + s, device)
        d_inv = 1.0 / torch.linalg.vector_norm(corpus, dim=1) if metric_id == 2 else None
        torch.cuda.synchronize(device)
        sx.shard(s).adopt_device_rows(corpus.data_ptr(), rows, args.dim, d_inv.data_ptr() if d_inv is not None else None)
        keep.append((corpus, d_inv))
    sx.sync_count()
    total_q = args.steps + args.warmup
    rng = np.random.default_rng(7)
    queries = rng.normal(0.0, 0.25, (total_q, args.dim)).astype(np.float32)
    if metric_id == 2:
        queries /= np.linalg.norm(queries, axis=1, keepdims=True)
    for i in range(args.warmup):
        sx.search_knn(queries[i:i + 1], kk)
    views = [sx.shard(s) for s in range(nshards)]
    for v in views:
        v.profile_enable(True)
    c0 = sx.collectives
    t0 = time.perf_counter()
    for i in range(args.warmup, total_q):
        dist, row, cnt = sx.search_knn(queries[i:i + 1], kk)
    elapsed = time.perf_counter() - t0
    collectives = sx.collectives - c0
    per_shard = [v.profile_read("scan") for v in views]
    for v in views:
        v.profile_enable(False)
    # parity of the last query against every shard searched on its own (single-device entry point) and merged here under (dist, global row)
    cand = []
    for s, v in enumerate(views):
        d1, r1, _ = v.search_knn(queries[total_q - 1:total_q], kk)
        cand += [(float(d), int(r) + s * rows) for d, r in zip(d1[0], r1[0])]
    cand.sort()
    merged_ok = [c[1] for c in cand[:kk]] == [int(r) for r in row[0]] and [np.float32(c[0]).view(np.uint32) for c in cand[:kk]] == list(dist[0].view(np.uint32))
    qps = args.steps / elapsed
    algo_bytes = rows * args.dim * 4
    launches = sum(n for n, _ in per_shard)
    avg_scan_s = sum(ms for _, ms in per_shard) / 1e3 / max(launches, 1)
    achieved = algo_bytes / avg_scan_s / 1e9 if launches else 0.0
    result = {
        "metric": "knn_queries_per_sec", "value": qps, "unit": "queries/s", "n_gpus": args.gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": (f"brute-force KNN, {rows * nshards} x {args.dim} fp32 row-sharded over {args.gpus} GPU(s) x {args.shards_per_gpu} shard(s) "
                         f"({rows} rows each), metric={args.metric}, k={args.k}, batch=1 (BASELINE configs[3], in-process C-ABI path)"),
            "rows_per_shard": rows, "total_rows": rows * nshards, "dim": args.dim, "k": args.k, "batch": 1,
            "sharding": "rxgpu_index_create_sharded: per-shard scans, ncclAllGather of kk x 8 B per shard inside librxgpu.so, merge kernel on device 0",
            "merge_mode": sx.merge_mode, "rccl_ranks": sx.ranks, "collectives_in_timed_region": collectives,
            "merged_equals_per_shard_lists": bool(merged_ok), "qps_over_full_corpus": qps, "shard_scans_per_sec": qps * nshards,
            "arch": capi.device_arch(0),
            "host_boundary": "queries enter as host pointers (3 KB H2D per step) and the merged list leaves by one D2H copy (88 B): both inside the timed region",
        },
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "knn_scan_fixed", "launches": launches, "avg_ms": avg_scan_s * 1e3, "algorithmic_bytes_per_launch": algo_bytes,
                     "scan_fraction_of_step": avg_scan_s / (elapsed / args.steps)},
        "bench_wall_seconds": time.perf_counter() - t_start,
    }
    emit(result, args)
    sx.close()
    del keep


def sharded_in_process_leg(args, ix, corpus, d_inv, queries, device, kk):
    """BASELINE configs[3]'s code path on this one GPU (a driver SCALE run needs a multi-GPU node): the SAME resident corpus as two row-range
    shards of one rxgpu_index_create_sharded handle — what the C++ Map builds from RX_GPU_VECTOR_INDEXES=0,0 — each step = host query in ->
    both shards' scans -> ONE ncclAllGather of kk x 8 B per shard inside librxgpu.so (RCCL communicator of the index) -> knn_merge_shards ->
    one D2H copy.  Asserted: merge mode 1 (device exchange), collectives == steps, and ids + distance bits identical to the unsharded index."""
    n = args.rows
    half = n // 2
    if n % 64:
        return {"skipped": "rows not a multiple of 64"}
    metric_id = capi.METRICS[args.metric]
    sx = capi.ShardedVectorIndex(metric_id, args.dim, n, [device.index, device.index])
    try:
        assert sx.shard_rows == half, (sx.shard_rows, half)
        esz = 4
        for s in range(2):
            sx.shard(s).adopt_device_rows(corpus.data_ptr() + s * half * args.dim * esz, half, args.dim,
                                          d_inv.data_ptr() + s * half * esz if d_inv is not None else None)
        sx.sync_count()
        steps = min(max(8, min(args.steps, 32)), int(queries.shape[0]) - 2)
        hq = queries[:steps + 2].cpu().numpy()
        for i in range(2):
            sx.search_knn(hq[i:i + 1], kk)
        c0 = sx.collectives
        got = []
        t0 = time.perf_counter()
        for i in range(2, steps + 2):
            got.append(sx.search_knn(hq[i:i + 1], kk))
        elapsed = time.perf_counter() - t0
        collectives = sx.collectives - c0
        same = 0
        for i, (d1, r1, c1) in enumerate(got):
            d0, r0, c0_ = ix.search_knn(hq[i + 2:i + 3], kk)
            same += int(np.array_equal(r0, r1) and np.array_equal(d0.view(np.uint32), d1.view(np.uint32)) and np.array_equal(c0_, c1))
        return {"workload": f"the headline corpus ({n} x {args.dim}) as 2 row-range shards on device {device.index}, batch = 1, host query in / merged list out",
                "shards": 2, "shard_merge_mode": 1 if sx.merge_mode == "rccl" else 0, "merge_note": sx.merge_note, "rccl_ranks": sx.ranks,
                "steps": steps, "collectives": collectives, "collectives_equal_steps": collectives == steps,
                "identical": same == steps and sx.merge_mode == "rccl" and collectives == steps, "identical_queries": same,
                "ms_per_query": elapsed / steps * 1e3, "queries_per_sec": steps / elapsed,
                "note": "both shards share one GPU here, so the scans run back to back: the figure prices the fan-out + RCCL + merge + copies around "
                        "one full scan, not a speedup"}
    finally:
        sx.close()


def main():
    t_start = time.perf_counter()
    args = parse_args()
    if args.in_process:
        return main_in_process(args, t_start)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # RXGPU_BENCH_FORCE_DIST=1 exercises the N > 1 code path (RCCL init, all-gather, device merge) on a single rank
    dist_on = world > 1 or bool(os.environ.get("RXGPU_BENCH_FORCE_DIST"))
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)

    metric_id = capi.METRICS[args.metric]
    kk = args.k + 1  # the Map asks for k+1 to detect a distance tie straddling the k-th boundary
    if args.scaling is None:   # N > 1 is BASELINE configs[3]: the 80M corpus fixed, value = queries/s over ALL of it
        args.scaling = "strong" if world > 1 else "weak"
    strong = args.scaling == "strong"
    if strong:   # BASELINE configs[3]: the corpus is FIXED (80M rows) and split into row ranges, one per rank
        if args.total_rows % world:
            raise SystemExit("--total-rows must be divisible by the number of ranks")
        args.rows = args.total_rows // world
    corpus = make_corpus(args.rows, args.dim, 20260924 + rank, device)
    total_q = args.steps + args.warmup
    gq = torch.Generator(device=device)
    gq.manual_seed(7)  # the same queries on every rank
    # (34: the in-process sharded leg runs 2 + 8..32 queries whatever --steps says)
    queries = torch.empty((max(total_q, args.cpu_queries, args.batch, 34), args.dim), dtype=torch.float32, device=device).normal_(0.0, 0.25, generator=gq)
    d_inv = None
    if metric_id == 2:
        d_inv = 1.0 / torch.linalg.vector_norm(corpus, dim=1)
        queries = queries / torch.linalg.vector_norm(queries, dim=1, keepdim=True)

    ix = capi.VectorIndex(metric_id, args.dim, device=local_rank)
    ix.adopt_device_rows(corpus.data_ptr(), args.rows, args.dim, d_inv.data_ptr() if d_inv is not None else None, keepalive=(corpus, d_inv))
    out_dist = torch.empty((total_q, kk), dtype=torch.float32, device=device)
    out_row = torch.empty((total_q, kk), dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream(device)
    esz = 4
    sharded = ShardedBruteforceGpu(ix, args.rows, device, 1, kk) if dist_on else None

    def step(i: int):
        if dist_on:   # local scan -> one RCCL all-gather of kk*8 B per rank -> device merge; out_row holds GLOBAL rows
            sharded.search_into(queries.data_ptr() + i * args.dim * esz, 1, out_dist[i:i + 1], out_row[i:i + 1])
        else:
            ix.search_knn_device(queries.data_ptr() + i * args.dim * esz, 1, kk, out_dist.data_ptr() + i * kk * esz,
                                 out_row.data_ptr() + i * kk * esz, None, stream.cuda_stream)

    def sync():
        torch.cuda.synchronize(device)
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize(device)

    for i in range(args.warmup):
        step(i)
    sync()
    ix.profile_enable(True)
    t0 = time.perf_counter()
    for i in range(args.warmup, total_q):
        step(i)
    sync()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    launches, scan_ms = ix.profile_read("scan")
    ix.profile_enable(False)

    qps_global = args.steps / elapsed            # queries/s over the whole (N x rows) corpus
    # N > 1: value = queries/s over the WHOLE corpus (every query scans every shard) in either mode — the metric's unit; the aggregate of
    # shard scans (rows_per_gpu-row scans/s, what "weak" used to report as value) is a side field.  N = 1: the two coincide.
    value = qps_global
    one_gpu = None
    if world > 1:
        # The same bytes on ONE GPU, measured here: rank 0 scans its shard `world` times back to back per query (what a single GPU holding the
        # whole corpus does: HBM-bound, rows x world x dim x 4 bytes; the shard is 100x the caches, nothing is reused) -> queries/s of one GPU
        # over a corpus of the full size, the denominator of the speedup.
        reps = max(2, min(8, args.steps))
        sync()
        t0 = time.perf_counter()
        if rank == 0:
            for j in range(reps):
                i = j % total_q
                for _ in range(world):
                    ix.search_knn_device(queries.data_ptr() + i * args.dim * esz, 1, kk, out_dist.data_ptr() + i * kk * esz,
                                         out_row.data_ptr() + i * kk * esz, None, stream.cuda_stream)
        sync()
        one_gpu = reps / (time.perf_counter() - t0)
    algo_bytes = args.rows * args.dim * 4        # N*D*4 per query (labels/norms excluded)
    avg_scan_s = (scan_ms / 1e3) / max(launches, 1)
    achieved = algo_bytes / avg_scan_s / 1e9 if launches else 0.0

    if rank == 0:
        traffic = pmc_traffic(algo_bytes)
        if strong:
            workload = (f"brute-force KNN, {args.total_rows} x {args.dim} fp32 FIXED corpus row-sharded over {world} GPU(s) ({args.rows} rows each), "
                        f"metric={args.metric}, k={args.k}, batch=1 (BASELINE configs[3], strong scaling)")
        else:
            workload = (f"brute-force KNN, {args.rows} x {args.dim} fp32 per GPU, metric={args.metric}, k={args.k}, batch=1 "
                        f"(BASELINE configs[1]{'; x' + str(world) + ' row-sharded = configs[3]' if world > 1 else ''})")
        result = {
            "metric": "knn_queries_per_sec", "value": value, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": workload,
                "rows_per_gpu": args.rows, "total_rows": args.rows * world, "dim": args.dim, "k": args.k, "batch": 1,
                "sharding": "row-range shards, RCCL all-gather of per-shard top-k + merge" if dist_on else "none",
                "rccl_ranks": (dist.get_world_size() if dist_on else 0),
                "value_definition": "queries/s over the whole corpus (total_rows): every query scans every shard",
                "qps_over_full_corpus": qps_global, "shard_scans_per_sec": qps_global * world,
                "one_gpu_same_bytes_qps": one_gpu, "speedup_vs_one_gpu": (qps_global / one_gpu if one_gpu else None),
                "one_gpu_note": ("rank 0 scanning its shard n_gpus times per query, measured in this run: what one GPU needs for the same corpus bytes"
                                 if one_gpu else None),
                "arch": capi.device_arch(local_rank),
            },
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic[0] if traffic else None, "traffic_source": traffic[1] if traffic else None,
                         "kernel": "knn_scan_fixed", "launches": launches, "avg_ms": avg_scan_s * 1e3,
                         "algorithmic_bytes_per_launch": algo_bytes},
        }

        def budget_left() -> float:
            return args.time_budget - (time.perf_counter() - t_start)

        def leg(name: str, need_s: float, fn):
            """Extra legs never take the bench line down: an exception or an exhausted time budget is recorded in place of the result."""
            if budget_left() < need_s:
                return {"skipped": f"time budget: {budget_left():.0f} s left, leg needs ~{need_s:.0f} s (--time-budget)"}
            t0 = time.perf_counter()
            try:
                out = fn()
            except Exception as e:
                out = {"error": repr(e)}
            if isinstance(out, dict):
                out.setdefault("leg_seconds", time.perf_counter() - t0)
            return out

        extra = world == 1 and not strong
        if extra:
            result["sharded_in_process"] = leg("sharded_in_process", 5, lambda: sharded_in_process_leg(args, ix, corpus, d_inv, queries, device, kk))
        if extra and args.batch > 1:
            result["batched"] = leg("batched", 5, lambda: batched_leg(args, ix, queries, device, kk))
            result["pruned_scan"] = leg("pruned_scan", 5, lambda: pruned_leg(args, ix, queries, device, kk))
            result["prefilter"] = leg("prefilter", 5, lambda: prefilter_leg(args, ix, queries, device, kk))
        if extra and not args.no_cpu:
            out = leg("cpu_baseline", 90, lambda: cpu_baseline_and_parity(args, corpus, queries, metric_id, ix))
            if isinstance(out, tuple):
                result["cpu_baseline"], result["parity"] = out
            else:
                result["cpu_baseline"] = {"value": None, **out}
        if extra and (args.hnsw_rows or args.hybrid_docs):
            d_inv = None
            # the other configs need the HBM the headline corpus holds only partly, but host RAM and time are shared: release first
            ix.close()
            ix = None
            corpus = None
            torch.cuda.empty_cache()
        if extra and args.hybrid_docs:
            import bench_hybrid
            result["hybrid"] = leg("hybrid", 20 + 14 * args.hybrid_docs / 1e6, lambda: bench_hybrid.run(dict(docs=args.hybrid_docs, device=local_rank)))
        if extra and args.ft_packed_words:
            import bench_ft_packed
            result["ft_packed"] = leg("ft_packed", 15, lambda: bench_ft_packed.run(dict(words=args.ft_packed_words)))
        if extra and args.hnsw_rows:   # last: at its true size (BASELINE configs[2], 10M x 768) the host build alone takes minutes
            import bench_hnsw
            from cpu_scaling import effective_cpus
            threads = 2 * effective_cpus()

            def need_s(rows):   # host build (measured: 20.6 k inserts/s from 16 threads, 25.8 k from 32, at 768 dims) + corpus, mirrors, reference engine
                return 40 + rows / (1290.0 * min(threads, 16) + 325.0 * max(0, min(threads, 32) - 16)) * 1.25 + 110 * rows / 1e7

            rows = args.hnsw_rows
            picked = None
            if budget_left() < need_s(rows) and args.hnsw_fallback_rows and args.hnsw_fallback_rows < rows:
                picked = (f"{rows} rows need ~{need_s(rows):.0f} s of the {budget_left():.0f} s left (--time-budget): "
                          f"configs[2] scaled to {args.hnsw_fallback_rows} rows instead")
                rows = args.hnsw_fallback_rows
            result["hnsw"] = leg("hnsw", need_s(rows), lambda: bench_hnsw.run(dict(rows=rows, queries=args.hnsw_queries, device=local_rank)))
            if picked and isinstance(result["hnsw"], dict):
                result["hnsw"]["size_fallback"] = picked
        result["bench_wall_seconds"] = time.perf_counter() - t_start
        emit(result, args)
    if ix is not None:
        ix.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
