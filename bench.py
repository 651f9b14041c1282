#!/usr/bin/env python3
"""Headline benchmark: brute-force KNN queries/s, 10M x 768 fp32 inner product, k = 10 (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: ONE query (batch = 1) scanned against the
whole HBM-resident corpus through the C-ABI (rxgpu_search_knn_device), results left in HBM.  With N > 1 every rank
holds its own 10M-row shard (weak scaling = BASELINE configs[3]: 80M rows on 8 GPUs); each step then also does the
per-shard top-k all-gather over RCCL and the k-way merge.  Rank 0 prints ONE JSON line.

Extra legs (rank 0, N = 1 only, outside the timed region):
  roofline      HIP events recorded by the library around every scan-kernel launch on the launch stream
  cpu_baseline  the reference's own BruteforceSearch (oracle/_ref, AVX-512 path) — or the plain-C port when the
                reference build is absent — timed on the host cores on a bounded row-prefix sample
  parity        GPU result vs that CPU result on the same sample (ids must be identical, distance bits too)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

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
        for name, e in summ.get("kernels", {}).items():
            t = e.get("hbm_traffic_bytes_per_launch")
            if "knn_scan" in name and t and abs(t / algo_bytes - 1.0) < 0.25:
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
    ap.add_argument("--cpu-sample-rows", type=int, default=1_000_000)
    ap.add_argument("--cpu-queries", type=int, default=16)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline / parity leg")
    ap.add_argument("--batch", type=int, default=256, help="extra leg (N=1, untimed region): batched queries on the MFMA path; 0 = skip")
    ap.add_argument("--batch-iters", type=int, default=3)
    return ap.parse_args()


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
    info = {"hardware_threads": os.cpu_count(), "RX_TARGET_INSTRUCTIONS": os.environ.get("RX_TARGET_INSTRUCTIONS", "avx512")}
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    info["model"] = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return info


def cpu_baseline_and_parity(args, corpus: torch.Tensor, queries: torch.Tensor, metric_id: int):
    from oracle import pyoracle  # checker / baseline only
    s_rows = min(args.cpu_sample_rows, corpus.shape[0])
    nq = min(args.cpu_queries, queries.shape[0])
    host_rows = corpus[:s_rows].cpu().numpy()
    host_q = queries[:nq].cpu().numpy()
    labels = np.arange(s_rows, dtype=np.uint64) << np.uint64(32)
    orc = pyoracle.Oracle()
    ref = pyoracle.ref_or_none()
    use_ref = ref is not None and ref.simd_level == 3
    inv = orc.l2_modules(host_rows) if metric_id == 2 else None
    if metric_id == 2:
        host_q = np.stack([orc.normalize_copy(q)[0] for q in host_q])
    ncores = os.cpu_count() or 1

    if use_ref:
        bf = pyoracle.RefBruteforce(ref, metric_id, args.dim, s_rows)
        bf.add(host_rows, labels)
        search = lambda q: bf.search_knn(q, args.k)  # noqa: E731
        kind = "reference"
    else:
        search = lambda q: orc.bf_search_knn(metric_id, host_rows, labels, inv, q, args.k)  # noqa: E731
        kind = "port"

    t0 = time.perf_counter()
    cpu_res = [search(host_q[i]) for i in range(nq)]
    t1 = time.perf_counter()
    qps_1 = nq / (t1 - t0)

    # all host cores: T concurrent query threads over one shared index (the reference's own concurrency model)
    threads = min(ncores, 64)
    per_thread = 2
    def worker(t):
        for j in range(per_thread):
            search(host_q[(t + j) % nq])
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    t2 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    t3 = time.perf_counter()
    qps_all = threads * per_thread / (t3 - t2)

    scale = s_rows / corpus.shape[0]  # linear scan: time is proportional to rows
    baseline = {
        "value": qps_1 * scale, "unit": "queries/s", "cores": 1, "kind": kind,
        "sample": f"{s_rows}-row prefix of the same corpus, {nq} queries, k={args.k}; measured {qps_1:.3f} q/s on the sample, "
                  f"scaled by {s_rows}/{corpus.shape[0]} rows (linear scan); SIMD=avx512" ,
        "all_cores": {"value": qps_all * scale, "cores": threads, "measured_on_sample": qps_all},
        "gbps_per_core": qps_1 * s_rows * args.dim * 4 / 1e9,
        "host": host_cpu_info(),
    }

    # parity on the same sample: GPU through the C-ABI vs the CPU result
    with capi.VectorIndex(metric_id, args.dim) as ix:
        d_inv = None
        if metric_id == 2:
            d_inv = torch.from_numpy(inv).to(corpus.device)
        ix.adopt_device_rows(corpus.data_ptr(), s_rows, corpus.shape[1], d_inv.data_ptr() if d_inv is not None else None,
                             keepalive=(corpus, d_inv))
        dist, row, cnt = ix.search_knn(host_q, args.k + 1)
    ids_equal, max_ulps = 0, 0
    for i in range(nq):
        wd, wl = cpu_res[i]
        gl = labels[row[i, :args.k]]
        ids_equal += int(np.array_equal(gl, wl))
        ulps = np.abs(dist[i, :args.k].view(np.int32).astype(np.int64) - wd.view(np.int32).astype(np.int64))
        max_ulps = max(max_ulps, int(ulps.max()))
    parity = {"ids_equal_frac": ids_equal / nq, "max_ulps_dist": max_ulps, "queries": nq, "rows": s_rows,
              "recall_at_k": ids_equal / nq if ids_equal == nq else None}
    if parity["recall_at_k"] is None:
        rec = 0.0
        for i in range(nq):
            rec += len(set(labels[row[i, :args.k]].tolist()) & set(cpu_res[i][1].tolist())) / args.k
        parity["recall_at_k"] = rec / nq
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
            "kernel": "knn_gemm_bf16_glds<FILTER>" if bf16 else "knn_gemm<FILTER>", "avg_ms": gemm_ms, "launches": n_gemm,
            "dtype": "bf16 nomination (v_mfma_f32_32x32x16_bf16) + exact f32 re-score" if bf16 else "f32 (v_mfma_f32_32x32x2_f32)"}
    if bf16 and n_gemm:
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


def main():
    args = parse_args()
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
    corpus = make_corpus(args.rows, args.dim, 20260924 + rank, device)
    total_q = args.steps + args.warmup
    gq = torch.Generator(device=device)
    gq.manual_seed(7)  # the same queries on every rank
    queries = torch.empty((max(total_q, args.cpu_queries, args.batch), args.dim), dtype=torch.float32, device=device).normal_(0.0, 0.25, generator=gq)
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
    value = qps_global * world                   # aggregate in 10M-row-shard scans/s (== queries/s at N = 1)
    algo_bytes = args.rows * args.dim * 4        # N*D*4 per query (labels/norms excluded)
    avg_scan_s = (scan_ms / 1e3) / max(launches, 1)
    achieved = algo_bytes / avg_scan_s / 1e9 if launches else 0.0

    if rank == 0:
        traffic = pmc_traffic(algo_bytes)
        result = {
            "metric": "knn_queries_per_sec", "value": value, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"brute-force KNN, {args.rows} x {args.dim} fp32 per GPU, metric={args.metric}, k={args.k}, batch=1 "
                            f"(BASELINE configs[1]{'; x' + str(world) + ' row-sharded = configs[3]' if world > 1 else ''})",
                "rows_per_gpu": args.rows, "total_rows": args.rows * world, "dim": args.dim, "k": args.k, "batch": 1,
                "sharding": "row-range shards, RCCL all-gather of per-shard top-k + merge" if world > 1 else "none",
                "value_definition": "queries/s over the full corpus x n_gpus (each query scans one rows_per_gpu shard per GPU)",
                "qps_over_full_corpus": qps_global, "arch": capi.device_arch(local_rank),
            },
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic[0] if traffic else None, "traffic_source": traffic[1] if traffic else None,
                         "kernel": "knn_scan_fixed", "launches": launches, "avg_ms": avg_scan_s * 1e3,
                         "algorithmic_bytes_per_launch": algo_bytes},
        }
        if world == 1 and args.batch > 1:
            try:
                result["batched"] = batched_leg(args, ix, queries, device, kk)
            except Exception as e:
                result["batched"] = {"error": repr(e)}
            try:
                result["pruned_scan"] = pruned_leg(args, ix, queries, device, kk)
            except Exception as e:
                result["pruned_scan"] = {"error": repr(e)}
            try:
                result["prefilter"] = prefilter_leg(args, ix, queries, device, kk)
            except Exception as e:
                result["prefilter"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu:
            try:
                base, parity = cpu_baseline_and_parity(args, corpus, queries, metric_id)
                result["cpu_baseline"] = base
                result["parity"] = parity
            except Exception as e:  # the bench line must still be printed
                result["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(result), flush=True)
    ix.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
