#!/usr/bin/env python3
"""Host diagnostics for the CPU baselines: what the box really gives a process (cgroup CPU quota, affinity, NUMA), and how the
reference's AVX-512 brute-force scan scales with T concurrent query threads over one shared index (the reference's concurrency
model).  bench.py picks the thread count of its all-core leg from the same probes.

    python tools/cpu_scaling.py [--rows 1000000] [--dim 768] > profiles/r2_cpu_scaling.json
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
os.environ.setdefault("RX_TARGET_INSTRUCTIONS", "avx512")


def host_limits() -> dict:
    out = {"os_cpu_count": os.cpu_count()}
    try:
        out["sched_affinity"] = len(os.sched_getaffinity(0))
    except Exception as e:
        out["sched_affinity"] = repr(e)
    for name, path in (("cgroup_v2_cpu_max", "/sys/fs/cgroup/cpu.max"), ("cgroup_v1_quota", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"),
                       ("cgroup_v1_period", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"), ("cpuset_effective", "/sys/fs/cgroup/cpuset.cpus.effective"),
                       ("numa_online", "/sys/devices/system/node/online"), ("loadavg", "/proc/loadavg")):
        try:
            out[name] = Path(path).read_text().strip()
        except Exception:
            out[name] = None
    return out


def effective_cpus() -> int:
    """Hardware threads this process may really use: min(affinity, cgroup quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, p = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    try:
        q = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
        p = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
        if q > 0 and p > 0:
            n = min(n, max(1, q // p))
    except Exception:
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--deadline", type=float, default=8.0)
    args = ap.parse_args()
    from oracle import pyoracle
    out = {"limits": host_limits(), "effective_cpus": effective_cpus()}
    ref = pyoracle.ref_or_none()
    if ref is None:
        out["error"] = "oracle/_ref missing"
        print(json.dumps(out))
        return
    rng = np.random.default_rng(1)
    rows = rng.standard_normal((args.rows, args.dim), dtype=np.float32) * np.float32(0.25)
    q = rng.standard_normal((64, args.dim), dtype=np.float32) * np.float32(0.25)
    bf = pyoracle.RefBruteforce(ref, 1, args.dim, args.rows)
    bf.add(rows, np.arange(args.rows, dtype=np.uint64))
    del rows
    gb = args.rows * args.dim * 4 / 1e9
    table = []
    t = 1
    while t <= (os.cpu_count() or 1):
        secs, done = bf.search_knn_mt(q, 10, t, 4, deadline_s=args.deadline)
        table.append({"threads": t, "queries": done, "seconds": secs, "qps": done / secs, "gbps": done / secs * gb})
        t *= 2
    out["bruteforce_scan_scaling"] = {"rows": args.rows, "dim": args.dim, "gb_per_query": gb, "table": table}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
