#!/usr/bin/env python3
"""Condense rocprofv3 CSV output of any command into profiles/<tag>_rocprof.json.

    python tools/summarize_prof2.py <dir> <tag> <kernel-substring> ["<command description>"]

<dir> holds trace/ (rocprofv3 --kernel-trace --stats --output-format csv) and optionally pmc_fetch/ and pmc_write/ (separate
--pmc FETCH_SIZE / --pmc WRITE_SIZE passes of the same command).  FETCH_SIZE is reported raw (KiB) and, for 16-byte-per-lane coalesced
loads, corrected x2 as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes for gfx950; WRITE_SIZE as reported.
"""
import csv
import json
import sys
from pathlib import Path


def find(d: Path, suffix: str):
    hits = sorted(d.rglob("*" + suffix)) if d.exists() else []
    return hits[0] if hits else None


def main():
    src, tag, sub = Path(sys.argv[1]), sys.argv[2], sys.argv[3]
    out = {"command": sys.argv[4] if len(sys.argv) > 4 else "", "kernels": {}}
    stats = find(src / "trace", "kernel_stats.csv")
    if stats:
        for r in csv.DictReader(open(stats)):
            if sub in r["Name"]:
                out["kernels"][r["Name"][:120]] = {"calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6,
                                                   "avg_ms": float(r["AverageNs"]) / 1e6, "min_ms": float(r["MinNs"]) / 1e6,
                                                   "max_ms": float(r["MaxNs"]) / 1e6, "pct_of_gpu_time": float(r["Percentage"])}
    for name, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        f = find(src / name, "counter_collection.csv")
        if not f:
            continue
        vals = {}
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                vals.setdefault(r["Kernel_Name"][:120], []).append(float(r["Counter_Value"]))
        for k, v in vals.items():
            e = out["kernels"].setdefault(k, {})
            e[ctr + "_KiB_per_launch"] = v
            e[ctr + "_KiB_total"] = sum(v)
    for e in out["kernels"].values():
        if "FETCH_SIZE_KiB_total" in e:
            e["hbm_read_bytes_total_raw"] = e["FETCH_SIZE_KiB_total"] * 1024
            e["hbm_read_bytes_total_corrected_x2"] = 2 * e["FETCH_SIZE_KiB_total"] * 1024
        if "WRITE_SIZE_KiB_total" in e:
            e["hbm_write_bytes_total"] = e["WRITE_SIZE_KiB_total"] * 1024
    dst = Path(__file__).resolve().parents[1] / "gpurun_out" / f"{tag}_rocprof.json"
    dst.parent.mkdir(exist_ok=True)
    dst.write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out, indent=1)[:3000])


if __name__ == "__main__":
    main()
