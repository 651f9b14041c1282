#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof/{trace,pmc_fetch,pmc_write}) into profiles/<tag>_*.{csv,json}.

    python tools/summarize_prof.py gpurun_out/prof r1

Kernel names are truncated (torch's template names run to kilobytes).  PMC correction per
/opt/skills/guides/MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE (KiB) counts 128-B requests as 64 B for wide
coalesced streaming reads, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is taken as reported (KiB).
"""
import csv
import json
import statistics
import sys
from pathlib import Path


def main():
    src, tag = Path(sys.argv[1]), sys.argv[2]
    out = Path(__file__).resolve().parents[1] / "profiles"
    out.mkdir(exist_ok=True)
    stats = list(csv.DictReader(open(src / "trace" / f"{tag}_kernel_stats.csv")))
    with open(out / f"{tag}_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in stats:
            w.writerow([r["Name"][:96], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    summary = {"command": "rocprofv3 --kernel-trace --stats -- python bench.py --steps 30 --warmup 5 --no-cpu ; "
                          "rocprofv3 --pmc FETCH_SIZE -- python bench.py --steps 10 --warmup 2 --no-cpu ; "
                          "rocprofv3 --pmc WRITE_SIZE -- (same)", "kernels": {}}
    for r in stats:
        if "rxgpu" in r["Name"]:
            summary["kernels"][r["Name"][:96]] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                                  "min_ms": float(r["MinNs"]) / 1e6, "max_ms": float(r["MaxNs"]) / 1e6}
    per_dispatch = {}   # kernel -> counter -> {dispatch id: KiB}; the passes run the same command, so dispatch ids pair up
    for name, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        p = src / name / f"{tag}_counter_collection.csv"
        if not p.exists():
            continue
        vals = {}
        for r in csv.DictReader(open(p)):
            if "rxgpu" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                vals.setdefault(r["Kernel_Name"][:96], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for k, dv in vals.items():
            v = [x for _, x in dv]
            e = summary["kernels"].setdefault(k, {})
            e[ctr + "_KiB_mean"] = statistics.mean(v)
            e[ctr + "_launches"] = len(v)
            per_dispatch.setdefault(k, {})[ctr] = dict(dv)
    for k, e in summary["kernels"].items():
        if "FETCH_SIZE_KiB_mean" in e:
            e["hbm_read_bytes_per_launch_corrected"] = 2 * e["FETCH_SIZE_KiB_mean"] * 1024
        if "WRITE_SIZE_KiB_mean" in e:
            e["hbm_write_bytes_per_launch"] = e["WRITE_SIZE_KiB_mean"] * 1024
        if "hbm_read_bytes_per_launch_corrected" in e:
            e["hbm_traffic_bytes_per_launch"] = e["hbm_read_bytes_per_launch_corrected"] + e.get("hbm_write_bytes_per_launch", 0.0)
    # A kernel launched over inputs of several sizes (bench.py's sharded leg scans half the corpus per launch): the mean over ALL launches
    # describes no launch.  The largest class = the launches whose FETCH_SIZE is within 25 % of the largest one, i.e. the whole-corpus scans.
    for k, ctrs in per_dispatch.items():
        f = ctrs.get("FETCH_SIZE")
        if not f or "knn_scan" not in k:
            continue
        top = max(f.values())
        ids = [d for d, x in f.items() if x >= 0.75 * top]
        if len(ids) == len(f):
            continue
        w = ctrs.get("WRITE_SIZE", {})
        wr = [w[d] for d in ids if d in w] or list(w.values()) or [0.0]
        e = summary["kernels"][k]
        e["largest_class_launches"] = len(ids)
        e["largest_class_hbm_read_bytes_per_launch_corrected"] = 2 * statistics.mean(f[d] for d in ids) * 1024
        e["largest_class_hbm_traffic_bytes_per_launch"] = e["largest_class_hbm_read_bytes_per_launch_corrected"] + statistics.mean(wr) * 1024
    # The headline kernel's whole-corpus launches in the kernel trace (by duration, the same 25 % rule) beside the counter bytes of that class.
    tr = src / "trace" / f"{tag}_kernel_trace.csv"
    if tr.exists():
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(tr)) if "knn_scan_fixed" in r["Kernel_Name"]]
        if dur:
            full = [d for d in dur if d >= 0.75 * max(dur)]
            scan = next((e for k, e in summary["kernels"].items() if "knn_scan_fixed" in k), {})
            summary["headline_kernel"] = {
                "kernel": "knn_scan_fixed", "launches_in_trace": len(dur), "full_corpus_launches": len(full),
                "avg_ms": statistics.mean(full), "median_ms": statistics.median(full), "min_ms": min(full), "max_ms": max(full),
                "hbm_traffic_bytes_per_full_corpus_launch": scan.get("largest_class_hbm_traffic_bytes_per_launch", scan.get("hbm_traffic_bytes_per_launch")),
                "hbm_traffic_bytes_per_launch_mean_over_all_launches": scan.get("hbm_traffic_bytes_per_launch"),
                "note": "whole-corpus launches = those lasting at least 0.75 x the longest (bench.py's in-process sharded leg launches the same kernel over "
                        "half the corpus); counter bytes = 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE of the launches whose FETCH_SIZE is within "
                        "25 % of the largest, from separate --pmc passes of the same command"}
    (out / f"{tag}_rocprof_summary.json").write_text(json.dumps(summary, indent=1) + "\n")
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
