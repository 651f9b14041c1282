#!/usr/bin/env python3
"""rocprofv3 counter files of tools/gpu_session_r4_gemm_pmc.sh -> one JSON: per nomination kernel (single ring / split rings, ip, filter
pass, 256-query tile) the average per-launch value of every counter collected, the kernel's average duration from the kernel trace of the
same runs, and the derived fractions (matrix-core busy share of the SIMD cycles, wait share of the wave cycles, HBM bytes by the gfx950
FETCH_SIZE rule: KB units, x2 for 16-byte-per-lane loads as in MI355X_MICROARCH.md).
    python tools/summarize_gemm_pmc.py <dir with pass subdirs> <out.json> ["<source line>"]"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    root, out = sys.argv[1], sys.argv[2]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    dur = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "knn_gemm_bf16" not in k:
                continue
            a = acc[k][row["Counter_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "knn_gemm_bf16" not in k:
                continue
            d = dur[k]
            d[0] += 1
            d[1] += (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e6
    res = {}
    for k, counters in acc.items():
        e = {c: v[1] / max(v[0], 1) for c, v in counters.items()}
        e["launches_per_counter"] = {c: v[0] for c, v in counters.items()}
        if k in dur:
            e["avg_ms_under_profiler"] = dur[k][1] / max(dur[k][0], 1)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e and e["GRBM_GUI_ACTIVE"]:
            e["mfma_busy_frac_of_simd_cycles"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (e["GRBM_GUI_ACTIVE"] / 8.0)
        if "SQ_WAIT_INST_ANY" in e and "SQ_WAVE_CYCLES" in e and e["SQ_WAVE_CYCLES"]:
            e["wait_inst_any_frac_of_wave_cycles"] = e["SQ_WAIT_INST_ANY"] / e["SQ_WAVE_CYCLES"]
        if "FETCH_SIZE" in e:
            e["hbm_read_GB_corrected"] = e["FETCH_SIZE"] * 1024.0 * 2.0 / 1e9
        res[k] = e
    source = sys.argv[3] if len(sys.argv) > 3 else None
    json.dump({"source": source or "rocprofv3 --kernel-trace --pmc <one set per pass>, tools/bench_gemm_ab.py --metrics ip --rounds 1 --iters 2 --modes split_ring_blocked_shadow,single_ring_blocked_shadow (10M x 768, 256 queries; tile-blocked bf16 shadow)",
               "kernels": res}, open(out, "w"), indent=1)
    for k, e in res.items():
        print(k[:70], {x: (round(y, 4) if isinstance(y, float) else y) for x, y in e.items() if x != "launches_per_counter"})


if __name__ == "__main__":
    main()
