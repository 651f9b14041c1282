#!/bin/bash
# Round 6: calibrate the nomination GEMM against the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) — timing run, then a kernel trace of
# the same command (names and durations of the vendor kernels), then the shader clock under both (GRBM_GUI_ACTIVE / 8 XCDs / duration).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/bench_gemm_vendor.py --out gpurun_out/rd6_gemm_vendor.json > gpurun_out/rd6_gemm_vendor.log 2>&1; echo "timing rc=$?"
tail -30 gpurun_out/rd6_gemm_vendor.log | head -60
rm -rf gpurun_out/gv; mkdir -p gpurun_out/gv
cd /tmp
timeout 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/gv -o g -- python $R/tools/bench_gemm_vendor.py --iters 2 > /tmp/gv.log 2>&1; echo "clock rc=$?"
cd $R
python - <<'PY'
import csv, glob, collections, json
cyc = collections.defaultdict(list); dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/gv/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cyc[r["Kernel_Name"][:90]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/gv/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:90]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for k in dur:
    d = sorted(dur[k]); dm = d[len(d) // 2]
    if dm < 100_000: continue
    e = {"launches": len(d), "median_ms": dm / 1e6, "total_ms": sum(d) / 1e6}
    if cyc.get(k):
        c = sorted(cyc[k]); e["shader_clock_ghz"] = c[len(c) // 2] / 8 / dm
    out[k] = e
json.dump(out, open("gpurun_out/rd6_gemm_vendor_kernels.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["total_ms"])[:12]: print(k[:80], {a: round(b, 4) for a, b in v.items()})
PY
find gpurun_out/gv -name "*.csv" -size +1M -delete
