#!/bin/bash
# round 5: the shader clock the chip actually runs the nomination GEMM at, against the clock of the HBM-bound scan kernels — GRBM_GUI_ACTIVE
# (busy cycles, summed over the 8 XCDs) over the kernel's duration from the kernel trace of the same pass.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; rm -rf gpurun_out/clk; mkdir -p gpurun_out/clk; export TMPDIR=/tmp
cd /tmp
timeout 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/clk/gemm -o g -- python $R/tools/bench_gemm_ab.py --metrics ip --rounds 1 --iters 3 --modes split_ring_blocked_shadow,areg_blocked_shadow > /tmp/c1.log 2>&1; echo "gemm rc=$?"
timeout 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/clk/gemm128 -o g -- python $R/tools/bench_gemm_ab.py --metrics ip --batch 128 --rounds 1 --iters 3 --modes split_ring_blocked_shadow > /tmp/c2.log 2>&1; echo "gemm128 rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/clk/scan -o g -- python $R/bench.py --steps 10 --warmup 2 --no-cpu --hnsw-rows 0 --hybrid-docs 0 --ft-packed-words 0 --full-json /tmp/x.json > /tmp/c3.log 2>&1; echo "scan rc=$?"
cd $R
python - <<'PY'
import csv, glob, collections, json
out = {}
for tag in ("gemm", "gemm128", "scan"):
    cyc = collections.defaultdict(list); dur = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/clk/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cyc[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
    for f in glob.glob(f"gpurun_out/clk/{tag}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"][:70]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in cyc:
        if "rxgpu" not in k or not dur.get(k): continue
        c = sorted(cyc[k]); d = sorted(dur[k])
        cm, dm = c[len(c) // 2], d[len(d) // 2]   # medians (a kernel name may cover launches of several sizes: the median pair is the common one)
        if dm < 200_000: continue
        out[f"{tag}: {k}"] = {"launches": len(c), "median_cycles_per_xcd": cm / 8, "median_ms": dm / 1e6, "shader_clock_ghz": cm / 8 / dm}
json.dump(out, open("gpurun_out/rd5_clock.json", "w"), indent=1)
for k, v in out.items(): print(k[:90], {a: round(b, 4) for a, b in v.items()})
PY
find gpurun_out/clk -name "*.csv" -size +1M -delete
