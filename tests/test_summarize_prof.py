"""CPU: tools/summarize_prof.py on synthetic rocprofv3 output — a kernel launched over inputs of two sizes in the profiled command (bench.py's
in-process sharded leg scans half the corpus per launch with the headline kernel): the counter bytes reported for the headline launches must
be those of the whole-corpus class, not the mean over both (the round-6 bench lines up to rd6final carried that mean: 23.1 GB for a 30.7 GB scan),
and bench.py's pmc_traffic must pick the class figure."""
import csv
import importlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
SCAN = "void rxgpu::knn_scan_fixed<1, 12, true, true, true, rxgpu::WaveTopK>(rxgpu::ScanParams)"
MERGE = "void rxgpu::knn_merge_lists<rxgpu::WaveTopK>(float const*, unsigned int const*)"


def _write(path, header, rows):
    path.parent.mkdir(parents=True, exist_ok=True)
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(header)
        w.writerows(rows)


def test_launch_classes_are_kept_apart(tmp_path, monkeypatch):
    tag, src = "t1", tmp_path / "prof"
    full_kib, half_kib = 15_000_000.0, 7_500_000.0   # FETCH_SIZE counts 64 B per 128-B request on gfx950: x2 -> 30.72 / 15.36 GB
    launches = [("full", 4.4e6)] * 5 + [("half", 2.3e6)] * 4
    _write(src / "trace" / f"{tag}_kernel_stats.csv", ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"],
           [[SCAN, 9, 31.2e6, 3.47e6, 99.0, 2.3e6, 4.4e6, 1.0], [MERGE, 9, 1.5e5, 1.6e4, 1.0, 1.5e4, 2.0e4, 1.0]])
    trace, t = [], 1000
    for i, (_, ns) in enumerate(launches):
        trace.append(["KERNEL_DISPATCH", "Agent 2", 1, 0, 1, i + 1, 49, SCAN, i + 1, t, t + int(ns)])
        t += int(ns) + 1000
    _write(src / "trace" / f"{tag}_kernel_trace.csv",
           ["Kind", "Agent_Id", "Queue_Id", "Stream_Id", "Thread_Id", "Dispatch_Id", "Kernel_Id", "Kernel_Name", "Correlation_Id", "Start_Timestamp", "End_Timestamp"], trace)
    hdr = ["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Grid_Size", "Kernel_Id", "Kernel_Name", "Workgroup_Size",
           "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
    for d, ctr, val in (("pmc_fetch", "FETCH_SIZE", {"full": full_kib, "half": half_kib}), ("pmc_write", "WRITE_SIZE", {"full": 72.0, "half": 72.0})):
        rows = [[i + 1, i + 1, "Agent 2", 1, 1, 1, 131072, 49, SCAN, 256, 5120, 0, 64, 0, 48, ctr, val[c], 0, 1] for i, (c, _) in enumerate(launches)]
        rows.append([99, 99, "Agent 2", 1, 1, 1, 512, 32, MERGE, 512, 0, 0, 40, 0, 112, ctr, 26.0, 0, 1])
        _write(src / d / f"{tag}_counter_collection.csv", hdr, rows)
    # the tool writes into <repo>/profiles: point it at a scratch tree instead
    tool_dir = tmp_path / "repo" / "tools"
    tool_dir.mkdir(parents=True)
    (tool_dir / "summarize_prof.py").write_text((ROOT / "tools" / "summarize_prof.py").read_text())
    monkeypatch.syspath_prepend(str(tool_dir))
    monkeypatch.setattr(sys, "argv", ["summarize_prof.py", str(src), tag])
    sys.modules.pop("summarize_prof", None)
    importlib.import_module("summarize_prof").main()
    out = json.loads((tmp_path / "repo" / "profiles" / f"{tag}_rocprof_summary.json").read_text())
    e = out["kernels"][SCAN[:96]]
    mean_all = 2 * 1024 * (5 * full_kib + 4 * half_kib) / 9 + 72 * 1024
    assert abs(e["hbm_traffic_bytes_per_launch"] - mean_all) < 1.0                       # the mean over all launches describes no launch ...
    assert e["largest_class_launches"] == 5
    assert abs(e["largest_class_hbm_traffic_bytes_per_launch"] - (2 * 1024 * full_kib + 72 * 1024)) < 1.0   # ... the class figure does
    h = out["headline_kernel"]
    assert h["full_corpus_launches"] == 5 and h["launches_in_trace"] == 9 and abs(h["avg_ms"] - 4.4) < 1e-6
    assert h["hbm_traffic_bytes_per_full_corpus_launch"] == e["largest_class_hbm_traffic_bytes_per_launch"]
    assert "largest_class_launches" not in out["kernels"][MERGE[:96]]                    # one class only: nothing to separate


def test_bench_reports_the_whole_corpus_class():
    """The committed round-6 summaries: bench.py's roofline.traffic for the 10M x 768 scan = 30.72 GB x 1.0001, never the 23.1 GB mean."""
    sys.path.insert(0, str(ROOT))
    import bench
    got = bench.pmc_traffic(10_000_000 * 768 * 4)
    assert got is not None and abs(got[0] / 30.72e9 - 1.0) < 1e-3, got
