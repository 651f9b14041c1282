"""The driver keeps the last 8 KB of bench.py's stdout: the printed line must stay under that and still carry every
leg.  Checked on the CPU against full records of earlier rounds (profiles/*_bench_full.json): bench.emit() is what prints."""
import argparse
import glob
import io
import json
import os
import sys
from contextlib import redirect_stderr, redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

RECORDS = sorted(glob.glob(os.path.join(ROOT, "profiles", "*bench_full.json")))


@pytest.mark.parametrize("path", RECORDS[-4:] or [None])
def test_printed_line_is_small_and_complete(path, tmp_path):
    if path is None:
        pytest.skip("no full bench record under profiles/")
    import bench
    text = open(path).read().strip().splitlines()[-1]
    full = json.loads(text)
    out, err = io.StringIO(), io.StringIO()
    args = argparse.Namespace(full_json=str(tmp_path / "bench_full.json"))
    with redirect_stdout(out), redirect_stderr(err):
        bench.emit(full, args)
    lines = [ln for ln in out.getvalue().splitlines() if ln.strip()]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    assert len(lines[0]) < 8000, f"{len(lines[0])} bytes: the driver's tail would cut it"
    small = json.loads(lines[0])
    # the contract keys and the two objects of the tier
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in small, key
    assert small["value"] == pytest.approx(full["value"], rel=1e-4)
    for key in ("bound", "achieved", "peak", "unit", "frac"):
        assert key in small["roofline"], key
    assert small["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-4)
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in small["cpu_baseline"], key
    # every leg of the full record is still there with its roofline fraction
    for leg in ("batched", "pruned_scan", "prefilter", "hnsw", "hybrid", "ft_packed", "parity"):
        if leg in full:
            assert leg in small, leg

    def fracs(o):
        if isinstance(o, dict):
            for k, v in o.items():
                if k == "frac" and isinstance(v, (int, float)):
                    yield v
                else:
                    yield from fracs(v)
        elif isinstance(o, list):
            for v in o:
                yield from fracs(v)
    assert len(list(fracs(small))) >= min(4, len(list(fracs(full))))
    # the full record went to the side file and to stderr
    assert json.loads(open(args.full_json).read()) == full
    assert json.loads(err.getvalue().strip().splitlines()[-1]) == full
