"""-m gpu: the reference's engine-level scenario (gtests/tests/unit/hnsw_streaming_search_test.cc) compiled against OUR Map classes
(tests/cpp/gpu_map_streaming_test.cc, built by reindexer_amd.build) — the drop-in claim exercised from C++, no Python in the data path."""
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_reference_style_streaming_scenario_in_cpp(rxgpu):
    exe = ROOT / "tests" / "cpp" / "gpu_map_streaming_test"
    assert exe.exists(), "run python -m reindexer_amd.build"
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "all checks passed" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
