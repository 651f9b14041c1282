"""-m gpu: tools/multigpu_preflight.py — the tool a multi-GPU node runs first (peer matrix, pooled communicators, one 2-shard query per
engine against the single-device result).  On the 1-GPU test box the device is listed twice: same code path, one RCCL rank with two slots."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_preflight_says_ok_and_prints_one_json_line(rxgpu):
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "multigpu_preflight.py"), "--rows", "20000", "--dim", "64"], capture_output=True, text=True, timeout=280)
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])
    out = json.loads(lines[0])
    assert r.returncode == 0 and out["ok"], out
    for leg in ("brute_force", "hnsw", "bm25"):
        assert out[leg].get("identical") is True, (leg, out[leg])
    assert out["devices_used"] == [0, 0] or len(out["devices_used"]) == out["visible_devices"]
    assert out["brute_force"]["collectives"] >= 1
    assert out["bm25"]["phrase_identical"] and out["bm25"]["synonym_identical"] and out["bm25"]["phrase_documents"] > 0, out["bm25"]
