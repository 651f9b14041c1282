"""-m gpu: bench.py's N > 1 code path on the one GPU there is — launched the way the driver launches it (python -m torch.distributed.run, one
rank) with RXGPU_BENCH_FORCE_DIST=1: RCCL process group, local scan -> all_gather_into_tensor of the per-shard lists -> knn_merge_shards, the
barrier + max-over-ranks timing, ONE JSON line from rank 0 (the contract the SCALE run reads).  Small corpus: this checks the path, not a number."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_distributed_path_on_one_rank(rxgpu, scaling, tmp_path):
    env = dict(os.environ, RXGPU_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           str(ROOT / "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--rows", "200000", "--total-rows", "200000", "--scaling", scaling,
           "--no-cpu", "--batch", "0", "--hnsw-rows", "0", "--hybrid-docs", "0", "--ft-packed-words", "0", "--full-json", str(tmp_path / "full.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=str(tmp_path))
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-3000:])
    out = json.loads(lines[0])
    assert out["metric"] == "knn_queries_per_sec" and out["n_gpus"] == 1 and out["steps"] == 6 and out["warmup"] == 2 and out["scaling"] == scaling
    assert out["config"]["rccl_ranks"] == 1 and out["config"]["sharding"].startswith("row-range shards")
    assert out["value"] > 0 and abs(out["value"] - 1e3 / out["ms_per_step"]) / out["value"] < 1e-3
    assert out["roofline"]["launches"] == 6 and out["roofline"]["frac"] > 0
