"""CPU: build-time resource checks of the kernels whose correctness or speed depends on staying in registers.

hipcc cross-compiles gfx950 without a GPU; `-Rpass-analysis=kernel-resource-usage` prints what the code object header will say.
  * every knn_gemm_bf16_glds instantiation: 0 spilled VGPRs, no scratch — a spill inside the K loop makes the compiler drain the LDS-DMA
    queue every stage (the L2 filter form at 256 queries did exactly that, 61 VGPRs, before its epilogue was fenced block by block);
  * hybrid_prepare_kernel: at most 128 VGPRs (it has to fit beside the scan's two workgroups on a CU) and no spill; hybrid_join_kernel: no spill."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "reindexer_amd" / "csrc"
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def resource_usage(src: Path, tmp: Path) -> dict:
    from reindexer_amd.build import HIP_FLAGS
    flags = [f for f in HIP_FLAGS if f not in ("-shared", "-fPIC")]
    r = subprocess.run([HIPCC, *flags, "-c", str(src), "-o", str(tmp / (src.stem + ".o")), "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+Function Name:\s+(\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?:\s+(\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="needs hipcc")
def test_bf16_gemm_instantiations_do_not_spill(tmp_path):
    usage = resource_usage(CSRC / "knn_batched_bf16.hip", tmp_path)
    gemms = {k: v for k, v in usage.items() if "knn_gemm_bf16_glds" in k}
    assert len(gemms) == 12, sorted(gemms)   # 3 metrics x 2 modes x 2 query-tile widths
    for name, u in gemms.items():
        assert u["VGPRs Spill"] == 0 and u["ScratchSize"] == 0, (name, u)
        assert u["VGPRs"] <= 256, (name, u)


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="needs hipcc")
def test_hybrid_fusion_kernels_fit(tmp_path):
    usage = resource_usage(CSRC / "hybrid_fuse.hip", tmp_path)
    prep = next(v for k, v in usage.items() if "hybrid_prepare_kernel" in k)
    join = next(v for k, v in usage.items() if "hybrid_join_kernel" in k)
    assert prep["VGPRs"] <= 128 and prep["VGPRs Spill"] == 0 and prep["ScratchSize"] == 0, prep
    assert join["VGPRs Spill"] == 0 and join["ScratchSize"] == 0, join
