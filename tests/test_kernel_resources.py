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
    gemms = {k: v for k, v in usage.items() if "knn_gemm_bf16_glds" in k or "knn_gemm_bf16_split" in k or "knn_gemm_bf16_qreg" in k}
    # (single ring + split rings) x 3 metrics x 2 modes x 2 query-tile widths + the register-staged-query filter kernel x 3 metrics x 2 widths
    assert len(gemms) == 30, sorted(gemms)
    for name, u in gemms.items():
        assert u["VGPRs Spill"] == 0 and u["ScratchSize"] == 0, (name, u)
        assert u["VGPRs"] <= 256, (name, u)


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="needs hipcc")
def test_qreg_gemm_staging_registers_are_left_alone(tmp_path):
    """knn_gemm_bf16_qreg keeps a query stage IN FLIGHT in registers across loop iterations behind the compiler's back (inline-asm
    global_load_dwordx4 ... s_waitcnt ... ds_write_b128).  That is only sound while the compiler never touches those registers between the
    load and the store: in every instantiation, walking the stage loop in program order, no instruction may name a register between the asm
    load that targets it and the asm store that reads it."""
    from reindexer_amd.build import HIP_FLAGS
    flags = [f for f in HIP_FLAGS if f not in ("-shared",)]
    r = subprocess.run([HIPCC, *flags, "-S", "--cuda-device-only", str(CSRC / "knn_batched_bf16.hip"), "-o", str(tmp_path / "k.s")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    text = (tmp_path / "k.s").read_text()
    kernels = re.findall(r"^(_ZN5rxgpu18knn_gemm_bf16_qreg\w+):.*?\n(.*?)^\s*\.end_amdhsa_kernel|^(_ZN5rxgpu18knn_gemm_bf16_qreg\w+):", text, re.S | re.M)
    bodies = {}
    for m in re.finditer(r"^(_ZN5rxgpu18knn_gemm_bf16_qreg\w+):\s*;.*?\n(.*?)s_endpgm", text, re.S | re.M):
        bodies[m.group(1)] = m.group(2)
    assert len(bodies) == 6, sorted(bodies)   # 3 metrics x 2 query-tile widths

    def regs_of(tok):   # "v[128:131]" / "v155" -> set of register numbers
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.fullmatch(r"v(\d+)", tok)
        return {int(m.group(1))} if m else set()

    for name, body in bodies.items():
        lines = [ln.strip() for ln in body.splitlines() if ln.strip() and not ln.strip().startswith(";")]
        # the query loaders' stage loop = the outermost loop (label ... last branch back to it) that holds the asm loads; the prologue's
        # loads lie in front of it and may use any register
        labels = {ln.split(":")[0]: i for i, ln in enumerate(lines) if re.match(r"\.LBB\d+_\d+:", ln)}
        loops = {}
        for i, ln in enumerate(lines):
            m = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops[m.group(1)] = max(loops.get(m.group(1), 0), i)
        regions = [(labels[lab], end) for lab, end in loops.items()
                   if any(x.startswith("global_load_dwordx4") for x in lines[labels[lab]:end]) and sum(x.startswith("s_barrier") for x in lines[labels[lab]:end]) >= 2
                   and any(("atomic" in x or x.startswith("ds_add")) for x in lines[labels[lab]:end])]
        assert regions, name
        # two stages per iteration: two barriers; the tile epilogue (its atomics) is part of the loop; an enclosing region would hold the prologue too
        lo, hi = min(regions, key=lambda r: r[1] - r[0])
        region = lines[lo:hi + 1]
        # walk the loop in program order, twice (what is in flight at its end is in flight at its start): a register is IN FLIGHT from the asm
        # load that targets it to the asm store that reads it; in between nothing else may name it
        in_flight = set()
        loads = stores = 0
        for lap in range(2):
            for ln in region:
                toks = re.findall(r"v\[\d+:\d+\]|v\d+\b", ln)
                op = ln.split()[0]
                if op == "global_load_dwordx4":
                    dst = regs_of(toks[0])
                    addr = regs_of(toks[1]) if len(toks) > 1 else set()
                    assert not ((dst | addr) & in_flight) or lap == 0, (name, ln)
                    in_flight |= dst
                    loads += lap
                elif op == "ds_write_b128":
                    data = regs_of(toks[1]) if len(toks) > 1 else set()
                    assert not (regs_of(toks[0]) & in_flight), (name, ln)
                    in_flight -= data
                    stores += lap
                elif toks:
                    used = set().union(*[regs_of(t) for t in toks])
                    assert lap == 0 or not (used & in_flight), (name, ln, sorted(used & in_flight))
        assert loads in (4, 8) and stores == loads, (name, loads, stores)   # two stages per iteration, 2 or 4 pieces each


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="needs hipcc")
def test_hybrid_fusion_kernels_fit(tmp_path):
    usage = resource_usage(CSRC / "hybrid_fuse.hip", tmp_path)
    prep = next(v for k, v in usage.items() if "hybrid_prepare_kernel" in k)
    join = next(v for k, v in usage.items() if "hybrid_join_kernel" in k)
    assert prep["VGPRs"] <= 128 and prep["VGPRs Spill"] == 0 and prep["ScratchSize"] == 0, prep
    assert join["VGPRs Spill"] == 0 and join["ScratchSize"] == 0, join


# ---------------------------------------------------------------------------------------------- HNSW search kernels, from the built object
LLVM_BIN = Path("/opt/rocm/lib/llvm/bin")


def built_kernel_metadata(obj: Path, tmp: Path) -> dict:
    """Per-kernel metadata of the gfx950 code object embedded in a built .o (the note hipcc writes: what the loader will allocate).
    A recompile with -Rpass-analysis of hnsw_search.hip takes 90 s (220 instantiations); this takes one."""
    fat, dev = tmp / "fat.bin", tmp / "dev.co"
    subprocess.run([str(LLVM_BIN / "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(obj)], check=True, capture_output=True)
    subprocess.run([str(LLVM_BIN / "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--input={fat}", f"--output={dev}"], check=True, capture_output=True)
    notes = subprocess.run([str(LLVM_BIN / "llvm-readelf"), "--notes", str(dev)], check=True, capture_output=True, text=True).stdout
    out, block = {}, {}
    for line in notes.splitlines():   # the kernel's fields come in alphabetical order, .name among them: close a block at ".args:" / "- .agpr_count"
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(\S+)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key in ("args", "agpr_count") and line.lstrip().startswith("- "):
            if "name" in block:
                out[block["name"]] = block
            block = {}
        if key in ("name", "symbol", "vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"):
            block[key] = val if key in ("name", "symbol") else int(val)
    if "name" in block:
        out[block["name"]] = block
    return {k: v for k, v in out.items() if k == v.get("symbol", "").removesuffix(".kd") or "symbol" not in v}


@pytest.mark.skipif(not (LLVM_BIN / "llvm-readelf").exists(), reason="needs the ROCm llvm tools")
def test_hnsw_search_kernels_stay_in_registers(tmp_path):
    """The sorted-list searches keep their lists in registers (per-slot state as vector values): no instantiation may use scratch memory
    or spill VGPRs, and the throughput form of the headline shape (cosine / ip / l2, D = 768, two entries a lane, no deleted nodes) must
    fit five wavefronts per SIMD (<= 96 VGPRs) — what its __launch_bounds__ asks for."""
    obj = ROOT / "reindexer_amd" / "build" / "obj" / "hnsw_search.o"
    if not obj.exists():
        from reindexer_amd import build
        build.build_device()
    meta = built_kernel_metadata(obj, tmp_path)
    search = {k: v for k, v in meta.items() if "hnsw_search_kernel" in k}
    assert len(search) >= 200, len(search)
    for name, v in search.items():
        assert v["vgpr_spill_count"] == 0 and v["private_segment_fixed_size"] == 0, (name, v)
    headline = [v for k, v in search.items() if re.search(r"ILi[012]ELb0ELi12ELb0ELb0ELi2ELb0EE", k)]
    assert len(headline) == 3 and all(v["vgpr_count"] <= 96 for v in headline), headline
