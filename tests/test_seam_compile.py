"""CPU: the drop-in boundary, proved by compiling it.

The reference adapts its whole Index / FloatVectorIndex plugin surface to an implicit `Map` concept in
cpp_src/core/index/float_vector/hnsw_index.cc (HnswIndexBase<Map>: ctor :47-70, upsert / del :88-123, search / select / selectRaw
:159-288, streaming :318-361, cache / quantisation hooks :379-560, the factories :562-581).  This test
  1. applies integration/patches/*.patch to a COPY of the reference's files in a scratch directory (nothing under /root/reference is
     touched, nothing of it enters the repository),
  2. compiles the product's Maps (gpu_bruteforce_map.cc, gpu_hnsw_map.cc, hnsw_graph.cc, distance_cpu.cc) with RXGPU_IN_TREE, i.e. against
     the reference's OWN FloatVectorId / ConstFloatVectorView / hnswlib::SearchResultQueue / StreamingSearchOptions,
  3. compiles the patched hnsw_index.cc, which instantiates HnswIndexBase<GpuBruteforceMapInTree>, <GpuHnswMapT<None>> and
     <GpuHnswMapT<OnInsertions>> (every virtual of FloatVectorIndex), and checks that the object file defines them.
-c only: linking the full core needs the reference's CMake build (LevelDB / Snappy are fetched from the network)."""
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference/cpp_src")
PATCHES = ROOT / "integration" / "patches"

FLAGS = ["-std=c++20", "-O0", "-fPIC", "-w", "-DWITH_RXGPU=1", "-DRXGPU_IN_TREE=1", f"-I{REF}", f"-I{REF}/vendor", f"-I{REF}/vendor_subdirs",
         f"-I{REF}/core/index/float_vector", f"-I{ROOT}/include", f"-I{ROOT}/reindexer_amd/host", "-DRX_WITH_BUILTIN_ANN_INDEXES=1",
         "-DREINDEXER_WITH_SSE=1", "-DFMT_HEADER_ONLY=1", "-DSPDLOG_FMT_EXTERNAL=1", "-DREINDEX_CORE_BUILD=1", "-msse4.2", "-mpopcnt"]


def _patched_tree(tmp: Path) -> Path:
    """a/cpp_src/... layout holding copies of exactly the files the patches name, patched with `patch -p1`."""
    for patch in sorted(PATCHES.glob("*.patch")):
        for line in patch.read_text().splitlines():
            if line.startswith("+++ b/"):
                rel = line[len("+++ b/"):].split("\t")[0]
                dst = tmp / rel
                if not dst.exists():
                    dst.parent.mkdir(parents=True, exist_ok=True)
                    shutil.copy(Path("/root/reference") / rel, dst)
        r = subprocess.run(["patch", "-p1", "--no-backup-if-mismatch", "-i", str(patch)], cwd=tmp, capture_output=True, text=True)
        assert r.returncode == 0, f"{patch.name} does not apply:\n{r.stdout}\n{r.stderr}"
    return tmp


@pytest.mark.skipif(not REF.exists(), reason="needs the reference tree (/root/reference)")
def test_gpu_maps_instantiate_hnsw_index_base_against_reference_headers(tmp_path):
    assert shutil.which("patch"), "patch(1) is required"
    tree = _patched_tree(tmp_path)
    patched = tree / "cpp_src/core/index/float_vector/hnsw_index.cc"
    assert "rx_seam.h" in patched.read_text()
    jobs = {"hnsw_index": patched}
    for name in ("gpu_bruteforce_map", "gpu_hnsw_map", "hnsw_graph", "distance_cpu"):
        jobs[name] = ROOT / "reindexer_amd" / "host" / f"{name}.cc"

    def compile_one(item):
        name, src = item
        obj = tmp_path / f"{name}.o"
        r = subprocess.run(["g++", *FLAGS, "-c", str(src), "-o", str(obj)], capture_output=True, text=True)
        return name, obj, r

    with ThreadPoolExecutor(max_workers=5) as ex:
        results = list(ex.map(compile_one, jobs.items()))
    for name, obj, r in results:
        assert r.returncode == 0, f"{name} does not compile inside cpp_src:\n{r.stderr[-4000:]}"
    syms = subprocess.run(["nm", "-C", "--defined-only", str(tmp_path / "hnsw_index.o")], capture_output=True, text=True, check=True).stdout.replace("> >", ">>")
    for map_type in ("rxgpu::host::GpuBruteforceMapInTree", "rxgpu::host::GpuHnswMapT<(hnswlib::Synchronization)0>",
                     "rxgpu::host::GpuHnswMapT<(hnswlib::Synchronization)1>"):
        for member in ("select(", "selectRaw(", "upsert(", "del(", "beginStreaming(", "continueStreaming(", "Clone(", "GetMemStat(", "GrowFor("):
            assert f"reindexer::HnswIndexBase<{map_type}>::{member}" in syms, (map_type, member)
    # what the Maps themselves export with the reference's types in their signatures
    msyms = subprocess.run(["nm", "-C", "--defined-only", str(tmp_path / "gpu_bruteforce_map.o")], capture_output=True, text=True, check=True).stdout
    assert "rxgpu::host::GpuBruteforceMap::AddPointNoLock(reindexer::ConstFloatVectorView, reindexer::FloatVectorId)" in msyms
    hsyms = subprocess.run(["nm", "-C", "--defined-only", str(tmp_path / "gpu_hnsw_map.o")], capture_output=True, text=True, check=True).stdout
    assert "rxgpu::host::GpuHnswMap::BeginStreamingSearch(float const*, std::optional<float>, hnswlib::StreamingSearchOptions) const" in hsyms


@pytest.mark.skipif(not REF.exists(), reason="needs the reference tree (/root/reference)")
def test_patches_apply_cleanly(tmp_path):
    _patched_tree(tmp_path)
    cm = (tmp_path / "cpp_src/CMakeLists.txt").read_text()
    assert "WITH_RXGPU" in cm and "RXGPU_IN_TREE" in cm
