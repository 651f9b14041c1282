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
  4. the same for the ft_fast half (patch 0003: Selector<IdCont>::mergeResults, IndexText::commitFulltextImpl, rx_ft_seam.h).
-c only: linking the full core needs the reference's CMake build (LevelDB / Snappy are fetched from the network); the FT adapter is
additionally EXECUTED against the reference's merger in tests/test_gpu_ft_seam.py (oracle/_ref/libref_ft_seam.so)."""
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
    """a/cpp_src/... layout: a mirror of every directory that holds a patched file (symlinks to the reference's files — so that headers which
    include a patched header by its bare name, from the same directory, find the patched one), the patched files themselves materialised as
    copies and patched with `patch -p1`."""
    targets = []
    for patch in sorted(PATCHES.glob("*.patch")):
        for line in patch.read_text().splitlines():
            if line.startswith("+++ b/"):
                targets.append(line[len("+++ b/"):].split("\t")[0])
    for rel in targets:
        src_dir, dst_dir = (Path("/root/reference") / rel).parent, (tmp / rel).parent
        if src_dir.name == "cpp_src":   # CMakeLists.txt: the file alone
            dst_dir.mkdir(parents=True, exist_ok=True)
            continue
        # mirror the directory tree below the patched file's directory (ft/ holds config/, stopwords/, ... that headers name relatively)
        for f in src_dir.rglob("*"):
            d = dst_dir / f.relative_to(src_dir)
            if f.is_dir():
                d.mkdir(parents=True, exist_ok=True)
            elif not d.exists():
                d.parent.mkdir(parents=True, exist_ok=True)
                d.symlink_to(f)
    for rel in targets:
        dst = tmp / rel
        if dst.is_symlink() or not dst.exists():
            if dst.is_symlink():
                dst.unlink()
            shutil.copy(Path("/root/reference") / rel, dst)
    for patch in sorted(PATCHES.glob("*.patch")):
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
    # the brute-force Map the reference's factory constructs takes a DEVICE LIST (RX_GPU_VECTOR_INDEXES, device_list.h): the only constructor
    # the patched translation unit references is the std::vector<int> one -> rxgpu_index_create_sharded on a multi-GPU node (§8e / configs[3])
    usyms = subprocess.run(["nm", "-C", "-u", str(tmp_path / "hnsw_index.o")], capture_output=True, text=True, check=True).stdout
    assert "rxgpu::host::GpuBruteforceMap::GpuBruteforceMap(reindexer::VectorMetric, unsigned long, unsigned long, std::vector<int" in usyms
    assert "rxgpu::host::GpuBruteforceMap::GpuBruteforceMap(reindexer::VectorMetric, unsigned long, unsigned long, int)" not in usyms
    # what the Maps themselves export with the reference's types in their signatures
    msyms = subprocess.run(["nm", "-C", "--defined-only", str(tmp_path / "gpu_bruteforce_map.o")], capture_output=True, text=True, check=True).stdout
    assert "rxgpu::host::GpuBruteforceMap::AddPointNoLock(reindexer::ConstFloatVectorView, reindexer::FloatVectorId)" in msyms
    hsyms = subprocess.run(["nm", "-C", "--defined-only", str(tmp_path / "gpu_hnsw_map.o")], capture_output=True, text=True, check=True).stdout
    assert "rxgpu::host::GpuHnswMap::BeginStreamingSearch(float const*, std::optional<float>, hnswlib::StreamingSearchOptions) const" in hsyms


@pytest.mark.skipif(not REF.exists(), reason="needs the reference tree (/root/reference)")
def test_gpu_ft_merger_branch_compiles_inside_selecter_and_indextext(tmp_path):
    """The FT half: with integration/patches/0003 applied, the reference's own indextext.cc — which instantiates
    Selector<PackedIdRelVec / IdRelVec>::Process -> mergeResults for ft::MergeData and both MergeDataAreas flavours, and
    IndexText<key_string / PayloadValue>::commitFulltextImpl — compiles with the GPU branch (rx_ft_seam.h: TryMergeOnGpu over
    ft::QueryMergeData / FTConfig / FtDslOpts / FtMergeStatuses::Statuses, SyncGpuFtMirror over DataHolder<IdCont>::words_ and the index as
    DocsStatsGetter), and gpu_ft_merger.cc compiles beside it with the core's flags."""
    tree = _patched_tree(tmp_path)
    patched = tree / "cpp_src/core/index/indextext/indextext.cc"
    assert "SyncGpuFtMirror" in patched.read_text()
    assert "TryMergeOnGpu" in (tree / "cpp_src/core/ft/ft_fast/selecterimpl.h").read_text()
    flags = [f"-I{tree}/cpp_src"] + FLAGS   # the patched headers shadow the reference's
    jobs = {"indextext": patched, "gpu_ft_merger": ROOT / "reindexer_amd" / "host" / "gpu_ft_merger.cc"}

    def compile_one(item):
        name, src = item
        obj = tmp_path / f"{name}.o"
        return name, obj, subprocess.run(["g++", *flags, "-c", str(src), "-o", str(obj)], capture_output=True, text=True)

    with ThreadPoolExecutor(max_workers=2) as ex:
        results = list(ex.map(compile_one, jobs.items()))
    for name, obj, r in results:
        assert r.returncode == 0, f"{name} does not compile inside cpp_src:\n{r.stderr[-4000:]}"
    syms = subprocess.run(["nm", "-C", "--defined-only", str(tmp_path / "indextext.o")], capture_output=True, text=True, check=True).stdout
    for cont in ("reindexer::PackedIdRelVec", "reindexer::IdRelVec"):
        assert f"bool rxgpu::host::TryMergeOnGpu<{cont}, reindexer::ft::MergeData>(" in syms, cont
        assert f"bool rxgpu::host::ToGpuTerms<{cont}>(" in syms, cont
        assert f"rxgpu::host::GpuFtMirror::SyncWords(std::vector<reindexer::PackedWordEntry<{cont}>" in syms, cont
        assert f"reindexer::Selector<{cont}>::mergeResults<" in syms.replace("unsigned short, ", "").replace("unsigned int, ", "") or \
            f"Selector<{cont}>" in syms, cont
    for store in ("reindexer::key_string", "reindexer::PayloadValue"):
        assert f"void rxgpu::host::GpuFtMirror::SyncDocs<reindexer::IndexText<{store}> >(" in syms, store


@pytest.mark.skipif(not REF.exists(), reason="needs the reference tree (/root/reference)")
def test_gpu_ivf_branch_compiles_inside_ivf_index(tmp_path):
    """SURVEY §8f-3 bound into the reference: with integration/patches/0004 applied, the reference's ivf_index.cc — IvfIndex::upsert / del /
    select / selectRaw / reconstruct / getFloatVectorViewImpl / RebuildCentroids / the cache hooks (:88-141, 143-425, 455-497, 625-678) —
    compiles with the GPU branch: its own search helpers (templates over `map`) instantiated over rxgpu::host::GpuIvfInTree (rx_ivf_seam.h:
    faiss::Index's search / range_search signatures on top of GpuIvfFlat), against the reference's vendored FAISS headers; gpu_ivf_flat.cc
    compiles beside it with the core's flags."""
    tree = _patched_tree(tmp_path)
    patched = tree / "cpp_src/core/index/float_vector/ivf_index.cc"
    assert "gpu_->Upsert" in patched.read_text() and "rx_ivf_seam.h" in (tree / "cpp_src/core/index/float_vector/ivf_index.h").read_text()
    flags = [f"-I{tree}/cpp_src", f"-I{tree}/cpp_src/core/index/float_vector"] + FLAGS + ["-DRX_WITH_FAISS_ANN_INDEXES=1"]
    jobs = {"ivf_index": patched, "gpu_ivf_flat": ROOT / "reindexer_amd" / "host" / "gpu_ivf_flat.cc"}

    def compile_one(item):
        name, src = item
        obj = tmp_path / f"{name}.o"
        return name, obj, subprocess.run(["g++", *flags, "-c", str(src), "-o", str(obj)], capture_output=True, text=True)

    with ThreadPoolExecutor(max_workers=2) as ex:
        results = list(ex.map(compile_one, jobs.items()))
    for name, obj, r in results:
        assert r.returncode == 0, f"{name} does not compile inside cpp_src:\n{r.stderr[-4000:]}"
    defined = subprocess.run(["nm", "-C", "--defined-only", str(tmp_path / "ivf_index.o")], capture_output=True, text=True, check=True).stdout
    for member in ("upsert(", "del(", "select(", "selectRaw(", "RebuildCentroids(", "getFloatVectorViewImpl(", "WriteIndexCache(", "LoadIndexCache("):
        assert f"reindexer::IvfIndex::{member}" in defined, member
    # the reference's own search helpers instantiated over the GPU adapter (std::unique_ptr<GpuIvfInTree> as `map`)
    assert "rxgpu::host::GpuIvfInTree" in defined
    undefined = subprocess.run(["nm", "-C", "-u", str(tmp_path / "ivf_index.o")], capture_output=True, text=True, check=True).stdout
    for sym in ("rxgpu::host::GpuIvfFlat::AddWithIds(", "rxgpu::host::GpuIvfFlat::Train(", "rxgpu::host::GpuIvfFlat::RemoveIds(",
                "rxgpu::host::GpuIvfFlat::Search(", "rxgpu::host::GpuIvfFlat::RangeSearch(", "rxgpu::host::GpuIvfFlat::VectorById("):
        assert sym in undefined, sym


@pytest.mark.skipif(not REF.exists(), reason="needs the reference tree (/root/reference)")
def test_gpu_hybrid_fusion_branch_compiles_inside_selectiteratorcontainer(tmp_path):
    """SURVEY §8f-1 bound into the reference: with integration/patches/0005 applied, selectiteratorcontainer.cc — SelectIteratorContainer::
    mergeRanked<desc> (:1454-1559) — compiles with the branch that hands both ranked conditions (KnnRawResult variant, the full-text flat id
    set + RanksHolder ranks, the Reranker variant through the accessors the patch adds to reranker.h) to rxgpu_hybrid_fuse
    (rx_hybrid_seam.h FuseRankedOnGpu)."""
    tree = _patched_tree(tmp_path)
    patched = tree / "cpp_src/core/nsselecter/selectiteratorcontainer.cc"
    assert "FuseRankedOnGpu" in patched.read_text() and "RankConst()" in (tree / "cpp_src/core/sorting/reranker.h").read_text()
    flags = [f"-I{tree}/cpp_src"] + FLAGS + ["-DRX_WITH_FAISS_ANN_INDEXES=1"]
    obj = tmp_path / "selectiteratorcontainer.o"
    r = subprocess.run(["g++", *flags, "-c", str(patched), "-o", str(obj)], capture_output=True, text=True)
    assert r.returncode == 0, f"selectiteratorcontainer.cc does not compile with the GPU fusion branch:\n{r.stderr[-4000:]}"
    defined = subprocess.run(["nm", "-C", "--defined-only", str(obj)], capture_output=True, text=True, check=True).stdout
    assert "reindexer::SelectIteratorContainer::MergeRanked(" in defined
    assert "bool rxgpu::host::FuseRankedOnGpu<" in defined
    undefined = subprocess.run(["nm", "-u", str(obj)], capture_output=True, text=True, check=True).stdout
    assert "rxgpu_hybrid_fuse" in undefined


@pytest.mark.skipif(not REF.exists(), reason="needs the reference tree (/root/reference)")
def test_patches_apply_cleanly(tmp_path):
    _patched_tree(tmp_path)
    cm = (tmp_path / "cpp_src/CMakeLists.txt").read_text()
    assert "WITH_RXGPU" in cm and "RXGPU_IN_TREE" in cm and "gpu_ft_merger.cc" in cm and "gpu_ivf_flat.cc" in cm
