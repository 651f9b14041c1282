"""In-tree build of the native libraries (gfx950 only).

    python -m reindexer_amd.build            # build what is stale
    python -m reindexer_amd.build --force

Produces, next to this file:
    librxgpu.so        HIP kernels + the C-ABI of include/rxgpu.h   (hipcc --offload-arch=gfx950)
    librxgpu_host.so   C++ host mirror of the reference's Map / select interfaces (g++), links librxgpu.so

hipcc cross-compiles without a GPU, so this runs in the CPU-only container too.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
HOST = PKG / "host"
INCLUDE = ROOT / "include"

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# -ffp-contract=off: every fused multiply-add in the kernels is an explicit __builtin_fmaf; nothing else may be
# contracted, or the bit-exact summation order of the reference's AVX-512 kernels is lost.
HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
    "-fno-gpu-flush-denormals-to-zero",  # x86 keeps f32 subnormals; so must the kernels
    "-Wall", "-Wno-unused-result", f"-I{INCLUDE}", f"-I{CSRC}",
]
HIP_FLAGS += os.environ.get("RXGPU_HIP_DEFINES", "").split()   # e.g. -DRXGPU_HNSW_PHASES (a profiling build; not what ships)
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-ffp-contract=off", f"-I{INCLUDE}", f"-I{HOST}"]


def _stale(target: Path, sources: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(s.stat().st_mtime > t for s in sources)


def _run(cmd: list[str]) -> None:
    print("+", " ".join(str(c) for c in cmd), flush=True)
    subprocess.run([str(c) for c in cmd], check=True)


def build_device(force: bool = False) -> Path:
    """One object per .hip translation unit (compiled in parallel, only the stale ones), then one link.  No relocatable device code:
    no kernel calls device code of another unit."""
    from concurrent.futures import ThreadPoolExecutor
    out = PKG / "librxgpu.so"
    objdir = PKG / "build" / "obj"
    objdir.mkdir(parents=True, exist_ok=True)
    srcs = sorted(CSRC.glob("*.hip"))
    headers = sorted(CSRC.glob("*.h")) + sorted(INCLUDE.glob("*.h"))
    flags = [f for f in HIP_FLAGS if f != "-shared"]
    todo = []
    for src in srcs:
        obj = objdir / (src.stem + ".o")
        if force or _stale(obj, [src] + headers):
            todo.append((src, obj))
    if todo:
        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4)) as ex:
            list(ex.map(lambda so: _run([HIPCC, *flags, "-c", so[0], "-o", so[1]]), todo))
    objs = [objdir / (src.stem + ".o") for src in srcs]
    if force or todo or _stale(out, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out, "-ldl", "-Wl,-rpath,/opt/rocm/lib"])   # RCCL is dlopen'ed by the first sharded index (rxgpu_sharded.hip)
    return out


def build_host(force: bool = False) -> Path | None:
    srcs = sorted(HOST.glob("*.cc"))
    if not srcs:
        return None
    out = PKG / "librxgpu_host.so"
    deps = srcs + sorted(HOST.glob("*.h")) + sorted(INCLUDE.glob("*.h")) + [PKG / "librxgpu.so"]
    if force or _stale(out, deps):
        _run(["g++", *HOST_FLAGS, *srcs, "-o", out, f"-L{PKG}", "-lrxgpu", "-Wl,-rpath,$ORIGIN", "-lpthread"])
    return out


def build_cpp_tests(force: bool = False) -> list[Path]:
    """tests/cpp/*.cc: reference-style engine tests written against the Map classes (run by pytest -m gpu)."""
    outs = []
    tdir = ROOT / "tests" / "cpp"
    for src in sorted(tdir.glob("*_test.cc")):
        out = src.with_suffix("")
        deps = [src, PKG / "librxgpu_host.so"] + sorted(HOST.glob("*.h"))
        if force or _stale(out, deps):
            _run(["g++", "-O2", "-std=c++17", "-Wall", f"-I{INCLUDE}", f"-I{HOST}", src, "-o", out, f"-L{PKG}", "-lrxgpu_host", "-lrxgpu",
                  "-Wl,-rpath,$ORIGIN/../../reindexer_amd", "-lpthread"])
        outs.append(out)
    # the device decoder of packed postings compiled for the host (CPU check of the code the kernel runs)
    src = tdir / "ft_packed_decode_cpu.cc"
    if src.exists():
        out = tdir / "libft_packed_decode_cpu.so"
        if force or _stale(out, [src, CSRC / "ft_packed_decode.h"]):
            _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", f"-I{CSRC}", src, "-o", out])
        outs.append(out)
    return outs


def build_all(force: bool = False) -> None:
    build_device(force)
    build_host(force)
    build_cpp_tests(force)


if __name__ == "__main__":
    build_all("--force" in sys.argv)
