"""CPU-only: the C-ABI library loads and exports every symbol include/rxgpu.h declares (no compute calls)."""
import ctypes
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def header_symbols():
    text = (ROOT / "include" / "rxgpu.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rxgpu_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from reindexer_amd import capi
    assert header_symbols() == capi.declared_symbols()


def test_library_exports_every_declared_symbol():
    from reindexer_amd import capi
    lib = ctypes.CDLL(str(capi.LIB_PATH))
    for name in header_symbols():
        assert hasattr(lib, name), f"librxgpu.so does not export {name}"
    assert lib.rxgpu_abi_version() == 2


def test_product_never_imports_oracle():
    """The oracle is the checker: nothing under reindexer_amd/ or include/ may reference it."""
    for path in list((ROOT / "reindexer_amd").rglob("*")) + list((ROOT / "include").rglob("*")):
        if path.is_file() and path.suffix in {".py", ".h", ".hip", ".cc", ".cpp"}:
            text = path.read_text(errors="replace")
            assert "pyoracle" not in text and "liboracle" not in text and "libref_oracle" not in text, path
            assert not re.search(r"#include\s+[\"<].*oracle", text), path
