"""ctypes binding of librxgpu_host.so: the C++ host layer (GpuBruteforceMap, KnnSelect) behind a flat test shim
(reindexer_amd/host/host_capi.cc).  Used by tests and examples; the reference integrates the C++ classes directly."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

PKG = Path(__file__).resolve().parent
HOST_LIB_PATH = PKG / "librxgpu_host.so"

_vp, _sz, _u64, _f, _i, _l = C.c_void_p, C.c_size_t, C.c_uint64, C.c_float, C.c_int, C.c_long
_lib = None


class HostError(RuntimeError):
    pass


class HostLogicError(HostError):
    """std::logic_error on the C++ side."""


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not HOST_LIB_PATH.exists():
            raise RuntimeError(f"{HOST_LIB_PATH} is missing: run `python -m reindexer_amd.build`")
        from . import capi
        capi.lib()  # librxgpu.so first (same SONAME resolution for libamdhip64 as everything else in the process)
        L = _lib = C.CDLL(str(HOST_LIB_PATH))
        L.rxhost_last_error.restype = C.c_char_p
        L.rxhost_l2_module.restype = _f
        L.rxhost_l2_module.argtypes = [_vp, C.c_int32]
        L.rxhost_normalize_copy.restype = _f
        L.rxhost_normalize_copy.argtypes = [_vp, C.c_int32, _vp]
        L.rxhost_bf_create.restype = _vp
        L.rxhost_bf_create.argtypes = [_i, _sz, _sz, _i]
        L.rxhost_bf_clone.restype = _vp
        L.rxhost_bf_clone.argtypes = [_vp, _sz]
        L.rxhost_bf_destroy.argtypes = [_vp]
        L.rxhost_bf_add.argtypes = [_vp, _vp, _sz, _u64]
        L.rxhost_bf_add_many.argtypes = [_vp, _vp, _sz, _sz, _vp]
        L.rxhost_bf_add_concurrent.argtypes = [_vp, _vp, _sz, _u64]
        L.rxhost_bf_remove.argtypes = [_vp, _u64]
        L.rxhost_bf_resize.argtypes = [_vp, _sz]
        for name in ("count", "max_elements", "element_size", "tie_replays"):
            fn = getattr(L, "rxhost_bf_" + name)
            fn.restype = _sz
            fn.argtypes = [_vp]
        L.rxhost_bf_vector_by_label.argtypes = [_vp, _u64, _vp]
        L.rxhost_bf_search_knn.restype = _l
        L.rxhost_bf_search_knn.argtypes = [_vp, _vp, _sz, _vp, _vp]
        L.rxhost_bf_search_knn_filtered.restype = _l
        L.rxhost_bf_search_knn_filtered.argtypes = [_vp, _vp, _sz, _vp, _sz, _vp, _vp]
        L.rxhost_bf_search_range.restype = _l
        L.rxhost_bf_search_range.argtypes = [_vp, _vp, _f, _vp, _vp, _sz]
        L.rxhost_bf_select.restype = _l
        L.rxhost_bf_select.argtypes = [_vp, _vp, _sz, _l, _i, _f, _i, _i, _vp, _vp, _sz]
    return _lib


def _raise(rc=None):
    msg = lib().rxhost_last_error().decode(errors="replace")
    raise (HostLogicError if rc == -4 else HostError)(msg)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _save_bytes(fn, h) -> bytes:
    n = fn(h, None, 0)
    if n < 0:
        _raise()
    buf = np.empty(max(n, 1), np.uint8)
    if fn(h, buf.ctypes.data, n) != n:
        _raise()
    return buf[:n].tobytes()


def _load_bytes(fn, h, data: bytes, labels, vectors, dim: int):
    raw = np.frombuffer(data, np.uint8)
    labels = np.ascontiguousarray(labels, np.uint64).reshape(-1)
    vectors = _f32(vectors).reshape(-1, dim)
    assert vectors.shape[0] == labels.shape[0]
    rc = fn(h, raw.ctypes.data if raw.size else None, raw.size, labels.ctypes.data, vectors.ctypes.data, labels.shape[0])
    if rc:
        _raise(rc)


def l2_module(x) -> np.float32:
    x = _f32(x)
    return np.float32(lib().rxhost_l2_module(x.ctypes.data, x.shape[0]))


def l2_modules_many(rows, threads: int = 0) -> np.ndarray:
    """1/|row| of every row as a cosine index stores it (AddNorm, hnswlib.h:80-92), on `threads` host threads."""
    rows = _f32(rows)
    out = np.empty(rows.shape[0], np.float32)
    L = lib()
    L.rxhost_l2_modules_many.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p, C.c_uint]
    L.rxhost_l2_modules_many.restype = None
    L.rxhost_l2_modules_many(rows.ctypes.data, rows.shape[0], rows.shape[1], out.ctypes.data, max(1, threads or (os.cpu_count() or 1)))
    return out


def normalize_copy(x):
    x = _f32(x)
    out = np.empty_like(x)
    k = lib().rxhost_normalize_copy(x.ctypes.data, x.shape[0], out.ctypes.data)
    return out, np.float32(k)


def parse_device_list(text: str | None) -> list[int]:
    """host/device_list.h: "3" | "0,1,2,3" | "0-7" | "0-3,6" -> device list; empty = no GPU engine."""
    L = lib()
    L.rxhost_parse_device_list.restype = _sz
    L.rxhost_parse_device_list.argtypes = [C.c_char_p, _vp, _sz]
    out = np.zeros(64, np.int32)
    n = L.rxhost_parse_device_list(None if text is None else text.encode(), out.ctypes.data, 64)
    return [int(x) for x in out[:n]]


class GpuBruteforceMap:
    """rxgpu::host::GpuBruteforceMap (drop-in for hnswlib::BruteforceSearch)."""

    def __init__(self, metric: int, dim: int, max_elements: int, device: int = 0, _handle=None, devices=None):
        """devices=[d0, d1, ...]: the Map over a device list (row-range shards; a device may be listed more than once)."""
        self.dim = dim
        if _handle is None and devices is not None:
            L = lib()
            L.rxhost_bf_create_sharded.restype = _vp
            L.rxhost_bf_create_sharded.argtypes = [_i, _sz, _sz, _vp, _sz]
            dv = np.ascontiguousarray(devices, np.int32)
            _handle = L.rxhost_bf_create_sharded(metric, dim, max_elements, dv.ctypes.data, dv.shape[0])
            if not _handle:
                _raise()
        self.h = _handle if _handle is not None else lib().rxhost_bf_create(metric, dim, max_elements, device)
        if not self.h:
            _raise()

    @classmethod
    def from_env(cls, metric: int, dim: int, max_elements: int) -> "GpuBruteforceMap":
        """What the reference's factory constructs through the in-tree adapter (rx_seam.h): `Map(metric, dim, maxElements)`, the device
        list taken from RX_GPU_VECTOR_INDEXES ("3", "0,1,2,3", "0-7"; unset: device 0)."""
        L = lib()
        L.rxhost_bf_create_from_env.restype = _vp
        L.rxhost_bf_create_from_env.argtypes = [_i, _sz, _sz]
        h = L.rxhost_bf_create_from_env(metric, dim, max_elements)
        if not h:
            _raise()
        return cls(metric, dim, max_elements, _handle=h)

    @property
    def sharded(self) -> bool:
        lib().rxhost_bf_is_sharded.argtypes = [_vp]
        return bool(lib().rxhost_bf_is_sharded(self.h))

    def enable_coalescing(self, on: bool) -> None:
        lib().rxhost_bf_enable_coalescing.argtypes = [_vp, _i]
        lib().rxhost_bf_enable_coalescing(self.h, int(on))

    def coalescing_stats(self):
        """(device batches run, queries served) by the query coalescer."""
        lib().rxhost_bf_coalescing_stats.argtypes = [_vp, _vp, _vp]
        a, b = _u64(0), _u64(0)
        lib().rxhost_bf_coalescing_stats(self.h, C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def clone(self, new_max_elements: int) -> "GpuBruteforceMap":
        h = lib().rxhost_bf_clone(self.h, new_max_elements)
        if not h:
            _raise()
        return GpuBruteforceMap(0, self.dim, 0, _handle=h)

    def close(self):
        if getattr(self, "h", None):
            lib().rxhost_bf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add(self, vecs, labels):
        vecs = _f32(vecs).reshape(-1, self.dim)
        labels = np.ascontiguousarray(labels, np.uint64).reshape(-1)
        rc = lib().rxhost_bf_add_many(self.h, vecs.ctypes.data, vecs.shape[0], self.dim, labels.ctypes.data)
        if rc:
            _raise(rc)

    def add_concurrent(self, vec, label):
        vec = _f32(vec)
        rc = lib().rxhost_bf_add_concurrent(self.h, vec.ctypes.data, self.dim, int(label))
        if rc:
            _raise(rc)

    def remove(self, label):
        rc = lib().rxhost_bf_remove(self.h, int(label))
        if rc:
            _raise(rc)

    def resize(self, n):
        rc = lib().rxhost_bf_resize(self.h, n)
        if rc:
            _raise(rc)

    count = property(lambda self: lib().rxhost_bf_count(self.h))
    max_elements = property(lambda self: lib().rxhost_bf_max_elements(self.h))
    element_size = property(lambda self: lib().rxhost_bf_element_size(self.h))
    tie_replays = property(lambda self: lib().rxhost_bf_tie_replays(self.h))

    def vector_by_label(self, label):
        out = np.empty(self.dim, np.float32)
        rc = lib().rxhost_bf_vector_by_label(self.h, int(label), out.ctypes.data)
        if rc:
            _raise(rc)
        return out

    def search_knn(self, q, k):
        q = _f32(q)
        od, ol = np.empty(max(k, 1), np.float32), np.empty(max(k, 1), np.uint64)
        n = lib().rxhost_bf_search_knn(self.h, q.ctypes.data, k, od.ctypes.data, ol.ctypes.data)
        if n < 0:
            _raise()
        return od[:n].copy(), ol[:n].copy()

    def search_knn_filtered(self, q, k, allowed_labels):
        """SearchKnnFiltered: the k nearest among the points whose labels are listed (`WHERE cond AND KNN(...)`)."""
        q = _f32(q)
        al = np.ascontiguousarray(allowed_labels, dtype=np.uint64).reshape(-1)
        od, ol = np.empty(max(k, 1), np.float32), np.empty(max(k, 1), np.uint64)
        n = lib().rxhost_bf_search_knn_filtered(self.h, q.ctypes.data, k, al.ctypes.data if al.size else None, al.size, od.ctypes.data,
                                                ol.ctypes.data)
        if n < 0:
            _raise()
        return od[:n].copy(), ol[:n].copy()

    def search_range(self, q, radius, cap=1 << 20):
        q = _f32(q)
        od, ol = np.empty(cap, np.float32), np.empty(cap, np.uint64)
        n = lib().rxhost_bf_search_range(self.h, q.ctypes.data, radius, od.ctypes.data, ol.ctypes.data, cap)
        if n < 0:
            _raise()
        assert n <= cap
        return od[:n].copy(), ol[:n].copy()

    def select(self, key, k=None, radius=None, need_sort=True, is_array=False, cap=1 << 20):
        key = _f32(key)
        ids, ranks = np.empty(cap, np.int32), np.empty(cap, np.float32)
        n = lib().rxhost_bf_select(self.h, key.ctypes.data, key.shape[0], -1 if k is None else k, int(radius is not None),
                                   0.0 if radius is None else radius, int(need_sort), int(is_array), ids.ctypes.data, ranks.ctypes.data, cap)
        if n < 0:
            _raise()
        return ids[:n].copy(), ranks[:n].copy()


class HnswGraph:
    """rxgpu::host::HnswGraph — the host-side graph builder (no GPU needed to BUILD)."""

    @staticmethod
    def _bind():
        L = lib()
        if not hasattr(L, "_graph_bound"):
            L.rxhost_graph_create.restype = _vp
            L.rxhost_graph_create.argtypes = [_i, _sz, _sz, _sz, _sz]
            L.rxhost_graph_destroy.argtypes = [_vp]
            L.rxhost_graph_add_many.argtypes = [_vp, _vp, _sz, _sz, _vp]
            L.rxhost_graph_add_many_mt.argtypes = [_vp, _vp, _sz, _sz, _vp, C.c_uint]
            L.rxhost_graph_mark_delete.argtypes = [_vp, _u64]
            L.rxhost_graph_info.argtypes = [_vp, _vp]
            L.rxhost_graph_export.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp]
            L.rxhost_graph_vectors.restype = C.POINTER(C.c_float)
            L.rxhost_graph_vectors.argtypes = [_vp]
            L.rxhost_graph_inv_norms.restype = C.POINTER(C.c_float)
            L.rxhost_graph_inv_norms.argtypes = [_vp]
            L.rxhost_graph_save_index.restype = _l
            L.rxhost_graph_save_index.argtypes = [_vp, _vp, _sz]
            L.rxhost_graph_load_index.argtypes = [_vp, _vp, _sz, _vp, _vp, _sz]
            L.rxhost_graph_clear.argtypes = [_vp]
            L._graph_bound = True
        return L

    def __init__(self, metric: int, dim: int, max_elements: int, M: int = 16, ef_construction: int = 200):
        L = self._bind()
        self.dim, self.metric = dim, metric
        self.h = L.rxhost_graph_create(metric, dim, max_elements, M, ef_construction)
        if not self.h:
            _raise()

    def close(self):
        if getattr(self, "h", None):
            lib().rxhost_graph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add(self, vecs, labels, threads: int = 0):
        """threads == 0: sequential AddPoint (the reference's single-threaded graph, link for link); threads >= 2: concurrent construction
        (AddPointConcurrent from `threads` workers, like the reference's multithreaded index build); threads == 1: the concurrent code
        path driven from one thread."""
        vecs = _f32(vecs).reshape(-1, self.dim)
        labels = np.ascontiguousarray(labels, np.uint64).reshape(-1)
        if threads:
            rc = lib().rxhost_graph_add_many_mt(self.h, vecs.ctypes.data, vecs.shape[0], self.dim, labels.ctypes.data, threads)
        else:
            rc = lib().rxhost_graph_add_many(self.h, vecs.ctypes.data, vecs.shape[0], self.dim, labels.ctypes.data)
        if rc:
            _raise(rc)

    def mark_delete(self, label):
        rc = lib().rxhost_graph_mark_delete(self.h, int(label))
        if rc:
            _raise(rc)

    def export(self) -> dict:
        info = np.zeros(7, np.int64)
        lib().rxhost_graph_info(self.h, info.ctypes.data)
        n, M, maxM0, maxlevel, entry, ndel, blocks = (int(x) for x in info)
        links0 = np.zeros((n, 1 + maxM0), np.uint32)
        levels = np.zeros(n, np.int32)
        labels = np.zeros(n, np.uint64)
        deleted = np.zeros(n, np.uint8)
        upper_off = np.zeros(n + 1, np.uint64)
        upper = np.zeros((max(blocks, 1), 1 + M), np.uint32)
        lib().rxhost_graph_export(self.h, links0.ctypes.data, levels.ctypes.data, labels.ctypes.data, deleted.ctypes.data,
                                  upper_off.ctypes.data, upper.ctypes.data)
        return dict(metric=self.metric, n=n, dim=self.dim, M=M, maxM0=maxM0, maxlevel=maxlevel, entry=entry & 0xFFFFFFFF, num_deleted=ndel,
                    links0=links0, upper_off=upper_off, upper=upper, levels=levels, labels=labels, deleted=deleted)

    def save_index(self) -> bytes:
        """The reference's ANN disk cache of this graph (hnswalg.h:1213-1263 behind the float flag of hnsw.cc:56-62), in the test
        encoding of the in-memory writer: 8 bytes per var-int, u64 length + bytes per string, 8-byte label per primary key."""
        return _save_bytes(lib().rxhost_graph_save_index, self.h)

    def load_index(self, data: bytes, labels, vectors):
        """LoadIndex into this EMPTY graph; (labels, vectors) are the namespace's rows the cache's primary keys resolve to."""
        _load_bytes(lib().rxhost_graph_load_index, self.h, data, labels, vectors, self.dim)

    def clear(self):
        lib().rxhost_graph_clear(self.h)

    def vector_views(self, n: int):
        """Zero-copy numpy views of the builder's vectors [n][dim] and (cosine) stored 1/|v| [n], in internal-id order — the order a
        concurrently built graph's ids refer to.  Valid while the graph lives and is not resized."""
        L = lib()
        vp = L.rxhost_graph_vectors(self.h)
        vec = np.ctypeslib.as_array(vp, shape=(n, self.dim))
        ip = L.rxhost_graph_inv_norms(self.h)
        inv = np.ctypeslib.as_array(ip, shape=(n,)) if ip else None
        return vec, inv


class _HnswStream:
    def __init__(self, owner, q, ef, norm=None):
        L = lib()
        L.rxhost_hnsw_stream_begin.restype = _vp
        L.rxhost_hnsw_stream_begin.argtypes = [_vp, _vp, _sz]
        L.rxhost_hnsw_stream_begin_norm.restype = _vp
        L.rxhost_hnsw_stream_begin_norm.argtypes = [_vp, _vp, _i, _f, _sz]
        L.rxhost_hnsw_stream_continue.restype = _l
        L.rxhost_hnsw_stream_continue.argtypes = [_vp, _vp, _sz, _vp, _vp, _vp]
        L.rxhost_hnsw_stream_end.argtypes = [_vp]
        self.owner, self.q = owner, q
        if norm is None:
            self.s = L.rxhost_hnsw_stream_begin(owner.h, q.ctypes.data, ef)
        else:
            self.s = L.rxhost_hnsw_stream_begin_norm(owner.h, q.ctypes.data, 1, float(norm), ef)
        if not self.s:
            _raise()

    def next(self, batch):
        cap = max(1, min(batch, int(self.owner.count) + 1))
        od, ol = np.empty(cap, np.float32), np.empty(cap, np.uint64)
        ex = C.c_int(0)
        n = lib().rxhost_hnsw_stream_continue(self.owner.h, self.s, batch, od.ctypes.data, ol.ctypes.data, C.byref(ex))
        if n < 0:
            _raise()
        return od[:n].copy(), ol[:n].copy(), bool(ex.value)

    def close(self):
        if getattr(self, "s", None):
            lib().rxhost_hnsw_stream_end(self.s)
            self.s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _KnnStream:
    def __init__(self, owner, key, ef):
        L = lib()
        L.rxhost_hnsw_knn_stream_begin.restype = _vp
        L.rxhost_hnsw_knn_stream_begin.argtypes = [_vp, _vp, _sz, _sz]
        L.rxhost_hnsw_knn_stream_continue.restype = _l
        L.rxhost_hnsw_knn_stream_continue.argtypes = [_vp, _vp, _sz, _vp, _vp, _vp]
        L.rxhost_hnsw_knn_stream_end.argtypes = [_vp, _vp]
        self.owner, self.key = owner, key
        self.s = L.rxhost_hnsw_knn_stream_begin(owner.h, key.ctypes.data, key.shape[0], ef)
        if not self.s:
            _raise()

    def next(self, batch):
        cap = max(1, min(batch, int(self.owner.count) + 1))
        ids, ranks = np.empty(cap, np.int32), np.empty(cap, np.float32)
        ex = C.c_int(0)
        n = lib().rxhost_hnsw_knn_stream_continue(self.owner.h, self.s, batch, ids.ctypes.data, ranks.ctypes.data, C.byref(ex))
        if n < 0:
            _raise()
        return ids[:n].copy(), ranks[:n].copy(), bool(ex.value)

    def close(self):
        if getattr(self, "s", None):
            lib().rxhost_hnsw_knn_stream_end(self.owner.h, self.s)
            self.s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def select_postprocess(metric: int, dist, label, k=None, has_radius=False, need_sort=True, is_array=False, raw=False):
    """KnnSelect / KnnSelectRaw (host/knn_select.h) applied to a given Map search result: (ids, ranks)."""
    L = lib()
    L.rxhost_select_postprocess.restype = _l
    L.rxhost_select_postprocess.argtypes = [_i, _vp, _vp, _sz, _l, _i, _i, _i, _i, _vp, _vp]
    d = _f32(dist)
    lab = np.ascontiguousarray(label, np.uint64)
    ids, ranks = np.empty(max(d.shape[0], 1), np.int32), np.empty(max(d.shape[0], 1), np.float32)
    n = L.rxhost_select_postprocess(metric, d.ctypes.data, lab.ctypes.data, d.shape[0], -1 if k is None else k, int(has_radius), int(need_sort),
                                    int(is_array), int(raw), ids.ctypes.data, ranks.ctypes.data)
    if n < 0:
        _raise()
    return ids[:n].copy(), ranks[:n].copy()


def sq8_quantize(metric: int, min_q: float, max_q: float, vec, scale: float = 1.0):
    """Sq8Quantize of the host library (sq8_quantizer.h): (codes, corrective offset, (alpha, alpha_2, delta))."""
    L = lib()
    L.rxhost_sq8_quantize.restype = _f
    L.rxhost_sq8_quantize.argtypes = [_i, _f, _f, _sz, _vp, _f, _vp, _vp]
    v = _f32(vec)
    to = np.empty(v.shape[0], np.uint8)
    params = np.zeros(3, np.float32)
    corr = L.rxhost_sq8_quantize(metric, float(min_q), float(max_q), v.shape[0], v.ctypes.data, float(scale), to.ctypes.data, params.ctypes.data)
    return to, np.float32(corr), params


def sq8_quantize_many(metric: int, min_q: float, max_q: float, vecs, scales=None, threads: int = 0):
    """(codes [n][dim], corr [n]) of n vectors; scales: one multiplier per vector (the 1 / normCoef of cosine queries)."""
    import os
    L = lib()
    L.rxhost_sq8_quantize_many.restype = None
    L.rxhost_sq8_quantize_many.argtypes = [_i, _f, _f, _sz, _vp, _sz, _vp, _vp, _vp, C.c_uint]
    v = np.ascontiguousarray(vecs, np.float32)
    n, dim = v.shape
    codes, corr = np.empty((n, dim), np.uint8), np.empty(n, np.float32)
    sc = None if scales is None else np.ascontiguousarray(scales, np.float32)
    L.rxhost_sq8_quantize_many(metric, float(min_q), float(max_q), dim, v.ctypes.data, n, None if sc is None else sc.ctypes.data, codes.ctypes.data,
                               corr.ctypes.data, threads or len(os.sched_getaffinity(0)))
    return codes, corr


class GpuHnswMap:
    """rxgpu::host::GpuHnswMap (drop-in for hnswlib::HierarchicalNSW<Synchronization::None>; multithread=True: <OnInsertions>, the Map of
    the reference's multithreaded index build — add(..., threads=T) then inserts from T threads through AddPointConcurrent)."""

    def __init__(self, metric: int, dim: int, max_elements: int, M: int = 16, ef_construction: int = 200, device: int = 0, _handle=None,
                 multithread: bool = False, devices=None, _borrowed: bool = False):
        """devices=[d0, d1, ...]: the Map over a device list (a graph per shard; a device may be listed more than once); devices=[] takes
        the list from RX_GPU_VECTOR_INDEXES like the in-tree adapter."""
        L = lib()
        self._borrowed = _borrowed
        if not hasattr(L, "_hnsw_bound"):
            L.rxhost_hnsw_create_sharded.restype = _vp
            L.rxhost_hnsw_create_sharded.argtypes = [_i, _sz, _sz, _sz, _sz, _vp, _sz, _i]
            L.rxhost_hnsw_shard_count.restype = _sz
            L.rxhost_hnsw_shard_count.argtypes = [_vp]
            L.rxhost_hnsw_shard_rows.restype = _sz
            L.rxhost_hnsw_shard_rows.argtypes = [_vp]
            L.rxhost_hnsw_shard.restype = _vp
            L.rxhost_hnsw_shard.argtypes = [_vp, _sz]
            L.rxhost_hnsw_device_index.restype = _vp
            L.rxhost_hnsw_device_index.argtypes = [_vp]
            L.rxhost_hnsw_create.restype = _vp
            L.rxhost_hnsw_create.argtypes = [_i, _sz, _sz, _sz, _sz, _i]
            L.rxhost_hnsw_create_mt.restype = _vp
            L.rxhost_hnsw_create_mt.argtypes = [_i, _sz, _sz, _sz, _sz, _i]
            L.rxhost_hnsw_add_many_mt.argtypes = [_vp, _vp, _sz, _sz, _vp, C.c_uint]
            L.rxhost_hnsw_clone.restype = _vp
            L.rxhost_hnsw_clone.argtypes = [_vp, _sz]
            L.rxhost_hnsw_destroy.argtypes = [_vp]
            L.rxhost_hnsw_add_many.argtypes = [_vp, _vp, _sz, _sz, _vp]
            L.rxhost_hnsw_add_concurrent.argtypes = [_vp, _vp, _sz, _u64]
            L.rxhost_hnsw_mark_delete.argtypes = [_vp, _u64]
            L.rxhost_hnsw_resize.argtypes = [_vp, _sz]
            L.rxhost_hnsw_count.restype = _sz
            L.rxhost_hnsw_count.argtypes = [_vp]
            L.rxhost_hnsw_deleted_count.restype = _sz
            L.rxhost_hnsw_deleted_count.argtypes = [_vp]
            L.rxhost_hnsw_graph.restype = _vp
            L.rxhost_hnsw_graph.argtypes = [_vp]
            L.rxhost_hnsw_search_knn.restype = _l
            L.rxhost_hnsw_search_knn.argtypes = [_vp, _vp, _sz, _sz, _vp, _vp]
            L.rxhost_hnsw_quantize.argtypes = [_vp, _f, _f]
            L.rxhost_hnsw_is_quantized.argtypes = [_vp]
            L.rxhost_hnsw_search_knn_norm.restype = _l
            L.rxhost_hnsw_search_knn_norm.argtypes = [_vp, _vp, _i, _f, _sz, _sz, _vp, _vp]
            L.rxhost_hnsw_search_range.restype = _l
            L.rxhost_hnsw_search_range.argtypes = [_vp, _vp, _f, _sz, _vp, _vp, _sz]
            L.rxhost_hnsw_select.restype = _l
            L.rxhost_hnsw_select.argtypes = [_vp, _vp, _sz, _l, _sz, _i, _f, _i, _i, _vp, _vp, _sz]
            L.rxhost_hnsw_posted_queries.restype = _l
            L.rxhost_hnsw_posted_queries.argtypes = [_vp]
            L.rxhost_hnsw_tie_reruns.restype = _l
            L.rxhost_hnsw_tie_reruns.argtypes = [_vp]
            L.rxhost_hnsw_lds_reruns.restype = _l
            L.rxhost_hnsw_lds_reruns.argtypes = [_vp]
            L.rxhost_hnsw_save_index.restype = _l
            L.rxhost_hnsw_save_index.argtypes = [_vp, _vp, _sz]
            L.rxhost_hnsw_load_index.argtypes = [_vp, _vp, _sz, _vp, _vp, _sz]
            L.rxhost_hnsw_load_index_quantized.argtypes = [_vp, _vp, _sz, _vp, _vp, _sz]
            L._hnsw_bound = True
        self.dim, self.metric = dim, metric
        create = L.rxhost_hnsw_create_mt if multithread else L.rxhost_hnsw_create
        if _handle is not None:
            self.h = _handle
        elif devices is not None:
            dv = np.ascontiguousarray(devices, np.int32)
            self.h = L.rxhost_hnsw_create_sharded(metric, dim, max_elements, M, ef_construction, dv.ctypes.data, dv.shape[0], int(multithread))
        else:
            self.h = create(metric, dim, max_elements, M, ef_construction, device)
        if not self.h:
            _raise()

    shard_count = property(lambda self: lib().rxhost_hnsw_shard_count(self.h))
    shard_rows = property(lambda self: lib().rxhost_hnsw_shard_rows(self.h))

    def shard(self, s: int) -> "GpuHnswMap":
        """Shard s of a Map over a device list as the (borrowed) single-device Map it is."""
        h = lib().rxhost_hnsw_shard(self.h, s)
        if not h:
            _raise()
        return GpuHnswMap(self.metric, self.dim, 0, _handle=h, _borrowed=True)

    @property
    def device_index(self) -> int:
        """rxgpu_index* of the Map's device mirror (sharded: the rxgpu_index_create_sharded handle)."""
        return lib().rxhost_hnsw_device_index(self.h)

    def clone(self, new_capacity: int) -> "GpuHnswMap":
        h = lib().rxhost_hnsw_clone(self.h, new_capacity)
        if not h:
            _raise()
        return GpuHnswMap(self.metric, self.dim, 0, _handle=h)

    def close(self):
        if getattr(self, "h", None):
            if not getattr(self, "_borrowed", False):
                lib().rxhost_hnsw_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add(self, vecs, labels, threads: int = 0):
        vecs = _f32(vecs).reshape(-1, self.dim)
        labels = np.ascontiguousarray(labels, np.uint64).reshape(-1)
        if threads:
            rc = lib().rxhost_hnsw_add_many_mt(self.h, vecs.ctypes.data, vecs.shape[0], self.dim, labels.ctypes.data, threads)
        else:
            rc = lib().rxhost_hnsw_add_many(self.h, vecs.ctypes.data, vecs.shape[0], self.dim, labels.ctypes.data)
        if rc:
            _raise(rc)

    def add_concurrent(self, vec, label):
        vec = _f32(vec)
        rc = lib().rxhost_hnsw_add_concurrent(self.h, vec.ctypes.data, self.dim, int(label))
        if rc:
            _raise(rc)

    def mark_delete(self, label):
        rc = lib().rxhost_hnsw_mark_delete(self.h, int(label))
        if rc:
            _raise(rc)

    def clear(self):
        """Map::Clear (HnswIndexBase::clearMap)"""
        lib().rxhost_hnsw_clear.argtypes = [_vp]
        rc = lib().rxhost_hnsw_clear(self.h)
        if rc:
            _raise(rc)

    def resize(self, n):
        rc = lib().rxhost_hnsw_resize(self.h, n)
        if rc:
            _raise(rc)

    def save_index(self) -> bytes:
        """HnswIndexBase::WriteIndexCache's Map part (hnsw_index.cc:388-437 -> hnsw.cc:56-62): quantisation flag + the graph."""
        return _save_bytes(lib().rxhost_hnsw_save_index, self.h)

    def load_index(self, data: bytes, labels, vectors, with_quantizer: bool = False):
        """LoadIndexCache's Map part (hnsw_index.cc:439-507) into an empty Map; on any error the Map is cleared, as clearMap() does.
        with_quantizer = LoadWithQuantizer: a cache written from a quantised Map brings it back quantised (same parameters)."""
        _load_bytes(lib().rxhost_hnsw_load_index_quantized if with_quantizer else lib().rxhost_hnsw_load_index, self.h, data, labels, vectors, self.dim)

    def lds_reruns(self) -> int:
        """Searches re-run with the largest LDS candidate heap since the last call (their first heap area overflowed)."""
        n = lib().rxhost_hnsw_lds_reruns(self.h)
        if n < 0:
            _raise()
        return n

    def posted_queries(self) -> int:
        """One-shot searches answered through the index's resident search kernel so far (rxgpu_hnsw_search_knn_posted)."""
        return int(lib().rxhost_hnsw_posted_queries(self.h))

    def tie_reruns(self) -> int:
        """Searches re-run on the heap kernel since the last call (the sorted-list search met equal distances)."""
        n = lib().rxhost_hnsw_tie_reruns(self.h)
        if n < 0:
            _raise()
        return n

    count = property(lambda self: lib().rxhost_hnsw_count(self.h))
    deleted_count = property(lambda self: lib().rxhost_hnsw_deleted_count(self.h))

    def export_graph(self, with_views: bool = False) -> dict:
        """Flat graph of the host builder (borrowed HnswGraph handle).  with_views: adds 'vectors' / 'inv_norms' = zero-copy views of
        the builder's storage in internal-id order (valid while this Map lives and is not resized)."""
        HnswGraph._bind()
        g = HnswGraph.__new__(HnswGraph)
        g.dim, g.metric, g.h = self.dim, self.metric, lib().rxhost_hnsw_graph(self.h)
        try:
            e = g.export()
            if with_views:
                e["vectors"], e["inv_norms"] = g.vector_views(e["n"])
            return e
        finally:
            g.h = None   # borrowed: owned by the Map

    def search_knn(self, q, k, ef=0):
        q = _f32(q)
        od, ol = np.empty(max(k, 1), np.float32), np.empty(max(k, 1), np.uint64)
        n = lib().rxhost_hnsw_search_knn(self.h, q.ctypes.data, k, ef, od.ctypes.data, ol.ctypes.data)
        if n < 0:
            _raise()
        return od[:n].copy(), ol[:n].copy()

    def set_coalescer_lanes(self, lanes: int) -> None:
        """Device batches the query coalescer keeps in flight at once (GpuHnswMap::SetCoalescerLanes)."""
        lib().rxhost_hnsw_set_coalescer_lanes.argtypes = [_vp, C.c_uint]
        lib().rxhost_hnsw_set_coalescer_lanes(self.h, int(lanes))

    def search_knn_mt(self, queries, k: int, ef: int, threads: int, per_thread: int, deadline_s: float = 30.0):
        """T native planner threads, one query per SearchKnn call each, over this shared Map (the reference's concurrency model).
        -> (seconds, searches completed, device batches the coalescer ran)."""
        L = lib()
        L.rxhost_hnsw_search_knn_mt.argtypes = [_vp, _vp, _sz, _sz, _sz, _sz, C.c_uint, _sz, C.c_double, _vp, _vp, _vp]
        q = _f32(queries).reshape(-1, self.dim)
        secs, done, batches = C.c_double(0.0), C.c_size_t(0), C.c_size_t(0)
        rc = L.rxhost_hnsw_search_knn_mt(self.h, q.ctypes.data, q.shape[0], self.dim, k, ef, threads, per_thread, float(deadline_s),
                                         C.addressof(secs), C.addressof(done), C.addressof(batches))
        if rc:
            _raise(rc)
        return float(secs.value), int(done.value), int(batches.value)

    def quantize(self, min_q: float, max_q: float) -> None:
        """Quantize: from here on searches run over SQ8 codes on the device (HierarchicalNSWImpl<uint8_t>)."""
        rc = lib().rxhost_hnsw_quantize(self.h, float(min_q), float(max_q))
        if rc:
            _raise(rc)

    def quantize_config(self, sample_size: int = 20000, quantile: float = 0.0, switch: bool = True):
        """HnswIndexBase::Quantize() (+ SwitchMapOnQuantized()): the parameters are sampled from the Map's rows the way QuantizingParams does
        (std::rand reservoir sample, batches of 20 rows, n-th min / max, means).  -> (minQ, maxQ, alpha, alpha_2, delta) when switched on."""
        L = lib()
        L.rxhost_hnsw_quantize_config.argtypes = [_vp, _sz, _f, _i, _vp]
        p = np.zeros(5, np.float32)
        rc = L.rxhost_hnsw_quantize_config(self.h, sample_size, float(quantile), int(switch), p.ctypes.data)
        if rc:
            _raise(rc)
        return p

    def switch_on_quantized(self) -> None:
        lib().rxhost_hnsw_switch_on_quantized.argtypes = [_vp]
        rc = lib().rxhost_hnsw_switch_on_quantized(self.h)
        if rc:
            _raise(rc)

    @property
    def quantizing_params(self):
        L = lib()
        L.rxhost_hnsw_quantizing_params.argtypes = [_vp, _vp]
        p = np.zeros(5, np.float32)
        L.rxhost_hnsw_quantizing_params(self.h, p.ctypes.data)
        return p

    @property
    def is_quantized(self) -> bool:
        return bool(lib().rxhost_hnsw_is_quantized(self.h))

    def search_knn_norm(self, q, k, ef=0, norm=None):
        """SearchKnn(query, query_data_norm, k, ef): the norm is required by a quantised cosine graph."""
        q = _f32(q)
        od, ol = np.empty(max(k, 1), np.float32), np.empty(max(k, 1), np.uint64)
        n = lib().rxhost_hnsw_search_knn_norm(self.h, q.ctypes.data, int(norm is not None), 0.0 if norm is None else float(norm), k, ef,
                                              od.ctypes.data, ol.ctypes.data)
        if n < 0:
            _raise()
        return od[:n].copy(), ol[:n].copy()

    def stream(self, q, ef=0, norm=None):
        """BeginStreamingSearch: session with .next(batch) -> (dist, label, exhausted) (worst first) and .close().  norm: the query's
        norm as HnswIndexBase::search passes it (needed by a quantised cosine graph)."""
        return _HnswStream(self, _f32(q), ef, norm)

    def knn_stream(self, key, ef=0):
        """Index-level streaming (HnswIndexBase<Map>::beginStreaming / continueStreaming): raw key in; .next(batch) -> (row ids,
        user-visible ranks, exhausted), best first."""
        return _KnnStream(self, _f32(key), ef)

    def search_range(self, q, radius, ef, cap=1 << 20, norm=None):
        q = _f32(q)
        od, ol = np.empty(cap, np.float32), np.empty(cap, np.uint64)
        L = lib()
        if norm is None:
            n = L.rxhost_hnsw_search_range(self.h, q.ctypes.data, radius, ef, od.ctypes.data, ol.ctypes.data, cap)
        else:
            L.rxhost_hnsw_search_range_norm.restype = _l
            L.rxhost_hnsw_search_range_norm.argtypes = [_vp, _vp, _i, _f, _f, _sz, _vp, _vp, _sz]
            n = L.rxhost_hnsw_search_range_norm(self.h, q.ctypes.data, 1, float(norm), radius, ef, od.ctypes.data, ol.ctypes.data, cap)
        if n < 0:
            _raise()
        return od[:n].copy(), ol[:n].copy()

    def select(self, key, k=None, ef=0, radius=None, need_sort=True, is_array=False, cap=1 << 16):
        key = _f32(key)
        ids, ranks = np.empty(cap, np.int32), np.empty(cap, np.float32)
        n = lib().rxhost_hnsw_select(self.h, key.ctypes.data, key.shape[0], -1 if k is None else k, ef, int(radius is not None),
                                     0.0 if radius is None else radius, int(need_sort), int(is_array), ids.ctypes.data, ranks.ctypes.data, cap)
        if n < 0:
            _raise()
        return ids[:n].copy(), ranks[:n].copy()


def set_ft_train_mode(mode: int) -> None:
    """rxgpu_ft_set_train_mode: -1 the host picks the launch train per query, 0 always the dense train, 1 the sparse train whenever eligible."""
    from . import capi
    capi.lib().rxgpu_ft_set_train_mode(int(mode))


class GpuFtMerger:
    """rxgpu::host::GpuFtMerger — ft_fast single-term BM25 merge on the GPU (stand-in for ft::Merger::Merge<Bm25Rx>)."""

    def __init__(self, num_fields: int, device: int = 0, devices=None):
        """devices=[d0, d1, ...]: the merger over a device list — document-range shards (a device may be listed more than once)."""
        L = lib()
        if not hasattr(L, "_ft_bound"):
            L.rxhost_ft_create_sharded.restype = _vp
            L.rxhost_ft_create_sharded.argtypes = [_sz, _vp, _sz]
            L.rxhost_ft_device_index.restype = _vp
            L.rxhost_ft_device_index.argtypes = [_vp]
            L.rxhost_ft_create.restype = _vp
            L.rxhost_ft_create.argtypes = [_sz, _i]
            L.rxhost_ft_destroy.argtypes = [_vp]
            L.rxhost_ft_set_docs.argtypes = [_vp, _sz, _vp, _vp, _vp]
            L.rxhost_ft_set_word.argtypes = [_vp, C.c_uint32, _sz, _vp, _vp, _vp, _vp]
            L.rxhost_ft_merge.restype = _l
            L.rxhost_ft_merge.argtypes = [_vp, _sz, _vp, _vp, _vp, _f, _f, _vp, _vp, _sz, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz]
            L.rxhost_ft_read_stats.argtypes = [_vp, _vp, _vp]
            L._ft_bound = True
        self.nf = num_fields
        if devices is not None:
            dv = np.ascontiguousarray(devices, np.int32)
            self.h = L.rxhost_ft_create_sharded(num_fields, dv.ctypes.data, dv.shape[0])
        else:
            self.h = L.rxhost_ft_create(num_fields, device)
        if not self.h:
            _raise()

    @property
    def device_index(self) -> int:
        """rxgpu_ft_index* behind the merger (sharded: the rxgpu_ft_create_sharded handle)."""
        return lib().rxhost_ft_device_index(self.h)

    def close(self):
        if getattr(self, "h", None):
            lib().rxhost_ft_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_docs(self, words, avg, removed=None):
        words = _f32(words).reshape(-1, self.nf)
        avg = _f32(avg)
        rem = np.ascontiguousarray(removed, np.uint8) if removed is not None else None
        rc = lib().rxhost_ft_set_docs(self.h, words.shape[0], words.ctypes.data, avg.ctypes.data, rem.ctypes.data if rem is not None else None)
        if rc:
            _raise(rc)

    def set_word_flat(self, word_id, s):
        """s: flat sub-term dict (doc, ent_off, ent_field, ent_tf, ent_first_pos) -> re-expanded to (field,pos) records whose grouping
        reproduces the same entries (tf positions per field, the first one at ent_first_pos)."""
        doc = np.ascontiguousarray(s["doc"], np.uint32)
        pos_off, pf, pp = [0], [], []
        for i in range(doc.shape[0]):
            for e in range(int(s["ent_off"][i]), int(s["ent_off"][i + 1])):
                for t in range(int(s["ent_tf"][e])):
                    pf.append(int(s["ent_field"][e]))
                    pp.append(int(s["ent_first_pos"][e]) + t)
            pos_off.append(len(pf))
        po, pfa, ppa = np.array(pos_off, np.uint32), np.array(pf, np.uint32), np.array(pp, np.uint32)
        rc = lib().rxhost_ft_set_word(self.h, word_id, doc.shape[0], doc.ctypes.data, po.ctypes.data, pfa.ctypes.data, ppa.ctypes.data)
        if rc:
            _raise(rc)

    def merge(self, cfg: dict, opts: dict, subterms, excluded=None, sort_by_rank=True):
        """cfg/opts: dicts as built by default_ft_config()/default_ft_opts(); subterms: [(word_id, proc), ...]."""
        nf = self.nf
        cfg_d = np.array([cfg["k1"], cfg["b"], cfg["summation_ratio"], cfg["full_match_boost"]], np.float64)
        cfg_i = np.array([cfg["min_rank"], cfg["merge_limit"], {"rx": 0, "classic": 1, "word_count": 2}[cfg.get("bm25_type", "rx")]], np.int32)
        fc = np.stack([np.asarray(cfg[k], np.float64) for k in ("bm25_boost", "bm25_weight", "term_len_boost", "term_len_weight",
                                                               "position_boost", "position_weight")], axis=1).copy()
        fb = _f32(opts["field_boost"])
        ns = np.ascontiguousarray(opts["need_sum_rank"], np.uint8)
        wid = np.array([s[0] for s in subterms], np.uint32)
        pr = np.array([s[1] for s in subterms], np.float32)
        exc = np.ascontiguousarray(excluded, np.uint8) if excluded is not None else None
        cap = int(cfg["merge_limit"])
        oid, op = np.zeros(cap, np.int32), np.zeros(cap, np.float32)
        of, on = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8)
        n = lib().rxhost_ft_merge(self.h, nf, cfg_d.ctypes.data, cfg_i.ctypes.data, fc.ctypes.data, opts["boost"], opts["term_len_boost"],
                                  fb.ctypes.data, ns.ctypes.data, len(subterms), wid.ctypes.data, pr.ctypes.data,
                                  exc.ctypes.data if exc is not None else None, int(sort_by_rank), oid.ctypes.data, op.ctypes.data,
                                  of.ctypes.data, on.ctypes.data, cap)
        if n < 0:
            _raise()
        return oid[:n].copy(), op[:n].copy(), of[:n].copy(), on[:n].copy()

    def set_word_fpos(self, word_id, s):
        """s: positions-format sub-term dict (doc, pos_off, fpos[u64 PosType words]); serves merge() and merge_query()."""
        L = lib()
        L.rxhost_ft_set_word_fpos.argtypes = [_vp, C.c_uint32, _sz, _vp, _vp, _vp]
        doc = np.ascontiguousarray(s["doc"], np.uint32)
        po = np.ascontiguousarray(s["pos_off"], np.uint32)
        fp = np.ascontiguousarray(s["fpos"], np.uint64)
        rc = L.rxhost_ft_set_word_fpos(self.h, word_id, doc.shape[0], doc.ctypes.data, po.ctypes.data, fp.ctypes.data)
        if rc:
            _raise(rc)

    def set_words_packed(self, words, host_from_bytes=1 << 18):
        """words: [(word_id, bytes (uint8 array: a PackedIdRelVec stream), array_found_pos), ...] — uploaded in one piece and decoded on the
        device (GpuFtMerger::SetWordsPacked); streams of host_from_bytes bytes or more are decoded on the host."""
        L = lib()
        L.rxhost_ft_set_words_packed.argtypes = [_vp, C.c_uint32, _vp, _vp, _vp, _vp, _sz]
        ids = np.array([w[0] for w in words], np.uint32)
        datas = [np.ascontiguousarray(w[1], np.uint8) for w in words]
        off = np.zeros(len(words) + 1, np.uint64)
        off[1:] = np.cumsum([d.shape[0] for d in datas])
        blob = np.concatenate(datas) if datas else np.zeros(0, np.uint8)
        blob = np.ascontiguousarray(np.append(blob, np.zeros(1, np.uint8)))   # never a null pointer
        afp = np.array([min(int(w[2]), 1 << 62) for w in words], np.uint64)
        rc = L.rxhost_ft_set_words_packed(self.h, len(words), ids.ctypes.data, off.ctypes.data, blob.ctypes.data, afp.ctypes.data, host_from_bytes)
        if rc:
            _raise(rc)

    def read_packed_wall(self) -> float:
        """ms spent inside the library's packed-upload calls since the last call (the commit-side cost at the C-ABI boundary)"""
        L = lib()
        L.rxhost_ft_read_packed_wall.restype = C.c_double
        L.rxhost_ft_read_packed_wall.argtypes = [_vp]
        return float(L.rxhost_ft_read_packed_wall(self.h))

    def get_word(self, word_id):
        """The word's device arrays read back: dict(doc, pos_off, fpos, ent_off, ent_field, ent_tf, ent_first, range_off)."""
        L = lib()
        L.rxhost_ft_get_word.argtypes = [_vp, C.c_uint32] + [_vp] * 9
        sizes = np.zeros(4, np.uint64)
        rc = L.rxhost_ft_get_word(self.h, word_id, sizes.ctypes.data, None, None, None, None, None, None, None, None)
        if rc:
            _raise(rc)
        n, npos, nent, nr = (int(x) for x in sizes)
        out = dict(doc=np.zeros(n, np.uint32), pos_off=np.zeros(n + 1, np.uint32), fpos=np.zeros(npos, np.uint64), ent_off=np.zeros(n + 1, np.uint32),
                   ent_field=np.zeros(nent, np.uint8), ent_tf=np.zeros(nent, np.uint32), ent_first=np.zeros(nent, np.uint32),
                   range_off=np.zeros(nr, np.uint32))
        rc = L.rxhost_ft_get_word(self.h, word_id, sizes.ctypes.data, *(out[k].ctypes.data for k in
                                  ("doc", "pos_off", "fpos", "ent_off", "ent_field", "ent_tf", "ent_first", "range_off")))
        if rc:
            _raise(rc)
        return out

    OP_OR, OP_AND, OP_NOT = 1, 2, 3

    def merge_query(self, cfg: dict, terms, excluded=None, sort_by_rank=True, synonyms=None, part_synonyms=None):
        """synonyms: [[term, ...], ...] multi-word synonyms (Synonym::Terms()), part_synonyms[i]: ids of the synonyms of query part i
        (PhraseOrTerm::SynonymsIds) -> GpuFtMerger::MergeQuery with QuerySynonyms.
        Multi-term Merger::Merge.  terms: [dict(op, opts, subs=[(word_id, proc), ...][, phrase=<phraseNum>, distance=<d>]), ...];
        consecutive terms with the same phrase number >= 0 are one phrase (FtDslOpts::phraseNum / distance).
        Returns (ids, proc, field, norm, preselected)."""
        L = lib()
        n_part_terms = len(terms)
        syn_off = [0]
        if synonyms:
            terms = list(terms)
            for syn in synonyms:
                terms.extend(syn)
                syn_off.append(len(terms) - n_part_terms)
        L.rxhost_ft_merge_query_phrases.restype = _l
        L.rxhost_ft_merge_query_phrases.argtypes = [_vp, _sz, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp,
                                                    _vp, _sz, _vp]
        nf = self.nf
        cfg_d = np.array([cfg["k1"], cfg["b"], cfg["summation_ratio"], cfg["full_match_boost"], cfg.get("distance_boost", 1.0),
                          cfg.get("distance_weight", 0.5)], np.float64)
        cfg_i = np.array([cfg["min_rank"], cfg["merge_limit"], {"rx": 0, "classic": 1, "word_count": 2}[cfg.get("bm25_type", "rx")]], np.int32)
        fc = np.stack([np.asarray(cfg[k], np.float64) for k in ("bm25_boost", "bm25_weight", "term_len_boost", "term_len_weight",
                                                               "position_boost", "position_weight")], axis=1).copy()
        ops = np.array([t["op"] for t in terms], np.int32)
        boosts = np.array([t["opts"]["boost"] for t in terms], np.float32)
        tlb = np.array([t["opts"]["term_len_boost"] for t in terms], np.float32)
        fb = np.array([t["opts"]["field_boost"] for t in terms], np.float32).reshape(len(terms), nf).copy()
        ns = np.array([t["opts"]["need_sum_rank"] for t in terms], np.uint8).reshape(len(terms), nf).copy()
        phr = np.array([t.get("phrase", -1) for t in terms], np.int32)
        dst = np.array([t.get("distance", 1) for t in terms], np.int32)
        sub_off, wid, pr = [0], [], []
        for t in terms:
            for w, p in t["subs"]:
                wid.append(w)
                pr.append(p)
            sub_off.append(len(wid))
        sub_off, wid, pr = np.array(sub_off, np.uint32), np.array(wid, np.uint32), np.array(pr, np.float32)
        exc = np.ascontiguousarray(excluded, np.uint8) if excluded is not None else None
        cap = int(cfg["merge_limit"])
        oid, op = np.zeros(cap, np.int32), np.zeros(cap, np.float32)
        of, on = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8)
        pre = C.c_int(0)
        if synonyms:
            nparts = sum(1 for i in range(n_part_terms) if phr[i] < 0 or i == 0 or phr[i - 1] != phr[i])
            ps_off, ps = [0], []
            for pi in range(nparts):
                ps.extend(part_synonyms[pi] if part_synonyms and pi < len(part_synonyms) else [])
                ps_off.append(len(ps))
            syn_off_a, ps_off_a, ps_a = np.array(syn_off, np.uint32), np.array(ps_off, np.uint32), np.array(ps + [0], np.uint32)
            L.rxhost_ft_merge_query_full.restype = _l
            L.rxhost_ft_merge_query_full.argtypes = [_vp, _sz, _vp, _vp, _vp, _sz, _sz] + [_vp] * 10 + [_sz, _vp, _sz, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]
            n = L.rxhost_ft_merge_query_full(self.h, nf, cfg_d.ctypes.data, cfg_i.ctypes.data, fc.ctypes.data, n_part_terms, len(terms) - n_part_terms,
                                             ops.ctypes.data, boosts.ctypes.data, tlb.ctypes.data, fb.ctypes.data, ns.ctypes.data, phr.ctypes.data,
                                             dst.ctypes.data, sub_off.ctypes.data, wid.ctypes.data, pr.ctypes.data, len(synonyms), syn_off_a.ctypes.data,
                                             nparts, ps_off_a.ctypes.data, ps_a.ctypes.data, exc.ctypes.data if exc is not None else None,
                                             int(sort_by_rank), oid.ctypes.data, op.ctypes.data, of.ctypes.data, on.ctypes.data, cap, C.byref(pre))
            if n < 0:
                _raise()
            return oid[:n].copy(), op[:n].copy(), of[:n].copy(), on[:n].copy(), bool(pre.value)
        n = L.rxhost_ft_merge_query_phrases(self.h, nf, cfg_d.ctypes.data, cfg_i.ctypes.data, fc.ctypes.data, len(terms), ops.ctypes.data,
                                            boosts.ctypes.data, tlb.ctypes.data, fb.ctypes.data, ns.ctypes.data, phr.ctypes.data, dst.ctypes.data,
                                            sub_off.ctypes.data, wid.ctypes.data, pr.ctypes.data, exc.ctypes.data if exc is not None else None,
                                            int(sort_by_rank), oid.ctypes.data, op.ctypes.data, of.ctypes.data, on.ctypes.data, cap, C.byref(pre))
        if n < 0:
            _raise()
        return oid[:n].copy(), op[:n].copy(), of[:n].copy(), on[:n].copy(), bool(pre.value)

    def merge_query_batch(self, cfg: dict, queries, sort_by_rank=True):
        """GpuFtMerger::MergeQueryBatch: `queries` = a list of term lists as merge_query takes them (no synonyms), merged in ONE launch train.
        Returns a list of (ids, proc, field, norm, preselected) — what merge_query returns per query."""
        L = lib()
        L.rxhost_ft_merge_query_batch.restype = _i
        L.rxhost_ft_merge_query_batch.argtypes = [_vp, _sz, _vp, _vp, _vp, _sz] + [_vp] * 11 + [_i, _vp, _vp, _vp, _vp, _sz, _vp, _vp]
        nf = self.nf
        cfg_d = np.array([cfg["k1"], cfg["b"], cfg["summation_ratio"], cfg["full_match_boost"], cfg.get("distance_boost", 1.0),
                          cfg.get("distance_weight", 0.5)], np.float64)
        cfg_i = np.array([cfg["min_rank"], cfg["merge_limit"], {"rx": 0, "classic": 1, "word_count": 2}[cfg.get("bm25_type", "rx")]], np.int32)
        fc = np.stack([np.asarray(cfg[k], np.float64) for k in ("bm25_boost", "bm25_weight", "term_len_boost", "term_len_weight",
                                                               "position_boost", "position_weight")], axis=1).copy()
        terms = [t for q in queries for t in q]
        term_off = np.zeros(len(queries) + 1, np.uint32)
        term_off[1:] = np.cumsum([len(q) for q in queries])
        nt = max(len(terms), 1)
        ops = np.array([t["op"] for t in terms] or [1], np.int32)
        boosts = np.array([t["opts"]["boost"] for t in terms] or [1.0], np.float32)
        tlb = np.array([t["opts"]["term_len_boost"] for t in terms] or [1.0], np.float32)
        fb = np.array([t["opts"]["field_boost"] for t in terms] or [[1.0] * nf], np.float32).reshape(nt, nf).copy()
        ns = np.array([t["opts"]["need_sum_rank"] for t in terms] or [[0] * nf], np.uint8).reshape(nt, nf).copy()
        phr = np.array([t.get("phrase", -1) for t in terms] or [-1], np.int32)
        dst = np.array([t.get("distance", 1) for t in terms] or [1], np.int32)
        sub_off, wid, pr = [0], [], []
        for t in terms:
            for w, p in t["subs"]:
                wid.append(w)
                pr.append(p)
            sub_off.append(len(wid))
        sub_off, wid, pr = np.array(sub_off, np.uint32), np.array(wid + [0], np.uint32), np.array(pr + [0.0], np.float32)
        cap, nq = int(cfg["merge_limit"]), len(queries)
        oid, op = np.zeros((nq, cap), np.int32), np.zeros((nq, cap), np.float32)
        of, on = np.zeros((nq, cap), np.uint8), np.zeros((nq, cap), np.uint8)
        cnt, pre = np.zeros(nq, np.int64), np.zeros(nq, np.int32)
        rc = L.rxhost_ft_merge_query_batch(self.h, nf, cfg_d.ctypes.data, cfg_i.ctypes.data, fc.ctypes.data, nq, term_off.ctypes.data, ops.ctypes.data,
                                           boosts.ctypes.data, tlb.ctypes.data, fb.ctypes.data, ns.ctypes.data, phr.ctypes.data, dst.ctypes.data,
                                           sub_off.ctypes.data, wid.ctypes.data, pr.ctypes.data, int(sort_by_rank), oid.ctypes.data, op.ctypes.data,
                                           of.ctypes.data, on.ctypes.data, cap, cnt.ctypes.data, pre.ctypes.data)
        if rc:
            _raise()
        return [(oid[i, :cnt[i]].copy(), op[i, :cnt[i]].copy(), of[i, :cnt[i]].copy(), on[i, :cnt[i]].copy(), bool(pre[i])) for i in range(nq)]

    def merge_query_concurrent(self, cfg: dict, terms, threads: int, repeats: int, excluded=None):
        """`threads` native threads issue the same query `repeats` times each against this merger (GpuFtMerger::MergeQuery is what several
        planner threads call at once; the merges spread over the handle's lanes).  Every result is checked against the first.
        Returns (results per merge, wall ms of the whole run)."""
        L = lib()
        L.rxhost_ft_merge_query_concurrent.restype = _l
        L.rxhost_ft_merge_query_concurrent.argtypes = [_vp, _sz, _vp, _vp, _vp, _sz] + [_vp] * 11 + [_i, _i, _vp]
        nf = self.nf
        cfg_d = np.array([cfg["k1"], cfg["b"], cfg["summation_ratio"], cfg["full_match_boost"], cfg.get("distance_boost", 1.0),
                          cfg.get("distance_weight", 0.5)], np.float64)
        cfg_i = np.array([cfg["min_rank"], cfg["merge_limit"], {"rx": 0, "classic": 1, "word_count": 2}[cfg.get("bm25_type", "rx")]], np.int32)
        fc = np.stack([np.asarray(cfg[k], np.float64) for k in ("bm25_boost", "bm25_weight", "term_len_boost", "term_len_weight",
                                                               "position_boost", "position_weight")], axis=1).copy()
        ops = np.array([t["op"] for t in terms], np.int32)
        boosts = np.array([t["opts"]["boost"] for t in terms], np.float32)
        tlb = np.array([t["opts"]["term_len_boost"] for t in terms], np.float32)
        fb = np.array([t["opts"]["field_boost"] for t in terms], np.float32).reshape(len(terms), nf).copy()
        ns = np.array([t["opts"]["need_sum_rank"] for t in terms], np.uint8).reshape(len(terms), nf).copy()
        phr = np.array([t.get("phrase", -1) for t in terms], np.int32)
        dst = np.array([t.get("distance", 1) for t in terms], np.int32)
        sub_off, wid, pr = [0], [], []
        for t in terms:
            for w, p in t["subs"]:
                wid.append(w)
                pr.append(p)
            sub_off.append(len(wid))
        sub_off, wid, pr = np.array(sub_off, np.uint32), np.array(wid, np.uint32), np.array(pr, np.float32)
        exc = np.ascontiguousarray(excluded, np.uint8) if excluded is not None else None
        wall = C.c_double(0.0)
        n = L.rxhost_ft_merge_query_concurrent(self.h, nf, cfg_d.ctypes.data, cfg_i.ctypes.data, fc.ctypes.data, len(terms), ops.ctypes.data,
                                               boosts.ctypes.data, tlb.ctypes.data, fb.ctypes.data, ns.ctypes.data, phr.ctypes.data, dst.ctypes.data,
                                               sub_off.ctypes.data, wid.ctypes.data, pr.ctypes.data, exc.ctypes.data if exc is not None else None,
                                               int(threads), int(repeats), C.byref(wall))
        if n < 0:
            _raise()
        return int(n), float(wall.value)

    def hybrid_query(self, cfg: dict, terms, knn_dist_ptr: int, knn_row_ptr: int, knn_entries: int, k: int, metric: int, kind="rrf", params=(60.0,),
                     union=True, desc=True, excluded=None, knn_count_ptr: int = 0, knn_stream: int = 0, row_of_doc_ptr: int = 0, rowid_of_row_ptr: int = 0):
        """Hybrid query with everything resident: the FT merge stays in HBM (GpuFtMerger::MergeQueryResident), the KNN result lies in HBM
        ((dist, row) best first at the given device pointers, e.g. from VectorIndex.search_knn_device), the fusion kernel applies
        postProcessResults + MergerRankedImpl and ONE list of (row id, fused rank) comes back.  Returns (ids, ranks, boundary_tie)."""
        L = lib()
        L.rxhost_ft_hybrid_query.restype = _l
        L.rxhost_ft_hybrid_query.argtypes = [_vp, _sz, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp,
                                             C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]
        nf = self.nf
        cfg_d = np.array([cfg["k1"], cfg["b"], cfg["summation_ratio"], cfg["full_match_boost"], cfg.get("distance_boost", 1.0),
                          cfg.get("distance_weight", 0.5)], np.float64)
        cfg_i = np.array([cfg["min_rank"], cfg["merge_limit"], {"rx": 0, "classic": 1, "word_count": 2}[cfg.get("bm25_type", "rx")]], np.int32)
        fc = np.stack([np.asarray(cfg[k_], np.float64) for k_ in ("bm25_boost", "bm25_weight", "term_len_boost", "term_len_weight",
                                                                 "position_boost", "position_weight")], axis=1).copy()
        ops = np.array([t["op"] for t in terms], np.int32)
        boosts = np.array([t["opts"]["boost"] for t in terms], np.float32)
        tlb = np.array([t["opts"]["term_len_boost"] for t in terms], np.float32)
        fb = np.array([t["opts"]["field_boost"] for t in terms], np.float32).reshape(len(terms), nf).copy()
        ns = np.array([t["opts"]["need_sum_rank"] for t in terms], np.uint8).reshape(len(terms), nf).copy()
        sub_off, wid, pr = [0], [], []
        for t in terms:
            for w, p_ in t["subs"]:
                wid.append(w)
                pr.append(p_)
            sub_off.append(len(wid))
        sub_off, wid, pr = np.array(sub_off, np.uint32), np.array(wid, np.uint32), np.array(pr, np.float32)
        exc = np.ascontiguousarray(excluded, np.uint8) if excluded is not None else None
        hyb = np.array([0 if kind == "rrf" else 1, int(union), int(desc)], np.int32)
        par = np.zeros(5, np.float64)
        par[:len(params)] = params
        cap = int(cfg["merge_limit"]) + int(k)
        oid, orank = np.zeros(cap, np.int32), np.zeros(cap, np.float32)
        tie = C.c_int(0)
        n = L.rxhost_ft_hybrid_query(self.h, nf, cfg_d.ctypes.data, cfg_i.ctypes.data, fc.ctypes.data, len(terms), ops.ctypes.data, boosts.ctypes.data,
                                     tlb.ctypes.data, fb.ctypes.data, ns.ctypes.data, sub_off.ctypes.data, wid.ctypes.data, pr.ctypes.data,
                                     exc.ctypes.data if exc is not None else None, hyb.ctypes.data, par.ctypes.data, metric, knn_dist_ptr, knn_row_ptr,
                                     knn_count_ptr or None, knn_entries, k, knn_stream or None, row_of_doc_ptr or None, rowid_of_row_ptr or None,
                                     oid.ctypes.data, orank.ctypes.data, cap, C.byref(tie))
        if n < 0:
            _raise()
        return oid[:n].copy(), orank[:n].copy(), bool(tie.value)

    def read_packed_stats(self):
        """set_words_packed since the last call: (count kernel ms, write kernel ms, stream bytes read per pass, array bytes produced)."""
        L = lib()
        L.rxhost_ft_read_packed_stats.restype = None
        L.rxhost_ft_read_packed_stats.argtypes = [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_u64), C.POINTER(_u64)]
        a, b, c, d = C.c_double(0), C.c_double(0), _u64(0), _u64(0)
        L.rxhost_ft_read_packed_stats(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return float(a.value), float(b.value), int(c.value), int(d.value)

    def read_fuse_stats(self):
        """(fusions, device ms of the join kernel — the part on the critical path —, device ms of the overlapped FT-only prepare kernel)
        since the last call."""
        L = lib()
        L.rxhost_ft_read_fuse_stats.restype = None
        L.rxhost_ft_read_fuse_stats.argtypes = [_vp, C.POINTER(_u64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        calls, ms, pms = _u64(0), C.c_double(0), C.c_double(0)
        L.rxhost_ft_read_fuse_stats(self.h, C.byref(calls), C.byref(ms), C.byref(pms))
        return int(calls.value), float(ms.value), float(pms.value)

    def read_train_stats(self):
        """(dense merges, sparse merges) since the last call: which launch train ran them (rxgpu_ft_read_train_stats)."""
        from . import capi
        a, b = _u64(0), _u64(0)
        capi.lib().rxgpu_ft_read_train_stats(self.device_index, C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def read_timing(self):
        """(calls, total ms) spent inside the C++ Merger since the last call — the end-to-end time of the drop-in boundary, without this
        Python wrapper's argument marshalling."""
        L = lib()
        L.rxhost_ft_read_timing.restype = None
        L.rxhost_ft_read_timing.argtypes = [_vp, C.POINTER(_u64), C.POINTER(C.c_double)]
        calls, ms = _u64(0), C.c_double(0.0)
        L.rxhost_ft_read_timing(self.h, C.byref(calls), C.byref(ms))
        return int(calls.value), float(ms.value)

    def read_stats(self):
        a, b = _u64(0), C.c_double(0.0)
        lib().rxhost_ft_read_stats(self.h, C.byref(a), C.byref(b))
        return int(a.value), float(b.value)


def hybrid_query_resident(vmap: "GpuBruteforceMap", ftm: "GpuFtMerger", cfg: dict, terms, key, k: int, kind="rrf", params=(60.0,), union=True,
                          desc=True, excluded=None, row_of_doc_ptr: int = 0, host_row_of_doc=None, synonyms=None, part_synonyms=None):
    """rxgpu::host::HybridQueryResident (hybrid_query.h): the hybrid query through the Map and the Merger with both halves left in HBM and
    fused there.  key: the query vector as given by the user.  synonyms / part_synonyms: as GpuFtMerger.merge_query takes them (multi-word
    synonyms; the resident merge keeps the documents it removes marked, the fusion skips them).
    Returns (row ids, fused ranks, boundary_tie_redone_on_host)."""
    L = lib()
    if synonyms:
        return _hybrid_query_resident_full(vmap, ftm, cfg, terms, key, k, kind, params, union, desc, excluded, row_of_doc_ptr, host_row_of_doc, synonyms,
                                           part_synonyms)
    L.rxhost_hybrid_query_resident.restype = _l
    L.rxhost_hybrid_query_resident.argtypes = [_vp, _vp, _sz, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp,
                                               _vp, _vp, _sz, _vp]
    nf = ftm.nf
    cfg_d = np.array([cfg["k1"], cfg["b"], cfg["summation_ratio"], cfg["full_match_boost"], cfg.get("distance_boost", 1.0),
                      cfg.get("distance_weight", 0.5)], np.float64)
    cfg_i = np.array([cfg["min_rank"], cfg["merge_limit"], {"rx": 0, "classic": 1, "word_count": 2}[cfg.get("bm25_type", "rx")]], np.int32)
    fc = np.stack([np.asarray(cfg[k_], np.float64) for k_ in ("bm25_boost", "bm25_weight", "term_len_boost", "term_len_weight",
                                                             "position_boost", "position_weight")], axis=1).copy()
    ops = np.array([t["op"] for t in terms], np.int32)
    boosts = np.array([t["opts"]["boost"] for t in terms], np.float32)
    tlb = np.array([t["opts"]["term_len_boost"] for t in terms], np.float32)
    fb = np.array([t["opts"]["field_boost"] for t in terms], np.float32).reshape(len(terms), nf).copy()
    ns = np.array([t["opts"]["need_sum_rank"] for t in terms], np.uint8).reshape(len(terms), nf).copy()
    sub_off, wid, pr = [0], [], []
    for t in terms:
        for w, p_ in t["subs"]:
            wid.append(w)
            pr.append(p_)
        sub_off.append(len(wid))
    sub_off, wid, pr = np.array(sub_off, np.uint32), np.array(wid, np.uint32), np.array(pr, np.float32)
    exc = np.ascontiguousarray(excluded, np.uint8) if excluded is not None else None
    hyb = np.array([0 if kind == "rrf" else 1, int(union), int(desc)], np.int32)
    par = np.zeros(5, np.float64)
    par[:len(params)] = params
    keyf = _f32(key)
    hmap = np.ascontiguousarray(host_row_of_doc, np.int32) if host_row_of_doc is not None else None
    cap = int(cfg["merge_limit"]) + int(k)
    oid, orank = np.zeros(cap, np.int32), np.zeros(cap, np.float32)
    tie = C.c_int(0)
    n = L.rxhost_hybrid_query_resident(vmap.h, ftm.h, nf, cfg_d.ctypes.data, cfg_i.ctypes.data, fc.ctypes.data, len(terms), ops.ctypes.data,
                                       boosts.ctypes.data, tlb.ctypes.data, fb.ctypes.data, ns.ctypes.data, sub_off.ctypes.data, wid.ctypes.data,
                                       pr.ctypes.data, exc.ctypes.data if exc is not None else None, hyb.ctypes.data, par.ctypes.data, keyf.ctypes.data, k,
                                       row_of_doc_ptr or None, hmap.ctypes.data if hmap is not None else None, oid.ctypes.data, orank.ctypes.data, cap,
                                       C.byref(tie))
    if n < 0:
        _raise()
    return oid[:n].copy(), orank[:n].copy(), bool(tie.value)


def _hybrid_query_resident_full(vmap, ftm, cfg, terms, key, k, kind, params, union, desc, excluded, row_of_doc_ptr, host_row_of_doc, synonyms, part_synonyms):
    L = lib()
    L.rxhost_hybrid_query_resident_full.restype = _l
    L.rxhost_hybrid_query_resident_full.argtypes = [_vp, _vp, _sz, _vp, _vp, _vp, _sz, _sz] + [_vp] * 10 + [_sz, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp,
                                                                                                        _vp, _vp, _vp, _sz, _vp]
    nf = ftm.nf
    n_part_terms = len(terms)
    terms = list(terms)
    syn_off = [0]
    for syn in synonyms:
        terms.extend(syn)
        syn_off.append(len(terms) - n_part_terms)
    cfg_d = np.array([cfg["k1"], cfg["b"], cfg["summation_ratio"], cfg["full_match_boost"], cfg.get("distance_boost", 1.0),
                      cfg.get("distance_weight", 0.5)], np.float64)
    cfg_i = np.array([cfg["min_rank"], cfg["merge_limit"], {"rx": 0, "classic": 1, "word_count": 2}[cfg.get("bm25_type", "rx")]], np.int32)
    fc = np.stack([np.asarray(cfg[k_], np.float64) for k_ in ("bm25_boost", "bm25_weight", "term_len_boost", "term_len_weight",
                                                             "position_boost", "position_weight")], axis=1).copy()
    ops = np.array([t["op"] for t in terms], np.int32)
    boosts = np.array([t["opts"]["boost"] for t in terms], np.float32)
    tlb = np.array([t["opts"]["term_len_boost"] for t in terms], np.float32)
    fb = np.array([t["opts"]["field_boost"] for t in terms], np.float32).reshape(len(terms), nf).copy()
    ns = np.array([t["opts"]["need_sum_rank"] for t in terms], np.uint8).reshape(len(terms), nf).copy()
    phr = np.array([t.get("phrase", -1) for t in terms], np.int32)
    dst = np.array([t.get("distance", 1) for t in terms], np.int32)
    sub_off, wid, pr = [0], [], []
    for t in terms:
        for w, p_ in t["subs"]:
            wid.append(w)
            pr.append(p_)
        sub_off.append(len(wid))
    sub_off, wid, pr = np.array(sub_off, np.uint32), np.array(wid, np.uint32), np.array(pr, np.float32)
    nparts = sum(1 for i in range(n_part_terms) if phr[i] < 0 or i == 0 or phr[i - 1] != phr[i])
    ps_off, ps = [0], []
    for pi in range(nparts):
        ps.extend(part_synonyms[pi] if part_synonyms and pi < len(part_synonyms) else [])
        ps_off.append(len(ps))
    syn_off_a, ps_off_a, ps_a = np.array(syn_off, np.uint32), np.array(ps_off, np.uint32), np.array(ps + [0], np.uint32)
    exc = np.ascontiguousarray(excluded, np.uint8) if excluded is not None else None
    hyb = np.array([0 if kind == "rrf" else 1, int(union), int(desc)], np.int32)
    par = np.zeros(5, np.float64)
    par[:len(params)] = params
    keyf = _f32(key)
    hmap = np.ascontiguousarray(host_row_of_doc, np.int32) if host_row_of_doc is not None else None
    cap = int(cfg["merge_limit"]) + int(k)
    oid, orank = np.zeros(cap, np.int32), np.zeros(cap, np.float32)
    tie = C.c_int(0)
    n = L.rxhost_hybrid_query_resident_full(vmap.h, ftm.h, nf, cfg_d.ctypes.data, cfg_i.ctypes.data, fc.ctypes.data, n_part_terms, len(terms) - n_part_terms,
                                            ops.ctypes.data, boosts.ctypes.data, tlb.ctypes.data, fb.ctypes.data, ns.ctypes.data, phr.ctypes.data, dst.ctypes.data,
                                            sub_off.ctypes.data, wid.ctypes.data, pr.ctypes.data, len(synonyms), syn_off_a.ctypes.data, nparts,
                                            ps_off_a.ctypes.data, ps_a.ctypes.data, exc.ctypes.data if exc is not None else None, hyb.ctypes.data,
                                            par.ctypes.data, keyf.ctypes.data, k, row_of_doc_ptr or None, hmap.ctypes.data if hmap is not None else None,
                                            oid.ctypes.data, orank.ctypes.data, cap, C.byref(tie))
    if n < 0:
        _raise()
    return oid[:n].copy(), orank[:n].copy(), bool(tie.value)


def ft_unpack(data, array_found_pos: int) -> dict:
    """PositionPostings::AppendPacked on the host (the product's host decoder of a PackedIdRelVec stream): dict(doc, pos_off, fpos)."""
    L = lib()
    L.rxhost_ft_unpack.restype = _l
    L.rxhost_ft_unpack.argtypes = [_vp, _sz, _sz, _vp, _vp, _vp, _vp]
    data = np.ascontiguousarray(data, np.uint8)
    afp = min(int(array_found_pos), 1 << 62)
    npos = _sz(0)
    n = L.rxhost_ft_unpack(data.ctypes.data, data.shape[0], afp, None, None, None, C.byref(npos))
    if n < 0:
        _raise()
    doc, po, fp = np.zeros(n, np.uint32), np.zeros(n + 1, np.uint32), np.zeros(int(npos.value), np.uint64)
    if L.rxhost_ft_unpack(data.ctypes.data, data.shape[0], afp, doc.ctypes.data, po.ctypes.data, fp.ctypes.data, None) != n:
        _raise()
    return dict(doc=doc, pos_off=po, fpos=fp)


def default_ft_config(num_fields=1, **kw):
    """The FTConfig members the merge reads, with the reference's defaults (ftconfig.h:118-124,151-220)."""
    cfg = dict(k1=2.0, b=0.75, summation_ratio=0.0, full_match_boost=1.1, min_rank=5, merge_limit=20000, num_fields=num_fields,
               bm25_boost=[1.0] * num_fields, bm25_weight=[0.1] * num_fields, term_len_boost=[1.0] * num_fields,
               term_len_weight=[0.3] * num_fields, position_boost=[1.0] * num_fields, position_weight=[0.1] * num_fields)
    cfg.update(kw)
    return cfg


def default_ft_opts(num_fields=1, **kw):
    """FtDslOpts defaults (ftdsl.h:13-35)."""
    o = dict(boost=1.0, term_len_boost=1.0, field_boost=[1.0] * num_fields, need_sum_rank=[0] * num_fields)
    o.update(kw)
    return o


def merge_ranked(kind: str, params, knn_ids, knn_ranks, ft_ids, ft_ranks, union=False, desc=True, metric=1, ft_order="id"):
    """Hybrid FT+KNN rank fusion (hybrid_rerank.h). kind: 'rrf' (params=[rank_const]) or 'linear' (params=[kKnn, knnDefault, kFt, ftDefault, c]).
    knn_* best-first as KnnSelectRaw returns them.  ft_order='id': ft_ids ascending with ft_ranks aligned (the selector's ftIds_ view);
    ft_order='rank': the FT result as the merger returns it (best rank first) — the id view is derived inside, no sort needed by the caller."""
    L = lib()
    fn = L.rxhost_merge_ranked if ft_order == "id" else L.rxhost_merge_ranked_ft_order
    fn.restype = _l
    fn.argtypes = [_i, _vp, _i, _i, _i, _vp, _vp, _sz, _vp, _vp, _sz, _vp, _vp, _sz]
    p = np.ascontiguousarray(params, np.float64)
    ki, kr = np.ascontiguousarray(knn_ids, np.int32), _f32(knn_ranks)
    fi, fr = np.ascontiguousarray(ft_ids, np.int32), _f32(ft_ranks)
    cap = ki.shape[0] + fi.shape[0] + 1
    oi, orr = np.empty(cap, np.int32), np.empty(cap, np.float32)
    n = fn(0 if kind == "rrf" else 1, p.ctypes.data, int(union), int(desc), metric, ki.ctypes.data, kr.ctypes.data, ki.shape[0],
           fi.ctypes.data, fr.ctypes.data, fi.shape[0], oi.ctypes.data, orr.ctypes.data, cap)
    if n < 0:
        _raise()
    return oi[:n].copy(), orr[:n].copy()


class GpuIvfFlat:
    """rxgpu::host::GpuIvfFlat — IVF-Flat on the GPU engines (the faiss::IndexFlat -> faiss::IndexIVFFlat pair of the reference's IvfIndex).
    Distances follow FAISS: L2 squared distance ascending, inner product / cosine similarity descending; labels are -1 past the last hit."""

    def __init__(self, metric: int, dim: int, nlist: int, device: int = 0):
        L = lib()
        if not hasattr(L, "_ivf_bound"):
            L.rxhost_ivf_create.restype = _vp
            L.rxhost_ivf_create.argtypes = [_i, _sz, _sz, _i]
            L.rxhost_ivf_destroy.argtypes = [_vp]
            L.rxhost_ivf_add.argtypes = [_vp, _vp, _sz, _vp]
            L.rxhost_ivf_train.argtypes = [_vp, _i]
            L.rxhost_ivf_remove.restype = _l
            L.rxhost_ivf_remove.argtypes = [_vp, _vp, _sz]
            L.rxhost_ivf_reset.argtypes = [_vp]
            L.rxhost_ivf_search.argtypes = [_vp, _vp, _sz, _sz, _vp, _vp]
            L.rxhost_ivf_search_batch.argtypes = [_vp, _sz, _vp, _sz, _sz, _vp, _vp]
            L.rxhost_ivf_range.restype = _l
            L.rxhost_ivf_range.argtypes = [_vp, _vp, _f, _sz, _vp, _vp, _sz]
            L.rxhost_ivf_probed_rows.restype = _l
            L.rxhost_ivf_probed_rows.argtypes = [_vp, _vp, _sz, _vp, _sz]
            L.rxhost_ivf_info.argtypes = [_vp, _vp]
            L.rxhost_ivf_list_sizes.argtypes = [_vp, _vp]
            L.rxhost_ivf_centroids.argtypes = [_vp, _vp]
            L._ivf_bound = True
        self.dim, self.metric, self.nlist = dim, metric, nlist
        self.h = L.rxhost_ivf_create(metric, dim, nlist, device)
        if not self.h:
            _raise()

    def close(self):
        if getattr(self, "h", None):
            lib().rxhost_ivf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _info(self):
        info = np.zeros(4, np.int64)
        lib().rxhost_ivf_info(self.h, info.ctypes.data)
        return info

    ntotal = property(lambda self: int(self._info()[0]))
    is_trained = property(lambda self: bool(self._info()[1]))

    def add_with_ids(self, x, ids):
        x = _f32(x).reshape(-1, self.dim)
        ids = np.ascontiguousarray(ids, np.int64).reshape(-1)
        assert ids.shape[0] == x.shape[0]
        rc = lib().rxhost_ivf_add(self.h, x.ctypes.data, x.shape[0], ids.ctypes.data)
        if rc:
            _raise(rc)

    def train(self, seed: int = 1234):
        rc = lib().rxhost_ivf_train(self.h, seed)
        if rc:
            _raise(rc)

    def remove_ids(self, ids) -> int:
        ids = np.ascontiguousarray(ids, np.int64).reshape(-1)
        n = lib().rxhost_ivf_remove(self.h, ids.ctypes.data, ids.shape[0])
        if n < 0:
            _raise()
        return int(n)

    def reset(self):
        rc = lib().rxhost_ivf_reset(self.h)
        if rc:
            _raise(rc)

    def search(self, x, k: int, nprobe: int = 1):
        x = _f32(x)
        d, l = np.empty(max(k, 1), np.float32), np.empty(max(k, 1), np.int64)
        rc = lib().rxhost_ivf_search(self.h, x.ctypes.data, k, nprobe, d.ctypes.data, l.ctypes.data)
        if rc:
            _raise(rc)
        return d[:k], l[:k]

    def search_batch(self, x, k: int, nprobe: int = 1):
        """x [n][dim] -> (distances [n][k], labels [n][k]): GpuIvfFlat::SearchBatch, the queries side by side on a few streams."""
        x = _f32(x).reshape(-1, self.dim)
        n = x.shape[0]
        d, l = np.empty((n, max(k, 1)), np.float32), np.empty((n, max(k, 1)), np.int64)
        rc = lib().rxhost_ivf_search_batch(self.h, n, x.ctypes.data, k, nprobe, d.ctypes.data, l.ctypes.data)
        if rc:
            _raise(rc)
        return d[:, :k], l[:, :k]

    def range_search(self, x, radius: float, nprobe: int = 1, cap: int = 1 << 16):
        x = _f32(x)
        while True:
            d, l = np.empty(cap, np.float32), np.empty(cap, np.int64)
            n = lib().rxhost_ivf_range(self.h, x.ctypes.data, radius, nprobe, d.ctypes.data, l.ctypes.data, cap)
            if n < 0:
                _raise()
            if n <= cap:
                return d[:n].copy(), l[:n].copy()
            cap = int(n)

    def probed_rows(self, x, nprobe: int):
        x = _f32(x)
        cap = max(self.ntotal, 1)
        out = np.empty(cap, np.uint32)
        n = lib().rxhost_ivf_probed_rows(self.h, x.ctypes.data, nprobe, out.ctypes.data, cap)
        if n < 0:
            _raise()
        return out[:n].copy()

    def list_sizes(self):
        out = np.zeros(self.nlist, np.uint32)
        rc = lib().rxhost_ivf_list_sizes(self.h, out.ctypes.data)
        if rc:
            _raise(rc)
        return out

    def list_ids(self, lst: int):
        L = lib()
        L.rxhost_ivf_list_ids.argtypes = [_vp, _sz, _vp]
        out = np.zeros(int(self.list_sizes()[lst]), np.int64)
        if out.size:
            rc = L.rxhost_ivf_list_ids(self.h, lst, out.ctypes.data)
            if rc:
                _raise(rc)
        return out

    def centroids(self):
        out = np.zeros((self.nlist, self.dim), np.float32)
        rc = lib().rxhost_ivf_centroids(self.h, out.ctypes.data)
        if rc:
            _raise(rc)
        return out


def sq8_find_nth_min_max(values, data_size: int, quantile: float):
    """Sq8FindNthMinMax (FindNthMinMax, quantization_params.h:12-44) -> (min, max)"""
    L = lib()
    L.rxhost_sq8_find_nth_min_max.argtypes = [_vp, _sz, _sz, _f, _vp]
    v = _f32(values).reshape(-1)
    out = np.zeros(2, np.float32)
    L.rxhost_sq8_find_nth_min_max(v.ctypes.data, v.shape[0], data_size, float(quantile), out.ctypes.data)
    return float(out[0]), float(out[1])


def sq8_sample_indexes(sample_size: int, size: int):
    """Sq8SampleIndexes (HNSWView::GetSampleIndexes): consumes the C library's rand() like the reference"""
    L = lib()
    L.rxhost_sq8_sample_indexes.restype = _sz
    L.rxhost_sq8_sample_indexes.argtypes = [_sz, _sz, _vp]
    out = np.zeros(min(sample_size, size), np.uint32)
    n = L.rxhost_sq8_sample_indexes(sample_size, size, out.ctypes.data)
    return out[:n]


def sq8_sample_params(rows, sample_size: int = 20000, quantile: float = 0.0):
    """Sq8SampleParams (QuantizingParams(hnsw, config)) over plain rows -> (minQ, maxQ, alpha, alpha_2, delta)"""
    L = lib()
    L.rxhost_sq8_sample_params.argtypes = [_vp, _sz, _sz, _sz, _f, _vp]
    r = _f32(rows)
    p = np.zeros(5, np.float32)
    rc = L.rxhost_sq8_sample_params(r.ctypes.data, r.shape[0], r.shape[1], sample_size, float(quantile), p.ctypes.data)
    if rc:
        _raise(rc)
    return p
