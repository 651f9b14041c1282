"""TEST INFRASTRUCTURE ONLY.  ctypes access to the CPU checkers:

    liboracle.so              our plain-C restatement of the reference algorithms (oracle/*.c)
    _ref/libref_oracle.so     the real reference engines compiled in place from /root/reference (may be absent)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  Nothing under
reindexer_amd/ may.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ORACLE_SO = HERE / "liboracle.so"
REF_SO = HERE / "_ref" / "libref_oracle.so"
REFERENCE_TREE = Path("/root/reference/cpp_src")

_vp, _sz, _u64, _f, _i = C.c_void_p, C.c_size_t, C.c_uint64, C.c_float, C.c_int

# The reference picks its SIMD level at static-init time (tools/cpucheck.cc:201-231); parity is pinned to AVX-512.
os.environ.setdefault("RX_TARGET_INSTRUCTIONS", "avx512")


def build_oracle(force: bool = False) -> Path:
    srcs = list(HERE.glob("*.c")) + list(HERE.glob("*.h"))
    if force or not ORACLE_SO.exists() or any(s.stat().st_mtime > ORACLE_SO.stat().st_mtime for s in srcs):
        subprocess.run(["make", "-C", str(HERE), "oracle"], check=True, capture_output=True)
    return ORACLE_SO


def build_ref() -> Path | None:
    """Builds oracle/_ref from the reference tree when it is present (this container only)."""
    if REFERENCE_TREE.exists():
        subprocess.run(["make", "-C", str(HERE), "-j8", "ref"], check=True, capture_output=True)
    return REF_SO if REF_SO.exists() else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Oracle:
    """liboracle.so — restatement."""

    def __init__(self):
        build_oracle()
        L = self.L = C.CDLL(str(ORACLE_SO))
        L.orc_l2sqr.restype = _f
        L.orc_l2sqr.argtypes = [_vp, _vp, _sz]
        L.orc_ip.restype = _f
        L.orc_ip.argtypes = [_vp, _vp, _sz]
        L.orc_l2sqr_many.argtypes = [_vp, _vp, _sz, _sz, _vp]
        L.orc_ip_many.argtypes = [_vp, _vp, _sz, _sz, _vp]
        L.orc_l2_module.restype = _f
        L.orc_l2_module.argtypes = [_vp, C.c_int32]
        L.orc_normalize_copy.restype = _f
        L.orc_normalize_copy.argtypes = [_vp, C.c_int32, _vp]
        L.orc_dist.restype = _f
        L.orc_dist.argtypes = [_i, _vp, _vp, _sz, _f]
        L.orc_bf_search_knn.restype = _sz
        L.orc_bf_search_knn.argtypes = [_i, _vp, _vp, _vp, _sz, _sz, _vp, _sz, _vp, _vp]
        L.orc_bf_search_range.restype = _sz
        L.orc_bf_search_range.argtypes = [_i, _vp, _vp, _vp, _sz, _sz, _vp, _f, _vp, _vp, _sz]
        L.orc_bf_search_knn_batch.argtypes = [_i, _vp, _vp, _vp, _sz, _sz, _vp, _sz, _sz, _vp, _vp, _vp, _i]
        L.orc_select_postprocess.restype = _sz
        L.orc_select_postprocess.argtypes = [_i, _vp, _vp, _sz, _i, _i, _i, _sz, _i, _vp, _vp]

    def l2sqr(self, a, b):
        a, b = _f32(a), _f32(b)
        return np.float32(self.L.orc_l2sqr(a.ctypes.data, b.ctypes.data, a.shape[0]))

    def ip(self, a, b):
        a, b = _f32(a), _f32(b)
        return np.float32(self.L.orc_ip(a.ctypes.data, b.ctypes.data, a.shape[0]))

    def dist_many(self, metric, q, rows, inv_norms=None):
        q, rows = _f32(q), _f32(rows)
        n, d = rows.shape
        out = np.empty(n, np.float32)
        if metric == 0:
            self.L.orc_l2sqr_many(q.ctypes.data, rows.ctypes.data, n, d, out.ctypes.data)
            return out
        self.L.orc_ip_many(q.ctypes.data, rows.ctypes.data, n, d, out.ctypes.data)
        out = -(out + np.float32(0.0))
        if metric == 2:
            out = out * _f32(inv_norms)
        return out.astype(np.float32)

    def l2_module(self, x):
        x = _f32(x)
        return np.float32(self.L.orc_l2_module(x.ctypes.data, x.shape[0]))

    def l2_modules(self, rows):
        rows = _f32(rows)
        return np.array([self.L.orc_l2_module(rows[i].ctypes.data, rows.shape[1]) for i in range(rows.shape[0])], np.float32)

    def normalize_copy(self, x):
        x = _f32(x)
        out = np.empty_like(x)
        k = self.L.orc_normalize_copy(x.ctypes.data, x.shape[0], out.ctypes.data)
        return out, np.float32(k)

    def bf_search_knn(self, metric, rows, labels, inv_norms, q, k):
        rows, q = _f32(rows), _f32(q)
        labels = np.ascontiguousarray(labels, np.uint64)
        n, d = rows.shape
        inv = _f32(inv_norms) if inv_norms is not None else None
        od, ol = np.empty(max(k, 1), np.float32), np.empty(max(k, 1), np.uint64)
        c = self.L.orc_bf_search_knn(metric, rows.ctypes.data, labels.ctypes.data, inv.ctypes.data if inv is not None else None,
                                     n, d, q.ctypes.data, k, od.ctypes.data, ol.ctypes.data)
        return od[:c].copy(), ol[:c].copy()

    def bf_search_range(self, metric, rows, labels, inv_norms, q, radius, cap=1 << 20):
        rows, q = _f32(rows), _f32(q)
        labels = np.ascontiguousarray(labels, np.uint64)
        n, d = rows.shape
        inv = _f32(inv_norms) if inv_norms is not None else None
        od, ol = np.empty(cap, np.float32), np.empty(cap, np.uint64)
        c = self.L.orc_bf_search_range(metric, rows.ctypes.data, labels.ctypes.data, inv.ctypes.data if inv is not None else None,
                                       n, d, q.ctypes.data, radius, od.ctypes.data, ol.ctypes.data, cap)
        assert c <= cap
        return od[:c].copy(), ol[:c].copy()

    def bf_search_knn_batch(self, metric, rows, labels, inv_norms, queries, k, threads=1):
        rows, queries = _f32(rows), _f32(queries)
        labels = np.ascontiguousarray(labels, np.uint64)
        n, d = rows.shape
        nq = queries.shape[0]
        inv = _f32(inv_norms) if inv_norms is not None else None
        od, ol = np.empty((nq, k), np.float32), np.empty((nq, k), np.uint64)
        cnt = np.zeros(nq, np.uint64)
        self.L.orc_bf_search_knn_batch(metric, rows.ctypes.data, labels.ctypes.data, inv.ctypes.data if inv is not None else None,
                                       n, d, queries.ctypes.data, nq, k, od.ctypes.data, ol.ctypes.data, cnt.ctypes.data, threads)
        return od, ol, cnt

    def select_postprocess(self, metric, dist, label, need_sort=True, is_array=False, k=None, has_radius=False):
        dist = _f32(dist)
        label = np.ascontiguousarray(label, np.uint64)
        n = dist.shape[0]
        ids, ranks = np.empty(max(n, 1), np.int32), np.empty(max(n, 1), np.float32)
        c = self.L.orc_select_postprocess(metric, dist.ctypes.data, label.ctypes.data, n, int(need_sort), int(is_array),
                                          int(k is not None), k or 0, int(has_radius), ids.ctypes.data, ranks.ctypes.data)
        return ids[:c].copy(), ranks[:c].copy()


class Ref:
    """_ref/libref_oracle.so — the real reference engines (oracle/ref/ref_shim.cc)."""

    def __init__(self):
        if not REF_SO.exists():
            raise FileNotFoundError(REF_SO)
        L = self.L = C.CDLL(str(REF_SO))
        L.ref_simd_level.restype = _i
        L.ref_l2sqr.restype = _f
        L.ref_l2sqr.argtypes = [_vp, _vp, _sz]
        L.ref_ip.restype = _f
        L.ref_ip.argtypes = [_vp, _vp, _sz]
        L.ref_l2sqr_many.argtypes = [_vp, _vp, _sz, _sz, _vp]
        L.ref_ip_many.argtypes = [_vp, _vp, _sz, _sz, _vp]
        L.ref_l2_module.restype = _f
        L.ref_l2_module.argtypes = [_vp, C.c_int32]
        L.ref_normalize_copy.restype = _f
        L.ref_normalize_copy.argtypes = [_vp, C.c_int32, _vp]
        L.ref_bf_create.restype = _vp
        L.ref_bf_create.argtypes = [_i, _sz, _sz]
        L.ref_bf_destroy.argtypes = [_vp]
        L.ref_bf_add_many.argtypes = [_vp, _vp, _sz, _sz, _vp]
        L.ref_bf_remove.argtypes = [_vp, _u64]
        L.ref_bf_count.restype = _sz
        L.ref_bf_count.argtypes = [_vp]
        L.ref_bf_search_knn.restype = _sz
        L.ref_bf_search_knn.argtypes = [_vp, _vp, _sz, _vp, _vp]
        L.ref_bf_search_range.restype = _sz
        L.ref_bf_search_range.argtypes = [_vp, _vp, _f, _vp, _vp, _sz]
        L.ref_hnsw_create.restype = _vp
        L.ref_hnsw_create.argtypes = [_i, _sz, _sz, _sz, _sz]
        L.ref_hnsw_destroy.argtypes = [_vp]
        L.ref_hnsw_add_many.argtypes = [_vp, _vp, _sz, _sz, _vp]
        L.ref_hnsw_mark_delete.argtypes = [_vp, _u64]
        L.ref_hnsw_count.restype = _sz
        L.ref_hnsw_count.argtypes = [_vp]
        L.ref_hnsw_search_knn.restype = _sz
        L.ref_hnsw_search_knn.argtypes = [_vp, _vp, _sz, _sz, _vp, _vp]
        L.ref_hnsw_stream_begin.restype = _vp
        L.ref_hnsw_stream_begin.argtypes = [_vp, _vp, _sz, _sz]
        L.ref_hnsw_stream_continue.restype = _sz
        L.ref_hnsw_stream_continue.argtypes = [_vp, _vp, _sz, _vp, _vp, _vp]
        L.ref_hnsw_stream_end.argtypes = [_vp]
        L.ref_hnsw_search_range.restype = _sz
        L.ref_hnsw_search_range.argtypes = [_vp, _vp, _f, _sz, _vp, _vp, _sz]
        L.ref_hnsw_info.argtypes = [_vp, _vp]
        L.ref_hnsw_export_level0.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp]
        L.ref_hnsw_export_upper.restype = _sz
        L.ref_hnsw_export_upper.argtypes = [_vp, _vp, _vp]
        if hasattr(L, "ref_bf_search_knn_mt"):   # timed multi-thread baselines + graph import (bench.py legs)
            L.ref_bf_search_knn_mt.restype = C.c_double
            L.ref_bf_search_knn_mt.argtypes = [_vp, _vp, _sz, _sz, _sz, _sz, _sz, C.c_double, _vp, _vp]
            L.ref_hnsw_search_knn_mt.restype = C.c_double
            L.ref_hnsw_search_knn_mt.argtypes = [_vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, C.c_double, _vp, _vp]
            L.ref_hnsw_search_knn_many.argtypes = [_vp, _vp, _sz, _sz, _sz, _sz, _vp, _vp, _vp]
            L.ref_hnsw_import_graph.restype = _i
            L.ref_hnsw_import_graph.argtypes = [_vp, _sz, _i, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
            L.ref_last_error.restype = C.c_char_p

    @property
    def simd_level(self) -> int:
        return self.L.ref_simd_level()

    def dist_many(self, metric, q, rows):
        q, rows = _f32(q), _f32(rows)
        n, d = rows.shape
        out = np.empty(n, np.float32)
        (self.L.ref_l2sqr_many if metric == 0 else self.L.ref_ip_many)(q.ctypes.data, rows.ctypes.data, n, d, out.ctypes.data)
        return out

    def l2_module(self, x):
        x = _f32(x)
        return np.float32(self.L.ref_l2_module(x.ctypes.data, x.shape[0]))

    def normalize_copy(self, x):
        x = _f32(x)
        out = np.empty_like(x)
        k = self.L.ref_normalize_copy(x.ctypes.data, x.shape[0], out.ctypes.data)
        return out, np.float32(k)


class RefBruteforce:
    """hnswlib::BruteforceSearch of the reference."""

    def __init__(self, ref: Ref, metric: int, dim: int, capacity: int):
        self.ref, self.dim = ref, dim
        self.h = ref.L.ref_bf_create(metric, dim, capacity)
        assert self.h

    def close(self):
        if self.h:
            self.ref.L.ref_bf_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def add(self, vecs, labels):
        vecs = _f32(vecs).reshape(-1, self.dim)
        labels = np.ascontiguousarray(labels, np.uint64)
        assert self.ref.L.ref_bf_add_many(self.h, vecs.ctypes.data, vecs.shape[0], self.dim, labels.ctypes.data) == 0

    def remove(self, label):
        self.ref.L.ref_bf_remove(self.h, int(label))

    @property
    def count(self):
        return self.ref.L.ref_bf_count(self.h)

    def search_knn(self, q, k):
        q = _f32(q)
        od, ol = np.empty(max(k, 1), np.float32), np.empty(max(k, 1), np.uint64)
        c = self.ref.L.ref_bf_search_knn(self.h, q.ctypes.data, k, od.ctypes.data, ol.ctypes.data)
        return od[:c].copy(), ol[:c].copy()

    def search_range(self, q, radius, cap=1 << 20):
        q = _f32(q)
        od, ol = np.empty(cap, np.float32), np.empty(cap, np.uint64)
        c = self.ref.L.ref_bf_search_range(self.h, q.ctypes.data, radius, od.ctypes.data, ol.ctypes.data, cap)
        assert c <= cap
        return od[:c].copy(), ol[:c].copy()

    def search_knn_mt(self, queries, k, threads, per_thread, deadline_s=0.0):
        """T threads x per_thread searches over the shared index, clock from the common start flag to the last finish (threads are
        created before the clock starts).  Returns (seconds, searches completed)."""
        queries = _f32(queries).reshape(-1, self.dim)
        done, chk = C.c_size_t(0), C.c_uint64(0)
        s = self.ref.L.ref_bf_search_knn_mt(self.h, queries.ctypes.data, queries.shape[0], self.dim, k, threads, per_thread, float(deadline_s),
                                            C.byref(done), C.byref(chk))
        return float(s), int(done.value)


def ref_or_none() -> Ref | None:
    try:
        return Ref()
    except (FileNotFoundError, OSError):
        return None


class RefHnsw:
    """hnswlib::HierarchicalNSWImpl<float, None> of the reference (seed 100, ReplaceDeleted_True — hnsw.cc:74-78)."""

    def __init__(self, ref: Ref, metric: int, dim: int, capacity: int, M: int = 16, ef_construction: int = 200, _handle=None):
        self.ref, self.dim, self.metric = ref, dim, metric
        self.h = _handle if _handle is not None else ref.L.ref_hnsw_create(metric, dim, capacity, M, ef_construction)
        assert self.h

    def save_index(self) -> bytes:
        """HierarchicalNSW::SaveIndex behind the float flag (hnsw.cc:56-62, hnswalg.h:1213-1263) through an in-memory IWriter."""
        L = self.ref.L
        L.ref_hnsw_save_index.restype = C.c_long
        L.ref_hnsw_save_index.argtypes = [_vp, _vp, _sz]
        n = L.ref_hnsw_save_index(self.h, None, 0)
        if n < 0:
            raise RuntimeError(L.ref_last_error().decode())
        buf = np.empty(max(n, 1), np.uint8)
        assert L.ref_hnsw_save_index(self.h, buf.ctypes.data, n) == n
        return buf[:n].tobytes()

    @classmethod
    def load_index(cls, ref: Ref, data: bytes, metric: int, dim: int, labels, vectors) -> "RefHnsw":
        """The reference's reader constructor (hnswalg.h:297-409) on an in-memory IReader whose primary keys resolve to (labels, vectors)."""
        L = ref.L
        L.ref_hnsw_load_index.restype = _vp
        L.ref_hnsw_load_index.argtypes = [_vp, _sz, _i, _sz, _vp, _vp, _sz]
        raw = np.frombuffer(data, np.uint8)
        labels = np.ascontiguousarray(labels, np.uint64).reshape(-1)
        vectors = _f32(vectors).reshape(-1, dim)
        h = L.ref_hnsw_load_index(raw.ctypes.data if raw.size else None, raw.size, metric, dim, labels.ctypes.data, vectors.ctypes.data, labels.shape[0])
        if not h:
            raise RuntimeError(L.ref_last_error().decode())
        return cls(ref, metric, dim, 0, _handle=h)

    def close(self):
        if self.h:
            self.ref.L.ref_hnsw_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def add(self, vecs, labels):
        vecs = _f32(vecs).reshape(-1, self.dim)
        labels = np.ascontiguousarray(labels, np.uint64)
        assert self.ref.L.ref_hnsw_add_many(self.h, vecs.ctypes.data, vecs.shape[0], self.dim, labels.ctypes.data) == 0

    def mark_delete(self, label):
        assert self.ref.L.ref_hnsw_mark_delete(self.h, int(label)) == 0

    @property
    def count(self):
        return self.ref.L.ref_hnsw_count(self.h)

    def search_knn(self, q, k, ef=0):
        q = _f32(q)
        od, ol = np.empty(max(k, 1), np.float32), np.empty(max(k, 1), np.uint64)
        c = self.ref.L.ref_hnsw_search_knn(self.h, q.ctypes.data, k, ef, od.ctypes.data, ol.ctypes.data)
        return od[:c].copy(), ol[:c].copy()

    def search_range(self, q, radius, ef, cap=1 << 20):
        q = _f32(q)
        od, ol = np.empty(cap, np.float32), np.empty(cap, np.uint64)
        c = self.ref.L.ref_hnsw_search_range(self.h, q.ctypes.data, radius, ef, od.ctypes.data, ol.ctypes.data, cap)
        assert c <= cap
        return od[:c].copy(), ol[:c].copy()

    def stream(self, q, ef=0):
        """BeginStreamingSearch: returns a session object with .next(batch) -> (dist, label, exhausted) and .close()."""
        return _RefStream(self, _f32(q), ef)

    def search_knn_many(self, queries, k, ef=0):
        queries = _f32(queries).reshape(-1, self.dim)
        nq = queries.shape[0]
        od, ol, cnt = np.zeros((nq, k), np.float32), np.zeros((nq, k), np.uint64), np.zeros(nq, np.uint32)
        self.ref.L.ref_hnsw_search_knn_many(self.h, queries.ctypes.data, nq, self.dim, k, ef, od.ctypes.data, ol.ctypes.data, cnt.ctypes.data)
        return od, ol, cnt

    def search_knn_mt(self, queries, k, ef, threads, per_thread, deadline_s=0.0):
        """Timed like RefBruteforce.search_knn_mt.  Returns (seconds, searches completed)."""
        queries = _f32(queries).reshape(-1, self.dim)
        done, chk = C.c_size_t(0), C.c_uint64(0)
        s = self.ref.L.ref_hnsw_search_knn_mt(self.h, queries.ctypes.data, queries.shape[0], self.dim, k, ef, threads, per_thread,
                                              float(deadline_s), C.byref(done), C.byref(chk))
        return float(s), int(done.value)

    def import_graph(self, g: dict, vectors=None):
        """Writes a flat graph (the export() layout; e.g. one built by the product's HnswGraph) INTO this (empty) engine, so that the
        reference's own SearchKnn runs on it."""
        vec = _f32(g["vectors"] if vectors is None else vectors)
        n = int(g["n"])
        links0 = np.ascontiguousarray(g["links0"], np.uint32)
        levels = np.ascontiguousarray(g["levels"], np.int32)
        labels = np.ascontiguousarray(g["labels"], np.uint64)
        deleted = np.ascontiguousarray(g["deleted"], np.uint8)
        upper_off = np.ascontiguousarray(g["upper_off"], np.uint64)
        upper = np.ascontiguousarray(g["upper"], np.uint32)
        assert links0.shape[0] >= n and vec.shape[0] >= n
        rc = self.ref.L.ref_hnsw_import_graph(self.h, n, int(g["maxlevel"]), int(g["entry"]) & 0xFFFFFFFF, links0.ctypes.data, levels.ctypes.data,
                                              labels.ctypes.data, deleted.ctypes.data, vec.ctypes.data, upper_off.ctypes.data, upper.ctypes.data)
        if rc != 0:
            raise RuntimeError(self.ref.L.ref_last_error().decode())

    def export(self, with_vectors=True) -> dict:
        """Flat graph in the layout shared by the oracle restatement and the GPU engine."""
        info = np.zeros(6, np.int64)
        self.ref.L.ref_hnsw_info(self.h, info.ctypes.data)
        n, M, maxM0, maxlevel, entry, ndel = (int(x) for x in info)
        links0 = np.zeros((n, 1 + maxM0), np.uint32)
        levels = np.zeros(n, np.int32)
        labels = np.zeros(n, np.uint64)
        deleted = np.zeros(n, np.uint8)
        vectors = np.zeros((n, self.dim), np.float32) if with_vectors else None
        self.ref.L.ref_hnsw_export_level0(self.h, links0.ctypes.data, levels.ctypes.data, labels.ctypes.data, deleted.ctypes.data,
                                          vectors.ctypes.data if with_vectors else None)
        upper_off = np.zeros(n + 1, np.uint64)
        blocks = self.ref.L.ref_hnsw_export_upper(self.h, upper_off.ctypes.data, None)
        upper = np.zeros((max(blocks, 1), 1 + M), np.uint32)
        self.ref.L.ref_hnsw_export_upper(self.h, upper_off.ctypes.data, upper.ctypes.data)
        return dict(metric=self.metric, n=n, dim=self.dim, M=M, maxM0=maxM0, maxlevel=maxlevel, entry=entry, num_deleted=ndel,
                    links0=links0, upper_off=upper_off, upper=upper, levels=levels, labels=labels, deleted=deleted, vectors=vectors)


def ref_hnsw_build_mt(ref: "Ref", metric: int, vecs, labels, M: int = 16, ef_construction: int = 200, threads: int = 16) -> dict:
    """The reference's MULTITHREADED index build: HierarchicalNSW<Synchronization::OnInsertions> filled through AddPointConcurrent from
    `threads` inserting threads (hnsw_index.cc:19, 105-111), exported as the flat graph RefHnsw.export returns."""
    L = ref.L
    L.ref_hnswmt_build.restype = _vp
    L.ref_hnswmt_build.argtypes = [_i, _sz, _sz, _sz, _sz, _sz, _vp, _vp]
    L.ref_hnswmt_destroy.argtypes = [_vp]
    L.ref_hnswmt_info.argtypes = [_vp, _vp]
    L.ref_hnswmt_export_level0.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp]
    L.ref_hnswmt_export_upper.restype = _sz
    L.ref_hnswmt_export_upper.argtypes = [_vp, _vp, _vp]
    vecs = _f32(vecs)
    n, dim = vecs.shape
    labels = np.ascontiguousarray(labels, np.uint64)
    h = L.ref_hnswmt_build(metric, dim, n, M, ef_construction, threads, vecs.ctypes.data, labels.ctypes.data)
    if not h:
        raise RuntimeError(L.ref_last_error().decode())
    try:
        info = np.zeros(6, np.int64)
        L.ref_hnswmt_info(h, info.ctypes.data)
        n, M, maxM0, maxlevel, entry, ndel = (int(x) for x in info)
        links0 = np.zeros((n, 1 + maxM0), np.uint32)
        levels = np.zeros(n, np.int32)
        out_labels = np.zeros(n, np.uint64)
        deleted = np.zeros(n, np.uint8)
        vectors = np.zeros((n, dim), np.float32)
        L.ref_hnswmt_export_level0(h, links0.ctypes.data, levels.ctypes.data, out_labels.ctypes.data, deleted.ctypes.data, vectors.ctypes.data)
        upper_off = np.zeros(n + 1, np.uint64)
        blocks = L.ref_hnswmt_export_upper(h, upper_off.ctypes.data, None)
        upper = np.zeros((max(blocks, 1), 1 + M), np.uint32)
        L.ref_hnswmt_export_upper(h, upper_off.ctypes.data, upper.ctypes.data)
    finally:
        L.ref_hnswmt_destroy(h)
    return dict(metric=metric, n=n, dim=dim, M=M, maxM0=maxM0, maxlevel=maxlevel, entry=entry, num_deleted=ndel, links0=links0, upper_off=upper_off,
                upper=upper, levels=levels, labels=out_labels, deleted=deleted, vectors=vectors)


def _hnsw_bind(L):
    L.orc_hnsw_search_knn.restype = _sz
    L.orc_hnsw_search_knn.argtypes = [_i, _sz, _sz, _sz, _sz, _i, C.c_uint32, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _sz, _vp, _vp]
    L.orc_hnsw_last_stats.argtypes = [_vp, _vp]


class _RefStream:
    def __init__(self, owner, q, ef):
        self.owner, self.q = owner, q
        self.s = owner.ref.L.ref_hnsw_stream_begin(owner.h, q.ctypes.data, owner.dim, ef)

    def next(self, batch):
        od, ol = np.empty(max(batch, 1), np.float32), np.empty(max(batch, 1), np.uint64)
        ex = C.c_int(0)
        n = self.owner.ref.L.ref_hnsw_stream_continue(self.owner.h, self.s, batch, od.ctypes.data, ol.ctypes.data, C.byref(ex))
        return od[:n].copy(), ol[:n].copy(), bool(ex.value)

    def close(self):
        if self.s:
            self.owner.ref.L.ref_hnsw_stream_end(self.s)
            self.s = None


class OracleHnswStream:
    """Restated BeginStreamingSearch / ContinueStreamingSearch on a flat graph (oracle/oracle_hnsw.c)."""

    def __init__(self, orc: Oracle, g: dict, q, ef: int = 0, inv_norms=None):
        L = self.L = orc.L
        L.orc_hnsw_stream_begin.restype = _vp
        L.orc_hnsw_stream_begin.argtypes = [_i, _sz, _sz, _sz, _sz, _i, C.c_uint32, _sz] + [_vp] * 9 + [_sz]
        L.orc_hnsw_stream_continue.restype = _sz
        L.orc_hnsw_stream_continue.argtypes = [_vp, _sz, _vp, _vp, _vp]
        L.orc_hnsw_stream_end.argtypes = [_vp]
        self.keep = (g, _f32(q), _f32(inv_norms) if inv_norms is not None else None)
        q, inv = self.keep[1], self.keep[2]
        self.s = L.orc_hnsw_stream_begin(g["metric"], g["n"], g["dim"], g["M"], g["maxM0"], g["maxlevel"], g["entry"], g["num_deleted"],
                                         g["links0"].ctypes.data, g["upper_off"].ctypes.data, g["upper"].ctypes.data, g["levels"].ctypes.data,
                                         g["labels"].ctypes.data, g["deleted"].ctypes.data, g["vectors"].ctypes.data,
                                         inv.ctypes.data if inv is not None else None, q.ctypes.data, ef)

    def next(self, batch):
        od, ol = np.empty(max(batch, 1), np.float32), np.empty(max(batch, 1), np.uint64)
        ex = C.c_int(0)
        n = self.L.orc_hnsw_stream_continue(self.s, batch, od.ctypes.data, ol.ctypes.data, C.byref(ex))
        return od[:n].copy(), ol[:n].copy(), bool(ex.value)

    def close(self):
        if self.s:
            self.L.orc_hnsw_stream_end(self.s)
            self.s = None


def oracle_hnsw_search_knn(orc: Oracle, g: dict, q, k: int, ef: int = 0, inv_norms=None, with_stats=False):
    """Restated HierarchicalNSWImpl::SearchKnn on a flat graph (oracle/oracle_hnsw.c)."""
    if not getattr(orc, "_hnsw_bound", False):
        _hnsw_bind(orc.L)
        orc._hnsw_bound = True
    q = _f32(q)
    od, ol = np.empty(max(k, 1), np.float32), np.empty(max(k, 1), np.uint64)
    inv = _f32(inv_norms) if inv_norms is not None else None
    c = orc.L.orc_hnsw_search_knn(g["metric"], g["n"], g["dim"], g["M"], g["maxM0"], g["maxlevel"], g["entry"], g["num_deleted"],
                                  g["links0"].ctypes.data, g["upper_off"].ctypes.data, g["upper"].ctypes.data, g["levels"].ctypes.data,
                                  g["labels"].ctypes.data, g["deleted"].ctypes.data, g["vectors"].ctypes.data,
                                  inv.ctypes.data if inv is not None else None, q.ctypes.data, k, ef, od.ctypes.data, ol.ctypes.data)
    if with_stats:
        nd, nh = C.c_long(0), C.c_long(0)
        orc.L.orc_hnsw_last_stats(C.byref(nd), C.byref(nh))
        return od[:c].copy(), ol[:c].copy(), int(nd.value), int(nh.value)
    return od[:c].copy(), ol[:c].copy()


# ------------------------------------------------------------------------------------------------ BM25 (oracle_bm25.c)
class _FtConfig(C.Structure):
    _fields_ = [("k1", C.c_double), ("b", C.c_double), ("summation_ratio", C.c_double), ("full_match_boost", C.c_double),
                ("min_rank", C.c_int), ("merge_limit", C.c_uint32), ("num_fields", C.c_uint32),
                ("bm25_boost", _vp), ("bm25_weight", _vp), ("term_len_boost", _vp), ("term_len_weight", _vp),
                ("position_boost", _vp), ("position_weight", _vp), ("bm25_type", C.c_int)]


class _FtTermOpts(C.Structure):
    _fields_ = [("boost", _f), ("term_len_boost", _f), ("field_boost", _vp), ("need_sum_rank", _vp)]


class _FtPostings(C.Structure):
    _fields_ = [("n", _u64), ("doc", _vp), ("ent_off", _vp), ("ent_field", _vp), ("ent_tf", _vp), ("ent_first_pos", _vp), ("proc", _f)]


class _FtPPostings(C.Structure):
    _fields_ = [("n", _u64), ("doc", _vp), ("pos_off", _vp), ("fpos", _vp), ("proc", _f)]


class _FtTerm(C.Structure):
    _fields_ = [("op", _i), ("opts", _FtTermOpts), ("nsub", C.c_uint32), ("subs", _vp)]


class FtOracle:
    """Restated ft_fast merge (single-term mergeSimple and multi-term mergeTerm).  cfg / opts are plain dicts (defaults = the reference's FTConfig defaults)."""

    def __init__(self, orc: Oracle):
        L = self.L = orc.L
        L.orc_bm25rx_idf.restype = C.c_double
        L.orc_bm25rx_idf.argtypes = [C.c_double, C.c_double]
        L.orc_bm25rx_get.restype = C.c_double
        L.orc_bm25rx_get.argtypes = [C.c_double] * 6
        L.orc_pos2rank.restype = _f
        L.orc_pos2rank.argtypes = [C.c_uint]
        L.orc_bound.restype = _f
        L.orc_bound.argtypes = [_f, _f, _f]
        L.orc_calc_term_rank.restype = _f
        L.orc_calc_term_rank.argtypes = [_vp, _vp, C.c_double, _f, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
        L.orc_ft_merge_query.restype = _sz
        L.orc_ft_merge_query.argtypes = [_vp, C.c_double, C.c_double, _vp, C.c_uint32, _u64, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]
        L.orc_positions_distance.restype = C.c_uint
        L.orc_positions_distance.argtypes = [_vp, C.c_uint32, _vp, C.c_uint32]
        L.orc_ft_merge_simple.restype = _sz
        L.orc_ft_merge_simple.argtypes = [_vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp, C.c_uint32, _i, _vp, _vp, _vp, _vp]

    @staticmethod
    def default_config(num_fields=1, **kw):
        cfg = dict(k1=2.0, b=0.75, summation_ratio=0.0, full_match_boost=1.1, min_rank=5, merge_limit=20000, num_fields=num_fields,
                   bm25_boost=[1.0] * num_fields, bm25_weight=[0.1] * num_fields, term_len_boost=[1.0] * num_fields,
                   term_len_weight=[0.3] * num_fields, position_boost=[1.0] * num_fields, position_weight=[0.1] * num_fields)
        cfg.update(kw)
        return cfg

    @staticmethod
    def default_opts(num_fields=1, **kw):
        o = dict(boost=1.0, term_len_boost=1.0, field_boost=[1.0] * num_fields, need_sum_rank=[0] * num_fields)
        o.update(kw)
        return o

    def _cfg(self, cfg):
        keep = [np.ascontiguousarray(cfg[k], np.float64) for k in ("bm25_boost", "bm25_weight", "term_len_boost", "term_len_weight",
                                                                   "position_boost", "position_weight")]
        c = _FtConfig(cfg["k1"], cfg["b"], cfg["summation_ratio"], cfg["full_match_boost"], cfg["min_rank"], cfg["merge_limit"],
                      cfg["num_fields"], *[a.ctypes.data for a in keep], {"rx": 0, "classic": 1, "word_count": 2}[cfg.get("bm25_type", "rx")])
        return c, keep

    def _opts(self, opts):
        fb = np.ascontiguousarray(opts["field_boost"], np.float32)
        ns = np.ascontiguousarray(opts["need_sum_rank"], np.uint8)
        return _FtTermOpts(opts["boost"], opts["term_len_boost"], fb.ctypes.data, ns.ctypes.data), (fb, ns)

    def idf(self, total_docs, matched):
        return self.L.orc_bm25rx_idf(total_docs, matched)

    def term_rank(self, cfg, opts, idf, proc, ent_field, ent_tf, ent_first_pos, words_in_field, avg_words):
        c, k1 = self._cfg(cfg)
        o, k2 = self._opts(opts)
        ef = np.ascontiguousarray(ent_field, np.uint8)
        et = np.ascontiguousarray(ent_tf, np.uint32)
        ep = np.ascontiguousarray(ent_first_pos, np.uint32)
        w = _f32(words_in_field)
        a = _f32(avg_words)
        field = C.c_uint8(0)
        bn, tl, pr = _f(0), _f(0), _f(0)
        r = self.L.orc_calc_term_rank(C.byref(c), C.byref(o), idf, proc, ef.shape[0], ef.ctypes.data, et.ctypes.data, ep.ctypes.data,
                                      w.ctypes.data, a.ctypes.data, C.byref(field), C.byref(bn), C.byref(tl), C.byref(pr))
        return np.float32(r), int(field.value), np.float32(bn.value), np.float32(tl.value), np.float32(pr.value)

    def merge_simple(self, cfg, opts, total_docs, words, avg_words, removed, excluded, subs, sort_by_rank=True):
        """subs: list of dicts(doc, ent_off, ent_field, ent_tf, ent_first_pos, proc) already sorted by proc desc."""
        c, k1 = self._cfg(cfg)
        o, k2 = self._opts(opts)
        words = _f32(words)
        avg = _f32(avg_words)
        rem = np.ascontiguousarray(removed, np.uint8) if removed is not None else None
        exc = np.ascontiguousarray(excluded, np.uint8) if excluded is not None else None
        arr = (_FtPostings * len(subs))()
        keep = []
        total = 0
        for i, s in enumerate(subs):
            d = np.ascontiguousarray(s["doc"], np.uint32)
            eo = np.ascontiguousarray(s["ent_off"], np.uint32)
            ef = np.ascontiguousarray(s["ent_field"], np.uint8)
            et = np.ascontiguousarray(s["ent_tf"], np.uint32)
            ep = np.ascontiguousarray(s["ent_first_pos"], np.uint32)
            keep += [d, eo, ef, et, ep]
            arr[i] = _FtPostings(d.shape[0], d.ctypes.data, eo.ctypes.data, ef.ctypes.data, et.ctypes.data, ep.ctypes.data, s["proc"])
            total += d.shape[0]
        cap = max(1, min(cfg["merge_limit"], total))
        od, op = np.zeros(cap, np.uint32), np.zeros(cap, np.float32)
        of, on = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8)
        n = self.L.orc_ft_merge_simple(C.byref(c), C.byref(o), total_docs, words.ctypes.data, avg.ctypes.data,
                                       rem.ctypes.data if rem is not None else None, exc.ctypes.data if exc is not None else None,
                                       arr, len(subs), int(sort_by_rank), od.ctypes.data, op.ctypes.data, of.ctypes.data, on.ctypes.data)
        return od[:n].copy(), op[:n].copy(), of[:n].copy(), on[:n].copy()


    def merge_query(self, cfg, terms, total_docs, words, avg_words, removed, excluded, sort_by_rank=True, distance_boost=1.0,
                    distance_weight=0.5):
        """Multi-term Merger::Merge.  terms: list of dict(op (1 OR / 2 AND / 3 NOT), opts, subs=[dict(doc, pos_off, fpos, proc), ...]),
        sub-terms already in SortSubterms order.  Returns (doc, proc, field, norm, preselected)."""
        c, k1 = self._cfg(cfg)
        words = _f32(words)
        avg = _f32(avg_words)
        rem = np.ascontiguousarray(removed, np.uint8) if removed is not None else None
        exc = np.ascontiguousarray(excluded, np.uint8) if excluded is not None else None
        tarr = (_FtTerm * len(terms))()
        keep = []
        total = 0
        for ti, t in enumerate(terms):
            o, k2 = self._opts(t["opts"])
            parr = (_FtPPostings * max(1, len(t["subs"])))()
            for i, s in enumerate(t["subs"]):
                d = np.ascontiguousarray(s["doc"], np.uint32)
                po = np.ascontiguousarray(s["pos_off"], np.uint32)
                fp = np.ascontiguousarray(s["fpos"], np.uint64)
                keep += [d, po, fp]
                parr[i] = _FtPPostings(d.shape[0], d.ctypes.data, po.ctypes.data, fp.ctypes.data, s["proc"])
                total += d.shape[0]
            keep += [k2, parr]
            tarr[ti] = _FtTerm(t["op"], o, len(t["subs"]), C.cast(parr, _vp))
        cap = max(1, min(cfg["merge_limit"], total))
        od, op = np.zeros(cap, np.uint32), np.zeros(cap, np.float32)
        of, on = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8)
        pre = C.c_uint8(0)
        n = self.L.orc_ft_merge_query(C.byref(c), distance_boost, distance_weight, tarr, len(terms), total_docs, words.ctypes.data,
                                      avg.ctypes.data, rem.ctypes.data if rem is not None else None,
                                      exc.ctypes.data if exc is not None else None, int(sort_by_rank), od.ctypes.data, op.ctypes.data,
                                      of.ctypes.data, on.ctypes.data, C.byref(pre))
        return od[:n].copy(), op[:n].copy(), of[:n].copy(), on[:n].copy(), bool(pre.value)

    def positions_distance(self, a, b):
        a = np.ascontiguousarray(a, np.uint64)
        b = np.ascontiguousarray(b, np.uint64)
        return int(self.L.orc_positions_distance(a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0]))


def make_fpos(pos, field, array_idx=0):
    """PosType word (idrelset.h:14-32): pos | arrayIdx << 28 | field << 56."""
    return (np.asarray(pos, np.uint64) | (np.asarray(array_idx, np.uint64) << np.uint64(28)) | (np.asarray(field, np.uint64) << np.uint64(56)))


def positions_to_entries(s):
    """Positions-format sub-term dict -> the flat (field, tf, first pos) entry format of merge_simple / rxgpu_ft_set_word."""
    doc = np.asarray(s["doc"], np.uint32)
    po = np.asarray(s["pos_off"], np.int64)
    fp = np.asarray(s["fpos"], np.uint64)
    fld = (fp >> np.uint64(56)).astype(np.int64)
    owner = np.repeat(np.arange(doc.shape[0]), np.diff(po))
    start = np.ones(fp.shape[0], bool)
    if fp.shape[0] > 1:
        start[1:] = (owner[1:] != owner[:-1]) | (fld[1:] != fld[:-1])
    idx = np.flatnonzero(start)
    ent_field = fld[idx].astype(np.uint8)
    ent_first = (fp[idx] & np.uint64((1 << 28) - 1)).astype(np.uint32)
    ent_tf = np.diff(np.append(idx, fp.shape[0])).astype(np.uint32)
    ent_off = np.zeros(doc.shape[0] + 1, np.uint32)
    np.add.at(ent_off, owner[idx] + 1, 1)
    ent_off = np.cumsum(ent_off).astype(np.uint32)
    out = dict(doc=doc, ent_off=ent_off, ent_field=ent_field, ent_tf=ent_tf, ent_first_pos=ent_first)
    if "proc" in s:
        out["proc"] = s["proc"]
    return out


# ------------------------------------------------------------------------------------------------ the REAL ft_fast merger (_ref)
REF_FT_SO = HERE / "_ref" / "libref_ft.so"


class RefFt:
    """reindexer::ft::Merger<IdRelVec, MergeData, uint32_t>::Merge<Bm25Rx> of the reference (oracle/ref/ref_ft_shim.cc)."""

    OP_OR, OP_AND, OP_NOT = 1, 2, 3

    def __init__(self, num_fields: int):
        if not REF_FT_SO.exists():
            raise FileNotFoundError(REF_FT_SO)
        L = self.L = C.CDLL(str(REF_FT_SO))
        L.ref_ft_create.restype = _vp
        L.ref_ft_create.argtypes = [_sz]
        L.ref_ft_destroy.argtypes = [_vp]
        L.ref_ft_set_docs.argtypes = [_vp, _sz, _vp, _vp, _vp]
        L.ref_ft_set_word.argtypes = [_vp, C.c_uint32, _sz, _vp, _vp, _vp, _vp, _vp]
        L.ref_ft_set_config.argtypes = [_vp, _vp, _vp, _vp]
        L.ref_ft_merge.restype = C.c_long
        L.ref_ft_merge.argtypes = [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz]
        self.nf = num_fields
        self.h = L.ref_ft_create(num_fields)

    def close(self):
        if self.h:
            self.L.ref_ft_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_docs(self, words, avg, removed=None):
        words = _f32(words).reshape(-1, self.nf)
        avg = _f32(avg)
        rem = np.ascontiguousarray(removed, np.uint8) if removed is not None else None
        self.total = words.shape[0]
        self.L.ref_ft_set_docs(self.h, words.shape[0], words.ctypes.data, avg.ctypes.data, rem.ctypes.data if rem is not None else None)

    def set_word_positions(self, word_id, doc, pos_off, pos_field, pos_pos):
        doc = np.ascontiguousarray(doc, np.uint32)
        po, pf, pp = (np.ascontiguousarray(a, np.uint32) for a in (pos_off, pos_field, pos_pos))
        self.L.ref_ft_set_word(self.h, word_id, doc.shape[0], doc.ctypes.data, po.ctypes.data, pf.ctypes.data, pp.ctypes.data, None)

    def set_word_fpos(self, word_id, s):
        """Positions-format sub-term dict (doc, pos_off, fpos)."""
        fp = np.asarray(s["fpos"], np.uint64)
        self.L.ref_ft_set_word.argtypes = [_vp, C.c_uint32, _sz, _vp, _vp, _vp, _vp, _vp]
        doc = np.ascontiguousarray(s["doc"], np.uint32)
        po = np.ascontiguousarray(s["pos_off"], np.uint32)
        pf = (fp >> np.uint64(56)).astype(np.uint32)
        pp = (fp & np.uint64((1 << 28) - 1)).astype(np.uint32)
        pa = ((fp >> np.uint64(28)) & np.uint64((1 << 28) - 1)).astype(np.uint32)
        self.L.ref_ft_set_word(self.h, word_id, doc.shape[0], doc.ctypes.data, po.ctypes.data, pf.ctypes.data, pp.ctypes.data, pa.ctypes.data)

    def pack(self, s):
        """Positions-format sub-term dict -> (PackedIdRelVec bytes, arrayFoundPos) by the reference's own packer."""
        L = self.L
        L.ref_ft_pack.restype = _sz
        L.ref_ft_pack.argtypes = [_sz, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]
        fp = np.asarray(s["fpos"], np.uint64)
        doc = np.ascontiguousarray(s["doc"], np.uint32)
        po = np.ascontiguousarray(s["pos_off"], np.uint32)
        pf = (fp >> np.uint64(56)).astype(np.uint32)
        pp = (fp & np.uint64((1 << 28) - 1)).astype(np.uint32)
        pa = ((fp >> np.uint64(28)) & np.uint64((1 << 28) - 1)).astype(np.uint32)
        cap = 16 * (doc.shape[0] + fp.shape[0]) + 64
        out = np.zeros(cap, np.uint8)
        afp = C.c_uint64(0)
        n = L.ref_ft_pack(doc.shape[0], doc.ctypes.data, po.ctypes.data, pf.ctypes.data, pp.ctypes.data, pa.ctypes.data, out.ctypes.data, cap,
                          C.byref(afp))
        assert n <= cap
        return out[:n].copy(), int(afp.value)

    def set_word_flat(self, word_id, s):
        """Flat sub-term dict (doc, ent_off, ent_field, ent_tf, ent_first_pos) -> positions first_pos, first_pos+1, ... per field."""
        pos_off, pf, pp = [0], [], []
        for i in range(len(s["doc"])):
            for e in range(int(s["ent_off"][i]), int(s["ent_off"][i + 1])):
                for t in range(int(s["ent_tf"][e])):
                    pf.append(int(s["ent_field"][e]))
                    pp.append(int(s["ent_first_pos"][e]) + t)
            pos_off.append(len(pf))
        self.set_word_positions(word_id, s["doc"], pos_off, pf, pp)

    def set_config(self, cfg: dict, distance_boost=1.0, distance_weight=0.5):
        cfg_d = np.array([cfg["k1"], cfg["b"], cfg["summation_ratio"], cfg["full_match_boost"], distance_boost, distance_weight], np.float64)
        cfg_i = np.array([cfg["min_rank"], cfg["merge_limit"]], np.int32)
        fc = np.stack([np.asarray(cfg[k], np.float64) for k in ("bm25_boost", "bm25_weight", "term_len_boost", "term_len_weight",
                                                               "position_boost", "position_weight")], axis=1).copy()
        self.L.ref_ft_set_config(self.h, cfg_d.ctypes.data, cfg_i.ctypes.data, fc.ctypes.data)
        bm25_type = cfg.get("bm25_type", "rx")
        fn = getattr(self.L, "ref_ft_set_bm25_type", None)   # absent in a libref_ft.so built before the switch existed
        if fn is not None:
            fn.argtypes = [_vp, _i]
            fn(self.h, {"rx": 0, "classic": 1, "word_count": 2}[bm25_type])
        elif bm25_type != "rx":
            raise RuntimeError("oracle/_ref/libref_ft.so predates the bm25Type switch: rebuild with `make -C oracle ref`")

    def merge(self, terms, excluded=None, rank_sort_type=1, cap=1 << 16, synonyms=None, part_synonyms=None):
        """terms: list of dict(op, opts (FtOracle.default_opts-like), subs=[(word_id, proc), ...][, phrase=<phraseNum>, distance=<d>]);
        consecutive terms with the same phrase number >= 0 form one phrase (the shim groups them like Selector::Process).
        synonyms: [[term, ...], ...] multi-word synonyms (Synonym::Terms()); part_synonyms[i]: ids of the synonyms of query part i."""
        nf = self.nf
        n_part_terms = len(terms)
        syn_off = [0]
        if synonyms:
            terms = list(terms)
            for syn in synonyms:
                terms.extend(syn)
                syn_off.append(len(terms) - n_part_terms)
        phr = np.array([t.get("phrase", -1) for t in terms], np.int32)
        dst = np.array([t.get("distance", 1) for t in terms], np.int32)
        ops = np.array([t["op"] for t in terms], np.int32)
        boosts = np.array([t["opts"]["boost"] for t in terms], np.float32)
        tlb = np.array([t["opts"]["term_len_boost"] for t in terms], np.float32)
        fb = np.array([t["opts"]["field_boost"] for t in terms], np.float32).reshape(len(terms), nf).copy()
        ns = np.array([t["opts"]["need_sum_rank"] for t in terms], np.uint8).reshape(len(terms), nf).copy()
        sub_off, sw, sp = [0], [], []
        for t in terms:
            for w, p in t["subs"]:
                sw.append(w)
                sp.append(p)
            sub_off.append(len(sw))
        sub_off, sw, sp = np.array(sub_off, np.uint32), np.array(sw, np.uint32), np.array(sp, np.float32)
        exc = np.ascontiguousarray(excluded, np.uint8) if excluded is not None else None
        oid, op = np.zeros(cap, np.int32), np.zeros(cap, np.float32)
        of, on = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8)
        if synonyms:
            nparts = sum(1 for i in range(n_part_terms) if phr[i] < 0 or i == 0 or phr[i - 1] != phr[i])
            ps_off, ps = [0], []
            for pi in range(nparts):
                ps.extend(part_synonyms[pi] if part_synonyms and pi < len(part_synonyms) else [])
                ps_off.append(len(ps))
            syn_off_a, ps_off_a, ps_a = np.array(syn_off, np.uint32), np.array(ps_off, np.uint32), np.array(ps + [0], np.uint32)
            fn = self.L.ref_ft_merge_full
            fn.restype = C.c_long
            fn.argtypes = [_vp, _sz, _sz] + [_vp] * 10 + [_sz] + [_vp] * 4 + [_i, _vp, _vp, _vp, _vp, _sz]
            n = fn(self.h, n_part_terms, len(terms) - n_part_terms, ops.ctypes.data, boosts.ctypes.data, tlb.ctypes.data, fb.ctypes.data, ns.ctypes.data,
                   phr.ctypes.data, dst.ctypes.data, sub_off.ctypes.data, sw.ctypes.data, sp.ctypes.data, len(synonyms), syn_off_a.ctypes.data,
                   ps_off_a.ctypes.data, ps_a.ctypes.data, exc.ctypes.data if exc is not None else None, rank_sort_type, oid.ctypes.data, op.ctypes.data,
                   of.ctypes.data, on.ctypes.data, cap)
            assert 0 <= n <= cap, n
            return oid[:n].copy(), op[:n].copy(), of[:n].copy(), on[:n].copy()
        fn = getattr(self.L, "ref_ft_merge_phrases", None)
        if fn is None:
            if (phr >= 0).any():
                raise RuntimeError("oracle/_ref/libref_ft.so predates phrases: rebuild with `make -C oracle ref`")
            n = self.L.ref_ft_merge(self.h, len(terms), ops.ctypes.data, boosts.ctypes.data, tlb.ctypes.data, fb.ctypes.data, ns.ctypes.data,
                                    sub_off.ctypes.data, sw.ctypes.data, sp.ctypes.data, exc.ctypes.data if exc is not None else None,
                                    rank_sort_type, oid.ctypes.data, op.ctypes.data, of.ctypes.data, on.ctypes.data, cap)
        else:
            fn.restype = C.c_long
            fn.argtypes = [_vp, _sz] + [_vp] * 11 + [_i, _vp, _vp, _vp, _vp, _sz]
            n = fn(self.h, len(terms), ops.ctypes.data, boosts.ctypes.data, tlb.ctypes.data, fb.ctypes.data, ns.ctypes.data, phr.ctypes.data,
                   dst.ctypes.data, sub_off.ctypes.data, sw.ctypes.data, sp.ctypes.data, exc.ctypes.data if exc is not None else None,
                   rank_sort_type, oid.ctypes.data, op.ctypes.data, of.ctypes.data, on.ctypes.data, cap)
        assert 0 <= n <= cap, n
        return oid[:n].copy(), op[:n].copy(), of[:n].copy(), on[:n].copy()


REF_FT_SEAM_SO = HERE / "_ref" / "libref_ft_seam.so"


class RefFtSeam(RefFt):
    """The FT half of the drop-in boundary, executed over the reference's own types (oracle/ref/ref_ft_seam_shim.cc): one word table held
    as PackedWordEntry<PackedIdRelVec> and <IdRelVec>, merged by the reference's ft::Merger (gpu=False) or by the adapter the patched
    Selector::mergeResults calls first (gpu=True: rx_ft_seam.h -> GpuFtMerger -> kernels)."""

    def __init__(self, num_fields: int):
        if not REF_FT_SEAM_SO.exists():
            raise FileNotFoundError(REF_FT_SEAM_SO)
        # this checker library links the product (librxgpu_host.so -> librxgpu.so -> HIP / RCCL): the product's loader goes first, so that the
        # process ends up with ONE HIP runtime whatever test runs first (reindexer_amd.capi.lib: torch's bundled runtime wins when torch is installed)
        from reindexer_amd import hostapi as _product
        _product.lib()
        L = self.L = C.CDLL(str(REF_FT_SEAM_SO))
        L.ref_seam_create.restype = _vp
        L.ref_seam_create.argtypes = [_sz]
        L.ref_seam_destroy.argtypes = [_vp]
        L.ref_seam_last_error.restype = C.c_char_p
        L.ref_seam_last_error.argtypes = [_vp]
        L.ref_seam_set_docs.argtypes = [_vp, _sz, _vp, _vp, _vp]
        L.ref_seam_set_word.argtypes = [_vp, C.c_uint32, _sz, _vp, _vp, _vp, _vp, _vp]
        L.ref_seam_set_config.argtypes = [_vp, _vp, _vp, _vp, _i]
        L.ref_seam_commit.restype = C.c_long
        L.ref_seam_commit.argtypes = [_vp, _i]
        L.ref_seam_merge.restype = C.c_long
        L.ref_seam_merge.argtypes = [_vp, _i, _i, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz]
        self.nf = num_fields
        self.h = L.ref_seam_create(num_fields)

    def close(self):
        if self.h:
            self.L.ref_seam_destroy(self.h)
            self.h = None

    def set_docs(self, words, avg, removed=None):
        words = _f32(words).reshape(-1, self.nf)
        avg = _f32(avg)
        rem = np.ascontiguousarray(removed, np.uint8) if removed is not None else None
        self.total = words.shape[0]
        self.L.ref_seam_set_docs(self.h, words.shape[0], words.ctypes.data, avg.ctypes.data, rem.ctypes.data if rem is not None else None)

    def set_word_fpos(self, word_id, s):
        fp = np.asarray(s["fpos"], np.uint64)
        doc = np.ascontiguousarray(s["doc"], np.uint32)
        po = np.ascontiguousarray(s["pos_off"], np.uint32)
        pf = (fp >> np.uint64(56)).astype(np.uint32)
        pp = (fp & np.uint64((1 << 28) - 1)).astype(np.uint32)
        pa = ((fp >> np.uint64(28)) & np.uint64((1 << 28) - 1)).astype(np.uint32)
        self.L.ref_seam_set_word(self.h, word_id, doc.shape[0], doc.ctypes.data, po.ctypes.data, pf.ctypes.data, pp.ctypes.data, pa.ctypes.data)

    def set_config(self, cfg: dict, distance_boost=1.0, distance_weight=0.5):
        cfg_d = np.array([cfg["k1"], cfg["b"], cfg["summation_ratio"], cfg["full_match_boost"], distance_boost, distance_weight], np.float64)
        cfg_i = np.array([cfg["min_rank"], cfg["merge_limit"]], np.int32)
        fc = np.stack([np.asarray(cfg[k], np.float64) for k in ("bm25_boost", "bm25_weight", "term_len_boost", "term_len_weight",
                                                               "position_boost", "position_weight")], axis=1).copy()
        self.L.ref_seam_set_config(self.h, cfg_d.ctypes.data, cfg_i.ctypes.data, fc.ctypes.data,
                                   {"rx": 0, "classic": 1, "word_count": 2}[cfg.get("bm25_type", "rx")])

    def commit(self, device: int = 0, devices=None) -> int:
        """The end of IndexText::commitFulltextImpl in the patched tree: statistics + changed words go to the device mirrors.
        devices=[d0, d1, ...]: mirrors over a device list (document-range shards), what RX_GPU_FT_INDEXES=<list> makes the patched tree build."""
        if devices is not None and len(devices) > 1:
            dv = np.ascontiguousarray(devices, np.int32)
            self.L.ref_seam_commit_devices.restype = C.c_long
            self.L.ref_seam_commit_devices.argtypes = [_vp, _vp, _sz]
            n = self.L.ref_seam_commit_devices(self.h, dv.ctypes.data, dv.shape[0])
        else:
            n = self.L.ref_seam_commit(self.h, device)
        if n < 0:
            raise RuntimeError(self.L.ref_seam_last_error(self.h).decode(errors="replace"))
        return n

    def merge_areas(self, terms, max_areas=5, excluded=None, rank_sort_type=1, cap=1 << 16, packed=True, gpu=False, area_cap=1 << 20):
        """The merge behind highlight() / snippet(): MergedDataType = MergeDataAreas<Area>, through the reference's ft::Merger (gpu=False) or the
        adapter the patched mergeResults calls first (gpu=True; None when it declines).  -> (ids, proc, field, norm, raw, committed): raw /
        committed = per merged document a list over the fields of (k, 3) uint32 arrays {start, end, arrayIdx} — AreasInField::data_ as the merge
        left it / after Commit()."""
        nf = self.nf
        ops = np.array([t["op"] for t in terms], np.int32)
        boosts = np.array([t["opts"]["boost"] for t in terms], np.float32)
        tlb = np.array([t["opts"]["term_len_boost"] for t in terms], np.float32)
        fb = np.array([t["opts"]["field_boost"] for t in terms], np.float32).reshape(len(terms), nf).copy()
        ns = np.array([t["opts"]["need_sum_rank"] for t in terms], np.uint8).reshape(len(terms), nf).copy()
        phr = np.array([t.get("phrase", -1) for t in terms], np.int32)
        dst = np.array([t.get("distance", 1) for t in terms], np.int32)
        sub_off, sw, sp = [0], [], []
        for t in terms:
            for w, p in t["subs"]:
                sw.append(w)
                sp.append(p)
            sub_off.append(len(sw))
        sub_off, sw, sp = np.array(sub_off, np.uint32), np.array(sw + [0], np.uint32), np.array(sp + [0.0], np.float32)
        exc = np.ascontiguousarray(excluded, np.uint8) if excluded is not None else None
        oid, op = np.zeros(cap, np.int32), np.zeros(cap, np.float32)
        of, on = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8)
        roff, coff = np.zeros(cap * nf + 1, np.uint32), np.zeros(cap * nf + 1, np.uint32)
        ra, ca = np.zeros((area_cap, 3), np.uint32), np.zeros((area_cap, 3), np.uint32)
        fn = self.L.ref_seam_merge_areas
        fn.restype = C.c_long
        fn.argtypes = [_vp, _i, _i, _i, _sz] + [_vp] * 11 + [_i, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _sz]
        n = fn(self.h, int(packed), int(gpu), int(max_areas), len(terms), ops.ctypes.data, boosts.ctypes.data, tlb.ctypes.data, fb.ctypes.data, ns.ctypes.data,
               phr.ctypes.data, dst.ctypes.data, sub_off.ctypes.data, sw.ctypes.data, sp.ctypes.data, exc.ctypes.data if exc is not None else None,
               rank_sort_type, oid.ctypes.data, op.ctypes.data, of.ctypes.data, on.ctypes.data, cap, roff.ctypes.data, ra.ctypes.data, coff.ctypes.data,
               ca.ctypes.data, area_cap)
        if n == -2:
            return None
        if n < 0:
            raise RuntimeError(self.L.ref_seam_last_error(self.h).decode(errors="replace"))
        assert n <= cap
        raw = [[ra[roff[i * nf + f]:roff[i * nf + f + 1]].copy() for f in range(nf)] for i in range(n)]
        com = [[ca[coff[i * nf + f]:coff[i * nf + f + 1]].copy() for f in range(nf)] for i in range(n)]
        return oid[:n].copy(), op[:n].copy(), of[:n].copy(), on[:n].copy(), raw, com

    def merge(self, terms, excluded=None, rank_sort_type=1, cap=1 << 16, packed=True, gpu=False, synonyms=None, part_synonyms=None):
        nf = self.nf
        n_part_terms = len(terms)
        syn_off = [0]
        if synonyms:
            terms = list(terms)
            for syn in synonyms:
                terms.extend(syn)
                syn_off.append(len(terms) - n_part_terms)
        ops = np.array([t["op"] for t in terms], np.int32)
        boosts = np.array([t["opts"]["boost"] for t in terms], np.float32)
        tlb = np.array([t["opts"]["term_len_boost"] for t in terms], np.float32)
        fb = np.array([t["opts"]["field_boost"] for t in terms], np.float32).reshape(len(terms), nf).copy()
        ns = np.array([t["opts"]["need_sum_rank"] for t in terms], np.uint8).reshape(len(terms), nf).copy()
        sub_off, sw, sp = [0], [], []
        for t in terms:
            for w, p in t["subs"]:
                sw.append(w)
                sp.append(p)
            sub_off.append(len(sw))
        sub_off, sw, sp = np.array(sub_off, np.uint32), np.array(sw, np.uint32), np.array(sp, np.float32)
        exc = np.ascontiguousarray(excluded, np.uint8) if excluded is not None else None
        oid, op = np.zeros(cap, np.int32), np.zeros(cap, np.float32)
        of, on = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8)
        phr = np.array([t.get("phrase", -1) for t in terms], np.int32)
        dst = np.array([t.get("distance", 1) for t in terms], np.int32)
        nparts = sum(1 for i in range(n_part_terms) if phr[i] < 0 or i == 0 or phr[i - 1] != phr[i])
        ps_off, ps = [0], []
        for pi in range(nparts):
            ps.extend(part_synonyms[pi] if part_synonyms and pi < len(part_synonyms) else [])
            ps_off.append(len(ps))
        syn_off_a, ps_off_a, ps_a = np.array(syn_off, np.uint32), np.array(ps_off, np.uint32), np.array(ps + [0], np.uint32)
        fn = self.L.ref_seam_merge_full
        fn.restype = C.c_long
        fn.argtypes = [_vp, _i, _i, _sz, _sz] + [_vp] * 10 + [_sz] + [_vp] * 4 + [_i, _vp, _vp, _vp, _vp, _sz]
        n = fn(self.h, int(packed), int(gpu), n_part_terms, len(terms) - n_part_terms, ops.ctypes.data, boosts.ctypes.data, tlb.ctypes.data, fb.ctypes.data,
               ns.ctypes.data, phr.ctypes.data, dst.ctypes.data, sub_off.ctypes.data, sw.ctypes.data, sp.ctypes.data, len(synonyms) if synonyms else 0,
               syn_off_a.ctypes.data, ps_off_a.ctypes.data, ps_a.ctypes.data, exc.ctypes.data if exc is not None else None, rank_sort_type,
               oid.ctypes.data, op.ctypes.data, of.ctypes.data, on.ctypes.data, cap)
        if n == -2:
            return None   # the GPU branch declined: the CPU merger would run
        if n < 0:
            raise RuntimeError(self.L.ref_seam_last_error(self.h).decode(errors="replace"))
        assert n <= cap, n
        return oid[:n].copy(), op[:n].copy(), of[:n].copy(), on[:n].copy()


def ref_ft_seam_or_none(num_fields: int):
    try:
        return RefFtSeam(num_fields)
    except (FileNotFoundError, OSError):
        return None


def ref_ft_or_none(num_fields: int):
    try:
        return RefFt(num_fields)
    except (FileNotFoundError, OSError):
        return None


# ---------------------------------------------------------------------------------------------------- SQ8 (uint8) distance path
def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


class _Sq8Base:
    """Same method names over the restatement (oracle_sq8.c) and over the real reference (ref_shim.cc): codes are uint8 arrays, `corr` the
    corrective offset Quantizer::quantize returns, distances follow DistCalculator<uint8_t> (smaller = closer for every metric)."""
    prefix = ""

    def _bind(self, L):
        p = self.prefix
        for name in ("l2sqr_u8", "ip_u8"):
            fn = getattr(L, p + name)
            fn.restype = _f
            fn.argtypes = [_vp, _vp, _sz]
        getattr(L, p + "sq8_params").argtypes = [_f, _f, _sz, _vp, _vp, _vp]

    def l2sqr_u8(self, a, b):
        a, b = _u8(a), _u8(b)
        return np.float32(getattr(self.L, self.prefix + "l2sqr_u8")(a.ctypes.data, b.ctypes.data, a.shape[0]))

    def ip_u8(self, a, b):
        a, b = _u8(a), _u8(b)
        return np.float32(getattr(self.L, self.prefix + "ip_u8")(a.ctypes.data, b.ctypes.data, a.shape[0]))

    def params(self, min_q, max_q, dim):
        out = np.zeros(3, np.float32)
        getattr(self.L, self.prefix + "sq8_params")(min_q, max_q, dim, out[0:].ctypes.data, out[1:].ctypes.data, out[2:].ctypes.data)
        return dict(min_q=np.float32(min_q), max_q=np.float32(max_q), alpha=out[0], alpha_2=out[1], delta=out[2])


class Sq8Oracle(_Sq8Base):
    prefix = "orc_"

    def __init__(self, orc: Oracle):
        self.L = orc.L
        self._bind(self.L)
        self.L.orc_sq8_quantize.restype = _f
        self.L.orc_sq8_quantize.argtypes = [_i, _sz, _f, _f, _f, _vp, _f, _vp]
        self.L.orc_sq8_dist.restype = _f
        self.L.orc_sq8_dist.argtypes = [_i, _sz, _f, _vp, _f, _f, _vp, _f, _f]
        self.L.orc_sq8_dist_query.restype = _f
        self.L.orc_sq8_dist_query.argtypes = [_i, _sz, _f, _vp, _f, _vp, _f, _f]
        self.L.orc_sq8_dist_query_many.argtypes = [_i, _sz, _f, _vp, _f, _vp, _vp, _vp, _sz, _vp]

    def quantize(self, metric, p, vec, scale=1.0):
        vec = _f32(vec)
        to = np.empty(vec.shape[0], np.uint8)
        corr = self.L.orc_sq8_quantize(metric, vec.shape[0], p["min_q"], p["alpha"], p["delta"], vec.ctypes.data, scale, to.ctypes.data)
        return to, np.float32(corr)

    def dist_pair(self, metric, p, a, corr_a, fa, b, corr_b, fb, orc: Oracle):
        a, b = _u8(a), _u8(b)
        na = orc.l2_module(fa) if metric == 2 else 1.0
        nb = orc.l2_module(fb) if metric == 2 else 1.0
        return np.float32(self.L.orc_sq8_dist(metric, a.shape[0], p["alpha_2"], a.ctypes.data, corr_a, na, b.ctypes.data, corr_b, nb))

    def dist_query(self, metric, p, q, corr_q, row, corr_row, frow, orc: Oracle):
        q, row = _u8(q), _u8(row)
        n = orc.l2_module(frow) if metric == 2 else 1.0
        return np.float32(self.L.orc_sq8_dist_query(metric, q.shape[0], p["alpha_2"], q.ctypes.data, corr_q, row.ctypes.data, corr_row, n))

    def dist_query_many(self, metric, p, q, corr_q, rows, corr, inv_norms=None):
        q, rows, corr = _u8(q), _u8(rows), _f32(corr)
        out = np.empty(rows.shape[0], np.float32)
        inv = _f32(inv_norms) if inv_norms is not None else None
        self.L.orc_sq8_dist_query_many(metric, rows.shape[1], p["alpha_2"], q.ctypes.data, corr_q, rows.ctypes.data, corr.ctypes.data,
                                       inv.ctypes.data if inv is not None else None, rows.shape[0], out.ctypes.data)
        return out


class Sq8Ref(_Sq8Base):
    prefix = "ref_"

    def __init__(self, ref: Ref):
        self.L = ref.L
        self._bind(self.L)   # AttributeError on a libref_oracle.so built before the SQ8 shims existed
        self.L.ref_sq8_quantize.restype = _f
        self.L.ref_sq8_quantize.argtypes = [_i, _sz, _f, _f, _vp, _f, _vp]
        self.L.ref_sq8_dist_pair.restype = _f
        self.L.ref_sq8_dist_pair.argtypes = [_i, _sz, _f, _vp, _f, _vp, _vp, _f, _vp]
        self.L.ref_sq8_dist_query.restype = _f
        self.L.ref_sq8_dist_query.argtypes = [_i, _sz, _f, _vp, _f, _vp, _f, _vp]

    def quantize(self, metric, p, vec, scale=1.0):
        vec = _f32(vec)
        to = np.empty(vec.shape[0], np.uint8)
        corr = self.L.ref_sq8_quantize(metric, vec.shape[0], p["min_q"], p["max_q"], vec.ctypes.data, scale, to.ctypes.data)
        return to, np.float32(corr)

    def dist_pair(self, metric, p, a, corr_a, fa, b, corr_b, fb, orc=None):
        a, b, fa, fb = _u8(a), _u8(b), _f32(fa), _f32(fb)
        return np.float32(self.L.ref_sq8_dist_pair(metric, a.shape[0], p["alpha_2"], a.ctypes.data, corr_a, fa.ctypes.data, b.ctypes.data, corr_b,
                                                   fb.ctypes.data))

    def dist_query(self, metric, p, q, corr_q, row, corr_row, frow, orc=None):
        q, row, frow = _u8(q), _u8(row), _f32(frow)
        return np.float32(self.L.ref_sq8_dist_query(metric, q.shape[0], p["alpha_2"], q.ctypes.data, corr_q, row.ctypes.data, corr_row,
                                                    frow.ctypes.data))


class RefHnswQ:
    """The reference's quantised engine (HierarchicalNSWImpl<uint8_t>) built from a RefHnsw by the copy constructor its own Quantize() uses."""

    def __init__(self, float_graph: "RefHnsw", sample_size: int = 20000, quantile: float = 0.0):
        L = self.L = float_graph.ref.L
        L.ref_hnsw_quantize.restype = _vp
        L.ref_hnsw_quantize.argtypes = [_vp, _sz, _f]
        L.ref_hnswq_destroy.argtypes = [_vp]
        L.ref_hnswq_export.argtypes = [_vp, _vp, _vp, _vp]
        L.ref_hnswq_search_knn.restype = C.c_long
        L.ref_hnswq_search_knn.argtypes = [_vp, _vp, _i, _f, _sz, _sz, _vp, _vp]
        self.dim, self.n = float_graph.dim, float_graph.count
        self.h = L.ref_hnsw_quantize(float_graph.h, sample_size, quantile)
        if not self.h:
            raise RuntimeError(L.ref_last_error().decode())

    def close(self):
        if self.h:
            self.L.ref_hnswq_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def export(self) -> dict:
        params = np.zeros(5, np.float32)
        codes = np.zeros((self.n, self.dim), np.uint8)
        corr = np.zeros(self.n, np.float32)
        self.L.ref_hnswq_export(self.h, params.ctypes.data, codes.ctypes.data, corr.ctypes.data)
        return dict(min_q=params[0], max_q=params[1], alpha=params[2], alpha_2=params[3], delta=params[4], codes=codes, corr=corr)

    def search_knn(self, q, k, ef=0, norm=None):
        q = _f32(q)
        od, ol = np.empty(max(k, 1), np.float32), np.empty(max(k, 1), np.uint64)
        c = self.L.ref_hnswq_search_knn(self.h, q.ctypes.data, int(norm is not None), 0.0 if norm is None else norm, k, ef, od.ctypes.data, ol.ctypes.data)
        if c < 0:
            raise RuntimeError(self.L.ref_last_error().decode())
        return od[:c].copy(), ol[:c].copy()


    def search_range(self, q, radius, ef, norm=None, cap=1 << 20):
        L = self.L
        L.ref_hnswq_search_range.restype = C.c_long
        L.ref_hnswq_search_range.argtypes = [_vp, _vp, _i, _f, _f, _sz, _vp, _vp, _sz]
        q = _f32(q)
        od, ol = np.empty(cap, np.float32), np.empty(cap, np.uint64)
        c = L.ref_hnswq_search_range(self.h, q.ctypes.data, int(norm is not None), 0.0 if norm is None else norm, radius, ef, od.ctypes.data, ol.ctypes.data, cap)
        if c < 0:
            raise RuntimeError(L.ref_last_error().decode())
        assert c <= cap
        return od[:c].copy(), ol[:c].copy()

    def stream(self, q, ef=0, norm=None):
        return _RefStreamQ(self, q, ef, norm)


class _RefStreamQ:
    """BeginStreamingSearch / ContinueStreamingSearch of the quantised engine; next(batch) -> (dist, label, exhausted), worst first."""

    def __init__(self, owner: "RefHnswQ", q, ef, norm):
        L = self.L = owner.L
        L.ref_hnswq_stream_begin.restype = _vp
        L.ref_hnswq_stream_begin.argtypes = [_vp, _vp, _sz, _i, _f, _sz]
        L.ref_hnswq_stream_continue.restype = C.c_long
        L.ref_hnswq_stream_continue.argtypes = [_vp, _vp, _sz, _vp, _vp, _vp]
        L.ref_hnsw_stream_end.argtypes = [_vp]
        self.owner = owner
        q = _f32(q)
        self.s = L.ref_hnswq_stream_begin(owner.h, q.ctypes.data, owner.dim, int(norm is not None), 0.0 if norm is None else norm, ef)
        if not self.s:
            raise RuntimeError(L.ref_last_error().decode())

    def next(self, batch):
        cap = max(1, min(batch, self.owner.n + 1))
        od, ol = np.empty(cap, np.float32), np.empty(cap, np.uint64)
        ex = C.c_int(0)
        c = self.L.ref_hnswq_stream_continue(self.owner.h, self.s, batch, od.ctypes.data, ol.ctypes.data, C.byref(ex))
        if c < 0:
            raise RuntimeError(self.L.ref_last_error().decode())
        return od[:c].copy(), ol[:c].copy(), bool(ex.value)

    def close(self):
        if self.s:
            self.L.ref_hnsw_stream_end(self.s)
            self.s = None

    def __del__(self):
        self.close()


def oracle_hnsw_search_knn_sq8(orc: Oracle, g: dict, sq: dict, q, k: int, ef: int = 0, inv_norms=None, qnorm=None):
    """Restated SearchKnn over an SQ8 graph: g = the flat graph (links of the float graph), sq = RefHnswQ.export()-shaped dict."""
    L = orc.L
    L.orc_hnsw_search_knn_sq8.restype = _sz
    L.orc_hnsw_search_knn_sq8.argtypes = [_i, _sz, _sz, _sz, _sz, _i, C.c_uint32, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                          _f, _f, _f, _f, _vp, _i, _f, _sz, _sz, _vp, _vp]
    q = _f32(q)
    od, ol = np.empty(max(k, 1), np.float32), np.empty(max(k, 1), np.uint64)
    inv = _f32(inv_norms) if inv_norms is not None else None
    codes, corr = _u8(sq["codes"]), _f32(sq["corr"])
    c = L.orc_hnsw_search_knn_sq8(g["metric"], g["n"], g["dim"], g["M"], g["maxM0"], g["maxlevel"], g["entry"], g["num_deleted"],
                                  g["links0"].ctypes.data, g["upper_off"].ctypes.data, g["upper"].ctypes.data, g["levels"].ctypes.data,
                                  g["labels"].ctypes.data, g["deleted"].ctypes.data, codes.ctypes.data, corr.ctypes.data,
                                  inv.ctypes.data if inv is not None else None, sq["min_q"], sq["alpha"], sq["alpha_2"], sq["delta"],
                                  q.ctypes.data, int(qnorm is not None), 0.0 if qnorm is None else qnorm, k, ef, od.ctypes.data, ol.ctypes.data)
    return od[:c].copy(), ol[:c].copy()


# ---------------------------------------------------------------------------------------------------- the reference's IVF backend (FAISS)
REF_IVF_SO = HERE / "_ref" / "libref_ivf.so"


class RefIvf:
    """The patched FAISS vendored in the reference, compiled in place (oracle/ref/ref_ivf_shim.cc), driven like IvfIndex drives it: a flat
    index trained on and drained into an IndexIVFFlat.  Distances follow FAISS (L2 ascending, similarity descending)."""

    def __init__(self, metric: int, dim: int, nlist: int, x, ids, exact_assignment: bool = False):
        """exact_assignment: train with FAISS's BLAS shortcut off (distance_compute_blas_threshold = INT_MAX): the k-means assignment then
        uses the reference's exact per-pair distance functions instead of |x|^2 + |y|^2 - 2 x.y through sgemm."""
        if not REF_IVF_SO.exists():
            raise FileNotFoundError(REF_IVF_SO)
        L = self.L = C.CDLL(str(REF_IVF_SO))
        old = L.ref_ivf_set_blas_threshold(0x7FFFFFFF) if exact_assignment else None
        L.ref_ivf_build.restype = _vp
        L.ref_ivf_build.argtypes = [_i, _sz, _sz, _sz, _vp, _vp]
        L.ref_ivf_destroy.argtypes = [_vp]
        L.ref_ivf_search.argtypes = [_vp, _vp, _sz, _sz, _vp, _vp]
        L.ref_ivf_range.restype = C.c_long
        L.ref_ivf_range.argtypes = [_vp, _vp, _f, _sz, _vp, _vp, _sz]
        L.ref_ivf_export.argtypes = [_vp, _vp, _vp]
        L.ref_ivf_list_ids.argtypes = [_vp, _sz, _vp]
        L.ref_ivf_remove.restype = C.c_long
        L.ref_ivf_remove.argtypes = [_vp, _vp, _sz]
        L.ref_ivf_last_error.restype = C.c_char_p
        x = _f32(x).reshape(-1, dim)
        ids = np.ascontiguousarray(ids, np.int64)
        self.metric, self.dim, self.nlist = metric, dim, nlist
        self.h = L.ref_ivf_build(metric, dim, nlist, x.shape[0], x.ctypes.data, ids.ctypes.data)
        if old is not None:
            L.ref_ivf_set_blas_threshold(old)
        if not self.h:
            raise RuntimeError(L.ref_ivf_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.ref_ivf_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def search(self, q, k, nprobe):
        q = _f32(q)
        d, l = np.empty(k, np.float32), np.empty(k, np.int64)
        if self.L.ref_ivf_search(self.h, q.ctypes.data, k, nprobe, d.ctypes.data, l.ctypes.data):
            raise RuntimeError(self.L.ref_ivf_last_error().decode())
        return d, l

    def range_search(self, q, radius, nprobe, cap=1 << 16):
        q = _f32(q)
        while True:
            d, l = np.empty(cap, np.float32), np.empty(cap, np.int64)
            n = self.L.ref_ivf_range(self.h, q.ctypes.data, radius, nprobe, d.ctypes.data, l.ctypes.data, cap)
            if n < 0:
                raise RuntimeError(self.L.ref_ivf_last_error().decode())
            if n <= cap:
                return d[:n].copy(), l[:n].copy()
            cap = int(n)

    def export(self):
        cent = np.zeros((self.nlist, self.dim), np.float32)
        sizes = np.zeros(self.nlist, np.uint64)
        self.L.ref_ivf_export(self.h, cent.ctypes.data, sizes.ctypes.data)
        lists = []
        for i in range(self.nlist):
            ids = np.zeros(int(sizes[i]), np.int64)
            if ids.size:
                self.L.ref_ivf_list_ids(self.h, i, ids.ctypes.data)
            lists.append(ids)
        return cent, lists

    def remove_ids(self, ids):
        ids = np.ascontiguousarray(ids, np.int64)
        return int(self.L.ref_ivf_remove(self.h, ids.ctypes.data, ids.shape[0]))


def ref_ivf_available() -> bool:
    return REF_IVF_SO.exists()


# ------------------------------------------------------------------------------------------------ the REAL hybrid rank fusion (_ref)
REF_RANK_SO = HERE / "_ref" / "libref_rank.so"


class RefRank:
    """SelectIteratorContainer::MergerRankedImpl + the drain of mergeRanked + RanksHolder::InitRRFPositions of the reference
    (oracle/ref/ref_rank_shim.cc: selectiteratorcontainer.cc compiled in place)."""

    def __init__(self):
        if not REF_RANK_SO.exists():
            raise FileNotFoundError(REF_RANK_SO)
        L = self.L = C.CDLL(str(REF_RANK_SO))
        L.ref_rank_uses_pmr.restype = _i
        L.ref_rank_init_rrf_positions.argtypes = [_vp, _sz, _vp]
        L.ref_rank_merge_rrf.restype = C.c_long
        L.ref_rank_merge_rrf.argtypes = [C.c_double, _i, _i, _i, _vp, _vp, _sz, _vp, _vp, _vp, _sz, _vp, _vp, _sz]
        L.ref_rank_merge_linear.restype = C.c_long
        L.ref_rank_merge_linear.argtypes = [C.c_double] * 5 + [_i, _i, _i, _vp, _vp, _sz, _vp, _vp, _sz, _vp, _vp, _sz]

    @property
    def uses_pmr(self) -> bool:
        return bool(self.L.ref_rank_uses_pmr())

    def rrf_positions(self, ranks_ft_order):
        r = _f32(ranks_ft_order)
        out = np.zeros(r.shape[0], np.uint64)
        self.L.ref_rank_init_rrf_positions(r.ctypes.data, r.shape[0], out.ctypes.data)
        return out

    def merge(self, kind, params, knn_ids, knn_ranks, ft_ids, ft_ranks, union=False, desc=True, metric=1, ft_positions=None):
        """ft_ids ascending with ft_ranks (and, for RRF, ft_positions) aligned — the selector's view (selectiteratorcontainer.cc:1494)."""
        ki, kr = np.ascontiguousarray(knn_ids, np.int32), _f32(knn_ranks)
        fi, fr = np.ascontiguousarray(ft_ids, np.int32), _f32(ft_ranks)
        cap = ki.shape[0] + fi.shape[0] + 1
        oi, orr = np.empty(cap, np.int32), np.empty(cap, np.float32)
        if kind == "rrf":
            fp = np.ascontiguousarray(ft_positions, np.uint64)
            assert fp.shape[0] == fi.shape[0]
            n = self.L.ref_rank_merge_rrf(float(params[0]), int(union), int(desc), metric, ki.ctypes.data, kr.ctypes.data, ki.shape[0],
                                          fi.ctypes.data, fr.ctypes.data, fp.ctypes.data, fi.shape[0], oi.ctypes.data, orr.ctypes.data, cap)
        else:
            n = self.L.ref_rank_merge_linear(*[float(x) for x in params], int(union), int(desc), metric, ki.ctypes.data, kr.ctypes.data,
                                             ki.shape[0], fi.ctypes.data, fr.ctypes.data, fi.shape[0], oi.ctypes.data, orr.ctypes.data, cap)
        assert 0 <= n <= cap
        return oi[:n].copy(), orr[:n].copy()


def ref_rank_or_none():
    try:
        return RefRank()
    except (FileNotFoundError, OSError):
        return None


# ---------------------------------------------------------------------------------------------------- the reference's select post-processing
REF_SELECT_SO = HERE / "_ref" / "libref_select.so"


class RefSelect:
    """_ref/libref_select.so (oracle/ref/ref_select_shim.cc): HnswIndexBase<BruteforceSearch>::select / selectRaw of the reference over its own
    brute-force map."""

    def __init__(self, metric: int, dim: int, max_elements: int, is_array: bool = False, index_radius=None):
        if not REF_SELECT_SO.exists():
            raise FileNotFoundError(REF_SELECT_SO)
        L = self.L = C.CDLL(str(REF_SELECT_SO))
        L.ref_select_create.restype = _vp
        L.ref_select_create.argtypes = [_i, _sz, _sz, _i, _i, _f]
        L.ref_select_destroy.argtypes = [_vp]
        L.ref_select_add.argtypes = [_vp, _vp, _sz, _vp]
        L.ref_select.restype = C.c_long
        L.ref_select.argtypes = [_vp, _vp, C.c_long, _i, _f, _i, _vp, _vp, _sz]
        L.ref_select_raw.restype = C.c_long
        L.ref_select_raw.argtypes = [_vp, _vp, C.c_long, _i, _f, _vp, _vp, _sz]
        self.dim = dim
        self.h = L.ref_select_create(metric, dim, max_elements, int(is_array), int(index_radius is not None),
                                     0.0 if index_radius is None else index_radius)

    def close(self):
        if self.h:
            self.L.ref_select_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def add(self, vecs, labels):
        v, lab = _f32(vecs), np.ascontiguousarray(labels, np.uint64)
        self.L.ref_select_add(self.h, v.ctypes.data, v.shape[0], lab.ctypes.data)

    def select(self, key, k=None, radius=None, need_sort=True, cap=1 << 20):
        key = _f32(key)
        ids, ranks = np.empty(cap, np.int32), np.empty(cap, np.float32)
        n = self.L.ref_select(self.h, key.ctypes.data, -1 if k is None else k, int(radius is not None), 0.0 if radius is None else radius,
                              int(need_sort), ids.ctypes.data, ranks.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("ref_select: ids / ranks size mismatch")
        return ids[:n].copy(), ranks[:n].copy()

    def select_raw(self, key, k=None, radius=None, cap=1 << 20):
        key = _f32(key)
        ids, ranks = np.empty(cap, np.int32), np.empty(cap, np.float32)
        n = self.L.ref_select_raw(self.h, key.ctypes.data, -1 if k is None else k, int(radius is not None), 0.0 if radius is None else radius,
                                  ids.ctypes.data, ranks.ctypes.data, cap)
        return ids[:n].copy(), ranks[:n].copy()


def ref_select_available() -> bool:
    return REF_SELECT_SO.exists()


REF_KNN_SEAM_SO = HERE / "_ref" / "libref_knn_seam.so"


class RefKnnSeam:
    """_ref/libref_knn_seam.so (oracle/ref/ref_knn_seam_shim.cc): the reference's HnswIndexBase<Map> — its hnsw_index.cc with
    integration/patches/0001 applied — over one of four Maps: kind 0 the reference's BruteforceSearch, 1 the MI355X brute-force Map, 2 the
    reference's HierarchicalNSW (single-thread build), 3 the MI355X HNSW Map.  upsert / del / select / selectRaw / streaming as the planner
    reaches them.  Kinds 1 and 3 need a GPU (the Maps are the product's, compiled against the reference's types)."""
    KINDS = {"ref_bf": 0, "gpu_bf": 1, "ref_hnsw": 2, "gpu_hnsw": 3}

    def __init__(self, kind: str, metric: int, dim: int, max_elements: int, is_array: bool = False, M: int = 16, ef_construction: int = 200):
        if not REF_KNN_SEAM_SO.exists():
            raise FileNotFoundError(REF_KNN_SEAM_SO)
        L = self.L = C.CDLL(str(REF_KNN_SEAM_SO))
        L.ref_knn_seam_error.restype = C.c_char_p
        L.ref_knn_seam_create.restype = _vp
        L.ref_knn_seam_create.argtypes = [_i, _i, _sz, _sz, _i, _sz, _sz]
        L.ref_knn_seam_destroy.argtypes = [_vp]
        for name, args in (("upsert", [_vp, _vp, _sz, _sz, _vp]), ("del", [_vp, C.c_uint64]), ("count", [_vp]),
                           ("select", [_vp, _i, _vp, C.c_long, _i, _f, _sz, _i, _vp, _vp, _sz]),
                           ("select_raw", [_vp, _i, _vp, C.c_long, _i, _f, _sz, _vp, _vp, _sz]),
                           ("begin_streaming", [_vp, _vp, _sz]), ("continue_streaming", [_vp, _sz, _vp, _vp, _vp])):
            f = getattr(L, "ref_knn_seam_" + name)
            f.restype = C.c_long
            f.argtypes = args
        self.kind, self.dim, self.hnsw = kind, dim, int(self.KINDS[kind] >= 2)
        self.h = L.ref_knn_seam_create(self.KINDS[kind], metric, dim, max_elements, int(is_array), M, ef_construction)
        if not self.h:
            raise RuntimeError("ref_knn_seam_create: " + L.ref_knn_seam_error().decode(errors="replace"))

    def _check(self, rc):
        if rc == -2:
            raise RuntimeError(self.L.ref_knn_seam_error().decode(errors="replace"))
        if rc < 0:
            raise RuntimeError("ref_knn_seam: ids / ranks size mismatch")
        return rc

    def close(self):
        if getattr(self, "h", None):
            self.L.ref_knn_seam_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def upsert(self, vecs, labels):
        v, lab = _f32(vecs).reshape(-1, self.dim), np.ascontiguousarray(labels, np.uint64)
        self._check(self.L.ref_knn_seam_upsert(self.h, v.ctypes.data, v.shape[0], self.dim, lab.ctypes.data))

    def delete(self, label):
        self._check(self.L.ref_knn_seam_del(self.h, int(label)))

    @property
    def count(self) -> int:
        return int(self.L.ref_knn_seam_count(self.h))

    def select(self, key, k=None, radius=None, ef=0, need_sort=True, cap=1 << 16):
        key = _f32(key)
        ids, ranks = np.empty(cap, np.int32), np.empty(cap, np.float32)
        n = self._check(self.L.ref_knn_seam_select(self.h, self.hnsw, key.ctypes.data, -1 if k is None else k, int(radius is not None),
                                                   0.0 if radius is None else radius, ef, int(need_sort), ids.ctypes.data, ranks.ctypes.data, cap))
        return ids[:n].copy(), ranks[:n].copy()

    def select_raw(self, key, k=None, radius=None, ef=0, cap=1 << 16):
        key = _f32(key)
        ids, ranks = np.empty(cap, np.int32), np.empty(cap, np.float32)
        n = self._check(self.L.ref_knn_seam_select_raw(self.h, self.hnsw, key.ctypes.data, -1 if k is None else k, int(radius is not None),
                                                       0.0 if radius is None else radius, ef, ids.ctypes.data, ranks.ctypes.data, cap))
        return ids[:n].copy(), ranks[:n].copy()

    def begin_streaming(self, key, ef):
        self._key = _f32(key)   # (the session of a non-cosine search keeps the caller's pointer)
        self._check(self.L.ref_knn_seam_begin_streaming(self.h, self._key.ctypes.data, ef))

    def continue_streaming(self, batch):
        ids, ranks, ex = np.empty(batch, np.int32), np.empty(batch, np.float32), C.c_int(0)
        n = self._check(self.L.ref_knn_seam_continue_streaming(self.h, batch, ids.ctypes.data, ranks.ctypes.data, C.byref(ex)))
        return ids[:n].copy(), ranks[:n].copy(), bool(ex.value)


def ref_knn_seam_available() -> bool:
    return REF_KNN_SEAM_SO.exists()
