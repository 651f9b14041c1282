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
        L.ref_hnsw_search_range.restype = _sz
        L.ref_hnsw_search_range.argtypes = [_vp, _vp, _f, _sz, _vp, _vp, _sz]
        L.ref_hnsw_info.argtypes = [_vp, _vp]
        L.ref_hnsw_export_level0.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp]
        L.ref_hnsw_export_upper.restype = _sz
        L.ref_hnsw_export_upper.argtypes = [_vp, _vp, _vp]

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


def ref_or_none() -> Ref | None:
    try:
        return Ref()
    except (FileNotFoundError, OSError):
        return None


class RefHnsw:
    """hnswlib::HierarchicalNSWImpl<float, None> of the reference (seed 100, ReplaceDeleted_True — hnsw.cc:74-78)."""

    def __init__(self, ref: Ref, metric: int, dim: int, capacity: int, M: int = 16, ef_construction: int = 200):
        self.ref, self.dim, self.metric = ref, dim, metric
        self.h = ref.L.ref_hnsw_create(metric, dim, capacity, M, ef_construction)
        assert self.h

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


def _hnsw_bind(L):
    L.orc_hnsw_search_knn.restype = _sz
    L.orc_hnsw_search_knn.argtypes = [_i, _sz, _sz, _sz, _sz, _i, C.c_uint32, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _sz, _vp, _vp]
    L.orc_hnsw_last_stats.argtypes = [_vp, _vp]


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
