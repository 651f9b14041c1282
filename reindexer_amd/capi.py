"""ctypes binding of the C-ABI in include/rxgpu.h (librxgpu.so).

Python here is plumbing for tests / bench / multi-GPU launch only: numpy arrays or torch device pointers in,
numpy arrays out.  There is NO fallback: if the HIP library is missing or fails to load, importing the symbols
raises — a product path must never silently route around the kernels.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / "librxgpu.so"

METRIC_L2, METRIC_IP, METRIC_COSINE = 0, 1, 2
METRICS = {"l2": METRIC_L2, "ip": METRIC_IP, "inner_product": METRIC_IP, "cosine": METRIC_COSINE}

RXGPU_OK = 0
RXGPU_ERR_PARAMS = -3
RXGPU_ERR_LOGIC = -4
RXGPU_ERR_NOMEM = -5
RXGPU_ERR_DEVICE = -6
RXGPU_ERR_NOTFOUND = -7
RXGPU_ERR_OVERFLOW = -8

# every symbol include/rxgpu.h declares (tests/test_abi_symbols.py cross-checks this list against the header)
_vp, _u32, _u64, _i, _f = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_float
_SIGNATURES = {
    "rxgpu_last_error": (C.c_char_p, []),
    "rxgpu_abi_version": (_i, []),
    "rxgpu_device_count": (_i, []),
    "rxgpu_device_arch": (_i, [_i, C.c_char_p, C.c_size_t]),
    "rxgpu_index_create": (_i, [_i, _u32, _u64, _i, C.POINTER(_vp)]),
    "rxgpu_index_destroy": (None, [_vp]),
    "rxgpu_index_create_sharded": (_i, [_i, _u32, _u64, _u32, _vp, C.POINTER(_vp)]),
    "rxgpu_index_shard_count": (_u32, [_vp]),
    "rxgpu_index_shard_rows": (_u64, [_vp]),
    "rxgpu_index_shard_merge_mode": (_i, [_vp]),
    "rxgpu_index_shard_merge_note": (C.c_char_p, [_vp]),
    "rxgpu_index_shard_ranks": (_u32, [_vp]),
    "rxgpu_index_shard_collectives": (_u64, [_vp]),
    "rxgpu_index_shard": (_vp, [_vp, _u32]),
    "rxgpu_index_shard_sync_count": (_i, [_vp]),
    "rxgpu_index_download_row": (_i, [_vp, _u64, _vp, _vp]),
    "rxgpu_index_reserve": (_i, [_vp, _u64]),
    "rxgpu_index_upload_rows": (_i, [_vp, _u64, _u64, _vp, _vp]),
    "rxgpu_index_adopt_device_rows": (_i, [_vp, _vp, _u64, _u32, _vp]),
    "rxgpu_index_move_row": (_i, [_vp, _u64, _u64]),
    "rxgpu_index_truncate": (_i, [_vp, _u64]),
    "rxgpu_index_count": (_u64, [_vp]),
    "rxgpu_index_capacity": (_u64, [_vp]),
    "rxgpu_index_dim": (_u32, [_vp]),
    "rxgpu_index_row_stride": (_u32, [_vp]),
    "rxgpu_index_metric": (_i, [_vp]),
    "rxgpu_index_device": (_i, [_vp]),
    "rxgpu_index_device_bytes": (_u64, [_vp]),
    "rxgpu_search_knn": (_i, [_vp, _vp, _u32, _u32, _vp, _vp, _vp]),
    "rxgpu_index_upload_row_ids": (_i, [_vp, _u64, _u64, _vp]),
    "rxgpu_index_row_ids_device": (_vp, [_vp]),
    "rxgpu_search_knn_resident": (_i, [_vp, _vp, _u32, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_u32)]),
    "rxgpu_index_resident_contexts": (_u32, [_vp]),
    "rxgpu_search_knn_device": (_i, [_vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp]),
    "rxgpu_search_knn_subset": (_i, [_vp, _vp, _u32, _u32, _vp, _u64, _vp, _vp, _vp]),
    "rxgpu_search_knn_bitmap": (_i, [_vp, _vp, _u32, _u32, _vp, _u64, _vp, _vp, _vp, C.POINTER(_u64)]),
    "rxgpu_index_set_lists": (_i, [_vp, _u32, _vp, _vp]),
    "rxgpu_search_knn_lists": (_i, [_vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, C.POINTER(_u64)]),
    "rxgpu_search_range_lists": (_i, [_vp, _vp, _vp, _u32, C.c_float, _i, _vp, _vp, _u64, C.POINTER(_u64), C.POINTER(_u64)]),
    "rxgpu_search_knn_subset_device": (_i, [_vp, _vp, _u32, _u32, _vp, _u64, _vp, _vp, _vp, _vp]),
    "rxgpu_check_row_list_device": (_i, [_vp, _vp, _u64, _vp, C.POINTER(C.c_int32)]),
    "rxgpu_search_range_subset": (_i, [_vp, _vp, _f, _i, _vp, _u64, _vp, _vp, _u64, C.POINTER(_u64)]),
    "rxgpu_merge_shards_device": (_i, [_vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp]),
    "rxgpu_search_range": (_i, [_vp, _vp, _f, _i, _vp, _vp, _u64, C.POINTER(_u64)]),
    "rxgpu_distances": (_i, [_vp, _vp, _vp, _u32, _vp]),
    "rxgpu_hnsw_attach_graph": (_i, [_vp, _vp, _vp, _vp, _u64, _vp, _u32, _u32, C.c_int32, _u32, _u64]),
    "rxgpu_hnsw_update_deleted": (_i, [_vp, _vp, _u64]),
    "rxgpu_hnsw_patch_graph": (_i, [_vp, _u32, _vp, _vp, _vp, _vp, _vp, C.c_int32, _u32, _u64]),
    "rxgpu_hnsw_search_knn": (_i, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp]),
    "rxgpu_hnsw_search_knn_posted": (_i, [_vp, _vp, _u32, _u32, _vp, _vp, _vp, C.POINTER(C.c_int32)]),
    "rxgpu_hnsw_server_stats": (_i, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "rxgpu_hnsw_server_times": (_i, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "rxgpu_hnsw_read_stats4": (_i, [_vp, _vp]),
    "rxgpu_hnsw_attach_sq8": (_i, [_vp, _vp, _vp, _u64, _f]),
    "rxgpu_hnsw_upload_sq8_rows": (_i, [_vp, _u64, _u64, _vp, _vp, _f]),
    "rxgpu_hnsw_search_knn_sq8": (_i, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp]),
    "rxgpu_hnsw_read_stats": (_i, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "rxgpu_hnsw_read_tie_reruns": (_i, [_vp, C.POINTER(_u64)]),
    "rxgpu_hnsw_read_lds_reruns": (_i, [_vp, C.POINTER(_u64)]),
    "rxgpu_hnsw_search_range": (_i, [_vp, _vp, _f, _u32, _vp, _vp, _u64, C.POINTER(_u64)]),
    "rxgpu_hnsw_search_range_sq8": (_i, [_vp, _vp, _f, _f, _f, _u32, _vp, _vp, _u64, C.POINTER(_u64)]),
    "rxgpu_hnsw_stream_begin_sq8": (_i, [_vp, _vp, _f, _f, _u32, C.POINTER(_vp)]),
    "rxgpu_hnsw_stream_begin": (_i, [_vp, _vp, _u32, C.POINTER(_vp)]),
    "rxgpu_hnsw_stream_continue": (_i, [_vp, _u32, _vp, _vp, C.POINTER(_u32), C.POINTER(_i)]),
    "rxgpu_hnsw_stream_end": (None, [_vp]),
    "rxgpu_ft_create": (_i, [_u32, _i, C.POINTER(_vp)]),
    "rxgpu_ft_create_sharded": (_i, [_u32, _u32, _vp, C.POINTER(_vp)]),
    "rxgpu_ft_shard_count": (_u32, [_vp]),
    "rxgpu_ft_shard_imbalance": (C.c_double, [_vp]),
    "rxgpu_ft_shard_exchange_mode": (_i, [_vp]),
    "rxgpu_ft_shard_collectives": (_u64, [_vp]),
    "rxgpu_ft_shard_ranges": (_i, [_vp, _u32, _vp, _vp]),
    "rxgpu_ft_word_df": (_i, [_vp, _u32, C.POINTER(_u64)]),
    "rxgpu_ft_destroy": (None, [_vp]),
    "rxgpu_ft_set_docs": (_i, [_vp, _u64, _vp, _vp, _vp]),
    "rxgpu_ft_set_word": (_i, [_vp, _u32, _u64, _vp, _vp, _vp, _vp, _vp]),
    "rxgpu_ft_merge_simple_raw": (_i, [_vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _u64, C.POINTER(_u64)]),
    "rxgpu_ft_set_word_positions": (_i, [_vp, _u32, _u64, _vp, _vp, _vp]),
    "rxgpu_ft_merge_terms_raw": (_i, [_vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u64, C.POINTER(_u64), C.POINTER(_i)]),
    "rxgpu_ft_merge_query_raw": (_i, [_vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u64, C.POINTER(_u64), C.POINTER(_i)]),
    "rxgpu_ft_merge_query2_raw": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u64, C.POINTER(_u64), C.POINTER(_i)]),
    "rxgpu_ft_merge_query_areas_raw": (_i, [_vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _u64, C.POINTER(_u64), C.POINTER(_i), _vp, _vp]),
    "rxgpu_ft_merge_batch_raw": (_i, [_vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _u64, _vp, _vp]),
    "rxgpu_ft_read_batch_stats": (_i, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "rxgpu_ft_set_train_mode": (None, [_i]),
    "rxgpu_ft_read_train_stats": (_i, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "rxgpu_ft_merge_query_resident": (_i, [_vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i)]),
    "rxgpu_ft_merge_query2_resident": (_i, [_vp, _vp, _vp, _vp, C.POINTER(_i)]),
    "rxgpu_ft_read_stats": (_i, [_vp, C.POINTER(_u64), C.POINTER(C.c_double)]),
    "rxgpu_ft_set_words_packed": (_i, [_vp, _u32, _vp, _vp, _vp, _vp]),
    "rxgpu_ft_set_words_packed_ptrs": (_i, [_vp, _u32, _vp, _vp, _vp, _vp]),
    "rxgpu_ft_read_packed_wall": (_i, [_vp, C.POINTER(C.c_double)]),
    "rxgpu_ft_read_packed_stats": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_u64), C.POINTER(_u64)]),
    "rxgpu_ft_get_word": (_i, [_vp, _u32, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64), _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_u32), _vp]),
    "rxgpu_ft_merge_simple_resident": (_i, [_vp, _vp, _vp, _u32, _vp, _vp, _vp]),
    "rxgpu_ft_merge_terms_resident": (_i, [_vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rxgpu_hybrid_prepare_resident": (_i, [_vp, C.c_int32, _vp, _i, _vp]),
    "rxgpu_hybrid_fuse_resident": (_i, [_vp, C.c_int32, _vp, _i, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _u64, C.POINTER(_u64),
                                        C.POINTER(_u32)]),
    "rxgpu_hybrid_read_stats": (_i, [_vp, C.POINTER(_u64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "rxgpu_hybrid_fuse": (_i, [_i, _vp, _i, _vp, _vp, _u32, _vp, _vp, _u32, _vp, _vp, _u64, C.POINTER(_u64)]),
    "rxgpu_profile_enable": (_i, [_vp, _i]),
    "rxgpu_profile_read": (_i, [_vp, C.c_char_p, C.POINTER(_u64), C.POINTER(C.c_double)]),
}

_lib = None


class RxGpuError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rxgpu error {code}: {msg}")
        self.code = code


def declared_symbols() -> list[str]:
    return sorted(_SIGNATURES)


def lib() -> C.CDLL:
    """Load librxgpu.so (once). Raises if it is not built — there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m reindexer_amd.build` "
                "(hipcc --offload-arch=gfx950). The GPU engine has no CPU fallback.")
        # One process can hold only one HIP runtime.  PyTorch wheels bundle their own libamdhip64.so.7 (same SONAME as
        # /opt/rocm's): whichever is loaded first serves both.  torch must win that race or its device enumeration fails
        # ("No HIP GPUs are available"), so when torch is installed it is imported before librxgpu.so is opened.
        if not os.environ.get("RXGPU_NO_TORCH"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        _lib = C.CDLL(str(LIB_PATH))
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib


def _check(rc: int) -> None:
    if rc != RXGPU_OK:
        raise RxGpuError(rc, lib().rxgpu_last_error().decode(errors="replace"))


def device_count() -> int:
    n = lib().rxgpu_device_count()
    if n < 0:
        _check(n)
    return n


def device_arch(device: int = 0) -> str:
    buf = C.create_string_buffer(256)
    _check(lib().rxgpu_device_arch(device, buf, 256))
    return buf.value.decode()


def _f32c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


class VectorIndex:
    """One float_vector index shard on one GPU (thin wrapper over rxgpu_index*)."""

    def __init__(self, metric: int | str, dim: int, capacity: int = 0, device: int = 0):
        if isinstance(metric, str):
            metric = METRICS[metric.lower()]
        self.metric, self.dim = int(metric), int(dim)
        h = _vp()
        _check(lib().rxgpu_index_create(self.metric, self.dim, capacity, device, C.byref(h)))
        self._h = h
        self._keepalive = None

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().rxgpu_index_destroy(self._h)
            self._h = None
            self._keepalive = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- storage
    @property
    def count(self) -> int:
        return lib().rxgpu_index_count(self._h)

    @property
    def capacity(self) -> int:
        return lib().rxgpu_index_capacity(self._h)

    @property
    def row_stride(self) -> int:
        return lib().rxgpu_index_row_stride(self._h)

    @property
    def device_bytes(self) -> int:
        return lib().rxgpu_index_device_bytes(self._h)

    def reserve(self, capacity: int) -> None:
        _check(lib().rxgpu_index_reserve(self._h, capacity))

    def upload_rows(self, first_row: int, rows, inv_norms=None) -> None:
        rows = _f32c(rows).reshape(-1, self.dim)
        nptr = None
        if inv_norms is not None:
            inv_norms = _f32c(inv_norms).reshape(-1)
            assert inv_norms.shape[0] == rows.shape[0]
            nptr = inv_norms.ctypes.data
        _check(lib().rxgpu_index_upload_rows(self._h, first_row, rows.shape[0], rows.ctypes.data, nptr))

    def adopt_device_rows(self, d_rows_ptr: int, n: int, row_stride: int, d_inv_norms_ptr: int | None = None, keepalive=None) -> None:
        _check(lib().rxgpu_index_adopt_device_rows(self._h, d_rows_ptr, n, row_stride, d_inv_norms_ptr))
        self._keepalive = keepalive

    def move_row(self, src: int, dst: int) -> None:
        _check(lib().rxgpu_index_move_row(self._h, src, dst))

    def truncate(self, count: int) -> None:
        _check(lib().rxgpu_index_truncate(self._h, count))

    # ---- search
    def search_knn(self, queries, kk: int):
        """-> (dist[nq,kk] f32, row[nq,kk] u32, count[nq] u32); exact top-kk under (dist,row) order."""
        q = _f32c(queries).reshape(-1, self.dim)
        nq = q.shape[0]
        dist = np.full((nq, max(kk, 1)), np.inf, np.float32)
        row = np.full((nq, max(kk, 1)), 0xFFFFFFFF, np.uint32)
        cnt = np.zeros(nq, np.uint32)
        _check(lib().rxgpu_search_knn(self._h, q.ctypes.data, nq, kk, dist.ctypes.data, row.ctypes.data, cnt.ctypes.data))
        return dist[:, :kk], row[:, :kk], cnt

    def search_knn_device(self, d_queries_ptr: int, nq: int, kk: int, d_out_dist_ptr: int, d_out_row_ptr: int,
                          d_out_count_ptr: int | None, stream_ptr: int) -> None:
        _check(lib().rxgpu_search_knn_device(self._h, d_queries_ptr, nq, kk, d_out_dist_ptr, d_out_row_ptr, d_out_count_ptr, stream_ptr))

    # ---- pre-filtered search (`WHERE cond AND KNN(...)`)
    def search_knn_resident(self, query, kk: int):
        """rxgpu_search_knn_resident: one host query, the (dist, row) list left in the index's device buffers.
        Returns (d_dist_ptr, d_row_ptr, d_count_ptr, stream_ptr, entries)."""
        q = _f32c(query)
        dd, dr, dc, st = _vp(), _vp(), _vp(), _vp()
        n = _u32(0)
        _check(lib().rxgpu_search_knn_resident(self._h, q.ctypes.data, kk, C.byref(dd), C.byref(dr), C.byref(dc), C.byref(st), C.byref(n)))
        return dd.value, dr.value, dc.value, st.value, int(n.value)

    def search_knn_subset(self, queries, kk: int, row_ids):
        """Exact top-kk among the listed rows only (strictly increasing uint32 internal rows) -> (dist, row, count)."""
        q = _f32c(queries).reshape(-1, self.dim)
        ids = np.ascontiguousarray(row_ids, dtype=np.uint32).reshape(-1)
        nq = q.shape[0]
        dist = np.full((nq, max(kk, 1)), np.inf, np.float32)
        row = np.full((nq, max(kk, 1)), 0xFFFFFFFF, np.uint32)
        cnt = np.zeros(nq, np.uint32)
        _check(lib().rxgpu_search_knn_subset(self._h, q.ctypes.data, nq, kk, ids.ctypes.data if ids.size else None, ids.size,
                                             dist.ctypes.data, row.ctypes.data, cnt.ctypes.data))
        return dist[:, :kk], row[:, :kk], cnt

    def search_knn_bitmap(self, queries, kk: int, allowed_words):
        """Same, the allowed rows given as a bitmap (uint32 words, bit r % 32 of word r // 32) -> (dist, row, count, n_allowed)."""
        q = _f32c(queries).reshape(-1, self.dim)
        words = np.ascontiguousarray(allowed_words, dtype=np.uint32).reshape(-1)
        nq = q.shape[0]
        dist = np.full((nq, max(kk, 1)), np.inf, np.float32)
        row = np.full((nq, max(kk, 1)), 0xFFFFFFFF, np.uint32)
        cnt = np.zeros(nq, np.uint32)
        allowed = _u64(0)
        _check(lib().rxgpu_search_knn_bitmap(self._h, q.ctypes.data, nq, kk, words.ctypes.data, words.size, dist.ctypes.data, row.ctypes.data,
                                             cnt.ctypes.data, C.byref(allowed)))
        return dist[:, :kk], row[:, :kk], cnt, int(allowed.value)

    def search_knn_subset_device(self, d_queries_ptr: int, nq: int, kk: int, d_row_ids_ptr: int, n_ids: int, d_out_dist_ptr: int,
                                 d_out_row_ptr: int, d_out_count_ptr: int | None, stream_ptr: int) -> None:
        _check(lib().rxgpu_search_knn_subset_device(self._h, d_queries_ptr, nq, kk, d_row_ids_ptr, n_ids, d_out_dist_ptr, d_out_row_ptr,
                                                    d_out_count_ptr, stream_ptr))

    def check_row_list_device(self, d_row_ids_ptr: int, n_ids: int, stream_ptr: int) -> bool:
        ok = C.c_int32(0)
        _check(lib().rxgpu_check_row_list_device(self._h, d_row_ids_ptr, n_ids, stream_ptr, C.byref(ok)))
        return bool(ok.value)

    def search_range_subset(self, query, radius: float, row_ids, inclusive: bool = False, cap: int = 1 << 16):
        """SearchRange over the listed rows only -> (dist, row) sorted by (dist, row)."""
        q = _f32c(query).reshape(self.dim)
        ids = np.ascontiguousarray(row_ids, dtype=np.uint32).reshape(-1)
        while True:
            dist = np.empty(max(cap, 1), np.float32)
            row = np.empty(max(cap, 1), np.uint32)
            total = _u64(0)
            rc = lib().rxgpu_search_range_subset(self._h, q.ctypes.data, radius, int(inclusive), ids.ctypes.data if ids.size else None, ids.size,
                                                 dist.ctypes.data, row.ctypes.data, cap, C.byref(total))
            if rc == RXGPU_ERR_OVERFLOW:
                cap = int(total.value)
                continue
            _check(rc)
            return dist[:total.value].copy(), row[:total.value].copy()

    def search_range(self, query, radius: float, inclusive: bool = False, cap: int = 1 << 16):
        q = _f32c(query).reshape(self.dim)
        while True:
            dist = np.empty(cap, np.float32)
            row = np.empty(cap, np.uint32)
            total = _u64(0)
            rc = lib().rxgpu_search_range(self._h, q.ctypes.data, radius, int(inclusive), dist.ctypes.data, row.ctypes.data, cap, C.byref(total))
            if rc == RXGPU_ERR_OVERFLOW:
                cap = int(total.value)
                continue
            _check(rc)
            return dist[: total.value].copy(), row[: total.value].copy()

    def distances(self, query, rows) -> np.ndarray:
        q = _f32c(query).reshape(self.dim)
        r = np.ascontiguousarray(rows, dtype=np.uint32).reshape(-1)
        out = np.empty(r.shape[0], np.float32)
        _check(lib().rxgpu_distances(self._h, q.ctypes.data, r.ctypes.data, r.shape[0], out.ctypes.data))
        return out

    # ---- HNSW
    # ---- IVF: device-resident inverted lists
    def set_lists(self, lists) -> None:
        """lists: sequence of row arrays (disjoint) -> CSR in HBM (rxgpu_index_set_lists)."""
        off = np.zeros(len(lists) + 1, np.uint64)
        off[1:] = np.cumsum([len(l) for l in lists])
        rows = np.ascontiguousarray(np.concatenate([np.asarray(l, np.uint32) for l in lists]) if len(lists) else np.empty(0, np.uint32), np.uint32)
        _check(lib().rxgpu_index_set_lists(self._h, len(lists), off.ctypes.data, rows.ctypes.data if rows.size else None))

    def search_knn_lists(self, coarse: "VectorIndex", query, nprobe: int, k: int):
        """IndexIVFFlat::search in one call: (dist[k'], row[k'], rows scanned)."""
        q = _f32c(query).reshape(self.dim)
        dist = np.full(max(k, 1), np.inf, np.float32)
        row = np.full(max(k, 1), 0xFFFFFFFF, np.uint32)
        cnt = np.zeros(1, np.uint32)
        scanned = _u64(0)
        _check(lib().rxgpu_search_knn_lists(self._h, coarse._h, q.ctypes.data, nprobe, k, dist.ctypes.data, row.ctypes.data, cnt.ctypes.data,
                                            C.byref(scanned)))
        c = int(cnt[0])
        return dist[:c], row[:c], int(scanned.value)

    def search_range_lists(self, coarse: "VectorIndex", query, nprobe: int, radius: float, inclusive: bool = False, cap: int = 1 << 14):
        """IndexIVFFlat::range_search over the device lists in one call: (dist[n], row[n], rows scanned), ascending by (dist, row)."""
        q = _f32c(query).reshape(self.dim)
        while True:
            dist, row = np.empty(max(cap, 1), np.float32), np.empty(max(cap, 1), np.uint32)
            total, scanned = _u64(0), _u64(0)
            rc = lib().rxgpu_search_range_lists(self._h, coarse._h, q.ctypes.data, nprobe, float(radius), int(inclusive), dist.ctypes.data, row.ctypes.data,
                                                cap, C.byref(total), C.byref(scanned))
            if rc == RXGPU_ERR_OVERFLOW:
                cap = int(total.value)
                continue
            _check(rc)
            n = int(total.value)
            return dist[:n].copy(), row[:n].copy(), int(scanned.value)

    def hnsw_attach_graph(self, g: dict) -> None:
        """g: flat graph dict (links0, upper_off, upper, deleted, M, maxM0, maxlevel, entry, num_deleted)."""
        links0 = np.ascontiguousarray(g["links0"], np.uint32)
        upper_off = np.ascontiguousarray(g["upper_off"], np.uint64)
        upper = np.ascontiguousarray(g["upper"], np.uint32)
        deleted = np.ascontiguousarray(g["deleted"], np.uint8)
        blocks = int(upper_off[-1])
        _check(lib().rxgpu_hnsw_attach_graph(self._h, links0.ctypes.data, upper_off.ctypes.data, upper.ctypes.data, blocks, deleted.ctypes.data,
                                             g["M"], g["maxM0"], g["maxlevel"], g["entry"], g["num_deleted"]))

    def hnsw_update_deleted(self, deleted, num_deleted: int) -> None:
        deleted = np.ascontiguousarray(deleted, np.uint8)
        _check(lib().rxgpu_hnsw_update_deleted(self._h, deleted.ctypes.data, num_deleted))

    def hnsw_search_knn(self, queries, k: int, ef: int = 0):
        q = _f32c(queries).reshape(-1, self.dim)
        nq = q.shape[0]
        dist = np.full((nq, max(k, 1)), np.inf, np.float32)
        row = np.full((nq, max(k, 1)), 0xFFFFFFFF, np.uint32)
        cnt = np.zeros(nq, np.uint32)
        _check(lib().rxgpu_hnsw_search_knn(self._h, q.ctypes.data, nq, k, ef, dist.ctypes.data, row.ctypes.data, cnt.ctypes.data))
        return dist[:, :k], row[:, :k], cnt

    def hnsw_search_knn_posted(self, query, k: int, ef: int = 0):
        """One query through the index's resident search kernel (rxgpu_hnsw_search_knn_posted): (dist, row, count, served)."""
        q = _f32c(query).reshape(self.dim)
        dist = np.full(max(k, 1), np.inf, np.float32)
        row = np.full(max(k, 1), 0xFFFFFFFF, np.uint32)
        cnt = _u32(0)
        served = C.c_int32(0)
        _check(lib().rxgpu_hnsw_search_knn_posted(self._h, q.ctypes.data, k, ef, dist.ctypes.data, row.ctypes.data, C.addressof(cnt), C.byref(served)))
        return dist[:k], row[:k], int(cnt.value), bool(served.value)

    def hnsw_read_stats4(self):
        """(distance evaluations, hops, in-kernel restarts, distance trips of the look-ahead team searches) since the last read."""
        v = (_u64 * 4)()
        _check(lib().rxgpu_hnsw_read_stats4(self._h, C.addressof(v)))
        return tuple(int(x) for x in v)

    def hnsw_server_times(self):
        """(microseconds on the device, microseconds at the callers) summed over the queries the mailbox has answered."""
        a, b = _u64(0), _u64(0)
        _check(lib().rxgpu_hnsw_server_times(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def hnsw_server_stats(self):
        a, b = _u64(0), _u64(0)
        _check(lib().rxgpu_hnsw_server_stats(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def hnsw_attach_sq8(self, codes, corr, alpha_2: float) -> None:
        """SQ8 copy of the rows (Quantizer::Quantize, quantizer.h:93-124): codes [count][dim] u8, corr [count] f32."""
        codes = np.ascontiguousarray(codes, np.uint8).reshape(-1, self.dim)
        corr = _f32c(corr).reshape(-1)
        if codes.shape[0] != corr.shape[0]:
            raise ValueError("one corrective offset per code row")
        _check(lib().rxgpu_hnsw_attach_sq8(self._h, codes.ctypes.data, corr.ctypes.data, codes.shape[0], float(alpha_2)))

    def hnsw_search_knn_sq8(self, qcodes, qcorr, qnorm, k: int, ef: int = 0):
        qc = np.ascontiguousarray(qcodes, np.uint8).reshape(-1, self.dim)
        nq = qc.shape[0]
        qcorr = _f32c(qcorr).reshape(-1)
        qnorm = _f32c(qnorm).reshape(-1)
        if qcorr.shape[0] != nq or qnorm.shape[0] != nq:
            raise ValueError("one corrective offset and one normCoef per query")
        dist = np.full((nq, max(k, 1)), np.inf, np.float32)
        row = np.full((nq, max(k, 1)), 0xFFFFFFFF, np.uint32)
        cnt = np.zeros(nq, np.uint32)
        _check(lib().rxgpu_hnsw_search_knn_sq8(self._h, qc.ctypes.data, qcorr.ctypes.data, qnorm.ctypes.data, nq, k, ef, dist.ctypes.data,
                                               row.ctypes.data, cnt.ctypes.data))
        return dist[:, :k], row[:, :k], cnt

    def hnsw_read_stats(self):
        a, b = _u64(0), _u64(0)
        _check(lib().rxgpu_hnsw_read_stats(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def hnsw_read_tie_reruns(self) -> int:
        a = _u64(0)
        _check(lib().rxgpu_hnsw_read_tie_reruns(self._h, C.byref(a)))
        return int(a.value)

    def hnsw_read_lds_reruns(self) -> int:
        a = _u64(0)
        _check(lib().rxgpu_hnsw_read_lds_reruns(self._h, C.byref(a)))
        return int(a.value)

    # ---- instrumentation
    def profile_enable(self, on: bool = True) -> None:
        _check(lib().rxgpu_profile_enable(self._h, int(on)))

    def profile_read(self, name: str) -> tuple[int, float]:
        n, ms = _u64(0), C.c_double(0.0)
        _check(lib().rxgpu_profile_read(self._h, name.encode(), C.byref(n), C.byref(ms)))
        return int(n.value), float(ms.value)


class ShardedVectorIndex(VectorIndex):
    """rxgpu_index_create_sharded: the same index row-range sharded over a device list of THIS process (a device may repeat).  Rows in and out
    are global rows; SearchKnn's per-shard lists meet in one RCCL all-gather on the devices (merge_mode == 1) unless RXGPU_SHARD_MERGE=host
    was set when the index was created."""

    def __init__(self, metric: int | str, dim: int, capacity: int, devices):
        if isinstance(metric, str):
            metric = METRICS[metric.lower()]
        self.metric, self.dim = int(metric), int(dim)
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = _vp()
        _check(lib().rxgpu_index_create_sharded(self.metric, self.dim, capacity, len(devices), devs, C.byref(h)))
        self._h = h
        self._keepalive = None

    @property
    def merge_mode(self) -> str:
        return {1: "rccl", 0: "host"}[lib().rxgpu_index_shard_merge_mode(self._h)]

    @property
    def merge_note(self) -> str:
        """why the index merges on the host ("" in rccl mode)"""
        return (lib().rxgpu_index_shard_merge_note(self._h) or b"").decode()

    @property
    def ranks(self) -> int:
        return lib().rxgpu_index_shard_ranks(self._h)

    @property
    def collectives(self) -> int:
        return lib().rxgpu_index_shard_collectives(self._h)

    @property
    def shard_count(self) -> int:
        return lib().rxgpu_index_shard_count(self._h)

    @property
    def shard_rows(self) -> int:
        return lib().rxgpu_index_shard_rows(self._h)

    def shard(self, s: int) -> "VectorIndex":
        """Shard s as a NON-owning single-device view (adopt_device_rows / profile_enable / profile_read on it)."""
        hs = lib().rxgpu_index_shard(self._h, s)
        if not hs:
            raise RxGpuError("no such shard")
        return _ShardView(self.metric, self.dim, _vp(hs))

    def sync_count(self) -> None:
        _check(lib().rxgpu_index_shard_sync_count(self._h))


class _ShardView(VectorIndex):
    def __init__(self, metric: int, dim: int, handle):
        self.metric, self.dim, self._keepalive = metric, dim, None
        self._h = handle

    def close(self) -> None:   # owned by the sharded index
        self._h = None
        self._keepalive = None


def merge_shards_device(d_gathered_ptr: int, world: int, nq: int, kk: int, shard_rows: int, d_out_dist_ptr: int, d_out_row_ptr: int,
                        d_out_count_ptr: int | None, stream_ptr: int) -> None:
    _check(lib().rxgpu_merge_shards_device(d_gathered_ptr, world, nq, kk, shard_rows, d_out_dist_ptr, d_out_row_ptr, d_out_count_ptr, stream_ptr))


class HybridParams(C.Structure):
    """rxgpu_hybrid_params"""
    _fields_ = [("kind", C.c_int32), ("is_union", C.c_int32), ("desc", C.c_int32), ("reserved", C.c_int32), ("params", C.c_double * 5)]


def hybrid_fuse(kind: str, params, knn_ids, knn_ranks, ft_ids, ft_ranks_u8, union=True, desc=True, metric=METRIC_IP, device: int = 0):
    """rxgpu_hybrid_fuse: the device rank fusion on host arrays.  knn_* best first (the planner's ranks), ft_ids unique in any order with
    their uint8 ranks.  Returns (ids, ranks) in Merged<desc> order."""
    hp = HybridParams()
    hp.kind = 0 if kind == "rrf" else 1
    hp.is_union, hp.desc = int(union), int(desc)
    for i, v in enumerate(params):
        hp.params[i] = float(v)
    ki, kr = np.ascontiguousarray(knn_ids, np.int32), _f32c(knn_ranks)
    fi, fr = np.ascontiguousarray(ft_ids, np.int32), np.ascontiguousarray(ft_ranks_u8, np.uint8)
    cap = ki.shape[0] + fi.shape[0] + 1
    oi, orank = np.empty(cap, np.int32), np.empty(cap, np.float32)
    n = _u64(0)
    _check(lib().rxgpu_hybrid_fuse(device, C.addressof(hp), metric, ki.ctypes.data, kr.ctypes.data, ki.shape[0], fi.ctypes.data, fr.ctypes.data,
                                   fi.shape[0], oi.ctypes.data, orank.ctypes.data, cap, C.byref(n)))
    return oi[:n.value].copy(), orank[:n.value].copy()


def gpu_available() -> bool:
    if os.environ.get("RXGPU_FORCE_NO_GPU"):
        return False
    try:
        return device_count() > 0
    except Exception:
        return False
