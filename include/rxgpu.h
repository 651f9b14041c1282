/*
 * rxgpu — C-ABI of the MI355X (gfx950) engines behind Reindexer's float_vector / ft_fast index plugins.
 *
 * This header is the drop-in boundary: plain C types, caller-allocated outputs, int return codes
 * (0 = ok, negative = rxgpu_status), no C++/torch types.  Nothing in the reference knows this ABI; it is
 * what the GPU `Map` policy of HnswIndexBase<Map> (cpp_src/core/index/float_vector/hnsw_index.h:16-57) and
 * the GPU branch of Selector<IdCont>::mergeResults (cpp_src/core/ft/ft_fast/selecterimpl.h:611-628) bind.
 * Each entry point cites the reference interface it stands in for.  See INTEGRATION.md for the binding.
 *
 * Conventions
 *   - "row" = internal index of a vector inside one index (the reference's `idx` in bruteforce.cc / `tableint`
 *     in hnswalg.h).  Labels (FloatVectorId = rowId<<32 | arrayIdx) never cross this boundary: the host Map
 *     owns the row->label table exactly like the reference owns it inside its AoS rows (bruteforce.h:47-48).
 *   - Distances are "smaller is better": L2 -> squared distance, IP -> -dot, cosine -> -dot * inv_norm[row]
 *     (hnswlib.h:147-165,192-197).  They are bit-identical to the reference's AVX-512 path for every dim.
 *   - All functions are thread-safe for concurrent searches on one index (the reference's read path is
 *     re-entrant under the namespace shared lock); mutations require external exclusion (namespace write lock).
 *   - *_device variants take device pointers and a hipStream_t (as void*), enqueue work and return without
 *     synchronising; everything else synchronises before returning.
 */
#ifndef RXGPU_H
#define RXGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RXGPU_ABI_VERSION 2

typedef enum rxgpu_status {
	RXGPU_OK = 0,
	RXGPU_ERR_PARAMS = -3,    /* reindexer errParams (type_consts.h:139-148) */
	RXGPU_ERR_LOGIC = -4,     /* reindexer errLogic */
	RXGPU_ERR_NOMEM = -5,     /* std::runtime_error("Not enough memory ...") in bruteforce.cc:15,90 */
	RXGPU_ERR_DEVICE = -6,    /* HIP runtime failure; text in rxgpu_last_error() */
	RXGPU_ERR_NOTFOUND = -7,
	RXGPU_ERR_OVERFLOW = -8   /* caller-provided output buffer too small; required size reported */
} rxgpu_status;

typedef enum rxgpu_metric { RXGPU_METRIC_L2 = 0, RXGPU_METRIC_IP = 1, RXGPU_METRIC_COSINE = 2 } rxgpu_metric; /* core/enums.h:101 VectorMetric */

typedef struct rxgpu_index rxgpu_index; /* one float_vector index shard resident on one GPU */

/* Thread-local text of the last failure on the calling thread. */
const char* rxgpu_last_error(void);
int rxgpu_abi_version(void);
/* Number of visible HIP devices; <0 on runtime failure. */
int rxgpu_device_count(void);
/* Fills name with the gcnArchName of `device` ("gfx950:..."), mirrors the SIMD-level log line of hnsw_index.cc:25-44. */
int rxgpu_device_arch(int device, char* name, size_t cap);

/* ---------------------------------------------------------------------------------------------------------
 * Vector storage (device mirror of BruteforceSearch / HierarchicalNSW row storage)
 * ------------------------------------------------------------------------------------------------------- */

/* BruteforceSearch::BruteforceSearch(metric, dim, maxElements)  bruteforce.cc:11-18.
 * Allocates capacity * row_stride floats of HBM on `device` (row_stride = dim rounded up to 4 floats). */
int rxgpu_index_create(int metric, uint32_t dim, uint64_t capacity, int device, rxgpu_index** out);
void rxgpu_index_destroy(rxgpu_index* h);

/* The same index row-range SHARDED over several GPUs of this process (BASELINE configs[3]: the Map owns a device list; `devices` may
 * repeat a device — several shards on one GPU).  Shard s holds the global rows [s * shard_rows, (s + 1) * shard_rows), shard_rows =
 * ceil(capacity / n_devices) rounded up to 32.  The handle is used like any other with rxgpu_index_upload_rows (no holes: first_row <= count)
 * / move_row / truncate / count / capacity / device_bytes, rxgpu_search_knn, rxgpu_search_knn_subset, rxgpu_search_range,
 * rxgpu_search_range_subset and rxgpu_distances: rows in and out are GLOBAL rows, results are the single-device results bit for bit
 * (per-shard exact lists merged under (dist, global row), i.e. BruteforceSearch's scan order, bruteforce.cc:103-127), every shard's
 * kernels run concurrently on their own device.  The capacity is fixed (create a new index to grow); device-pointer entry points,
 * bitmap filters, SQ8, streaming sessions and profiling are single-device only and return RXGPU_ERR_LOGIC here.
 * HNSW over shards (SURVEY 8e "HNSW": "per-shard independent graphs + the same all-gather merge"): every shard holds the graph of ITS rows —
 * rows, graph, patches and delete marks go to the rxgpu_index_shard(h, s) handles (local row ids), each of which may hold any number of rows
 * up to rxgpu_index_shard_rows — and rxgpu_hnsw_search_knn / rxgpu_hnsw_search_range / rxgpu_hnsw_read_* on the sharded handle fan out:
 * every shard runs HierarchicalNSWImpl::SearchKnn (hnswalg.h:1988-2012) over its graph concurrently, the per-shard results stay in HBM and
 * meet in the same ncclAllGather + (dist, global row) merge as brute force (k <= 64; host merge above that or in host mode).  The merged
 * list of a query = the k best of the union of the per-shard results, sorted, rows GLOBAL (shard * shard_rows + local row). */
int rxgpu_index_create_sharded(int metric, uint32_t dim, uint64_t capacity, uint32_t n_devices, const int* devices, rxgpu_index** out);
uint32_t rxgpu_index_shard_count(const rxgpu_index* h);   /* 0 for an unsharded index */
uint64_t rxgpu_index_shard_rows(const rxgpu_index* h);
/* How SearchKnn's per-shard lists meet (north_star: "RCCL all-gather of per-shard top-k over xGMI"): 1 = on the devices — the index owns one
 * RCCL communicator over its distinct devices (ncclCommInitAll at creation), each search is the shards' scans, ONE ncclAllGather of
 * kk x 8 B x nq per shard on the shards' streams, the (dist, global row) merge kernel on the first device and one D2H copy; 0 = on the host
 * (RXGPU_SHARD_MERGE=host in the environment at creation: D2H per shard + host merge — also what range searches, pre-filtered searches and
 * kk > 64 use in either mode); -1 = not a sharded index.  RCCL is opened when the first sharded index asks for it (dlopen: a single-GPU
 * deployment does not need the library); if it is missing or the communicator cannot be built (no peer access, no /dev/shm, ...) the index is
 * still created, in host mode — one line on stderr, and rxgpu_index_shard_merge_note says why ("" in device mode). */
int rxgpu_index_shard_merge_mode(const rxgpu_index* h);
const char* rxgpu_index_shard_merge_note(const rxgpu_index* h);
uint32_t rxgpu_index_shard_ranks(const rxgpu_index* h);         /* RCCL ranks = distinct devices of the shard list (0 in host mode) */
uint64_t rxgpu_index_shard_collectives(const rxgpu_index* h);   /* all-gathers issued so far (tests / bench assert the path that ran) */
/* Shard s as an ordinary single-device index, for filling it in place (rxgpu_index_adopt_device_rows with memory of THAT device; the
 * sharded handle then takes the row counts over with rxgpu_index_shard_sync_count: full shards, then at most one partial, then empty). */
rxgpu_index* rxgpu_index_shard(rxgpu_index* h, uint32_t s);
int rxgpu_index_shard_sync_count(rxgpu_index* h);
/* One row back to the host (and its 1/|row| for cosine; out_inv_norm may be NULL): the cross-device half of a sharded swap-delete. */
int rxgpu_index_download_row(rxgpu_index* h, uint64_t row, float* out_row, float* out_inv_norm);

/* BruteforceSearch::ResizeIndex  bruteforce.cc:88-101 (contents preserved; shrinking below count is an error). */
int rxgpu_index_reserve(rxgpu_index* h, uint64_t capacity);

/* Device side of AddPointNoLock (bruteforce.cc:44-64): copy n host rows (packed [n][dim]) to rows
 * [first_row, first_row+n) and their 1/|row| coefficients (cosine only, may be NULL otherwise; the host computes
 * them exactly as DistCalculator::AddNorm does, hnswlib.h:80-92).  count becomes max(count, first_row+n). */
int rxgpu_index_upload_rows(rxgpu_index* h, uint64_t first_row, uint64_t n, const float* rows, const float* inv_norms);

/* Zero-copy alternative: adopt caller-owned device memory (e.g. a torch tensor) as the row storage.
 * d_rows: [n][row_stride] floats, row_stride >= dim, row_stride % 4 == 0, 16-byte aligned; d_inv_norms: [n] or NULL.
 * The caller keeps ownership and must keep the memory alive and unchanged while adopted. */
int rxgpu_index_adopt_device_rows(rxgpu_index* h, const void* d_rows, uint64_t n, uint32_t row_stride, const void* d_inv_norms);

/* Device side of RemovePoint's swap-with-last (bruteforce.cc:70-86): copy row `from` over row `to` (incl. norm). */
int rxgpu_index_move_row(rxgpu_index* h, uint64_t from, uint64_t to);
/* Set the number of live rows (after a swap-delete, or Reset()). */
int rxgpu_index_truncate(rxgpu_index* h, uint64_t count);

uint64_t rxgpu_index_count(const rxgpu_index* h);
uint64_t rxgpu_index_capacity(const rxgpu_index* h);
uint32_t rxgpu_index_dim(const rxgpu_index* h);
uint32_t rxgpu_index_row_stride(const rxgpu_index* h);
int rxgpu_index_metric(const rxgpu_index* h);
int rxgpu_index_device(const rxgpu_index* h);
/* HBM bytes held by the index (AllocatedMemSize analogue, bruteforce.h:41-44). */
uint64_t rxgpu_index_device_bytes(const rxgpu_index* h);

/* ---------------------------------------------------------------------------------------------------------
 * Brute-force search (BruteforceSearch::SearchKnn / SearchRange)
 * ------------------------------------------------------------------------------------------------------- */

/* Exact top-kk of each of nq queries under the total order (dist, row) ascending — the order in which the
 * reference's (dist,label) max-heap evicts, with row standing in for label (bruteforce.cc:103-127).
 * queries: host [nq][dim] (cosine: already normalised by the caller, hnsw_index.cc:166-171).
 * out_dist/out_row: host [nq][kk]; out_count[q] = min(kk, count).
 * The GPU Map asks for kk = k+1 so it can detect a distance tie straddling the k-th boundary and replay the
 * reference's admission rule (strict `dist < worst`, bruteforce.cc:121) with rxgpu_search_range(inclusive = 1). */
int rxgpu_search_knn(rxgpu_index* h, const float* queries, uint32_t nq, uint32_t kk, float* out_dist, uint32_t* out_row,
					 uint32_t* out_count);

/* internal row -> row id (label >> 32, FloatVectorId::RowId) table kept beside the rows for consumers on the device: the hybrid fusion
 * translates the scan's rows there.  Rows [first_row, first_row + n); the table is sized by the index capacity.  The device pointer
 * (int32 [capacity]) or NULL when nothing was uploaded. */
int rxgpu_index_upload_row_ids(rxgpu_index* h, uint64_t first_row, uint64_t n, const int32_t* row_ids);
const void* rxgpu_index_row_ids_device(const rxgpu_index* h);
/* One host query searched with the result LEFT IN HBM (the hybrid query's KNN half): the search is enqueued on a stream of the
 * index and the call returns without waiting; *d_dist / *d_row ((dist, row) best first, *entries of them) and *d_count (device uint32)
 * are buffers of the index that hold this result until the CALLING THREAD's next resident search on it (every thread has buffers and a
 * stream of its own there); *stream is the stream it runs on — a consumer
 * (rxgpu_hybrid_fuse_resident) orders itself behind it on the device.  kk in [1, 128]. */
int rxgpu_search_knn_resident(rxgpu_index* h, const float* query, uint32_t kk, void** d_dist, void** d_row, void** d_count, void** stream,
							  uint32_t* entries);
/* Diagnostics: how many threads hold a resident context (stream + result buffers) on the index right now.  A thread that ends gives its
 * context back to the index's pool, so the number follows the LIVE searching threads, not every thread that ever searched. */
uint32_t rxgpu_index_resident_contexts(rxgpu_index* h);
/* Same, device-resident in/out on `stream` (hipStream_t); d_out_count may be NULL.  No synchronisation. */
int rxgpu_search_knn_device(rxgpu_index* h, const void* d_queries, uint32_t nq, uint32_t kk, void* d_out_dist, void* d_out_row,
							void* d_out_count, void* stream);

/* ---- Pre-filtered brute force: the caller side of `WHERE cond AND KNN(...)` -------------------
 * The reference evaluates such a query by taking the KNN result and filtering it on the host (selectLoop,
 * nsselecter.cc:841-875); only HNSW can stream more candidates (knn_streaming_index_iterator.cc).  Here the rows that
 * passed `cond` are handed to the scan, which then reads ONLY those rows: the result is exactly what
 * BruteforceSearch::SearchKnn (bruteforce.cc:103-127) returns over an index that holds just the allowed rows —
 * same distance bits, same (dist, row) order — and HBM traffic shrinks with the selectivity of the filter.
 *
 * rxgpu_search_knn_subset: row_ids = host [n_ids], strictly increasing internal rows (the shape of the selector's
 *   sorted IdSet, core/idset.h); best for selective filters (4 bytes per allowed row on the wire).
 * rxgpu_search_knn_bitmap: allowed_words = host [n_words >= ceil(count / 32)], bit (r % 32) of word r / 32 set = row r
 *   allowed (bits at and above `count` are ignored); expanded to the row list on the device; best for dense filters
 *   (count / 8 bytes on the wire).  *out_allowed (optional) = number of allowed rows.
 * Output layout as rxgpu_search_knn: host [nq][kk], out_count[q] = min(kk, allowed rows); any kk.
 * Errors: RXGPU_ERR_PARAMS for an unsorted / out-of-range list or a short bitmap. */
int rxgpu_search_knn_subset(rxgpu_index* h, const float* queries, uint32_t nq, uint32_t kk, const uint32_t* row_ids, uint64_t n_ids,
							float* out_dist, uint32_t* out_row, uint32_t* out_count);
int rxgpu_search_knn_bitmap(rxgpu_index* h, const float* queries, uint32_t nq, uint32_t kk, const uint32_t* allowed_words,
							uint64_t n_words, float* out_dist, uint32_t* out_row, uint32_t* out_count, uint64_t* out_allowed);

/* ---- IVF-Flat: faiss::IndexIVFFlat::search as IvfIndex drives it (ivf_index.cc:355-372) in ONE call ----------
 * rxgpu_index_set_lists: the inverted lists over this index's rows as CSR — list_off [nlist + 1] (list_off[0] = 0), list_rows
 *   [list_off[nlist]] internal rows (lists are disjoint); uploaded and kept until the next call.  Set them again after the rows change.
 * rxgpu_search_knn_lists: `coarse` = a flat index of the nlist centroids (same metric family, same device).  The nprobe nearest centroids
 *   are found on the device, their lists are marked in an allowed-rows bitmap, expanded to the ascending row list and scanned
 *   (rxgpu_search_knn_subset's kernels) without a host round trip in between: same result as rxgpu_search_knn_subset over the union of
 *   the probed lists.  query = host [dim], already prepared for the metric (cosine: normalised).  Any nprobe (clamped to nlist): up to
 *   128 lists the coarse result never leaves HBM, wider probes fetch the nprobe list ids and send them back.  *out_scanned (optional)
 *   = rows in the probed lists.  Output as rxgpu_search_knn for nq = 1.
 * rxgpu_search_range_lists: IndexIVFFlat::range_search (ivf_index.cc:212-272 drives it) over the same device lists — probed lists ->
 *   row list on the device -> the range kernel of rxgpu_search_range_subset; output, overflow protocol and ordering as
 *   rxgpu_search_range_subset. */
int rxgpu_index_set_lists(rxgpu_index* h, uint32_t nlist, const uint64_t* list_off, const uint32_t* list_rows);
int rxgpu_search_knn_lists(rxgpu_index* h, rxgpu_index* coarse, const float* query, uint32_t nprobe, uint32_t kk, float* out_dist,
						   uint32_t* out_row, uint32_t* out_count, uint64_t* out_scanned);
int rxgpu_search_range_lists(rxgpu_index* h, rxgpu_index* coarse, const float* query, uint32_t nprobe, float radius, int inclusive, float* out_dist,
							 uint32_t* out_row, uint64_t cap, uint64_t* out_total, uint64_t* out_scanned);

/* Device-resident variant on `stream` (no synchronisation): d_row_ids = device [n_ids] uint32, 1 <= n_ids <= count,
 * kk in [1, 128]; d_out_count may be NULL.  The list is TRUSTED (strictly increasing, below count): an id beyond the
 * index would fault the device.  rxgpu_check_row_list_device() verifies a device list (synchronises `stream`). */
int rxgpu_search_knn_subset_device(rxgpu_index* h, const void* d_queries, uint32_t nq, uint32_t kk, const void* d_row_ids,
								   uint64_t n_ids, void* d_out_dist, void* d_out_row, void* d_out_count, void* stream);
int rxgpu_check_row_list_device(rxgpu_index* h, const void* d_row_ids, uint64_t n_ids, void* stream, int32_t* out_ok);

/* SearchRange over a row list (same list contract as rxgpu_search_knn_subset): every listed row with dist < radius (<= when
 * `inclusive`), sorted by (dist, row); *out_total = number of hits; RXGPU_ERR_OVERFLOW when it exceeds `cap` (call again with
 * cap >= *out_total).  Stands in for the list scan of faiss::IndexIVFFlat::range_search as the reference's IvfIndex calls it
 * (ivf_index.cc:212-272) — the probed inverted lists are the row list. */
int rxgpu_search_range_subset(rxgpu_index* h, const float* query, float radius, int inclusive, const uint32_t* row_ids, uint64_t n_ids,
							  float* out_dist, uint32_t* out_row, uint64_t cap, uint64_t* out_total);

/* Multi-GPU merge step (no reference counterpart: the reference has no device notion).  d_gathered = the all-gather of every
 * rank's search output for one query batch: [world][2][nq][kk] 32-bit words — per rank the [nq][kk] distances followed by the
 * [nq][kk] shard-local rows (what rxgpu_search_knn_device writes when d_out_row == d_out_dist + nq*kk words).
 * Writes the global top-kk under (dist, global row = shard*shard_rows + local row): d_out_dist [nq][kk], d_out_row [nq][kk] u32.
 * Pure device work on `stream`, no synchronisation. */
int rxgpu_merge_shards_device(const void* d_gathered, uint32_t world, uint32_t nq, uint32_t kk, uint32_t shard_rows, void* d_out_dist,
							  void* d_out_row, void* d_out_count, void* stream);

/* BruteforceSearch::SearchRange (bruteforce.cc:129-143): every row with dist < radius (inclusive != 0: dist <= radius;
 * the inclusive form serves the tie replay above).  Rows are returned sorted by (dist,row); *out_total receives the
 * number of hits; at most cap are written; RXGPU_ERR_OVERFLOW if out_total > cap (call again with a larger buffer). */
int rxgpu_search_range(rxgpu_index* h, const float* query, float radius, int inclusive, float* out_dist, uint32_t* out_row,
					   uint64_t cap, uint64_t* out_total);

/* Distances of one query against an explicit list of rows (DistCalculator::operator()(q,row,id), hnswlib.h:147-165);
 * used for exact re-scoring and by tests. */
int rxgpu_distances(rxgpu_index* h, const float* query, const uint32_t* rows, uint32_t n, float* out_dist);

/* ---------------------------------------------------------------------------------------------------------
 * HNSW search (HierarchicalNSWImpl<float>::SearchKnn, hnswalg.h:1988-2012) over a host-built graph
 * ------------------------------------------------------------------------------------------------------- */

/* Mirror the graph of an HNSW index into HBM.  The vectors are the index rows (rxgpu_index_upload_rows, internal id == row);
 * the graph is the flat form of HierarchicalNSWImpl's storage (hnswalg.h:216-240), built on the host by the GPU Map:
 *   links0     u32 [count][1 + max_m0]   slot 0 = neighbour count, then the level-0 links in stored order
 *   upper      u32 [upper_blocks][1 + M] per-level lists of the upper layers; node i owns its `level_i` blocks starting at
 *   upper_off  u64 [count + 1]           block upper_off[i] (level 1 first)
 *   deleted    u8  [count]               DELETE_MARK flags (hnswalg.h:1366-1374)
 * entry / maxlevel = enterpoint_node_ / maxlevel_; num_deleted selects the bare-bone search (hnswalg.h:1982).
 * Limits of the GPU engine: max_m0 <= 128 (M <= 64). */
int rxgpu_hnsw_attach_graph(rxgpu_index* h, const uint32_t* links0, const uint64_t* upper_off, const uint32_t* upper, uint64_t upper_blocks,
							const uint8_t* deleted, uint32_t M, uint32_t max_m0, int32_t maxlevel, uint32_t entry, uint64_t num_deleted);
/* In-place mirror of a host-side insert or update (addPoint / updatePoint, hnswalg.h:1472-1852, touch the new element and a few dozen
 * neighbours): instead of re-uploading the whole graph, the nodes whose lists changed are scattered into the resident arrays.  Call after
 * rxgpu_index_upload_rows for the new rows.  dirty_ids [n_dirty]: every node with a changed level-0 list, upper list or delete flag; new
 * nodes (id >= the attached node count) all listed, ascending; links0_rows [n_dirty][1 + max_m0]; deleted_flags [n_dirty]; levels [n_dirty]
 * = the node's level (upper blocks); upper_rows = those blocks concatenated in dirty order, (1 + M) words each.
 * RXGPU_ERR_OVERFLOW: the arrays allocated by the last attach cannot take the growth — attach the graph again. */
int rxgpu_hnsw_patch_graph(rxgpu_index* h, uint32_t n_dirty, const uint32_t* dirty_ids, const uint32_t* links0_rows, const uint8_t* deleted_flags,
						   const int32_t* levels, const uint32_t* upper_rows, int32_t maxlevel, uint32_t entry, uint64_t num_deleted);
/* MarkDelete mirror (hnswalg.h:1303-1339): refresh only the flags. */
int rxgpu_hnsw_update_deleted(rxgpu_index* h, const uint8_t* deleted, uint64_t num_deleted);

/* SearchKnn for nq queries (host in/out).  ef == 0 means k*3/2 like the engine (hnswalg.h:1995); ef <= 4096
 * (RXGPU_ERR_PARAMS above that; from ef = 1025 on the candidate heap lives in global scratch instead of LDS).
 * Writes, per query, the members of the reference's top_candidates after trimming to k (UNORDERED: the Map pushes them into the
 * (dist,label) result heap exactly as hnswalg.h:2002-2010 does): out_dist/out_row [nq][k], out_count[q] <= k.
 * The traversal replays the reference's heaps step for step, so on the same graph the sets are identical. */
int rxgpu_hnsw_search_knn(rxgpu_index* h, const float* queries, uint32_t nq, uint32_t k, uint32_t ef, float* out_dist, uint32_t* out_row,
						  uint32_t* out_count);
/* ONE query through the index's RESIDENT search kernel — the planner's call: HnswIndexBase::select issues one SearchKnn per query from as
 * many threads as there are connections (hnsw_index.cc:159-288; gtests/tests/unit/float_vector_index.cc:258-294 runs 16 of them).  The call
 * claims a slot of a mailbox in pinned host memory, stores the query and polls the answer; the kernel behind the mailbox is launched by
 * whichever caller finds none alive, serves every thread's requests side by side (one workgroup of four wavefronts per slot) and ends by
 * itself — when the index is about to change (every mutating entry point tells it), after RXGPU_HNSW_SERVER_IDLE_US (2000) without a
 * request, after RXGPU_HNSW_SERVER_LIFE_MS (50) in any case.  No launch, copy or completion signal lies on the path of a served query.
 * *served = 1: out_* hold what rxgpu_hnsw_search_knn returns for this query (the same device code runs the search).  *served = 0: this query
 * is not taken — every slot (RXGPU_HNSW_SERVER_SLOTS, 256) is busy, ef > 256 (224 on a graph with deleted nodes; up to 128 / 96 and above
 * that are served by two resident kernels of their own), a dimension other than
 * 128 / 512 / 768, a profiled or sharded index, RXGPU_HNSW_SERVER=0, or the search needs the re-run tiers — and the caller uses
 * rxgpu_hnsw_search_knn, which tries the mailbox itself for nq == 1 and launches otherwise. */
int rxgpu_hnsw_search_knn_posted(rxgpu_index* h, const float* query, uint32_t k, uint32_t ef, float* out_dist, uint32_t* out_row, uint32_t* out_count,
								 int32_t* served);
/* The search kernels' own counters since the last read (reset by the call, like rxgpu_hnsw_read_stats): out4[0] = distance evaluations the
 * traversal used, [1] = hops, [2] = searches that started over on the reference's heaps inside the kernel, [3] = two 32-bit counts of the team
 * searches (small launches and the resident kernel): high half = hops that found their link block in LDS because it had come along with the
 * previous hop's rows; low half = distance trips of the look-ahead experiment (RXGPU_HNSW_SPEC=1, off by default; hnsw_search_core.hip.h). */
int rxgpu_hnsw_read_stats4(rxgpu_index* h, uint64_t* out4);
/* queries answered through the mailbox / resident kernels launched so far (instrumentation; 0 / 0 before the first such query) */
int rxgpu_hnsw_server_stats(rxgpu_index* h, uint64_t* served, uint64_t* generations);
/* ... and where their time went, summed over the queries answered so far: on the device (the search itself, by the kernel's wall clock) and at
 * the caller (from the store of the request until the answer was seen: the search + the mailbox + the caller's wake-up) — microseconds */
int rxgpu_hnsw_server_times(rxgpu_index* h, uint64_t* device_us, uint64_t* caller_us);

/* SQ8 graphs — HierarchicalNSWImpl<uint8_t> after HnswIndexBase::Quantize (hnsw_index.cc:626-660, hnswalg.h:353-420): the level-0 payload
 * is one byte per component plus the row's corrective offset, distances are DistCalculator<uint8_t>::operator() (hnswlib.h:147-165) over
 * L2SqrDistance<uint8_t> / InnerProductDistance<uint8_t> (tools/distances/l2_dist.cc:168-199, ip_dist.cc:163-192):
 *   alpha_2 * sum + corr(query) + corr(row)   (negated for IP / cosine, times 1/|row| for cosine), times the query's normCoef.
 * The summation order of the reference's AVX-512 form is part of the contract and is reproduced on the device (v_dot4_u32_u8 partial
 * sums re-associated into the 16 zmm lanes), so results are bit-identical to the quantised engine on the same graph.
 * codes [count][dim] u8, corr [count] (Quantizer::Quantize + CorrectiveOffsets, quantizer.h:93-124); count == the index row count.
 * Attach after rxgpu_hnsw_attach_graph / upload; the float rows of the index are not read by the SQ8 search. */
int rxgpu_hnsw_attach_sq8(rxgpu_index* h, const uint8_t* codes, const float* corr, uint64_t count, float alpha_2);
/* Rows [first_row, first_row + n) of the code table (first_row <= rows that have codes; first_row + n <= count): the device side of a point
 * added to or updated in a quantised graph (addPoint with a quantizer, hnswalg.h:1480-1495) — D + 4 bytes per row instead of the whole
 * table again.  The table is sized by the index capacity. */
int rxgpu_hnsw_upload_sq8_rows(rxgpu_index* h, uint64_t first_row, uint64_t n, const uint8_t* codes, const float* corr, float alpha_2);
/* SearchKnn on the SQ8 graph.  The caller quantises the queries the way prepareData does (hnswalg.h:510-529): query_codes [nq][dim],
 * query_corr [nq], query_norm_coef [nq] (1 for L2 / IP, queryNormCoef for cosine, hnswalg.h:1855-1863).  Same outputs as
 * rxgpu_hnsw_search_knn. */
int rxgpu_hnsw_search_knn_sq8(rxgpu_index* h, const uint8_t* query_codes, const float* query_corr, const float* query_norm_coef, uint32_t nq,
							  uint32_t k, uint32_t ef, float* out_dist, uint32_t* out_row, uint32_t* out_count);
/* Counters accumulated since the last call (for the roofline accounting: bytes = evals*dim*4 + hops*(1+2M)*4). */
int rxgpu_hnsw_read_stats(rxgpu_index* h, uint64_t* distance_evals, uint64_t* hops);
/* Searches with ef <= 256 (<= 224 when the graph has deleted nodes) keep top_candidates + candidate_set (hnswalg.h:741-777) as one sorted
 * list in registers; a search in which EQUAL distances could change what the reference's binary heaps do (CompareByFirst,
 * hnswalg.h:581-585), or whose list runs out of registers, starts over on the kernel that replays those heaps.  Reads and resets the
 * number of such restarts.  RXGPU_HNSW_SORTED=0 in the environment keeps every search on the heap kernel. */
int rxgpu_hnsw_read_tie_reruns(rxgpu_index* h, uint64_t* reruns);
/* A search whose candidate_set (hnswalg.h:741-777, unbounded in the reference) outgrows the LDS area of its first pass is run again on the
 * heap kernel with the largest LDS heap (2048 entries) before the global-scratch tiers are tried.  Reads and resets the number of such
 * re-runs. */
int rxgpu_hnsw_read_lds_reruns(rxgpu_index* h, uint64_t* reruns);

/* HierarchicalNSW::SearchRange (hnswalg.h:2015-2070) with BOTH halves on the device: the ef-search, then the closure of its hits over the
 * level-0 links while dist < radius (a fresh visited set in which only the ef hits are marked; deleted neighbours skipped) in ONE launch.
 * Hits come back unordered ((dist, row), dist as the engine reports it: negated similarity for inner product / cosine).  cap too small:
 * RXGPU_ERR_OVERFLOW with *out_total = the hits counted so far (a lower bound) — retry with more room.  The _sq8 form runs over the
 * attached codes with the query as prepareData leaves it. */
int rxgpu_hnsw_search_range(rxgpu_index* h, const float* query, float radius, uint32_t ef, float* out_dist, uint32_t* out_row, uint64_t cap,
							uint64_t* out_total);
int rxgpu_hnsw_search_range_sq8(rxgpu_index* h, const uint8_t* query_codes, float query_corr, float query_norm_coef, float radius, uint32_t ef,
								float* out_dist, uint32_t* out_row, uint64_t cap, uint64_t* out_total);
/* Streaming (batched) KNN over the attached graph: HierarchicalNSWImpl::BeginStreamingSearch / ContinueStreamingSearch
 * (cpp_src/core/index/float_vector/hnswlib/hnswalg.h:1865-1975; interface hnsw_interface.h:18-45, 99-102).  The session owns its
 * Layer0SearchState (candidate_set, top_candidates, top_candidates_extras, visited set) in device memory.  query: [dim] floats, already
 * normalised for cosine (hnsw_index.cc:303-314); ef == 0 -> 100.  Every continue(batch) returns <= batch (dist, internal row) pairs,
 * worst first like emitStreamingBatch pops them, and *exhausted as the reference reports it.  The graph must not change during a
 * session (the reference requires an external read lock for the same reason). */
typedef struct rxgpu_hnsw_stream rxgpu_hnsw_stream;
int rxgpu_hnsw_stream_begin(rxgpu_index* h, const float* query, uint32_t ef, rxgpu_hnsw_stream** out);
/* The same session over the attached SQ8 codes (HierarchicalNSWImpl<uint8_t>): query codes + corrective offset + normCoef as for
 * rxgpu_hnsw_search_knn_sq8; rxgpu_hnsw_stream_continue / _end as above. */
int rxgpu_hnsw_stream_begin_sq8(rxgpu_index* h, const uint8_t* query_codes, float query_corr, float query_norm_coef, uint32_t ef, rxgpu_hnsw_stream** out);
int rxgpu_hnsw_stream_continue(rxgpu_hnsw_stream* s, uint32_t batch, float* out_dist, uint32_t* out_row, uint32_t* out_count, int32_t* exhausted);
void rxgpu_hnsw_stream_end(rxgpu_hnsw_stream* s);

/* ---------------------------------------------------------------------------------------------------------
 * ft_fast BM25 score accumulation (ft::Merger<..>::mergeSimple, cpp_src/core/ft/ft_fast/mergerimpl.h:194-250)
 * ------------------------------------------------------------------------------------------------------- */

typedef struct rxgpu_ft_index rxgpu_ft_index; /* device mirror of one ft_fast DataHolder: vdoc statistics + flattened postings */

/* FTConfig fields read by the merge (cpp_src/core/ft/config/ftconfig.h:118-124,151-220); per-field arrays have num_fields entries. */
typedef struct rxgpu_ft_config {
	double bm25_k1, bm25_b;                    /* Bm25Config: 2.0, 0.75 */
	double summation_ranks_by_fields_ratio;    /* 0.0 */
	double full_match_boost;                   /* 1.1; addFullMatchBoost (merger.h:100-109) is applied on the device (the word counts are resident) */
	int32_t min_rank;                          /* 5   (host) */
	uint32_t merge_limit;                      /* 20000 */
	uint32_t num_fields;
	const double *bm25_boost, *bm25_weight, *term_len_boost, *term_len_weight, *position_boost, *position_weight;
	double distance_boost, distance_weight;    /* 1.0, 0.5 (ftconfig.h:180-181); read by the multi-term merge only */
	int32_t bm25_type;                         /* FTConfig::Bm25Config::bm25Type (ftconfig.h:199-206; calculators in core/ft/bm25.h:8-68):
	                                              0 = rx (the default), 1 = classic, 2 = wordCount.  ABI 2. */
} rxgpu_ft_config;

/* FtDslOpts of the query term (cpp_src/core/ft/ftdsl.h:13-35).  At most 4096 sub-terms per term on the GPU engine. */
typedef struct rxgpu_ft_term_opts {
	float boost, term_len_boost;
	const float* field_boost;       /* FtDslFieldOpts::boost per field */
	const uint8_t* need_sum_rank;   /* FtDslFieldOpts::needSumRank per field (at most 8 set on the GPU engine) */
} rxgpu_ft_term_opts;

int rxgpu_ft_create(uint32_t num_fields, int device, rxgpu_ft_index** out);
void rxgpu_ft_destroy(rxgpu_ft_index* h);
/* The same index cut into DOCUMENT-RANGE shards over a device list (SURVEY 8e "BM25": "shard by doc-id range (each GPU holds the posting
 * fragments of its docs; idf uses global N and df ...); exchange = ... the uint16 pre-score histogram for the global threshold"): contiguous
 * runs of 8192-document ranges, one run per listed device (1..64 entries; a device may repeat).  rxgpu_ft_set_docs replicates the per-document
 * statistics and fixes the cut (call it BEFORE the words); rxgpu_ft_set_word / rxgpu_ft_set_word_positions hand every shard the postings of
 * its documents with the whole list's length as document frequency.  rxgpu_ft_merge_simple_raw / _terms_raw / _query_raw (without phrases)
 * run the ordinary launch train on every shard at once and exchange, between its kernels, what spans the shards — every shard's pre-score
 * histogram + mask popcount (the 2-phase gate and preselectMostRelevantDocs' threshold, mergerimpl.h:386-464, with the ties at the threshold
 * handed out in document order) and every shard's table of first-met documents per (sub-term row, range) (the merge slots of addDoc,
 * merger.h:161-180, and the cut at maxMergedDocs) — one all-gather each, stream-ordered between the shards' kernels with no host round trip:
 * an ncclAllGather over the listed devices (RCCL opened on demand; one communicator per index), a plain device copy when every shard lives on
 * one device; without RCCL on a multi-device node, or with RXGPU_SHARD_MERGE=host at creation, the pieces travel through the host
 * (rxgpu_ft_shard_exchange_mode 1 = on the devices, 0 = through the host).  The result is the single
 * index's, bit for bit, in merge order.  rxgpu_ft_merge_query2_raw's multi-word synonyms run there too (a synonym's mask, the term
 * counting and the removal of documents that hold only parts of it, mergerimpl.h:347-361 / 509-555, are decided per document, and a
 * document lies in one shard; the marked documents go after the union of the shards' slots), and so do phrases: PhraseMerger runs on every
 * shard over its fragments — a phrase is decided inside a document —, the rows of a phrase are numbered alike on every shard, the admission
 * cut of the whole index (at most merge_limit candidates of the first term in (row, document) order, phrasemerger.h:341) is settled between
 * the shards' admission passes, NumDocsMerged() in the 2-phase estimate is the sum.
 * rxgpu_ft_merge_query_areas_raw works there as well (a document's areas are built by the shard that holds it, at its global merge slot).
 * rxgpu_ft_merge_batch_raw runs its merges one after the other there.  Packed uploads (the host decodes, GpuFtMerger::SetWordsPacked) and
 * resident (hybrid) merges are single-device features: RXGPU_ERR_LOGIC here. */
int rxgpu_ft_create_sharded(uint32_t num_fields, uint32_t n_devices, const int* devices, rxgpu_ft_index** out);
uint32_t rxgpu_ft_shard_count(const rxgpu_ft_index* h);           /* 0 for an unsharded index */
/* The cut of a sharded index is fixed by the first rxgpu_ft_set_docs and kept while the shards hold words: an index that grows through
 * step commits (a larger total_docs, only the changed words uploaded again) keeps its fragments, the new document ranges go to the last
 * shard.  Ranges of the fullest shard / ranges of an even cut; 1.0 for an even cut or an unsharded index. */
double rxgpu_ft_shard_imbalance(const rxgpu_ft_index* h);
int rxgpu_ft_shard_exchange_mode(const rxgpu_ft_index* h);        /* 1 on the devices, 0 through the host, -1 not sharded */
uint64_t rxgpu_ft_shard_collectives(const rxgpu_ft_index* h);     /* all-gathers issued so far */
int rxgpu_ft_shard_ranges(const rxgpu_ft_index* h, uint32_t shard, uint32_t* range_begin, uint32_t* range_count);   /* the shard's run of 8192-document ranges */
/* Documents that hold the word over the WHOLE index (a sharded handle: the length of the whole list, not of a shard's fragment); 0 for a
 * word the index does not know (MaxVDocs of a sub-term: what the merge limits and the 2-phase estimate are computed from). */
int rxgpu_ft_word_df(rxgpu_ft_index* h, uint32_t word_id, uint64_t* out_df);
/* The DocsStatsGetter of IndexText (cpp_src/core/index/indextext/indextext.h:245-258): total_docs counts the empty sentinel
 * vdoc 0 ("first doc is always empty"); words_in_field [total_docs][num_fields] = VDoc::wordCounts_; avg_words [num_fields];
 * removed [total_docs] or NULL. */
int rxgpu_ft_set_docs(rxgpu_ft_index* h, uint64_t total_docs, const float* words_in_field, const float* avg_words, const uint8_t* removed);
/* Posting list of one dictionary word (IdRelVec / PackedIdRelVec, cpp_src/core/ft/idrelset.h:62-294) flattened to SoA: ascending
 * vdoc ids; per posting a run [ent_off[i], ent_off[i+1]) of (field, occurrences in that field, first position in that field),
 * fields ascending — exactly what calcTermRankImpl derives from IdRelType::Pos() (phrasemergerimpl.h:24-49). */
int rxgpu_ft_set_word(rxgpu_ft_index* h, uint32_t word_id, uint64_t n, const uint32_t* doc, const uint32_t* ent_off, const uint8_t* ent_field,
					  const uint32_t* ent_tf, const uint32_t* ent_first_pos);
/* Device half of Merger::mergeSimple for a Simple() query: nsub sub-terms (word_ids[], procs[], caller-sorted by proc desc like
 * SortSubterms), docsExcluded bitmap or NULL.  Writes the admitted documents IN MERGE ORDER with their rank (addFullMatchBoost applied,
 * merger.h:100-109) and field — before postProcessResults (merger.h:111-155), which the host merger applies to these <= merge_limit rows. */
int rxgpu_ft_merge_simple_raw(rxgpu_ft_index* h, const rxgpu_ft_config* cfg, const rxgpu_ft_term_opts* opts, uint32_t nsub,
							  const uint32_t* word_ids, const float* procs, const uint8_t* excluded, uint32_t* out_doc, float* out_proc,
							  uint8_t* out_field, uint64_t cap, uint64_t* out_n);
/* Posting list of one dictionary word WITH its positions, as the multi-term merge needs them (PositionsDistance,
 * cpp_src/core/ft/ft_fast/mergerimpl.h:20-37): per posting a run [pos_off[i], pos_off[i+1]) of PosType words
 * (cpp_src/core/ft/idrelset.h:14-32: pos | arrayIdx << 28 | field << 56), ascending like IdRelType::SortAndUnique leaves them.
 * The (field, tf, first position) entries of rxgpu_ft_set_word are derived from them, so the word serves both merges. */
int rxgpu_ft_set_word_positions(rxgpu_ft_index* h, uint32_t word_id, uint64_t n, const uint32_t* doc, const uint32_t* pos_off, const uint64_t* fpos);
/* Posting lists in the reference's own storage format, decoded ON THE DEVICE: `bytes` holds the PackedIdRelVec streams of
 * nwords dictionary words back to back (cpp_src/core/ft/idrelset.h:155-280; IdRelType::pack / unpack, idrelset.cc:8-139; varints
 * tools/varint.h:122-176), word w = bytes[byte_off[w] .. byte_off[w + 1]); array_found_pos[w] = PackedIdRelVec's offset (relative to the
 * word's first byte) from which elements carry array indexes (>= the stream length if none do).  Replaces, for these words, the host-side
 * flattening + rxgpu_ft_set_word_positions: documents, positions (PosType words), the (field, tf, first position) entries of
 * calcTermRankImpl (phrasemergerimpl.h:24-49) and the range index are produced by a kernel (one thread per word) into one device
 * allocation per call.  A malformed stream (truncated varint, ids not ascending, field >= num_fields) is RXGPU_ERR_PARAMS naming the word. */
int rxgpu_ft_set_words_packed(rxgpu_ft_index* h, uint32_t nwords, const uint32_t* word_ids, const uint64_t* byte_off, const uint8_t* bytes,
							  const uint64_t* array_found_pos);
/* The same with word w's stream at data[w] (len[w] bytes) — PackedIdRelVec::RawData() of every dictionary entry, where the engine keeps it:
 * the streams are gathered once, in launch order, into pinned staging memory and travel in one asynchronous copy. */
int rxgpu_ft_set_words_packed_ptrs(rxgpu_ft_index* h, uint32_t nwords, const uint32_t* word_ids, const uint8_t* const* data, const uint64_t* len,
								   const uint64_t* array_found_pos);
/* Wall time spent inside rxgpu_ft_set_words_packed / _ptrs since the last call (the commit-side cost at the C-ABI boundary). */
int rxgpu_ft_read_packed_wall(rxgpu_ft_index* h, double* wall_ms);
/* Device time of the two decode kernels of rxgpu_ft_set_words_packed (count pass, write pass), the stream bytes they read (each pass)
 * and the bytes of the arrays they produced (256-byte aligned slices included), since the last call. */
int rxgpu_ft_read_packed_stats(rxgpu_ft_index* h, double* count_ms, double* write_ms, uint64_t* bytes_in, uint64_t* bytes_out);
/* Reads a word's device arrays back (tests, diagnostics).  Sizes first (array pointers null), then the arrays the caller wants. */
int rxgpu_ft_get_word(rxgpu_ft_index* h, uint32_t word_id, uint64_t* n, uint64_t* npos, uint64_t* nent, uint32_t* doc, uint32_t* pos_off, uint64_t* fpos,
					  uint32_t* ent_off, uint8_t* ent_field, uint32_t* ent_tf, uint32_t* ent_first_pos, uint32_t* n_ranges, uint32_t* range_off);

/* Device half of Merger::Merge for a query of nterms >= 2 terms without phrases / multi-word synonyms (mergerimpl.h:466-566):
 * buildRestrictingBitmask (:326-384), the 2-phase gate + preselectMostRelevantDocs (:386-464, 486-490) and mergeTerm (:107-192) for
 * every term that is not a NOT.  ops[t]: OpType 1 OR / 2 AND / 3 NOT (core/type_consts.h); opts[t]: the term's FtDslOpts; the
 * sub-terms of term t are word_ids/procs[sub_off[t] .. sub_off[t+1]) sorted by proc descending (SortSubterms).
 * Writes the merged documents IN MERGE ORDER with proc (addFullMatchBoost applied: canBeBoostedByFullMatch :527-531, merger.h:100-109),
 * field and MergerDocumentData::termsCounter (merger.h:24); the host merger applies postProcessResults (merger.h:111-155).
 * *out_preselected = 1 if the preselect phase ran.  cap >= min(merge_limit, total postings). */
int rxgpu_ft_merge_terms_raw(rxgpu_ft_index* h, const rxgpu_ft_config* cfg, uint32_t nterms, const int32_t* ops, const rxgpu_ft_term_opts* opts,
							 const uint32_t* sub_off, const uint32_t* word_ids, const float* procs, const uint8_t* excluded, uint32_t* out_doc,
							 float* out_proc, uint8_t* out_field, uint16_t* out_terms_counter, uint64_t cap, uint64_t* out_n, int32_t* out_preselected);
/* Device half of Merger::Merge for ANY query made of terms and phrases (QueryMergeData::queryParts, querymergedata.h:191-242; the
 * selecter builds them in selecterimpl.h:482-572): the arguments of rxgpu_ft_merge_terms_raw plus, per term, FtDslOpts::phraseNum and
 * FtDslOpts::distance (ftdsl.h:13-35) — consecutive terms with the same phrase_num >= 0 form one phrase (PhraseResults; its operator is
 * its first term's), phrase_num < 0 is a plain term; both arrays may be NULL (no phrases).  Every phrase goes through PhraseMerger::Merge
 * on the device first (phrasemergerimpl.h:161-329: preselectDocsContainingAllTerms, mergePhraseTerm with MergePositionsWithDist /
 * SwitchPositions, phrasemerger.h:24-55, 107-140), then the query parts are merged: mergeTerm for terms, mergePhrase (mergerimpl.h:39-90)
 * for phrases, the restricting bitmask / pre-scores with GetMergedDocsBitmask / ExcludeMergedDocsFromBitmask / GetMergedDocsScore
 * (phrasemerger.h:309-333).  A single plain term is the Simple() merge; an Empty() query writes nothing.  Multi-word synonyms
 * (QueryMergeData::synonyms) are not covered: such queries stay on the CPU merger.  Outputs as rxgpu_ft_merge_terms_raw;
 * cap >= min(merge_limit, total postings of all terms). */
int rxgpu_ft_merge_query_raw(rxgpu_ft_index* h, const rxgpu_ft_config* cfg, uint32_t nterms, const int32_t* ops, const rxgpu_ft_term_opts* opts,
							 const int32_t* phrase_num, const int32_t* distance, const uint32_t* sub_off, const uint32_t* word_ids, const float* procs,
							 const uint8_t* excluded, uint32_t* out_doc, float* out_proc, uint8_t* out_field, uint16_t* out_terms_counter, uint64_t cap,
							 uint64_t* out_n, int32_t* out_preselected);
/* The whole QueryMergeData (querymergedata.h:191-242) in one description: query parts (terms, phrases) AND multi-word synonyms.
 * Terms [0, nterms) make up the query parts exactly as in rxgpu_ft_merge_query_raw; terms [nterms, nterms + nsyn_terms) are the terms of the
 * nsyn multi-word synonyms (Synonym::Terms(), querymergedata.h:178-192), synonym s owning terms nterms + syn_term_off[s] .. nterms +
 * syn_term_off[s + 1]; every per-term array has nterms + nsyn_terms entries (phrase_num / distance of a synonym's term are ignored).
 * part_syn_off [nparts + 1] / part_syn: PhraseOrTerm::SynonymsIds() of every query part (nparts = the number of parts the terms form).
 * suppressed (may be NULL): SubtermResults::Suppressed() per sub-term — what QueryMergeData::SupressDuplicatesInSynonyms (:221-241) set.
 * The merge: buildRestrictingBitmask with the synonyms' masks (mergerimpl.h:347-361), their terms in the pre-scores (:393-397) and in the
 * 2-phase estimate (merger.h:251-255), mergeTerm for every synonym term behind the query parts with the term counting, the
 * containsFullMultiWordSynonym rule and the removal of documents that hold only parts of a synonym (:509-555).  A query without
 * synonyms is rxgpu_ft_merge_query_raw's. */
typedef struct rxgpu_ft_query {
	uint32_t nterms, nsyn_terms;
	const int32_t* ops;
	const rxgpu_ft_term_opts* opts;
	const int32_t* phrase_num;      /* may be NULL */
	const int32_t* distance;        /* may be NULL */
	const uint32_t* sub_off;        /* [nterms + nsyn_terms + 1] */
	const uint32_t* word_ids;
	const float* procs;
	const uint8_t* suppressed;      /* per sub-term, may be NULL */
	uint32_t nsyn;
	const uint32_t* syn_term_off;   /* [nsyn + 1] */
	const uint32_t* part_syn_off;   /* [nparts + 1], may be NULL when nsyn == 0 */
	const uint32_t* part_syn;
} rxgpu_ft_query;
int rxgpu_ft_merge_query2_raw(rxgpu_ft_index* h, const rxgpu_ft_config* cfg, const rxgpu_ft_query* query, const uint8_t* excluded, uint32_t* out_doc,
							  float* out_proc, uint8_t* out_field, uint16_t* out_terms_counter, uint64_t cap, uint64_t* out_n, int32_t* out_preselected);
/* Merger<IdCont, MergeDataAreas<Area>, ..>::Merge — the merge behind highlight() / snippet() (merger.h:36-57 with kWithRegularAreas; addAreas
 * :196-204; AreasInDocument / AreasInField::Insert / Area::Concat, core/ft/areaholder.h:9-37, 56-155): the result of rxgpu_ft_merge_query2_raw
 * plus, per merged document i (merge order, like out_doc) and field f, the areas its postings left: out_area_cnt[i * num_fields + f] entries
 * of {start, end, arrayIdx} at out_areas[((i * num_fields + f) * max_areas_in_doc + j) * 3], in the order AreasInField::data_ holds them
 * BEFORE Commit() (the engine sorts and joins them when they are read).  max_areas_in_doc = FTConfig::maxAreasInDoc (default 5), >= 1.
 * out_area_cnt: cap * num_fields words; out_areas: cap * num_fields * max_areas_in_doc * 3 words.  Queries of plain terms (Simple() included;
 * the words need their positions); phrases, multi-word synonyms and MergeDataAreas<AreaDebug> return RXGPU_ERR_LOGIC: the CPU merger. */
int rxgpu_ft_merge_query_areas_raw(rxgpu_ft_index* h, const rxgpu_ft_config* cfg, const rxgpu_ft_query* query, const uint8_t* excluded, uint32_t max_areas_in_doc,
								   uint32_t* out_doc, float* out_proc, uint8_t* out_field, uint16_t* out_terms_counter, uint64_t cap, uint64_t* out_n,
								   int32_t* out_preselected, uint32_t* out_area_cnt, uint32_t* out_areas);
/* nq queries (each as in rxgpu_ft_merge_query2_raw) over one index in ONE launch train: the kernels of the merge run with the query as the
 * second grid dimension, so the per-launch floors and the ramp of every grid are paid once per train and the device works on all the
 * queries' document ranges at a time — the form for a caller with several Merge() calls in hand (the hybrid path's query batch, T planner
 * threads behind one combiner).  Per query i: excluded[i] (the array or any entry may be NULL), the output arrays out_*[i] of `cap` entries
 * each, out_n[i], out_preselected[i] (may be NULL).  Results are those of nq single calls, bit for bit.  Queries with phrases or
 * multi-word synonyms (kernels of their own in front of the train) are run one by one inside the call; at most 64 queries share a train
 * (longer batches are cut). */
int rxgpu_ft_merge_batch_raw(rxgpu_ft_index* h, const rxgpu_ft_config* cfg, uint32_t nq, const rxgpu_ft_query* queries, const uint8_t* const* excluded,
							 uint32_t* const* out_doc, float* const* out_proc, uint8_t* const* out_field, uint16_t* const* out_terms_counter, uint64_t cap,
							 uint64_t* out_n, int32_t* out_preselected);
/* Launch trains run by rxgpu_ft_merge_batch_raw and the merges they carried, since the index was created. */
int rxgpu_ft_read_batch_stats(rxgpu_ft_index* h, uint64_t* trains, uint64_t* merges);
/* Two launch trains produce the same merge: the dense one (a workgroup per 8192-document range, per-document arrays in HBM between its
 * kernels) and the one for SPARSELY hit ranges (a wavefront per (query, range), bitmaps per sub-term in LDS, nothing per document in HBM;
 * eligible: plain terms whose fields share one positive boost, <= 16 sub-terms, Bm25Rx / TermCount, weights below 1).  By default the host
 * picks per query (eligible and postings on <= 30 % of the documents -> sparse).  mode: -1 that default, 0 always dense, 1 sparse whenever
 * eligible.  Process-wide; RXGPU_FT_TRAIN=dense|sparse presets it.  rxgpu_ft_read_train_stats: merges run by either since the last call. */
void rxgpu_ft_set_train_mode(int mode);
int rxgpu_ft_read_train_stats(rxgpu_ft_index* h, uint64_t* dense_merges, uint64_t* sparse_merges);
/* ---------------------------------------------------------------------------------------------------------
 * Hybrid rank fusion on the device: MergerRankedImpl + mergeRanked (cpp_src/core/nsselecter/selectiteratorcontainer.cc:
 * 1343-1423, 1454-1559), RanksHolder::InitRRFPositions (ranks_holder.h:61-76), RerankerRRF / RerankerLinear (core/sorting/reranker.h:11-39),
 * the Merged<desc> order (selectiteratorcontainer.cc:1258-1283: desc = rank descending, ties by DESCENDING id).
 * ------------------------------------------------------------------------------------------------------- */
typedef struct rxgpu_hybrid_params {
	int32_t kind;       /* 0 = RRF: params[0] = rank_const; 1 = linear: params = kKnn, knnDefault, kFt, ftDefault, c */
	int32_t is_union;   /* the two ranked conditions joined by OR (union) / AND (intersection), selectiteratorcontainer.cc:1473-1480 */
	int32_t desc;       /* 1: the normal rank() / RRF() ordering */
	int32_t reserved;
	double params[5];
} rxgpu_hybrid_params;
/* The merge of rxgpu_ft_merge_simple_raw / _terms_raw, but the result STAYS IN HBM (ft_finish's output; no export, no wait, the call
 * returns as soon as the train is enqueued) for rxgpu_hybrid_fuse_resident to read.
 * A resident merge opens a SESSION on the index that belongs to the calling thread and ends with that thread's rxgpu_hybrid_fuse_resident:
 * ordinary merges of other threads run on the handle's other lanes meanwhile, another thread's resident merge (or session-less fusion)
 * WAITS for the session to end.  A session that is not fused within 2 s is taken over; its owner's prepare / fuse then fail with
 * RXGPU_ERR_LOGIC — a resident result is never silently replaced by an empty or a foreign one. */
int rxgpu_ft_merge_simple_resident(rxgpu_ft_index* h, const rxgpu_ft_config* cfg, const rxgpu_ft_term_opts* opts, uint32_t nsub,
								   const uint32_t* word_ids, const float* procs, const uint8_t* excluded);
int rxgpu_ft_merge_terms_resident(rxgpu_ft_index* h, const rxgpu_ft_config* cfg, uint32_t nterms, const int32_t* ops, const rxgpu_ft_term_opts* opts,
								  const uint32_t* sub_off, const uint32_t* word_ids, const float* procs, const uint8_t* excluded);
/* ... the same for rxgpu_ft_merge_query_raw's queries (phrases included).  *out_enqueued = 0 when the query is Empty(): there is then no
 * resident result and the fusion sees an empty FT side. */
int rxgpu_ft_merge_query_resident(rxgpu_ft_index* h, const rxgpu_ft_config* cfg, uint32_t nterms, const int32_t* ops, const rxgpu_ft_term_opts* opts,
								  const int32_t* phrase_num, const int32_t* distance, const uint32_t* sub_off, const uint32_t* word_ids, const float* procs,
								  const uint8_t* excluded, int32_t* out_enqueued);
/* The resident form of rxgpu_ft_merge_query2_raw (terms, phrases AND multi-word synonyms, QueryMergeData::synonyms): Merger::Merge
 * (mergerimpl.h:466-566) left in HBM for rxgpu_hybrid_prepare_resident / rxgpu_hybrid_fuse_resident.  The documents Merge() removes because
 * they hold only parts of a synonym (mergerimpl.h:533-555) are skipped by the fusion kernels. */
int rxgpu_ft_merge_query2_resident(rxgpu_ft_index* h, const rxgpu_ft_config* cfg, const rxgpu_ft_query* query, const uint8_t* excluded, int32_t* out_enqueued);
/* The FT-only half of the fusion (postProcessResults, the documents' order among themselves, the rank-class tables), enqueued behind the
 * resident merge: it needs nothing from the KNN side, so a caller that enqueues it BEFORE it starts the KNN search has it run while the
 * scan streams the corpus, and only the short join is left on the query's critical path.  Optional — rxgpu_hybrid_fuse_resident
 * enqueues it itself when it was not called with the same min_rank / params / d_row_of_doc. */
int rxgpu_hybrid_prepare_resident(rxgpu_ft_index* h, int32_t min_rank, const rxgpu_hybrid_params* params, int metric, const void* d_row_of_doc);
/* Fuses the resident merge of `h` (postProcessResults — merger.h:111-140: proc < min_rank dropped, scaled to 0..255, uint8 — is applied on
 * the device) with a KNN result that lies in HBM as rxgpu_search_knn_device left it: d_knn_dist / d_knn_row best first, d_knn_count (device
 * uint32, or NULL) of knn_n entries valid, the first k taking part (ranks as the planner sees them: the distance for L2, its negation for
 * inner product / cosine, hnsw_index.cc:261-270).  knn_stream: the stream that search was enqueued on (the fusion waits for it on the
 * device) or NULL.  d_row_of_doc: int32 [total_docs], vdoc -> row id when texts and rows correspond 1:1 otherwise than by number, or NULL;
 * d_rowid_of_row: int32 [rows], internal row -> row id (label >> 32), or NULL = identity.  One list of (row id, fused rank) comes back,
 * ordered like Merged<desc>.  *out_flags bit 0: the k-th and (k+1)-th KNN distances are equal — the boundary tie is decided by labels on
 * the host (GpuBruteforceMap replays it), the caller should take that path.  cap >= merge_limit + k. */
int rxgpu_hybrid_fuse_resident(rxgpu_ft_index* h, int32_t min_rank, const rxgpu_hybrid_params* params, int metric, const void* d_knn_dist,
							   const void* d_knn_row, const void* d_knn_count, uint32_t knn_n, uint32_t k, void* knn_stream, const void* d_row_of_doc,
							   const void* d_rowid_of_row, int32_t* out_ids, float* out_ranks, uint64_t cap, uint64_t* out_n, uint32_t* out_flags);
/* Fusions run by rxgpu_hybrid_fuse_resident and the device time of their JOIN kernel — what is left on the critical path once both
 * halves are there (HIP events on the merger's stream) — since the last call. */
int rxgpu_hybrid_read_stats(rxgpu_ft_index* h, uint64_t* calls, double* kernel_ms, double* prepare_ms /* the overlapped FT-only kernel; may be NULL */);
/* The same kernel on host arrays: knn_ids / knn_ranks best first (at most 1024), ft_ids (unique, any order) with their uint8 ranks
 * (MergeInfo::normalizedProc).  cap >= n_knn + n_ft. */
int rxgpu_hybrid_fuse(int device, const rxgpu_hybrid_params* params, int metric, const int32_t* knn_ids, const float* knn_ranks, uint32_t n_knn,
					  const int32_t* ft_ids, const uint8_t* ft_ranks, uint32_t n_ft, int32_t* out_ids, float* out_ranks, uint64_t cap, uint64_t* out_n);

/* Postings scored / kernel milliseconds since the last call (roofline accounting: 20 B per posting). */
int rxgpu_ft_read_stats(rxgpu_ft_index* h, uint64_t* postings, double* kernel_ms);

/* ---------------------------------------------------------------------------------------------------------
 * Instrumentation (bench.py roofline leg): HIP-event timing of the dominant kernel on its own stream.
 * ------------------------------------------------------------------------------------------------------- */
int rxgpu_profile_enable(rxgpu_index* h, int on);
/* name: "scan" | "merge" | "range" ...; returns launches recorded since enable and their summed milliseconds. */
int rxgpu_profile_read(rxgpu_index* h, const char* name, uint64_t* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* RXGPU_H */
