// GpuHnswMap — the GPU `Map` policy for HnswIndexBase<Map> in its HNSW form (hnsw_index.cc:47-58), a drop-in for
// hnswlib::HierarchicalNSW<Synchronization::None> (cpp_src/core/index/float_vector/hnswlib/hnsw.h:14-126).
//   build  : host, HnswGraph (hnsw_graph.h) — same graph as the reference builds
//   search : MI355X, rxgpu_hnsw_search_knn (hnsw_search.hip) — same traversal as the reference, no CPU search path
#pragma once

#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <optional>
#include <unordered_map>
#include <vector>

#include "hnsw_graph.h"
#include "rx_types.h"
#include "sq8_quantizer.h"

struct rxgpu_index;
struct rxgpu_hnsw_stream;

namespace rxgpu::host {

// hnsw_interface.h:18-45
#if defined(RXGPU_IN_TREE)
using hnswlib::StreamingSearchOptions;
using hnswlib::Synchronization;
#else
struct StreamingSearchOptions {
	size_t ef = 0;
};
#endif
struct StreamingBatch {
	SearchResultQueue results;
	bool exhausted = false;
};
class StreamingSearchSession {
public:
	StreamingSearchSession();
	StreamingSearchSession(StreamingSearchSession&& o) noexcept;
	StreamingSearchSession& operator=(StreamingSearchSession&& o) noexcept;
	StreamingSearchSession(const StreamingSearchSession&) = delete;
	~StreamingSearchSession();

private:
	friend class GpuHnswMap;
	rxgpu_hnsw_stream* impl_ = nullptr;
	const void* graph_ = nullptr;   // the Map that began the session (hnswalg.h:1953-1956: a foreign session is reported exhausted)
	struct Sharded;                 // a Map over a device list: one session per shard + what each has delivered and the merge not emitted
	std::unique_ptr<Sharded> sharded_;
};

// hnswlib::Synchronization (hnswlib.h): None = HierarchicalNSWST (AddPointConcurrent throws), OnInsertions = HierarchicalNSWMT (the index
// type the reference builds from several upsert threads, hnsw_index.cc:18-19, 105-116, 566-573)
#if !defined(RXGPU_IN_TREE)
enum class Synchronization { None, OnInsertions };
#endif

class GpuHnswMap {
public:
	GpuHnswMap(VectorMetric metric, size_t dim, size_t maxElements, size_t M, size_t efConstruction, int device = 0,
			   Synchronization synchronization = Synchronization::None);
	// SURVEY 8(e) "HNSW": the same Map over a DEVICE LIST (more than one entry; a device may repeat).  The points are split into row ranges in
	// insertion order — shard s takes the points that arrive while shards 0 .. s - 1 are full (ceil(maxElements / N) each) — and every
	// shard is a complete single-device Map of its own: its graph is what the reference builds over those points in that order, mirrored
	// on that shard's GPU (one rxgpu_index_create_sharded handle owns the N device indexes).  SearchKnn runs every shard's search at once and
	// the per-shard results meet in the same ncclAllGather + (dist, global row) merge as brute force (rxgpu_hnsw_search_knn on the sharded
	// handle): the k best of the union of the per-shard engine results, recall >= the single graph's at equal ef.  SQ8 (one quantiser for the Map, a
	// code table per shard) and streaming sessions (a session per shard, merged batch by batch) work over a device list too; the ANN disk
	// cache does not (it reports "not available").
	GpuHnswMap(VectorMetric metric, size_t dim, size_t maxElements, size_t M, size_t efConstruction, std::vector<int> devices,
			   Synchronization synchronization = Synchronization::None);
	GpuHnswMap(const GpuHnswMap& other, size_t newCapacity);
	~GpuHnswMap();
	GpuHnswMap& operator=(const GpuHnswMap&) = delete;

	size_t MaxElements() const noexcept { return sh_ ? shMaxElements() : graph_.MaxElements(); }
	size_t CurrentElementCount() const noexcept { return sh_ ? shCount(false) : graph_.Count(); }
	size_t DeletedCountUnsafe() const noexcept { return sh_ ? shCount(true) : graph_.DeletedCount(); }
	size_t AllocatedMemSize() const noexcept { return sh_ ? shAllocated() : graph_.AllocatedMemSize(); }
	// hnswalg.h:216-228: links0 + data + label + hash per element
	size_t ElementSize() const noexcept { return (1 + graph_.MaxM0()) * sizeof(uint32_t) + graph_.Dim() * sizeof(float) + 16; }

	// internal ids of a Map over a device list are GLOBAL rows: shard * ShardRows() + the shard's internal id
	labeltype ExternalLabel(tableint id) const { return sh_ ? shLabel(id) : graph_.Label(id); }
	bool IsMarkedDeleted(tableint id) const noexcept { return sh_ ? shIsDeleted(id) : graph_.IsDeleted(id); }
	const float* FloatPtrByExternalLabel(labeltype label) const { return sh_ ? shFloatPtr(label) : graph_.Vector(graph_.InternalId(label)); }
	bool Sharded() const noexcept { return bool(sh_); }
	size_t ShardCount() const noexcept;
	size_t ShardRows() const noexcept;
	const GpuHnswMap& Shard(size_t s) const;   // shard s as the single-device Map it is (tests pin each to the reference engine)
	rxgpu_index* DeviceIndex() const noexcept;   // the device handle (sharded: the rxgpu_index_create_sharded handle)

	void MarkDelete(FloatVectorId id);
	void AddPointNoLock(ConstFloatVectorView vect, FloatVectorId id);
	void AddPointConcurrent(ConstFloatVectorView vect, FloatVectorId id);   // Synchronization::None: throws, like the reference's ST map
	void ResizeIndex(size_t newMaxElements);

	SearchResultQueue SearchKnn(const float* queryDataRaw, std::optional<float> queryDataNorm, size_t k, size_t ef = 0) const;
	SearchResultQueue SearchRange(const float* queryDataRaw, std::optional<float> queryDataNorm, float radius, size_t ef) const;

	// Streaming (batched) KNN, hnswalg.h:1865-1975.  The whole session must run under the caller's read lock, like the reference's.
	StreamingSearchSession BeginStreamingSearch(const float* queryDataRaw, std::optional<float> queryDataNorm, StreamingSearchOptions opts) const;
	StreamingBatch ContinueStreamingSearch(StreamingSearchSession& session, size_t batchSize) const;

	// Query coalescing, as in GpuBruteforceMap: one-shot searches that arrive while the device is busy and ask for the same (k, ef) share
	// one launch of the batched search kernel (a wavefront per query) instead of one single-wavefront launch each.
	void EnableQueryCoalescing(bool on) noexcept { coalesce_ = on; }
	size_t CoalescedBatches() const noexcept { return coBatches_; }
	// one-shot searches answered through the index's resident search kernel (no launch, no batching: rxgpu_hnsw_search_knn_posted)
	size_t PostedQueries() const noexcept { return coPosted_.load(); }
	// device batches the coalescer keeps in flight at once (1 .. 64; default kMaxLeaders)
	void SetCoalescerLanes(unsigned lanes) noexcept { coLanes_ = lanes < 1 ? 1 : (lanes > 64 ? 64 : lanes); }
	// searches the device re-ran on its heap kernel because the sorted-list search met equal distances (rxgpu_hnsw_read_tie_reruns); resets
	uint64_t TieReruns() const;
	// searches whose candidate heap outgrew the LDS area of their first pass and ran again with the largest one (rxgpu_hnsw_read_lds_reruns); resets
	uint64_t LdsReruns() const;

	// SQ8 (HierarchicalNSW::Quantize, hnsw.h:104-118; hnswalg.h:411-470): from here on SearchKnn runs over one byte per component on the
	// device (rxgpu_hnsw_search_knn_sq8) and returns what HierarchicalNSWImpl<uint8_t> returns on the same graph, bit for bit.  The range
	// [minQ, maxQ] is what QuantizingParams derives from its sample (quantization_params.h:48-63); the caller supplies it (in-tree:
	// QuantizingParams over the Map's rows).  Points added later are quantised with the same parameters at the next sync
	// (addPoint, hnswalg.h:1480-1495).  Streaming and range searches of a quantised Map are not implemented: they throw.
	void Quantize(float minQ, float maxQ);
	// HierarchicalNSW::Quantize(config) + SwitchMapOnQuantized() (hnsw.h:61-62, hnsw.cc:131-137, Impl::get :108-114) as HnswIndexBase drives them
	// (hnsw_index.cc:532-551): Quantize derives the parameters from a sample of the stored rows exactly as QuantizingParams does
	// (quantization_params.h:48-66: reservoir sample by std::rand, batches of 20 rows, n-th min / max per batch, means) and keeps them PENDING —
	// searches go on over the float rows, like the reference's readers on ptr_ while quantizedPtr_ is built — until SwitchMapOnQuantized()
	// (under the namespace write lock) makes the Map a quantised one.
	void Quantize(const Sq8QuantizationConfig& config);
	void SwitchMapOnQuantized();
	bool IsQuantized() const noexcept { return quantized_; }
	bool QuantizationAvailable() const noexcept { return !quantized_ && !pendingSq8_; }
	const Sq8Params& QuantizingParams() const noexcept { return sq8_; }

	VectorMetric Metric() const noexcept { return graph_.Metric(); }
	size_t Dim() const noexcept { return graph_.Dim(); }
	const HnswGraph& Graph() const noexcept { return graph_; }

private:
	void syncDevice() const;
	// ---- the Map over a device list
	struct ShardedState;
	GpuHnswMap(VectorMetric metric, size_t dim, size_t maxElements, size_t M, size_t efConstruction, int device, Synchronization synchronization,
			   rxgpu_index* external);                                                     // a shard: the device index belongs to the sharded handle
	GpuHnswMap(const GpuHnswMap& other, size_t newCapacity, rxgpu_index* external);
	void rebindDevice(rxgpu_index* external);                                               // the sharded handle was re-created: mirror everything again
	void shCreateParent(size_t shardRows);
	size_t shMaxElements() const noexcept;
	size_t shCount(bool deleted) const noexcept;
	size_t shAllocated() const noexcept;
	labeltype shLabel(tableint id) const;
	bool shIsDeleted(tableint id) const noexcept;
	const float* shFloatPtr(labeltype label) const;
	GpuHnswMap& shRoute(labeltype label);
	void shSyncAll() const;
	void shLoadGraphs(AnnCacheReader& reader);                                               // the sharded ANN cache behind the quantising parameters
	void saveQuantizingParams(AnnCacheWriter& writer) const;
	static bool loadQuantizingParams(AnnCacheReader& reader, Sq8Params& stored, Sq8QuantizationConfig& cfg);
public:
	// The ANN disk cache (ann_cache.h): HierarchicalNSW::SaveIndex / LoadIndex (hnswlib/hnsw.cc:41-72).  The stream starts with the
	// "quantised" flag; a quantised Map writes 1 + its QuantizingParams (version, QuantizationConfig, minQ, maxQ, alpha, alpha_2, delta:
	// quantization_params.h:83-96, quantization_config.cc:44-49), a float graph 0.  LoadIndex reads them back: parameters in the stream and
	// reader.WithQuantizer() -> the Map comes back QUANTISED with exactly those parameters (the codes are rebuilt from the rows: the cache holds
	// links and keys); otherwise a float graph (hnsw.cc:47-53).  The Map must be empty; the next search uploads the whole graph to the device.
	// A Map over a device list writes a stream of its own shape — a graph per shard behind a header that every single-graph reader (the
	// reference's engine, a single-device Map) refuses at once with the reference's "Current elements count is larger than max elements count",
	// i.e. a cache miss and a rebuild; it loads only into a Map over as many shards (gpu_hnsw_map.cc: shLoadGraphs).
	void SaveIndex(AnnCacheWriter& writer, const std::atomic_int32_t& cancel) const;
	void LoadIndex(AnnCacheReader& reader);
	void LoadGraph(AnnCacheReader& reader);
	void Clear();   // an empty Map again (HnswIndexBase::clearMap)
private:
	struct PendingQuery;
	void fetchKnn(const float* query, uint32_t k, uint32_t ef, float* dist, uint32_t* row, uint32_t* count) const;

	mutable HnswGraph graph_;   // mutable: syncDevice() (const, under syncMtx_) drains the graph's change tracker
	const int device_;
	mutable std::mutex syncMtx_;
	mutable rxgpu_index* dev_ = nullptr;
	mutable size_t syncedRows_ = 0;
	mutable bool graphOnDevice_ = false;   // a full attach happened: later changes can be patched in place (rxgpu_hnsw_patch_graph)
	const Synchronization synchronization_;
	mutable std::atomic<bool> graphDirty_{true};
	mutable bool deletedDirty_ = false;
	bool quantized_ = false;
	Sq8Params sq8_;
	Sq8QuantizationConfig sq8Config_;               // what the parameters were sampled under (travels with them in the ANN cache)
	std::optional<Sq8Params> pendingSq8_;           // Quantize(config) ran, SwitchMapOnQuantized() has not yet
	mutable bool codesDirty_ = false;               // the whole code table has to be (re)built and sent
	mutable size_t syncedCodes_ = 0;                // rows whose codes are on the device
	mutable std::vector<tableint> codesDirtyRows_;  // what the graph's change tracker named at this sync (when it could)
	mutable bool codesIncremental_ = false;
	void attachCodes() const;
	void patchCodes(const std::vector<tableint>& dirty, size_t n) const;   // codes of the changed / new rows only
	float quantizeQuery(const float* queryDataRaw, std::optional<float> queryDataNorm, std::vector<uint8_t>& qcodes, float& normCoef) const;

	bool coalesce_ = true;
	mutable std::mutex coMtx_;
	mutable std::condition_variable coCv_;
	mutable std::deque<PendingQuery*> coQueue_;
	static constexpr unsigned kMaxLeaders = 4;   // device batches of the coalescer in flight at once (gpu_hnsw_map.cc: fetchKnn)
	unsigned coLanes_ = kMaxLeaders;
	mutable unsigned coLeaders_ = 0;
	mutable size_t coBatches_ = 0;
	mutable std::atomic<size_t> coPosted_{0};
	bool ownsDev_ = true;                 // false: a shard (the device index belongs to the sharded handle) or the Map over a device list itself
	std::unique_ptr<ShardedState> sh_;    // non-null: the Map over a device list
};

}  // namespace rxgpu::host
