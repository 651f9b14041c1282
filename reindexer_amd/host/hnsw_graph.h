// HnswGraph — host-side construction of the HNSW graph in the FLAT layout the GPU search kernel consumes.
//
// Replaces the build half of hnswlib::HierarchicalNSWImpl<float, None> (cpp_src/core/index/float_vector/hnswlib/hnswalg.h):
//   level draw            getRandomLevel :624-635, seed 100 (hnsw.h:73), mult = 1/ln(M) :219
//   insertion             addPoint :1694-1852
//   construction search   searchBaseLayer :644-749
//   neighbour selection   getNeighborsByHeuristic2 :977-1024
//   linking               mutuallyConnectNewElement :1042-1180
//   delete mark           MarkDelete / markDeletedInternal :1303-1339
// Same algorithm, same RNG stream, same distance bits (distance_cpu.h), same heap tie mechanics (ResultHeap, rx_types.h) =>
// for sequential inserts the graph equals the reference's link for link (tests/test_hnsw_builder.py).
//
// Concurrent construction (HierarchicalNSWMT = Synchronization::OnInsertions, hnsw.h:60-91; the reference's multithreaded index build):
// AddPointConcurrent follows addPoint<RegularLocker> lock for lock — label table mutex, level generator mutex, `global` held only while
// a new top level is being created, the new element's link-list lock held for the whole insertion, every other list locked while it is read
// or rewritten (hnswalg.h:1694-1852, 644-749, 1042-1180).  Like the reference's, such a graph depends on thread timing; inserted from
// ONE thread it equals the sequential graph link for link (tests/test_hnsw_builder.py).
//
//   slot reuse            addPoint<LockerT>(data, label) :1401-1470: a new point takes `*deleted_elements.begin()` when a deleted slot exists
//                         (the reference's Maps are built with ReplaceDeleted_True, hnsw.h:72); the hash set's iteration order is restated in
//                         hopscotch_set.h, so the SAME slot is recycled
//   in-place update       updatePoint :1472-1587 (one- / two-hop neighbourhood re-selected per neighbour, sets walked in the reference's hash
//                         order), repairConnectionsForUpdate :1589-1680, mutuallyConnectNewElement(isUpdate = true) :1042-1180
// A delete + upsert sequence therefore yields the reference's graph link for link and the capacity does not leak
// (tests/test_hnsw_builder.py: delete / re-insert cycles against the real engine).
#pragma once

#include <atomic>
#include <cstdint>
#include <memory>
#include <mutex>
#include <random>
#include <shared_mutex>
#include <unordered_map>
#include <vector>

#include "ann_cache.h"
#include "hopscotch_set.h"
#include "rx_types.h"

namespace rxgpu::host {

class HnswGraph {
public:
	HnswGraph(VectorMetric metric, size_t dim, size_t maxElements, size_t M, size_t efConstruction, size_t randomSeed = 100);
	HnswGraph(const HnswGraph& other, size_t newMaxElements);

	// addPoint<DummyLocker, ExpectConcurrentUpdates::No>(data, label, -1); returns the internal id
	tableint AddPoint(const float* data, labeltype label);
	// addPoint<RegularLocker, ExpectConcurrentUpdates::No>: callable from many threads at once (never together with AddPoint / Resize /
	// MarkDelete).  AddPoints() is the driver the tools use: the first element goes in alone, the rest from `threads` workers.
	tableint AddPointConcurrent(const float* data, labeltype label);
	// allocates the per-element locks and reserves the label table (no rehash under the table mutex); call it before the first concurrent
	// insert when several threads may race to be first (AddPointConcurrent does it lazily otherwise).  Survives Resize().
	void EnableConcurrentInserts();
	void AddPoints(const float* data, const labeltype* labels, size_t n, unsigned threads);
	void MarkDelete(labeltype label);
	void Resize(size_t newMaxElements);

	size_t MaxElements() const noexcept { return maxElements_; }
	size_t Count() const noexcept { return count_; }
	size_t DeletedCount() const noexcept { return numDeleted_; }
	size_t Dim() const noexcept { return dim_; }
	size_t M() const noexcept { return M_; }
	size_t MaxM0() const noexcept { return maxM0_; }
	int MaxLevel() const noexcept { return maxLevel_; }
	tableint EntryPoint() const noexcept { return entryPoint_; }
	VectorMetric Metric() const noexcept { return metric_; }

	bool HasLabel(labeltype label) const { return labelLookup_.count(label) != 0; }
	bool HasLabelSync(labeltype label) const {   // ... while other threads may be inside AddPointConcurrent
		std::lock_guard<std::mutex> lk(labelMtx_);
		return labelLookup_.count(label) != 0;
	}
	tableint InternalId(labeltype label) const;   // throws std::runtime_error("Label not found")
	labeltype Label(tableint id) const noexcept { return labels_[id]; }
	bool IsDeleted(tableint id) const noexcept { return deleted_[id] != 0; }
	const float* Vector(tableint id) const noexcept { return vectors_.data() + size_t(id) * dim_; }
	float InvNorm(tableint id) const noexcept { return invNorms_.empty() ? 1.f : invNorms_[id]; }

	// flat views for the device upload (see include/rxgpu.h: rxgpu_hnsw_attach_graph)
	const float* Vectors() const noexcept { return vectors_.data(); }
	const float* InvNorms() const noexcept { return invNorms_.empty() ? nullptr : invNorms_.data(); }
	const uint32_t* Links0() const noexcept { return links0_.data(); }      // [count][1 + maxM0]
	const int32_t* Levels() const noexcept { return levels_.data(); }
	const uint8_t* Deleted() const noexcept { return deleted_.data(); }
	const labeltype* Labels() const noexcept { return labels_.data(); }
	// upper levels as CSR blocks of (1 + M) u32: node i owns levels[i] consecutive blocks starting at off[i]
	void ExportUpper(std::vector<uint64_t>& off, std::vector<uint32_t>& blocks) const;
	void AppendUpper(tableint id, std::vector<uint32_t>& blocks) const { blocks.insert(blocks.end(), upper_[id].begin(), upper_[id].end()); }

	size_t AllocatedMemSize() const noexcept;

	// The ANN disk cache (ann_cache.h): HierarchicalNSWImpl::SaveIndex (hnswalg.h:1213-1263) and its reader constructor + initTree
	// (:297-409, 1264-1281), field for field — header (max elements, count, max level, entry point, M, efConstruction), per element the
	// level-0 list (size word with the delete mark, links) and either the vector (deleted) or the row's primary key, then per element the raw
	// upper-level lists.  LoadIndex needs an EMPTY graph built with the cache's M / efConstruction (anything else throws: the caller then
	// rebuilds the index, exactly what HnswIndexBase::LoadIndexCache's error path leads to).
	void SaveIndex(AnnCacheWriter& writer, const std::atomic_int32_t& cancel) const;
	void LoadIndex(AnnCacheReader& reader);
	// back to the freshly constructed state (same capacity, same parameters, level generator reseeded): what HnswIndexBase::clearMap
	// (hnsw_index.cc:72-83) does with `map_ = Map(...)` after a cache load failed half way
	void Clear();

	// Change tracking for the device mirror: the nodes whose lists / vector changed since the last TakeDirty().  Returns false when the
	// tracker gave up (more than a quarter of the graph touched, e.g. a bulk build): everything has to be re-sent.
	bool TakeDirty(std::vector<tableint>& out);

private:
	using Pair = std::pair<float, tableint>;
	struct ByFirst {
		bool operator()(const Pair& a, const Pair& b) const noexcept { return a.first < b.first; }
	};
	using Heap = ResultHeap<Pair, ByFirst>;

	float distIds(tableint a, tableint b) const noexcept;               // DistCalculator(v1,id1,v2,id2) hnswlib.h:123-145
	void loadIndex(AnnCacheReader& reader);
	uint32_t* list(tableint id, int level) noexcept;
	const uint32_t* list(tableint id, int level) const noexcept;
	int randomLevel();
	// construction scratch: visit stamps (VisitedListPool, visited_list_pool.h:13-36); one per inserting thread
	struct Visited {
		std::vector<uint16_t> stamp;
		uint16_t cur = 0;
	};
	template <bool kMT>
	Heap searchBaseLayer(tableint ep, tableint self, int layer, Visited& vis);
	void selectNeighbors(Heap& candidates, size_t M) const;
	template <bool kMT>
	tableint connect(tableint cur, Heap& candidates, int level, bool isUpdate = false);
	template <bool kMT>
	tableint addPoint(const float* data, labeltype label);
	void updatePoint(const float* data, tableint id);                                   // hnswalg.h:1472-1587 (callers hold the graph exclusively)
	void repairConnectionsForUpdate(tableint id, tableint entryPoint, int level, int maxLevel);
	void rebuildDeletedSet();
	void markDirty(tableint id);
	std::unique_ptr<Visited> acquireVisited();
	void releaseVisited(std::unique_ptr<Visited> v);

	const VectorMetric metric_;
	const size_t dim_;
	size_t maxElements_;
	const size_t M_, maxM0_, efConstruction_;
	const double mult_;
	size_t count_ = 0, numDeleted_ = 0;
	int maxLevel_ = -1;
	tableint entryPoint_ = 0xFFFFFFFFu;

	std::vector<float> vectors_;
	std::vector<float> invNorms_;
	std::vector<uint32_t> links0_;
	std::vector<std::vector<uint32_t>> upper_;   // per node: levels * (1 + M)
	std::vector<int32_t> levels_;
	std::vector<labeltype> labels_;
	std::vector<uint8_t> deleted_;
	std::unordered_map<labeltype, tableint> labelLookup_;
	std::default_random_engine levelGenerator_;
	size_t randomSeed_ = 100;
	DeletedIdSet deletedElements_;   // HierarchicalNSWImpl::deleted_elements (allow_replace_deleted_): ids in the reference's hash-set order

	Visited visited_;   // the sequential builder's scratch

	std::vector<tableint> dirty_;
	std::atomic<bool> dirtyAll_{true};   // nothing has been mirrored yet
	std::mutex dirtyMtx_;

	// concurrent construction only
	std::unique_ptr<std::atomic<uint8_t>[]> nodeLocks_;   // link_list_locks_: one byte spin lock per element
	size_t nodeLocksSize_ = 0;
	bool concurrent_ = false;
	mutable std::mutex labelMtx_, generatorMtx_, globalMtx_, entryMtx_, visitedPoolMtx_, deletedMtx_;
	std::shared_mutex updateMtx_;   // concurrent inserts share it; an insert that recycles a slot (updatePoint) holds it exclusively
	std::vector<std::unique_ptr<Visited>> visitedPool_;
};

}  // namespace rxgpu::host
