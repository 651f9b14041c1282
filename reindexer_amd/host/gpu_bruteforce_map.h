// GpuBruteforceMap — the GPU `Map` policy for HnswIndexBase<Map> (cpp_src/core/index/float_vector/hnsw_index.h:16-57),
// a drop-in for hnswlib::BruteforceSearch (cpp_src/core/index/float_vector/hnswlib/bruteforce.h:14-66): same member
// names, argument meaning, result type and exceptions, so Index::New (index.cc:78-116) can instantiate
// HnswIndexBase<GpuBruteforceMap> as a new index type and the planner sees no difference.
//
// Host side keeps the master copy (rows, labels, 1/|row|) because results may hand vectors back to the user
// (FloatPtrByExternalLabel, hnsw_index.cc:363-370); HBM holds a mirror that is synchronised lazily at the first search
// after a mutation (vector indexes have no Commit(), float_vector_index.cc:200-204).  All arithmetic of a search runs
// in the HIP kernels behind include/rxgpu.h; there is NO CPU search fallback.
#pragma once

#include <memory>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <optional>
#include <unordered_map>
#include <vector>

#include "rx_types.h"

struct rxgpu_index;

namespace rxgpu::host {

// tools/normalize.h:16-22 — host-side, exactly as DistCalculator::AddNorm / HnswIndexBase::search use them.
float CalculateL2Module(const float* x, int32_t d) noexcept;
float NormalizeCopyVector(const float* x, int32_t d, float* out) noexcept;

class GpuBruteforceMap {
public:
	GpuBruteforceMap(VectorMetric metric, size_t dim, size_t maxElements, int device = 0);
	// BASELINE configs[3]: the same Map over a DEVICE LIST — the rows are range-sharded over the listed GPUs (rxgpu_index_create_sharded; a
	// device may be listed more than once), every search runs on all of them at once and the per-shard top-k lists are merged under the
	// reference's (dist, row) order, so results — including the k-th-boundary tie replay by label — are those of the single-device Map.
	GpuBruteforceMap(VectorMetric metric, size_t dim, size_t maxElements, std::vector<int> devices);
	GpuBruteforceMap(const GpuBruteforceMap& other, size_t newMaxElements);   // copy-on-write tx clone (hnsw_index.cc:68-70)
	~GpuBruteforceMap();
	GpuBruteforceMap& operator=(const GpuBruteforceMap&) = delete;

	size_t MaxElements() const noexcept { return maxElements_; }
	size_t CurrentElementCount() const noexcept { return curElementCount_; }
	size_t ElementSize() const noexcept { return dim_ * sizeof(float) + sizeof(labeltype); }
	size_t AllocatedMemSize() const noexcept;
	size_t DeviceMemSize() const noexcept;

	const float* FloatPtrByExternalLabel(labeltype label) const;

	void AddPointNoLock(ConstFloatVectorView vect, FloatVectorId id);
	[[noreturn]] void AddPointConcurrent(ConstFloatVectorView, FloatVectorId);
	void RemovePoint(labeltype curExternal);
	void ResizeIndex(size_t newMaxElements);

	SearchResultQueue SearchKnn(const float* queryData, std::optional<float> queryDataNorm, size_t k, size_t ef = 0) const;
	SearchResultQueue SearchRange(const float* queryData, std::optional<float> queryDataNorm, float radius, size_t ef) const;
	// Extension (`WHERE cond AND KNN(...)`): the k nearest among the points whose labels are listed — what SearchKnn returns
	// over an index holding only those points.  The device scan reads the allowed rows only (rxgpu_search_knn_subset / _bitmap).
	SearchResultQueue SearchKnnFiltered(const float* queryData, std::optional<float> queryDataNorm, size_t k, const labeltype* allowed,
										size_t nAllowed) const;

	// Hybrid query, KNN half: ONE query's search enqueued, its exact top-(k + 1) (dist, internal row) list LEFT IN HBM for the
	// rank fusion on the device (GpuFtMerger::FuseResident) — nothing is waited for, nothing comes back.  The extra entry lets the consumer
	// see a distance tie straddling the k-th place (which only the label-aware replay of SearchKnn can decide).  rowIds: device table
	// internal row -> row id (label >> 32), null while every label is (row << 32) (the fusion then takes the row as the id).
	struct ResidentKnn {
		const void* dDist = nullptr;
		const void* dRow = nullptr;
		const void* dCount = nullptr;
		void* stream = nullptr;
		uint32_t entries = 0;
		const void* dRowIds = nullptr;
	};
	ResidentKnn SearchKnnResident(const float* queryData, size_t k) const;
	rxgpu_index* DeviceIndex() const noexcept { return dev_; }

	bool IsQuantized() const noexcept { return false; }
	bool QuantizationAvailable() const noexcept { return false; }

	VectorMetric Metric() const noexcept { return metric_; }
	bool Sharded() const noexcept { return devices_.size() > 1; }   // the device mirror is row-range sharded over a device list
	size_t Dim() const noexcept { return dim_; }
	labeltype LabelByIdx(size_t idx) const noexcept { return labels_[idx]; }
	// statistics for tests: how many searches needed the tie replay
	size_t TieReplays() const noexcept { return tieReplays_; }

	// Query coalescing (on by default).  The reference's concurrency model is T planner threads each running its own SearchKnn over the shared
	// index (SURVEY §8b "Threading"); on a GPU a single scan already uses the whole HBM bandwidth, so concurrent scans would just queue.
	// Instead, calls that arrive while the device is busy are merged into ONE batched search (rxgpu_search_knn, nq <= 256): the thread
	// that finds the device idle runs the batch for everybody, the others sleep until their rows are back.  No waiting window is added:
	// a lone caller goes straight through.  Results are the batch-1 results, bit for bit.
	void EnableQueryCoalescing(bool on) noexcept { coalesce_ = on; }
	size_t CoalescedBatches() const noexcept { return coBatches_; }
	size_t CoalescedQueries() const noexcept { return coQueries_; }

private:
	void syncDevice() const;
	void markDirty(size_t idx);
	struct PendingQuery;
	void fetchTopK(const float* query, uint32_t kk, float* dist, uint32_t* row, uint32_t* count) const;
	void runBatch(std::vector<PendingQuery*>& batch) const;
	SearchResultQueue replayTies(const float* queryData, size_t k, float dk, const std::vector<uint32_t>* allowedRows) const;

	const VectorMetric metric_;
	const size_t dim_;
	const int device_;
	const std::vector<int> devices_;   // more than one entry: a sharded device mirror
	void createDeviceIndex();
	size_t maxElements_;
	size_t curElementCount_ = 0;

	std::vector<float> rows_;       // [maxElements][dim]  host master copy
	std::vector<labeltype> labels_; // [maxElements]
	std::vector<float> invNorms_;   // [maxElements], cosine only (DistCalculator::normCoefs_, hnswlib.h:80-92)
	std::unordered_map<labeltype, size_t> dictExternalToInternal_;

	// device mirror
	mutable std::mutex syncMtx_;
	mutable rxgpu_index* dev_ = nullptr;
	mutable std::vector<uint32_t> dirtyRows_;
	mutable bool dirtyAll_ = false;
	mutable bool needSync_ = false;
	mutable size_t tieReplays_ = 0;
	bool labelsIdentity_ = true;            // every label is (internal row << 32): the device needs no row-id table
	mutable bool rowIdsOnDevice_ = false;   // the table has been uploaded in full once

	bool coalesce_ = true;
	mutable std::mutex coMtx_;
	mutable std::condition_variable coCv_;
	mutable std::deque<PendingQuery*> coQueue_;
	mutable bool coLeader_ = false;
	mutable size_t coBatches_ = 0, coQueries_ = 0;
};

}  // namespace rxgpu::host
