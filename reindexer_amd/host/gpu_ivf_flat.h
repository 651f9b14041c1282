// GpuIvfFlat — IVF-Flat on the GPU engines (first cut).  Stands in for the pair the reference's IvfIndex drives
// (cpp_src/core/index/float_vector/ivf_index.cc): faiss::IndexFlat `space_` while fewer than 39 * nCentroids vectors are indexed
// (ivf_index.h:62, ivf_index.cc:88-108) and faiss::IndexIVFFlat `map_` afterwards — train / add_with_ids / remove_ids / search / range_search
// with faiss::IVFSearchParameters::nprobe (ivf_index.cc:355-372, 143-272, 469-487).
//
// FAISS is a patched copy vendored in the reference (cpp_src/vendor_subdirs/faiss; its fvec_L2sqr / fvec_inner_product call the reference's
// own vector_dists functions).  Parity status:
//   * SEARCH is pinned: the definition the tests hold this class to — the exact (dist,row)-ordered search, in the engine's distance arithmetic,
//     over the rows of the nprobe nearest lists; cosine through the stored 1/|row| — returns the real FAISS's labels and distance bits for
//     search / range_search / remove_ids given FAISS's trained state (tests/test_ivf_oracle.py; FAISS compiled in place into
//     oracle/_ref/libref_ivf.so with a triple-loop sgemm standing in for BLAS).  tests/test_gpu_ivf.py checks the GPU index against that
//     definition on its own centroids.
//   * TRAINING is pinned too: Train() restates faiss::Clustering::train_encoded as Level1Quantizer::train_q1 configures it for
//     IndexIVFFlat (IndexIVF.cpp:43-49, 76-88; Clustering.cpp:83-137, 153-290, 330-560) —
//       subsample        more than 256 points per centroid: the first nlist * 256 of rand_perm(n, seed 1234)
//       initial centroids the first nlist points of rand_perm(n, seed + 1) (faiss::RandomGenerator = std::mt19937, rand_int = mt() % max)
//       10 iterations     k = 1 search of the coarse quantiser per point -> compute_centroids (single-precision sums in data order, x 1/count)
//                         -> split_clusters (its own RandomGenerator(1234) every call, +-1/1024) -> spherical renormalisation for
//                         inner product / cosine (fvec_renorm_L2: a sequential fmaf chain of squares in the AVX-512 build, 1.0 / sqrtf)
//       add_with_ids      vector -> list of its nearest centroid (cosine: x * 1/|x|; the coarse quantiser is IndexFlatCosine:
//                         inner product x the stored 1/|centroid|)
//     and tests/test_gpu_ivf.py holds centroids and inverted lists to the BITS of the vendored FAISS built in place, L2 / IP / cosine, incl.
//     the subsample and the empty-cluster split.  One qualification: the k-means assignment is the exact k = 1 search on the device (the
//     reference's own distance functions); FAISS takes its BLAS shortcut (|x|^2 + |y|^2 - 2 x.y through sgemm) for batches of 20+ queries,
//     whose rounding depends on the BLAS library the reference finds at run time — the pin is against FAISS with that shortcut off
//     (faiss::distance_compute_blas_threshold = INT_MAX), the form that is a function of the reference's code alone.
//
// MI355X mapping: both halves are the brute-force engine.  The coarse quantiser is a KNN over the nlist centroids (training assigns 256
// points per call: the matrix-core batch path); the list scan is knn_scan_subset / knn_range_subset over the row list of the probed
// inverted lists, so a query reads nprobe / nlist of the corpus.  The lists are mirrored into HBM as CSR (rebuilt after a mutation) and a
// search with nprobe <= 64 is ONE C-ABI call (rxgpu_search_knn_lists): coarse search, list union (bitmap), row list and scan all on the
// device; the host copy of the lists serves mutations, range search and wider probes.
#pragma once

#include <cstdint>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "rx_types.h"

struct rxgpu_index;

namespace rxgpu::host {

// Threading: Search / RangeSearch / ProbedRows are const and may run concurrently (every call checks out its own device scratch); Train,
// AddWithIds, RemoveIds and Reset need exclusive access — the reference serialises them the same way under the namespace lock.
class GpuIvfFlat {
public:
	using idx_t = int64_t;   // faiss::idx_t

	GpuIvfFlat(VectorMetric metric, size_t dim, size_t nlist, int device = 0);
	// a copy with a device mirror of its own (IvfIndex's copy-on-write clone, ivf_index.cc:72-79: faiss::clone_index): vectors, ids, centroids
	// and inverted lists as they are — nothing is re-trained or re-assigned
	GpuIvfFlat(const GpuIvfFlat& other, int device);
	~GpuIvfFlat();
	GpuIvfFlat(const GpuIvfFlat&) = delete;
	GpuIvfFlat& operator=(const GpuIvfFlat&) = delete;

	static constexpr size_t TrainingSize(size_t nCentroids) noexcept { return nCentroids * 39; }   // ivf_index.h:62

	bool IsTrained() const noexcept { return trained_; }
	size_t NTotal() const noexcept { return count_; }
	size_t NList() const noexcept { return nlist_; }
	size_t Dim() const noexcept { return dim_; }
	int Device() const noexcept { return device_; }
	VectorMetric Metric() const noexcept { return metric_; }
	// the stored vector of an id (IvfIndex::reconstruct / getFloatVectorViewImpl, ivf_index.cc:455-467, 489-497); throws when absent
	const float* VectorById(idx_t id) const;
	size_t ListSize(size_t list) const { return lists_.at(list).size(); }
	// the ids held by an inverted list, in row order (tests: compared with faiss::InvertedLists::get_ids)
	void ListIds(size_t list, idx_t* out) const {
		const auto& l = lists_.at(list);
		for (size_t i = 0; i < l.size(); ++i) out[i] = ids_[l[i]];
	}
	const std::vector<float>& Centroids() const noexcept { return centroids_; }

	// IndexIVF::train over the vectors ALREADY added (the reference trains on space_'s content, ivf_index.cc:96-108) and moves every
	// vector into its list.  seed: Clustering::seed (1234).
	void Train(int seed = 1234);
	// add_with_ids: before training the vectors only join the flat storage (the reference's `space_` phase); ids must be unique
	void AddWithIds(size_t n, const float* x, const idx_t* ids);
	// remove_ids(IDSelectorArray): returns how many were present
	size_t RemoveIds(const idx_t* ids, size_t n);
	void Reset();

	// search(1, x, k, distances, labels, nprobe): best first; L2: squared distance ascending, inner product / cosine: similarity descending;
	// labels[i] = -1 past the last hit (FAISS convention).  Untrained: exact search over everything (IndexFlat).
	void Search(const float* x, size_t k, size_t nprobe, float* distances, idx_t* labels) const;
	// n queries, x [n][dim] -> distances / labels [n][k]: the searches run side by side on a few streams
	void SearchBatch(size_t n, const float* x, size_t k, size_t nprobe, float* distances, idx_t* labels) const;
	// range_search: L2: dist < radius; inner product / cosine: similarity > radius; sorted best first
	void RangeSearch(const float* x, float radius, size_t nprobe, std::vector<float>& distances, std::vector<idx_t>& labels) const;

	// the rows of the `nprobe` nearest lists for x (ascending) — what the list scan is run over; exposed for tests / tools
	std::vector<uint32_t> ProbedRows(const float* x, size_t nprobe) const;

private:
	void prepareQuery(const float* x, std::vector<float>& q) const;
	void coarse(const float* q, size_t nprobe, std::vector<uint32_t>& lists) const;
	void assign(const float* xPrepared, size_t n, std::vector<uint32_t>& out) const;   // nearest centroid per (prepared) vector, GPU
	void uploadCentroids() const;
	void reserveRows(size_t need);
	void listInsert(uint32_t list, uint32_t row);
	void listErase(uint32_t list, uint32_t row);
	void listRename(uint32_t list, uint32_t from, uint32_t to);   // a row number changes (swap-delete of the flat storage)
	void scanAppend(uint32_t list, uint32_t row);
	// faiss tie semantics over the candidates with internal distance <= worst (see Search)
	void replayTies(const float* q, size_t k, size_t nprobe, float worst, float* distances, idx_t* labels) const;
	void orderTies(size_t n, float* distances, idx_t* labels) const;
	float toFaiss(float internal) const noexcept { return metric_ == VectorMetric::L2 ? internal : -internal; }

	const VectorMetric metric_;
	const size_t dim_, nlist_;
	const int device_;
	bool trained_ = false;
	size_t count_ = 0, capacity_ = 0;

	std::vector<float> rows_;        // host master copy [count][dim] (raw vectors)
	std::vector<float> invNorms_;    // cosine: 1 / |row|
	std::vector<idx_t> ids_;         // [count]
	std::vector<uint32_t> listOf_;   // [count], valid once trained
	std::unordered_map<idx_t, uint32_t> idToRow_;   // DirectMap::Hashtable (ivf_index.cc:470)
	std::vector<std::vector<uint32_t>> lists_;      // per centroid: its rows, ascending
	// the same lists in faiss::ArrayInvertedLists order — append on add, the list's last entry moves into the hole on remove
	// (DirectMap::remove_ids, Hashtable type) — and every row's position there.  Only exact ties in a search read it: the scanner keeps
	// the first-scanned of equal distances at the k-th place (IndexIVFFlat.cpp scan_codes: `if (C::cmp(simi[0], dis))` is strict).
	std::vector<std::vector<uint32_t>> scan_;
	std::vector<uint32_t> scanPos_;                 // [count]
	std::vector<float> centroids_;   // [nlist][dim]

	void syncLists() const;                     // CSR mirror of lists_ in HBM (rxgpu_index_set_lists), rebuilt after a mutation
	mutable bool listsDirty_ = true;
	mutable std::mutex listsMtx_;               // concurrent searches: one of them uploads the lists
	mutable rxgpu_index* dev_ = nullptr;        // the vectors
	mutable rxgpu_index* devCentroids_ = nullptr;
};

}  // namespace rxgpu::host
