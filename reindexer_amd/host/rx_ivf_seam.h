// In-tree adapter (RXGPU_IN_TREE only) for the IVF index: GpuIvfFlat behind the calls IvfIndex makes on its FAISS pair
// (cpp_src/core/index/float_vector/ivf_index.cc) — faiss::IndexFlat `space_` + faiss::IndexIVFFlat `map_`:
//   upsert            :88-108    space_->add / map_->add_with_ids, training once more than 39 x nCentroids vectors are indexed (:96-104, trainIdx :469-487)
//   del               :114-141   remove_ids(IDSelectorArray)
//   select / selectRaw :143-272, 305-425   map->search(1, x, k, D, I, &IVFSearchParameters) / map->range_search(1, x, radius, &RangeSearchResult, ...)
//                                 — IvfIndex's search helpers are templates over `map`: GpuIvfInTree offers exactly those two members
//   reconstruct / getFloatVectorViewImpl :455-467, 489-497
//   RebuildCentroids  :625-678
// integration/patches/0004-ivf_index-gpu-ivf.patch routes them here when RX_GPU_VECTOR_INDEXES is set; tests/test_seam_compile.py compiles it.
// Ids are FloatVectorId numbers from the first vector on (GpuIvfFlat keeps ids in its flat phase too), so `prepareId` is the identity.
#pragma once
#if !defined(RXGPU_IN_TREE)
#error "rx_ivf_seam.h is for the build inside cpp_src (define RXGPU_IN_TREE)"
#endif

#include <algorithm>
#include <cstring>
#include <memory>
#include <vector>

#include "core/enums.h"
#include "device_list.h"
#include "faiss/IndexIVF.h"
#include "faiss/impl/AuxIndexStructures.h"
#include "gpu_ivf_flat.h"

namespace rxgpu::host {

class GpuIvfInTree {
public:
	using idx_t = faiss::idx_t;

	GpuIvfInTree(VectorMetric metric, size_t dim, size_t nCentroids, int device) : ivf_(std::make_unique<GpuIvfFlat>(metric, dim, nCentroids, device)) {}
	// IvfIndex's copy constructor (copy-on-write tx clone, ivf_index.cc:72-79: faiss::clone_index of whichever index is live)
	GpuIvfInTree(const GpuIvfInTree& o) : ivf_(std::make_unique<GpuIvfFlat>(*o.ivf_, o.ivf_->Device())) {}

	// upsert, both phases (ivf_index.cc:88-108): the vector joins the flat storage; the first one past the training size trains the lists
	void Upsert(const float* vec, idx_t id) {
		ivf_->AddWithIds(1, vec, &id);
		if (!ivf_->IsTrained() && ivf_->NTotal() > GpuIvfFlat::TrainingSize(ivf_->NList())) ivf_->Train();
	}
	size_t Remove(idx_t id) { return ivf_->RemoveIds(&id, 1); }
	bool IsTrained() const noexcept { return ivf_->IsTrained(); }
	size_t NTotal() const noexcept { return ivf_->NTotal(); }
	const float* VectorById(idx_t id) const { return ivf_->VectorById(id); }
	// RebuildCentroids(dataPart) (:625-678): a new training over the stored vectors (FAISS's own subsample rule — at most 256 points per
	// centroid — picks the training set; dataPart only lowers that bound in the reference), every vector re-assigned
	void RebuildCentroids() {
		if (ivf_->IsTrained()) ivf_->Train();
	}
	size_t AllocatedMemSize() const noexcept { return ivf_->NTotal() * (ivf_->Dim() * sizeof(float) + sizeof(idx_t) + 2 * sizeof(uint32_t)); }

	// faiss::Index::search as IvfIndex's helpers call it: n == 1
	void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels, const faiss::SearchParameters* params = nullptr) const {
		for (idx_t i = 0; i < n; ++i) ivf_->Search(x + size_t(i) * ivf_->Dim(), size_t(k), nprobeOf(params), distances + size_t(i) * k, labels + size_t(i) * k);
	}
	// faiss::Index::range_search: result->lims is allocated by the caller's RangeSearchResult(nq); labels / distances by do_allocation()
	void range_search(idx_t n, const float* x, float radius, faiss::RangeSearchResult* result, const faiss::SearchParameters* params = nullptr) const {
		const size_t nq = static_cast<size_t>(n);
		std::vector<std::vector<float>> d(nq);
		std::vector<std::vector<GpuIvfFlat::idx_t>> l(nq);
		for (idx_t i = 0; i < n; ++i) {
			ivf_->RangeSearch(x + size_t(i) * ivf_->Dim(), radius, nprobeOf(params), d[size_t(i)], l[size_t(i)]);
			result->lims[i] = d[size_t(i)].size();   // do_allocation() turns the per-query counts into offsets (AuxIndexStructures.cpp:37-50)
		}
		result->do_allocation();
		for (idx_t i = 0; i < n; ++i) {
			std::copy(d[size_t(i)].begin(), d[size_t(i)].end(), result->distances + result->lims[i]);
			std::copy(l[size_t(i)].begin(), l[size_t(i)].end(), result->labels + result->lims[i]);
		}
	}

private:
	static size_t nprobeOf(const faiss::SearchParameters* params) {
		const auto* p = dynamic_cast<const faiss::IVFSearchParameters*>(params);
		return p ? p->nprobe : 1;
	}
	std::unique_ptr<GpuIvfFlat> ivf_;
};

}  // namespace rxgpu::host
