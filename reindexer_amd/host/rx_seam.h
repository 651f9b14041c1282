// In-tree adapters (RXGPU_IN_TREE only): the constructor shapes and the few HNSW-only members HnswIndexBase<Map> expects from its Map
// (cpp_src/core/index/float_vector/hnsw_index.cc:47-70, 425-560; hnswlib::HierarchicalNSW, hnsw.h:14-126), on top of the GPU Maps.
// integration/patches/0001-hnsw_index-gpu-maps.patch instantiates HnswIndexBase over these three types; tests/test_seam_compile.py compiles it.
#pragma once
#if !defined(RXGPU_IN_TREE)
#error "rx_seam.h is for the build inside cpp_src (define RXGPU_IN_TREE)"
#endif

#include <atomic>
#include <cstdlib>
#include <span>
#include <stdexcept>

#include "core/enums.h"
#include "core/index/float_vector/scalar_quantization/quantization_params.h"
#include "gpu_bruteforce_map.h"
#include "gpu_hnsw_map.h"

namespace rxgpu::host {

// RX_GPU_VECTOR_INDEXES=<device> routes `vec_bf` / `hnsw` index definitions to the MI355X engines (unset / empty: the CPU engines).
inline int GpuDeviceFromEnv() noexcept {
	const char* e = std::getenv("RX_GPU_VECTOR_INDEXES");
	return (e && *e) ? std::atoi(e) : -1;
}

// hnswlib::BruteforceSearch's shape: (metric, dim, maxElements) + copy-with-capacity (bruteforce.h:16-17)
class GpuBruteforceMapInTree : public GpuBruteforceMap {
public:
	GpuBruteforceMapInTree(VectorMetric metric, size_t dim, size_t maxElements) : GpuBruteforceMap(metric, dim, maxElements, std::max(0, GpuDeviceFromEnv())) {}
	GpuBruteforceMapInTree(const GpuBruteforceMapInTree& other, size_t newMaxElements) : GpuBruteforceMap(other, newMaxElements) {}
};

// hnswlib::HierarchicalNSW<synchronization>'s shape: (IsArray, metric, dim, maxElements, M, efConstruction) + copy-with-capacity (hnsw.h:16-20)
template <Synchronization synchronization>
class GpuHnswMapT : public GpuHnswMap {
public:
	GpuHnswMapT(reindexer::IsArray, VectorMetric metric, size_t dim, size_t maxElements, size_t M, size_t efConstruction)
		: GpuHnswMap(metric, dim, maxElements, M, efConstruction, std::max(0, GpuDeviceFromEnv()), synchronization) {}
	GpuHnswMapT(const GpuHnswMapT& other, size_t newCapacity) : GpuHnswMap(other, newCapacity) {}

	// hnsw.h:32 / hnswalg.h:586-592
	size_t GetHash(FloatVectorId id) const {
		return ConstFloatVectorView{std::span<const float>{FloatPtrByExternalLabel(id.AsNumber()), Dim()}}.Hash();
	}
	// The ANN disk cache is not provided by the GPU engine: the index type is registered as non-cacheable (like brute force).  SQ8: the Map
	// itself quantises (GpuHnswMap::Quantize(minQ, maxQ) + the device search over codes), but HnswIndexBase::Quantize() derives the range by
	// sampling the reference's own graph storage (QuantizingParams over an HNSWView, quantization_params.h:48-63), which this adapter does
	// not expose yet — so through the seam QuantizationAvailable() stays false and these are never reached; they fail loudly if called.
	bool QuantizationAvailable() const noexcept { return false; }
	void SaveIndex(hnswlib::IWriter&, const std::atomic_int32_t&) const { throw std::logic_error("GpuHnswMap: the ANN disk cache is not supported"); }
	void LoadIndex(hnswlib::IReader&) { throw std::logic_error("GpuHnswMap: the ANN disk cache is not supported"); }
	void Quantize(const hnswlib::QuantizationConfig&) { throw std::logic_error("GpuHnswMap: quantization is not supported"); }
	void SwitchMapOnQuantized() { throw std::logic_error("GpuHnswMap: quantization is not supported"); }
};
using GpuHnswMapST = GpuHnswMapT<Synchronization::None>;
using GpuHnswMapMT = GpuHnswMapT<Synchronization::OnInsertions>;

}  // namespace rxgpu::host
