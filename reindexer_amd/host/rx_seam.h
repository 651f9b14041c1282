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
#include "device_list.h"
#include "gpu_bruteforce_map.h"
#include "gpu_hnsw_map.h"

namespace rxgpu::host {

// RX_GPU_VECTOR_INDEXES=<device list> (device_list.h: "3", "0,1,2,3", "0-7") routes `vec_bf` / `hnsw` index definitions to the MI355X
// engines (unset / empty / malformed: the CPU engines).  More than one device: the brute-force Map range-shards its rows over the list
// (rxgpu_index_create_sharded, one RCCL all-gather per query batch) — BruteForceVectorIndex_New (hnsw_index.cc:578-581) reaches BASELINE
// configs[3] with no further change.
inline int GpuDeviceFromEnv() noexcept {
	const std::vector<int> d = GpuDevicesFromEnv();
	return d.empty() ? -1 : d[0];
}
inline std::vector<int> GpuDevicesOrDefault() {
	std::vector<int> d = GpuDevicesFromEnv();
	if (d.empty()) d.push_back(0);
	return d;
}

// hnswlib::BruteforceSearch's shape: (metric, dim, maxElements) + copy-with-capacity (bruteforce.h:16-17)
class GpuBruteforceMapInTree : public GpuBruteforceMap {
public:
	GpuBruteforceMapInTree(VectorMetric metric, size_t dim, size_t maxElements) : GpuBruteforceMap(metric, dim, maxElements, GpuDevicesOrDefault()) {}
	GpuBruteforceMapInTree(const GpuBruteforceMapInTree& other, size_t newMaxElements) : GpuBruteforceMap(other, newMaxElements) {}
};

// hnswlib::HierarchicalNSW<synchronization>'s shape: (IsArray, metric, dim, maxElements, M, efConstruction) + copy-with-capacity (hnsw.h:16-20)
template <Synchronization synchronization>
class GpuHnswMapT : public GpuHnswMap {
public:
	GpuHnswMapT(reindexer::IsArray, VectorMetric metric, size_t dim, size_t maxElements, size_t M, size_t efConstruction)
		: GpuHnswMap(metric, dim, maxElements, M, efConstruction, GpuDevicesOrDefault(), synchronization) {}
	GpuHnswMapT(const GpuHnswMapT& other, size_t newCapacity) : GpuHnswMap(other, newCapacity) {}

	// hnsw.h:32 / hnswalg.h:586-592
	size_t GetHash(FloatVectorId id) const {
		return ConstFloatVectorView{std::span<const float>{FloatPtrByExternalLabel(id.AsNumber()), Dim()}}.Hash();
	}
	// The ANN disk cache (HnswIndexBase::WriteIndexCache / LoadIndexCache, hnsw_index.cc:388-507): the reference's writer / reader objects
	// forwarded to the Map's own interfaces (ann_cache.h) — the stream is the CPU engine's, field for field, so a cache written by either
	// engine loads into the other, QuantizingParams of a quantised graph included (the Map comes back quantised when the reader asks for it).
	void SaveIndex(hnswlib::IWriter& writer, const std::atomic_int32_t& cancel) const {
		struct W final : AnnCacheWriter {
			hnswlib::IWriter& w;
			explicit W(hnswlib::IWriter& w_) : w(w_) {}
			void PutVarUInt(uint64_t v) override { w.PutVarUInt(v); }
			void PutVarUInt(uint32_t v) override { w.PutVarUInt(v); }
			void PutVarInt(int64_t v) override { w.PutVarInt(v); }
			void PutVarInt(int32_t v) override { w.PutVarInt(v); }
			void PutVString(std::string_view v) override { w.PutVString(v); }
			void PutFloat(float v) override { w.PutFloat(v); }
			void AppendPKByID(labeltype l) override { w.AppendPKByID(l); }
		} fw(writer);
		GpuHnswMap::SaveIndex(fw, cancel);
	}
	void LoadIndex(hnswlib::IReader& reader) {
		struct R final : AnnCacheReader {
			hnswlib::IReader& r;
			explicit R(hnswlib::IReader& r_) : r(r_) {}
			uint64_t GetVarUInt() override { return r.GetVarUInt(); }
			int64_t GetVarInt() override { return r.GetVarInt(); }
			std::string_view GetVString() override { return r.GetVString(); }
			float GetFloat() override { return r.GetFloat(); }
			labeltype ReadPkEncodedData(float* dest) override { return r.ReadPkEncodedData(dest); }
			bool WithQuantizer() const override { return r.WithQuantizer(); }
		} fr(reader);
		GpuHnswMap::LoadIndex(fr);   // reads the "quantised" flag and QuantizingParams itself (the reference's stream, field for field)
	}
	// SQ8 through the seam (HnswIndexBase::Quantize / SwitchMapOnQuantized, hnsw_index.cc:532-551): the Map samples its own rows like
	// QuantizingParams does (quantization_params.h:48-66) and keeps the parameters pending until the switch.
	void Quantize(const hnswlib::QuantizationConfig& c) {
		Sq8QuantizationConfig cfg;
		cfg.quantile = c.quantile;
		cfg.sampleSize = c.sampleSize;
		cfg.quantizationThreshold = c.quantizationThreshold;
		GpuHnswMap::Quantize(cfg);
	}
};
using GpuHnswMapST = GpuHnswMapT<Synchronization::None>;
using GpuHnswMapMT = GpuHnswMapT<Synchronization::OnInsertions>;

}  // namespace rxgpu::host
