// The reference's ANN disk cache of an HNSW index: what HnswIndexBase<Map>::WriteIndexCache / LoadIndexCache hand to the Map
// (cpp_src/core/index/float_vector/hnsw_index.cc:388-507) and what HierarchicalNSW::SaveIndex / LoadIndex + HierarchicalNSWImpl::SaveIndex
// and its reader constructor put into / take out of it (hnswlib/hnsw.cc:41-70, hnswlib/hnswalg.h:1213-1263, 297-409).
//
// The stream is written through an abstract writer (the reference's hnswlib::IWriter, hnsw_interface.h:47-58: var-ints, length-prefixed
// strings, floats and "the primary key of this row" — the encoding belongs to the caller) and read back through hnswlib::IReader (:60-70),
// whose ReadPkEncodedData() resolves a primary key to the row and copies its vector out of the namespace.  These two interfaces mirror
// them name for name, so that rx_seam.h adapts the reference's objects with one-line forwarders and a cache written by the CPU engine
// loads into the GPU Map and vice versa (tests/test_ann_cache.py: both directions against the reference engine compiled in place).
#pragma once

#include <cstddef>
#include <cstdint>
#include <string_view>

#include "rx_types.h"

namespace rxgpu::host {

class AnnCacheWriter {
public:
	virtual ~AnnCacheWriter() = default;
	virtual void PutVarUInt(uint64_t) = 0;
	virtual void PutVarUInt(uint32_t) = 0;
	virtual void PutVarInt(int64_t) = 0;
	virtual void PutVarInt(int32_t) = 0;
	virtual void PutVString(std::string_view) = 0;
	virtual void PutFloat(float) = 0;
	virtual void AppendPKByID(labeltype) = 0;
};

class AnnCacheReader {
public:
	virtual ~AnnCacheReader() = default;
	virtual uint64_t GetVarUInt() = 0;
	virtual int64_t GetVarInt() = 0;
	virtual std::string_view GetVString() = 0;
	virtual float GetFloat() = 0;
	virtual labeltype ReadPkEncodedData(float* destBuf) = 0;   // the row's label; its vector is copied to destBuf
	virtual bool WithQuantizer() const = 0;
};

// QuantizingParams::Serialize / Deserialize (scalar_quantization/quantization_params.h:69-96) read past, for a cache the CPU engine wrote
// from a quantised graph: the links are the same graph, the codes are not stored (they are recomputed from the rows on load).
constexpr uint64_t kAnnCacheQuantizationParamsVersion = 1;

}  // namespace rxgpu::host
