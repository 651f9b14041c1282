#include "gpu_hnsw_map.h"

#include <cstring>

#include <algorithm>
#include <queue>
#include <stdexcept>
#include <string>
#include <vector>

#include "rxgpu.h"

namespace rxgpu::host {

namespace {
[[noreturn]] void throwDevice(const char* what) { throw std::runtime_error(std::string(what) + ": " + rxgpu_last_error()); }
}  // namespace

GpuHnswMap::GpuHnswMap(VectorMetric metric, size_t dim, size_t maxElements, size_t M, size_t efConstruction, int device,
					   Synchronization synchronization)
	: graph_(metric, dim, maxElements, M, efConstruction), device_(device), synchronization_(synchronization) {
	if (2 * graph_.M() > 128) throw std::logic_error("GpuHnswMap: the GPU engine supports M <= 64");
	if (synchronization_ == Synchronization::OnInsertions) graph_.EnableConcurrentInserts();
	if (rxgpu_index_create(int(metric), uint32_t(dim), maxElements, device_, &dev_) != RXGPU_OK) {
		throwDevice("GpuHnswMap: device index creation failed");
	}
}

// ------------------------------------------------------------------------------------------------ the Map over a device list
struct GpuHnswMap::ShardedState {
	std::vector<int> devices;
	rxgpu_index* parent = nullptr;                    // rxgpu_index_create_sharded: owns the shards' device indexes
	std::vector<std::unique_ptr<GpuHnswMap>> maps;    // shard s: a single-device Map over rxgpu_index_shard(parent, s)
	size_t shardRows = 0;                             // capacity of every shard = the parent's shard_rows (global row = s * shardRows + local)
	size_t maxElements = 0;                           // what the caller asked for (MaxElements())
	size_t M = 0, efConstruction = 0;
	std::mutex routeMtx;                              // label -> shard decisions (the insert itself runs outside it)
	std::unordered_map<labeltype, uint32_t> shardOf;  // every label ever routed (a recycled slot leaves a harmless stale entry)
	std::vector<size_t> routed;                       // new labels sent to shard s so far
	~ShardedState() {
		maps.clear();
		if (parent) rxgpu_index_destroy(parent);
	}
};

namespace {
size_t shardRowsFor(size_t maxElements, size_t n) { return ((std::max<size_t>(maxElements, 1) + n - 1) / n + 31) & ~size_t(31); }
}  // namespace

void GpuHnswMap::shCreateParent(size_t shardRows) {
	ShardedState& S = *sh_;
	rxgpu_index* parent = nullptr;
	const uint64_t cap = uint64_t(shardRows) * S.devices.size();   // shard_rows of the handle == shardRows (a multiple of 32)
	if (rxgpu_index_create_sharded(int(graph_.Metric()), uint32_t(graph_.Dim()), cap, uint32_t(S.devices.size()), S.devices.data(), &parent) != RXGPU_OK) {
		throwDevice("GpuHnswMap: sharded device index creation failed");
	}
	if (rxgpu_index_shard_rows(parent) != shardRows) {
		rxgpu_index_destroy(parent);
		throw std::logic_error("GpuHnswMap: unexpected shard size of the device index");
	}
	for (size_t s = 0; s < S.maps.size(); ++s) S.maps[s]->rebindDevice(rxgpu_index_shard(parent, uint32_t(s)));   // before the old handles go
	if (S.parent) rxgpu_index_destroy(S.parent);
	S.parent = parent;
	S.shardRows = shardRows;
}

GpuHnswMap::GpuHnswMap(VectorMetric metric, size_t dim, size_t maxElements, size_t M, size_t efConstruction, std::vector<int> devices,
					   Synchronization synchronization)
	: graph_(metric, dim, 1, M, efConstruction), device_(devices.empty() ? 0 : devices[0]), synchronization_(synchronization), ownsDev_(false) {
	if (devices.empty()) throw std::logic_error("GpuHnswMap: empty device list");
	if (2 * graph_.M() > 128) throw std::logic_error("GpuHnswMap: the GPU engine supports M <= 64");
	if (devices.size() == 1) {   // an ordinary single-device Map
		ownsDev_ = true;
		graph_.Resize(maxElements);
		if (synchronization_ == Synchronization::OnInsertions) graph_.EnableConcurrentInserts();
		if (rxgpu_index_create(int(metric), uint32_t(dim), maxElements, device_, &dev_) != RXGPU_OK) throwDevice("GpuHnswMap: device index creation failed");
		return;
	}
	sh_ = std::make_unique<ShardedState>();
	ShardedState& S = *sh_;
	S.devices = std::move(devices);
	S.maxElements = maxElements;
	S.M = M;
	S.efConstruction = efConstruction;
	S.routed.assign(S.devices.size(), 0);
	const size_t rows = shardRowsFor(maxElements, S.devices.size());
	shCreateParent(rows);
	for (size_t s = 0; s < S.devices.size(); ++s) {
		S.maps.emplace_back(new GpuHnswMap(metric, dim, rows, M, efConstruction, S.devices[s], synchronization, rxgpu_index_shard(S.parent, uint32_t(s))));
	}
}

GpuHnswMap::GpuHnswMap(VectorMetric metric, size_t dim, size_t maxElements, size_t M, size_t efConstruction, int device, Synchronization synchronization,
					   rxgpu_index* external)
	: graph_(metric, dim, maxElements, M, efConstruction), device_(device), dev_(external), synchronization_(synchronization), ownsDev_(false) {
	if (synchronization_ == Synchronization::OnInsertions) graph_.EnableConcurrentInserts();
	coalesce_ = false;   // the fan-out above this shard batches nothing; a direct call to a shard stays a direct call
}

GpuHnswMap::GpuHnswMap(const GpuHnswMap& other, size_t newCapacity, rxgpu_index* external)
	: graph_(other.graph_, newCapacity), device_(other.device_), dev_(external), synchronization_(other.synchronization_), ownsDev_(false) {
	if (synchronization_ == Synchronization::OnInsertions) graph_.EnableConcurrentInserts();
	coalesce_ = false;
}

void GpuHnswMap::rebindDevice(rxgpu_index* external) {
	std::lock_guard<std::mutex> lk(syncMtx_);
	dev_ = external;
	syncedRows_ = 0;
	syncedCodes_ = 0;
	graphOnDevice_ = false;
	graphDirty_ = true;
	codesDirty_ = quantized_;
}

GpuHnswMap::GpuHnswMap(const GpuHnswMap& other, size_t newCapacity)
	: graph_(other.graph_, other.sh_ ? 1 : newCapacity), device_(other.device_), synchronization_(other.synchronization_), ownsDev_(!other.sh_) {
	if (other.sh_) {   // copy-on-write tx clone of a Map over a device list: every shard cloned, a sharded handle of its own
		const ShardedState& O = *other.sh_;
		sh_ = std::make_unique<ShardedState>();
		ShardedState& S = *sh_;
		S.devices = O.devices;
		S.maxElements = std::max(O.maxElements, newCapacity);
		S.M = O.M;
		S.efConstruction = O.efConstruction;
		S.shardOf = O.shardOf;
		S.routed = O.routed;
		const size_t rows = std::max(O.shardRows, shardRowsFor(S.maxElements, S.devices.size()));
		shCreateParent(rows);
		for (size_t s = 0; s < O.maps.size(); ++s) {
			S.maps.emplace_back(new GpuHnswMap(*O.maps[s], rows, rxgpu_index_shard(S.parent, uint32_t(s))));
		}
		return;
	}
	if (synchronization_ == Synchronization::OnInsertions) graph_.EnableConcurrentInserts();
	if (rxgpu_index_create(int(graph_.Metric()), uint32_t(graph_.Dim()), graph_.MaxElements(), device_, &dev_) != RXGPU_OK) {
		throwDevice("GpuHnswMap: device index creation failed");
	}
}

GpuHnswMap::~GpuHnswMap() {
	sh_.reset();
	if (dev_ && ownsDev_) rxgpu_index_destroy(dev_);
}

size_t GpuHnswMap::ShardCount() const noexcept { return sh_ ? sh_->maps.size() : 0; }
size_t GpuHnswMap::ShardRows() const noexcept { return sh_ ? sh_->shardRows : 0; }
const GpuHnswMap& GpuHnswMap::Shard(size_t s) const {
	if (!sh_ || s >= sh_->maps.size()) throw std::logic_error("GpuHnswMap: no such shard");
	return *sh_->maps[s];
}
rxgpu_index* GpuHnswMap::DeviceIndex() const noexcept { return sh_ ? sh_->parent : dev_; }
size_t GpuHnswMap::shMaxElements() const noexcept { return sh_->maxElements; }
size_t GpuHnswMap::shCount(bool deleted) const noexcept {
	size_t n = 0;
	for (const auto& m : sh_->maps) n += deleted ? m->graph_.DeletedCount() : m->graph_.Count();
	return n;
}
size_t GpuHnswMap::shAllocated() const noexcept {
	size_t n = sizeof(ShardedState) + sh_->shardOf.size() * (sizeof(labeltype) + sizeof(uint32_t) + 2 * sizeof(void*));
	for (const auto& m : sh_->maps) n += m->graph_.AllocatedMemSize();
	return n;
}
labeltype GpuHnswMap::shLabel(tableint id) const { return Shard(id / sh_->shardRows).graph_.Label(tableint(id % sh_->shardRows)); }
bool GpuHnswMap::shIsDeleted(tableint id) const noexcept { return sh_->maps[id / sh_->shardRows]->graph_.IsDeleted(tableint(id % sh_->shardRows)); }
const float* GpuHnswMap::shFloatPtr(labeltype label) const {
	ShardedState& S = *sh_;
	uint32_t s;
	{
		std::lock_guard<std::mutex> lk(S.routeMtx);
		const auto it = S.shardOf.find(label);
		if (it == S.shardOf.end()) throw std::runtime_error("Label not found");
		s = it->second;
	}
	const HnswGraph& g = S.maps[s]->graph_;
	return g.Vector(g.InternalId(label));
}

// Which shard takes the point: the one that holds the label already (an update in place, or a re-insert over its delete-marked slot), else
// the first shard that still has room — shards fill in order, i.e. contiguous ranges of the insertion sequence ("row range").
GpuHnswMap& GpuHnswMap::shRoute(labeltype label) {
	ShardedState& S = *sh_;
	std::lock_guard<std::mutex> lk(S.routeMtx);
	if (const auto it = S.shardOf.find(label); it != S.shardOf.end()) {
		// the entry is only a hint: the label's slot may have been recycled by another label since (a delete-marked slot taken over,
		// hnswalg.h:1401-1470) — then the label is new again and goes wherever there is room
		if (S.maps[it->second]->graph_.HasLabelSync(label)) return *S.maps[it->second];
		S.shardOf.erase(it);
	}
	for (size_t s = 0; s < S.maps.size(); ++s) {
		if (S.routed[s] < S.shardRows) {
			++S.routed[s];
			S.shardOf.emplace(label, uint32_t(s));
			return *S.maps[s];
		}
	}
	for (size_t s = 0; s < S.maps.size(); ++s) {   // every range is full: a delete-marked slot is recycled (addPoint, hnswalg.h:1401-1470)
		if (S.maps[s]->graph_.DeletedCount()) {
			S.shardOf[label] = uint32_t(s);
			return *S.maps[s];
		}
	}
	throw std::runtime_error("The number of elements exceeds the specified limit");   // hnswalg.h:1445
}

void GpuHnswMap::shSyncAll() const {
	for (const auto& m : sh_->maps) {
		if (m->graph_.Count()) m->syncDevice();
	}
}

void GpuHnswMap::AddPointNoLock(ConstFloatVectorView vect, FloatVectorId id) {
	if (sh_) return shRoute(id.AsNumber()).AddPointNoLock(vect, id);
	graph_.AddPoint(vect.Data(), id.AsNumber());
	graphDirty_ = true;
}

void GpuHnswMap::AddPointConcurrent(ConstFloatVectorView vect, FloatVectorId id) {
	if (synchronization_ == Synchronization::None) {
		throw std::logic_error("This HNSW index does not support concurrent insertions");   // hnswalg.h:1393-1399 (Synchronization::None)
	}
	if (sh_) return shRoute(id.AsNumber()).AddPointConcurrent(vect, id);
	graph_.AddPointConcurrent(vect.Data(), id.AsNumber());
	graphDirty_ = true;
}

void GpuHnswMap::MarkDelete(FloatVectorId id) {
	if (sh_) {
		ShardedState& S = *sh_;
		uint32_t s;
		{
			std::lock_guard<std::mutex> lk(S.routeMtx);
			const auto it = S.shardOf.find(id.AsNumber());
			if (it == S.shardOf.end()) throw std::runtime_error("Label not found");   // hnswalg.h:1310
			s = it->second;
		}
		return S.maps[s]->MarkDelete(id);
	}
	graph_.MarkDelete(id.AsNumber());
	deletedDirty_ = true;
}

void GpuHnswMap::ResizeIndex(size_t newMaxElements) {
	if (sh_) {   // every range grows; the sharded device handle has a fixed shard size, so a new one takes over (the next search mirrors again)
		ShardedState& S = *sh_;
		if (newMaxElements < shCount(false)) throw std::runtime_error("Cannot resize, max element is less than the current number of elements");   // hnswalg.h:1187
		S.maxElements = newMaxElements;
		const size_t rows = shardRowsFor(newMaxElements, S.devices.size());
		if (rows > S.shardRows) {
			for (auto& m : S.maps) m->ResizeIndex(rows);
			shCreateParent(rows);
		}
		return;
	}
	graph_.Resize(newMaxElements);
	graphDirty_ = true;
}

// The cache of a Map over a device list.  The reference's stream holds ONE graph (hnswalg.h:1213-1263); here there is a graph per shard, so
// the stream says so in a way every single-graph reader understands: where one graph states its capacity and its element count, this one
// states capacity 0 and count = the number of shards — HierarchicalNSWImpl's reader (hnswalg.h:297-306), HnswGraph::LoadIndex and a
// single-device Map all stop at those two fields with "Current elements count is larger than max elements count", which
// HnswIndexBase::LoadIndexCache (hnsw_index.cc:452-507) takes as a cache to drop and rebuild.  Behind them: format version, the Map's
// capacity, the rows of a shard, then every shard's graph as HnswGraph::SaveIndex writes it.  Labels route back to their shard on load.
namespace {
constexpr uint64_t kShardedAnnCacheVersion = 1;
}  // namespace

void GpuHnswMap::saveQuantizingParams(AnnCacheWriter& writer) const {
	// serializeQuantizingParams (hnsw.cc:56-62) + QuantizingParams::Serialize (quantization_params.h:83-96)
	writer.PutVarUInt(uint32_t(quantized_ ? 1 : 0));
	if (quantized_) {
		writer.PutVarUInt(uint64_t(kAnnCacheQuantizationParamsVersion));
		writer.PutVarInt(int32_t(0));                                   // QuantizationType::ScalarQuantization8bit
		writer.PutFloat(sq8Config_.quantile ? *sq8Config_.quantile : 0.f);
		writer.PutVarUInt(uint64_t(sq8Config_.sampleSize));
		writer.PutVarUInt(uint64_t(sq8Config_.quantizationThreshold));
		writer.PutFloat(sq8_.minQ);
		writer.PutFloat(sq8_.maxQ);
		writer.PutFloat(sq8_.alpha);
		writer.PutFloat(sq8_.alpha_2);
		writer.PutFloat(sq8_.delta);
	}
}

void GpuHnswMap::SaveIndex(AnnCacheWriter& writer, const std::atomic_int32_t& cancel) const {
	saveQuantizingParams(writer);
	if (sh_) {
		const ShardedState& S = *sh_;
		writer.PutVarUInt(uint64_t(0));
		writer.PutVarUInt(uint64_t(S.maps.size()));
		writer.PutVarUInt(kShardedAnnCacheVersion);
		writer.PutVarUInt(uint64_t(S.maxElements));
		writer.PutVarUInt(uint64_t(S.shardRows));
		for (const auto& m : S.maps) m->graph_.SaveIndex(writer, cancel);
		return;
	}
	graph_.SaveIndex(writer, cancel);
}

bool GpuHnswMap::loadQuantizingParams(AnnCacheReader& reader, Sq8Params& stored, Sq8QuantizationConfig& cfg) {
	if (reader.GetVarUInt() == 0) return false;   // deserializeQuantizingParams (hnsw.cc:64-72)
	if (reader.GetVarUInt() != kAnnCacheQuantizationParamsVersion) throw std::runtime_error("Invalid quantization parameters version during deserialization");
	if (reader.GetVarInt() != 0) throw std::runtime_error("Unsupported quantization type");
	const float q = reader.GetFloat();   // QuantizationConfig::Deserialize (quantization_config.cc:30-41)
	if (q >= 0.95f && q <= 1.f) {
		cfg.quantile = q;
	} else if (q != 0.f) {
		throw std::runtime_error("Incorrect deserialized quantile value: must be within [0.95; 1.0]");
	}
	cfg.sampleSize = size_t(reader.GetVarUInt());
	cfg.quantizationThreshold = size_t(reader.GetVarUInt());
	stored.minQ = reader.GetFloat();
	stored.maxQ = reader.GetFloat();
	stored.alpha = reader.GetFloat();
	stored.alpha_2 = reader.GetFloat();
	stored.delta = reader.GetFloat();
	return true;
}

void GpuHnswMap::LoadIndex(AnnCacheReader& reader) {
	Sq8Params stored;
	Sq8QuantizationConfig cfg;
	const bool hasParams = loadQuantizingParams(reader, stored, cfg);
	LoadGraph(reader);
	quantized_ = false;
	pendingSq8_.reset();
	if (sh_) {
		for (const auto& m : sh_->maps) {
			m->quantized_ = false;
			m->pendingSq8_.reset();
		}
	}
	if (hasParams && reader.WithQuantizer()) {   // hnsw.cc:47-53: Load<QuantizedHnswT> only then; else the float graph
		sq8_ = stored;
		sq8Config_ = cfg;
		quantized_ = true;
		codesDirty_ = true;
		if (sh_) {   // ONE quantiser, a code table per shard (Quantize above)
			for (const auto& m : sh_->maps) m->Quantize(stored.minQ, stored.maxQ);
		}
	}
}

void GpuHnswMap::shLoadGraphs(AnnCacheReader& reader) {
	ShardedState& S = *sh_;
	if (shCount(false) != 0) throw std::logic_error("HnswGraph::LoadIndex: the graph is not empty");
	if (reader.GetVarUInt() != 0) {
		throw std::runtime_error("GpuHnswMap: the ANN cache holds one graph, this index is defined over a list of " + std::to_string(S.maps.size()) +
								 " devices (a graph per shard)");
	}
	const uint64_t shards = reader.GetVarUInt();
	if (shards != S.maps.size()) {
		throw std::runtime_error("GpuHnswMap: the ANN cache was written over " + std::to_string(shards) + " shards, this index is defined over " +
								 std::to_string(S.maps.size()));
	}
	if (reader.GetVarUInt() != kShardedAnnCacheVersion) throw std::runtime_error("GpuHnswMap: unknown version of the sharded ANN cache");
	const uint64_t maxElements = reader.GetVarUInt(), shardRows = reader.GetVarUInt();
	if (shardRows >= 0xFFFFFFFFull || maxElements > shardRows * S.maps.size()) throw std::runtime_error("GpuHnswMap: sharded ANN cache: sizes out of range");
	if (shardRows > S.shardRows) {   // the writer's shards were larger: every range grows first (a global row is shard * shardRows + local)
		if ((shardRows & 31) != 0) throw std::runtime_error("GpuHnswMap: sharded ANN cache: sizes out of range");
		for (auto& m : S.maps) m->ResizeIndex(size_t(shardRows));
		shCreateParent(size_t(shardRows));
	}
	S.maxElements = std::max(S.maxElements, size_t(maxElements));
	for (auto& m : S.maps) {
		m->LoadGraph(reader);   // (a failure leaves the earlier shards loaded: the caller clears the Map, HnswIndexBase::LoadIndexCache's clearMap())
		if (m->graph_.MaxElements() > S.shardRows) throw std::runtime_error("GpuHnswMap: sharded ANN cache: a shard's graph is larger than the shard");
	}
	std::lock_guard<std::mutex> lk(S.routeMtx);
	S.shardOf.clear();
	for (size_t s = 0; s < S.maps.size(); ++s) {
		const HnswGraph& g = S.maps[s]->graph_;
		S.routed[s] = g.Count();   // slots in use, delete-marked ones included (they are recycled only when every range is full)
		for (size_t i = 0; i < g.Count(); ++i) {
			if (!g.IsDeleted(tableint(i))) S.shardOf[g.Label(tableint(i))] = uint32_t(s);
		}
	}
}

void GpuHnswMap::Clear() {
	if (sh_) {
		for (auto& m : sh_->maps) m->Clear();
		{
			std::lock_guard<std::mutex> lk(sh_->routeMtx);
			sh_->shardOf.clear();
			sh_->routed.assign(sh_->maps.size(), 0);
		}
		// the shards' DEVICE indexes still hold the old rows and graphs, and shSyncAll() mirrors only shards that hold points: a fresh
		// parent handle rebinds every shard to an empty device index (a search over a shard left empty then finds nothing there)
		shCreateParent(sh_->shardRows);
		return;
	}
	graph_.Clear();
	graphDirty_ = true;
	deletedDirty_ = true;
}

void GpuHnswMap::LoadGraph(AnnCacheReader& reader) {
	if (sh_) return shLoadGraphs(reader);
	graph_.LoadIndex(reader);
	graphDirty_ = true;
	deletedDirty_ = true;
}

void GpuHnswMap::syncDevice() const {
	std::lock_guard<std::mutex> lk(syncMtx_);
	if (!graphDirty_ && !deletedDirty_) return;
	const size_t n = graph_.Count();
	if (graphDirty_) {
		if (rxgpu_index_capacity(dev_) < graph_.MaxElements()) {
			if (rxgpu_index_reserve(dev_, graph_.MaxElements()) != RXGPU_OK) throwDevice("Not enough memory: resizeIndex failed to allocate base layer");
		}
		// What changed since the last sync: a handful of nodes after an upsert (the new element, the neighbours it was linked to; an
		// in-place update also its one-hop neighbourhood) — their vectors and lists are patched in place; a bulk build re-sends everything.
		std::vector<tableint> dirty;
		const bool incremental = graph_.TakeDirty(dirty) && graphOnDevice_;
		codesIncremental_ = incremental;
		if (incremental) codesDirtyRows_ = dirty;
		const float* norms = graph_.InvNorms();
		bool patched = false;
		if (incremental) {
			// vectors: new rows as one run, updated rows (recycled slots / in-place updates) one by one
			for (const tableint id : dirty) {
				if (id >= syncedRows_) break;   // ids ascend; the rest are new rows
				if (rxgpu_index_upload_rows(dev_, id, 1, graph_.Vector(id), norms ? norms + id : nullptr) != RXGPU_OK) throwDevice("row upload failed");
			}
			if (n > syncedRows_) {
				if (rxgpu_index_upload_rows(dev_, syncedRows_, n - syncedRows_, graph_.Vectors() + syncedRows_ * graph_.Dim(),
											norms ? norms + syncedRows_ : nullptr) != RXGPU_OK) {
					throwDevice("row upload failed");
				}
			}
			const size_t stride0 = 1 + graph_.MaxM0();
			std::vector<uint32_t> rows0(dirty.size() * stride0), upperRows;
			std::vector<uint8_t> del(dirty.size());
			std::vector<int32_t> levels(dirty.size());
			for (size_t j = 0; j < dirty.size(); ++j) {
				const tableint id = dirty[j];
				std::memcpy(rows0.data() + j * stride0, graph_.Links0() + size_t(id) * stride0, stride0 * sizeof(uint32_t));
				del[j] = graph_.Deleted()[id];
				levels[j] = graph_.Levels()[id];
				graph_.AppendUpper(id, upperRows);
			}
			const int rc = rxgpu_hnsw_patch_graph(dev_, uint32_t(dirty.size()), dirty.data(), rows0.data(), del.data(), levels.data(),
												  upperRows.empty() ? nullptr : upperRows.data(), graph_.MaxLevel(), n ? graph_.EntryPoint() : 0,
												  graph_.DeletedCount());
			if (rc == RXGPU_OK) {
				patched = true;
			} else if (rc != RXGPU_ERR_OVERFLOW) {
				throwDevice("graph patch failed");
			}
			syncedRows_ = n;
		}
		if (!patched) {
			if (!incremental) {   // the vectors of recycled slots may have changed too: the tracker lost them, re-send all rows
				syncedRows_ = 0;
			}
			if (n > syncedRows_) {
				if (rxgpu_index_upload_rows(dev_, syncedRows_, n - syncedRows_, graph_.Vectors() + syncedRows_ * graph_.Dim(),
											norms ? norms + syncedRows_ : nullptr) != RXGPU_OK) {
					throwDevice("row upload failed");
				}
				syncedRows_ = n;
			}
			std::vector<uint64_t> off;
			std::vector<uint32_t> upper;
			graph_.ExportUpper(off, upper);
			if (rxgpu_hnsw_attach_graph(dev_, graph_.Links0(), off.data(), upper.data(), off.empty() ? 0 : off[n], graph_.Deleted(), uint32_t(graph_.M()),
										uint32_t(graph_.MaxM0()), graph_.MaxLevel(), n ? graph_.EntryPoint() : 0, graph_.DeletedCount()) != RXGPU_OK) {
				throwDevice("graph upload failed");
			}
			graphOnDevice_ = true;
		}
	} else if (deletedDirty_) {
		if (rxgpu_hnsw_update_deleted(dev_, graph_.Deleted(), graph_.DeletedCount()) != RXGPU_OK) throwDevice("delete-mark upload failed");
	}
	if (quantized_ && (graphDirty_ || codesDirty_)) {
		// points added to (or updated in) a quantised graph are quantised with the same parameters (addPoint, hnswalg.h:1480-1495): only their
		// codes travel when the change tracker knows what changed; a bulk change (or a fresh quantisation) sends the whole table
		if (!codesDirty_ && codesIncremental_ && syncedCodes_ <= n) {
			patchCodes(codesDirtyRows_, n);
		} else {
			attachCodes();
		}
	}
	codesDirtyRows_.clear();
	codesIncremental_ = false;
	graphDirty_ = false;
	deletedDirty_ = false;
}

// hnswalg.h:443-470 (the quantising copy) and :1480-1495 (points added to a quantised graph): codes + corrective offsets of every stored
// vector under the Map's parameters.  The whole code table is re-sent after a mutation — D bytes a row, a quarter of the float rows.
void GpuHnswMap::attachCodes() const {
	const size_t n = graph_.Count(), dim = graph_.Dim();
	std::vector<uint8_t> codes(n * dim);
	std::vector<float> corr(n);
	for (size_t i = 0; i < n; ++i) corr[i] = Sq8Quantize(graph_.Metric(), sq8_, graph_.Vector(tableint(i)), dim, 1.f, codes.data() + i * dim);
	if (rxgpu_hnsw_attach_sq8(dev_, codes.data(), corr.data(), n, sq8_.alpha_2) != RXGPU_OK) throwDevice("SQ8 code upload failed");
	codesDirty_ = false;
	syncedCodes_ = n;
}

void GpuHnswMap::patchCodes(const std::vector<tableint>& dirty, size_t n) const {
	const size_t dim = graph_.Dim();
	std::vector<uint8_t> code(dim);
	for (const tableint id : dirty) {   // updated rows (recycled slots, in-place updates; neighbours whose links changed cost a row each, a handful)
		if (id >= syncedCodes_) break;   // ids ascend; the rest are new rows
		const float corr = Sq8Quantize(graph_.Metric(), sq8_, graph_.Vector(id), dim, 1.f, code.data());
		if (rxgpu_hnsw_upload_sq8_rows(dev_, id, 1, code.data(), &corr, sq8_.alpha_2) != RXGPU_OK) throwDevice("SQ8 code upload failed");
	}
	if (n > syncedCodes_) {
		const size_t add = n - syncedCodes_;
		std::vector<uint8_t> codes(add * dim);
		std::vector<float> corr(add);
		for (size_t i = 0; i < add; ++i) corr[i] = Sq8Quantize(graph_.Metric(), sq8_, graph_.Vector(tableint(syncedCodes_ + i)), dim, 1.f, codes.data() + i * dim);
		if (rxgpu_hnsw_upload_sq8_rows(dev_, syncedCodes_, add, codes.data(), corr.data(), sq8_.alpha_2) != RXGPU_OK) throwDevice("SQ8 code upload failed");
	}
	syncedCodes_ = n;
}

void GpuHnswMap::Quantize(float minQ, float maxQ) {
	if (!(maxQ > minQ)) throw std::runtime_error("Quantize: empty quantisation range");
	if (sh_) {   // ONE quantiser for the whole Map (the reference quantises the index, not its parts): every shard codes its rows with it
		for (const auto& m : sh_->maps) m->Quantize(minQ, maxQ);
	}
	sq8_ = Sq8Params::FromRange(minQ, maxQ, graph_.Dim());
	pendingSq8_.reset();
	quantized_ = true;
	codesDirty_ = true;
	graphDirty_ = true;
}

void GpuHnswMap::Quantize(const Sq8QuantizationConfig& config) {
	if (quantized_ || pendingSq8_) throw std::logic_error("Quantize: the Map is quantised already");
	Sq8Params p;
	if (sh_) {   // the sample runs over the points of all shards, numbered shard after shard (what internal ids are to a single graph)
		std::vector<size_t> first(sh_->maps.size() + 1, 0);
		for (size_t i = 0; i < sh_->maps.size(); ++i) first[i + 1] = first[i] + sh_->maps[i]->graph_.Count();
		p = Sq8SampleParams(first.back(), graph_.Dim(), config, [this, &first](uint32_t id) {
			const size_t sh = size_t(std::upper_bound(first.begin(), first.end(), size_t(id)) - first.begin()) - 1;
			return sh_->maps[sh]->graph_.Vector(tableint(id - first[sh]));
		});
	} else {
		p = Sq8SampleParams(graph_.Count(), graph_.Dim(), config, [this](uint32_t id) { return graph_.Vector(tableint(id)); });
	}
	if (!(p.maxQ > p.minQ)) throw std::runtime_error("Quantize: empty quantisation range");
	sq8Config_ = config;
	pendingSq8_ = p;
}

void GpuHnswMap::SwitchMapOnQuantized() {
	if (!pendingSq8_) return;   // Impl::get(): nothing pending, nothing to swap in
	sq8_ = *pendingSq8_;
	pendingSq8_.reset();
	if (sh_) {
		for (const auto& m : sh_->maps) m->Quantize(sq8_.minQ, sq8_.maxQ);
	}
	quantized_ = true;
	codesDirty_ = true;
	graphDirty_ = true;
}

struct GpuHnswMap::PendingQuery {
	const float* query;
	uint32_t k, ef;
	float* dist;
	uint32_t* row;
	uint32_t* count;
	bool done;
	int rc;
	std::string error;
};

void GpuHnswMap::fetchKnn(const float* query, uint32_t k, uint32_t ef, float* dist, uint32_t* row, uint32_t* count) const {
	PendingQuery p{query, k, ef, dist, row, count, false, 0, {}};
	// over a device list the queries of T planner threads meet here too: one fan-out + one all-gather per BATCH instead of one per query
	rxgpu_index* const target = sh_ ? sh_->parent : dev_;
	if (!sh_) {   // the index's resident search kernel first (rxgpu_hnsw_search_knn_posted): a store and a poll, side by side with every other thread's
		int32_t served = 0;
		if (rxgpu_hnsw_search_knn_posted(target, query, k, ef, dist, row, count, &served) != RXGPU_OK) throw std::runtime_error(std::string("SearchKnn: ") + rxgpu_last_error());
		if (served) {
			coPosted_.fetch_add(1, std::memory_order_relaxed);
			return;
		}
	}
	if (!coalesce_) {
		p.rc = rxgpu_hnsw_search_knn(target, query, 1, k, ef, dist, row, count);
		if (p.rc != RXGPU_OK) p.error = rxgpu_last_error();
	} else {
		std::unique_lock<std::mutex> lk(coMtx_);
		coQueue_.push_back(&p);
		while (!p.done) {
			// How many batches at once?  Measured at 1M x 768 (tools/bench_hnsw_nq_sweep.py, profiles/rd5_hnsw_nq_sweep*.json): one call costs
			// 0.60 ms for 1 query, 0.75 for 8, 1.25 for 128, 2.3 for 1024 — a search is a chain of ~140 dependent hops on one wavefront, so a
			// batch is nearly free until the chip fills — but calls of different threads overlap only in part on the device.  T = 16 / 64 / 256
			// planner threads: one lane 9 / 28 / 66-70 k q/s, FOUR 12 / 34-37 / 57-75 k, eight 8-9 / 15-23 / 37-55 k, sixteen 6 / 6 / 9 k (the
			// reference's 16 cores on the same graph: 39 k).  A rule that made a long queue wait for an empty device so as to leave as one
			// batch lost the overlap and measured worse than any fixed number.
			if (coLeaders_ >= coLanes_ || coQueue_.empty()) {
				coCv_.wait(lk);
				continue;
			}
			++coLeaders_;   // a free lane: serve the queue head and everything queued with the same (k, ef) — this thread's own query possibly later
			std::vector<PendingQuery*> batch;
			const uint32_t bk = coQueue_.front()->k, bef = coQueue_.front()->ef;
			for (auto it = coQueue_.begin(); it != coQueue_.end() && batch.size() < 4096;) {
				if ((*it)->k == bk && (*it)->ef == bef) {
					batch.push_back(*it);
					it = coQueue_.erase(it);
				} else {
					++it;
				}
			}
			lk.unlock();
			const size_t nq = batch.size(), dim = graph_.Dim();
			int rc = RXGPU_OK;
			std::string error;
			try {
				if (nq == 1) {
					PendingQuery& q = *batch[0];
					rc = rxgpu_hnsw_search_knn(target, q.query, 1, bk, bef, q.dist, q.row, q.count);
				} else {
					std::vector<float> queries(nq * dim), d(nq * bk);
					std::vector<uint32_t> r(nq * bk), c(nq);
					for (size_t i = 0; i < nq; ++i) std::copy(batch[i]->query, batch[i]->query + dim, queries.begin() + i * dim);
					rc = rxgpu_hnsw_search_knn(target, queries.data(), uint32_t(nq), bk, bef, d.data(), r.data(), c.data());
					if (rc == RXGPU_OK) {
						for (size_t i = 0; i < nq; ++i) {
							std::copy(d.begin() + i * bk, d.begin() + i * bk + c[i], batch[i]->dist);
							std::copy(r.begin() + i * bk, r.begin() + i * bk + c[i], batch[i]->row);
							*batch[i]->count = c[i];
						}
					}
				}
				if (rc != RXGPU_OK) error = rxgpu_last_error();
			} catch (const std::exception& e) {   // e.g. bad_alloc while staging: every caller of the batch gets the error, the lane is released
				rc = RXGPU_ERR_NOMEM;
				error = e.what();
			}
			lk.lock();
			for (PendingQuery* q : batch) {
				q->rc = rc;
				q->error = error;
				q->done = true;
			}
			++coBatches_;
			--coLeaders_;
			coCv_.notify_all();
		}
	}
	if (p.rc != RXGPU_OK) throw std::runtime_error("SearchKnn: " + p.error);
}

// queryNormCoef (hnswalg.h:1855-1863) and prepareData (:510-529): the query is scaled back to its original length, quantised, and every
// distance is multiplied by 1 / |q|.  Returns the query's corrective offset.
float GpuHnswMap::quantizeQuery(const float* queryDataRaw, std::optional<float> queryDataNorm, std::vector<uint8_t>& qcodes, float& normCoef) const {
	const bool cosine = graph_.Metric() == VectorMetric::Cosine;
	if (cosine && !queryDataNorm) {
		throw std::runtime_error("Norm is required for Cosine-metric during corrective offsets calculation in quantized graph");
	}
	normCoef = cosine ? 1.f / *queryDataNorm : 1.f;
	qcodes.resize(graph_.Dim());
	return Sq8Quantize(graph_.Metric(), sq8_, queryDataRaw, graph_.Dim(), 1.f / normCoef, qcodes.data());
}

// hnswalg.h:1988-2012
SearchResultQueue GpuHnswMap::SearchKnn(const float* queryDataRaw, std::optional<float> queryDataNorm, size_t k, size_t ef) const {
	SearchResultQueue result;
	if (sh_) {   // every shard's SearchKnn at once; the lists meet on the devices (rxgpu_hnsw_search_knn on the sharded handle)
		const size_t total = shCount(false);
		if (total == 0 || k == 0) return result;
		shSyncAll();
		k = std::min(k, total);
		std::vector<float> dist(k);
		std::vector<uint32_t> row(k);
		uint32_t count = 0;
		if (quantized_) {   // every shard over its code table, the query quantised once (one quantiser for the Map)
			float normCoef = 1.f;
			std::vector<uint8_t> qcodes;
			const float qcorr = quantizeQuery(queryDataRaw, queryDataNorm, qcodes, normCoef);
			if (rxgpu_hnsw_search_knn_sq8(sh_->parent, qcodes.data(), &qcorr, &normCoef, 1, uint32_t(k), uint32_t(ef), dist.data(), row.data(), &count) != RXGPU_OK) {
				throwDevice("SearchKnn");
			}
		} else {
			fetchKnn(queryDataRaw, uint32_t(k), uint32_t(ef), dist.data(), row.data(), &count);   // the coalescer in front of the sharded handle
		}
		ReserveQueue(result, count);
		for (uint32_t i = 0; i < count; ++i) result.emplace(dist[i], shLabel(row[i]));
		return result;
	}
	const size_t n = graph_.Count();
	if (n == 0 || k == 0) return result;
	syncDevice();
	k = std::min(k, n);
	std::vector<float> dist(k);
	std::vector<uint32_t> row(k);
	uint32_t count = 0;
	if (quantized_) {
		float normCoef = 1.f;
		std::vector<uint8_t> qcodes;
		const float qcorr = quantizeQuery(queryDataRaw, queryDataNorm, qcodes, normCoef);
		if (rxgpu_hnsw_search_knn_sq8(dev_, qcodes.data(), &qcorr, &normCoef, 1, uint32_t(k), uint32_t(ef), dist.data(), row.data(), &count) !=
			RXGPU_OK) {
			throwDevice("SearchKnn");
		}
	} else {
		fetchKnn(queryDataRaw, uint32_t(k), uint32_t(ef), dist.data(), row.data(), &count);
	}
	ReserveQueue(result, count);
	for (uint32_t i = 0; i < count; ++i) result.emplace(dist[i], graph_.Label(row[i]));
	return result;
}

uint64_t GpuHnswMap::TieReruns() const {
	uint64_t n = 0;
	if (rxgpu_hnsw_read_tie_reruns(DeviceIndex(), &n) != RXGPU_OK) throwDevice("TieReruns");
	return n;
}

uint64_t GpuHnswMap::LdsReruns() const {
	uint64_t n = 0;
	if (rxgpu_hnsw_read_lds_reruns(DeviceIndex(), &n) != RXGPU_OK) throwDevice("LdsReruns");
	return n;
}

// A session over a device list: one session per shard (each walks ITS graph exactly as a single-device Map's does) and, per shard, what that
// session has delivered and the merged stream has not emitted yet.
struct StreamingSearchSession::Sharded {
	struct Part {
		StreamingSearchSession session;
		std::vector<std::pair<float, labeltype>> held;   // ascending (dist, label)
		bool exhausted = false;
	};
	std::vector<Part> parts;
};

StreamingSearchSession::StreamingSearchSession() = default;
StreamingSearchSession::StreamingSearchSession(StreamingSearchSession&& o) noexcept : impl_(o.impl_), graph_(o.graph_), sharded_(std::move(o.sharded_)) { o.impl_ = nullptr; }
StreamingSearchSession::~StreamingSearchSession() {
	if (impl_) rxgpu_hnsw_stream_end(impl_);
}
StreamingSearchSession& StreamingSearchSession::operator=(StreamingSearchSession&& o) noexcept {
	if (this != &o) {
		if (impl_) rxgpu_hnsw_stream_end(impl_);
		impl_ = o.impl_;
		graph_ = o.graph_;
		sharded_ = std::move(o.sharded_);
		o.impl_ = nullptr;
	}
	return *this;
}

// hnswalg.h:1865-1891
StreamingSearchSession GpuHnswMap::BeginStreamingSearch(const float* queryDataRaw, std::optional<float> queryDataNorm, StreamingSearchOptions opts) const {
	if (sh_) {   // a session per shard; ContinueStreamingSearch merges what they deliver
		StreamingSearchSession session;
		session.graph_ = this;
		session.sharded_ = std::make_unique<StreamingSearchSession::Sharded>();
		for (const auto& m : sh_->maps) {
			StreamingSearchSession::Sharded::Part part;
			if (m->graph_.Count()) {
				part.session = m->BeginStreamingSearch(queryDataRaw, queryDataNorm, opts);
			} else {
				part.exhausted = true;
			}
			session.sharded_->parts.push_back(std::move(part));
		}
		return session;
	}
	StreamingSearchSession session;
	session.graph_ = this;
	syncDevice();
	if (quantized_) {   // HierarchicalNSWImpl<uint8_t>: the session runs over the codes, the query prepared as in SearchKnn (hnswalg.h:1872-1878)
		float normCoef = 1.f;
		std::vector<uint8_t> qcodes;
		const float qcorr = quantizeQuery(queryDataRaw, queryDataNorm, qcodes, normCoef);
		if (rxgpu_hnsw_stream_begin_sq8(dev_, qcodes.data(), qcorr, normCoef, uint32_t(opts.ef), &session.impl_) != RXGPU_OK) {
			throwDevice("BeginStreamingSearch");
		}
		return session;
	}
	if (rxgpu_hnsw_stream_begin(dev_, queryDataRaw, uint32_t(opts.ef), &session.impl_) != RXGPU_OK) throwDevice("BeginStreamingSearch");
	return session;
}

// hnswalg.h:1947-1975
StreamingBatch GpuHnswMap::ContinueStreamingSearch(StreamingSearchSession& session, size_t batchSize) const {
	StreamingBatch batch;
	if (sh_) {
		// The next batchSize results of the UNION of the shards' streams: every shard's session is asked for as many results as the batch
		// could take from it (batchSize minus what it delivered earlier and is still held), then the batchSize nearest of everything held
		// leave, (dist, label) ascending — each shard's own stream is the reference's (hnswalg.h:1947-1975) over that shard's graph.
		if (session.graph_ != this || !session.sharded_) {
			batch.exhausted = true;
			return batch;
		}
		if (batchSize == 0) return batch;
		auto& parts = session.sharded_->parts;
		for (size_t i = 0; i < parts.size(); ++i) {
			auto& part = parts[i];
			if (part.exhausted || part.held.size() >= batchSize) continue;
			StreamingBatch got = sh_->maps[i]->ContinueStreamingSearch(part.session, batchSize - part.held.size());
			part.exhausted = got.exhausted;
			for (; !got.results.empty(); got.results.pop()) part.held.emplace_back(got.results.top().first, got.results.top().second);
			std::sort(part.held.begin(), part.held.end());
		}
		std::vector<size_t> at(parts.size(), 0);
		ReserveQueue(batch.results, batchSize);
		for (size_t n = 0; n < batchSize; ++n) {
			int best = -1;
			for (size_t i = 0; i < parts.size(); ++i) {
				if (at[i] < parts[i].held.size() && (best < 0 || parts[i].held[at[i]] < parts[size_t(best)].held[at[size_t(best)]])) best = int(i);
			}
			if (best < 0) break;
			const auto& e = parts[size_t(best)].held[at[size_t(best)]++];
			batch.results.emplace(e.first, e.second);
		}
		bool all = true;
		for (size_t i = 0; i < parts.size(); ++i) {
			parts[i].held.erase(parts[i].held.begin(), parts[i].held.begin() + long(at[i]));
			all = all && parts[i].exhausted && parts[i].held.empty();
		}
		batch.exhausted = all;
		return batch;
	}
	if (session.graph_ != this || !session.impl_) {
		batch.exhausted = true;
		return batch;
	}
	if (batchSize == 0) return batch;
	const size_t cap = std::min(batchSize, std::max<size_t>(1, graph_.Count()));
	std::vector<float> dist(cap);
	std::vector<uint32_t> row(cap);
	uint32_t count = 0;
	int32_t exhausted = 0;
	if (rxgpu_hnsw_stream_continue(session.impl_, uint32_t(std::min<size_t>(batchSize, 0xFFFFFFFFu)), dist.data(), row.data(), &count, &exhausted) != RXGPU_OK) {
		throwDevice("ContinueStreamingSearch");
	}
	ReserveQueue(batch.results, count);
	for (uint32_t i = 0; i < count; ++i) batch.results.emplace(dist[i], graph_.Label(row[i]));   // emitStreamingBatch: (dist, ExternalLabel)
	batch.exhausted = exhausted != 0;
	return batch;
}

// hnswalg.h:2015-2070: ef-search, then the closure over level-0 links while dist < radius — both on the device (rxgpu_hnsw_search_range:
// the expansion is one launch of hnsw_range_kernel whatever its depth; the result is a set, its order does not depend on the walk).
SearchResultQueue GpuHnswMap::SearchRange(const float* queryDataRaw, std::optional<float> queryDataNorm, float radius, size_t ef) const {
	SearchResultQueue result;
	if (sh_ && quantized_) {   // the shards' own (quantised) Maps one after the other: the result is a set, the union of theirs
		for (const auto& m : sh_->maps) {
			if (!m->graph_.Count()) continue;
			SearchResultQueue part = m->SearchRange(queryDataRaw, queryDataNorm, radius, ef);
			for (; !part.empty(); part.pop()) result.emplace(part.top().first, part.top().second);
		}
		return result;
	}
	if (sh_) {   // every shard's ef-search + closure (rxgpu_hnsw_search_range on the sharded handle), hits concatenated
		const size_t total = shCount(false);
		if (total == 0) return result;
		shSyncAll();
		const size_t efEff = ef ? ef : 1;
		std::vector<float> dist;
		std::vector<uint32_t> row;
		uint64_t hits = 0;
		for (size_t cap = std::max<size_t>(4 * efEff * sh_->maps.size(), 1024);; cap = std::min<size_t>(total, std::max<size_t>(2 * cap, size_t(hits)))) {
			cap = std::min(cap, total);
			dist.resize(cap);
			row.resize(cap);
			const int rc = rxgpu_hnsw_search_range(sh_->parent, queryDataRaw, radius, uint32_t(efEff), dist.data(), row.data(), cap, &hits);
			if (rc == RXGPU_OK) break;
			if (rc != RXGPU_ERR_OVERFLOW || cap >= total) throwDevice("SearchRange");
		}
		ReserveQueue(result, size_t(hits));
		for (uint64_t i = 0; i < hits; ++i) result.emplace(dist[i], shLabel(row[i]));
		return result;
	}
	const size_t n = graph_.Count();
	if (n == 0) return result;
	syncDevice();
	const size_t efEff = ef ? ef : 1;   // no clamp: an ef beyond the device engine's limit (4096) is an error from the C-ABI, never a silently smaller search
	float normCoef = 1.f, qcorr = 0.f;
	std::vector<uint8_t> qcodes;
	if (quantized_) qcorr = quantizeQuery(queryDataRaw, queryDataNorm, qcodes, normCoef);
	std::vector<float> dist;
	std::vector<uint32_t> row;
	uint64_t total = 0;
	for (size_t cap = std::max<size_t>(4 * efEff, 1024);; cap = std::min<size_t>(n, std::max<size_t>(2 * cap, size_t(total)))) {
		cap = std::min(cap, n);
		dist.resize(cap);
		row.resize(cap);
		const int rc = quantized_ ? rxgpu_hnsw_search_range_sq8(dev_, qcodes.data(), qcorr, normCoef, radius, uint32_t(efEff), dist.data(), row.data(), cap, &total)
								  : rxgpu_hnsw_search_range(dev_, queryDataRaw, radius, uint32_t(efEff), dist.data(), row.data(), cap, &total);
		if (rc == RXGPU_OK) break;
		if (rc != RXGPU_ERR_OVERFLOW || cap >= n) throwDevice("SearchRange");
	}
	ReserveQueue(result, size_t(total));
	for (uint64_t i = 0; i < total; ++i) result.emplace(dist[i], graph_.Label(row[i]));
	return result;
}

}  // namespace rxgpu::host
