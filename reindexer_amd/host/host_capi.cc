// Flat C entry points over the C++ host layer so that pytest (ctypes) can drive GpuBruteforceMap / KnnSelect the way
// the reference's own engine-level tests drive BruteforceSearch (gtests/tests/unit/hnsw_streaming_search_test.cc).
// Exceptions become return codes + thread-local text, like the Reindexer API boundary turns them into Error values.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <stdexcept>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>
#include <cstring>
#include <string>

#include "device_list.h"
#include "gpu_bruteforce_map.h"
#include "knn_select.h"

using namespace rxgpu::host;

namespace {
thread_local std::string g_err;
template <typename F>
int guarded(F&& f) {
	try {
		f();
		return 0;
	} catch (const std::logic_error& e) {
		g_err = e.what();
		return -4;
	} catch (const std::exception& e) {
		g_err = e.what();
		return -1;
	}
}
size_t drain(SearchResultQueue& q, float* outDist, uint64_t* outLabel, size_t cap) {
	const size_t n = q.size();
	for (size_t i = n; !q.empty(); q.pop()) {
		--i;
		if (i < cap) {
			outDist[i] = q.top().first;
			outLabel[i] = q.top().second;
		}
	}
	return n;
}
}  // namespace

extern "C" {

const char* rxhost_last_error() { return g_err.c_str(); }

float rxhost_l2_module(const float* x, int32_t d) { return CalculateL2Module(x, d); }
float rxhost_normalize_copy(const float* x, int32_t d, float* out) { return NormalizeCopyVector(x, d, out); }
// AddNorm over a block of rows (hnswlib.h:80-92): the 1/|row| table a cosine index stores, the product's own host arithmetic
void rxhost_l2_modules_many(const float* rows, size_t n, int32_t d, float* out, unsigned threads) {
	threads = std::max(1u, std::min<unsigned>(threads ? threads : 1u, unsigned(std::max<size_t>(n / 1024, 1))));
	auto work = [&](size_t a, size_t b) {
		for (size_t i = a; i < b; ++i) out[i] = CalculateL2Module(rows + i * size_t(d), d);
	};
	std::vector<std::thread> pool;
	const size_t per = (n + threads - 1) / threads;
	for (unsigned t = 1; t < threads; ++t) {
		if (t * per < n) pool.emplace_back(work, t * per, std::min(n, (t + 1) * per));
	}
	work(0, std::min(n, per));
	for (auto& th : pool) th.join();
}

void* rxhost_bf_create(int metric, size_t dim, size_t maxElements, int device) {
	GpuBruteforceMap* m = nullptr;
	guarded([&] { m = new GpuBruteforceMap(VectorMetric(metric), dim, maxElements, device); });
	return m;
}
// the Map over a device list (row-range shards, BASELINE configs[3]); a device may be listed more than once
void* rxhost_bf_create_sharded(int metric, size_t dim, size_t maxElements, const int* devices, size_t nDevices) {
	GpuBruteforceMap* m = nullptr;
	guarded([&] { m = new GpuBruteforceMap(VectorMetric(metric), dim, maxElements, std::vector<int>(devices, devices + nDevices)); });
	return m;
}
// The shape the in-tree adapter constructs (rx_seam.h GpuBruteforceMapInTree: `Map(metric, dim, maxElements)`, hnsw_index.cc:61-66): the
// device list comes from RX_GPU_VECTOR_INDEXES (device_list.h); unset -> device 0.
void* rxhost_bf_create_from_env(int metric, size_t dim, size_t maxElements) {
	GpuBruteforceMap* m = nullptr;
	guarded([&] {
		std::vector<int> d = GpuDevicesFromEnv();
		if (d.empty()) d.push_back(0);
		m = new GpuBruteforceMap(VectorMetric(metric), dim, maxElements, std::move(d));
	});
	return m;
}
// device_list.h's parser: the list into out[0..cap), returns its length (0: no GPU engine)
size_t rxhost_parse_device_list(const char* text, int* out, size_t cap) {
	const std::vector<int> d = ParseDeviceList(text);
	for (size_t i = 0; i < d.size() && i < cap; ++i) out[i] = d[i];
	return d.size();
}
int rxhost_bf_is_sharded(void* h) { return static_cast<GpuBruteforceMap*>(h)->Sharded() ? 1 : 0; }
void* rxhost_bf_clone(void* h, size_t newMaxElements) {
	GpuBruteforceMap* m = nullptr;
	guarded([&] { m = new GpuBruteforceMap(*static_cast<GpuBruteforceMap*>(h), newMaxElements); });
	return m;
}
void rxhost_bf_destroy(void* h) { delete static_cast<GpuBruteforceMap*>(h); }
int rxhost_bf_add(void* h, const float* vec, size_t dim, uint64_t label) {
	return guarded([&] { static_cast<GpuBruteforceMap*>(h)->AddPointNoLock(ConstFloatVectorView(vec, dim), FloatVectorId::FromNumber(label)); });
}
int rxhost_bf_add_many(void* h, const float* vecs, size_t n, size_t dim, const uint64_t* labels) {
	return guarded([&] {
		auto* m = static_cast<GpuBruteforceMap*>(h);
		for (size_t i = 0; i < n; ++i) m->AddPointNoLock(ConstFloatVectorView(vecs + i * dim, dim), FloatVectorId::FromNumber(labels[i]));
	});
}
int rxhost_bf_add_concurrent(void* h, const float* vec, size_t dim, uint64_t label) {
	return guarded([&] { static_cast<GpuBruteforceMap*>(h)->AddPointConcurrent(ConstFloatVectorView(vec, dim), FloatVectorId::FromNumber(label)); });
}
int rxhost_bf_remove(void* h, uint64_t label) {
	return guarded([&] { static_cast<GpuBruteforceMap*>(h)->RemovePoint(label); });
}
int rxhost_bf_resize(void* h, size_t n) {
	return guarded([&] { static_cast<GpuBruteforceMap*>(h)->ResizeIndex(n); });
}
size_t rxhost_bf_count(void* h) { return static_cast<GpuBruteforceMap*>(h)->CurrentElementCount(); }
size_t rxhost_bf_max_elements(void* h) { return static_cast<GpuBruteforceMap*>(h)->MaxElements(); }
size_t rxhost_bf_element_size(void* h) { return static_cast<GpuBruteforceMap*>(h)->ElementSize(); }
size_t rxhost_bf_tie_replays(void* h) { return static_cast<GpuBruteforceMap*>(h)->TieReplays(); }
int rxhost_bf_vector_by_label(void* h, uint64_t label, float* out) {
	return guarded([&] {
		auto* m = static_cast<GpuBruteforceMap*>(h);
		std::memcpy(out, m->FloatPtrByExternalLabel(label), m->Dim() * sizeof(float));
	});
}
// returns hit count (best first in out*), or -1 on error
long rxhost_bf_search_knn(void* h, const float* q, size_t k, float* outDist, uint64_t* outLabel) {
	long n = -1;
	guarded([&] {
		auto res = static_cast<const GpuBruteforceMap*>(h)->SearchKnn(q, std::nullopt, k, 0);
		n = long(drain(res, outDist, outLabel, k));
	});
	return n;
}
long rxhost_bf_search_knn_filtered(void* h, const float* q, size_t k, const uint64_t* allowed, size_t nAllowed, float* outDist,
								   uint64_t* outLabel) {
	long n = -1;
	guarded([&] {
		auto res = static_cast<const GpuBruteforceMap*>(h)->SearchKnnFiltered(q, std::nullopt, k, allowed, nAllowed);
		n = long(drain(res, outDist, outLabel, k));
	});
	return n;
}
long rxhost_bf_search_range(void* h, const float* q, float radius, float* outDist, uint64_t* outLabel, size_t cap) {
	long n = -1;
	guarded([&] {
		auto res = static_cast<const GpuBruteforceMap*>(h)->SearchRange(q, std::nullopt, radius, 0);
		n = long(drain(res, outDist, outLabel, cap));
	});
	return n;
}
// HnswIndexBase::select through the Map: k < 0 => no k, has_radius == 0 => no radius.  Returns count or -1.
// KnnSelect over a GIVEN search result (what the Map's SearchKnn / SearchRange returned): the host post-processing alone, so that it can be
// checked on CPU against the reference's HnswIndexBase::select (tests/test_select_pin.py)
namespace {
struct FixedResultMap {
	VectorMetric metric;
	size_t dim;
	const float* dist;
	const uint64_t* label;
	size_t n;
	VectorMetric Metric() const noexcept { return metric; }
	size_t Dim() const noexcept { return dim; }
	SearchResultQueue fill() const {
		SearchResultQueue q;
		for (size_t i = 0; i < n; ++i) q.emplace(dist[i], label[i]);
		return q;
	}
	SearchResultQueue SearchKnn(const float*, std::optional<float>, size_t, size_t) const { return fill(); }
	SearchResultQueue SearchRange(const float*, std::optional<float>, float, size_t) const { return fill(); }
};
}  // namespace
long rxhost_select_postprocess(int metric, const float* dist, const uint64_t* label, size_t n, long k, int has_radius, int need_sort, int is_array,
							   int raw, int32_t* outIds, float* outRanks) {
	long cnt = -1;
	guarded([&] {
		const FixedResultMap map{VectorMetric(metric), 1, dist, label, n};
		KnnSearchParams p;
		if (k >= 0) p.k = size_t(k);
		if (has_radius) p.radius = 0.f;   // only its presence matters after the search (removeOverK)
		const float key = 1.f;
		auto res = raw ? KnnSelectRaw(map, ConstFloatVectorView(&key, 1), p, is_array != 0)
					   : KnnSelect(map, ConstFloatVectorView(&key, 1), p, need_sort != 0, is_array != 0);
		cnt = long(res.ids.size());
		for (size_t i = 0; i < res.ids.size(); ++i) {
			outIds[i] = res.ids[i];
			outRanks[i] = res.ranks[i];
		}
	});
	return cnt;
}

long rxhost_bf_select(void* h, const float* key, size_t dim, long k, int has_radius, float radius, int need_sort, int is_array,
					  int32_t* outIds, float* outRanks, size_t cap) {
	long n = -1;
	guarded([&] {
		KnnSearchParams p;
		if (k >= 0) p.k = size_t(k);
		if (has_radius) p.radius = radius;
		p.Validate(false);
		auto res = KnnSelect(*static_cast<const GpuBruteforceMap*>(h), ConstFloatVectorView(key, dim), p, need_sort != 0, is_array != 0);
		n = long(res.ids.size());
		for (size_t i = 0; i < res.ids.size() && i < cap; ++i) {
			outIds[i] = res.ids[i];
			outRanks[i] = res.ranks[i];
		}
	});
	return n;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------- HNSW graph builder
#include "hnsw_graph.h"

// ---- the admission cut of a phrase over document-range shards (csrc/ft_phrase_cut.h: what rxgpu_ft_capi.hip runs between the shards' admission
// passes), for the CPU suite: counts [shards][rows] (a shard whose first entry is 0xFFFFFFFF admitted nothing), keep [shards] out.
#include "../csrc/ft_phrase_cut.h"
extern "C" void rxhost_ft_shard_phrase_cut(const uint32_t* counts, size_t shards, size_t rows, uint64_t mergeLimit, uint64_t* keep) {
	std::vector<std::vector<uint32_t>> c(shards);
	for (size_t s = 0; s < shards; ++s) {
		if (rows && counts[s * rows] == 0xFFFFFFFFu) continue;
		c[s].assign(counts + s * rows, counts + (s + 1) * rows);
	}
	const std::vector<uint64_t> k = rxgpu::ft_shard_phrase_cut(c, rows, mergeLimit);
	for (size_t s = 0; s < shards; ++s) keep[s] = k[s];
}

// ---- the ANN disk cache through plain memory (tests, tools): fixed-width little-endian fields — 8 bytes per var-int, u64 length + bytes per
// string, 4 bytes per float, the 8-byte label for a primary key (the reader resolves it against the (labels, vectors) table it was given).
// oracle/ref/ref_shim.cc implements hnswlib::IWriter / IReader with the SAME encoding around the reference engine, so the two sides exchange
// caches byte for byte.
namespace {
class MemAnnWriter final : public rxgpu::host::AnnCacheWriter {
public:
	std::vector<uint8_t> buf;
	void put(const void* p, size_t n) { buf.insert(buf.end(), static_cast<const uint8_t*>(p), static_cast<const uint8_t*>(p) + n); }
	void PutVarUInt(uint64_t v) override { put(&v, 8); }
	void PutVarUInt(uint32_t v) override { PutVarUInt(uint64_t(v)); }
	void PutVarInt(int64_t v) override { put(&v, 8); }
	void PutVarInt(int32_t v) override { PutVarInt(int64_t(v)); }
	void PutVString(std::string_view v) override {
		const uint64_t n = v.size();
		put(&n, 8);
		put(v.data(), v.size());
	}
	void PutFloat(float v) override { put(&v, 4); }
	void AppendPKByID(labeltype l) override { put(&l, 8); }
};
class MemAnnReader final : public rxgpu::host::AnnCacheReader {
public:
	MemAnnReader(const uint8_t* d, size_t n, size_t dim, const uint64_t* labels, const float* vectors, size_t rows) : d_(d), n_(n), dim_(dim), vectors_(vectors) {
		for (size_t i = 0; i < rows; ++i) rowOf_[labels[i]] = i;
	}
	size_t Remaining() const noexcept { return n_ - at_; }
	uint64_t GetVarUInt() override { return get<uint64_t>(); }
	int64_t GetVarInt() override { return get<int64_t>(); }
	float GetFloat() override { return get<float>(); }
	std::string_view GetVString() override {
		const uint64_t n = get<uint64_t>();
		need(n);
		std::string_view v(reinterpret_cast<const char*>(d_ + at_), n);
		at_ += n;
		return v;
	}
	labeltype ReadPkEncodedData(float* dest) override {
		const uint64_t label = get<uint64_t>();
		const auto it = rowOf_.find(label);
		if (it == rowOf_.end()) throw std::runtime_error("ANN cache: no row with the stored key");
		std::memcpy(dest, vectors_ + it->second * dim_, dim_ * sizeof(float));
		return label;
	}
	bool WithQuantizer() const override { return withQuantizer; }
	bool withQuantizer = false;   // LoadWithQuantizer of HnswIndexBase::LoadIndexCache (hnsw_index.cc:452-484)

private:
	void need(size_t n) const {
		if (n_ - at_ < n) throw std::runtime_error("ANN cache: truncated stream");
	}
	template <typename T>
	T get() {
		need(sizeof(T));
		T v;
		std::memcpy(&v, d_ + at_, sizeof(T));
		at_ += sizeof(T);
		return v;
	}
	const uint8_t* d_;
	size_t n_, at_ = 0, dim_;
	const float* vectors_;
	std::unordered_map<uint64_t, size_t> rowOf_;
};
}  // namespace


extern "C" {

void* rxhost_graph_create(int metric, size_t dim, size_t maxElements, size_t M, size_t efConstruction) {
	HnswGraph* g = nullptr;
	guarded([&] { g = new HnswGraph(VectorMetric(metric), dim, maxElements, M, efConstruction); });
	return g;
}
void rxhost_graph_destroy(void* h) { delete static_cast<HnswGraph*>(h); }
int rxhost_graph_add_many(void* h, const float* vecs, size_t n, size_t dim, const uint64_t* labels) {
	return guarded([&] {
		auto* g = static_cast<HnswGraph*>(h);
		for (size_t i = 0; i < n; ++i) g->AddPoint(vecs + i * dim, labels[i]);
	});
}
// threads >= 2: HnswGraph::AddPoints (concurrent construction); threads == 1: every point through AddPointConcurrent from this one thread
// (the concurrent code path, deterministic — tests compare it with the sequential graph)
int rxhost_graph_add_many_mt(void* h, const float* vecs, size_t n, size_t dim, const uint64_t* labels, unsigned threads) {
	return guarded([&] {
		auto* g = static_cast<HnswGraph*>(h);
		if (threads >= 2) {
			g->AddPoints(vecs, labels, n, threads);
		} else {
			for (size_t i = 0; i < n; ++i) g->AddPointConcurrent(vecs + i * dim, labels[i]);
		}
	});
}
// borrowed views of the builder's own storage, in internal-id order (valid until the next insert that resizes / the graph's destruction)
const float* rxhost_graph_vectors(void* h) { return static_cast<HnswGraph*>(h)->Vectors(); }
const float* rxhost_graph_inv_norms(void* h) { return static_cast<HnswGraph*>(h)->InvNorms(); }
int rxhost_graph_mark_delete(void* h, uint64_t label) {
	return guarded([&] { static_cast<HnswGraph*>(h)->MarkDelete(label); });
}
// info[0]=count [1]=M [2]=maxM0 [3]=maxlevel [4]=entry [5]=numDeleted [6]=upper blocks
// HnswGraph::SaveIndex behind the "not quantised" flag of HierarchicalNSW::SaveIndex (hnsw.cc:41-44).  Returns the byte count (copied when it fits cap).
long rxhost_graph_save_index(void* h, uint8_t* out, size_t cap) {
	long n = -1;
	guarded([&] {
		MemAnnWriter w;
		w.PutVarUInt(uint32_t(0));
		const std::atomic_int32_t cancel{0};
		static_cast<const HnswGraph*>(h)->SaveIndex(w, cancel);
		if (w.buf.size() <= cap) std::memcpy(out, w.buf.data(), w.buf.size());
		n = long(w.buf.size());
	});
	return n;
}
// ... and LoadIndex into an EMPTY graph; (labels, vectors): the namespace's rows the primary keys of the cache resolve to
int rxhost_graph_load_index(void* h, const uint8_t* data, size_t len, const uint64_t* labels, const float* vectors, size_t rows) {
	return guarded([&] {
		auto* g = static_cast<HnswGraph*>(h);
		MemAnnReader r(data, len, g->Dim(), labels, vectors, rows);
		if (r.GetVarUInt() != 0) throw std::runtime_error("ANN cache: quantization parameters in the stream");
		g->LoadIndex(r);
		if (r.Remaining()) throw std::runtime_error("ANN cache: unparsed data behind the graph");
	});
}
void rxhost_graph_clear(void* h) { static_cast<HnswGraph*>(h)->Clear(); }

void rxhost_graph_info(void* h, int64_t* info) {
	auto* g = static_cast<HnswGraph*>(h);
	info[0] = int64_t(g->Count());
	info[1] = int64_t(g->M());
	info[2] = int64_t(g->MaxM0());
	info[3] = g->MaxLevel();
	info[4] = int64_t(g->EntryPoint());
	info[5] = int64_t(g->DeletedCount());
	int64_t blocks = 0;
	for (size_t i = 0; i < g->Count(); ++i) blocks += g->Levels()[i];
	info[6] = blocks;
}
void rxhost_graph_export(void* h, uint32_t* links0, int32_t* levels, uint64_t* labels, uint8_t* deleted, uint64_t* upperOff, uint32_t* upper) {
	auto* g = static_cast<HnswGraph*>(h);
	const size_t n = g->Count();
	std::memcpy(links0, g->Links0(), n * (1 + g->MaxM0()) * sizeof(uint32_t));
	std::memcpy(levels, g->Levels(), n * sizeof(int32_t));
	std::memcpy(labels, g->Labels(), n * sizeof(uint64_t));
	std::memcpy(deleted, g->Deleted(), n);
	std::vector<uint64_t> off;
	std::vector<uint32_t> blocks;
	g->ExportUpper(off, blocks);
	std::memcpy(upperOff, off.data(), off.size() * sizeof(uint64_t));
	if (off[n]) std::memcpy(upper, blocks.data(), off[n] * (1 + g->M()) * sizeof(uint32_t));
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------- GpuHnswMap
#include "gpu_hnsw_map.h"

extern "C" {

void* rxhost_hnsw_create(int metric, size_t dim, size_t maxElements, size_t M, size_t efConstruction, int device) {
	GpuHnswMap* m = nullptr;
	guarded([&] { m = new GpuHnswMap(VectorMetric(metric), dim, maxElements, M, efConstruction, device); });
	return m;
}
// HierarchicalNSWMT: the Map the reference instantiates for multithreaded index builds (hnsw_index.cc:566-573)
void* rxhost_hnsw_create_mt(int metric, size_t dim, size_t maxElements, size_t M, size_t efConstruction, int device) {
	GpuHnswMap* m = nullptr;
	guarded([&] { m = new GpuHnswMap(VectorMetric(metric), dim, maxElements, M, efConstruction, device, Synchronization::OnInsertions); });
	return m;
}
// The Map over a device list (SURVEY 8e "HNSW": a graph per shard, the per-shard results meet in the all-gather + merge of brute force);
// nDevices == 0: the list comes from RX_GPU_VECTOR_INDEXES like the in-tree adapter's (rx_seam.h GpuHnswMapT), unset -> device 0
void* rxhost_hnsw_create_sharded(int metric, size_t dim, size_t maxElements, size_t M, size_t efConstruction, const int* devices, size_t nDevices,
								 int multithread) {
	GpuHnswMap* m = nullptr;
	guarded([&] {
		std::vector<int> d = nDevices ? std::vector<int>(devices, devices + nDevices) : GpuDevicesFromEnv();
		if (d.empty()) d.push_back(0);
		m = new GpuHnswMap(VectorMetric(metric), dim, maxElements, M, efConstruction, std::move(d),
						   multithread ? Synchronization::OnInsertions : Synchronization::None);
	});
	return m;
}
size_t rxhost_hnsw_shard_count(void* h) { return static_cast<GpuHnswMap*>(h)->ShardCount(); }
size_t rxhost_hnsw_shard_rows(void* h) { return static_cast<GpuHnswMap*>(h)->ShardRows(); }
// shard s as a borrowed single-device Map (owned by the sharded one): graph export, direct searches
void* rxhost_hnsw_shard(void* h, size_t s) {
	void* out = nullptr;
	guarded([&] { out = const_cast<GpuHnswMap*>(&static_cast<GpuHnswMap*>(h)->Shard(s)); });
	return out;
}
// the device handle of the Map (sharded: the rxgpu_index_create_sharded handle) — tests read its merge mode / collectives counter
void* rxhost_hnsw_device_index(void* h) { return static_cast<GpuHnswMap*>(h)->DeviceIndex(); }
// `threads` upsert threads calling AddPointConcurrent, as HnswIndexBase<HierarchicalNSWMT>::upsertConcurrent does (hnsw_index.cc:105-116)
int rxhost_hnsw_add_many_mt(void* h, const float* vecs, size_t n, size_t dim, const uint64_t* labels, unsigned threads) {
	return guarded([&] {
		auto* m = static_cast<GpuHnswMap*>(h);
		size_t first = 0;
		if (n && m->CurrentElementCount() == 0) {
			m->AddPointConcurrent(ConstFloatVectorView(vecs, dim), FloatVectorId::FromNumber(labels[0]));
			first = 1;
		}
		std::atomic<size_t> next{first};
		std::mutex errMtx;
		std::string error;
		auto worker = [&] {
			try {
				for (;;) {
					const size_t i = next.fetch_add(1, std::memory_order_relaxed);
					if (i >= n) break;
					m->AddPointConcurrent(ConstFloatVectorView(vecs + i * dim, dim), FloatVectorId::FromNumber(labels[i]));
				}
			} catch (const std::exception& e) {
				next.store(n, std::memory_order_relaxed);
				std::lock_guard<std::mutex> lk(errMtx);
				if (error.empty()) error = e.what();
			}
		};
		std::vector<std::thread> pool;
		for (unsigned t = 0; t < std::max(1u, threads); ++t) pool.emplace_back(worker);
		for (auto& t : pool) t.join();
		if (!error.empty()) throw std::logic_error(error);
	});
}
void* rxhost_hnsw_clone(void* h, size_t newCapacity) {
	GpuHnswMap* m = nullptr;
	guarded([&] { m = new GpuHnswMap(*static_cast<GpuHnswMap*>(h), newCapacity); });
	return m;
}
void rxhost_hnsw_destroy(void* h) { delete static_cast<GpuHnswMap*>(h); }
int rxhost_hnsw_add_many(void* h, const float* vecs, size_t n, size_t dim, const uint64_t* labels) {
	return guarded([&] {
		auto* m = static_cast<GpuHnswMap*>(h);
		for (size_t i = 0; i < n; ++i) m->AddPointNoLock(ConstFloatVectorView(vecs + i * dim, dim), FloatVectorId::FromNumber(labels[i]));
	});
}
int rxhost_hnsw_add_concurrent(void* h, const float* vec, size_t dim, uint64_t label) {
	return guarded([&] { static_cast<GpuHnswMap*>(h)->AddPointConcurrent(ConstFloatVectorView(vec, dim), FloatVectorId::FromNumber(label)); });
}
int rxhost_hnsw_mark_delete(void* h, uint64_t label) {
	return guarded([&] { static_cast<GpuHnswMap*>(h)->MarkDelete(FloatVectorId::FromNumber(label)); });
}
int rxhost_hnsw_clear(void* h) {
	return guarded([&] { static_cast<GpuHnswMap*>(h)->Clear(); });
}
int rxhost_hnsw_resize(void* h, size_t n) {
	return guarded([&] { static_cast<GpuHnswMap*>(h)->ResizeIndex(n); });
}
size_t rxhost_hnsw_count(void* h) { return static_cast<GpuHnswMap*>(h)->CurrentElementCount(); }
size_t rxhost_hnsw_deleted_count(void* h) { return static_cast<GpuHnswMap*>(h)->DeletedCountUnsafe(); }
long rxhost_hnsw_tie_reruns(void* h) {
	long n = -1;
	guarded([&] { n = long(static_cast<const GpuHnswMap*>(h)->TieReruns()); });
	return n;
}
// one-shot searches of this Map that the index's resident search kernel answered (GpuHnswMap::PostedQueries)
long rxhost_hnsw_posted_queries(void* h) { return long(static_cast<const GpuHnswMap*>(h)->PostedQueries()); }
long rxhost_hnsw_lds_reruns(void* h) {
	long n = -1;
	guarded([&] { n = long(static_cast<const GpuHnswMap*>(h)->LdsReruns()); });
	return n;
}
void* rxhost_hnsw_graph(void* h) { return const_cast<HnswGraph*>(&static_cast<GpuHnswMap*>(h)->Graph()); }
// the Map's ANN disk cache through memory (same encoding as rxhost_graph_save_index / _load_index)
long rxhost_hnsw_save_index(void* h, uint8_t* out, size_t cap) {
	long n = -1;
	guarded([&] {
		MemAnnWriter w;
		const std::atomic_int32_t cancel{0};
		static_cast<const GpuHnswMap*>(h)->SaveIndex(w, cancel);
		if (w.buf.size() <= cap) std::memcpy(out, w.buf.data(), w.buf.size());
		n = long(w.buf.size());
	});
	return n;
}
static int hnswLoadIndex(void* h, const uint8_t* data, size_t len, const uint64_t* labels, const float* vectors, size_t rows, bool withQuantizer) {
	return guarded([&] {
		auto* m = static_cast<GpuHnswMap*>(h);
		MemAnnReader r(data, len, m->Dim(), labels, vectors, rows);
		r.withQuantizer = withQuantizer;
		try {
			m->LoadIndex(r);
			if (r.Remaining()) throw std::runtime_error("ANN cache: unparsed data behind the graph");
		} catch (...) {
			m->Clear();   // HnswIndexBase::LoadIndexCache's error path (clearMap)
			throw;
		}
	});
}
int rxhost_hnsw_load_index(void* h, const uint8_t* data, size_t len, const uint64_t* labels, const float* vectors, size_t rows) {
	return hnswLoadIndex(h, data, len, labels, vectors, rows, false);
}
// ... with LoadWithQuantizer_True: a cache that carries QuantizingParams brings the Map back quantised
int rxhost_hnsw_load_index_quantized(void* h, const uint8_t* data, size_t len, const uint64_t* labels, const float* vectors, size_t rows) {
	return hnswLoadIndex(h, data, len, labels, vectors, rows, true);
}
long rxhost_hnsw_search_knn(void* h, const float* q, size_t k, size_t ef, float* outDist, uint64_t* outLabel) {
	long n = -1;
	guarded([&] {
		auto res = static_cast<const GpuHnswMap*>(h)->SearchKnn(q, std::nullopt, k, ef);
		n = long(drain(res, outDist, outLabel, k));
	});
	return n;
}
// The reference's concurrency model on the GPU Map (SURVEY 8b "Threading", 8d): T planner threads, each running its own SearchKnn with ONE
// query over the shared Map (cf. runMultithreadQueries, gtests/tests/unit/float_vector_index.cc:258-294) — native threads, started outside the
// timed region; a thread stops STARTING searches once deadlineS has passed.  Thread t searches queries[(t * perThread + j) % nq].
int rxhost_hnsw_search_knn_mt(void* h, const float* queries, size_t nq, size_t dim, size_t k, size_t ef, unsigned threads, size_t perThread,
							  double deadlineS, double* outSeconds, size_t* outDone, size_t* outBatches) {
	return guarded([&] {
		const auto* m = static_cast<const GpuHnswMap*>(h);
		std::atomic<int> gate{0};
		std::atomic<size_t> done{0}, ready{0};
		std::mutex errMtx;
		std::string error;
		std::chrono::steady_clock::time_point t0;
		auto worker = [&](unsigned t) {
			ready.fetch_add(1);
			while (!gate.load(std::memory_order_acquire)) std::this_thread::yield();
			try {
				for (size_t j = 0; j < perThread; ++j) {
					if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > deadlineS) break;
					auto res = m->SearchKnn(queries + ((size_t(t) * perThread + j) % nq) * dim, std::nullopt, k, ef);
					if (res.empty()) throw std::logic_error("empty SearchKnn result");
					done.fetch_add(1, std::memory_order_relaxed);
				}
			} catch (const std::exception& e) {
				std::lock_guard<std::mutex> lk(errMtx);
				if (error.empty()) error = e.what();
			}
		};
		const size_t b0 = m->CoalescedBatches();
		std::vector<std::thread> pool;
		for (unsigned t = 0; t < std::max(1u, threads); ++t) pool.emplace_back(worker, t);
		while (ready.load() < pool.size()) std::this_thread::yield();
		t0 = std::chrono::steady_clock::now();
		gate.store(1, std::memory_order_release);
		for (auto& th : pool) th.join();
		*outSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		*outDone = done.load();
		if (outBatches) *outBatches = m->CoalescedBatches() - b0;
		if (!error.empty()) throw std::logic_error(error);
	});
}
// the quantised Map: Quantize(minQ, maxQ), SearchKnn with query_data_norm (hnsw_index.cc:168: normL2 = 1.f / NormalizeCopyVector(...))
int rxhost_hnsw_quantize(void* h, float minQ, float maxQ) {
	return guarded([&] { static_cast<GpuHnswMap*>(h)->Quantize(minQ, maxQ); });
}
int rxhost_hnsw_is_quantized(void* h) { return static_cast<GpuHnswMap*>(h)->IsQuantized() ? 1 : 0; }
// HnswIndexBase::Quantize() / SwitchMapOnQuantized() (hnsw_index.cc:532-551) through the Map: the parameters are sampled from the stored rows
// like QuantizingParams does (quantile <= 0: the default of the dimension); params5 (may be null) = minQ, maxQ, alpha, alpha_2, delta
int rxhost_hnsw_quantize_config(void* h, size_t sampleSize, float quantile, int switchOn, float* params5) {
	return guarded([&] {
		auto* m = static_cast<GpuHnswMap*>(h);
		rxgpu::host::Sq8QuantizationConfig cfg;
		cfg.sampleSize = sampleSize;
		if (quantile > 0.f) cfg.quantile = quantile;
		m->Quantize(cfg);
		if (m->QuantizationAvailable()) throw std::logic_error("QuantizationAvailable() after Quantize(config)");
		if (m->IsQuantized()) throw std::logic_error("IsQuantized() before SwitchMapOnQuantized()");
		if (switchOn) m->SwitchMapOnQuantized();
		if (params5 && switchOn) {
			const auto& p = m->QuantizingParams();
			params5[0] = p.minQ;
			params5[1] = p.maxQ;
			params5[2] = p.alpha;
			params5[3] = p.alpha_2;
			params5[4] = p.delta;
		}
	});
}
int rxhost_hnsw_switch_on_quantized(void* h) {
	return guarded([&] { static_cast<GpuHnswMap*>(h)->SwitchMapOnQuantized(); });
}
void rxhost_hnsw_quantizing_params(void* h, float* params5) {
	const auto& p = static_cast<GpuHnswMap*>(h)->QuantizingParams();
	params5[0] = p.minQ;
	params5[1] = p.maxQ;
	params5[2] = p.alpha;
	params5[3] = p.alpha_2;
	params5[4] = p.delta;
}
// Sq8FindNthMinMax / Sq8SampleIndexes (sq8_quantizer.h) for the CPU parity tests against the reference's sampler
void rxhost_sq8_find_nth_min_max(const float* v, size_t count, size_t dataSize, float quantile, float* out2) {
	const auto mm = rxgpu::host::Sq8FindNthMinMax(v, count, dataSize, quantile);
	out2[0] = mm.first;
	out2[1] = mm.second;
}
size_t rxhost_sq8_sample_indexes(size_t sampleSize, size_t size, uint32_t* out) {
	const auto ids = rxgpu::host::Sq8SampleIndexes(sampleSize, size);
	std::copy(ids.begin(), ids.end(), out);
	return ids.size();
}
// QuantizingParams(hnsw, config) over plain rows [n][dim] (Sq8SampleParams): minQ, maxQ, alpha, alpha_2, delta
int rxhost_sq8_sample_params(const float* rows, size_t n, size_t dim, size_t sampleSize, float quantile, float* params5) {
	return guarded([&] {
		rxgpu::host::Sq8QuantizationConfig cfg;
		cfg.sampleSize = sampleSize;
		if (quantile > 0.f) cfg.quantile = quantile;
		const auto p = rxgpu::host::Sq8SampleParams(n, dim, cfg, [&](uint32_t id) { return rows + size_t(id) * dim; });
		params5[0] = p.minQ;
		params5[1] = p.maxQ;
		params5[2] = p.alpha;
		params5[3] = p.alpha_2;
		params5[4] = p.delta;
	});
}
long rxhost_hnsw_search_knn_norm(void* h, const float* q, int hasNorm, float norm, size_t k, size_t ef, float* outDist, uint64_t* outLabel) {
	long n = -1;
	guarded([&] {
		auto res = static_cast<const GpuHnswMap*>(h)->SearchKnn(q, hasNorm ? std::optional<float>(norm) : std::nullopt, k, ef);
		n = long(drain(res, outDist, outLabel, k));
	});
	return n;
}
// Sq8Quantize / Sq8Params (sq8_quantizer.h) for the CPU parity tests against Quantizer::Quantize
float rxhost_sq8_quantize(int metric, float minQ, float maxQ, size_t dim, const float* from, float scale, uint8_t* to, float* params3) {
	const Sq8Params p = Sq8Params::FromRange(minQ, maxQ, dim);
	if (params3) {
		params3[0] = p.alpha;
		params3[1] = p.alpha_2;
		params3[2] = p.delta;
	}
	return Sq8Quantize(VectorMetric(metric), p, from, dim, scale, to);
}
// n vectors at once (rows of a corpus, or a query batch with one scale per query), split over `threads` host threads
void rxhost_sq8_quantize_many(int metric, float minQ, float maxQ, size_t dim, const float* from, size_t n, const float* scales, uint8_t* to,
							  float* corr, unsigned threads) {
	const Sq8Params p = Sq8Params::FromRange(minQ, maxQ, dim);
	threads = std::max(1u, std::min<unsigned>(threads ? threads : 1u, unsigned(std::max<size_t>(n / 64, 1))));
	auto work = [&](size_t a, size_t b) {
		for (size_t i = a; i < b; ++i) corr[i] = Sq8Quantize(VectorMetric(metric), p, from + i * dim, dim, scales ? scales[i] : 1.f, to + i * dim);
	};
	std::vector<std::thread> pool;
	const size_t per = (n + threads - 1) / threads;
	for (unsigned t = 1; t < threads; ++t) {
		if (t * per < n) pool.emplace_back(work, t * per, std::min(n, (t + 1) * per));
	}
	work(0, std::min(n, per));
	for (auto& th : pool) th.join();
}
// streaming session, driven like HnswIndexBase<Map>::beginStreaming / continueStreaming (hnsw_index.cc:318-351)
void* rxhost_hnsw_stream_begin(void* h, const float* q, size_t ef) {
	StreamingSearchSession* s = nullptr;
	guarded([&] { s = new StreamingSearchSession(static_cast<const GpuHnswMap*>(h)->BeginStreamingSearch(q, std::nullopt, StreamingSearchOptions{ef})); });
	return s;
}
// with the query's norm (a quantised cosine graph needs it: queryNormCoef, hnswalg.h:1855-1863)
void* rxhost_hnsw_stream_begin_norm(void* h, const float* q, int hasNorm, float norm, size_t ef) {
	StreamingSearchSession* s = nullptr;
	guarded([&] {
		s = new StreamingSearchSession(static_cast<const GpuHnswMap*>(h)->BeginStreamingSearch(q, hasNorm ? std::optional<float>(norm) : std::nullopt,
																								   StreamingSearchOptions{ef}));
	});
	return s;
}
long rxhost_hnsw_search_range_norm(void* h, const float* q, int hasNorm, float norm, float radius, size_t ef, float* outDist, uint64_t* outLabel, size_t cap) {
	long n = -1;
	guarded([&] {
		auto res = static_cast<const GpuHnswMap*>(h)->SearchRange(q, hasNorm ? std::optional<float>(norm) : std::nullopt, radius, ef);
		n = long(drain(res, outDist, outLabel, cap));
	});
	return n;
}
// pops the batch's result queue: worst first under (dist, label).  Returns the count, -1 on error.
long rxhost_hnsw_stream_continue(void* h, void* session, size_t batch, float* outDist, uint64_t* outLabel, int* exhausted) {
	long n = -1;
	guarded([&] {
		auto b = static_cast<const GpuHnswMap*>(h)->ContinueStreamingSearch(*static_cast<StreamingSearchSession*>(session), batch);
		*exhausted = b.exhausted ? 1 : 0;
		n = 0;
		for (; !b.results.empty(); b.results.pop()) {
			outDist[n] = b.results.top().first;
			outLabel[n] = b.results.top().second;
			++n;
		}
	});
	return n;
}
void rxhost_hnsw_stream_end(void* session) { delete static_cast<StreamingSearchSession*>(session); }
long rxhost_hnsw_search_range(void* h, const float* q, float radius, size_t ef, float* outDist, uint64_t* outLabel, size_t cap) {
	long n = -1;
	guarded([&] {
		auto res = static_cast<const GpuHnswMap*>(h)->SearchRange(q, std::nullopt, radius, ef);
		n = long(drain(res, outDist, outLabel, cap));
	});
	return n;
}
long rxhost_hnsw_select(void* h, const float* key, size_t dim, long k, size_t ef, int has_radius, float radius, int need_sort, int is_array,
						int32_t* outIds, float* outRanks, size_t cap) {
	long n = -1;
	guarded([&] {
		KnnSearchParams p;
		if (k >= 0) p.k = size_t(k);
		if (has_radius) p.radius = radius;
		p.ef = ef;
		p.Validate(true);
		auto res = KnnSelect(*static_cast<const GpuHnswMap*>(h), ConstFloatVectorView(key, dim), p, need_sort != 0, is_array != 0);
		n = long(res.ids.size());
		for (size_t i = 0; i < res.ids.size() && i < cap; ++i) {
			outIds[i] = res.ids[i];
			outRanks[i] = res.ranks[i];
		}
	});
	return n;
}

}  // extern "C"

extern "C" {
// index-level streaming (HnswIndexBase<Map>::beginStreaming / continueStreaming): raw key in, (row id, user-visible rank) best first out
void* rxhost_hnsw_knn_stream_begin(void* h, const float* key, size_t dim, size_t ef) {
	using S = decltype(KnnBeginStreaming(*static_cast<const GpuHnswMap*>(h), ConstFloatVectorView(key, dim), ef));
	S* s = nullptr;
	guarded([&] { s = new S(KnnBeginStreaming(*static_cast<const GpuHnswMap*>(h), ConstFloatVectorView(key, dim), ef)); });
	return s;
}
long rxhost_hnsw_knn_stream_continue(void* h, void* session, size_t batch, int32_t* outIds, float* outRanks, int* exhausted) {
	using S = decltype(KnnBeginStreaming(*static_cast<const GpuHnswMap*>(h), ConstFloatVectorView(nullptr, 0), 0));
	long n = -1;
	guarded([&] {
		KnnStreamingBatch b;
		KnnContinueStreaming(*static_cast<const GpuHnswMap*>(h), *static_cast<S*>(session), batch, b);
		*exhausted = b.exhausted ? 1 : 0;
		n = long(b.ids.size());
		for (size_t i = 0; i < b.ids.size(); ++i) {
			outIds[i] = b.ids[i];
			outRanks[i] = b.ranks[i];
		}
	});
	return n;
}
void rxhost_hnsw_knn_stream_end(void* h, void* session) {
	using S = decltype(KnnBeginStreaming(*static_cast<const GpuHnswMap*>(h), ConstFloatVectorView(nullptr, 0), 0));
	delete static_cast<S*>(session);
}
}  // extern "C"

extern "C" {
void rxhost_hnsw_enable_coalescing(void* h, int on) { static_cast<GpuHnswMap*>(h)->EnableQueryCoalescing(on != 0); }
void rxhost_hnsw_set_coalescer_lanes(void* h, unsigned lanes) { static_cast<GpuHnswMap*>(h)->SetCoalescerLanes(lanes); }
void rxhost_bf_enable_coalescing(void* h, int on) { static_cast<GpuBruteforceMap*>(h)->EnableQueryCoalescing(on != 0); }
void rxhost_bf_coalescing_stats(void* h, uint64_t* batches, uint64_t* queries) {
	*batches = static_cast<const GpuBruteforceMap*>(h)->CoalescedBatches();
	*queries = static_cast<const GpuBruteforceMap*>(h)->CoalescedQueries();
}
}  // extern "C"

// ---------------------------------------------------------------------------------------------- GpuFtMerger
#include "gpu_ft_merger.h"

extern "C" {

void* rxhost_ft_create(size_t numFields, int device) {
	GpuFtMerger* m = nullptr;
	guarded([&] { m = new GpuFtMerger(numFields, device); });
	return m;
}
// the merger over a device list: document-range shards (SURVEY 8e "BM25")
void* rxhost_ft_create_sharded(size_t numFields, const int* devices, size_t nDevices) {
	GpuFtMerger* m = nullptr;
	guarded([&] { m = new GpuFtMerger(numFields, std::vector<int>(devices, devices + nDevices)); });
	return m;
}
void* rxhost_ft_device_index(void* h) { return static_cast<GpuFtMerger*>(h)->DeviceIndex(); }
void rxhost_ft_destroy(void* h) { delete static_cast<GpuFtMerger*>(h); }
int rxhost_ft_set_docs(void* h, size_t totalDocs, const float* words, const float* avg, const uint8_t* removed) {
	return guarded([&] { static_cast<GpuFtMerger*>(h)->SetDocs(totalDocs, words, avg, removed); });
}
// postings given as IdRelType-like records: doc[i] with positions [pos_off[i], pos_off[i+1]) of (field, pos) pairs
int rxhost_ft_set_word(void* h, uint32_t wordId, size_t n, const uint32_t* doc, const uint32_t* posOff, const uint32_t* posField, const uint32_t* posPos) {
	return guarded([&] {
		FlatPostings fp;
		std::vector<std::pair<uint32_t, uint32_t>> tmp;
		for (size_t i = 0; i < n; ++i) {
			tmp.clear();
			for (uint32_t j = posOff[i]; j < posOff[i + 1]; ++j) tmp.emplace_back(posField[j], posPos[j]);
			fp.Add(doc[i], tmp.data(), tmp.size());
		}
		static_cast<GpuFtMerger*>(h)->SetWord(wordId, fp);
	});
}
// cfgD: [k1, b, summationRatio, fullMatchBoost] ; cfgI: [minRank, mergeLimit] ; fieldCfg: [nf][6] doubles (bm25Boost, bm25Weight,
// termLenBoost, termLenWeight, positionBoost, positionWeight) ; opts: boost, termLenBoost, fieldBoost[nf], needSum[nf]
long rxhost_ft_merge(void* h, size_t nf, const double* cfgD, const int* cfgI, const double* fieldCfg, float boost, float termLenBoost,
					 const float* fieldBoost, const uint8_t* needSum, size_t nsub, const uint32_t* wordIds, const float* procs,
					 const uint8_t* excluded, int sortByRank, int32_t* outId, float* outProc, uint8_t* outField, uint8_t* outNorm, size_t cap) {
	long n = -1;
	guarded([&] {
		FtConfig cfg(nf);
		cfg.bm25k1 = cfgD[0];
		cfg.bm25b = cfgD[1];
		cfg.summationRanksByFieldsRatio = cfgD[2];
		cfg.fullMatchBoost = cfgD[3];
		cfg.minRank = cfgI[0];
		cfg.mergeLimit = uint32_t(cfgI[1]);
		cfg.bm25Type = cfgI[2] == 1 ? FtConfig::Bm25Type::Classic : (cfgI[2] == 2 ? FtConfig::Bm25Type::WordCount : FtConfig::Bm25Type::Rx);
		FtDslOpts opts;
		opts.boost = boost;
		opts.termLenBoost = termLenBoost;
		opts.fieldsOpts.resize(nf);
		for (size_t f = 0; f < nf; ++f) {
			cfg.fieldsCfg[f] = FtFieldConfig{fieldCfg[f * 6 + 0], fieldCfg[f * 6 + 1], fieldCfg[f * 6 + 2], fieldCfg[f * 6 + 3], fieldCfg[f * 6 + 4], fieldCfg[f * 6 + 5]};
			opts.fieldsOpts[f] = FtDslFieldOpts{fieldBoost[f], needSum[f] != 0};
		}
		std::vector<SubtermRef> subs(nsub);
		for (size_t i = 0; i < nsub; ++i) subs[i] = SubtermRef{wordIds[i], procs[i]};
		auto res = static_cast<const GpuFtMerger*>(h)->Merge(cfg, opts, std::move(subs), excluded, sortByRank ? RankSortType::RankOnly : RankSortType::RankAndID);
		n = long(res.size());
		for (size_t i = 0; i < res.size() && i < cap; ++i) {
			outId[i] = res[i].id;
			outProc[i] = res[i].proc;
			outField[i] = res[i].field;
			outNorm[i] = res[i].normalizedProc;
		}
	});
	return n;
}
int rxhost_ft_read_stats(void* h, uint64_t* postings, double* ms) {
	return guarded([&] { static_cast<const GpuFtMerger*>(h)->ReadStats(*postings, *ms); });
}

}  // extern "C"

extern "C" int rxhost_ft_set_word_flat(void* h, uint32_t wordId, size_t n, const uint32_t* doc, const uint32_t* entOff, const uint8_t* entField,
										const uint32_t* entTf, const uint32_t* entFirstPos) {
	return guarded([&] {
		FlatPostings fp;
		fp.doc.assign(doc, doc + n);
		fp.entOff.assign(entOff, entOff + n + 1);
		const size_t ne = n ? entOff[n] : 0;
		fp.entField.assign(entField, entField + ne);
		fp.entTf.assign(entTf, entTf + ne);
		fp.entFirstPos.assign(entFirstPos, entFirstPos + ne);
		static_cast<GpuFtMerger*>(h)->SetWord(wordId, fp);
	});
}

extern "C" int rxhost_ft_set_word_fpos(void* h, uint32_t wordId, size_t n, const uint32_t* doc, const uint32_t* posOff, const uint64_t* fpos) {
	return guarded([&] {
		PositionPostings pp;
		pp.doc.assign(doc, doc + n);
		pp.posOff.assign(posOff, posOff + n + 1);
		pp.fpos.assign(fpos, fpos + (n ? posOff[n] : 0));
		static_cast<GpuFtMerger*>(h)->SetWord(wordId, pp);
	});
}
// cfgD: [k1, b, summationRatio, fullMatchBoost, distanceBoost, distanceWeight]; per term: op, boost, termLenBoost, fieldBoost[nf],
// needSum[nf], sub-term slice [subOff[t], subOff[t+1]) of (wordId, proc).  Returns the result count, -1 on error.
extern "C" void rxhost_ft_read_packed_stats(void* h, double* countMs, double* writeMs, uint64_t* bytesIn, uint64_t* bytesOut) {
	static_cast<const GpuFtMerger*>(h)->ReadPackedStats(*countMs, *writeMs, *bytesIn, *bytesOut);
}
extern "C" void rxhost_ft_read_fuse_stats(void* h, uint64_t* calls, double* kernelMs, double* prepareMs) {
	static_cast<const GpuFtMerger*>(h)->ReadFuseStats(*calls, *kernelMs, prepareMs);
}
extern "C" void rxhost_ft_read_timing(void* h, uint64_t* calls, double* totalMs) { static_cast<const GpuFtMerger*>(h)->ReadTiming(*calls, *totalMs); }

namespace {
FtConfig parseFtConfig(size_t nf, const double* cfgD, const int* cfgI, const double* fieldCfg) {
	FtConfig cfg(nf);
	cfg.bm25k1 = cfgD[0];
	cfg.bm25b = cfgD[1];
	cfg.summationRanksByFieldsRatio = cfgD[2];
	cfg.fullMatchBoost = cfgD[3];
	cfg.distanceBoost = cfgD[4];
	cfg.distanceWeight = cfgD[5];
	cfg.minRank = cfgI[0];
	cfg.mergeLimit = uint32_t(cfgI[1]);
	cfg.bm25Type = cfgI[2] == 1 ? FtConfig::Bm25Type::Classic : (cfgI[2] == 2 ? FtConfig::Bm25Type::WordCount : FtConfig::Bm25Type::Rx);
	for (size_t f = 0; f < nf; ++f) {
		cfg.fieldsCfg[f] = FtFieldConfig{fieldCfg[f * 6 + 0], fieldCfg[f * 6 + 1], fieldCfg[f * 6 + 2], fieldCfg[f * 6 + 3], fieldCfg[f * 6 + 4], fieldCfg[f * 6 + 5]};
	}
	return cfg;
}
std::vector<QueryTerm> parseFtTerms(size_t nf, size_t nTerms, const int* ops, const float* boosts, const float* termLenBoosts, const float* fieldBoost,
									const uint8_t* needSum, const uint32_t* subOff, const uint32_t* wordIds, const float* procs) {
	std::vector<QueryTerm> terms(nTerms);
	for (size_t t = 0; t < nTerms; ++t) {
		terms[t].op = OpType(ops[t]);
		terms[t].opts.boost = boosts[t];
		terms[t].opts.termLenBoost = termLenBoosts[t];
		terms[t].opts.fieldsOpts.resize(nf);
		for (size_t f = 0; f < nf; ++f) terms[t].opts.fieldsOpts[f] = FtDslFieldOpts{fieldBoost[t * nf + f], needSum[t * nf + f] != 0};
		for (uint32_t s = subOff[t]; s < subOff[t + 1]; ++s) terms[t].subterms.push_back(SubtermRef{wordIds[s], procs[s]});
	}
	return terms;
}
}  // namespace

// phraseNum / distance: FtDslOpts::phraseNum (-1: a plain term) and FtDslOpts::distance per term, or null (no phrases)
extern "C" long rxhost_ft_merge_query_phrases(void* h, size_t nf, const double* cfgD, const int* cfgI, const double* fieldCfg, size_t nTerms, const int* ops,
											  const float* boosts, const float* termLenBoosts, const float* fieldBoost, const uint8_t* needSum,
											  const int* phraseNum, const int* distance, const uint32_t* subOff, const uint32_t* wordIds, const float* procs,
											  const uint8_t* excluded, int sortByRank, int32_t* outId, float* outProc, uint8_t* outField, uint8_t* outNorm,
											  size_t cap, int* outPreselected) {
	long n = -1;
	guarded([&] {
		const FtConfig cfg = parseFtConfig(nf, cfgD, cfgI, fieldCfg);
		std::vector<QueryTerm> terms = parseFtTerms(nf, nTerms, ops, boosts, termLenBoosts, fieldBoost, needSum, subOff, wordIds, procs);
		for (size_t t = 0; t < nTerms; ++t) {
			if (phraseNum) terms[t].phraseNum = phraseNum[t];
			if (distance) terms[t].distance = distance[t];
		}
		bool pre = false;
		auto res = static_cast<const GpuFtMerger*>(h)->MergeQuery(cfg, std::move(terms), excluded, sortByRank ? RankSortType::RankOnly : RankSortType::RankAndID, &pre);
		if (outPreselected) *outPreselected = pre ? 1 : 0;
		n = long(res.size());
		for (size_t i = 0; i < res.size() && i < cap; ++i) {
			outId[i] = res[i].id;
			outProc[i] = res[i].proc;
			outField[i] = res[i].field;
			outNorm[i] = res[i].normalizedProc;
		}
	});
	return n;
}
// ... plus multi-word synonyms: per-term arrays hold nTerms + nSynTerms entries; synonym s = terms nTerms + synTermOff[s] .. nTerms + synTermOff[s + 1];
// partSynOff [nParts + 1] / partSyn: the synonyms of every query part.  Sub-terms of the synonyms whose word the query's own plain terms found
// are marked suppressed here, like QueryMergeData::SupressDuplicatesInSynonyms does in front of the reference's merge (selecterimpl.h:606).
namespace {
// the per-term arrays of a query with multi-word synonyms (rxhost_ft_merge_query_full's layout) -> the parts' terms + QuerySynonyms
void splitSynonyms(std::vector<QueryTerm>& all, size_t nTerms, const int* phraseNum, const int* distance, size_t nSyn, const uint32_t* synTermOff, size_t nParts,
				   const uint32_t* partSynOff, const uint32_t* partSyn, std::vector<QueryTerm>& terms, QuerySynonyms& syn) {
	for (size_t t = 0; t < nTerms; ++t) {
		if (phraseNum) all[t].phraseNum = phraseNum[t];
		if (distance) all[t].distance = distance[t];
	}
	terms.assign(all.begin(), all.begin() + nTerms);
	std::vector<uint32_t> found;   // words of the plain (non-phrase) query terms
	for (const QueryTerm& t : terms) {
		if (t.phraseNum < 0) {
			for (const SubtermRef& s : t.subterms) found.push_back(s.wordId);
		}
	}
	std::sort(found.begin(), found.end());
	for (size_t sy = 0; sy < nSyn; ++sy) {
		syn.synonyms.emplace_back();
		for (uint32_t k = synTermOff[sy]; k < synTermOff[sy + 1]; ++k) {
			QueryTerm t = all[nTerms + k];
			for (SubtermRef& s : t.subterms) s.suppressed = std::binary_search(found.begin(), found.end(), s.wordId);
			syn.synonyms.back().push_back(std::move(t));
		}
	}
	for (size_t pi = 0; pi < nParts; ++pi) syn.partSynonyms.emplace_back(partSyn + partSynOff[pi], partSyn + partSynOff[pi + 1]);
}
}  // namespace

extern "C" long rxhost_ft_merge_query_full(void* h, size_t nf, const double* cfgD, const int* cfgI, const double* fieldCfg, size_t nTerms, size_t nSynTerms,
										   const int* ops, const float* boosts, const float* termLenBoosts, const float* fieldBoost, const uint8_t* needSum,
										   const int* phraseNum, const int* distance, const uint32_t* subOff, const uint32_t* wordIds, const float* procs,
										   size_t nSyn, const uint32_t* synTermOff, size_t nParts, const uint32_t* partSynOff, const uint32_t* partSyn,
										   const uint8_t* excluded, int sortByRank, int32_t* outId, float* outProc, uint8_t* outField, uint8_t* outNorm,
										   size_t cap, int* outPreselected) {
	long n = -1;
	guarded([&] {
		const FtConfig cfg = parseFtConfig(nf, cfgD, cfgI, fieldCfg);
		std::vector<QueryTerm> all = parseFtTerms(nf, nTerms + nSynTerms, ops, boosts, termLenBoosts, fieldBoost, needSum, subOff, wordIds, procs);
		std::vector<QueryTerm> terms;
		QuerySynonyms syn;
		splitSynonyms(all, nTerms, phraseNum, distance, nSyn, synTermOff, nParts, partSynOff, partSyn, terms, syn);
		bool pre = false;
		auto res = static_cast<const GpuFtMerger*>(h)->MergeQuery(cfg, std::move(terms), std::move(syn), excluded,
																  sortByRank ? RankSortType::RankOnly : RankSortType::RankAndID, &pre);
		if (outPreselected) *outPreselected = pre ? 1 : 0;
		n = long(res.size());
		for (size_t i = 0; i < res.size() && i < cap; ++i) {
			outId[i] = res[i].id;
			outProc[i] = res[i].proc;
			outField[i] = res[i].field;
			outNorm[i] = res[i].normalizedProc;
		}
	});
	return n;
}
extern "C" long rxhost_ft_merge_query(void* h, size_t nf, const double* cfgD, const int* cfgI, const double* fieldCfg, size_t nTerms, const int* ops,
									  const float* boosts, const float* termLenBoosts, const float* fieldBoost, const uint8_t* needSum,
									  const uint32_t* subOff, const uint32_t* wordIds, const float* procs, const uint8_t* excluded, int sortByRank,
									  int32_t* outId, float* outProc, uint8_t* outField, uint8_t* outNorm, size_t cap, int* outPreselected) {
	return rxhost_ft_merge_query_phrases(h, nf, cfgD, cfgI, fieldCfg, nTerms, ops, boosts, termLenBoosts, fieldBoost, needSum, nullptr, nullptr, subOff, wordIds,
										 procs, excluded, sortByRank, outId, outProc, outField, outNorm, cap, outPreselected);
}

// Q queries in ONE launch train (GpuFtMerger::MergeQueryBatch).  The queries' terms back to back: query i = terms [termOff[i], termOff[i + 1]);
// the per-term arrays and subOff ([all terms + 1], into wordIds / procs) cover all of them.  Outputs: row i of [nQueries][cap] arrays, outN[i].
extern "C" int rxhost_ft_merge_query_batch(void* h, size_t nf, const double* cfgD, const int* cfgI, const double* fieldCfg, size_t nQueries, const uint32_t* termOff,
										   const int* ops, const float* boosts, const float* termLenBoosts, const float* fieldBoost, const uint8_t* needSum,
										   const int* phraseNum, const int* distance, const uint32_t* subOff, const uint32_t* wordIds, const float* procs,
										   int sortByRank, int32_t* outId, float* outProc, uint8_t* outField, uint8_t* outNorm, size_t cap, long* outN,
										   int* outPreselected) {
	int rc = -1;
	guarded([&] {
		const FtConfig cfg = parseFtConfig(nf, cfgD, cfgI, fieldCfg);
		const size_t nTerms = termOff[nQueries];
		std::vector<QueryTerm> all = parseFtTerms(nf, nTerms, ops, boosts, termLenBoosts, fieldBoost, needSum, subOff, wordIds, procs);
		for (size_t t = 0; t < nTerms; ++t) {
			if (phraseNum) all[t].phraseNum = phraseNum[t];
			if (distance) all[t].distance = distance[t];
		}
		std::vector<std::vector<QueryTerm>> queries(nQueries);
		for (size_t i = 0; i < nQueries; ++i) queries[i].assign(all.begin() + termOff[i], all.begin() + termOff[i + 1]);
		std::vector<uint8_t> pre;
		const auto res = static_cast<const GpuFtMerger*>(h)->MergeQueryBatch(cfg, std::move(queries), {}, sortByRank ? RankSortType::RankOnly : RankSortType::RankAndID, &pre);
		for (size_t i = 0; i < nQueries; ++i) {
			outN[i] = long(res[i].size());
			if (outPreselected) outPreselected[i] = pre[i];
			for (size_t j = 0; j < res[i].size() && j < cap; ++j) {
				outId[i * cap + j] = res[i][j].id;
				outProc[i * cap + j] = res[i][j].proc;
				outField[i * cap + j] = res[i][j].field;
				outNorm[i * cap + j] = res[i][j].normalizedProc;
			}
		}
		rc = 0;
	});
	return rc;
}

// `threads` callers issue the same query `repeats` times each against ONE merger (what several planner threads of a server do to one text
// index): wall time of the whole run in *wallMs, the callers' merges spread over the handle's lanes.  Returns the result count of a merge
// (every one is checked against the first: same count, same ids), -1 on error.
extern "C" long rxhost_ft_merge_query_concurrent(void* h, size_t nf, const double* cfgD, const int* cfgI, const double* fieldCfg, size_t nTerms,
												 const int* ops, const float* boosts, const float* termLenBoosts, const float* fieldBoost,
												 const uint8_t* needSum, const int* phraseNum, const int* distance, const uint32_t* subOff,
												 const uint32_t* wordIds, const float* procs, const uint8_t* excluded, int threads, int repeats, double* wallMs) {
	long n = -1;
	guarded([&] {
		const FtConfig cfg = parseFtConfig(nf, cfgD, cfgI, fieldCfg);
		std::vector<QueryTerm> terms = parseFtTerms(nf, nTerms, ops, boosts, termLenBoosts, fieldBoost, needSum, subOff, wordIds, procs);
		for (size_t t = 0; t < nTerms; ++t) {
			if (phraseNum) terms[t].phraseNum = phraseNum[t];
			if (distance) terms[t].distance = distance[t];
		}
		const auto* m = static_cast<const GpuFtMerger*>(h);
		const MergeData first = m->MergeQuery(cfg, terms, excluded, RankSortType::RankAndID);
		std::atomic<int> bad{0};
		std::vector<std::thread> pool;
		const auto t0 = std::chrono::steady_clock::now();
		for (int t = 0; t < threads; ++t) {
			pool.emplace_back([&] {
				try {
					for (int r = 0; r < repeats; ++r) {
						const MergeData res = m->MergeQuery(cfg, terms, excluded, RankSortType::RankAndID);
						bool same = res.size() == first.size();
						for (size_t i = 0; same && i < res.size(); ++i) same = res[i].id == first[i].id && res[i].normalizedProc == first[i].normalizedProc;
						if (!same) bad.fetch_add(1);
					}
				} catch (...) {
					bad.fetch_add(1);
				}
			});
		}
		for (auto& th : pool) th.join();
		if (wallMs) *wallMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
		if (bad.load()) throw std::runtime_error("rxhost_ft_merge_query_concurrent: a concurrent merge differs from the first one");
		n = long(first.size());
	});
	return n;
}

// Hybrid query through the Merger class: the FT merge stays in HBM (MergeQueryResident), then the fusion with a KNN result that lies in
// HBM (FuseResident).  dKnn*: device pointers ((dist, row) best first as rxgpu_search_knn_device left them), knnStream: that search's stream.
// hybrid: kind (0 RRF / 1 linear), isUnion, desc; params[5].  Returns the fused count (-1 on error); *outTie = boundary-tie flag.
extern "C" long rxhost_ft_hybrid_query(void* h, size_t nf, const double* cfgD, const int* cfgI, const double* fieldCfg, size_t nTerms, const int* ops,
									   const float* boosts, const float* termLenBoosts, const float* fieldBoost, const uint8_t* needSum,
									   const uint32_t* subOff, const uint32_t* wordIds, const float* procs, const uint8_t* excluded, const int* hybrid,
									   const double* params, int metric, const void* dKnnDist, const void* dKnnRow, const void* dKnnCount, uint32_t knnEntries,
									   uint32_t k, void* knnStream, const void* dRowOfDoc, const void* dRowIdOfRow, int32_t* outId, float* outRank, size_t cap,
									   int* outTie) {
	long n = -1;
	guarded([&] {
		const FtConfig cfg = parseFtConfig(nf, cfgD, cfgI, fieldCfg);
		std::vector<QueryTerm> terms = parseFtTerms(nf, nTerms, ops, boosts, termLenBoosts, fieldBoost, needSum, subOff, wordIds, procs);
		const auto* m = static_cast<const GpuFtMerger*>(h);
		m->MergeQueryResident(cfg, std::move(terms), excluded);
		HybridFuseParams hp;
		hp.linear = hybrid[0] == 1;
		hp.isUnion = hybrid[1] != 0;
		hp.desc = hybrid[2] != 0;
		for (int i = 0; i < 5; ++i) hp.params[i] = params[i];
		const HybridFused res = m->FuseResident(cfg, hp, metric, dKnnDist, dKnnRow, dKnnCount, knnEntries, k, knnStream, dRowOfDoc, dRowIdOfRow);
		if (outTie) *outTie = res.knnBoundaryTie ? 1 : 0;
		n = long(res.ids.size());
		for (size_t i = 0; i < res.ids.size() && i < cap; ++i) {
			outId[i] = res.ids[i];
			outRank[i] = res.ranks[i];
		}
	});
	return n;
}

// The whole hybrid query through the two engines' classes (hybrid_query.h): KNN half + FT half left in HBM, fused there, one list back.
// key: the query vector as the user gave it (normalised inside for cosine).  Returns the fused count (-1 on error).
#include "hybrid_query.h"
extern "C" long rxhost_hybrid_query_resident(void* mapHandle, void* ftHandle, size_t nf, const double* cfgD, const int* cfgI, const double* fieldCfg,
											 size_t nTerms, const int* ops, const float* boosts, const float* termLenBoosts, const float* fieldBoost,
											 const uint8_t* needSum, const uint32_t* subOff, const uint32_t* wordIds, const float* procs, const uint8_t* excluded,
											 const int* hybrid, const double* params, const float* key, size_t k, const void* dRowOfDoc,
											 const int32_t* hostRowOfDoc, int32_t* outId, float* outRank, size_t cap, int* outTie) {
	long n = -1;
	guarded([&] {
		const FtConfig cfg = parseFtConfig(nf, cfgD, cfgI, fieldCfg);
		const std::vector<QueryTerm> terms = parseFtTerms(nf, nTerms, ops, boosts, termLenBoosts, fieldBoost, needSum, subOff, wordIds, procs);
		HybridFuseParams hp;
		hp.linear = hybrid[0] == 1;
		hp.isUnion = hybrid[1] != 0;
		hp.desc = hybrid[2] != 0;
		for (int i = 0; i < 5; ++i) hp.params[i] = params[i];
		const HybridFused res = HybridQueryResident(*static_cast<const GpuBruteforceMap*>(mapHandle), *static_cast<const GpuFtMerger*>(ftHandle), cfg, terms,
													excluded, key, k, hp, dRowOfDoc, hostRowOfDoc);
		if (outTie) *outTie = res.knnBoundaryTie ? 1 : 0;
		n = long(res.ids.size());
		for (size_t i = 0; i < res.ids.size() && i < cap; ++i) {
			outId[i] = res.ids[i];
			outRank[i] = res.ranks[i];
		}
	});
	return n;
}

// ... for a query with multi-word synonyms (per-term arrays as in rxhost_ft_merge_query_full)
extern "C" long rxhost_hybrid_query_resident_full(void* mapHandle, void* ftHandle, size_t nf, const double* cfgD, const int* cfgI, const double* fieldCfg,
												  size_t nTerms, size_t nSynTerms, const int* ops, const float* boosts, const float* termLenBoosts,
												  const float* fieldBoost, const uint8_t* needSum, const int* phraseNum, const int* distance, const uint32_t* subOff,
												  const uint32_t* wordIds, const float* procs, size_t nSyn, const uint32_t* synTermOff, size_t nParts,
												  const uint32_t* partSynOff, const uint32_t* partSyn, const uint8_t* excluded, const int* hybrid, const double* params,
												  const float* key, size_t k, const void* dRowOfDoc, const int32_t* hostRowOfDoc, int32_t* outId, float* outRank,
												  size_t cap, int* outTie) {
	long n = -1;
	guarded([&] {
		const FtConfig cfg = parseFtConfig(nf, cfgD, cfgI, fieldCfg);
		std::vector<QueryTerm> all = parseFtTerms(nf, nTerms + nSynTerms, ops, boosts, termLenBoosts, fieldBoost, needSum, subOff, wordIds, procs);
		std::vector<QueryTerm> terms;
		QuerySynonyms syn;
		splitSynonyms(all, nTerms, phraseNum, distance, nSyn, synTermOff, nParts, partSynOff, partSyn, terms, syn);
		HybridFuseParams hp;
		hp.linear = hybrid[0] == 1;
		hp.isUnion = hybrid[1] != 0;
		hp.desc = hybrid[2] != 0;
		for (int i = 0; i < 5; ++i) hp.params[i] = params[i];
		const HybridFused res = HybridQueryResident(*static_cast<const GpuBruteforceMap*>(mapHandle), *static_cast<const GpuFtMerger*>(ftHandle), cfg, terms,
													excluded, key, k, hp, dRowOfDoc, hostRowOfDoc, &syn);
		if (outTie) *outTie = res.knnBoundaryTie ? 1 : 0;
		n = long(res.ids.size());
		for (size_t i = 0; i < res.ids.size() && i < cap; ++i) {
			outId[i] = res.ids[i];
			outRank[i] = res.ranks[i];
		}
	});
	return n;
}

// PackedIdRelVec byte stream -> positions-format postings (host decode, no device needed).  Returns the posting count, -1 on error;
// with null outputs only counts (*nPositions = total positions).
extern "C" long rxhost_ft_unpack(const uint8_t* data, size_t len, size_t arrayFoundPos, uint32_t* outDoc, uint32_t* outPosOff, uint64_t* outFpos,
								 size_t* nPositions) {
	long n = -1;
	guarded([&] {
		PositionPostings pp;
		pp.AppendPacked(data, len, arrayFoundPos);
		if (nPositions) *nPositions = pp.fpos.size();
		if (outDoc) std::copy(pp.doc.begin(), pp.doc.end(), outDoc);
		if (outPosOff) std::copy(pp.posOff.begin(), pp.posOff.end(), outPosOff);
		if (outFpos) std::copy(pp.fpos.begin(), pp.fpos.end(), outFpos);
		n = long(pp.doc.size());
	});
	return n;
}
extern "C" int rxhost_ft_set_word_packed(void* h, uint32_t wordId, const uint8_t* data, size_t len, size_t arrayFoundPos) {
	return guarded([&] {
		PositionPostings pp;
		pp.AppendPacked(data, len, arrayFoundPos);
		static_cast<GpuFtMerger*>(h)->SetWord(wordId, pp);
	});
}

// bulk: nwords PackedIdRelVec streams back to back, decoded on the device (streams of hostFromBytes bytes or more: on the host)
extern "C" int rxhost_ft_set_words_packed(void* h, uint32_t nwords, const uint32_t* wordIds, const uint64_t* byteOff, const uint8_t* bytes,
										  const uint64_t* arrayFoundPos, size_t hostFromBytes) {
	return guarded([&] {
		std::vector<GpuFtMerger::PackedWord> ws(nwords);
		for (uint32_t i = 0; i < nwords; ++i) ws[i] = {wordIds[i], bytes + byteOff[i], size_t(byteOff[i + 1] - byteOff[i]), size_t(arrayFoundPos[i])};
		static_cast<GpuFtMerger*>(h)->SetWordsPacked(ws, hostFromBytes);
	});
}
extern "C" double rxhost_ft_read_packed_wall(void* h) {
	double ms = -1.0;
	guarded([&] { static_cast<const GpuFtMerger*>(h)->ReadPackedWall(ms); });
	return ms;
}
// a word's device arrays; sizes[4] = {n, npos, nent, nRanges}; with null arrays only the sizes
extern "C" int rxhost_ft_get_word(void* h, uint32_t wordId, uint64_t* sizes, uint32_t* doc, uint32_t* posOff, uint64_t* fpos, uint32_t* entOff,
								  uint8_t* entField, uint32_t* entTf, uint32_t* entFirstPos, uint32_t* rangeOff) {
	return guarded([&] {
		PositionPostings pp;
		FlatPostings fp;
		std::vector<uint32_t> ro;
		static_cast<const GpuFtMerger*>(h)->GetWord(wordId, pp, fp, ro);
		sizes[0] = pp.doc.size();
		sizes[1] = pp.fpos.size();
		sizes[2] = fp.entField.size();
		sizes[3] = ro.size();
		if (!doc) return;
		std::copy(pp.doc.begin(), pp.doc.end(), doc);
		std::copy(pp.posOff.begin(), pp.posOff.end(), posOff);
		std::copy(pp.fpos.begin(), pp.fpos.end(), fpos);
		std::copy(fp.entOff.begin(), fp.entOff.end(), entOff);
		std::copy(fp.entField.begin(), fp.entField.end(), entField);
		std::copy(fp.entTf.begin(), fp.entTf.end(), entTf);
		std::copy(fp.entFirstPos.begin(), fp.entFirstPos.end(), entFirstPos);
		std::copy(ro.begin(), ro.end(), rangeOff);
	});
}

// ---------------------------------------------------------------------------------------------- hybrid rank fusion
#include "hybrid_rerank.h"

extern "C" {

void rxhost_rrf_positions(const float* ranks, size_t n, uint64_t* out) {
	std::vector<float> r(ranks, ranks + n);
	auto p = InitRRFPositions(r);
	for (size_t i = 0; i < n; ++i) out[i] = p[i];
}
// kind 0 = RRF (params[0] = rank_const), 1 = linear (params = kKnn, knnDefault, kFt, ftDefault, c). Returns count.
long rxhost_merge_ranked(int kind, const double* params, int isUnion, int desc, int metric, const int32_t* knnIds, const float* knnRanks, size_t nKnn,
						 const int32_t* ftIds, const float* ftRanks, size_t nFt, int32_t* outIds, float* outRanks, size_t cap) {
	long n = -1;
	guarded([&] {
		std::vector<int32_t> ki(knnIds, knnIds + nKnn), fi(ftIds, ftIds + nFt);
		std::vector<float> kr(knnRanks, knnRanks + nKnn), fr(ftRanks, ftRanks + nFt);
		HybridResult res;
		const auto type = isUnion ? HybridMergeType::Union : HybridMergeType::Intersection;
		if (kind == 0) {
			// FT positions follow the FT result order (rank-sorted, ties in the given order), then are re-indexed by ascending id like ftIds_.
			// Stable descending sort through one 64-bit key per entry: inverted order-preserving image of the rank, then the index.
			std::vector<uint64_t> keys(nFt);
			std::vector<uint32_t> idx(nFt);
			for (size_t i = 0; i < nFt; ++i) {
				keys[i] = (uint64_t(~detail::SortableBits(fr[i])) << 32) | uint32_t(i);
				idx[i] = uint32_t(i);
			}
			detail::RadixSortPairs(keys, idx);
			std::vector<float> sorted(nFt);
			for (size_t i = 0; i < nFt; ++i) sorted[i] = fr[idx[i]];
			auto posSorted = InitRRFPositions(sorted);
			std::vector<size_t> pos(nFt);
			for (size_t i = 0; i < nFt; ++i) pos[idx[i]] = posSorted[i];
			res = MergeRankedRRF(RerankerRRF{params[0]}, type, desc != 0, VectorMetric(metric), ki, kr, fi, pos);
		} else {
			res = MergeRankedLinear(RerankerLinear{params[0], params[1], params[2], params[3], params[4]}, type, desc != 0, ki, kr, fi, fr);
		}
		n = long(res.ids.size());
		for (size_t i = 0; i < res.ids.size() && i < cap; ++i) {
			outIds[i] = res.ids[i];
			outRanks[i] = res.ranks[i];
		}
	});
	return n;
}

// The same fusion fed with the FT result exactly as the engine returns it (FT result order, best rank first): the id-ascending view and the
// RRF positions are derived here (PrepareFtById), so the caller needs no sort of its own.
long rxhost_merge_ranked_ft_order(int kind, const double* params, int isUnion, int desc, int metric, const int32_t* knnIds, const float* knnRanks,
								  size_t nKnn, const int32_t* ftIds, const float* ftRanks, size_t nFt, int32_t* outIds, float* outRanks, size_t cap) {
	long n = -1;
	guarded([&] {
		std::vector<int32_t> ki(knnIds, knnIds + nKnn), fi(ftIds, ftIds + nFt);
		std::vector<float> kr(knnRanks, knnRanks + nKnn), fr(ftRanks, ftRanks + nFt);
		const FtById ft = PrepareFtById(fi, fr);
		for (size_t i = 1; i < ft.ids.size(); ++i) {
			if (ft.ids[i - 1] == ft.ids[i]) throw std::invalid_argument("merge_ranked: duplicate id in the FT result");
		}
		const auto type = isUnion ? HybridMergeType::Union : HybridMergeType::Intersection;
		HybridResult res = kind == 0 ? MergeRankedRRF(RerankerRRF{params[0]}, type, desc != 0, VectorMetric(metric), ki, kr, ft.ids, ft.positions)
									 : MergeRankedLinear(RerankerLinear{params[0], params[1], params[2], params[3], params[4]}, type, desc != 0, ki, kr,
														 ft.ids, ft.ranks);
		n = long(res.ids.size());
		for (size_t i = 0; i < res.ids.size() && i < cap; ++i) {
			outIds[i] = res.ids[i];
			outRanks[i] = res.ranks[i];
		}
	});
	return n;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------- GpuIvfFlat
#include "gpu_ivf_flat.h"

extern "C" {

void* rxhost_ivf_create(int metric, size_t dim, size_t nlist, int device) {
	GpuIvfFlat* m = nullptr;
	guarded([&] { m = new GpuIvfFlat(VectorMetric(metric), dim, nlist, device); });
	return m;
}
void rxhost_ivf_destroy(void* h) { delete static_cast<GpuIvfFlat*>(h); }
int rxhost_ivf_add(void* h, const float* x, size_t n, const int64_t* ids) {
	return guarded([&] { static_cast<GpuIvfFlat*>(h)->AddWithIds(n, x, ids); });
}
int rxhost_ivf_train(void* h, int seed) {
	return guarded([&] { static_cast<GpuIvfFlat*>(h)->Train(seed); });
}
long rxhost_ivf_remove(void* h, const int64_t* ids, size_t n) {
	long removed = -1;
	guarded([&] { removed = long(static_cast<GpuIvfFlat*>(h)->RemoveIds(ids, n)); });
	return removed;
}
int rxhost_ivf_reset(void* h) {
	return guarded([&] { static_cast<GpuIvfFlat*>(h)->Reset(); });
}
int rxhost_ivf_search(void* h, const float* x, size_t k, size_t nprobe, float* dist, int64_t* labels) {
	return guarded([&] { static_cast<const GpuIvfFlat*>(h)->Search(x, k, nprobe, dist, labels); });
}
int rxhost_ivf_search_batch(void* h, size_t n, const float* x, size_t k, size_t nprobe, float* dist, int64_t* labels) {
	return guarded([&] { static_cast<const GpuIvfFlat*>(h)->SearchBatch(n, x, k, nprobe, dist, labels); });
}
// returns the number of hits (the first min(hits, cap) are written), or -1
long rxhost_ivf_range(void* h, const float* x, float radius, size_t nprobe, float* dist, int64_t* labels, size_t cap) {
	long n = -1;
	guarded([&] {
		std::vector<float> d;
		std::vector<int64_t> l;
		static_cast<const GpuIvfFlat*>(h)->RangeSearch(x, radius, nprobe, d, l);
		n = long(d.size());
		for (size_t i = 0; i < d.size() && i < cap; ++i) {
			dist[i] = d[i];
			labels[i] = l[i];
		}
	});
	return n;
}
long rxhost_ivf_probed_rows(void* h, const float* x, size_t nprobe, uint32_t* out, size_t cap) {
	long n = -1;
	guarded([&] {
		const auto rows = static_cast<const GpuIvfFlat*>(h)->ProbedRows(x, nprobe);
		n = long(rows.size());
		for (size_t i = 0; i < rows.size() && i < cap; ++i) out[i] = rows[i];
	});
	return n;
}
// info[0] = ntotal, [1] = trained, [2] = nlist, [3] = dim
void rxhost_ivf_info(void* h, int64_t* info) {
	const auto* m = static_cast<const GpuIvfFlat*>(h);
	info[0] = int64_t(m->NTotal());
	info[1] = m->IsTrained() ? 1 : 0;
	info[2] = int64_t(m->NList());
	info[3] = int64_t(m->Dim());
}
int rxhost_ivf_list_sizes(void* h, uint32_t* out) {
	return guarded([&] {
		const auto* m = static_cast<const GpuIvfFlat*>(h);
		for (size_t i = 0; i < m->NList(); ++i) out[i] = uint32_t(m->ListSize(i));
	});
}
int rxhost_ivf_list_ids(void* h, size_t list, int64_t* out) {
	return guarded([&] { static_cast<const GpuIvfFlat*>(h)->ListIds(list, out); });
}
int rxhost_ivf_centroids(void* h, float* out) {
	return guarded([&] {
		const auto& c = static_cast<const GpuIvfFlat*>(h)->Centroids();
		std::memcpy(out, c.data(), c.size() * sizeof(float));
	});
}

}  // extern "C"

// test hook for sorted_union.h: runs given as one concatenated array + offsets; strategy 0 = cost model, 1 = merge, 2 = bitmap
#include "sorted_union.h"
extern "C" long rxhost_sorted_union(const uint32_t* rows, const uint64_t* off, size_t nruns, size_t universe, int strategy, uint32_t* out, size_t cap) {
	long n = -1;
	guarded([&] {
		std::vector<std::vector<uint32_t>> runs(nruns);
		std::vector<const std::vector<uint32_t>*> ptrs;
		size_t total = 0;
		for (size_t i = 0; i < nruns; ++i) {
			runs[i].assign(rows + off[i], rows + off[i + 1]);
			ptrs.push_back(&runs[i]);
			total += runs[i].size();
		}
		const auto u = strategy == 1 ? SortedUnionByMerge(ptrs) : strategy == 2 ? SortedUnionByBitmap(ptrs, universe, total) : SortedUnion(ptrs, universe);
		n = long(u.size());
		for (size_t i = 0; i < u.size() && i < cap; ++i) out[i] = u[i];
	});
	return n;
}
