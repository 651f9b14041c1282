// TEST INFRASTRUCTURE ONLY — never linked into the product library.
//
// The VECTOR half of the drop-in boundary, EXECUTED (the FT half: ref_ft_seam_shim.cc).  This TU #includes the reference's
//   cpp_src/core/index/float_vector/hnsw_index.cc
// as integration/patches/0001-hnsw_index-gpu-maps.patch leaves it (applied to a scratch mirror by oracle/Makefile; nothing under
// /root/reference is touched) and builds four instantiations of the reference's own adapter side by side:
//   HnswIndexBase<hnswlib::BruteforceSearch>          the reference's brute-force engine            (kind 0)
//   HnswIndexBase<rxgpu::host::GpuBruteforceMapInTree> the MI355X brute-force Map behind the seam    (kind 1)
//   HnswIndexBase<HierarchicalNSWST>                   the reference's HNSW engine                   (kind 2)
//   HnswIndexBase<rxgpu::host::GpuHnswMapST>           the MI355X HNSW Map behind the seam           (kind 3)
// and drives the members of FloatVectorIndex the planner reaches — upsert (hnsw_index.cc:88-97), del (:116-123), select / selectRaw
// (:159-288), beginStreaming / continueStreaming (:318-361) — so that tests/test_gpu_knn_seam.py can compare what the PATCHED REFERENCE CODE
// returns over the GPU Maps with what it returns over its own engines on the same rows: IdSet, ranks, tie order, array de-duplication, the
// k + radius truncation.  The Maps are the product's sources compiled with RXGPU_IN_TREE (against the reference's FloatVectorId /
// ConstFloatVectorView / SearchResultQueue), exactly as INTEGRATION.md builds them inside cpp_src.
// The index objects are NOT constructed through Index's constructor (that would pull the namespace / payload machinery in): with
// -fno-access-control the members the path reads — metric_, opts_, map_ — are built in place in zeroed storage and the members are called
// non-virtually (the recipe of ref_select_shim.cc).  Every symbol the TU references but this path never calls becomes a trap stub.
// Output: oracle/_ref/libref_knn_seam.so.
#include "core/index/float_vector/hnsw_index.cc"

#include <dlfcn.h>
#include <signal.h>
#include <unistd.h>

#include <cstring>
#include <new>
#include <string>

#include "core/nsselecter/ranks_holder.h"

namespace {
using namespace reindexer;

std::string g_error;

struct Engine {
	virtual ~Engine() = default;
	virtual void Upsert(const float* vec, uint64_t label) = 0;
	virtual void Del(uint64_t label) = 0;
	virtual size_t Count() const = 0;
	virtual long Select(const float* key, const KnnSearchParams& params, int needSort, int32_t* outIds, float* outRanks, size_t cap) = 0;
	virtual long SelectRaw(const float* key, const KnnSearchParams& params, int32_t* outIds, float* outRanks, size_t cap) = 0;
	virtual void BeginStreaming(const float* key, size_t ef) = 0;
	virtual long ContinueStreaming(size_t batch, int32_t* outIds, float* outRanks, int* exhausted) = 0;
};

template <typename Map>
struct EngineT final : Engine {
	using Idx = HnswIndexBase<Map>;
	void* mem = nullptr;
	Idx* ix = nullptr;
	size_t dim = 0;
	std::unique_ptr<KnnStreamingSession> session;   // (no default constructor: held by pointer)

	template <typename... MapArgs>
	EngineT(VectorMetric metric, size_t dim_, bool isArray, MapArgs&&... mapArgs) : dim(dim_) {
		mem = ::operator new(sizeof(Idx), std::align_val_t(alignof(Idx)));
		std::memset(mem, 0, sizeof(Idx));
		ix = reinterpret_cast<Idx*>(mem);
		ix->metric_ = metric;
		// opts_ stays the zeroed storage (IndexOpts' constructor lives in a TU this library does not have): `options` is all the path reads
		if (isArray) ix->opts_.options |= kIndexOptArray;
		new (&ix->map_) Map(std::forward<MapArgs>(mapArgs)...);
	}
	~EngineT() override {
		session.reset();
		ix->map_.~Map();
		::operator delete(mem, std::align_val_t(alignof(Idx)));
	}
	ConstFloatVectorView view(const float* v) const { return ConstFloatVectorView(std::span<const float>(v, dim)); }
	// HnswIndexBase<Map>::upsert (hnsw_index.cc:88-97) up to the Variant it returns (keyvalue machinery this library does not link):
	// grow when full, AddPointNoLock
	void Upsert(const float* vec, uint64_t label) override {
		if (auto cur = ix->map_.MaxElements(); ix->map_.CurrentElementCount() >= cur) ix->map_.ResizeIndex(Idx::newSize(cur));
		ix->map_.AddPointNoLock(view(vec), FloatVectorId::FromNumber(label));
	}
	void Del(uint64_t label) override { ix->Idx::del(FloatVectorId::FromNumber(label), MustExist_True); }
	size_t Count() const override { return ix->map_.CurrentElementCount(); }
	long Select(const float* key, const KnnSearchParams& params, int needSort, int32_t* outIds, float* outRanks, size_t cap) override {
		auto ranks = make_intrusive<RanksHolder>();
		KnnCtx ctx{ranks};
		ctx.NeedSort(needSort ? NeedSort_True : NeedSort_False);
		SelectKeyResult res = ix->Idx::select(view(key), params, ctx);
		size_t n = 0;
		for (const auto& single : res) {
			for (const IdType id : single.flatIds_.view) {
				if (n < cap) outIds[n] = int32_t(id.ToNumber());
				++n;
			}
		}
		const auto span = ranks->GetRanksSpan();
		for (size_t i = 0; i < span.size() && i < cap; ++i) outRanks[i] = span[i].Value();
		return span.size() == n ? long(n) : -1;
	}
	long SelectRaw(const float* key, const KnnSearchParams& params, int32_t* outIds, float* outRanks, size_t cap) override {
		KnnRawResult raw = ix->Idx::selectRaw(view(key), params);
		auto& r = std::get<HnswKnnRawResult>(raw.AsVariant());
		const size_t n = r.Ids().size();
		for (size_t i = 0; i < n && i < cap; ++i) {
			outIds[i] = int32_t(r.Ids()[i].ToNumber());
			outRanks[i] = r.Dists()[i].Value();
		}
		return long(n);
	}
	void BeginStreaming(const float* key, size_t ef) override { session = std::make_unique<KnnStreamingSession>(ix->Idx::beginStreaming(view(key), ef)); }
	long ContinueStreaming(size_t batch, int32_t* outIds, float* outRanks, int* exhausted) override {
		KnnStreamingBatch out;
		ix->Idx::continueStreaming(*session, batch, out);
		for (size_t i = 0; i < out.ids.size(); ++i) {
			outIds[i] = int32_t(out.ids[i].ToNumber());
			outRanks[i] = out.ranks[i].Value();
		}
		*exhausted = out.exhausted ? 1 : 0;
		return long(out.ids.size());
	}
};

KnnSearchParams makeParams(int hnsw, long k, int hasRadius, float radius, size_t ef) {
	if (hnsw) {
		HnswSearchParams p;
		if (k >= 0) p.K(size_t(k));
		if (hasRadius) p.Radius(radius);
		p.Ef(ef);
		return KnnSearchParams{p};
	}
	BruteForceSearchParams p;
	if (k >= 0) p.K(size_t(k));
	if (hasRadius) p.Radius(radius);
	return KnnSearchParams{p};
}

template <typename F>
long guarded(F&& f) {
	try {
		return f();
	} catch (const Error& e) {
		g_error = std::string(e.what());
	} catch (const std::exception& e) {
		g_error = e.what();
	}
	return -2;
}
}  // namespace

// a trap stub was reached: say which symbol (the stubs keep their names) instead of dying silently
static void onTrap(int, siginfo_t* si, void*) {
	Dl_info info{};
	const char* name = (dladdr(si->si_addr, &info) && info.dli_sname) ? info.dli_sname : "?";
	const char msg[] = "libref_knn_seam.so: trap stub reached: ";
	(void)!write(2, msg, sizeof(msg) - 1);
	(void)!write(2, name, strlen(name));
	(void)!write(2, "\n", 1);
	_exit(132);
}

extern "C" {

const char* ref_knn_seam_error() { return g_error.c_str(); }

// kind: 0 reference brute force, 1 GPU brute force, 2 reference HNSW (single-thread build), 3 GPU HNSW.  The GPU Maps take their device list
// from RX_GPU_VECTOR_INDEXES like the reference's factory would (rx_seam.h; unset: device 0).
void* ref_knn_seam_create(int kind, int metric, size_t dim, size_t maxElements, int isArray, size_t M, size_t efConstruction) {
	struct sigaction sa {};
	sa.sa_sigaction = onTrap;
	sa.sa_flags = SA_SIGINFO;
	sigaction(SIGILL, &sa, nullptr);
	Engine* e = nullptr;
	const long rc = guarded([&]() -> long {
		const VectorMetric m = VectorMetric(metric);
		const IsArray arr = isArray ? IsArray_True : IsArray_False;
		switch (kind) {
			case 0: e = new EngineT<hnswlib::BruteforceSearch>(m, dim, isArray != 0, m, dim, maxElements); break;
			case 1: e = new EngineT<GpuBruteforce>(m, dim, isArray != 0, m, dim, maxElements); break;
			case 2: e = new EngineT<HierarchicalNSWST>(m, dim, isArray != 0, arr, m, dim, maxElements, M, efConstruction); break;
			case 3: e = new EngineT<GpuHnswST>(m, dim, isArray != 0, arr, m, dim, maxElements, M, efConstruction); break;
			default: g_error = "no such engine kind"; return -2;
		}
		return 0;
	});
	return rc == 0 ? e : nullptr;
}
void ref_knn_seam_destroy(void* h) { delete static_cast<Engine*>(h); }

long ref_knn_seam_upsert(void* h, const float* vecs, size_t n, size_t dim, const uint64_t* labels) {
	return guarded([&]() -> long {
		for (size_t i = 0; i < n; ++i) static_cast<Engine*>(h)->Upsert(vecs + i * dim, labels[i]);
		return long(n);
	});
}
long ref_knn_seam_del(void* h, uint64_t label) {
	return guarded([&]() -> long {
		static_cast<Engine*>(h)->Del(label);
		return 0;
	});
}
long ref_knn_seam_count(void* h) { return long(static_cast<Engine*>(h)->Count()); }

// select(): ids + ranks as the planner receives them.  k < 0: no K.  Returns the count (-1: ids and ranks disagree, -2: see ref_knn_seam_error)
long ref_knn_seam_select(void* h, int hnsw, const float* key, long k, int hasRadius, float radius, size_t ef, int needSort, int32_t* outIds, float* outRanks,
						 size_t cap) {
	return guarded([&]() -> long { return static_cast<Engine*>(h)->Select(key, makeParams(hnsw, k, hasRadius, radius, ef), needSort, outIds, outRanks, cap); });
}
long ref_knn_seam_select_raw(void* h, int hnsw, const float* key, long k, int hasRadius, float radius, size_t ef, int32_t* outIds, float* outRanks, size_t cap) {
	return guarded([&]() -> long { return static_cast<Engine*>(h)->SelectRaw(key, makeParams(hnsw, k, hasRadius, radius, ef), outIds, outRanks, cap); });
}
long ref_knn_seam_begin_streaming(void* h, const float* key, size_t ef) {
	return guarded([&]() -> long {
		static_cast<Engine*>(h)->BeginStreaming(key, ef);
		return 0;
	});
}
long ref_knn_seam_continue_streaming(void* h, size_t batch, int32_t* outIds, float* outRanks, int* exhausted) {
	return guarded([&]() -> long { return static_cast<Engine*>(h)->ContinueStreaming(batch, outIds, outRanks, exhausted); });
}
}
