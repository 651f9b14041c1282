// TEST INFRASTRUCTURE ONLY — never linked into the product library.
//
// The reference's KNN select post-processing, compiled in place: this TU #includes
//   /root/reference/cpp_src/core/index/float_vector/hnsw_index.cc
// and drives HnswIndexBase<hnswlib::BruteforceSearch>::select / selectRaw (hnsw_index.cc:160-288: search -> queue drain with sign flip,
// the equal-distance id sort, removeDuplicateRowId for array fields, removeOverK) over the reference's own BruteforceSearch map.
// The index object is NOT constructed through Index's constructor (that would pull the namespace / payload machinery in): with
// -fno-access-control the three members the path reads — metric_, opts_, map_ — are built in place in zeroed storage and select() is called
// non-virtually.  Every symbol the TU references but this path never calls becomes a trap stub (same recipe as libref_ft.so / libref_rank.so).
// Output: oracle/_ref/libref_select.so.
#include "core/index/float_vector/hnsw_index.cc"

#include <dlfcn.h>
#include <signal.h>
#include <unistd.h>

#include <cstring>
#include <new>

#include "core/nsselecter/ranks_holder.h"

namespace {
using namespace reindexer;
using Idx = HnswIndexBase<hnswlib::BruteforceSearch>;

struct Holder {
	void* mem = nullptr;
	Idx* ix = nullptr;
	VectorMetric metric;
	size_t dim;
};
}  // namespace

// a trap stub was reached: say which symbol (the stubs keep their names) instead of dying silently
static void onTrap(int, siginfo_t* si, void*) {
	Dl_info info{};
	const char* name = (dladdr(si->si_addr, &info) && info.dli_sname) ? info.dli_sname : "?";
	const char msg[] = "libref_select.so: trap stub reached: ";
	(void)!write(2, msg, sizeof(msg) - 1);
	(void)!write(2, name, strlen(name));
	(void)!write(2, "\n", 1);
	_exit(132);
}

extern "C" {

void* ref_select_create(int metric, size_t dim, size_t maxElements, int isArray, int hasIndexRadius, float indexRadius) {
	struct sigaction sa {};
	sa.sa_sigaction = onTrap;
	sa.sa_flags = SA_SIGINFO;
	sigaction(SIGILL, &sa, nullptr);
	auto* h = new Holder();
	h->mem = ::operator new(sizeof(Idx), std::align_val_t(alignof(Idx)));
	std::memset(h->mem, 0, sizeof(Idx));
	h->ix = reinterpret_cast<Idx*>(h->mem);
	h->metric = VectorMetric(metric);
	h->dim = dim;
	h->ix->metric_ = h->metric;
	// opts_ stays the zeroed storage (IndexOpts' constructor lives in a TU this library does not have): `options` and the empty optional
	// floatVector_ are all the path reads; the never-touched std::string config_ is simply not destroyed
	if (isArray) h->ix->opts_.options |= kIndexOptArray;
	if (hasIndexRadius) {
		h->ix->opts_.floatVector_.emplace();
		h->ix->opts_.floatVector_->SetRadius(indexRadius);
	}
	new (&h->ix->map_) hnswlib::BruteforceSearch(h->metric, dim, maxElements);
	return h;
}

void ref_select_destroy(void* hp) {
	auto* h = static_cast<Holder*>(hp);
	h->ix->map_.~BruteforceSearch();
	::operator delete(h->mem, std::align_val_t(alignof(Idx)));
	delete h;
}

// label = FloatVectorId number: (rowId << 32) | arrayIndex as the product packs it
void ref_select_add(void* hp, const float* vecs, size_t n, const uint64_t* labels) {
	auto* h = static_cast<Holder*>(hp);
	for (size_t i = 0; i < n; ++i) {
		h->ix->map_.AddPointNoLock(ConstFloatVectorView(std::span<const float>(vecs + i * h->dim, h->dim)), FloatVectorId::FromNumber(labels[i]));
	}
}

// select(): ids + ranks as the planner receives them.  k < 0: no K.  Returns the count.
long ref_select(void* hp, const float* key, long k, int hasRadius, float radius, int needSort, int32_t* outIds, float* outRanks, size_t cap) {
	auto* h = static_cast<Holder*>(hp);
	BruteForceSearchParams bf;
	if (k >= 0) bf.K(size_t(k));
	if (hasRadius) bf.Radius(radius);
	KnnSearchParams params{bf};
	auto ranks = make_intrusive<RanksHolder>();
	KnnCtx ctx{ranks};
	ctx.NeedSort(needSort ? NeedSort_True : NeedSort_False);
	SelectKeyResult res = h->ix->Idx::select(ConstFloatVectorView(std::span<const float>(key, h->dim)), params, ctx);
	size_t n = 0;
	for (const auto& single : res) {
		for (const IdType id : single.flatIds_.view) {   // select() returns one plain id set (SingleSelectKeyResult(IdSetPlain::Ptr&&))
			if (n < cap) outIds[n] = int32_t(id.ToNumber());
			++n;
		}
	}
	const auto span = ranks->GetRanksSpan();
	for (size_t i = 0; i < span.size() && i < cap; ++i) outRanks[i] = span[i].Value();
	return span.size() == n ? long(n) : -1;
}

// selectRaw(): the hybrid path's raw result (no id sort)
long ref_select_raw(void* hp, const float* key, long k, int hasRadius, float radius, int32_t* outIds, float* outRanks, size_t cap) {
	auto* h = static_cast<Holder*>(hp);
	BruteForceSearchParams bf;
	if (k >= 0) bf.K(size_t(k));
	if (hasRadius) bf.Radius(radius);
	KnnSearchParams params{bf};
	KnnRawResult raw = h->ix->Idx::selectRaw(ConstFloatVectorView(std::span<const float>(key, h->dim)), params);
	auto& r = std::get<HnswKnnRawResult>(raw.AsVariant());
	const size_t n = r.Ids().size();
	for (size_t i = 0; i < n && i < cap; ++i) {
		outIds[i] = int32_t(r.Ids()[i].ToNumber());
		outRanks[i] = r.Dists()[i].Value();
	}
	return long(n);
}
}
